// mhx_emcee_kernels.h -- affine-invariant ensemble sampler (Goodman & Weare stretch move), one
// wavefront lane per walker.
//
// Replaces Ensemble{StretchProposal}: the sweep `propose` (src/emcee.jl:39-58) and `move`
// (src/emcee.jl:70-102).  The reference updates walkers one after another (Gauss-Seidel: walker i
// may pair with an already-updated walker).  That loop cannot run in parallel, so the device runs
// the standard parallel form (Foreman-Mackey et al. 2013): the ensemble is split into halves
// [0, W/2) and [W/2, W); every walker of one half moves at once, pairing with a uniformly chosen
// walker of the OTHER half (whose positions are frozen during that half-step); then the roles
// swap.  Same stationary distribution, different Markov kernel -- the oracle implements both
// (mode 0 sequential, mode 1 split) and the HIP kernel is bit-exact against mode 1.
//
// One launch = one half-step; consecutive launches on the stream order the halves.  The ABI layout of
// the walkers is [dim][W] (walker fastest) and the run-time-dimension kernel works on it directly (own
// accesses coalesced, the partner gather a scattered 4-byte read per dimension).  The compile-time-
// dimension kernels -- lane per walker in registers, and the cooperative one below -- keep the walkers in
// a walker-major copy instead ([W][round4(dim)], refreshed from / written back to the ABI layout by the
// host side): a walker and its partner are then one contiguous row each (4 cache lines at d = 50
// instead of 50 per partner), read and written as float4.
#pragma once
#include "mhx_targets.h"

MHX_NS_BEGIN

struct mhx_emcee_args {
    mhx_real* x;                 // [dim][W]   (ABI layout; the lane-per-walker kernels work on it)
    mhx_real* xw;                // [W][round4(dim)] walker-major copy, zero padded: the state of the cooperative kernel
    mhx_real* lp;                // [W]
    mhx_real* xw_out;            // the sweep kernels (one launch per sweep) read xw / lp and write the new state here
    mhx_real* lp_out;
    int all_rows;                // sweep kernels: 1 = every walker's row goes to xw_out (the first launch after the state was set from
                                 // outside); 0 = only the rows xw_out does not hold already (see mhx_emcee_coop_sweep_body)
    mhx_u32* acc_count;       // [W]
    mhx_u64* acc_total;
    mhx_real* samples;           // [slots][dim+1][W] or null
    unsigned char* accepted;  // [slots][W] or null
    unsigned char* last_acc;  // [W]
    mhx_real* ybuf;              // [dim][W] candidate scratch (run-time-dimension kernel)
    mhx_u64 seed;
    mhx_u64 ensemble_id;
    int nwalkers;
    int dim;
    int target_kind;
    int ntparams;
    mhx_real tconst;
    mhx_real stretch;            // a
    mhx_u32 sweep;            // RNG step counter of this sweep
    int half;                 // 0: walkers [0, W/2) move; 1: walkers [W/2, W) move
    long save_slot;           // slot to record this sweep into, or -1
    long rec_other_slot;      // cooperative kernel: slot into which THIS launch records the half that is NOT moving (at rest since the
                              // previous half-step, hence final for its sweep), or -1.  The record of a half then leaves at the START of the
                              // next launch, off its critical path, instead of at the end of the launch that moved it (host: emcee_advance)
    int reduce_lanes;         // lanes per walker (cooperative kernel), >= 1
    int t_begin, t_count;     // the slice of the moving half this launch moves (an ensemble sharded over GPUs moves
                              // one slice per rank and exchanges the slices; a single GPU moves [0, size of the half))
    // initial walkers drawn on the device (src/emcee.jl:29-34: W draws from the wrapped prior, here (Mv)Normal): x_i = mu + L z
    int init_kind;            // MHX_PROP_ISO / DIAG / DENSE
    mhx_real init_scale;
    const mhx_real* init_vec;    // DIAG: sigma[dim]; DENSE: chol(Sigma) packed lower
    const mhx_real* init_mean;   // mu[dim] or null
    // the reference's sequential sweep (one launch = the whole schedule, one wave): see mhx_emcee_seq_body
    int nsweeps;              // sweeps of this launch
    mhx_u32 save_next;        // first sweep whose state is recorded (0xffffffff: none), then every `thinning`-th
    int thinning;
    // MANY ensembles in one launch (mhx_emcee_cfg.n_ensembles; README.md:135-148: `sample(model, Ensemble(..), MCMCThreads(), N,
    // nchains)` runs nchains independent ensembles): blockIdx.y is the ensemble.  Ensemble e's walkers are columns e W .. e W + W - 1
    // of every [..][E W] array (x, samples, accepted: leading dimension ld = E W), rows e W .. of the walker-major state, and its
    // Philox counters carry ensemble_id + e -- the kernels see ONE ensemble through mhx_emcee_pick.
    int ld;                   // E * nwalkers
    int ybuf_ens;             // reals between two ensembles' pieces of ybuf
};

// Pitch (in reals) of a walker's row in the walker-major state: round4(dim) reals.  MHX_XW_LINE = 128 (a build-time knob) rounds rows
// of at least a cache line up to whole 128-byte lines, so that a rewritten row is rewritten as FULL lines -- measured on C3's sweep
// kernel: 8.33 us per sweep against 8.07 with the tight pitch (416-byte rows in fp64: the extra lines cost more than the partial
// ones), so the tight pitch stays.
#ifndef MHX_XW_LINE
#define MHX_XW_LINE 0
#endif
MHX_HD constexpr int mhx_xw_pitch(const int d)
{
    const int xp = (d + 3) & ~3, rb = xp * (int)sizeof(mhx_real);
    return (MHX_XW_LINE > 0 && rb >= MHX_XW_LINE) ? ((rb + MHX_XW_LINE - 1) / MHX_XW_LINE) * MHX_XW_LINE / (int)sizeof(mhx_real) : xp;
}

#ifndef MHX_PROP_ISO
#define MHX_PROP_ISO   0
#define MHX_PROP_DIAG  1
#define MHX_PROP_DENSE 2
#endif

// the argument block of ensemble blockIdx.y (wave-uniform: scalar arithmetic on the kernel arguments)
MHX_DEV mhx_emcee_args mhx_emcee_pick(const mhx_emcee_args& a)
{
    mhx_emcee_args b = a;
    const long e = (long)blockIdx.y;
    if (e) {
        const long w = e * (long)a.nwalkers, wp = w * mhx_xw_pitch(a.dim);
        b.x += w;
        if (a.xw) b.xw += wp;
        if (a.xw_out) b.xw_out += wp;
        b.lp += w;
        if (a.lp_out) b.lp_out += w;
        b.acc_count += w;
        b.last_acc += w;
        if (a.samples) { b.samples += w; b.accepted += w; }
        if (a.ybuf) b.ybuf += e * (long)a.ybuf_ens;
        b.ensemble_id += (mhx_u64)e;
    }
    return b;
}

typedef mhx_real mhx_e4 __attribute__((ext_vector_type(4)));

// D > 0: compile-time dimension, candidate in registers; the walkers are read from the walker-major copy
// [W][round4(D)] (a.xw): a walker and its partner are one contiguous row each, float4 loads, 4 cache lines
// at d = 50 where the [dim][W] layout touches 50 per partner.  D == 0: run-time dimension, [dim][W] state,
// candidate staged in ybuf.
template <int D, int TK>
MHX_DEV void mhx_emcee_half_body(const mhx_emcee_args& a, const mhx_real* __restrict__ tparams)
{
    const int W = a.nwalkers;
    const int halfW = W / 2;
    const int lo = a.half ? halfW : 0;
    const int cnt = a.half ? W - halfW : halfW;
    const int t = a.t_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt || t >= a.t_begin + a.t_count) return;
    const int i = lo + t;
    const int ostart = a.half ? 0 : halfW;
    const int osize = a.half ? halfW : W - halfW;
    const int d = D > 0 ? D : a.dim;
    const long ld = a.ld;

    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep);
    // partner from the complementary half (src/emcee.jl:48,52 draws from all other walkers)
    const int j = ostart + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)osize) >> 32);
    // src/emcee.jl:81  z = ((a - 1) * rand(rng) + 1)^2 / a
    const mhx_real u = dr.u;
    const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), u, MHX_R(1.0));
    const mhx_real z = (tt * tt) / a.stretch;
    const mhx_real alphamult = (mhx_real)(d - 1) * mhx_log(z);                // :82

    mhx_real lpy;
    constexpr int XP = D > 0 ? ((D + 3) & ~3) : 4;
    mhx_real yreg[XP];
    mhx_real* ys = a.ybuf + i;
    mhx_e4* xrow_i = D > 0 ? (mhx_e4*)(a.xw + (long)i * mhx_xw_pitch(D)) : nullptr;
    if (D > 0) {
        const mhx_e4* xrow_j = (const mhx_e4*)(a.xw + (long)j * mhx_xw_pitch(D));
#pragma unroll
        for (int q = 0; q < XP / 4; ++q) {
            const mhx_e4 xi = xrow_i[q], xj = xrow_j[q];                  // the zero pad of the rows stays zero
            yreg[4 * q + 0] = mhx_fma(z, xi.x - xj.x, xj.x);              // :85
            yreg[4 * q + 1] = mhx_fma(z, xi.y - xj.y, xj.y);
            yreg[4 * q + 2] = mhx_fma(z, xi.z - xj.z, xj.z);
            yreg[4 * q + 3] = mhx_fma(z, xi.w - xj.w, xj.w);
        }
        lpy = mhx_target_eval<TK>(TK, yreg, D, tparams, a.ntparams, a.tconst);
    } else {
        for (int k = 0; k < d; ++k) {
            const mhx_real xi = a.x[(long)k * ld + i];
            const mhx_real xj = a.x[(long)k * ld + j];
            ys[(long)k * ld] = mhx_fma(z, xi - xj, xj);
        }
        mhx_strided_x yv;
        yv.base = ys;
        yv.ld = ld;
        lpy = mhx_target_eval<TK>(a.target_kind, yv, d, tparams, a.ntparams, a.tconst);
    }
    const mhx_real lpi = a.lp[i];
    const mhx_real alpha = (alphamult + lpy) - lpi;                        // :91
    const mhx_real logu = dr.logu;
    const bool acc = logu <= alpha;                                     // :93  -randexp <= alpha (non-strict)
    if (acc) {
        if (D > 0) {
#pragma unroll
            for (int q = 0; q < XP / 4; ++q) {
                mhx_e4 v;
                v.x = yreg[4 * q + 0]; v.y = yreg[4 * q + 1]; v.z = yreg[4 * q + 2]; v.w = yreg[4 * q + 3];
                xrow_i[q] = v;
            }
        } else {
            for (int k = 0; k < d; ++k) a.x[(long)k * ld + i] = ys[(long)k * ld];
        }
        a.lp[i] = lpy;
        a.acc_count[i] += 1u;
    }
    a.last_acc[i] = acc ? 1 : 0;
    if (a.save_slot >= 0) {
        mhx_real* row = a.samples + a.save_slot * (long)(d + 1) * ld + i;
        if (D > 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) row[(long)k * ld] = acc ? yreg[k] : a.xw[(long)i * mhx_xw_pitch(D) + k];
        } else {
            for (int k = 0; k < d; ++k) row[(long)k * ld] = a.x[(long)k * ld + i];
        }
        row[(long)d * ld] = acc ? lpy : lpi;
        a.accepted[a.save_slot * ld + i] = acc ? 1 : 0;
    }
    // the all-walker accept count is summed from acc_count by the host after the run (see the
    // cooperative kernel below for why there is no per-wave atomic in these short launches)
}

// The same kernel as ONE LAUNCH PER SWEEP (round 4; the idea and the double-buffered state are described at
// mhx_emcee_coop_sweep_body): blocks [0, nbB) move the second half, whose lanes first re-do their partner's move from the old
// state -- a second evaluation of the log-density (any target, user source included) instead of a second launch -- and then move
// against its result.  Two candidates live in registers at a time; rows that are needed again are re-read (they are in L1 / L2).
template <int D, int TK>
MHX_DEV void mhx_emcee_sweep_reg_body(const mhx_emcee_args& a, const mhx_real* __restrict__ tparams)
{
    static_assert(D > 0, "the sweep kernel works on the walker-major rows of the compile-time-dimension form");
    const int W = a.nwalkers;
    const int halfW = W / 2, cntB = W - halfW;
    const int nbB = (cntB + (int)blockDim.x - 1) / (int)blockDim.x;
    const bool second = (int)blockIdx.x < nbB;                               // (block-uniform)
    const int blk = second ? (int)blockIdx.x : (int)blockIdx.x - nbB;
    const int cnt = second ? cntB : halfW;
    const int t = blk * (int)blockDim.x + (int)threadIdx.x;
    if (t >= cnt) return;
    const int i = (second ? halfW : 0) + t;
    const long ld = a.ld;
    constexpr int XP = (D + 3) & ~3;
    // the walker's own row, lp and flag do not wait for the draws: their loads go out first (left where they are used, they were issued
    // behind the Philox rounds and, in the second half, behind the partner's whole evaluation)
    // (while four rows of registers fit a lane comfortably: fp32 up to d = 64; a d = 50 row in fp64 is 104 VGPRs and holding it through
    // both evaluations costs more than the early load buys -- 11.2 against 9.9 us per sweep -- so there the row is re-read where needed)
    constexpr bool EARLY = XP * (int)(sizeof(mhx_real) / 4) * 4 <= 256;
    mhx_real xi_reg[EARLY ? XP : 1];
    const mhx_real* xi_mem = a.xw + (long)i * mhx_xw_pitch(D);
    if constexpr (EARLY) {
        const mhx_e4* xr = (const mhx_e4*)xi_mem;
#pragma unroll
        for (int q = 0; q < XP / 4; ++q) { const mhx_e4 v = xr[q]; xi_reg[4 * q] = v.x; xi_reg[4 * q + 1] = v.y; xi_reg[4 * q + 2] = v.z; xi_reg[4 * q + 3] = v.w; }
    }
    auto xi = [&](const int k) -> mhx_real { if constexpr (EARLY) return xi_reg[k]; else return xi_mem[k]; };
    const mhx_real lpi = a.lp[i];
    const bool moved_before = a.all_rows != 0 || a.last_acc[i] != 0;         // xw_out does not hold this walker's row
    __builtin_amdgcn_sched_barrier(0);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep);
    const int j = (second ? 0 : halfW) + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)(second ? halfW : cntB)) >> 32);
    const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), dr.u, MHX_R(1.0));
    const mhx_real z = (tt * tt) / a.stretch;                                // src/emcee.jl:81
    const mhx_real alphamult = (mhx_real)(D - 1) * mhx_log(z);               // :82
    const mhx_e4* xrow_j = (const mhx_e4*)(a.xw + (long)j * mhx_xw_pitch(D));
    mhx_real yreg[XP];
    if (!second) {
#pragma unroll
        for (int q = 0; q < XP / 4; ++q) {
            const mhx_e4 xj = xrow_j[q];                                     // the zero pad of the rows stays zero
            yreg[4 * q + 0] = mhx_fma(z, xi(4 * q + 0) - xj.x, xj.x);        // :85
            yreg[4 * q + 1] = mhx_fma(z, xi(4 * q + 1) - xj.y, xj.y);
            yreg[4 * q + 2] = mhx_fma(z, xi(4 * q + 2) - xj.z, xj.z);
            yreg[4 * q + 3] = mhx_fma(z, xi(4 * q + 3) - xj.w, xj.w);
        }
    } else {
        // j is a walker of the first half: its own move of this sweep, from the state this launch found
        const mhx_emcee_draws da = mhx_emcee_draw(ks, (mhx_u32)j, (mhx_u32)a.ensemble_id, a.sweep);
        const int jb = halfW + (int)(((mhx_u64)da.partner * (mhx_u64)(mhx_u32)cntB) >> 32);
        const mhx_e4* xrow_b = (const mhx_e4*)(a.xw + (long)jb * mhx_xw_pitch(D));
        const mhx_real lpa = a.lp[j];
        const mhx_real ta = mhx_fma(a.stretch - MHX_R(1.0), da.u, MHX_R(1.0));
        const mhx_real za = (ta * ta) / a.stretch;
        const mhx_real alphamult_a = (mhx_real)(D - 1) * mhx_log(za);
        mhx_real ya[XP];
#pragma unroll
        for (int q = 0; q < XP / 4; ++q) {
            const mhx_e4 xa = xrow_j[q], xb = xrow_b[q];
            ya[4 * q + 0] = mhx_fma(za, xa.x - xb.x, xb.x);
            ya[4 * q + 1] = mhx_fma(za, xa.y - xb.y, xb.y);
            ya[4 * q + 2] = mhx_fma(za, xa.z - xb.z, xb.z);
            ya[4 * q + 3] = mhx_fma(za, xa.w - xb.w, xb.w);
        }
        const mhx_real lpya = mhx_target_eval<TK>(TK, ya, D, tparams, a.ntparams, a.tconst);
        const bool acc_a = da.logu <= (alphamult_a + lpya) - lpa;            // the partner's accept test (:91-93), as its own lane runs it
#pragma unroll
        for (int q = 0; q < XP / 4; ++q) {
            const mhx_e4 xa = xrow_j[q];
            const mhx_real p0 = acc_a ? ya[4 * q + 0] : xa.x, p1 = acc_a ? ya[4 * q + 1] : xa.y;
            const mhx_real p2 = acc_a ? ya[4 * q + 2] : xa.z, p3 = acc_a ? ya[4 * q + 3] : xa.w;
            yreg[4 * q + 0] = mhx_fma(z, xi(4 * q + 0) - p0, p0);
            yreg[4 * q + 1] = mhx_fma(z, xi(4 * q + 1) - p1, p1);
            yreg[4 * q + 2] = mhx_fma(z, xi(4 * q + 2) - p2, p2);
            yreg[4 * q + 3] = mhx_fma(z, xi(4 * q + 3) - p3, p3);
        }
    }
    const mhx_real lpy = mhx_target_eval<TK>(TK, yreg, D, tparams, a.ntparams, a.tconst);
    const mhx_real alpha = (alphamult + lpy) - lpi;                          // :91
    const bool acc = dr.logu <= alpha;                                       // :93
    if (!acc) {
#pragma unroll
        for (int k = 0; k < XP; ++k) yreg[k] = xi(k);
    }
    if (acc || moved_before) {
        mhx_e4* xrow_o = (mhx_e4*)(a.xw_out + (long)i * mhx_xw_pitch(D));
#pragma unroll
        for (int q = 0; q < XP / 4; ++q) {
            mhx_e4 v;
            v.x = yreg[4 * q + 0]; v.y = yreg[4 * q + 1]; v.z = yreg[4 * q + 2]; v.w = yreg[4 * q + 3];
            xrow_o[q] = v;
        }
    }
    a.lp_out[i] = acc ? lpy : lpi;
    if (acc) a.acc_count[i] += 1u;
    a.last_acc[i] = acc ? 1 : 0;
    if (a.save_slot >= 0) {
        mhx_real* row = a.samples + a.save_slot * (long)(D + 1) * ld + i;
#pragma unroll
        for (int k = 0; k < D; ++k) row[(long)k * ld] = yreg[k];
        row[(long)D * ld] = acc ? lpy : lpi;
        a.accepted[a.save_slot * ld + i] = acc ? 1 : 0;
    }
}

// A SMALL ensemble (W <= 1024: the sizes emcee is mostly run at -- a few walkers per dimension) as ONE persistent block: thread t owns
// walker t for the whole call, its row in registers and, for its partners to read, in the block's LDS ([W][XP + 1] reals); the
// half-steps of all the call's sweeps are separated by block barriers instead of kernel boundaries (2.7 us each, of the 3.4-4.9 us
// a sweep launch of such an ensemble takes).  Same draws, same arithmetic (lane per walker, reduction shape 1), same record as the
// sweep / half-step launches.  a.nsweeps sweeps from a.sweep on; a.save_next / a.thinning / a.save_slot as in the sequential form.
template <int D, int TK>
MHX_DEV void mhx_emcee_persist_body(const mhx_emcee_args& a, const mhx_real* __restrict__ tparams, mhx_real* xsh)
{
    static_assert(D > 0, "compile-time dimension");
    constexpr int XP = (D + 3) & ~3;
    constexpr int XS = XP + 1;                                   // LDS row pitch: odd, the walkers' columns spread over the banks
    const int W = a.nwalkers;
    const int halfW = W / 2, cntB = W - halfW;
    const int t = (int)threadIdx.x;
    const bool owner = t < W;
    const int i = owner ? t : W - 1;
    const long ld = a.ld;
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    mhx_real x[XP], y[XP];
    const mhx_real* xrow_g = a.xw + (long)i * mhx_xw_pitch(D);
#pragma unroll
    for (int k = 0; k < XP; ++k) x[k] = xrow_g[k];
    mhx_real* xrow_s = xsh + i * XS;
    if (owner) {
#pragma unroll
        for (int k = 0; k < XP; ++k) xrow_s[k] = x[k];
    }
    mhx_real lp = a.lp[i];
    mhx_u32 nacc = a.acc_count[i];
    bool last = a.last_acc[i] != 0;
    const bool second = t >= halfW;                              // which half this thread's walker is in
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    __syncthreads();
    for (int s = 0; s < a.nsweeps; ++s) {
        const mhx_u32 sweep = a.sweep + (mhx_u32)s;
        // the draws of this walker's move of the sweep do not depend on anything: taken before the barriers, off the critical path
        const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, sweep);
        const int j = (second ? 0 : halfW) + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)(second ? halfW : cntB)) >> 32);
        const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), dr.u, MHX_R(1.0));
        const mhx_real z = (tt * tt) / a.stretch;                            // src/emcee.jl:81
        const mhx_real alphamult = (mhx_real)(D - 1) * mhx_log(z);           // :82
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (owner && (h == 1) == second) {                               // (whole waves except at the seam of the halves)
                const mhx_real* xj = xsh + j * XS;
#pragma unroll
                for (int k = 0; k < XP; ++k) { const mhx_real p = xj[k]; y[k] = mhx_fma(z, x[k] - p, p); }      // :85 (the pad stays zero)
                const mhx_real lpy = mhx_target_eval<TK>(TK, y, D, tparams, a.ntparams, a.tconst);
                const mhx_real alpha = (alphamult + lpy) - lp;               // :91
                const bool acc = dr.logu <= alpha;                           // :93
                if (acc) {
#pragma unroll
                    for (int k = 0; k < XP; ++k) { x[k] = y[k]; xrow_s[k] = y[k]; }
                    lp = lpy;
                    nacc += 1u;
                }
                last = acc;
            }
            __syncthreads();
        }
        if (sweep == save_next) {
            if (owner) {
                mhx_real* row = a.samples + slot * (long)(D + 1) * ld + i;
#pragma unroll
                for (int k = 0; k < D; ++k) row[(long)k * ld] = x[k];
                row[(long)D * ld] = lp;
                a.accepted[slot * ld + i] = last ? 1 : 0;
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    if (owner) {
        mhx_real* xrow_o = a.xw + (long)i * mhx_xw_pitch(D);
#pragma unroll
        for (int k = 0; k < XP; ++k) xrow_o[k] = x[k];
        a.lp[i] = lp;
        a.acc_count[i] = nacc;
        a.last_acc[i] = last ? 1 : 0;
    }
}

// initial walkers (src/emcee.jl:29-34, :6-8): with `draw`, walker i is a draw mu + L z from the wrapped (Mv)Normal prior,
// z from Philox stream INIT of (ensemble, i); then W log-density evaluations, accepted = false
template <int TK>
MHX_DEV void mhx_emcee_init_body(const mhx_emcee_args& a, const mhx_real* __restrict__ tparams, const int draw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nwalkers) return;
    if (draw) {
        const mhx_philox_key ks = mhx_philox_schedule(a.seed);
        const long ld = a.ld;
        const int d = a.dim;
        mhx_real* xs = a.x + i;
        const int nblk = (d + 3) >> 2;
        for (int b = 0; b < nblk; ++b) {
            mhx_real n[4];
            mhx_normal4(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, 0u, MHX_STREAM_INIT, (mhx_u32)b, n);
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k < d) {
                    if (a.init_kind == MHX_PROP_DENSE) xs[(long)k * ld] = n[j];
                    else {
                        const mhx_real sc = a.init_kind == MHX_PROP_ISO ? a.init_scale : a.init_vec[k];
                        xs[(long)k * ld] = a.init_mean ? mhx_fma(sc, n[j], a.init_mean[k]) : mhx_fma(sc, n[j], MHX_R(0.0));
                    }
                }
            }
        }
        if (a.init_kind == MHX_PROP_DENSE) {
            for (int r = d - 1; r >= 0; --r) {                  // rows in descending order: z is overwritten in place
                const mhx_real* Lr = a.init_vec + (long)r * (r + 1) / 2;
                mhx_real w = MHX_R(0.0);
                for (int j = 0; j <= r; ++j) w = mhx_fma(Lr[j], xs[(long)j * ld], w);
                xs[(long)r * ld] = a.init_mean ? a.init_mean[r] + w : w;
            }
        }
    }
    mhx_strided_x xv;
    xv.base = a.x + i;
    xv.ld = a.ld;
    a.lp[i] = mhx_target_eval_lanes<TK>(a.target_kind, xv, a.dim, tparams, a.ntparams, a.tconst, a.reduce_lanes);
    a.acc_count[i] = 0u;
    a.last_acc[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// Cooperative stretch move for the dense-Gaussian target: L lanes share one walker (64/L walkers per
// wave).  A half-step of a 16 384-walker ensemble is only 128 waves of lane-per-walker work, each
// walking the 1275 entries of inv(chol(Sigma)) serially; with L = 16 it is 2048 waves (two per SIMD):
// every lane owns float4 slices of the move and D/L rows of A y.  The candidate is exchanged through
// LDS (broadcast reads within a walker's lane group); the L partial sums of squares meet in an
// xor-butterfly -- the reduction shape is part of the arithmetic spec (oracle: reduce_lanes = L).
// The same image / row-product machinery serves random-walk Metropolis on dense factors
// (mhx_rwmh_dense_kernels.h).
#ifndef MHX_EMCEE_COOP_WAVES
#define MHX_EMCEE_COOP_WAVES 4                   // waves per block: they share the LDS copy of the factor
#endif

// LDS image of the packed factor for L lanes per walker.  Lane l owns rows l, l+L, ...; row set m
// (rows L m .. L m + L - 1) is stored as LEN4(m) groups of L float4 -- group jj4 holds columns
// 4 jj4 .. 4 jj4 + 3 of the L rows, so a lane group reads L consecutive float4 (no bank conflicts) --
// zero-filled above the diagonal: fma(0, y, w) == w, the row dot needs no masks.
template <int D, int L>
struct mhx_emcee_geom {
    static constexpr int NK = (D + L - 1) / L;
    static constexpr int XP = (D + 3) & ~3;                   // pitch of a walker's row in the walker-major state
    static constexpr int NQ = XP / 4;                         // float4 per walker
    static constexpr int NQL = (NQ + 1 + L - 1) / L;          // float4 slots per lane (incl. the zero tail of the y row)
    static constexpr int DP4 = XP + 4;                        // y row pitch: 16-byte aligned, spread over the banks
    MHX_HD static constexpr int len4(int m) { return ((L * (m + 1) < D ? L * (m + 1) : D) + 3) / 4; }
    MHX_HD static constexpr int off4(int m) { int o = 0; for (int k = 0; k < m; ++k) o += len4(k) * L; return o; }
    static constexpr int TOTAL4 = off4(NK);
    static constexpr int THREADS = 64 * MHX_EMCEE_COOP_WAVES;
    MHX_HD static constexpr int nit(int m) { return (len4(m) * L + THREADS - 1) / THREADS; }   // float4 per thread of row set m
    MHX_HD static constexpr int maxit() { int x = 1; for (int k = 0; k < NK; ++k) x = nit(k) > x ? nit(k) : x; return x; }
};

// this thread's float4 of the factor image, straight from the packed factor: unconditional, index-clamped
// loads (zero selected afterwards), so that hipcc puts ALL of them in flight at once
template <int D, int L>
MHX_DEV void mhx_dense_image_load(const mhx_real* __restrict__ A, mhx_e4 (&areg)[mhx_emcee_geom<D, L>::NK][mhx_emcee_geom<D, L>::maxit()])
{
    typedef mhx_emcee_geom<D, L> GEO;
#pragma unroll
    for (int m = 0; m < GEO::NK; ++m) {
#pragma unroll
        for (int it = 0; it < GEO::nit(m); ++it) {
            const int g = threadIdx.x + GEO::THREADS * it;
            const bool ok = g < GEO::len4(m) * L;
            const int gg = ok ? g : 0;
            const int jj4 = gg / L, r = gg % L + L * m;
            const int base = r < D ? r * (r + 1) / 2 : 0;
            mhx_real e[4];
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                const bool in = ok && r < D && 4 * jj4 + cidx <= r;
                const mhx_real a0 = A[in ? base + 4 * jj4 + cidx : 0];
                e[cidx] = in ? a0 : MHX_R(0.0);
            }
            areg[m][it].x = e[0]; areg[m][it].y = e[1]; areg[m][it].z = e[2]; areg[m][it].w = e[3];
        }
    }
}
template <int D, int L>
MHX_DEV void mhx_dense_image_store(const mhx_e4 (&areg)[mhx_emcee_geom<D, L>::NK][mhx_emcee_geom<D, L>::maxit()], mhx_e4* Ash4)
{
    typedef mhx_emcee_geom<D, L> GEO;
#pragma unroll
    for (int m = 0; m < GEO::NK; ++m) {
#pragma unroll
        for (int it = 0; it < GEO::nit(m); ++it) {
            const int g = threadIdx.x + GEO::THREADS * it;
            if (g < GEO::len4(m) * L) Ash4[GEO::off4(m) + g] = areg[m][it];
        }
    }
}
// the whole image, in batches of row sets that keep at most 16 float4 of staging registers per thread (one
// batch up to d = 128; a persistent kernel pays the extra round trips of a larger image once per launch)
template <int D, int L>
MHX_DEV void mhx_dense_image_fill(const mhx_real* __restrict__ A, mhx_e4* img4)
{
    typedef mhx_emcee_geom<D, L> GEO;
    constexpr int GB = GEO::maxit() >= 16 ? 1 : 16 / GEO::maxit();          // row sets per batch
#pragma unroll
    for (int m0 = 0; m0 < GEO::NK; m0 += GB) {
        mhx_e4 regs[GB][GEO::maxit()];
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            const int m = m0 + i;
            if (m < GEO::NK) {
#pragma unroll
                for (int it = 0; it < GEO::nit(m); ++it) {
                    const int g = threadIdx.x + GEO::THREADS * it;
                    const bool ok = g < GEO::len4(m) * L;
                    const int gg = ok ? g : 0;
                    const int jj4 = gg / L, r = gg % L + L * m;
                    const int base = r < D ? r * (r + 1) / 2 : 0;
                    mhx_real e[4];
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) {
                        const bool in = ok && r < D && 4 * jj4 + cidx <= r;
                        const mhx_real a0 = A[in ? base + 4 * jj4 + cidx : 0];
                        e[cidx] = in ? a0 : MHX_R(0.0);
                    }
                    regs[i][it].x = e[0]; regs[i][it].y = e[1]; regs[i][it].z = e[2]; regs[i][it].w = e[3];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            const int m = m0 + i;
            if (m < GEO::NK) {
#pragma unroll
                for (int it = 0; it < GEO::nit(m); ++it) {
                    const int g = threadIdx.x + GEO::THREADS * it;
                    if (g < GEO::len4(m) * L) img4[GEO::off4(m) + g] = regs[i][it];
                }
            }
        }
    }
}

// rows l, l+L, ... of (lower-triangular image) x (row vector in LDS): ascending columns, one fmaf chain per row
template <int D, int L>
MHX_DEV void mhx_dense_rows(const mhx_e4* img4, const mhx_e4* row4, const int l, mhx_real (&w)[mhx_emcee_geom<D, L>::NK])
{
    typedef mhx_emcee_geom<D, L> GEO;
#pragma unroll
    for (int m = 0; m < GEO::NK; ++m) {
        mhx_real acc = MHX_R(0.0);
#pragma unroll
        for (int jj4 = 0; jj4 < GEO::len4(m); ++jj4) {
            const mhx_e4 av = img4[GEO::off4(m) + jj4 * L + l];
            const mhx_e4 yv = row4[jj4];
            acc = mhx_fma(av.x, yv.x, acc);
            acc = mhx_fma(av.y, yv.y, acc);
            acc = mhx_fma(av.z, yv.z, acc);
            acc = mhx_fma(av.w, yv.w, acc);
        }
        w[m] = acc;
    }
}
// lane l's share of |A y|^2: its rows of A y squared and summed in ascending row order; the L shares meet in the
// caller's butterfly
template <int D, int L>
MHX_DEV mhx_real mhx_dense_rows_sq(const mhx_e4* Ash4, const mhx_e4* yrow4, const int l)
{
    typedef mhx_emcee_geom<D, L> GEO;
    mhx_real w[GEO::NK];
    mhx_dense_rows<D, L>(Ash4, yrow4, l, w);
    mhx_real q = MHX_R(0.0);
#pragma unroll
    for (int m = 0; m < GEO::NK; ++m) q = (l + L * m) < D ? mhx_fma(w[m], w[m], q) : q;
    return q;
}

// A BANDED factor (A_rc == 0 for c < r - BW: the precision factor of a Markov / autoregressive Gaussian; BW = 1 for
// Sigma_ij = rho^|i-j|): lane l's rows l, l+L, ... need BW + 1 products each instead of r + 1.  The band entries come straight
// from the packed factor (index-clamped loads, issued at the top of the kernel), y from the wave's LDS row; the chain
// w_r = fma(A_rc, y_c, w_r) over c = r-BW .. r starts from +0 like the dense one, whose skipped terms fma(0, y_c, w) leave w
// unchanged bit for bit (y finite) -- same value, no factor image in LDS and no block barrier.
template <int D, int L, int BW>
MHX_DEV void mhx_band_load(const mhx_real* __restrict__ A, const int l, mhx_real (&ab)[mhx_emcee_geom<D, L>::NK][BW + 1])
{
#pragma unroll
    for (int m = 0; m < mhx_emcee_geom<D, L>::NK; ++m) {
        const int r = l + L * m;
        const int base = r < D ? r * (r + 1) / 2 : 0;
#pragma unroll
        for (int t = 0; t <= BW; ++t) {
            const int c = r - BW + t;
            const bool in = r < D && c >= 0;
            const mhx_real a0 = A[in ? base + c : 0];
            ab[m][t] = in ? a0 : MHX_R(0.0);
        }
    }
}
template <int D, int L, int BW>
MHX_DEV mhx_real mhx_band_rows_sq(const mhx_real (&ab)[mhx_emcee_geom<D, L>::NK][BW + 1], const mhx_real* yrow, const int l)
{
    mhx_real q = MHX_R(0.0);
#pragma unroll
    for (int m = 0; m < mhx_emcee_geom<D, L>::NK; ++m) {
        const int r = l + L * m;
        mhx_real w = MHX_R(0.0);
#pragma unroll
        for (int t = 0; t <= BW; ++t) {
            const int c = r - BW + t;
            const mhx_real yc = yrow[(r < D && c >= 0) ? c : 0];
            w = mhx_fma(ab[m][t], yc, w);
        }
        q = r < D ? mhx_fma(w, w, q) : q;
    }
    return q;
}

// Timing points: empty in the release library (the tools build defines them; its block is not part of the embedded source).
#ifdef MHX_TOOLS_BUILD
// libmhx_tools.so, options EMCEE_PROBE / EMCEE_STAMPS of mhx_ctx_set_option: end a half-step after its n-th phase (wrong chains,
// right latencies) / stamp s_memtime at the phase boundaries
#ifndef MHX_EMCEE_PROBE
#define MHX_EMCEE_PROBE 0
#endif
#define MHX_TP(n, val) do { if (MHX_EMCEE_PROBE == (n)) { if ((val) == MHX_R(12345.678)) a.lp[0] = (val); return; } } while (0)
#define MHX_TP_IS(n) (MHX_EMCEE_PROBE == (n))
#define MHX_TP_LT(n) (MHX_EMCEE_PROBE < (n))
// stamps: lane 0 of every wave stores s_memtime / s_memrealtime at its phase boundaries into a.ybuf ([waves of the launch][16]
// 64-bit words; the host writes the last launch's buffer to the file named by option EMCEE_STAMPS_FILE)
#ifndef MHX_EMCEE_STAMPS
#define MHX_EMCEE_STAMPS 0
#endif
#if MHX_EMCEE_STAMPS
#define MHX_STAMP(k) do { if ((threadIdx.x & 63) == 0) { mhx_u64* sp_ = (mhx_u64*)a.ybuf + ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16; \
        sp_[2 * (k)] = __builtin_amdgcn_s_memtime(); sp_[2 * (k) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MHX_STAMP(k) do { } while (0)
#endif
#else
#define MHX_TP(n, val) do { } while (0)
#define MHX_TP_IS(n) false
#define MHX_TP_LT(n) true
#define MHX_STAMP(k) do { } while (0)
#endif

#ifndef MHX_EMCEE_REC_STORE
#define MHX_EMCEE_REC_STORE 0       // tuning knob: how the record leaves -- 0 plain stores, 1 non-temporal
#endif
#if MHX_EMCEE_REC_STORE == 1
#define MHX_REC_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MHX_REC_ST(p, v) (*(p) = (v))
#endif
#ifndef MHX_EMCEE_ROW_STORE
#define MHX_EMCEE_ROW_STORE 1       // tuning knob (sweep kernels): how the new rows leave -- 0 plain stores, 1 non-temporal (8.34 -> 8.09 us per sweep on C3)
#endif
#if MHX_EMCEE_ROW_STORE == 1
#define MHX_ROW_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MHX_ROW_ST(p, v) (*(p) = (v))
#endif
#ifndef MHX_EMCEE_COOP_REC
#define MHX_EMCEE_COOP_REC 0                 // how the record leaves the lane-group form: 0 straight from the lanes (4-walker runs), 1 through the block's LDS (a block barrier: measured slower here, 5.96 against 5.27 us on C3 -- the waves of this form do not otherwise wait for each other; it pays in the scalar-factor form, which has its barriers anyway)
#endif
// BW < 0: dense factor (image in LDS, one block barrier); BW >= 0: a factor of bandwidth BW (no image, no barrier before the record)
template <int D, int L, int BW = -1>
MHX_DEV void mhx_emcee_coop_body(const mhx_emcee_args& a, const mhx_real* __restrict__ A, mhx_real* ysh_all, mhx_e4* Ash4)
{
    typedef mhx_emcee_geom<D, L> GEO;
    constexpr int CPW = 64 / L;                  // walkers per wave
    constexpr int NK = GEO::NK;                  // dimensions (and rows) per lane
    constexpr int DP4 = GEO::DP4;
    constexpr bool BAND = BW >= 0;
    // The factor goes to LDS once per block (a lane's row reads are scattered: from LDS, not L2); its loads fly
    // together with the walker rows below: the launch is a chain of memory latencies.
    // (images of more than 16 float4 per thread are filled in batches further down instead)
    constexpr bool ONE_BATCH = !BAND && NK * GEO::maxit() <= 16;
    mhx_e4 areg[ONE_BATCH ? NK : 1][GEO::maxit()];
    if constexpr (ONE_BATCH) mhx_dense_image_load<D, L>(A, areg);
    mhx_real ab[BAND ? NK : 1][BAND ? BW + 1 : 1];
    if constexpr (BAND) mhx_band_load<D, L, BW>(A, (int)((threadIdx.x & 63) / CPW), ab);
    const int wave = threadIdx.x >> 6;
    mhx_real* ysh = ysh_all + wave * (CPW * DP4);
    const int W = a.nwalkers;
    const int halfW = W / 2;
    const int lo = a.half ? halfW : 0;
    const int cnt = a.half ? W - halfW : halfW;
    const int lane = threadIdx.x & 63;
    const int cw = lane & (CPW - 1);
    const int l = lane / CPW;
    const int t_raw = a.t_begin + (blockIdx.x * MHX_EMCEE_COOP_WAVES + wave) * CPW + cw;
    const bool valid = t_raw < cnt && t_raw < a.t_begin + a.t_count;
    const int i = lo + (valid ? t_raw : cnt - 1);
    const int ostart = a.half ? 0 : halfW;
    const int osize = a.half ? halfW : W - halfW;
    const long ld = a.ld;

    MHX_TP(1, (mhx_real)i);                                               // launch + arguments
    // the walker's own row does not wait for the draw (the partner's does): its loads go out first
    constexpr int NQ = GEO::NQ, NQL = GEO::NQL;
    mhx_e4 xs[NQL], ysl[NQL];
    mhx_e4* xrow_i = (mhx_e4*)(a.xw + (long)i * mhx_xw_pitch(D));
    {
        const mhx_e4 zero4 = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
#pragma unroll
        for (int m = 0; m < NQL; ++m) { const int q4 = l + L * m; xs[m] = q4 < NQ ? xrow_i[q4] : zero4; }
    }
    const mhx_real lpi = a.lp[i];                                            // in flight with the rows
    const mhx_u32 acc_i = a.acc_count[i];
    // deferred record (a.rec_other_slot): this group's walker of the half at rest -- loads issued here, stores after the move
    const int io = (a.rec_other_slot >= 0 && valid && t_raw < osize) ? ostart + t_raw : -1;
    mhx_e4 xo[mhx_emcee_geom<D, L>::NQL];
    mhx_real lpo = MHX_R(0.0);
    unsigned char lao = 0;
    if (io >= 0) {
        const mhx_e4* xrow_o = (const mhx_e4*)(a.xw + (long)io * mhx_xw_pitch(D));
#pragma unroll
        for (int m = 0; m < mhx_emcee_geom<D, L>::NQL; ++m) { const int q4 = l + L * m; if (q4 < mhx_emcee_geom<D, L>::NQ) xo[m] = xrow_o[q4]; }
        if (l == 0) { lpo = a.lp[io]; lao = a.last_acc[io]; }
    }
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep);
    const int j = ostart + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)osize) >> 32);
    const mhx_real u = dr.u;
    const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), u, MHX_R(1.0));
    const mhx_real z = (tt * tt) / a.stretch;                               // src/emcee.jl:81
    const mhx_real alphamult = (mhx_real)(D - 1) * mhx_log(z);                 // :82
    MHX_TP(2, alphamult + (mhx_real)j);                                   // + the draws

    // the move, element-wise on float4 slices of the two rows (lane l: float4 l, l+L, ...); the zero pad of
    // the rows gives the zero pad of y that multiplies the zeros of the factor image
    mhx_real* yrow = ysh + cw * DP4;
    const mhx_e4* xrow_j = (const mhx_e4*)(a.xw + (long)j * mhx_xw_pitch(D));
#pragma unroll
    for (int m = 0; m < NQL; ++m) {
        const int q4 = l + L * m;
        const mhx_e4 zero4 = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
        ysl[m] = zero4;
        if (q4 < NQ) {
            const mhx_e4 xi = xs[m];
            const mhx_e4 xj = xrow_j[q4];
            ysl[m].x = mhx_fma(z, xi.x - xj.x, xj.x);                    // :85
            ysl[m].y = mhx_fma(z, xi.y - xj.y, xj.y);
            ysl[m].z = mhx_fma(z, xi.z - xj.z, xj.z);
            ysl[m].w = mhx_fma(z, xi.w - xj.w, xj.w);
        }
        if (q4 < DP4 / 4) ((mhx_e4*)yrow)[q4] = ysl[m];
    }
    MHX_TP(3, ysl[0].x);                                                  // + the two rows, the move
    if (!MHX_TP_IS(6) && a.rec_other_slot >= 0) {
        // the partner row x_j was requested after these loads: they have arrived.  The stores have the rest of the launch to drain.
        auto rec = [&](const int iw, const mhx_e4 (&xr)[mhx_emcee_geom<D, L>::NQL], const mhx_real lpr, const unsigned char lar) {
            mhx_real* row = a.samples + a.rec_other_slot * (long)(D + 1) * ld + iw;
#pragma unroll
            for (int m = 0; m < NQL; ++m) {
                const int k = 4 * (l + L * m);
                if (k + 0 < D) MHX_REC_ST(&row[(long)(k + 0) * ld], xr[m].x);
                if (k + 1 < D) MHX_REC_ST(&row[(long)(k + 1) * ld], xr[m].y);
                if (k + 2 < D) MHX_REC_ST(&row[(long)(k + 2) * ld], xr[m].z);
                if (k + 3 < D) MHX_REC_ST(&row[(long)(k + 3) * ld], xr[m].w);
            }
            if (l == 0) { MHX_REC_ST(&row[(long)D * ld], lpr); a.accepted[a.rec_other_slot * ld + iw] = lar; }
        };
        if (io >= 0) rec(io, xo, lpo, lao);
        // the half at rest may be one walker larger than the moving one (odd W): the first group takes it too
        if (valid && t_raw + cnt < osize) {
            const int ie = ostart + t_raw + cnt;
            mhx_e4 xe[mhx_emcee_geom<D, L>::NQL];
            const mhx_e4* xrow_e = (const mhx_e4*)(a.xw + (long)ie * mhx_xw_pitch(D));
#pragma unroll
            for (int m = 0; m < NQL; ++m) { const int q4 = l + L * m; if (q4 < NQ) xe[m] = xrow_e[q4]; }
            rec(ie, xe, l == 0 ? a.lp[ie] : MHX_R(0.0), l == 0 ? a.last_acc[ie] : (unsigned char)0);
        }
    }
    mhx_real q;
    if constexpr (BAND) {
        MHX_WAVE_SYNC();                                                     // the candidate rows of a wave are its own
        q = mhx_band_rows_sq<D, L, BW>(ab, yrow, l);
    } else {
        if constexpr (ONE_BATCH) mhx_dense_image_store<D, L>(areg, Ash4);
        else mhx_dense_image_fill<D, L>(A, Ash4);
        __syncthreads();
        MHX_TP(4, ysl[0].x + ((const mhx_real*)Ash4)[threadIdx.x]);       // + the factor image in LDS
        q = mhx_dense_rows_sq<D, L>(Ash4, (const mhx_e4*)yrow, l);
    }
    q = mhx_butterfly<L>(q);
    const mhx_real lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
    MHX_TP(5, lpy + ysl[0].x);                                            // + A y, the butterfly
    const mhx_real alpha = (alphamult + lpy) - lpi;                         // :91
    const mhx_real logu = dr.logu;
    const bool acc = logu <= alpha;                                      // :93
    if (valid) {
        if (acc) {
#pragma unroll
            for (int m = 0; m < NQL; ++m) { const int q4 = l + L * m; if (q4 < NQ) xrow_i[q4] = ysl[m]; }
            if (l == 0) { a.lp[i] = lpy; a.acc_count[i] = acc_i + 1u; }
        }
        if (l == 0) a.last_acc[i] = acc ? 1 : 0;
        if (MHX_TP_IS(6)) return;                                    // + accept and the state update, no record
        if (!MHX_EMCEE_COOP_REC && a.save_slot >= 0) {
            // the record is [dim+1][W] (walker fastest): 16-byte runs per dimension from this wave's walkers
            // (staging it through LDS for 64-byte runs measured no faster)
            mhx_real* row = a.samples + a.save_slot * (long)(D + 1) * ld + i;
#pragma unroll
            for (int m = 0; m < NQL; ++m) {
                const int k = 4 * (l + L * m);
                const mhx_e4 v = acc ? ysl[m] : xs[m];
                if (k + 0 < D) MHX_REC_ST(&row[(long)(k + 0) * ld], v.x);
                if (k + 1 < D) MHX_REC_ST(&row[(long)(k + 1) * ld], v.y);
                if (k + 2 < D) MHX_REC_ST(&row[(long)(k + 2) * ld], v.z);
                if (k + 3 < D) MHX_REC_ST(&row[(long)(k + 3) * ld], v.w);
            }
            if (l == 0) {
                MHX_REC_ST(&row[(long)D * ld], acc ? lpy : lpi);
                a.accepted[a.save_slot * ld + i] = acc ? 1 : 0;
            }
        }
    }
#if MHX_EMCEE_COOP_REC
    // The record leaves as whole row segments (round 4): the final rows of the block's walkers meet in its LDS -- the candidate
    // rows are there, a rejected move puts the walker back, lp and the accept flag ride in the row's tail -- and every row k of the
    // [dim+1][W] record is written for all the block's consecutive walkers at once (32 x sizeof(real) contiguous bytes at C3's shape
    // instead of 4 walkers per store).  A tuning knob (MHX_EMCEE_COOP_REC=1), off by default: see the macro.
    if (!MHX_TP_IS(6) && a.save_slot >= 0) {                          // (uniform)
        if (!acc) {
#pragma unroll
            for (int m = 0; m < NQL; ++m) { const int q4 = l + L * m; if (q4 < NQ) ((mhx_e4*)yrow)[q4] = xs[m]; }
        }
        if (l == 0) { yrow[GEO::XP] = acc ? lpy : lpi; yrow[GEO::XP + 1] = acc ? MHX_R(1.0) : MHX_R(0.0); }
        __syncthreads();
        constexpr int WPBC = MHX_EMCEE_COOP_WAVES * CPW;                     // walkers per block; threads / WPBC == L
        const int wr = threadIdx.x % WPBC, ks = threadIdx.x / WPBC;
        const int tw = a.t_begin + blockIdx.x * WPBC + wr;
        if (tw < cnt && tw < a.t_begin + a.t_count) {
            const mhx_real* fin = ysh_all + wr * DP4;
            mhx_real* col = a.samples + a.save_slot * (long)(D + 1) * ld + (lo + tw);
#pragma unroll
            for (int k = ks; k < D + 1; k += L) MHX_REC_ST(&col[(long)k * ld], fin[k < D ? k : GEO::XP]);
            if (ks == L - 1) a.accepted[a.save_slot * ld + lo + tw] = fin[GEO::XP + 1] != MHX_R(0.0) ? 1 : 0;
        }
    }
#endif
    // no per-wave atomic here: a half-step is one short launch of ~1000 waves, and 1000 atomics on one
    // address (~12 ns each) would cost more than the move; the host sums acc_count after the run.
}

// ---------------------------------------------------------------------------------------------
// ONE LAUNCH PER SWEEP (round 4).  A half-step is a 5-us launch of which 2.7 us is the kernel boundary, and the two half-steps
// of a sweep are dependent launches only because the second half moves against the NEW state of the first.  But the draws of a
// walker are a function of (seed, ensemble, sweep, walker) alone: a walker b of the second half knows its partner a before any
// memory has answered, and a's partner b' as well -- so the group that moves b RE-DOES a's move from the old state (rows a, b'
// and lp_a as the launch found them), which is bit for bit what the group that owns a computes, and then moves b against the
// result.  No group waits for another one; the price is a second (and, below, a third) row product for the second half's
// groups and a state that is double-buffered (xw / lp are read, xw_out / lp_out take EVERY walker's new row: a's old row must
// outlive the launch that replaces it; the host swaps the buffers -- and a walker that moved neither in this sweep nor in the one
// before is not written at all: the buffer this launch writes is the one the launch before READ, it holds the state of two sweeps ago,
// which for such a walker is its state now; last_acc carries "moved in the previous sweep", so a third of the rows are stored).  The chain of b is not even two moves deep: its candidate
// is formed for both outcomes of a's accept test -- y_b0 against x_a, y_b1 against y_a -- and the three row products run side by
// side; the test then selects.  Same chains, same records, same counters as two half-step launches (oracle: mode 1).
// Blocks [0, nbB) move the second half (the longer program: dispatched first), the others the first half.
template <int D, int L, int BW = -1>
MHX_DEV void mhx_emcee_coop_sweep_body(const mhx_emcee_args& a, const mhx_real* __restrict__ A, mhx_real* ysh_all, mhx_e4* Ash4)
{
    typedef mhx_emcee_geom<D, L> GEO;
    constexpr int CPW = 64 / L;
    constexpr int NK = GEO::NK;
    constexpr int DP4 = GEO::DP4;
    constexpr int NQ = GEO::NQ, NQL = GEO::NQL;
    constexpr bool BAND = BW >= 0;
    constexpr bool ONE_BATCH = !BAND && NK * GEO::maxit() <= 16;
    constexpr int WPBK = MHX_EMCEE_COOP_WAVES * CPW;                         // walkers per block
    mhx_e4 areg[ONE_BATCH ? NK : 1][GEO::maxit()];
    if constexpr (ONE_BATCH) mhx_dense_image_load<D, L>(A, areg);
    mhx_real ab[BAND ? NK : 1][BAND ? BW + 1 : 1];
    if constexpr (BAND) mhx_band_load<D, L, BW>(A, (int)((threadIdx.x & 63) / CPW), ab);
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int cw = lane & (CPW - 1);
    const int l = lane / CPW;
    mhx_real* yrow = ysh_all + (wave * 3 * CPW + cw) * DP4;                  // three candidate rows per walker: + CPW * DP4 each
    const int W = a.nwalkers;
    const int halfW = W / 2, cntB = W - halfW;
    const int nbB = (cntB + WPBK - 1) / WPBK;
    const bool second = (int)blockIdx.x < nbB;                               // (block-uniform)
    const int blk = second ? (int)blockIdx.x : (int)blockIdx.x - nbB;
    const int cnt = second ? cntB : halfW;
    const int t_raw = (blk * MHX_EMCEE_COOP_WAVES + wave) * CPW + cw;
    const bool valid = t_raw < cnt;
    const int i = (second ? halfW : 0) + (valid ? t_raw : cnt - 1);
    const long ld = a.ld;
    const mhx_e4 zero4 = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};

    mhx_e4 xs[NQL], ysl[NQL];
    const mhx_e4* xrow_i = (const mhx_e4*)(a.xw + (long)i * mhx_xw_pitch(D));
#pragma unroll
    for (int m = 0; m < NQL; ++m) { const int q4 = l + L * m; xs[m] = q4 < NQ ? xrow_i[q4] : zero4; }
    const mhx_real lpi = a.lp[i];
    const mhx_u32 acc_i = a.acc_count[i];
    const unsigned char last_i = a.last_acc[i];
    const bool moved_before = a.all_rows != 0 || last_i != 0;                // xw_out does not hold this walker's row
    // the record of the PREVIOUS sweep (a.rec_other_slot >= 0): the row just loaded is that sweep's final state of this walker, lp and
    // the flag ride along -- so its stores leave here, at the top of the launch, and have the whole kernel to drain, instead of at the
    // end of the launch that moved the walker, where the kernel boundary waits for them (no extra load: unlike the half-step form's
    // deferred record, every walker's row is read by its own group anyway)
    if (a.rec_other_slot >= 0 && valid) {
        mhx_real* row = a.samples + a.rec_other_slot * (long)(D + 1) * ld + i;
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int k = 4 * (l + L * m);
            if (k + 0 < D) MHX_REC_ST(&row[(long)(k + 0) * ld], xs[m].x);
            if (k + 1 < D) MHX_REC_ST(&row[(long)(k + 1) * ld], xs[m].y);
            if (k + 2 < D) MHX_REC_ST(&row[(long)(k + 2) * ld], xs[m].z);
            if (k + 3 < D) MHX_REC_ST(&row[(long)(k + 3) * ld], xs[m].w);
        }
        if (l == 0) {
            MHX_REC_ST(&row[(long)D * ld], lpi);
            a.accepted[a.rec_other_slot * ld + i] = last_i;
        }
    }
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep);
    // the partner: the second half draws from the first ([0, halfW)), the first from the second
    const int j = (second ? 0 : halfW) + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)(second ? halfW : cntB)) >> 32);
    const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), dr.u, MHX_R(1.0));
    const mhx_real z = (tt * tt) / a.stretch;                               // src/emcee.jl:81
    const mhx_real alphamult = (mhx_real)(D - 1) * mhx_log(z);              // :82
    const mhx_e4* xrow_j = (const mhx_e4*)(a.xw + (long)j * mhx_xw_pitch(D));
    // the partner's row goes out the moment the draw names it -- before the second half works out ITS partner's partner (left in the
    // candidate loop, these loads were issued behind the second draw's Philox rounds and logarithm)
    mhx_e4 xjs[NQL];
#pragma unroll
    for (int m = 0; m < NQL; ++m) { const int q4 = l + L * m; xjs[m] = q4 < NQ ? xrow_j[q4] : zero4; }
    __builtin_amdgcn_sched_barrier(0);
    MHX_TP(2, alphamult + (mhx_real)j + xs[0].x + lpi);                  // launch, own row, the draws
    auto stretch = [](const mhx_real zz, const mhx_e4 xi, const mhx_e4 xj) {   // :85, element-wise
        mhx_e4 y;
        y.x = mhx_fma(zz, xi.x - xj.x, xj.x);
        y.y = mhx_fma(zz, xi.y - xj.y, xj.y);
        y.z = mhx_fma(zz, xi.z - xj.z, xj.z);
        y.w = mhx_fma(zz, xi.w - xj.w, xj.w);
        return y;
    };
    auto row_products = [&](const mhx_real* yr) {
        mhx_real q;
        if constexpr (BAND) q = mhx_band_rows_sq<D, L, BW>(ab, yr, l);
        else q = mhx_dense_rows_sq<D, L>(Ash4, (const mhx_e4*)yr, l);
        return q;
    };
    auto image_ready = [&]() {
        if constexpr (BAND) MHX_WAVE_SYNC();                                 // the candidate rows of a wave are its own
        else {
            if constexpr (ONE_BATCH) mhx_dense_image_store<D, L>(areg, Ash4);
            else mhx_dense_image_fill<D, L>(A, Ash4);
            __syncthreads();
        }
    };
    mhx_real lpy;
    if (!second) {
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int q4 = l + L * m;
            ysl[m] = q4 < NQ ? stretch(z, xs[m], xjs[m]) : zero4;
            if (q4 < DP4 / 4) ((mhx_e4*)yrow)[q4] = ysl[m];
        }
        image_ready();
        MHX_TP(3, ysl[0].x);                                              // + the partner rows, the candidates in LDS
        const mhx_real q = mhx_butterfly<L>(row_products(yrow));
        lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
    } else {
        // j is a walker of the first half: its own move of this sweep, from the state this launch found
        const mhx_emcee_draws da = mhx_emcee_draw(ks, (mhx_u32)j, (mhx_u32)a.ensemble_id, a.sweep);
        const int jb = halfW + (int)(((mhx_u64)da.partner * (mhx_u64)(mhx_u32)cntB) >> 32);
        const mhx_e4* xrow_b = (const mhx_e4*)(a.xw + (long)jb * mhx_xw_pitch(D));
        const mhx_real lpa = a.lp[j];
        const mhx_real ta = mhx_fma(a.stretch - MHX_R(1.0), da.u, MHX_R(1.0));
        const mhx_real za = (ta * ta) / a.stretch;
        const mhx_real alphamult_a = (mhx_real)(D - 1) * mhx_log(za);
        mhx_e4 y1[NQL];
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int q4 = l + L * m;
            mhx_e4 ya = zero4;
            ysl[m] = zero4; y1[m] = zero4;
            if (q4 < NQ) {
                const mhx_e4 xa = xjs[m];
                ya = stretch(za, xa, xrow_b[q4]);                           // a's candidate
                ysl[m] = stretch(z, xs[m], xa);                              // this walker's candidate if a stays
                y1[m] = stretch(z, xs[m], ya);                               //                          if a moves
            }
            if (q4 < DP4 / 4) {
                ((mhx_e4*)yrow)[q4] = ya;
                ((mhx_e4*)(yrow + CPW * DP4))[q4] = ysl[m];
                ((mhx_e4*)(yrow + 2 * CPW * DP4))[q4] = y1[m];
            }
        }
        image_ready();
        MHX_TP(3, ysl[0].x + y1[0].x);
        const mhx_real qa = mhx_butterfly<L>(row_products(yrow));
        const mhx_real q0 = mhx_butterfly<L>(row_products(yrow + CPW * DP4));
        const mhx_real q1 = mhx_butterfly<L>(row_products(yrow + 2 * CPW * DP4));
        const mhx_real lpya = mhx_fma(-MHX_R(0.5), qa, a.tconst);
        const bool acc_a = da.logu <= (alphamult_a + lpya) - lpa;           // a's accept test (:91-93), as its own group runs it
        lpy = mhx_fma(-MHX_R(0.5), acc_a ? q1 : q0, a.tconst);
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            ysl[m].x = acc_a ? y1[m].x : ysl[m].x; ysl[m].y = acc_a ? y1[m].y : ysl[m].y;
            ysl[m].z = acc_a ? y1[m].z : ysl[m].z; ysl[m].w = acc_a ? y1[m].w : ysl[m].w;
        }
    }
    MHX_TP(5, lpy + ysl[0].x);                                            // + the row products, the butterflies
    const mhx_real alpha = (alphamult + lpy) - lpi;                         // :91
    const bool acc = dr.logu <= alpha;                                      // :93
    if (valid) {
        mhx_e4* xrow_o = (mhx_e4*)(a.xw_out + (long)i * mhx_xw_pitch(D));
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int q4 = l + L * m;
            mhx_e4 v;
            v.x = acc ? ysl[m].x : xs[m].x; v.y = acc ? ysl[m].y : xs[m].y; v.z = acc ? ysl[m].z : xs[m].z; v.w = acc ? ysl[m].w : xs[m].w;
            ysl[m] = v;
            if (q4 < NQ && (acc || (moved_before && MHX_TP_LT(7)))) MHX_ROW_ST(&xrow_o[q4], v);      // (probes 7, 8: timing only, accepted rows alone)
        }
        if (l == 0) {
            a.lp_out[i] = acc ? lpy : lpi;
            if (acc) a.acc_count[i] = acc_i + 1u;
            a.last_acc[i] = acc ? 1 : 0;
        }
        if (MHX_TP_IS(6) || MHX_TP_IS(7)) return;            // + accept and the new state, no record
        if (a.save_slot >= 0) {
            mhx_real* row = a.samples + a.save_slot * (long)(D + 1) * ld + i;
#pragma unroll
            for (int m = 0; m < NQL; ++m) {
                const int k = 4 * (l + L * m);
                if (k + 0 < D) MHX_REC_ST(&row[(long)(k + 0) * ld], ysl[m].x);
                if (k + 1 < D) MHX_REC_ST(&row[(long)(k + 1) * ld], ysl[m].y);
                if (k + 2 < D) MHX_REC_ST(&row[(long)(k + 2) * ld], ysl[m].z);
                if (k + 3 < D) MHX_REC_ST(&row[(long)(k + 3) * ld], ysl[m].w);
            }
            if (l == 0) {
                MHX_REC_ST(&row[(long)D * ld], acc ? lpy : lpi);
                a.accepted[a.save_slot * ld + i] = acc ? 1 : 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The SCALAR-FACTOR form of the cooperative stretch move (dense precision factor; round 4).
//
// The factor A = inv(chol Sigma) is the same for every walker.  In the lane-group form above a lane owns ROWS of A y, so the A
// operand differs from lane to lane and has to come from an LDS image that every block rebuilds in every 5-us launch (index
// arithmetic, zero selects, a block barrier), and every fma needs an LDS read of A.  Here a lane owns a WALKER during the
// mat-vec: all 64 lanes of a wave then multiply by the SAME A_rc, which is a wave-uniform value -- it streams through the scalar
// cache (s_load_dwordxN straight from the packed factor: a row is contiguous) into SGPR pairs and is the scalar operand of
// v_fma: no image, no LDS traffic for A, one fma instruction per product.  The rows are split over the NW waves of a block by
// r mod NW (wave g: rows g, g + NW, ...), which is the spec's reduction shape L = NW (oracle: reduce_lanes = NW): every row is one
// ascending fma chain from +0, wave g sums the squares of its rows in ascending order, the NW partial sums meet in the butterfly's
// tree ((q0 + q1) + (q2 + q3)) + ... -- bit for bit what the lane-group form with L = NW computes.
//   phase 1 (NW lanes per walker, as above): draws, the two rows, the move y = x_j + z (x_i - x_j); y goes to the block's LDS
//           as a [64 walkers][YS] array whose pitch is an odd number of 16-byte units (conflict-free b128 reads by walker);
//   phase 2 (lane = walker, wave = row class): y of the lane's walker into registers, the row products against SGPR operands,
//           the wave's partial sum of squares to LDS;
//   phase 3 (phase 1's mapping, whose registers still hold x_i and y): the butterfly tree, the accept test, state and record.
// Two block barriers; 64 walkers per block of NW waves.
#ifndef MHX_EMCEE_SCAL_WPB
#define MHX_EMCEE_SCAL_WPB 64                // walkers per block of the scalar-factor form (64: every lane of phase 2 owns one; 32: half)
#endif
template <int D, int NW>
struct mhx_emcee_sgeom {
    static constexpr int WPB = MHX_EMCEE_SCAL_WPB;             // walkers per block
    static constexpr int LM = 64 * NW / WPB;                   // lanes per walker in the move mapping
    static constexpr int XP = (D + 3) & ~3;                    // pitch of a walker's row in the walker-major state
    static constexpr int NQ = XP / 4;                          // e4 per walker
    static constexpr int NQL = (NQ + LM - 1) / LM;             // e4 slots per lane in the move mapping
    static constexpr int U = 16 / (int)sizeof(mhx_real);       // reals per 16 bytes
    static constexpr int YS = ((XP / U) | 1) * U;              // y pitch: >= XP, an odd number of 16-byte units
    static constexpr int LDS_REALS = WPB * YS + NW * 64;       // y rows + the partial sums
};

typedef mhx_real mhx_e2 __attribute__((ext_vector_type(2)));

// wave g's rows of A y, squared and summed: A_rc is wave-uniform (kernel argument + compile-time offset: a scalar load)
template <int D, int NW, int G>
MHX_DEV mhx_real mhx_scal_rows_sq(const mhx_real* __restrict__ A, const mhx_real (&y)[(D + 3) & ~3])
{
    mhx_real q = MHX_R(0.0);
#pragma unroll
    for (int r = G; r < D; r += NW) {
        const mhx_real* Ar = A + (r * (r + 1)) / 2;
        mhx_real w = MHX_R(0.0);
#pragma unroll
        for (int c = 0; c <= r; ++c) w = mhx_fma(Ar[c], y[c], w);
        q = mhx_fma(w, w, q);
    }
    return q;
}
template <int D, int NW, int G = 0>
MHX_DEV mhx_real mhx_scal_rows_dispatch(const int g, const mhx_real* __restrict__ A, const mhx_real (&y)[(D + 3) & ~3])
{
    if constexpr (G < NW) {
        if (g == G) return mhx_scal_rows_sq<D, NW, G>(A, y);       // g is wave-uniform: a scalar branch
        return mhx_scal_rows_dispatch<D, NW, G + 1>(g, A, y);
    } else {
        return MHX_R(0.0);
    }
}

// The same rows with the factor in VGPRs and a DPP broadcast as the multiplier (MHX_EMCEE_SCAL_MODE 1, the default): scalar loads
// return out of order, so a wave can only wait for ALL of them (lgkmcnt(0)) and the ~20 x16 pieces of a row class cannot be kept in
// flight behind the products that consume them -- the SGPR file holds two or three pieces.  Vector loads count in order and the
// VGPR file is large: each wave fetches ITS rows at the top of the kernel, 16 columns per load (lane k of every 16-lane row of the
// wave holds A[r][16 b + k]: one 128-byte segment), they land while phase 1 runs, and the product is
//     v_fmac_f64_dpp w, vA, y_c row_newbcast:k        (w += vA[lane k of the row] * y_c; f32 alike)
// -- still one instruction per product, the multiplier wave-uniform by construction.  The columns run in the outer loop and a wave's
// rows in the inner one, so consecutive instructions belong to different accumulator chains; every chain is ascending in c from +0.
// One column of a row class as ONE asm statement: the products w_m += A[r_m][c] * y_c of the N rows that reach column c, back to back.
// The compiler cannot put anything between them, so the two wait states a DPP source needs after a VALU write (a register copy or an
// AGPR reload of a factor piece -- the compiler does not look inside asm) are paid once per column (s_nop 1), not once per product.
#if MHX_REAL64
#define MHX_DPP_FMAC "v_fmac_f64_dpp"
#else
#define MHX_DPP_FMAC "v_fmac_f32_dpp"
#endif
#define MHX_DPP_TAIL " row_mask:0xf bank_mask:0xf\n\t"
template <int K> MHX_DEV void mhx_dpp_col1(mhx_real& w0, const mhx_real a0, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %1, %2 row_newbcast:%3" MHX_DPP_TAIL
                 : "+v"(w0) : "v"(a0), "v"(yc), "n"(K));
}
template <int K> MHX_DEV void mhx_dpp_col2(mhx_real& w0, mhx_real& w1, const mhx_real a0, const mhx_real a1, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %2, %4 row_newbcast:%5" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %1, %3, %4 row_newbcast:%5" MHX_DPP_TAIL
                 : "+v"(w0), "+v"(w1) : "v"(a0), "v"(a1), "v"(yc), "n"(K));
}
template <int K> MHX_DEV void mhx_dpp_col3(mhx_real& w0, mhx_real& w1, mhx_real& w2, const mhx_real a0, const mhx_real a1, const mhx_real a2, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %3, %6 row_newbcast:%7" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %1, %4, %6 row_newbcast:%7" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %2, %5, %6 row_newbcast:%7" MHX_DPP_TAIL
                 : "+v"(w0), "+v"(w1), "+v"(w2) : "v"(a0), "v"(a1), "v"(a2), "v"(yc), "n"(K));
}
template <int K> MHX_DEV void mhx_dpp_col4(mhx_real& w0, mhx_real& w1, mhx_real& w2, mhx_real& w3, const mhx_real a0, const mhx_real a1, const mhx_real a2, const mhx_real a3, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %4, %8 row_newbcast:%9" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %1, %5, %8 row_newbcast:%9" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %2, %6, %8 row_newbcast:%9" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %3, %7, %8 row_newbcast:%9" MHX_DPP_TAIL
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(yc), "n"(K));
}
template <int K> MHX_DEV void mhx_dpp_col5(mhx_real& w0, mhx_real& w1, mhx_real& w2, mhx_real& w3, mhx_real& w4, const mhx_real a0, const mhx_real a1, const mhx_real a2, const mhx_real a3, const mhx_real a4, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %5, %10 row_newbcast:%11" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %1, %6, %10 row_newbcast:%11" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %2, %7, %10 row_newbcast:%11" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %3, %8, %10 row_newbcast:%11" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %4, %9, %10 row_newbcast:%11" MHX_DPP_TAIL
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(yc), "n"(K));
}
template <int K> MHX_DEV void mhx_dpp_col6(mhx_real& w0, mhx_real& w1, mhx_real& w2, mhx_real& w3, mhx_real& w4, mhx_real& w5, const mhx_real a0, const mhx_real a1, const mhx_real a2, const mhx_real a3, const mhx_real a4, const mhx_real a5, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %6, %12 row_newbcast:%13" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %1, %7, %12 row_newbcast:%13" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %2, %8, %12 row_newbcast:%13" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %3, %9, %12 row_newbcast:%13" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %4, %10, %12 row_newbcast:%13" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %5, %11, %12 row_newbcast:%13" MHX_DPP_TAIL
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(yc), "n"(K));
}
template <int K> MHX_DEV void mhx_dpp_col7(mhx_real& w0, mhx_real& w1, mhx_real& w2, mhx_real& w3, mhx_real& w4, mhx_real& w5, mhx_real& w6, const mhx_real a0, const mhx_real a1, const mhx_real a2, const mhx_real a3, const mhx_real a4, const mhx_real a5, const mhx_real a6, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %7, %14 row_newbcast:%15" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %1, %8, %14 row_newbcast:%15" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %2, %9, %14 row_newbcast:%15" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %3, %10, %14 row_newbcast:%15" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %4, %11, %14 row_newbcast:%15" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %5, %12, %14 row_newbcast:%15" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %6, %13, %14 row_newbcast:%15" MHX_DPP_TAIL
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(yc), "n"(K));
}
template <int K> MHX_DEV void mhx_dpp_col8(mhx_real& w0, mhx_real& w1, mhx_real& w2, mhx_real& w3, mhx_real& w4, mhx_real& w5, mhx_real& w6, mhx_real& w7, const mhx_real a0, const mhx_real a1, const mhx_real a2, const mhx_real a3, const mhx_real a4, const mhx_real a5, const mhx_real a6, const mhx_real a7, const mhx_real yc)
{
    asm volatile("s_nop 1\n\t"
                 MHX_DPP_FMAC " %0, %8, %16 row_newbcast:%17" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %1, %9, %16 row_newbcast:%17" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %2, %10, %16 row_newbcast:%17" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %3, %11, %16 row_newbcast:%17" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %4, %12, %16 row_newbcast:%17" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %5, %13, %16 row_newbcast:%17" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %6, %14, %16 row_newbcast:%17" MHX_DPP_TAIL
                 MHX_DPP_FMAC " %7, %15, %16 row_newbcast:%17" MHX_DPP_TAIL
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(yc), "n"(K));
}
template <int D, int NW>
struct mhx_bcast_geom {
    static constexpr int NRMAX = (D + NW - 1) / NW;            // rows of a class, at most
    static constexpr int CHMAX = (D - 1) / 16 + 1;             // 16-column pieces of a row, at most
    static constexpr int TOT = D * (D + 1) / 2;
};
// this wave's pieces: the same code for every row class g (wave-uniform row arithmetic on the scalar side); row set m of class g
// is row g + NW m, and every class loads as many pieces of it as the longest row of the set needs -- a piece past the end of a
// short row holds the next rows of the packed factor (never used), past the end of the factor the last element
template <int D, int NW>
MHX_DEV void mhx_bcast_load(const int g, const mhx_real* __restrict__ A, const int lane16,
                            mhx_real (&av)[mhx_bcast_geom<D, NW>::NRMAX][mhx_bcast_geom<D, NW>::CHMAX])
{
    constexpr int TOT = mhx_bcast_geom<D, NW>::TOT;
#pragma unroll
    for (int m = 0; m < mhx_bcast_geom<D, NW>::NRMAX; ++m) {
        const int r0 = g + NW * m;
        const int r = r0 < D ? r0 : D - 1;
        const int base = (r * (r + 1)) / 2 + lane16;
        const int rmax = (NW * m + NW - 1) < D ? (NW * m + NW - 1) : D - 1;          // longest row of the set (folds after unrolling)
#pragma unroll
        for (int b = 0; 16 * b <= rmax; ++b) {
            const int idx = base + 16 * b;
            av[m][b] = A[idx < TOT ? idx : TOT - 1];
        }
    }
}
// rows M0 .. M0 + N - 1 of the class (N <= 8) at column C
template <int K, int N, int M0, int B, int NR, int CH>
MHX_DEV void mhx_bcast_group(mhx_real (&w)[NR], const mhx_real (&av)[NR][CH], const mhx_real yc)
{
    if constexpr (N == 1) mhx_dpp_col1<K>(w[M0], av[M0][B], yc);
    else if constexpr (N == 2) mhx_dpp_col2<K>(w[M0], w[M0 + 1], av[M0][B], av[M0 + 1][B], yc);
    else if constexpr (N == 3) mhx_dpp_col3<K>(w[M0], w[M0 + 1], w[M0 + 2], av[M0][B], av[M0 + 1][B], av[M0 + 2][B], yc);
    else if constexpr (N == 4) mhx_dpp_col4<K>(w[M0], w[M0 + 1], w[M0 + 2], w[M0 + 3], av[M0][B], av[M0 + 1][B], av[M0 + 2][B], av[M0 + 3][B], yc);
    else if constexpr (N == 5) mhx_dpp_col5<K>(w[M0], w[M0 + 1], w[M0 + 2], w[M0 + 3], w[M0 + 4], av[M0][B], av[M0 + 1][B], av[M0 + 2][B], av[M0 + 3][B],
                                               av[M0 + 4][B], yc);
    else if constexpr (N == 6) mhx_dpp_col6<K>(w[M0], w[M0 + 1], w[M0 + 2], w[M0 + 3], w[M0 + 4], w[M0 + 5], av[M0][B], av[M0 + 1][B], av[M0 + 2][B],
                                               av[M0 + 3][B], av[M0 + 4][B], av[M0 + 5][B], yc);
    else if constexpr (N == 7) mhx_dpp_col7<K>(w[M0], w[M0 + 1], w[M0 + 2], w[M0 + 3], w[M0 + 4], w[M0 + 5], w[M0 + 6], av[M0][B], av[M0 + 1][B],
                                               av[M0 + 2][B], av[M0 + 3][B], av[M0 + 4][B], av[M0 + 5][B], av[M0 + 6][B], yc);
    else mhx_dpp_col8<K>(w[M0], w[M0 + 1], w[M0 + 2], w[M0 + 3], w[M0 + 4], w[M0 + 5], w[M0 + 6], w[M0 + 7], av[M0][B], av[M0 + 1][B], av[M0 + 2][B],
                         av[M0 + 3][B], av[M0 + 4][B], av[M0 + 5][B], av[M0 + 6][B], av[M0 + 7][B], yc);
}
// the rows of class G that reach column C are the row sets M0 .. NR - 1 (row G + NW m >= C); in groups of at most 8
template <int D, int NW, int G, int C, int M0>
MHX_DEV void mhx_bcast_groups(const mhx_real (&av)[mhx_bcast_geom<D, NW>::NRMAX][mhx_bcast_geom<D, NW>::CHMAX], const mhx_real yc,
                              mhx_real (&w)[mhx_bcast_geom<D, NW>::NRMAX])
{
    constexpr int NR = (D - G + NW - 1) / NW;                 // rows of class G
    if constexpr (M0 < NR) {
        constexpr int N = NR - M0 < 8 ? NR - M0 : 8;
        mhx_bcast_group<C % 16, N, M0, C / 16>(w, av, yc);
        mhx_bcast_groups<D, NW, G, C, M0 + N>(av, yc, w);
    }
}
template <int D, int NW, int G, int C>
MHX_DEV void mhx_bcast_column(const mhx_real (&av)[mhx_bcast_geom<D, NW>::NRMAX][mhx_bcast_geom<D, NW>::CHMAX], const mhx_real (&y)[(D + 3) & ~3],
                              mhx_real (&w)[mhx_bcast_geom<D, NW>::NRMAX])
{
    if constexpr (C < D) {
        constexpr int M0 = C <= G ? 0 : (C - G + NW - 1) / NW;   // first row set whose row reaches column C
        mhx_bcast_groups<D, NW, G, C, M0>(av, y[C], w);
        mhx_bcast_column<D, NW, G, C + 1>(av, y, w);
    }
}
template <int D, int NW, int G>
MHX_DEV mhx_real mhx_bcast_rows_sq(const mhx_real (&av)[mhx_bcast_geom<D, NW>::NRMAX][mhx_bcast_geom<D, NW>::CHMAX], const mhx_real (&y)[(D + 3) & ~3])
{
    mhx_real w[mhx_bcast_geom<D, NW>::NRMAX];
#pragma unroll
    for (int m = 0; m < mhx_bcast_geom<D, NW>::NRMAX; ++m) w[m] = MHX_R(0.0);
    mhx_bcast_column<D, NW, G, 0>(av, y, w);
    mhx_real q = MHX_R(0.0);
#pragma unroll
    for (int m = 0; G + NW * m < D; ++m) q = mhx_fma(w[m], w[m], q);
    return q;
}
template <int D, int NW, int G = 0>
MHX_DEV mhx_real mhx_bcast_rows_dispatch(const int g, const mhx_real (&av)[mhx_bcast_geom<D, NW>::NRMAX][mhx_bcast_geom<D, NW>::CHMAX],
                                         const mhx_real (&y)[(D + 3) & ~3])
{
    if constexpr (G < NW) {
        if (g == G) return mhx_bcast_rows_sq<D, NW, G>(av, y);
        return mhx_bcast_rows_dispatch<D, NW, G + 1>(g, av, y);
    } else {
        return MHX_R(0.0);
    }
}

#ifndef MHX_EMCEE_SCAL_REC
#define MHX_EMCEE_SCAL_REC 1                 // how the record leaves: 0 straight from the move mapping (4-walker runs), 1 through the block's LDS (whole row segments)
#endif
#ifndef MHX_EMCEE_SCAL_MODE
#define MHX_EMCEE_SCAL_MODE 1                // how the wave-uniform factor reaches the products: 0 scalar loads -> SGPR operands, 1 vector loads -> DPP broadcast
#endif
template <int D, int NW>
MHX_DEV void mhx_emcee_scal_body(const mhx_emcee_args& a, const mhx_real* __restrict__ A, mhx_real* lds)
{
    typedef mhx_emcee_sgeom<D, NW> GEO;
    constexpr int XP = GEO::XP, NQ = GEO::NQ, NQL = GEO::NQL, YS = GEO::YS, WPB = GEO::WPB, LM = GEO::LM;
    mhx_real* ysh = lds;                         // [WPB][YS]
    mhx_real* qsh = lds + WPB * YS;              // [NW][64]
    const int tid = threadIdx.x;
    MHX_STAMP(0);
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);                  // this wave's row class in phase 2
    // ---- phase 1: LM lanes per walker.  A chain of latencies, so the loads go out in the order their addresses are known:
    // the walker's own row and lp at once, the partner's row as soon as the draw names it, then this wave's pieces of the factor
    // (needed only in phase 2: vector loads count in order, so they must follow the rows or the rows would wait for them too).
    const int wm = tid / LM, l = tid % LM;
    const int W = a.nwalkers;
    const int halfW = W / 2;
    const int lo = a.half ? halfW : 0;
    const int cnt = a.half ? W - halfW : halfW;
    const int t_raw = a.t_begin + blockIdx.x * WPB + wm;
    const bool valid = t_raw < cnt && t_raw < a.t_begin + a.t_count;
    const int i = lo + (valid ? t_raw : cnt - 1);
    const int ostart = a.half ? 0 : halfW;
    const int osize = a.half ? halfW : W - halfW;
    const long ld = a.ld;
    MHX_TP(1, (mhx_real)i);
    mhx_e4 xs[NQL], xjs[NQL], ysl[NQL];
    mhx_e4* xrow_i = (mhx_e4*)(a.xw + (long)i * mhx_xw_pitch(D));
    const mhx_e4 zero4 = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
#pragma unroll
    for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; xs[m] = q4 < NQ ? xrow_i[q4] : zero4; }
    const mhx_real lpi = a.lp[i];
    const mhx_u32 acc_i = a.acc_count[i];
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep);
    const int j = ostart + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)osize) >> 32);
    const mhx_e4* xrow_j = (const mhx_e4*)(a.xw + (long)j * mhx_xw_pitch(D));
#pragma unroll
    for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; xjs[m] = q4 < NQ ? xrow_j[q4] : zero4; }
    __builtin_amdgcn_sched_barrier(0);
    mhx_real av[mhx_bcast_geom<D, NW>::NRMAX][mhx_bcast_geom<D, NW>::CHMAX];
    if constexpr (MHX_EMCEE_SCAL_MODE == 1) mhx_bcast_load<D, NW>(g, A, tid & 15, av);   // in flight until phase 2
    __builtin_amdgcn_sched_barrier(0);
    const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), dr.u, MHX_R(1.0));
    const mhx_real z = (tt * tt) / a.stretch;                                // src/emcee.jl:81
    const mhx_real alphamult = (mhx_real)(D - 1) * mhx_log(z);               // :82
    MHX_TP(2, alphamult + (mhx_real)j);
    mhx_real* yrow = ysh + wm * YS;
#pragma unroll
    for (int m = 0; m < NQL; ++m) {
        const int q4 = l + LM * m;
        ysl[m] = zero4;
        if (q4 < NQ) {
            const mhx_e4 xi = xs[m];
            const mhx_e4 xj = xjs[m];
            ysl[m].x = mhx_fma(z, xi.x - xj.x, xj.x);                        // :85
            ysl[m].y = mhx_fma(z, xi.y - xj.y, xj.y);
            ysl[m].z = mhx_fma(z, xi.z - xj.z, xj.z);
            ysl[m].w = mhx_fma(z, xi.w - xj.w, xj.w);
            // (the row pitch is a multiple of 16 bytes, not of sizeof(e4) in fp64: two 16-byte halves there)
            if constexpr (sizeof(mhx_real) == 8) {
                mhx_e2 h0 = {ysl[m].x, ysl[m].y}, h1 = {ysl[m].z, ysl[m].w};
                ((mhx_e2*)(yrow + 4 * q4))[0] = h0;
                ((mhx_e2*)(yrow + 4 * q4))[1] = h1;
            } else {
                *(mhx_e4*)(yrow + 4 * q4) = ysl[m];
            }
        }
    }
    MHX_TP(3, ysl[0].x);
    MHX_STAMP(1);
    __syncthreads();
    MHX_STAMP(2);
    // ---- phase 2: lane = walker, wave = row class
    {
        const int wv = tid & 63;
        mhx_real y[XP];
        const mhx_real* yr = ysh + (wv < WPB ? wv : WPB - 1) * YS;     // (WPB < 64: the upper lanes idle along)
        if constexpr (sizeof(mhx_real) == 8) {
#pragma unroll
            for (int k = 0; k < XP / 2; ++k) { const mhx_e2 v = ((const mhx_e2*)yr)[k]; y[2 * k] = v.x; y[2 * k + 1] = v.y; }
        } else {
#pragma unroll
            for (int k = 0; k < XP / 4; ++k) { const mhx_e4 v = ((const mhx_e4*)yr)[k]; y[4 * k] = v.x; y[4 * k + 1] = v.y; y[4 * k + 2] = v.z; y[4 * k + 3] = v.w; }
        }
        MHX_TP(4, y[0]);
        MHX_STAMP(3);
        if constexpr (MHX_EMCEE_SCAL_MODE == 1) qsh[g * 64 + wv] = mhx_bcast_rows_dispatch<D, NW>(g, av, y);
        else qsh[g * 64 + wv] = mhx_scal_rows_dispatch<D, NW>(g, A, y);
    }
    MHX_STAMP(4);
    __syncthreads();
    MHX_STAMP(5);
    // ---- phase 3: back in the move mapping
    mhx_real qv[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) qv[k] = qsh[k * 64 + wm];
#pragma unroll
    for (int off = 1; off < NW; off <<= 1)
#pragma unroll
        for (int k = 0; k < NW; k += 2 * off) qv[k] = qv[k] + qv[k + off];          // the butterfly's tree
    const mhx_real lpy = mhx_fma(-MHX_R(0.5), qv[0], a.tconst);
    MHX_TP(5, lpy + ysl[0].x);
    const mhx_real alpha = (alphamult + lpy) - lpi;                          // :91
    const bool acc = dr.logu <= alpha;                                       // :93
#if MHX_EMCEE_SCAL_REC
    // the record leaves as whole row segments: the final rows (candidate or walker) meet in the block's LDS -- the candidate rows
    // are there already, a rejected move puts the walker back -- and every row k of the [dim+1][W] record is then written for the
    // block's WPB consecutive walkers at once (WPB * sizeof(real) contiguous bytes) instead of 4 walkers at a time
    if (valid && acc) {
#pragma unroll
        for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; if (q4 < NQ) xrow_i[q4] = ysl[m]; }
        if (l == 0) { a.lp[i] = lpy; a.acc_count[i] = acc_i + 1u; }
    }
    if (valid && l == 0) a.last_acc[i] = acc ? 1 : 0;
    MHX_STAMP(6);
    if (MHX_TP_IS(6)) return;
    if (a.save_slot >= 0) {                                                 // (uniform)
        if (!acc) {
#pragma unroll
            for (int m = 0; m < NQL; ++m) {
                const int q4 = l + LM * m;
                if (q4 < NQ) {
                    if constexpr (sizeof(mhx_real) == 8) {
                        mhx_e2 h0 = {xs[m].x, xs[m].y}, h1 = {xs[m].z, xs[m].w};
                        ((mhx_e2*)(yrow + 4 * q4))[0] = h0;
                        ((mhx_e2*)(yrow + 4 * q4))[1] = h1;
                    } else {
                        *(mhx_e4*)(yrow + 4 * q4) = xs[m];
                    }
                }
            }
        }
        if (l == 0) qsh[wm] = acc ? lpy : lpi;                              // (the partial sums have been read: phase 3 is past them)
        if (l == 1 || LM == 1) qsh[64 + wm] = acc ? MHX_R(1.0) : MHX_R(0.0);
        __syncthreads();
        const int wr = tid % WPB, ks = tid / WPB;                            // walker of the block, first row of this thread
        const int tw = a.t_begin + blockIdx.x * WPB + wr;
        if (tw < cnt && tw < a.t_begin + a.t_count) {
            mhx_real* col = a.samples + a.save_slot * (long)(D + 1) * ld + (lo + tw);
#pragma unroll
            for (int k = ks; k < D + 1; k += LM) MHX_REC_ST(&col[(long)k * ld], k < D ? ysh[wr * YS + k] : qsh[wr]);
            if (ks == LM - 1) a.accepted[a.save_slot * ld + lo + tw] = qsh[64 + wr] != MHX_R(0.0) ? 1 : 0;
        }
    }
#else
    if (valid) {
        if (acc) {
#pragma unroll
            for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; if (q4 < NQ) xrow_i[q4] = ysl[m]; }
            if (l == 0) { a.lp[i] = lpy; a.acc_count[i] = acc_i + 1u; }
        }
        if (l == 0) a.last_acc[i] = acc ? 1 : 0;
        MHX_STAMP(6);
        if (MHX_TP_IS(6)) return;
        if (a.save_slot >= 0) {
            mhx_real* row = a.samples + a.save_slot * (long)(D + 1) * ld + i;
#pragma unroll
            for (int m = 0; m < NQL; ++m) {
                const int k = 4 * (l + LM * m);
                const mhx_e4 v = acc ? ysl[m] : xs[m];
                if (k + 0 < D) MHX_REC_ST(&row[(long)(k + 0) * ld], v.x);
                if (k + 1 < D) MHX_REC_ST(&row[(long)(k + 1) * ld], v.y);
                if (k + 2 < D) MHX_REC_ST(&row[(long)(k + 2) * ld], v.z);
                if (k + 3 < D) MHX_REC_ST(&row[(long)(k + 3) * ld], v.w);
            }
            if (l == 0) {
                MHX_REC_ST(&row[(long)D * ld], acc ? lpy : lpi);
                a.accepted[a.save_slot * ld + i] = acc ? 1 : 0;
            }
        }
    }
#endif
    MHX_STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// The scalar-factor form as ONE LAUNCH PER SWEEP (see mhx_emcee_coop_sweep_body for the idea: the groups of the second half re-do
// their partners' moves from the old state; all three candidates of such a walker -- the partner's, and its own for both outcomes
// of the partner's accept test -- go through phase 2 side by side).  The row products are what this form is bound by in fp64 (a
// v_fmac_f64_dpp issues every 8 cycles, and phase 2 runs on 64 lanes whatever the number of candidate rows), so a block is MIXED:
// WPB / 2 walkers of the second half (three candidate rows each) and WPB / 2 of the first (one row) -- with WPB = 32 exactly 64 rows,
// every lane of phase 2 busy (the half-step kernel runs it half empty at this size), ONE pass.  Waves [0, NW / 2) hold the second
// half's walkers in the move mapping.  Rows in LDS: [0, HB) the partners' candidates, [HB, 2 HB) own against x_a, [2 HB, 3 HB) own
// against y_a, [3 HB, 4 HB) the first half's candidates.  State double-buffered as in the lane-group form (xw / lp read, xw_out /
// lp_out written; rows the other buffer already holds are not stored).  DPP operand mode, WPB = 32 (the latency shape) only.
template <int D, int NW>
MHX_DEV void mhx_emcee_scal_sweep_body(const mhx_emcee_args& a, const mhx_real* __restrict__ A, mhx_real* lds)
{
    typedef mhx_emcee_sgeom<D, NW> GEO;
    constexpr int XP = GEO::XP, NQ = GEO::NQ, NQL = GEO::NQL, YS = GEO::YS, WPB = GEO::WPB, LM = GEO::LM;
    constexpr int HB = WPB / 2;                  // walkers of each half per block
    static_assert(WPB == 32, "the mixed block fills the 64 lanes of phase 2 with 2 WPB candidate rows");
    mhx_real* ysh = lds;                         // [2 WPB][YS]
    mhx_real* qsh = lds + 2 * WPB * YS;          // [NW][64]
    mhx_real* fin = qsh + NW * 64;               // [2][WPB]: lp and accept flag of the final rows (the record's tail)
    const int tid = threadIdx.x;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);                  // this wave's row class in phase 2
    const int wm = tid / LM, l = tid % LM;
    const bool second = __builtin_amdgcn_readfirstlane(wm < HB ? 1 : 0) != 0;   // (wave-uniform: LM lanes per walker, 64 / LM walkers per wave)
    const int ws = second ? wm : wm - HB;                                    // walker of its half within the block
    const int W = a.nwalkers;
    const int halfW = W / 2, cntB = W - halfW;
    const int cnt = second ? cntB : halfW;
    const int t_raw = (int)blockIdx.x * HB + ws;
    const bool valid = t_raw < cnt;
    const int lo = second ? halfW : 0;
    const int i = lo + (valid ? t_raw : cnt - 1);
    const int r0 = second ? ws : 3 * HB + ws;                                // this walker's first candidate row
    const long ld = a.ld;
    const mhx_e4 zero4 = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
    // ---- phase 1 (LM lanes per walker): every address is a function of the counters -- own row, partner's row, and (second
    // half) the partner's partner's row go out before anything has come back; then this wave's pieces of the factor
    mhx_e4 xs[NQL], xjs[NQL], xbs[NQL], ysl[NQL], y1[NQL];
    const mhx_e4* xrow_i = (const mhx_e4*)(a.xw + (long)i * mhx_xw_pitch(D));
#pragma unroll
    for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; xs[m] = q4 < NQ ? xrow_i[q4] : zero4; }
    const mhx_real lpi = a.lp[i];
    const mhx_u32 acc_i = a.acc_count[i];
    const bool moved_before = a.all_rows != 0 || a.last_acc[i] != 0;         // xw_out does not hold this walker's row
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep);
    const int j = (second ? 0 : halfW) + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)(second ? halfW : cntB)) >> 32);
    const mhx_e4* xrow_j = (const mhx_e4*)(a.xw + (long)j * mhx_xw_pitch(D));
#pragma unroll
    for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; xjs[m] = q4 < NQ ? xrow_j[q4] : zero4; }
    mhx_emcee_draws da = dr;
    mhx_real lpa = MHX_R(0.0);
    if (second) {
        da = mhx_emcee_draw(ks, (mhx_u32)j, (mhx_u32)a.ensemble_id, a.sweep);   // j's own draws of this sweep (j is of the first half)
        const int jb = halfW + (int)(((mhx_u64)da.partner * (mhx_u64)(mhx_u32)cntB) >> 32);
        const mhx_e4* xrow_b = (const mhx_e4*)(a.xw + (long)jb * mhx_xw_pitch(D));
#pragma unroll
        for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; xbs[m] = q4 < NQ ? xrow_b[q4] : zero4; }
        lpa = a.lp[j];
    }
    MHX_TP(2, (mhx_real)j + xs[0].x + lpi + lpa + da.u);                  // launch, own row, the draws' integer part
    __builtin_amdgcn_sched_barrier(0);
    mhx_real av[mhx_bcast_geom<D, NW>::NRMAX][mhx_bcast_geom<D, NW>::CHMAX];
    mhx_bcast_load<D, NW>(g, A, tid & 15, av);                               // in flight until phase 2
    __builtin_amdgcn_sched_barrier(0);
    const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), dr.u, MHX_R(1.0));
    const mhx_real z = (tt * tt) / a.stretch;                                // src/emcee.jl:81
    const mhx_real alphamult = (mhx_real)(D - 1) * mhx_log(z);               // :82
    auto stretch = [](const mhx_real zz, const mhx_e4 xi, const mhx_e4 xj) {   // :85, element-wise
        mhx_e4 y;
        y.x = mhx_fma(zz, xi.x - xj.x, xj.x);
        y.y = mhx_fma(zz, xi.y - xj.y, xj.y);
        y.z = mhx_fma(zz, xi.z - xj.z, xj.z);
        y.w = mhx_fma(zz, xi.w - xj.w, xj.w);
        return y;
    };
    auto put = [](mhx_real* row, const int q4, const mhx_e4 v) {             // (the row pitch is a multiple of 16 bytes, not of sizeof(e4) in fp64)
        if constexpr (sizeof(mhx_real) == 8) {
            mhx_e2 h0 = {v.x, v.y}, h1 = {v.z, v.w};
            ((mhx_e2*)(row + 4 * q4))[0] = h0;
            ((mhx_e2*)(row + 4 * q4))[1] = h1;
        } else {
            *(mhx_e4*)(row + 4 * q4) = v;
        }
    };
    mhx_real* yrow = ysh + r0 * YS;
    mhx_real alphamult_a = MHX_R(0.0);
    if (!second) {
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int q4 = l + LM * m;
            ysl[m] = zero4; y1[m] = zero4;
            if (q4 < NQ) { ysl[m] = stretch(z, xs[m], xjs[m]); put(yrow, q4, ysl[m]); }
        }
    } else {
        const mhx_real ta = mhx_fma(a.stretch - MHX_R(1.0), da.u, MHX_R(1.0));
        const mhx_real za = (ta * ta) / a.stretch;
        alphamult_a = (mhx_real)(D - 1) * mhx_log(za);
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int q4 = l + LM * m;
            ysl[m] = zero4; y1[m] = zero4;
            if (q4 < NQ) {
                const mhx_e4 ya = stretch(za, xjs[m], xbs[m]);               // the partner's candidate
                ysl[m] = stretch(z, xs[m], xjs[m]);                          // this walker's, if the partner stays
                y1[m] = stretch(z, xs[m], ya);                               //                if it moves
                put(yrow, q4, ya);
                put(yrow + HB * YS, q4, ysl[m]);
                put(yrow + 2 * HB * YS, q4, y1[m]);
            }
        }
    }
    MHX_TP(3, ysl[0].x + y1[0].x + alphamult_a);                          // + the rows, the candidates in LDS
    __syncthreads();
    // ---- phase 2: lane = candidate row (all 64), wave = row class
    {
        const int wv = tid & 63;
        mhx_real y[XP];
        const mhx_real* yr = ysh + wv * YS;
        if constexpr (sizeof(mhx_real) == 8) {
#pragma unroll
            for (int k = 0; k < XP / 2; ++k) { const mhx_e2 v = ((const mhx_e2*)yr)[k]; y[2 * k] = v.x; y[2 * k + 1] = v.y; }
        } else {
#pragma unroll
            for (int k = 0; k < XP / 4; ++k) { const mhx_e4 v = ((const mhx_e4*)yr)[k]; y[4 * k] = v.x; y[4 * k + 1] = v.y; y[4 * k + 2] = v.z; y[4 * k + 3] = v.w; }
        }
        MHX_TP(4, y[0] + y[XP - 1]);                                      // + barrier, y in registers
        qsh[g * 64 + wv] = mhx_bcast_rows_dispatch<D, NW>(g, av, y);
    }
    __syncthreads();
    // ---- phase 3: back in the move mapping
    auto tree = [&](const int r) {                                           // the butterfly's tree over the NW partial sums of row r
        mhx_real qv[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) qv[k] = qsh[k * 64 + r];
#pragma unroll
        for (int off = 1; off < NW; off <<= 1)
#pragma unroll
            for (int k = 0; k < NW; k += 2 * off) qv[k] = qv[k] + qv[k + off];
        return qv[0];
    };
    mhx_real lpy = mhx_fma(-MHX_R(0.5), tree(r0), a.tconst);
    if (second) {
        const bool acc_a = da.logu <= (alphamult_a + lpy) - lpa;             // the partner's accept test, as its own group runs it
        lpy = mhx_fma(-MHX_R(0.5), acc_a ? tree(2 * HB + ws) : tree(HB + ws), a.tconst);
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            ysl[m].x = acc_a ? y1[m].x : ysl[m].x; ysl[m].y = acc_a ? y1[m].y : ysl[m].y;
            ysl[m].z = acc_a ? y1[m].z : ysl[m].z; ysl[m].w = acc_a ? y1[m].w : ysl[m].w;
        }
    }
    MHX_TP(5, lpy + ysl[0].x);                                            // + the row products, the trees
    const mhx_real alpha = (alphamult + lpy) - lpi;                          // :91
    const bool acc = dr.logu <= alpha;                                       // :93
#pragma unroll
    for (int m = 0; m < NQL; ++m) {
        ysl[m].x = acc ? ysl[m].x : xs[m].x; ysl[m].y = acc ? ysl[m].y : xs[m].y;
        ysl[m].z = acc ? ysl[m].z : xs[m].z; ysl[m].w = acc ? ysl[m].w : xs[m].w;
    }
    if (valid) {
        mhx_e4* xrow_o = (mhx_e4*)(a.xw_out + (long)i * mhx_xw_pitch(D));
        if (acc || moved_before) {
#pragma unroll
            for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; if (q4 < NQ) MHX_ROW_ST(&xrow_o[q4], ysl[m]); }
        }
        if (l == 0) {
            a.lp_out[i] = acc ? lpy : lpi;
            if (acc) a.acc_count[i] = acc_i + 1u;
            a.last_acc[i] = acc ? 1 : 0;
        }
    }
    if (MHX_TP_IS(6)) return;                                        // + accept and the new state, no record
    if (a.save_slot >= 0) {                                                 // (uniform)
        // the record leaves as whole row segments through the block's LDS: the final row replaces the walker's first candidate row
        // (phase 2 is past it), then every row k of the [dim+1][W] record is written for the HB consecutive walkers of each half
#pragma unroll
        for (int m = 0; m < NQL; ++m) { const int q4 = l + LM * m; if (q4 < NQ) put(yrow, q4, ysl[m]); }
        if (l == 0) fin[wm] = acc ? lpy : lpi;
        if (l == 1 || LM == 1) fin[WPB + wm] = acc ? MHX_R(1.0) : MHX_R(0.0);
        __syncthreads();
        const int wr = tid % WPB, kk = tid / WPB;                            // walker slot of the block, first row of this thread
        const bool sec = wr < HB;
        const int tw = (int)blockIdx.x * HB + (sec ? wr : wr - HB);
        if (tw < (sec ? cntB : halfW)) {
            const mhx_real* frow = ysh + (sec ? wr : 3 * HB + (wr - HB)) * YS;
            const long iw = (sec ? halfW : 0) + tw;
            mhx_real* col = a.samples + a.save_slot * (long)(D + 1) * ld + iw;
#pragma unroll
            for (int k = kk; k < D + 1; k += LM) MHX_REC_ST(&col[(long)k * ld], k < D ? frow[k] : fin[wr]);
            if (kk == LM - 1) a.accepted[a.save_slot * ld + iw] = fin[WPB + wr] != MHX_R(0.0) ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The reference's OWN sweep (src/emcee.jl:39-58): walkers move one after another, walker i pairs with idx = mod1(i + r, W),
// r uniform on 1..W-1, and uses the ALREADY UPDATED position when idx < i (:53) -- Gauss-Seidel, serial in W by
// construction.  One wave runs the whole schedule of a launch: lanes share the copy of a move's rows, lane 0 evaluates
// the log-density in the sequential order of the spec (every target, user sources included), all lanes take its
// decision.  It exists for fidelity, not speed (what Julia runs; MHX_FLAG_EMCEE_SEQUENTIAL): test/emcee.jl's 1000-walker
// ensemble is one block.  In-place update == the reference's new_walkers / walkers pair: entries below i are new.
template <int TK>
MHX_DEV void mhx_emcee_seq_body(const mhx_emcee_args& a, const mhx_real* __restrict__ tparams)
{
    const int lane = threadIdx.x & 63;
    const int W = a.nwalkers, d = a.dim;
    const long ld = a.ld;
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    for (int s = 0; s < a.nsweeps; ++s) {
        const mhx_u32 sweep = a.sweep + (mhx_u32)s;
        const bool rec = sweep == save_next;
        for (int i = 0; i < W; ++i) {
            const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, sweep);
            // src/emcee.jl:48,52  idx = mod1(i + rand(1:W-1), W): never i itself
            const mhx_u32 r = 1u + (mhx_u32)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)(W - 1)) >> 32);
            const int j = (int)(((mhx_u64)(mhx_u32)i + r) % (mhx_u64)(mhx_u32)W);
            const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), dr.u, MHX_R(1.0));
            const mhx_real z = (tt * tt) / a.stretch;                              // :81
            const mhx_real alphamult = (mhx_real)(d - 1) * mhx_log(z);             // :82
            mhx_real* ys = a.ybuf;                                                  // [dim] candidate of the current move
            for (int k = lane; k < d; k += 64) {
                const mhx_real xi = a.x[(long)k * ld + i], xj = a.x[(long)k * ld + j];
                ys[k] = mhx_fma(z, xi - xj, xj);                                    // :85
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            mhx_real lpy = MHX_R(0.0);
            if (lane == 0) {
                mhx_strided_x yv;
                yv.base = ys;
                yv.ld = 1;
                lpy = mhx_target_eval<TK>(a.target_kind, yv, d, tparams, a.ntparams, a.tconst);
            }
            lpy = mhx_readlane(lpy, 0);
            const mhx_real lpi = a.lp[i];
            const mhx_real alpha = (alphamult + lpy) - lpi;                         // :91
            const bool acc = dr.logu <= alpha;                                      // :93 (non-strict)
            if (acc) {
                for (int k = lane; k < d; k += 64) a.x[(long)k * ld + i] = ys[k];
                if (lane == 0) { a.lp[i] = lpy; a.acc_count[i] += 1u; }
            }
            if (lane == 0) a.last_acc[i] = acc ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        if (rec) {
            mhx_real* row = a.samples + slot * (long)(d + 1) * ld;
            for (long e = lane; e < (long)d * W; e += 64) { const long k = e / W, i = e - k * W; row[k * ld + i] = a.x[k * ld + i]; }
            for (int i = lane; i < W; i += 64) { row[(long)d * ld + i] = a.lp[i]; a.accepted[slot * ld + i] = a.last_acc[i]; }
            save_next += (mhx_u32)a.thinning;
            ++slot;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

#ifdef MHX_JIT_EMCEE
#ifndef MHX_JIT_SCAL
#define MHX_JIT_SCAL 0
#endif
#ifndef MHX_JIT_PRELOAD
#define MHX_JIT_PRELOAD 0
#endif
#if MHX_JIT_SCAL
extern "C" __global__ void __launch_bounds__(64 * MHX_JIT_L)
#elif MHX_JIT_L > 1
extern "C" __global__ void __launch_bounds__(64 * MHX_EMCEE_COOP_WAVES)
#else
extern "C" __global__ void __launch_bounds__(64)
#endif
#if MHX_JIT_PRELOAD
// kernarg preload (gfx950: the first user SGPRs of a wave are filled from the head of the kernarg segment by the dispatcher): the
// fields the half-step's first dependent chain starts from -- state rows, counters of the draw, the moving slice -- come FIRST and as
// plain scalars (a by-value struct is `byref` and cannot be preloaded), so no wave waits for an s_load of them; the rest of the
// argument block follows and is fetched as before.  The host passes the same values twice (emcee_launch_half).
mhx_jit_emcee_half(mhx_real* const xw, mhx_real* const lp, const mhx_u64 seed, const mhx_u64 ensemble_id, const mhx_u32 sweep,
                   const int half, const int nwalkers, const int t_begin, const int t_count,
                   const mhx_emcee_args a0, const mhx_real* __restrict__ tparams)
{
    mhx_emcee_args a = a0;
    a.xw = xw; a.lp = lp; a.seed = seed; a.ensemble_id = ensemble_id; a.sweep = sweep; a.half = half; a.nwalkers = nwalkers;
    a.t_begin = t_begin; a.t_count = t_count;
#else
mhx_jit_emcee_half(const mhx_emcee_args a_, const mhx_real* __restrict__ tparams)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
#endif
#if MHX_JIT_SCAL
    // the scalar-factor form: MHX_JIT_L waves per block = row classes = reduction shape; dynamic LDS = y rows + partial sums
    extern __shared__ mhx_e4 mhx_emcee_lds[];
    mhx_emcee_scal_body<MHX_JIT_DIM, MHX_JIT_L>(a, tparams, (mhx_real*)mhx_emcee_lds);
#elif MHX_JIT_L > 1
    // dynamic LDS (up to 160 KB per block on gfx950): [candidate rows][factor image]
    extern __shared__ mhx_e4 mhx_emcee_lds[];
    constexpr int YS4 = MHX_EMCEE_COOP_WAVES * (64 / MHX_JIT_L) * mhx_emcee_geom<MHX_JIT_DIM, MHX_JIT_L>::DP4 / 4;
#ifndef MHX_JIT_BW
#define MHX_JIT_BW -1
#endif
    mhx_emcee_coop_body<MHX_JIT_DIM, MHX_JIT_L, MHX_JIT_BW>(a, tparams, (mhx_real*)mhx_emcee_lds, mhx_emcee_lds + YS4);
#else
    mhx_emcee_half_body<MHX_JIT_DIM, MHX_JIT_TK>(a, tparams);
#endif
}
#if MHX_JIT_SCAL && MHX_EMCEE_SCAL_MODE == 1 && MHX_EMCEE_SCAL_WPB == 32
// one launch per sweep (scalar-factor form, mixed blocks): LDS = [64] candidate rows, [NW][64] partial sums, [2][WPB] lp and flag
extern "C" __global__ void __launch_bounds__(64 * MHX_JIT_L)
mhx_jit_emcee_sweep(const mhx_emcee_args a_, const mhx_real* __restrict__ tparams)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    extern __shared__ mhx_e4 mhx_emcee_lds[];
    mhx_emcee_scal_sweep_body<MHX_JIT_DIM, MHX_JIT_L>(a, tparams, (mhx_real*)mhx_emcee_lds);
}
#endif
#if !MHX_JIT_SCAL && MHX_JIT_L == 1 && MHX_JIT_DIM > 0
#ifdef MHX_JIT_PERSIST_THREADS
// a small ensemble as one persistent block (lane per walker, any target): dynamic LDS = [W][round4(D) + 1] reals
extern "C" __global__ void __launch_bounds__(MHX_JIT_PERSIST_THREADS)
mhx_jit_emcee_persist(const mhx_emcee_args a_, const mhx_real* __restrict__ tparams)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    extern __shared__ mhx_e4 mhx_emcee_lds[];
    mhx_emcee_persist_body<MHX_JIT_DIM, MHX_JIT_TK>(a, tparams, (mhx_real*)mhx_emcee_lds);
}
#endif
// one launch per sweep (lane per walker, any target)
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_emcee_sweep(const mhx_emcee_args a_, const mhx_real* __restrict__ tparams)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    mhx_emcee_sweep_reg_body<MHX_JIT_DIM, MHX_JIT_TK>(a, tparams);
}
#endif
#if !MHX_JIT_SCAL && MHX_JIT_L > 1
// one launch per sweep (lane-group form): three candidate rows per walker in LDS, then the factor image
extern "C" __global__ void __launch_bounds__(64 * MHX_EMCEE_COOP_WAVES)
mhx_jit_emcee_sweep(const mhx_emcee_args a_, const mhx_real* __restrict__ tparams)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    extern __shared__ mhx_e4 mhx_emcee_lds[];
    constexpr int YS4 = 3 * MHX_EMCEE_COOP_WAVES * (64 / MHX_JIT_L) * mhx_emcee_geom<MHX_JIT_DIM, MHX_JIT_L>::DP4 / 4;
    mhx_emcee_coop_sweep_body<MHX_JIT_DIM, MHX_JIT_L, MHX_JIT_BW>(a, tparams, (mhx_real*)mhx_emcee_lds, mhx_emcee_lds + YS4);
}
#endif
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_emcee_init(const mhx_emcee_args a_, const mhx_real* __restrict__ tparams, const int draw)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    mhx_emcee_init_body<MHX_JIT_TK>(a, tparams, draw);
}
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_emcee_seq(const mhx_emcee_args a_, const mhx_real* __restrict__ tparams)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    mhx_emcee_seq_body<MHX_JIT_TK>(a, tparams);
}
#endif
MHX_NS_END
