// mhx_rwmh_kernels.h -- random-walk Metropolis-Hastings, one wavefront lane per chain.
//
// Replaces the reference's per-chain step loop: AbstractMCMC.step (src/mh-core.jl:92-117) =
// propose (src/proposal.jl:49-56) -> logdensity (src/AdvancedMH.jl:74) -> accept test
// `-randexp(rng) < loga` (src/mh-core.jl:108), iterated by the upstream mcmcsample loop.
//
// Two kernels share the arithmetic:
//   mhx_rwmh_reg_kernel<D,...>  compile-time dimension; the chain's state x[D] and candidate y[D]
//                               live in VGPRs for a whole launch of `nsteps` transitions (the
//                               kernel is persistent over steps, not over chains); HBM sees only
//                               the sample records.  Built ahead of time for the headline shapes
//                               and specialised by hiprtc for any other (D, target, proposal).
//   mhx_rwmh_generic_kernel     run-time dimension; state in HBM as [dim][nchains] (chain
//                               fastest => every access is a coalesced 256 B row segment per
//                               wave), candidate staged in a [dim][nchains] scratch buffer.
// Both produce bit-identical chains: same Philox counters, same fmaf order.
#pragma once
#include "mhx_targets.h"

MHX_NS_BEGIN

#define MHX_PROP_ISO   0
#define MHX_PROP_DIAG  1
#define MHX_PROP_DENSE 2
#define MHX_PROP_DYNAMIC (-1)

#define MHX_NO_SAVE 0xffffffffu

// Cooperative kernel: blocks (4 dimensions each) a lane may own -- x and y are 8 reals per block, a double takes two
// VGPRs -- and the waves per SIMD its launch bound asks for (its point is >= 2 waves per SIMD)
#if MHX_REAL64
#define MHX_COOP_NBL_AUTO 7
#define MHX_COOP_NBL_MAX 13
#define MHX_COOP_WAVES(NBL) ((NBL) <= 2 ? 4 : ((NBL) <= 7 ? 2 : 1))
#else
#define MHX_COOP_NBL_AUTO 13
#define MHX_COOP_NBL_MAX 16
#define MHX_COOP_WAVES(NBL) ((NBL) <= 5 ? 4 : ((NBL) <= 13 ? 2 : 1))
#endif

struct mhx_rwmh_args {
    mhx_real* x;                 // [dim][ld]   chain state
    mhx_real* lp;                // [ld]
    mhx_u32* acc_count;       // [ld]        per-chain accepted proposals
    mhx_u64* acc_total;       // [1]         all-chain accepted proposals (ballot + popcount + 1 atomic / wave)
    mhx_real* samples;           // [slots][dim+1][ld] or null
    unsigned char* accepted;  // [slots][ld] or null
    unsigned char* last_acc;  // [ld]        accepted flag of each chain's latest transition
    mhx_real* ybuf;              // [dim][ld]   candidate scratch (generic kernel)
    mhx_u64 seed;
    mhx_u64 first_chain;
    int nchains;
    int ld;
    int dim;
    int target_kind;
    int ntparams;
    mhx_real tconst;
    int prop_kind;
    mhx_real pscale;
    mhx_u32 step0;            // first transition index of this launch (RNG step counter)
    int nsteps;
    mhx_u32 save_next;        // first transition >= step0 whose state is recorded (MHX_NO_SAVE: none)
    int save_slot;            // its slot in `samples`
    int thinning;
    int reduce_lanes;         // reduction shape of the separable targets (lanes per chain), >= 1
    // running moments instead of a sample tensor (runs too large to store): per chain and parameter the
    // Welford mean / M2 over the states the schedule would have recorded
    mhx_real* mom_mean;          // [dim+1][ld] or null
    mhx_real* mom_m2;            // [dim+1][ld]
    mhx_u32 mom_n0;           // states already folded in before this launch
    // drifting random walk (non-zero proposal mean; generic kernel only): mu[dim] followed by 2 L^-1 mu [dim]
    const mhx_real* pmean;       // null = zero mean (the Hastings ratio is then exactly 0 and is not computed)
    // static (independence) proposal, generic kernel only: q(x) = -1/2 |L^-1 (x - mu)|^2 of each chain's state
    mhx_real* qx;                // [ld] or null = random walk
    int tr_lds;                  // cooperative kernel, one or two chains per wave: the block has dim x (chains per block) reals of LDS to
                                 // move rows of the [dim][chains] arrays through (see mhx_rwmh_coop_body)
    int normal_gen;              // MHX_GEN_*: how stream bits become standard normals (a property of the run: initial draw and proposals)
};

// one Welford step with the wave-uniform 1/n
MHX_DEV void mhx_welford(mhx_real x, mhx_real rn, mhx_real& mean, mhx_real& m2)
{
    const mhx_real delta = x - mean;
    mean = mhx_fma(delta, rn, mean);
    m2 = mhx_fma(delta, x - mean, m2);
}

// ---------------------------------------------------------------------------------------------
// XR < D (round 4): the candidate y must be whole in a lane's registers (an arbitrary log-density reads all of it), the STATE need
// not be -- it is read once per step to form the candidate and written where a move is accepted.  Its first XR coordinates stay in
// registers, the rest live in the block's LDS as [k][lane] (one wave per block: conflict-free columns), so that x, y and the
// generator's temporaries fit 512 VGPRs without spilling: fp64 user targets of 80 < d <= 160 (fp32: 160 < d <= 320) run here
// instead of on the state-in-HBM kernel (d = 100, 65 536 chains, fp64: 4.9e8 -> see DESIGN 6.1).  Same arithmetic, same chains.
template <int D, int TK, int PK, int XR = D>
MHX_DEV void mhx_rwmh_reg_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ tparams,
                               const mhx_real* __restrict__ pvec, mhx_real* xl = nullptr)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (XR < D) xl += threadIdx.x;                  // this lane's column of the [D - XR][64] state tail
    if (c >= a.nchains) return;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;

    mhx_real x[XR > 0 ? XR : 1], y[D];
    auto getx = [&](const int k) -> mhx_real { return k < XR ? x[k < XR ? k : 0] : xl[(k - XR) * 64]; };
    const mhx_u32 cu = (mhx_u32)c * MHX_RB;  // row pointers are wave-uniform (scalar), the lane adds its byte offset
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const mhx_real v = mhx_ld_off(a.x + (long)k * ld, cu);
        if (k < XR) x[k < XR ? k : 0] = v; else xl[(k - XR) * 64] = v;
    }
    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;

    for (int i = 0; i < a.nsteps; ++i) {
        const mhx_u32 step = a.step0 + (mhx_u32)i;
        // ---- propose: y = x + L z   (src/proposal.jl:49-56; z from Philox stream PROPOSAL)
        if (PK == MHX_PROP_DENSE) {
            mhx_real z[D];
#pragma unroll
            for (int b = 0; b < (D + 3) / 4; ++b) {
                mhx_real n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (4 * b + j < D) z[4 * b + j] = n[j];
            }
            int off = 0;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                mhx_real w = MHX_R(0.0);
#pragma unroll
                for (int j = 0; j <= r; ++j) w = mhx_fma(pvec[off + j], z[j], w);
                y[r] = getx(r) + w;
                off += r + 1;
            }
        } else {
#pragma unroll
            for (int b = 0; b < (D + 3) / 4; ++b) {
                mhx_real n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * b + j;
                    if (k < D) y[k] = mhx_fma(PK == MHX_PROP_ISO ? a.pscale : pvec[k], n[j], getx(k));
                }
            }
        }
        // ---- log-density of the candidate and the accept test (src/mh-core.jl:103-108)
        const mhx_real lpy = mhx_target_eval<TK>(TK, y, D, tparams, a.ntparams, a.tconst);
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);          // strict; NaN compares false => reject
        // (fp64: the accepted lanes MOVE the candidate over the state under their execute mask -- where the state's registers are
        // AGPRs a select costs read + 2 v_cndmask + write per real, the move one write per word; fp32: a select is one instruction)
        if (MHX_REAL64) {
            if (acc) {
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    if (k < XR) x[k < XR ? k : 0] = y[k];
                    else xl[(k - XR) * 64] = y[k];
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < XR) x[k < XR ? k : 0] = acc ? y[k] : x[k < XR ? k : 0];
            else if (acc) xl[(k - XR) * 64] = y[k];
        }
        }
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc));
        // ---- record (ext/AdvancedMHMCMCChainsExt.jl:96-105 layout, chain fastest)
        if (step == save_next) {
            mhx_real* slotp = a.samples + slot * (long)(D + 1) * ld;
            const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(D + 1) * (mhx_u32)ld * MHX_RB);
            const mhx_u32 ldb = (mhx_u32)ld * MHX_RB;
            // (the row offset as a running scalar sum behind an opaque asm: see the record of mhx_rwmh_coop_body)
            mhx_u32 roff = 0u;
            asm volatile("" : "+s"(roff));
#pragma unroll
            for (int k = 0; k < D; ++k) { mhx_srd_store<MHX_REC_STORE_AUX>(srd, cu, roff, getx(k)); roff += ldb; }
            mhx_srd_store<MHX_REC_STORE_AUX>(srd, cu, roff, lp);
            a.accepted[slot * ld + c] = acc ? 1 : 0;
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) mhx_st_off(a.x + (long)k * ld, cu, getx(k));
    a.lp[c] = lp;
    a.acc_count[c] = nacc;
    a.last_acc[c] = last ? 1 : 0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// ---------------------------------------------------------------------------------------------
template <int TK>
MHX_DEV void mhx_rwmh_generic_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ tparams,
                                   const mhx_real* __restrict__ pvec)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;
    const int d = a.dim;
    mhx_real* xs = a.x + c;
    mhx_real* ys = a.ybuf + c;

    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    mhx_u32 mom_n = a.mom_n0;
    const int nblk = (d + 3) >> 2;
    mhx_real qxc = a.qx ? a.qx[c] : MHX_R(0.0);

    for (int i = 0; i < a.nsteps; ++i) {
        const mhx_u32 step = a.step0 + (mhx_u32)i;
        // a drifting walk (proposal mean mu != 0) keeps |z|^2 and |z + 2 L^-1 mu|^2 for its Hastings ratio
        const mhx_real* mu = a.pmean;
        const bool stat = a.qx != nullptr;             // StaticProposal: the candidate ignores x (src/proposal.jl:66-72)
        mhx_real fwd = MHX_R(0.0), bwd = MHX_R(0.0);
        if (a.prop_kind == MHX_PROP_DENSE) {
            for (int b = 0; b < nblk; ++b) {
                mhx_real n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * b + j;
                    if (k < d) {
                        ys[(long)k * ld] = n[j];
                        if (mu || stat) fwd = mhx_fma(n[j], n[j], fwd);
                        if (mu && !stat) { const mhx_real tk = n[j] + mu[d + k]; bwd = mhx_fma(tk, tk, bwd); }
                    }
                }
            }
            // y_r = x_r + sum_{j<=r} L_rj z_j ; rows in descending order so z can be overwritten in place
            for (int r = d - 1; r >= 0; --r) {
                const mhx_real* Lr = pvec + (long)r * (r + 1) / 2;
                mhx_real w = MHX_R(0.0);
                for (int j = 0; j <= r; ++j) w = mhx_fma(Lr[j], ys[(long)j * ld], w);
                const mhx_real xr = stat ? MHX_R(0.0) : xs[(long)r * ld];
                ys[(long)r * ld] = mu ? xr + (mu[r] + w) : xr + w;
            }
        } else {
            for (int b = 0; b < nblk; ++b) {
                mhx_real n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * b + j;
                    if (k < d) {
                        const mhx_real s = a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k];
                        const mhx_real xk = stat ? MHX_R(0.0) : xs[(long)k * ld];
                        if (mu) {
                            ys[(long)k * ld] = xk + mhx_fma(s, n[j], mu[k]);
                            fwd = mhx_fma(n[j], n[j], fwd);
                            const mhx_real tk = n[j] + mu[d + k];
                            bwd = mhx_fma(tk, tk, bwd);
                        } else {
                            ys[(long)k * ld] = mhx_fma(s, n[j], xk);
                            if (stat) fwd = mhx_fma(n[j], n[j], fwd);
                        }
                    }
                }
            }
        }
        mhx_strided_x yv;
        yv.base = ys;
        yv.ld = ld;
        const mhx_real lpy = mhx_target_eval<TK>(a.target_kind, yv, d, tparams, a.ntparams, a.tconst);
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        // src/mh-core.jl:104-105 with logratio_proposal_density (src/proposal.jl:190-192) when the walk drifts
        // static proposal: logpdf(p, x) - logpdf(p, y) (src/proposal.jl:74-83), q(y) = -1/2 |z|^2
        const mhx_real qy = -MHX_R(0.5) * fwd;
        const mhx_real loga = stat ? (lpy - lp) + (qxc - qy) : (mu ? (lpy - lp) + MHX_R(0.5) * (fwd - bwd) : (lpy - lp));
        const bool acc = logu < loga;
        lp = acc ? lpy : lp;
        qxc = (stat && acc) ? qy : qxc;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc));
        if (a.mom_mean && step == save_next) {
            // running moments kept in HBM (this kernel's state lives there anyway)
            ++mom_n;
            const mhx_real rn = MHX_R(1.0) / (mhx_real)mom_n;
            const bool first = mom_n == 1u;
            for (int k = 0; k <= d; ++k) {
                mhx_real v;
                if (k < d) {
                    v = acc ? ys[(long)k * ld] : xs[(long)k * ld];
                    if (acc) xs[(long)k * ld] = v;
                } else {
                    v = lp;
                }
                mhx_real mean = first ? MHX_R(0.0) : a.mom_mean[(long)k * ld + c];
                mhx_real m2v = first ? MHX_R(0.0) : a.mom_m2[(long)k * ld + c];
                mhx_welford(v, rn, mean, m2v);
                a.mom_mean[(long)k * ld + c] = mean;
                a.mom_m2[(long)k * ld + c] = m2v;
            }
            save_next += (mhx_u32)a.thinning;
        } else if (step == save_next) {
            mhx_real* row = a.samples + slot * (long)(d + 1) * ld + c;
            for (int k = 0; k < d; ++k) {
                const mhx_real v = acc ? ys[(long)k * ld] : xs[(long)k * ld];
                if (acc) xs[(long)k * ld] = v;
                row[(long)k * ld] = v;
            }
            row[(long)d * ld] = lp;
            a.accepted[slot * ld + c] = acc ? 1 : 0;
            save_next += (mhx_u32)a.thinning;
            ++slot;
        } else if (acc) {
            for (int k = 0; k < d; ++k) xs[(long)k * ld] = ys[(long)k * ld];
        }
    }
    a.lp[c] = lp;
    if (a.qx) a.qx[c] = qxc;
    a.acc_count[c] = nacc;
    a.last_acc[c] = last ? 1 : 0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// q(x) = -1/2 |L^-1 (x - mu)|^2 of every chain's current state: the static proposal's logpdf up to its
// constant (src/proposal.jl:31-35); forward substitution with the whitened vector in the scratch slab
MHX_DEV void mhx_rwmh_whiten_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ pvec)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const long ld = a.ld;
    const int d = a.dim;
    const mhx_real* xs = a.x + c;
    mhx_real* ts = a.ybuf + c;
    const mhx_real* mu = a.pmean;
    mhx_real q = MHX_R(0.0);
    for (int i = 0; i < d; ++i) {
        const mhx_real r = mu ? xs[(long)i * ld] - mu[i] : xs[(long)i * ld];
        mhx_real t;
        if (a.prop_kind == MHX_PROP_ISO) t = r / a.pscale;
        else if (a.prop_kind == MHX_PROP_DIAG) t = r / pvec[i];
        else {
            const mhx_real* Li = pvec + (long)i * (i + 1) / 2;
            mhx_real acc = MHX_R(0.0);
            for (int j = 0; j < i; ++j) acc = mhx_fma(Li[j], ts[(long)j * ld], acc);
            t = (r - acc) / Li[i];
        }
        ts[(long)i * ld] = t;
        q = mhx_fma(t, t, q);
    }
    a.qx[c] = -MHX_R(0.5) * q;
}

// ---------------------------------------------------------------------------------------------
// Cooperative kernel: L lanes share one chain (L = 2 .. 64, a power of two; 64/L chains per wave).
// Lane l owns the Philox blocks b = l, l+L, ... (4 dimensions each), i.e. NBL blocks of state
// x[NBL][4] and candidate y[NBL][4] in VGPRs.  Used for the separable catalogue targets
// (iso-Gaussian, banana, funnel), whose log-density is a sum over dimensions: every lane reduces its
// own blocks sequentially, the L partial sums meet in an xor-butterfly (offsets 1, 2, 4, ... in
// units of 64/L lanes) and every lane of the chain ends up with the same lp and takes the same
// accept decision.  Two regimes:
//   - few chains, moderate d (65 536 x 100): L = 2 doubles the wave count to 2 per SIMD, which is
//     what a CDNA4 SIMD needs to reach its VALU issue rate (one wave alone issues every ~5 cycles);
//   - large d (1000): L = 64 is wave-per-chain, the whole state (4 VGPRs per 256 dimensions) stays
//     in registers across a launch and nothing is streamed from HBM but the recorded samples.
// The reduction shape L is part of the arithmetic spec (the oracle takes the same L).
// WALK: 0 = zero-mean random walk (Hastings ratio exactly 0); 1 = drifting walk (proposal mean mu != 0:
// y = x + (mu + s n), logratio = 1/2 sum n^2 - 1/2 sum (n + t)^2, t = 2 L^-1 mu; src/proposal.jl:58-64,190-192);
// 2 = static proposal (independence sampler, src/proposal.jl:9-11,66-83: y = mu + s n whatever x is,
// logratio = q(x) - q(y), q(y) = -1/2 sum n^2, q(x) one more real of chain state).  The two sums of a drifting / static
// step are separable like the target's: per-lane partial sums in block order, the same butterfly (reduction shape L).
#define MHX_WALK_PLAIN  0
#define MHX_WALK_DRIFT  1
#define MHX_WALK_STATIC 2
// Dynamic LDS of the cooperative kernel with the ziggurat generator (GEN = MHX_GEN_ZIGGURAT), 256-thread blocks (sizes in the engine's
// own width: fp64 1024 layers and 8-byte normals, fp32 -- round 6 -- 256 layers and 4-byte normals):
//   [0, table bytes)                the layer table: fp64 x[0..N]; fp32 the signed pairs (+-x[l], x[l + 1]) a candidate's nine top bits index
//   then per wave  NBL*4*64 reals   the step's normals -- fp64: [pair of slots][lane][2], fp32: [block][lane][4]: a lane writes / reads
//                                   16 bytes, conflict-free
//                  64 u16           the queue of the candidates that left their rectangles: owner lane | slot << 6
//                  KS x 64 u64      the lanes' masks of failed slots, one per step of the group
// KS (round 4): the normals of KS consecutive steps are generated back to back and their failed candidates finished in ONE pass --
// the fix-up pass costs a wave what it costs whether it repairs 4 candidates or 40 (the queue, two wave syncs, one walk through the
// rejection code), and at d = 1000 (a wave per chain, 16 slots per lane: 4 failures per wave-step) it was a third of the kernel.
// KS is the largest group (<= 4) whose slabs leave the LDS for as many blocks per CU as the launch bound asks for.
#if MHX_REAL64
#define MHX_ZIG_TABLE_BYTES (((MHX_ZIG_N + 1) * (int)sizeof(mhx_real) + 15) / 16 * 16)
#else
#define MHX_ZIG_TABLE_BYTES (MHX_ZIG_PAIR_FLOATS * 4)       // the signed pair table (mhx_device_math.h): 2 x 256 x 8 bytes
#endif
#define MHX_ZIG_TABLE_BYTES_ANY MHX_ZIG_TABLE_BYTES
#define MHX_ZIG_SLAB_BYTES(NBL) ((NBL) * 4 * 64 * (int)sizeof(mhx_real))
#define MHX_ZIG_KS_FIT(NBL) ((163840 / MHX_COOP_WAVES(NBL) - MHX_ZIG_TABLE_BYTES - 512) / (4 * (MHX_ZIG_SLAB_BYTES(NBL) + 512 + 32)))
#define MHX_ZIG_KS(NBL) (MHX_ZIG_KS_FIT(NBL) < 1 ? 1 : (MHX_ZIG_KS_FIT(NBL) > 4 ? 4 : MHX_ZIG_KS_FIT(NBL)))
#define MHX_ZIG_WAVE_BYTES(NBL) (MHX_ZIG_KS(NBL) * (MHX_ZIG_SLAB_BYTES(NBL) + 512) + 128)
#define MHX_ZIG_LDS_BYTES(NBL) (MHX_ZIG_TABLE_BYTES + 4 * MHX_ZIG_WAVE_BYTES(NBL))
// where slot sl (= 4 i + j: block i of the lane, normal j of the block) of lane `ln` lives in a step's slab of normals
#if MHX_REAL64
#define MHX_ZIG_SLAB_AT(sl, ln) (((((sl) >> 1) * 64 + (ln)) << 1) + ((sl) & 1))
#else
#define MHX_ZIG_SLAB_AT(sl, ln) (((((sl) >> 2) * 64 + (ln)) << 2) + ((sl) & 3))
#endif

#if MHX_REAL64
// (experiment, MHX_ZADDC: the fp64 fast path noting its failures like the fp32 one -- compare + add-with-carry into four mask words)
#ifndef MHX_ZADDC
#define MHX_ZADDC 1
#endif
#define MHX_ZIG_NOTE64(m, ax, hi, slot) asm("v_cmp_nlt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(ax), "v"(hi) : "vcc")
#ifdef MHX_TOOLS_BUILD
#ifdef MHX_ZIG_FORCE_FAIL
#undef MHX_ZIG_NOTE64
#define MHX_ZIG_NOTE64(m, ax, hi, slot) m = ((m) << 1) | ((!((ax) < (hi)) || (((slot) + lane) % (MHX_ZIG_FORCE_FAIL) == 0)) ? 1u : 0u)
#endif
#endif
#endif
#if !MHX_REAL64
// fp32: "candidate x left its rectangle" noted by shifting the compare's carry into a mask word: `m = m + m + (|x| >= hi)` -- two
// instructions (compare with the magnitude as an operand modifier, add-with-carry); the bit of the FIRST candidate noted ends up highest
#define MHX_ZIG_NOTE(m, x, hi, slot) asm("v_cmp_nlt_f32 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(x), "v"(hi) : "vcc")
#ifdef MHX_TOOLS_BUILD
#ifdef MHX_ZIG_FORCE_FAIL
#undef MHX_ZIG_NOTE
#define MHX_ZIG_NOTE(m, x, hi, slot) m = ((m) << 1) | ((!(__builtin_fabsf(x) < (hi)) || (((slot) + lane) % (MHX_ZIG_FORCE_FAIL) == 0)) ? 1u : 0u)
#endif
#endif
// a candidate's table pair read at its LDS byte address itself: the signed pair table OPENS the kernel's LDS (no static LDS in the
// kernels that use it; each checks).  Through a generic pointer hipcc adds the array's base, a literal 0 it learns too late to fold.
typedef float mhx_f2v __attribute__((ext_vector_type(2)));
typedef float mhx_f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) mhx_f2v* mhx_lds_f2v;
#define MHX_ZIG_PAIR_OF(w) (*(mhx_lds_f2v)(size_t)(((w) >> 20) & 0xff8u))
#endif

// The candidates of this wave-step that left their rectangles (0.4 % of the draws: a dozen per wave-step at d = 100), gathered
// from all lanes into one queue and finished by as many lanes side by side -- ONE pass of the slow path per wave-step instead of
// one per failing block.  fm: the lane's failed slots (slot s = 4 i + j at bit s).  A fixer lane re-derives the failed candidate from
// its Philox block (nothing but the slot number was kept), runs the rejection loop of mhx_zig_slow and drops the normal into the
// owner's place in `zn`.
// (a group of `ng` steps: zfm[s][lane] = the lane's failed slots of step step0 + s, whose normals live in zn + s * slabd)
// (Measured and removed in round 5, profiles/r04y_fastpath_ab.log: the failure bits from the SIGN of |x| - x[layer + 1] shifted in by
// one v_alignbit_b32 per candidate -- 47 instructions fewer per wave-step and SLOWER, one dependent chain per candidate.)
template <int L>
MHX_DEV void mhx_zig_fixup(const mhx_philox_key& ks, const mhx_real* __restrict__ zt, mhx_real* __restrict__ zn,
                           unsigned short* __restrict__ zq, const mhx_u64* __restrict__ zfm, const int ng, const int slabd,
                           const int lane, const long wave,
                           const mhx_u64 first_chain, const int nchains, const mhx_u32 step0, const mhx_u32 stream,
                           const bool fm_in_reg = false, const mhx_u64 fm_reg = 0ull, const int rev_top = -1)
{
    // rev_top >= 0 (the fp32 kernel's masks): bit p of a mask is slot 4 i + j with j = p >> 4, i = rev_top - (p & 15) -- four
    // 16-bit fields, one per word of a Philox block, each shifted up by one per block; rev_top < 0: bit p is slot p
    constexpr int CPW = 64 / L;
    // Queue positions without a prefix sum over the lanes: round k takes the k-th failure of every lane that has one; the
    // lanes of a round are ranked by mbcnt over the round's ballot, the rounds follow each other in the queue.  (Nearly all
    // lanes have 0 or 1 failure: one or two rounds.)  One pass both numbers and stores the first 64 entries -- all of them
    // unless more than 64 candidates of the wave-step failed, in which case further windows of 64 follow.
    int total = 0;
    for (int win = 0; win == 0 || win < total; win += 64) {
        int base = -win;
        for (int s = 0; s < ng; ++s) {
            mhx_u64 f = fm_in_reg ? fm_reg : zfm[s * 64 + lane];      // (a group of one step: the mask never went to LDS)
            for (;;) {
                const mhx_u64 m = __ballot(f != 0ull);
                if (m == 0ull) break;
                if (f != 0ull) {
                    const int e = base + (int)__builtin_amdgcn_mbcnt_hi((mhx_u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((mhx_u32)m, 0u));
                    const int p = __ffsll((long long)f) - 1;
                    const int sl = rev_top < 0 ? p : 4 * (rev_top - (p & 15)) + (p >> 4);
                    if (e >= 0 && e < 64) zq[e] = (unsigned short)(lane | (sl << 6) | (s << 12));
                }
                f &= f - 1ull;
                base += __popcll(m);
            }
        }
        total = base + win;
        MHX_WAVE_SYNC();
        const int nent = total - win < 64 ? total - win : 64;
        if (lane < nent) {
            const int ent = zq[lane];
            const int ol = ent & 63, sl = (ent >> 6) & 63, sg = ent >> 12;
            const mhx_u32 step = step0 + (mhx_u32)sg;
            mhx_real* const zns = zn + sg * slabd;
            const long oc_raw = wave * CPW + (ol & (CPW - 1));
            const mhx_u64 oid = first_chain + (mhx_u64)(oc_raw < nchains ? oc_raw : (long)nchains - 1);
            const mhx_u32 b = (mhx_u32)(ol / CPW + L * (sl >> 2));
            const mhx_u32 n = 4u * b + (mhx_u32)(sl & 3);
            // nothing but the slot number was kept: the failed candidate is re-derived from its Philox block
            zns[MHX_ZIG_SLAB_AT(sl, ol)] = mhx_zig_refine(ks, zt, (mhx_u32)oid, (mhx_u32)(oid >> 32), step, stream, n);
        }
        MHX_WAVE_SYNC();
    }
}

// The D standard normals of (chain, step, stream) by the table ziggurat into a lane's REGISTER array y (lane per chain, every lane of
// the wave alive): phase A writes every fast-path normal into y[n] and notes the failures in a per-lane bit mask; the failures of
// the whole wave-step are queued, refined side by side by as many lanes (which leave the values in LDS by queue position), and every
// owner walks its failures once more and takes the round's value where the slot number says so.  Used by the register kernels of
// RWMH (mhx_rwmh_reg_zig_body) and MALA (mhx_mala_reg_body<.., ZIG>): the normals of mhx_zig_normal / orc_zig_normal.  Both widths
// (fp32 since round 6: a Philox block serves four normals, the signed pair table, failures noted by add-with-carry).
// zt: the layer table in LDS (offset 0), zq: 64 queue entries, zres: 64 refined values; chain_base: the chain of lane 0.
#define MHX_REG_ZIG_HDR_BYTES (MHX_ZIG_TABLE_BYTES + 128 + 64 * (int)sizeof(mhx_real))      // [table][queue][results]
// SLAB (fp32, where the LDS has room for [blocks][64][4] floats per wave): the step's normals meet in a slab as in the cooperative
// kernel -- the fast path writes a Philox block's four normals as one 16-byte store, a refined value is dropped into its owner's
// place, and the lane reads its column back when the queue is empty -- instead of the hand-back walk (compare + select per register
// and round: at fp32's failure rate, 1.5 % of 6 400 candidates = two queue windows of five rounds, it cost more than the fast path).
#define MHX_REG_ZIG_SLAB_BYTES(D) ((size_t)(((D) + 3) / 4) * 64 * 4 * sizeof(mhx_real))
template <int D, bool SLAB = false>
MHX_DEV void mhx_reg_zig_fill(mhx_real (&y)[D], const mhx_philox_key& ks, const mhx_u32 id_lo, const mhx_u32 id_hi, const mhx_u32 step,
                              const mhx_u32 stream, const mhx_real* __restrict__ zt, unsigned short* __restrict__ zq,
                              mhx_real* __restrict__ zres, const int lane, const long chain_base, const mhx_u64 first_chain,
                              const int nchains, const mhx_u32 zsign, mhx_real* __restrict__ zslab = nullptr)
{
    static_assert(!SLAB || !MHX_REAL64, "the slab form is fp32's");
    constexpr int NW = (D + 63) / 64;                          // words of the failure mask
    // ---- phase A: the fast path; failures noted
    // (the chain id behind an empty asm: otherwise hipcc hoists the id-only first round and a half of all Philox calls out
    // of the step loop -- 150 registers it does not have: they go to scratch memory and come back every step)
    mhx_u32 idl = id_lo, idh = id_hi;
    asm volatile("" : "+v"(idl), "+v"(idh));
    mhx_u64 fmw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) fmw[w] = 0ull;
    // Software pipeline over the Philox blocks: the table look-ups of block p are in flight while the rounds of block p + 1 run
    // (one wave per SIMD has nothing else to hide an LDS round trip behind; straight after each other the look-ups cost a
    // quarter of the kernel: 100 x ~100 cycles per wave-step)
#if MHX_REAL64
    // normal n of the step from Philox block n >> 1 (words x, y / z, w)
    constexpr int NP = (D + 1) / 2;
    mhx_u32x4 w4 = mhx_philox(ks, idl, idh, step, (stream << 28) | 0u);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        typedef double mhx_d2 __attribute__((ext_vector_type(2)));
        mhx_d2 xe[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const mhx_u32 ly = (h ? w4.w : w4.y) & (mhx_u32)(MHX_ZIG_N - 1);
            xe[h].x = zt[ly]; xe[h].y = zt[ly + 1];
        }
        __builtin_amdgcn_sched_barrier(0);
        mhx_u32x4 wn = w4;
        if (p + 1 < NP) wn = mhx_philox(ks, idl, idh, step, (stream << 28) | (mhx_u32)(p + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = 2 * p + h;
            if (n < D) {
                const mhx_u32 hi = h ? w4.z : w4.x, lo = h ? w4.w : w4.y;
                const double ax = mhx_zig_ax(hi, lo, xe[h].x);
                y[n] = mhx_zig_signed(ax, lo, zsign);
                const bool fail = !(ax < xe[h].y);
                fmw[n >> 6] |= (fail ? 1ull : 0ull) << (n & 63);
            }
        }
        w4 = wn;
    }
#else
    // normal n of the step from word n & 3 of Philox block n >> 2; 32 consecutive slots share a mask word (first slot highest)
    constexpr int NP = (D + 3) / 4, NM = (D + 31) / 32;
    mhx_u32 m32[NM];
#pragma unroll
    for (int w = 0; w < NM; ++w) m32[w] = 0u;
    mhx_u32x4 w4 = mhx_philox(ks, idl, idh, step, (stream << 28) | 0u);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const mhx_u32 kw[4] = {w4.x, w4.y, w4.z, w4.w};
        mhx_f2v xe[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xe[e] = MHX_ZIG_PAIR_OF(kw[e]);
        __builtin_amdgcn_sched_barrier(0);
        mhx_u32x4 wn = w4;
        if (p + 1 < NP) wn = mhx_philox(ks, idl, idh, step, (stream << 28) | (mhx_u32)(p + 1));
        __builtin_amdgcn_sched_barrier(0);
        float nn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * p + j;
            if (n < D) {
                nn[j] = mhx_zig_ax(kw[j], xe[j].x);            // signed: the table entry carries the sign
                MHX_ZIG_NOTE(m32[n >> 5], nn[j], xe[j].y, n);
                if (!SLAB) y[n] = nn[j];
            }
        }
        if (SLAB) {
            mhx_f4v v4;
            v4.x = nn[0]; v4.y = nn[1]; v4.z = nn[2]; v4.w = nn[3];
            *(mhx_f4v*)(zslab + ((p * 64 + lane) << 2)) = v4;
        }
        w4 = wn;
    }
#pragma unroll
    for (int w = 0; w < NM; ++w) {                             // slot n at bit n & 63 of word n >> 6, as the walks below expect
        const int cnt = D - 32 * w < 32 ? D - 32 * w : 32;
        const mhx_u32 nat = __builtin_bitreverse32(m32[w]) >> (32 - cnt);
        fmw[w >> 1] |= (mhx_u64)nat << (32 * (w & 1));
    }
    (void)zsign;
#endif
    // ---- the wave-step's failures: queue, refine side by side, hand back
    bool anyfail = false;
#pragma unroll
    for (int w = 0; w < NW; ++w) anyfail = anyfail || fmw[w] != 0ull;
    if (__ballot(anyfail)) {
        int total = 0;
        for (int win = 0; win == 0 || win < total; win += 64) {
            int base = -win;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                mhx_u64 f = fmw[w];
                for (;;) {
                    const mhx_u64 m = __ballot(f != 0ull);
                    if (m == 0ull) break;
                    if (f != 0ull) {
                        const int e = base + (int)__builtin_amdgcn_mbcnt_hi((mhx_u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((mhx_u32)m, 0u));
                        const int sl = 64 * w + __ffsll((long long)f) - 1;
                        if (e >= 0 && e < 64) zq[e] = (unsigned short)(lane | (sl << 6));
                    }
                    f &= f - 1ull;
                    base += __popcll(m);
                }
            }
            total = base + win;
            MHX_WAVE_SYNC();
            const int nent = total - win < 64 ? total - win : 64;
            mhx_u32 ent = 0u;
            mhx_real val = MHX_R(0.0);
            if (lane < nent) {
                ent = zq[lane];
                const int ol = (int)(ent & 63u);
                const long oc_raw = chain_base + ol;
                const mhx_u64 oid = first_chain + (mhx_u64)(oc_raw < nchains ? oc_raw : (long)nchains - 1);
                val = mhx_zig_refine(ks, zt, (mhx_u32)oid, (mhx_u32)(oid >> 32), step, stream, ent >> 6);
            }
            if (SLAB) {
                // into the owner's place of the slab; the next window (if any) rewrites the queue
                if (lane < nent) zslab[((((ent >> 6) >> 2) * 64 + (ent & 63u)) << 2) + ((ent >> 6) & 3u)] = val;
                MHX_WAVE_SYNC();
                continue;
            }
            if (lane < nent) zres[lane] = val;
            MHX_WAVE_SYNC();
            // back to the owners: the same walk as the queue's -- a lane's r-th failure of word w sits at the position it was
            // given there -- and every register of the word takes the value where the slot number says so: straight-line
            // compare + select per register and round (a wave-uniform branch per failure into one of D registers costs
            // hipcc's allocator a copy of the whole candidate per iteration; a run-time index sends the array to scratch)
            base = -win;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                mhx_u64 f = fmw[w];
                for (;;) {
                    const mhx_u64 m = __ballot(f != 0ull);
                    if (m == 0ull) break;
                    const int e = base + (int)__builtin_amdgcn_mbcnt_hi((mhx_u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((mhx_u32)m, 0u));
                    const bool has = f != 0ull && e >= 0 && e < 64;
                    int sls = has ? __ffsll((long long)f) - 1 : -1;
                    asm volatile("" : "+v"(sls));          // (else hipcc splits the test into `has` AND a compare: 5 instructions per register for 3)
                    const mhx_real pv = zres[has ? e : 0];
                    // (skipping a block of 8 registers when no lane's slot lies in it was measured: 3.35 against 3.25 ms per launch at
                    // c2_user -- the ballots and branches cost more than the skipped selects)
#pragma unroll
                    for (int blk = 0; blk < 8; ++blk) {
                        if (64 * w + 8 * blk < D) {
#pragma unroll
                            for (int b = 8 * blk; b < 8 * blk + 8; ++b)
                                if (64 * w + b < D) y[64 * w + b] = sls == b ? pv : y[64 * w + b];
                        }
                    }
                    f &= f - 1ull;
                    base += __popcll(m);
                }
            }
            MHX_WAVE_SYNC();                               // (the next window writes the queue and the results again)
        }
    }
#if !MHX_REAL64
    if (SLAB) {
        MHX_WAVE_SYNC();
#pragma unroll
        for (int p = 0; p < (D + 3) / 4; ++p) {
            const mhx_f4v v = *(const mhx_f4v*)(zslab + ((p * 64 + lane) << 2));
            if (4 * p < D) y[4 * p] = v.x;
            if (4 * p + 1 < D) y[4 * p + 1] = v.y;
            if (4 * p + 2 < D) y[4 * p + 2] = v.z;
            if (4 * p + 3 < D) y[4 * p + 3] = v.w;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// The register kernel (lane per chain, any target incl. a user's HIP source) with the ZIGGURAT generator (round 4, second
// session).  Box-Muller is 3/5 of that kernel's instructions (~73 per normal against ~25 for Philox + table fast path); what kept
// the ziggurat off it is the patch: the normals ARE the candidate's registers, and a register array cannot be indexed by a
// per-lane slot number.  Here: phase A writes every fast-path normal into y[k] and notes the failures in a per-lane bit mask; the
// failures of the whole wave-step (0.43 % x 64 D: 27 at d = 100) are queued as in mhx_zig_fixup and refined side by side by as
// many lanes, which leave the values in LDS by queue position; then every owner walks its failures once more, round by round like
// the queue did, fetches the round's value and every register of the mask word takes it where the slot number says so -- three
// straight-line instructions per register and round, ~600 per wave-step at d = 100 -- no slab of normals (which at a lane per
// chain would be 51 KB per wave).  Every lane stays alive (idle lanes shadow the last chain) because the queue and the
// patch are wave-wide.  LDS of the one-wave block: [layer table][queue][results][state tail (D - XR) x 64].  Same normals as
// mhx_zig_normal, hence the oracle's chains at reduction shape 1.
// (fp32, round 6: the same kernel in 4-byte reals over the signed pair table.)
#define MHX_REG_ZIG_LDS_BYTES(D, XR) ((size_t)((D) - (XR)) * 64 * sizeof(mhx_real) + MHX_REG_ZIG_HDR_BYTES)
template <int D, int TK, int PK, int XR = D, bool SLAB = false>
MHX_DEV void mhx_rwmh_reg_zig_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ tparams,
                                   const mhx_real* __restrict__ pvec, double* __restrict__ lds_)
{
    static_assert(PK != MHX_PROP_DENSE, "ISO / DIAG proposals");
    const int lane = (int)threadIdx.x;                         // one wave per block
    const long c_raw = (long)blockIdx.x * 64 + lane;
    const bool valid = c_raw < a.nchains;
    const long c = valid ? c_raw : (long)a.nchains - 1;        // idle lanes shadow the last chain (loads only)
    mhx_real* const zt = (mhx_real*)lds_;                      // (the table at offset 0: its look-up addresses need no base)
    unsigned short* const zq = (unsigned short*)((char*)lds_ + MHX_ZIG_TABLE_BYTES);  // 64 entries
    mhx_real* const zres = (mhx_real*)((char*)lds_ + MHX_ZIG_TABLE_BYTES + 128);      // 64 refined normals
    mhx_real* const xl = zres + 64 + lane;                     // this lane's column of the [D - XR][64] state tail
    mhx_real* const zslab = zres + 64 + (D - XR) * 64;         // (SLAB) the step's normals [block][lane][4], 16-byte aligned
#if MHX_REAL64
    for (int e = lane; e <= MHX_ZIG_N; e += 64) zt[e] = mhx_zig_x[e];
#else
    if ((mhx_u32)(mhx_u64)lds_ != 0u) __builtin_trap();        // (MHX_ZIG_PAIR_OF addresses the table at LDS byte 0)
    for (int e = lane; e < MHX_ZIG_PAIR_FLOATS; e += 64) zt[e] = mhx_zig_pair_entry(e);
#endif
    __syncthreads();
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;

    mhx_real x[XR > 0 ? XR : 1], y[D];
    auto getx = [&](const int k) -> mhx_real { return k < XR ? x[k < XR ? k : 0] : xl[(k - XR) * 64]; };
    const mhx_u32 cu = (mhx_u32)c * MHX_RB;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const mhx_real v = mhx_ld_off(a.x + (long)k * ld, cu);
        if (k < XR) x[k < XR ? k : 0] = v; else xl[(k - XR) * 64] = v;
    }
    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    mhx_u32 zsign = 0x80000000u;                               // (opaque: see mhx_zig_signed)
    asm volatile("" : "+s"(zsign));

    for (int i = 0; i < a.nsteps; ++i) {
        const mhx_u32 step = a.step0 + (mhx_u32)i;
        // ---- the step's D standard normals into the candidate's registers (fast path, queue, refinement, hand-back)
        mhx_reg_zig_fill<D, SLAB>(y, ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, zt, zq, zres, lane, (long)blockIdx.x * 64, a.first_chain, a.nchains, zsign, zslab);
        // ---- propose: y = x + s n   (src/proposal.jl:49-56)
#pragma unroll
        for (int k = 0; k < D; ++k) y[k] = mhx_fma(PK == MHX_PROP_ISO ? a.pscale : pvec[k], y[k], getx(k));
        // ---- log-density of the candidate and the accept test (src/mh-core.jl:103-108)
        const mhx_real lpy = mhx_target_eval<TK>(TK, y, D, tparams, a.ntparams, a.tconst);
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);          // strict; NaN compares false => reject
        // (the accepted lanes MOVE the candidate over the state under their execute mask: where the state's registers are AGPRs
        // -- d = 100: all 60 of them -- a select costs read + 2 v_cndmask + write per real, the move one write per word)
        if (acc) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (k < XR) x[k < XR ? k : 0] = y[k];
                else xl[(k - XR) * 64] = y[k];
            }
        }
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid));
        // ---- record (ext/AdvancedMHMCMCChainsExt.jl:96-105 layout, chain fastest)
        if (step == save_next) {
            if (valid) {
                mhx_real* slotp = a.samples + slot * (long)(D + 1) * ld;
                const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(D + 1) * (mhx_u32)ld * MHX_RB);
                const mhx_u32 ldb = (mhx_u32)ld * MHX_RB;
                mhx_u32 roff = 0u;                            // (running row offset: see the record of mhx_rwmh_coop_body)
                asm volatile("" : "+s"(roff));
#pragma unroll
                for (int k = 0; k < D; ++k) { mhx_srd_store<MHX_REC_STORE_AUX>(srd, cu, roff, getx(k)); roff += ldb; }
                mhx_srd_store<MHX_REC_STORE_AUX>(srd, cu, roff, lp);
                a.accepted[slot * ld + c] = acc ? 1 : 0;
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < D; ++k) mhx_st_off(a.x + (long)k * ld, cu, getx(k));
        a.lp[c] = lp;
        a.acc_count[c] = nacc;
        a.last_acc[c] = last ? 1 : 0;
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}


template <int L, int NBL, int TK, int PK, bool MOM, int WALK = MHX_WALK_PLAIN, int GEN = MHX_GEN_BOX_MULLER>
MHX_DEV void mhx_rwmh_coop_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ tparams,
                                const mhx_real* __restrict__ pvec)
{
    constexpr int CPW = 64 / L;                    // chains per wave
    constexpr bool ZIG = GEN == MHX_GEN_ZIGGURAT;
    constexpr bool ZKEEP = MHX_REAL64 && WALK == MHX_WALK_PLAIN && PK == MHX_PROP_ISO;   // (fp32: a select is one instruction already)
    // (re-reading the normals from LDS at the accept instead of keeping them in registers: C2 4.76e9 against 5.11e9 steps/s, removed --
    // profiles/r04i_zreload_ab.log)
    extern __shared__ double mhx_coop_lds[];
    constexpr int KS = ZIG ? MHX_ZIG_KS(NBL) : 1;                   // steps per fix-up group
#if MHX_REAL64
    typedef double mhx_d2 __attribute__((ext_vector_type(2)));
#else
    // the fp32 form (round 6): same layout in 4-byte reals -- table, then per wave the slab [block][lane][4], the queue, the masks
    const float* zt = (const float*)mhx_coop_lds;
    if (ZIG && (mhx_u32)(mhx_u64)mhx_coop_lds != 0u) __builtin_trap();
    constexpr int SLABD = NBL * 4 * 64;                            // floats of one step's normals
    float* zn0 = (float*)((char*)mhx_coop_lds + MHX_ZIG_TABLE_BYTES + (threadIdx.x >> 6) * MHX_ZIG_WAVE_BYTES(NBL));
    unsigned short* zq = (unsigned short*)(zn0 + KS * SLABD);
    mhx_u64* zfm = (mhx_u64*)((char*)(zn0 + KS * SLABD) + 128);
    if (ZIG) {
        for (int e = threadIdx.x; e < MHX_ZIG_PAIR_FLOATS; e += blockDim.x) ((float*)mhx_coop_lds)[e] = mhx_zig_pair_entry(e);
        __syncthreads();
    }
#endif
#if MHX_REAL64
    const double* zt = mhx_coop_lds;
    constexpr int SLABD = NBL * 4 * 64;                            // doubles of one step's normals
    double* zn0 = mhx_coop_lds + MHX_ZIG_TABLE_BYTES / 8 + (threadIdx.x >> 6) * (MHX_ZIG_WAVE_BYTES(NBL) / 8);
    unsigned short* zq = (unsigned short*)(zn0 + KS * SLABD);
    mhx_u64* zfm = (mhx_u64*)(zn0 + KS * SLABD + 16);
    if (ZIG) {
        for (int e = threadIdx.x; e <= MHX_ZIG_N; e += blockDim.x) mhx_coop_lds[e] = mhx_zig_x[e];
        __syncthreads();
    }
#endif
    // (a start-up stagger between blocks / between the waves of a block, so that the chip's record of a step does not leave as one
    // burst: built, no gain, removed -- profiles/r04x_record_ab.log)
    const int lane = threadIdx.x & 63;
    // One or two chains per wave (L = 64, 32): a block covers 32 or 64 bytes of a row of the [dim][chains] arrays, less than a cache
    // line -- blocks b, b + 8, b + 16, ... run on ONE XCD (round-robin dispatch), so they get CONSECUTIVE chain groups and the line
    // they share is assembled in one L2 instead of leaving four XCDs in four partial write-backs (the host rounds the grid to 8)
    const long blk = CPW <= 2 ? (long)(blockIdx.x & 7u) * (long)((gridDim.x + 7u) >> 3) + (long)(blockIdx.x >> 3) : (long)blockIdx.x;
    const long wave = (blk * blockDim.x + threadIdx.x) >> 6;
    const int cw = lane & (CPW - 1);
    const int l = lane / CPW;
    const long c_raw = wave * CPW + cw;
    const bool valid = c_raw < a.nchains;
    const long c = valid ? c_raw : (long)a.nchains - 1;      // idle lanes shadow the last chain (loads only)
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;
    const int d = a.dim;
    // Element (i, j) of this lane is dimension k = 4 (l + L i) + j.  Its address in a [dim][ld] array is
    //   base + ((4 L i + j) ld) * 4  [wave-uniform, scalar]  +  (4 l ld + c) * 4  [one VGPR for all (i, j)]
    // (the host guarantees (dim + 1) * ld * 4 < 2^32).  Only a lane's LAST block (i == NBL-1) can lie
    // past the end of the vector; every earlier block is complete for every lane.
    const mhx_u32 lane_off = ((mhx_u32)(4 * l) * (mhx_u32)ld + (mhx_u32)c) * MHX_RB;     // bytes
    const int k_last = 4 * (l + L * (NBL - 1));               // first dimension of the last block

    mhx_real x[NBL][4], y[NBL][4];
    // One or two chains per wave (L = 64, 32) touch 8 or 16 bytes per row of a [dim][chains] array: a fraction of every 32-byte
    // sector they read or dirty (PMC on C5: 3.3 x the algorithmic write bytes), and a d = 1000 run that records every state
    // ran at 0.6 TB/s.  The CB = 4 or 8 chains of a block therefore meet in LDS -- T[k][chain of the block], in the ziggurat's
    // slab or in a buffer of its own (a.tr_lds) -- and every row moves as ONE CB-real access: the state, the moment arrays
    // (once per launch each way) and the record of a saved step.
    constexpr int CB = 4 * CPW;                                    // chains per 256-thread block
    constexpr int NCH = CB * (int)sizeof(mhx_real) / 16;           // 16-byte pieces of a row segment
    mhx_real* const Tb = (mhx_real*)((char*)mhx_coop_lds + (ZIG ? MHX_ZIG_TABLE_BYTES_ANY : 0));
    const bool tr_io = CPW <= 2 && NCH >= 1 && a.tr_lds && blockDim.x == 256 && (((mhx_u64)ld * sizeof(mhx_real)) & 15ull) == 0 &&
                       ((wave & ~3L) * CPW + CB <= a.nchains);     // block-uniform: every chain of the block exists
    typedef mhx_u32 mhx_tr16 __attribute__((ext_vector_type(4)));
    auto fetch4 = [&](const mhx_real* base, auto& v) {
        const int cb = (int)(threadIdx.x >> 6) * CPW + cw;
        const long c0 = (wave & ~3L) * CPW;
        __syncthreads();
        for (int k = (int)threadIdx.x; k < d; k += 256) {
            const mhx_tr16* src = (const mhx_tr16*)(base + (long)k * ld + c0);
            mhx_tr16* dst = (mhx_tr16*)(Tb + (long)CB * k);
#pragma unroll
            for (int e = 0; e < (NCH > 0 ? NCH : 1); ++e) dst[e] = src[e];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * (l + L * i) + j;
                v[i][j] = (i < NBL - 1 || k < d) ? Tb[CB * k + cb] : MHX_R(0.0);
            }
        __syncthreads();                         // (the memory is the waves' own again only when every wave has taken its values)
    };
    auto flush4 = [&](mhx_real* base, const auto& v) {
        const int cb = (int)(threadIdx.x >> 6) * CPW + cw;
        const long c0 = (wave & ~3L) * CPW;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * (l + L * i) + j;
                if (i < NBL - 1 || k < d) Tb[CB * k + cb] = v[i][j];
            }
        __syncthreads();
        for (int k = (int)threadIdx.x; k < d; k += 256) {
            const mhx_tr16* src = (const mhx_tr16*)(Tb + (long)CB * k);
            mhx_tr16* dst = (mhx_tr16*)(base + (long)k * ld + c0);
#pragma unroll
            for (int e = 0; e < (NCH > 0 ? NCH : 1); ++e) dst[e] = src[e];
        }
        __syncthreads();                         // (a wave that runs ahead writes the next step's normals into this memory)
    };
    if (tr_io) {
        fetch4(a.x, x);
    } else {
#pragma unroll
    for (int i = 0; i < NBL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const mhx_real* col = a.x + (long)(4 * L * i + j) * ld;
            if (i < NBL - 1) x[i][j] = mhx_ld_off(col, lane_off);
            else x[i][j] = (k_last + j < d) ? mhx_ld_off(col, lane_off) : MHX_R(0.0);
        }
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) y[i][j] = MHX_R(0.0);
    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    // drifting / static walks: the lane's share of mu and of t = 2 L^-1 mu (a.pmean = mu[dim] then t[dim]); q(x) of the chain
    constexpr int NW = WALK != MHX_WALK_PLAIN ? NBL : 1;
    mhx_real pmu[NW][4], ptt[NW][4];
    mhx_real qxc = MHX_R(0.0);
    if (WALK != MHX_WALK_PLAIN) {
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * (l + L * i) + j;
                const bool in = (i < NBL - 1 || k < d) && a.pmean != nullptr;
                pmu[i][j] = in ? a.pmean[k] : MHX_R(0.0);
                ptt[i][j] = (in && WALK == MHX_WALK_DRIFT) ? a.pmean[d + k] : MHX_R(0.0);
            }
        if (WALK == MHX_WALK_STATIC) qxc = a.qx[c];
    }
    // running moments (MOM): continue the Welford recursion of the previous launches
    mhx_real mm[MOM ? NBL : 1][4], m2[MOM ? NBL : 1][4], lpm = MHX_R(0.0), lpm2 = MHX_R(0.0);
    mhx_u32 mom_n = a.mom_n0;
    if (MOM) {
        if (tr_io && mom_n) {
            if constexpr (MOM) { fetch4(a.mom_mean, mm); fetch4(a.mom_m2, m2); }
        } else {
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = (i < NBL - 1) || (k_last + j < d);
                const long e = (long)(4 * L * i + j) * ld;
                mm[i][j] = (in && mom_n) ? mhx_ld_off(a.mom_mean + e, lane_off) : MHX_R(0.0);
                m2[i][j] = (in && mom_n) ? mhx_ld_off(a.mom_m2 + e, lane_off) : MHX_R(0.0);
            }
        }
        if (mom_n) { lpm = a.mom_mean[(long)d * ld + c]; lpm2 = a.mom_m2[(long)d * ld + c]; }
    }

    // What the record of a saved step costs is the ISSUE of its 4 NBL stores (6.3e9 steps/s without it against 5.4e9), not their bytes
    // or their arriving in one burst.  Measured, bit-exact, no gain, removed in round 5 (A/B logs under profiles/): the stores spread
    // over the NEXT step's generation phase through an empty-range buffer descriptor (r04q_defer_rec_ab*.log), 16-byte stores of two
    // rows x two chains after a 2 x 2 lane transpose (r04x_record_ab.log).
    // (a saved step of the one- / two-chains-per-wave shapes stages its record in the slab memory: no groups then)
    const int ks_eff = (ZIG && !(tr_io && a.samples != nullptr && a.save_next != MHX_NO_SAVE)) ? KS : 1;
    for (int it0 = 0; it0 < a.nsteps; it0 += ks_eff) {
    const int ng = a.nsteps - it0 < ks_eff ? a.nsteps - it0 : ks_eff;
#if MHX_REAL64
    if (ZIG) {
        bool anyfail = false;
        mhx_u64 fm1 = 0ull;                                         // KS == 1: the step's failure mask, kept in a register
#pragma unroll 1
        for (int sg = 0; sg < ng; ++sg) {
            const mhx_u32 step = a.step0 + (mhx_u32)(it0 + sg);
            double* const zn = zn0 + sg * SLABD;
            {
            // phase A: every slot's candidate by the fast path -- table look-up, multiply, compare -- into LDS; the slots that
            // left their rectangles are noted in `fm` and finished by mhx_zig_fixup before the candidate state is formed
            mhx_u64 fm = 0ull;
            constexpr bool ZADDC = (MHX_ZADDC == 2 || (MHX_ZADDC == 1 && NBL > 4)) && NBL <= 16;
            mhx_u32 m4[4] = {0u, 0u, 0u, 0u};
            // Software pipeline over the lane's blocks: the table look-ups of block i are in flight while the Philox rounds of
            // block i + 1 run (one wave per SIMD has no other wave to hide an LDS round trip behind): per block -- issue the 4
            // look-ups (layers known since the previous stage), Philox of the next block, then consume.
            // GB blocks per pipeline stage = 2 GB independent Philox chains in flight.  Measured at C2: two blocks per stage (four
            // chains) 4.61e9 steps/s against 4.86e9 with one -- the rounds are not waiting on each other, the extra live words
            // cost more than the extra chains buy (C5: 4.88e8 against 5.36e8).
            constexpr int GB = 1;
            constexpr int NG = (NBL + GB - 1) / GB;                 // stages
            mhx_u32 khi[4 * GB], klo[4 * GB];                       // the candidates' raw words (hi:lo) of the stage in flight
            auto draw = [&](const int grp, mhx_u32 (&hi)[4 * GB], mhx_u32 (&lo)[4 * GB]) {
#pragma unroll
                for (int bb = 0; bb < GB; ++bb) {
                    const int i = grp * GB + bb;
                    if (i < NBL) {
                        const mhx_u32 b = (mhx_u32)(l + L * i);
                        const mhx_u32x4 w0 = mhx_philox(ks, id_lo, id_hi, step, (MHX_STREAM_PROPOSAL << 28) | (2u * b));
                        const mhx_u32x4 w1 = mhx_philox(ks, id_lo, id_hi, step, (MHX_STREAM_PROPOSAL << 28) | (2u * b + 1u));
                        hi[4 * bb + 0] = w0.x; lo[4 * bb + 0] = w0.y; hi[4 * bb + 1] = w0.z; lo[4 * bb + 1] = w0.w;
                        hi[4 * bb + 2] = w1.x; lo[4 * bb + 2] = w1.y; hi[4 * bb + 3] = w1.z; lo[4 * bb + 3] = w1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hi[4 * bb + e] = 0u; lo[4 * bb + e] = 0u; }
                    }
                }
            };
            draw(0, khi, klo);
            mhx_u32 zsign = 0x80000000u;                            // (opaque: see mhx_zig_signed)
            asm volatile("" : "+s"(zsign));
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                mhx_d2 xe[4 * GB];                                  // x[layer], x[layer + 1] of the stage's candidates
#pragma unroll
                for (int e = 0; e < 4 * GB; ++e) {                  // (two 8-byte reads of one address: ds_read2_b64)
                    const mhx_u32 ly = klo[e] & (mhx_u32)(MHX_ZIG_N - 1);
                    xe[e].x = zt[ly]; xe[e].y = zt[ly + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
                mhx_u32 nhi[4 * GB], nlo[4 * GB];
                if (grp + 1 < NG) draw(grp + 1, nhi, nlo);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bb = 0; bb < GB; ++bb) {
                    const int i = grp * GB + bb;
                    if (i < NBL) {
                        double nn[4];
                        mhx_u32 nib = 0u;                   // the block's four failure bits (1, 2, 4, 8: inline constants of the selects)
                        // Three ways to shorten the fast path, measured one by one (profiles/r04y_fastpath_ab.log; C2 = 13 blocks per lane at
                        // one wave per SIMD, C5 = 4 blocks at two): ANDOR -- the sign merge as one v_and_or_b32 (mhx_zig_signed) -- C2
                        // 2.955 -> 2.888 ms per launch; FABS -- the compare on |signed x|, so that the merge happens in place (-70 v_mov)
                        // -- and NIB -- the block's failure bits selected from inline constants, then one shift per block: fewer
                        // instructions both, C2 unchanged / 3.4 % SLOWER with them (e64 compares into SGPR pairs feed the selects),
                        // C5 1 % faster with all three.  Hence by shape.
                        constexpr bool ZFABS = NBL <= 4, ZNIB = NBL <= 4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const double ax = mhx_zig_ax(khi[4 * bb + j], klo[4 * bb + j], xe[4 * bb + j].x);
                            nn[j] = mhx_zig_signed(ax, klo[4 * bb + j], zsign);
                            // (|x| of the signed value: a source modifier of the compare, and `ax` dies at the sign merge -- the merge
                            // then happens in place instead of into a register that a v_mov brings back for the 16-byte LDS write)
                            if (ZADDC) { MHX_ZIG_NOTE64(m4[j], ax, xe[4 * bb + j].y, 4 * i + j); continue; }
                            bool fail = ZFABS ? !(__builtin_fabs(nn[j]) < xe[4 * bb + j].y) : !(ax < xe[4 * bb + j].y);
#ifdef MHX_TOOLS_BUILD
#ifdef MHX_ZIG_FORCE_FAIL       // test hook of the tools build (option ZIG_FORCE_FAIL): every n-th slot is sent through the fix-up although
                                // its candidate is inside its rectangle -- the refinement re-derives the same normal, so the chains
                                // are unchanged while the queue runs through several 64-entry windows per wave-step
                            fail = fail || ((4 * i + j + lane) % (MHX_ZIG_FORCE_FAIL) == 0);
#endif
#endif
                            if (i == NBL - 1) fail = fail && (k_last + j < d);   // padding dimensions past the end of the vector need no normal
                            if (ZNIB) nib |= fail ? (1u << j) : 0u;
                            else fm |= (fail ? 1ull : 0ull) << (4 * i + j);
                        }
                        if (ZNIB) fm |= (mhx_u64)nib << (4 * i);
                        mhx_d2 v2;
                        v2.x = nn[0]; v2.y = nn[1];
                        *(mhx_d2*)(zn + (((i * 2) * 64 + lane) << 1)) = v2;
                        v2.x = nn[2]; v2.y = nn[3];
                        *(mhx_d2*)(zn + (((i * 2 + 1) * 64 + lane) << 1)) = v2;
                    }
                }
                if (grp + 1 < NG) {
#pragma unroll
                    for (int e = 0; e < 4 * GB; ++e) { khi[e] = nhi[e]; klo[e] = nlo[e]; }
                }
            }
            if (ZADDC) {
                fm = (mhx_u64)(m4[0] | (m4[1] << 16)) | ((mhx_u64)(m4[2] | (m4[3] << 16)) << 32);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k_last + j >= d) fm &= ~(1ull << (16 * j));      // padding dimensions past the end of the vector need no normal
            }
            if (KS > 1) zfm[sg * 64 + lane] = fm;
            fm1 = fm;
            anyfail = anyfail || fm != 0ull;
            }
        }
        constexpr bool ZADDC_ = (MHX_ZADDC == 2 || (MHX_ZADDC == 1 && NBL > 4)) && NBL <= 16;
#ifdef MHX_TOOLS_BUILD
#ifndef MHX_ZIG_PROBE
#define MHX_ZIG_PROBE 0        // timing probe of the tools build (option ZIG_PROBE): 1 = skip the fix-up (WRONG normals)
#endif
        if (MHX_ZIG_PROBE != 1 && __ballot(anyfail))
#else
        if (__ballot(anyfail))
#endif
            mhx_zig_fixup<L>(ks, zt, zn0, zq, zfm, ng, SLABD, lane, wave, a.first_chain, a.nchains, a.step0 + (mhx_u32)it0, MHX_STREAM_PROPOSAL,
                             KS == 1, fm1, ZADDC_ ? NBL - 1 : -1);
    }
#else
    if (ZIG) {
        // the fp32 form of the same three phases: ONE Philox call per block of four normals (a word each); a candidate's nine top bits
        // fetch its signed table pair as one 8-byte read, its mantissa is used where it lies, the sign comes with the table entry and
        // the compare's carry is shifted into the mask of failures by one add-with-carry: and_or, shift, and, fma, compare, addc per
        // normal (9 before the word was laid out for it); the block's four normals go to the slab as one 16-byte write.
        // Masks: one 32-bit word per word j of the Philox blocks, block i at bit NBL - 1 - i (mhx_zig_fixup: rev_top).
        static_assert(NBL <= 16, "a 16-bit field of failure bits per Philox word");
        bool anyfail = false;
        mhx_u64 fm1 = 0ull;
        mhx_u64 pad_ok = ~0ull;                                     // padding dimensions past the end of the vector need no normal
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k_last + j >= d) pad_ok &= ~(1ull << (16 * j));
#pragma unroll 1
        for (int sg = 0; sg < ng; ++sg) {
            const mhx_u32 step = a.step0 + (mhx_u32)(it0 + sg);
            float* const zn = zn0 + sg * SLABD;
            mhx_u32 m4[4] = {0u, 0u, 0u, 0u};
            mhx_u32 kw[4];
            auto draw = [&](const int i, mhx_u32 (&w)[4]) {
                if (i < NBL) {
                    const mhx_u32x4 w4 = mhx_philox(ks, id_lo, id_hi, step, (MHX_STREAM_PROPOSAL << 28) | (mhx_u32)(l + L * i));
                    w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
                } else {
                    w[0] = w[1] = w[2] = w[3] = 0u;
                }
            };
            draw(0, kw);
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                mhx_f2v xe[4];                                      // +-x[layer], x[layer + 1] of the block's candidates
#pragma unroll
                for (int e = 0; e < 4; ++e) xe[e] = MHX_ZIG_PAIR_OF(kw[e]);
#ifndef MHX_ZIG32_FENCE
#define MHX_ZIG32_FENCE 3       // bit 0: the look-ups are issued before the next block's Philox rounds; bit 1: those before this block's fast path
#endif
                if (MHX_ZIG32_FENCE & 1) __builtin_amdgcn_sched_barrier(0);
                mhx_u32 nw[4];
                if (i + 1 < NBL) draw(i + 1, nw);                   // the next block's Philox rounds run while the look-ups are in flight
                if (MHX_ZIG32_FENCE & 2) __builtin_amdgcn_sched_barrier(0);
                mhx_f4v v4;
                float nn[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    nn[j] = mhx_zig_ax(kw[j], xe[j].x);             // signed: the table entry carries the sign
                    MHX_ZIG_NOTE(m4[j], nn[j], xe[j].y, 4 * i + j);
                }
                v4.x = nn[0]; v4.y = nn[1]; v4.z = nn[2]; v4.w = nn[3];
                *(mhx_f4v*)(zn + ((i * 64 + lane) << 2)) = v4;
                if (i + 1 < NBL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) kw[e] = nw[e];
                }
            }
            const mhx_u64 fm = ((mhx_u64)(m4[0] | (m4[1] << 16)) | ((mhx_u64)(m4[2] | (m4[3] << 16)) << 32)) & pad_ok;
            if (KS > 1) zfm[sg * 64 + lane] = fm;
            fm1 = fm;
            anyfail = anyfail || fm != 0ull;
        }
#ifdef MHX_TOOLS_BUILD
#ifndef MHX_ZIG_PROBE
#define MHX_ZIG_PROBE 0
#endif
        if (MHX_ZIG_PROBE != 1 && __ballot(anyfail))
#else
        if (__ballot(anyfail))
#endif
            mhx_zig_fixup<L>(ks, zt, zn0, zq, zfm, ng, SLABD, lane, wave, a.first_chain, a.nchains, a.step0 + (mhx_u32)it0, MHX_STREAM_PROPOSAL,
                             KS == 1, fm1, NBL - 1);
    }
#endif
#pragma unroll 1
    for (int sg = 0; sg < ng; ++sg) {
        const int it = it0 + sg;
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        // Branch-free over the lane's blocks: a last block past the end of the vector (and the padding
        // dimensions of the final block) computes on zeros -- y = 0 there, and fma(0, 0, q) == q bit for
        // bit, so the partial sums need no predication.
        mhx_real q = MHX_R(0.0), fwd = MHX_R(0.0), bwd = MHX_R(0.0), y00 = MHX_R(0.0);
#if MHX_REAL64
        const double* const zn_c = ZIG ? (const double*)(zn0 + sg * SLABD) : nullptr;
        if (ZIG) {
            // the step's normals, final: all of them on their way to the registers the candidate will occupy (one wait)
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                const mhx_d2 v0 = *(const mhx_d2*)(zn_c + (((i * 2) * 64 + lane) << 1));
                const mhx_d2 v1 = *(const mhx_d2*)(zn_c + (((i * 2 + 1) * 64 + lane) << 1));
                y[i][0] = v0.x; y[i][1] = v0.y; y[i][2] = v1.x; y[i][3] = v1.y;
            }
        }
#else
        if (ZIG) {
            const float* const zn_c = zn0 + sg * SLABD;
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                const mhx_f4v v = *(const mhx_f4v*)(zn_c + ((i * 64 + lane) << 2));
                y[i][0] = v.x; y[i][1] = v.y; y[i][2] = v.z; y[i][3] = v.w;
            }
        }
#endif
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            const int b = l + L * i;
            mhx_real n[4];
            if (ZIG) {
                n[0] = y[i][0]; n[1] = y[i][1]; n[2] = y[i][2]; n[3] = y[i][3];
            } else
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mhx_real sc = a.pscale;
                if (PK != MHX_PROP_ISO) {
                    const int k = 4 * b + j;
                    sc = pvec[(i < NBL - 1 || k < d) ? k : 0];
                }
                mhx_real yk;
                if (WALK == MHX_WALK_PLAIN) {
                    yk = mhx_fma(sc, n[j], x[i][j]);
                } else {
                    // the same expressions as the state-in-HBM kernel: x + fma(s, n, mu) with a mean, fma(s, n, x) without;
                    // a static proposal takes x = 0.  Padding dimensions contribute fma(0, 0, sum) == sum.
                    const bool in = i < NBL - 1 || k_last + j < d;
                    const mhx_real nk = in ? n[j] : MHX_R(0.0);
                    const mhx_real xk = WALK == MHX_WALK_STATIC ? MHX_R(0.0) : x[i][j];
                    yk = a.pmean ? xk + mhx_fma(sc, nk, pmu[i][j]) : mhx_fma(sc, nk, xk);
                    fwd = mhx_fma(nk, nk, fwd);
                    if (WALK == MHX_WALK_DRIFT) { const mhx_real tk = nk + ptt[i][j]; bwd = mhx_fma(tk, tk, bwd); }
                }
                if (i == NBL - 1) yk = (k_last + j < d) ? yk : MHX_R(0.0);
                // ZKEEP (plain walk, one scale for all dimensions): the registers of the candidate keep the NORMAL instead -- the
                // accepted state is re-formed as fma(s_acc, n, x) with s_acc = accepted ? s : 0, the same fma (or x itself, exactly)
                // in one instruction per real where a select of a double takes two.  A padding dimension keeps n = 0: x stays 0.
                // (fma(0, n, x) == x for every x but -0.0, which a chain never holds: mhx_run_init / set_state turn a caller's -0.0
                // into +0.0 -- rwmh_canonical_zero in mhx_api.hip -- and a rounded sum is -0 only if both terms are.)
                if (ZKEEP) y[i][j] = (i == NBL - 1 && !(k_last + j < d)) ? MHX_R(0.0) : n[j];
                else y[i][j] = yk;
                if (i == 0 && j == 0) y00 = yk;
                const mhx_real sq = mhx_fma(yk, yk, q);
                if (TK == MHX_TARGET_BANANA && i == 0 && j == 0) {
                    q = l == 0 ? (yk * yk) * MHX_R(0.01) : sq;           // x1 ~ N(0, 100)
                } else if (TK == MHX_TARGET_BANANA && i == 0 && j == 1) {
                    const mhx_real y0 = y00;
                    const mhx_real u = mhx_fma(tparams[0], mhx_fma(y0, y0, -MHX_R(100.0)), yk);
                    q = l == 0 ? mhx_fma(u, u, q) : sq;
                } else if (TK == MHX_TARGET_FUNNEL && i == 0 && j == 0) {
                    q = l == 0 ? q : sq;                           // x1 is the funnel's scale, not a summand
                } else {
                    q = sq;
                }
            }
            // keep the blocks in program order: with >= 2 waves per SIMD the other wave provides the
            // latency hiding, and interleaving blocks only inflates the live register set
            __builtin_amdgcn_sched_barrier(0);
        }
        q = mhx_butterfly<L>(q);
        if (WALK != MHX_WALK_PLAIN) {
            fwd = mhx_butterfly<L>(fwd);
            if (WALK == MHX_WALK_DRIFT) bwd = mhx_butterfly<L>(bwd);
        }
        mhx_real lpy;
        if (TK == MHX_TARGET_FUNNEL) {
            const mhx_real v = __shfl(y00, cw, 64);               // x1 lives in lane l == 0 of the chain
            const mhx_real ev = mhx_exp(-v);
            mhx_real r = (v * v) * MHX_ONE_18;
            r = mhx_fma(MHX_R(0.5) * (mhx_real)(d - 1), v, r);
            r = mhx_fma(MHX_R(0.5) * ev, q, r);
            lpy = a.tconst - r;
        } else {
            lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        }
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        // src/mh-core.jl:104-105 with logratio_proposal_density (src/proposal.jl:190-192 / :74-83) for the walks that have one
        const mhx_real qy = -MHX_R(0.5) * fwd;
        const mhx_real loga = WALK == MHX_WALK_STATIC ? (lpy - lp) + (qxc - qy)
                            : (WALK == MHX_WALK_DRIFT ? (lpy - lp) + MHX_R(0.5) * (fwd - bwd) : (lpy - lp));
        const bool acc = logu < loga;
        const mhx_real s_acc = acc ? a.pscale : MHX_R(0.0);
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = ZKEEP ? mhx_fma(s_acc, y[i][j], x[i][j]) : (acc ? y[i][j] : x[i][j]);
        lp = acc ? lpy : lp;
        if (WALK == MHX_WALK_STATIC) qxc = acc ? qy : qxc;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && l == 0));
        if (MOM && step == save_next) {
            ++mom_n;
            const mhx_real rn = MHX_R(1.0) / (mhx_real)mom_n;
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mhx_welford(x[i][j], rn, mm[i][j], m2[i][j]);
            mhx_welford(lp, rn, lpm, lpm2);
            save_next += (mhx_u32)a.thinning;
        } else if (step == save_next && tr_io) {
            // (one or two chains per wave: the record leaves through the block's LDS as whole row segments, see fetch4)
            mhx_real* slotp = a.samples + slot * (long)(d + 1) * ld;
            flush4(slotp, x);
            if (l == 0) {
                slotp[(long)d * ld + c] = lp;
                a.accepted[slot * ld + c] = acc ? 1 : 0;
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        } else if (step == save_next) {
            if (valid) {
                mhx_real* slotp = a.samples + slot * (long)(d + 1) * ld;
                const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(d + 1) * (mhx_u32)ld * MHX_RB);
                const mhx_u32 ldb = (mhx_u32)ld * MHX_RB;
                // The row offset as a running scalar sum behind an opaque asm (round 4).  As 4 NBL loop-invariant products hipcc
                // hoists the offsets out of the step loop, spills them to lanes of a VGPR and pays v_readlane + s_nop 4 per store --
                // and at one wave per SIMD the record costs by the instruction, not by the byte (profiles/r04x_record_ab.log)
                mhx_u32 roff = 0u;
                asm volatile("" : "+s"(roff));
#pragma unroll
                for (int i = 0; i < NBL; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const mhx_u32 rowb = roff;                                                 // wave-uniform -> soffset
                        roff += (j < 3 ? 1u : (mhx_u32)(4 * L - 3)) * ldb;
                        if (i < NBL - 1) mhx_srd_store<MHX_REC_STORE_AUX>(srd, lane_off, rowb, x[i][j]);
                        else if (k_last + j < d) mhx_srd_store<MHX_REC_STORE_AUX>(srd, lane_off, rowb, x[i][j]);
                    }
                if (l == 0) {
                    slotp[(long)d * ld + c] = lp;
                    a.accepted[slot * ld + c] = acc ? 1 : 0;
                }
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    }
    const bool tr_out = tr_io;
    if (tr_out) {
        flush4(a.x, x);
        if constexpr (MOM) { flush4(a.mom_mean, mm); flush4(a.mom_m2, m2); }
    }
    if (valid) {
        if (!tr_out) {
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mhx_real* col = a.x + (long)(4 * L * i + j) * ld;
                if (i < NBL - 1) mhx_st_off(col, lane_off, x[i][j]);
                else if (k_last + j < d) mhx_st_off(col, lane_off, x[i][j]);
            }
        }
        if (l == 0) {
            a.lp[c] = lp;
            a.acc_count[c] = nacc;
            a.last_acc[c] = last ? 1 : 0;
            if (WALK == MHX_WALK_STATIC) a.qx[c] = qxc;
        }
        if (MOM) {
            if (!tr_out) {
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = (i < NBL - 1) || (k_last + j < d);
                    const long e = (long)(4 * L * i + j) * ld;
                    if (in) { mhx_st_off(a.mom_mean + e, lane_off, mm[i][j]); mhx_st_off(a.mom_m2 + e, lane_off, m2[i][j]); }
                }
            }
            if (l == 0) { a.mom_mean[(long)d * ld + c] = lpm; a.mom_m2[(long)d * ld + c] = lpm2; }
        }
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// ---------------------------------------------------------------------------------------------
// initial state (src/mh-core.jl:83-84): x0 = initial_params, or a bare proposal draw
// (src/proposal.jl:41-47) from Philox stream INIT; lp0 = logdensity(model, x0).
template <int TK>
MHX_DEV void mhx_rwmh_init_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ tparams,
                                const mhx_real* __restrict__ pvec, const int draw)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const long ld = a.ld;
    const int d = a.dim;
    mhx_real* xs = a.x + c;
    if (draw) {
        const mhx_philox_key ks = mhx_philox_schedule(a.seed);
        const int nblk = (d + 3) >> 2;
        for (int b = 0; b < nblk; ++b) {
            mhx_real n[4];
            mhx_normal4_gen(a.normal_gen, ks, id_lo, id_hi, 0u, MHX_STREAM_INIT, (mhx_u32)b, n);
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k < d) {
                    if (a.prop_kind == MHX_PROP_DENSE) xs[(long)k * ld] = n[j];
                    else if (a.pmean) xs[(long)k * ld] = MHX_R(0.0) + mhx_fma(a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k], n[j], a.pmean[k]);
                    else xs[(long)k * ld] = mhx_fma(a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k], n[j], MHX_R(0.0));
                }
            }
        }
        if (a.prop_kind == MHX_PROP_DENSE) {
            for (int r = d - 1; r >= 0; --r) {
                const mhx_real* Lr = pvec + (long)r * (r + 1) / 2;
                mhx_real w = MHX_R(0.0);
                for (int j = 0; j <= r; ++j) w = mhx_fma(Lr[j], xs[(long)j * ld], w);
                xs[(long)r * ld] = a.pmean ? MHX_R(0.0) + (a.pmean[r] + w) : MHX_R(0.0) + w;
            }
        }
    }
    mhx_strided_x xv;
    xv.base = xs;
    xv.ld = ld;
    a.lp[c] = mhx_target_eval_lanes<TK>(a.target_kind, xv, d, tparams, a.ntparams, a.tconst, a.reduce_lanes);
    a.acc_count[c] = 0u;
    a.last_acc[c] = 0;          // Transition(params, lp, false), src/mh-core.jl:84
}

// ---------------------------------------------------------------------------------------------
// logdensity(model, x) for a batch of points, x [dim][n] -> lp [n]   (src/AdvancedMH.jl:74)
template <int TK>
MHX_DEV void mhx_target_eval_body(const mhx_real* __restrict__ x, mhx_real* __restrict__ lp, const int n,
                                  const int d, const int kind, const mhx_real* __restrict__ tparams,
                                  const int ntparams, const mhx_real tconst, const int lanes)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    mhx_strided_x xv;
    xv.base = x + c;
    xv.ld = n;
    lp[c] = mhx_target_eval_lanes<TK>(kind, xv, d, tparams, ntparams, tconst, lanes);
}

// record the current state into sample slot `slot` (sample 1 of a run with discard_initial == 0)
MHX_DEV void mhx_record_state_body(const mhx_real* __restrict__ x, const mhx_real* __restrict__ lp,
                                   const unsigned char* __restrict__ last_acc, mhx_real* samples,
                                   unsigned char* accepted, const int n, const long ld, const int d,
                                   const long slot)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    mhx_real* row = samples + slot * (long)(d + 1) * ld + c;
    for (int k = 0; k < d; ++k) row[(long)k * ld] = x[(long)k * ld + c];
    row[(long)d * ld] = lp[c];
    accepted[slot * ld + c] = last_acc[c];
}

// first folded sample = the current state (discard_initial == 0): mean = state, M2 = 0
MHX_DEV void mhx_moments_first_body(const mhx_real* __restrict__ x, const mhx_real* __restrict__ lp, mhx_real* mean, mhx_real* m2,
                                    const int n, const long ld, const int d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    for (int k = 0; k < d; ++k) { mean[(long)k * ld + c] = x[(long)k * ld + c]; m2[(long)k * ld + c] = MHX_R(0.0); }
    mean[(long)d * ld + c] = lp[c];
    m2[(long)d * ld + c] = MHX_R(0.0);
}

// JIT entry points: hiprtc compiles this header with the specialisation macros defined
// ---------------------------------------------------------------------------------------------
// A WAVE per chain (round 5; kernel_variant 11): the few-chain regime of the reference's own example -- README.md:25-40,
// test/runtests.jl:76-94: ONE chain, 100 000 draws of theta = (mu, sigma) under sum(logpdf.(Normal(mu, sigma), data)).  A lane per chain
// walks the data terms serially (2.4 us per transition: 23 x slower than one CPU thread); here the 64 lanes of a wave SPLIT THE TERMS
// (lane l: terms l, l + 64, ... in ascending order, one IEEE division each), the partial sums meet in the fixed xor-butterfly -- the
// spec's reduction shape 64, which the oracle takes for this target too -- and every lane carries the (tiny) state redundantly.  The
// random numbers do not depend on the state (counter RNG), so they leave the sequential chain altogether: lane l draws the normals and
// the accept uniform of step s0 + l for a batch of 64 steps at once, the chain then takes them by v_readlane; lane l also keeps the
// state after ITS step and writes that step's record after the batch (one chain: 64 consecutive slots, contiguous).  What is left per
// transition is the dependent chain itself -- two fma, a division, six butterfly steps beside log(sigma), the compare -- and that
// chain runs K = 4 or 8 steps at a time speculatively (below): same decisions, same bits, 1 / K of the latency while steps reject.
// N butterflies as ONE, transposed: at offset OFF the lane whose bit OFF is clear keeps the even member of every pair of running sums
// and the other lane the odd one -- each hands its partner the member it does not keep -- so the N sums halve at each of log2 N
// levels and lane l is left with the sum of candidate l & (N0 - 1) over its group of N0 lanes.  Every addition has the operands the
// plain xor-butterfly of that candidate has at that lane (in either order: the same bits).
template <int N, int OFF>
MHX_DEV mhx_real mhx_wave_transposed(const mhx_real (&v)[N], const int lane)
{
    if constexpr (N == 1) {
        return v[0];
    } else {
        const bool odd = (lane & OFF) != 0;
        mhx_real w[N / 2];
#pragma unroll
        for (int m = 0; m < N / 2; ++m) w[m] = (odd ? v[2 * m + 1] : v[2 * m]) + mhx_lane_xor<OFF>(odd ? v[2 * m] : v[2 * m + 1]);
        return mhx_wave_transposed<N / 2, 2 * OFF>(w, lane);
    }
}
// ... and the plain butterfly over the remaining offsets OFF, 2 OFF, ..., 32
template <int OFF>
MHX_DEV mhx_real mhx_wave_butterfly_from(mhx_real q)
{
    if constexpr (OFF <= 32) return mhx_wave_butterfly_from<2 * OFF>(mhx_butterfly_add<OFF>(q));
    else return q;
}
// K = candidates per round (4 or 8; the same chain for any K: a round settles the steps up to its first acceptance).  The host takes
// 8 while the previous sampling call accepted fewer than one step in eight (a round then settles 6.5 steps at the README model's 6 %
// against 3.65 with K = 4, for about 1.45 x the instructions), else 4.
template <int PK, int K = 4>
MHX_DEV void mhx_rwmh_wave_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    static_assert(K == 4 || K == 8, "candidates per round");
    static_assert(PK != MHX_PROP_DENSE, "ISO / DIAG proposals");
    const int lane = (int)threadIdx.x;                         // one wave per block, one chain per block
    const long c = (long)blockIdx.x;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;
    const int np = a.ntparams;
    mhx_real x0 = a.x[c], x1 = a.x[ld + c], lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    bool last = a.last_acc[c] != 0;
    const mhx_real s0 = PK == MHX_PROP_ISO ? a.pscale : pvec[0], s1 = PK == MHX_PROP_ISO ? a.pscale : pvec[1];
    const bool has_term = lane < np;
    const mhx_real t0 = has_term ? tparams[lane] : MHX_R(0.0);                  // this lane's first term stays in a register
    const mhx_real npf = (mhx_real)np;
    mhx_u32 total_acc = 0u;
    auto bcast = [](const mhx_real v, const int j) -> mhx_real {                   // lane j's value, for every lane (j is wave-uniform)
#if MHX_REAL64
        const mhx_u64 b = __builtin_bit_cast(mhx_u64, v);
        const mhx_u32 lo = (mhx_u32)__builtin_amdgcn_readlane((int)(mhx_u32)b, j), hi = (mhx_u32)__builtin_amdgcn_readlane((int)(mhx_u32)(b >> 32), j);
        return __builtin_bit_cast(double, (mhx_u64)lo | ((mhx_u64)hi << 32));
#else
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
#endif
    };
    for (int it0 = 0; it0 < a.nsteps; it0 += 64) {
        const int nb = a.nsteps - it0 < 64 ? a.nsteps - it0 : 64;
        // ---- the batch's draws, one step per lane (lanes >= nb draw for steps nobody takes)
        const mhx_u32 mystep = a.step0 + (mhx_u32)(it0 + lane);
        mhx_real n[4];
        mhx_normal4(ks, id_lo, id_hi, mystep, MHX_STREAM_PROPOSAL, 0u, n);      // normals 0, 1 of the step (src/proposal.jl:49-56)
        mhx_accept_cache ac;
        ac.group = 0xffffffffu;
        ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
        const mhx_real mylogu = mhx_accept_logu(ks, id_lo, id_hi, mystep, ac);
        mhx_real rx0 = x0, rx1 = x1, rlp = lp;
        bool racc = false;
        // ---- the chain, K steps at a time.  The candidates of steps j .. j+K-1 are all formed from the CURRENT state and evaluated side
        // by side (independent dependency chains: a wave alone on its SIMD has nothing else to fill the latencies with); the first
        // of them that is accepted ends the round -- the steps before it were rejections from exactly this state, so their outcome
        // is what the sequential loop computes; the ones after it are discarded and re-done from the new state.  Acceptance is low
        // where this kernel runs (the README example: 6 %), so a round advances almost K steps for the latency of one.
        const int kk = lane & (K - 1);                                           // the candidate whose SCALAR part this lane evaluates
        for (int j = 0; j < nb;) {
            mhx_real y0[K], y1[K], acc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int jj = j + k < nb ? j + k : nb - 1;                      // (past the batch: a copy of its last step, never used)
                y0[k] = mhx_fma(s0, bcast(n[0], jj), x0);
                y1[k] = mhx_fma(s1, bcast(n[1], jj), x1);
                // logdensity(model, candidate) (src/mh-core.jl:103): mhx_target_eval's IID_NORMAL expressions, the sum in shape 64.
                // (a SELECT, not a branch, for the lanes without a term: under `if (lane < np)` hipcc emits one exec-masked block per
                // candidate and the K divisions run one after the other)
                const mhx_real z = (t0 - y0[k]) / y1[k];
                const mhx_real zz = mhx_fma(z, z, MHX_R(0.0));
                acc[k] = has_term ? zz : MHX_R(0.0);
            }
            if (np > 64) {                                                       // (wave-uniform) more terms than lanes
                for (int i = lane + 64; i < np; i += 64) {
                    const mhx_real ti = tparams[i];
#pragma unroll
                    for (int k = 0; k < K; ++k) { const mhx_real z = (ti - y0[k]) / y1[k]; acc[k] = mhx_fma(z, z, acc[k]); }
                }
            }
            // The four butterflies as ONE, transposed: at offset 1 the even lane of a pair keeps candidate 0's (2's) sum and the odd
            // lane candidate 1's (3's) -- each gives the partner the one it does not keep --, at offset 2 the pairs (0, 1) and (2, 3)
            // merge the same way, offsets 4 .. 32 run on the one value left: 7 cross-lane steps instead of 24, and lane l ends with the
            // total of candidate l & 3.  Every addition has the operands the plain butterfly of that candidate has at that lane (in
            // either order: the same bits), so the totals are the shape-64 sums of the spec.  One wave alone on its SIMD issues an
            // instruction every ~5 cycles whatever its class: what a round costs is its instruction count, hence this and the next step.
            const mhx_real tot = mhx_wave_butterfly_from<K>(mhx_wave_transposed<K, 1>(acc, lane));
            // ... and the scalar part of candidate l & (K - 1) on lane l only (one logarithm per lane instead of K)
            const int myj = j + kk;
            mhx_real my_y1 = y1[0];
#pragma unroll
            for (int k = 1; k < K; ++k) my_y1 = kk == k ? y1[k] : my_y1;
            const mhx_real my_logu = __shfl(mylogu, myj < nb ? myj : nb - 1, 64);
            const mhx_real tt = mhx_log_sel(my_y1) + MHX_HALF_LOG_2PI;           // (mhx_log's value, branch-free)
            const mhx_real v = mhx_fma(-MHX_R(0.5), tot, -(npf * tt));
            const mhx_real my_lpy = (my_y1 > MHX_R(0.0)) ? v : -MHX_INF;         // theta[2] >= 0 support, -Inf at sigma == 0; NaN: reject
            // the first accepted step of the round (strict compare, src/mh-core.jl:108; NaN compares false): lanes 0 .. K-1 speak for
            // candidates 0 .. K-1
            const mhx_u32 okm = (mhx_u32)__ballot(myj < nb && my_logu < (my_lpy - lp)) & ((1u << K) - 1u);
            const int first = okm ? (int)__builtin_ctz(okm) : K;                 // wave-uniform
            const int adv = first < K ? first + 1 : (nb - j < K ? nb - j : K);   // steps this round settles
            // lanes j .. j+adv-2 (rejections) record the old state, lane j+adv-1 the new one if it was an acceptance
            if (lane >= j && lane < j + adv) { rx0 = x0; rx1 = x1; rlp = lp; racc = false; }
            if (first < K) {
                mhx_real a0 = y0[0], a1 = y1[0];
#pragma unroll
                for (int k = 1; k < K; ++k) { a0 = first == k ? y0[k] : a0; a1 = first == k ? y1[k] : a1; }
                x0 = a0; x1 = a1; lp = bcast(my_lpy, first);
                nacc += 1u;
                total_acc += 1u;
                if (lane == j + first) { rx0 = x0; rx1 = x1; rlp = lp; racc = true; }
            }
            last = first < K;
            j += adv;
        }
        // ---- the batch's records, one step per lane (ext/AdvancedMHMCMCChainsExt.jl:96-105 layout)
        if (lane < nb && a.save_next != MHX_NO_SAVE && mystep >= a.save_next) {
            const mhx_u32 since = mystep - a.save_next;
            if (since % (mhx_u32)a.thinning == 0u) {
                const long slot = (long)a.save_slot + (long)(since / (mhx_u32)a.thinning);
                mhx_real* row = a.samples + slot * 3L * ld + c;
                row[0] = rx0;
                row[ld] = rx1;
                row[2 * ld] = rlp;
                a.accepted[slot * ld + c] = racc ? 1 : 0;
            }
        }
    }
    if (lane == 0) {
        a.x[c] = x0;
        a.x[ld + c] = x1;
        a.lp[c] = lp;
        a.acc_count[c] = nacc;
        a.last_acc[c] = last ? 1 : 0;
        atomicAdd(a.acc_total, (mhx_u64)total_acc);
    }
}

#ifdef MHX_JIT_RWMH_REG
#ifndef MHX_JIT_XR
#define MHX_JIT_XR MHX_JIT_DIM
#endif
#ifndef MHX_JIT_REG_WAVES
#define MHX_JIT_REG_WAVES 1       // waves per SIMD the register budget is cut for (the host asks for 2 where the arrays leave room)
#endif
extern "C" __global__ void __launch_bounds__(64, MHX_JIT_REG_WAVES)
mhx_jit_rwmh_reg(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
#if defined(MHX_JIT_GEN) && MHX_JIT_GEN == 1
    extern __shared__ double mhx_reg_zig_lds[];                // MHX_REG_ZIG_LDS_BYTES(MHX_JIT_DIM, MHX_JIT_XR)
#ifndef MHX_JIT_ZSLAB
#define MHX_JIT_ZSLAB 0
#endif
    mhx_rwmh_reg_zig_body<MHX_JIT_DIM, MHX_JIT_TK, MHX_JIT_PK, MHX_JIT_XR, (MHX_JIT_ZSLAB != 0)>(a, tparams, pvec, mhx_reg_zig_lds);
#elif MHX_JIT_XR < MHX_JIT_DIM
    extern __shared__ mhx_real mhx_reg_state_tail[];           // [MHX_JIT_DIM - MHX_JIT_XR][64]
    mhx_rwmh_reg_body<MHX_JIT_DIM, MHX_JIT_TK, MHX_JIT_PK, MHX_JIT_XR>(a, tparams, pvec, mhx_reg_state_tail);
#else
    mhx_rwmh_reg_body<MHX_JIT_DIM, MHX_JIT_TK, MHX_JIT_PK>(a, tparams, pvec);
#endif
}
#endif
#ifdef MHX_JIT_RWMH_COOP
#ifndef MHX_JIT_WAVES
#define MHX_JIT_WAVES MHX_COOP_WAVES(MHX_JIT_NBL)
#endif
extern "C" __global__ void __launch_bounds__(256, MHX_JIT_WAVES)
mhx_jit_rwmh_coop(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
#ifndef MHX_JIT_WALK
#define MHX_JIT_WALK 0
#endif
#ifndef MHX_JIT_GEN
#define MHX_JIT_GEN MHX_GEN_BOX_MULLER
#endif
    mhx_rwmh_coop_body<MHX_JIT_L, MHX_JIT_NBL, MHX_JIT_TK, MHX_JIT_PK, (MHX_JIT_MOM != 0), MHX_JIT_WALK, MHX_JIT_GEN>(a, tparams, pvec);
}
#endif
#ifdef MHX_JIT_RWMH_GENERIC
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_rwmh_generic(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    mhx_rwmh_generic_body<MHX_JIT_TK>(a, tparams, pvec);
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_rwmh_init(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec,
                  const int draw)
{
    mhx_rwmh_init_body<MHX_JIT_TK>(a, tparams, pvec, draw);
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_target_eval(const mhx_real* __restrict__ x, mhx_real* __restrict__ lp, const int n, const int d,
                    const int kind, const mhx_real* __restrict__ tparams, const int ntparams, const mhx_real tconst,
                    const int lanes)
{
    mhx_target_eval_body<MHX_JIT_TK>(x, lp, n, d, kind, tparams, ntparams, tconst, lanes);
}
#endif
MHX_NS_END
