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
// One launch = one half-step; consecutive launches on the stream order the halves.  Walkers are
// stored [dim][W] (walker fastest): own reads/writes are coalesced, the partner gather is a
// scattered 4-byte read per dimension served by L2 (16 384 x 50 floats = 3.3 MB).
#pragma once
#include "mhx_targets.h"

struct mhx_emcee_args {
    float* x;                 // [dim][W]
    float* lp;                // [W]
    mhx_u32* acc_count;       // [W]
    mhx_u64* acc_total;
    float* samples;           // [slots][dim+1][W] or null
    unsigned char* accepted;  // [slots][W] or null
    unsigned char* last_acc;  // [W]
    float* ybuf;              // [dim][W] candidate scratch (run-time-dimension kernel)
    mhx_u64 seed;
    mhx_u64 ensemble_id;
    int nwalkers;
    int dim;
    int target_kind;
    int ntparams;
    float tconst;
    float stretch;            // a
    mhx_u32 sweep;            // RNG step counter of this sweep
    int half;                 // 0: walkers [0, W/2) move; 1: walkers [W/2, W) move
    long save_slot;           // slot to record this sweep into, or -1
};

// D > 0: compile-time dimension, candidate in registers.  D == 0: run-time dimension, candidate
// staged in ybuf.
template <int D, int TK>
MHX_DEV void mhx_emcee_half_body(const mhx_emcee_args& a, const float* __restrict__ tparams)
{
    const int W = a.nwalkers;
    const int halfW = W / 2;
    const int lo = a.half ? halfW : 0;
    const int cnt = a.half ? W - halfW : halfW;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const int i = lo + t;
    const int ostart = a.half ? 0 : halfW;
    const int osize = a.half ? halfW : W - halfW;
    const int d = D > 0 ? D : a.dim;
    const long ld = W;

    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_u32x4 w = mhx_philox(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep, MHX_STREAM_EMCEE << 28);
    // partner from the complementary half (src/emcee.jl:48,52 draws from all other walkers)
    const int j = ostart + (int)(((mhx_u64)w.x * (mhx_u64)(mhx_u32)osize) >> 32);
    // src/emcee.jl:81  z = ((a - 1) * rand(rng) + 1)^2 / a
    const float u = mhx_u01_half(w.y);
    const float tt = mhx_fma(a.stretch - 1.0f, u, 1.0f);
    const float z = (tt * tt) / a.stretch;
    const float alphamult = (float)(d - 1) * mhx_log(z);                // :82

    float lpy;
    float yreg[D > 0 ? D : 1];
    float* ys = a.ybuf + i;
    if (D > 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float xi = a.x[(long)k * ld + i];
            const float xj = a.x[(long)k * ld + j];
            yreg[k] = mhx_fma(z, xi - xj, xj);                          // :85
        }
        lpy = mhx_target_eval<TK>(TK, yreg, D, tparams, a.ntparams, a.tconst);
    } else {
        for (int k = 0; k < d; ++k) {
            const float xi = a.x[(long)k * ld + i];
            const float xj = a.x[(long)k * ld + j];
            ys[(long)k * ld] = mhx_fma(z, xi - xj, xj);
        }
        mhx_strided_x yv;
        yv.base = ys;
        yv.ld = ld;
        lpy = mhx_target_eval<TK>(a.target_kind, yv, d, tparams, a.ntparams, a.tconst);
    }
    const float lpi = a.lp[i];
    const float alpha = (alphamult + lpy) - lpi;                        // :91
    const float logu = mhx_log_pos(mhx_u01_open(w.z));
    const bool acc = logu <= alpha;                                     // :93  -randexp <= alpha (non-strict)
    if (acc) {
        if (D > 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) a.x[(long)k * ld + i] = yreg[k];
        } else {
            for (int k = 0; k < d; ++k) a.x[(long)k * ld + i] = ys[(long)k * ld];
        }
        a.lp[i] = lpy;
        a.acc_count[i] += 1u;
    }
    a.last_acc[i] = acc ? 1 : 0;
    if (a.save_slot >= 0) {
        float* row = a.samples + a.save_slot * (long)(d + 1) * ld + i;
        if (D > 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) row[(long)k * ld] = acc ? yreg[k] : a.x[(long)k * ld + i];
        } else {
            for (int k = 0; k < d; ++k) row[(long)k * ld] = a.x[(long)k * ld + i];
        }
        row[(long)d * ld] = acc ? lpy : lpi;
        a.accepted[a.save_slot * ld + i] = acc ? 1 : 0;
    }
    const mhx_u64 b = __ballot(acc);
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)__popcll(b));
}

// initial walkers (src/emcee.jl:6-8): W log-density evaluations, accepted = false
template <int TK>
MHX_DEV void mhx_emcee_init_body(const mhx_emcee_args& a, const float* __restrict__ tparams)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nwalkers) return;
    mhx_strided_x xv;
    xv.base = a.x + i;
    xv.ld = a.nwalkers;
    a.lp[i] = mhx_target_eval<TK>(a.target_kind, xv, a.dim, tparams, a.ntparams, a.tconst);
    a.acc_count[i] = 0u;
    a.last_acc[i] = 0;
}

#ifdef MHX_JIT_EMCEE
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_emcee_half(const mhx_emcee_args a, const float* __restrict__ tparams)
{
    mhx_emcee_half_body<MHX_JIT_DIM, MHX_JIT_TK>(a, tparams);
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_emcee_init(const mhx_emcee_args a, const float* __restrict__ tparams)
{
    mhx_emcee_init_body<MHX_JIT_TK>(a, tparams);
}
#endif
