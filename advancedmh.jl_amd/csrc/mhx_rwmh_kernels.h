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

#define MHX_PROP_ISO   0
#define MHX_PROP_DIAG  1
#define MHX_PROP_DENSE 2
#define MHX_PROP_DYNAMIC (-1)

#define MHX_NO_SAVE 0xffffffffu

struct mhx_rwmh_args {
    float* x;                 // [dim][ld]   chain state
    float* lp;                // [ld]
    mhx_u32* acc_count;       // [ld]        per-chain accepted proposals
    mhx_u64* acc_total;       // [1]         all-chain accepted proposals (ballot + popcount + 1 atomic / wave)
    float* samples;           // [slots][dim+1][ld] or null
    unsigned char* accepted;  // [slots][ld] or null
    unsigned char* last_acc;  // [ld]        accepted flag of each chain's latest transition
    float* ybuf;              // [dim][ld]   candidate scratch (generic kernel)
    mhx_u64 seed;
    mhx_u64 first_chain;
    int nchains;
    int ld;
    int dim;
    int target_kind;
    int ntparams;
    float tconst;
    int prop_kind;
    float pscale;
    mhx_u32 step0;            // first transition index of this launch (RNG step counter)
    int nsteps;
    mhx_u32 save_next;        // first transition >= step0 whose state is recorded (MHX_NO_SAVE: none)
    int save_slot;            // its slot in `samples`
    int thinning;
};

// ---------------------------------------------------------------------------------------------
template <int D, int TK, int PK>
MHX_DEV void mhx_rwmh_reg_body(const mhx_rwmh_args& a, const float* __restrict__ tparams,
                               const float* __restrict__ pvec)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;

    float x[D], y[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = a.x[(long)k * ld + c];
    float lp = a.lp[c];
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
            float z[D];
#pragma unroll
            for (int b = 0; b < (D + 3) / 4; ++b) {
                float n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (4 * b + j < D) z[4 * b + j] = n[j];
            }
            int off = 0;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                float w = 0.0f;
#pragma unroll
                for (int j = 0; j <= r; ++j) w = mhx_fma(pvec[off + j], z[j], w);
                y[r] = x[r] + w;
                off += r + 1;
            }
        } else {
#pragma unroll
            for (int b = 0; b < (D + 3) / 4; ++b) {
                float n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * b + j;
                    if (k < D) y[k] = mhx_fma(PK == MHX_PROP_ISO ? a.pscale : pvec[k], n[j], x[k]);
                }
            }
        }
        // ---- log-density of the candidate and the accept test (src/mh-core.jl:103-108)
        const float lpy = mhx_target_eval<TK>(TK, y, D, tparams, a.ntparams, a.tconst);
        const float logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);          // strict; NaN compares false => reject
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = acc ? y[k] : x[k];
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc));
        // ---- record (ext/AdvancedMHMCMCChainsExt.jl:96-105 layout, chain fastest)
        if (step == save_next) {
            float* row = a.samples + slot * (long)(D + 1) * ld + c;
#pragma unroll
            for (int k = 0; k < D; ++k) row[(long)k * ld] = x[k];
            row[(long)D * ld] = lp;
            a.accepted[slot * ld + c] = acc ? 1 : 0;
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) a.x[(long)k * ld + c] = x[k];
    a.lp[c] = lp;
    a.acc_count[c] = nacc;
    a.last_acc[c] = last ? 1 : 0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// ---------------------------------------------------------------------------------------------
template <int TK>
MHX_DEV void mhx_rwmh_generic_body(const mhx_rwmh_args& a, const float* __restrict__ tparams,
                                   const float* __restrict__ pvec)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;
    const int d = a.dim;
    float* xs = a.x + c;
    float* ys = a.ybuf + c;

    float lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    const int nblk = (d + 3) >> 2;

    for (int i = 0; i < a.nsteps; ++i) {
        const mhx_u32 step = a.step0 + (mhx_u32)i;
        if (a.prop_kind == MHX_PROP_DENSE) {
            for (int b = 0; b < nblk; ++b) {
                float n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
                for (int j = 0; j < 4; ++j) if (4 * b + j < d) ys[(long)(4 * b + j) * ld] = n[j];
            }
            // y_r = x_r + sum_{j<=r} L_rj z_j ; rows in descending order so z can be overwritten in place
            for (int r = d - 1; r >= 0; --r) {
                const float* Lr = pvec + (long)r * (r + 1) / 2;
                float w = 0.0f;
                for (int j = 0; j <= r; ++j) w = mhx_fma(Lr[j], ys[(long)j * ld], w);
                ys[(long)r * ld] = xs[(long)r * ld] + w;
            }
        } else {
            for (int b = 0; b < nblk; ++b) {
                float n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * b + j;
                    if (k < d) {
                        const float s = a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k];
                        ys[(long)k * ld] = mhx_fma(s, n[j], xs[(long)k * ld]);
                    }
                }
            }
        }
        mhx_strided_x yv;
        yv.base = ys;
        yv.ld = ld;
        const float lpy = mhx_target_eval<TK>(a.target_kind, yv, d, tparams, a.ntparams, a.tconst);
        const float logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc));
        if (step == save_next) {
            float* row = a.samples + slot * (long)(d + 1) * ld + c;
            for (int k = 0; k < d; ++k) {
                const float v = acc ? ys[(long)k * ld] : xs[(long)k * ld];
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
    a.acc_count[c] = nacc;
    a.last_acc[c] = last ? 1 : 0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// ---------------------------------------------------------------------------------------------
// initial state (src/mh-core.jl:83-84): x0 = initial_params, or a bare proposal draw
// (src/proposal.jl:41-47) from Philox stream INIT; lp0 = logdensity(model, x0).
template <int TK>
MHX_DEV void mhx_rwmh_init_body(const mhx_rwmh_args& a, const float* __restrict__ tparams,
                                const float* __restrict__ pvec, const int draw)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const long ld = a.ld;
    const int d = a.dim;
    float* xs = a.x + c;
    if (draw) {
        const mhx_philox_key ks = mhx_philox_schedule(a.seed);
        const int nblk = (d + 3) >> 2;
        for (int b = 0; b < nblk; ++b) {
            float n[4];
            mhx_normal4(ks, id_lo, id_hi, 0u, MHX_STREAM_INIT, (mhx_u32)b, n);
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k < d) {
                    if (a.prop_kind == MHX_PROP_DENSE) xs[(long)k * ld] = n[j];
                    else xs[(long)k * ld] = mhx_fma(a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k], n[j], 0.0f);
                }
            }
        }
        if (a.prop_kind == MHX_PROP_DENSE) {
            for (int r = d - 1; r >= 0; --r) {
                const float* Lr = pvec + (long)r * (r + 1) / 2;
                float w = 0.0f;
                for (int j = 0; j <= r; ++j) w = mhx_fma(Lr[j], xs[(long)j * ld], w);
                xs[(long)r * ld] = 0.0f + w;
            }
        }
    }
    mhx_strided_x xv;
    xv.base = xs;
    xv.ld = ld;
    a.lp[c] = mhx_target_eval<TK>(a.target_kind, xv, d, tparams, a.ntparams, a.tconst);
    a.acc_count[c] = 0u;
    a.last_acc[c] = 0;          // Transition(params, lp, false), src/mh-core.jl:84
}

// ---------------------------------------------------------------------------------------------
// logdensity(model, x) for a batch of points, x [dim][n] -> lp [n]   (src/AdvancedMH.jl:74)
template <int TK>
MHX_DEV void mhx_target_eval_body(const float* __restrict__ x, float* __restrict__ lp, const int n,
                                  const int d, const int kind, const float* __restrict__ tparams,
                                  const int ntparams, const float tconst)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    mhx_strided_x xv;
    xv.base = x + c;
    xv.ld = n;
    lp[c] = mhx_target_eval<TK>(kind, xv, d, tparams, ntparams, tconst);
}

// record the current state into sample slot `slot` (sample 1 of a run with discard_initial == 0)
MHX_DEV void mhx_record_state_body(const float* __restrict__ x, const float* __restrict__ lp,
                                   const unsigned char* __restrict__ last_acc, float* samples,
                                   unsigned char* accepted, const int n, const long ld, const int d,
                                   const long slot)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float* row = samples + slot * (long)(d + 1) * ld + c;
    for (int k = 0; k < d; ++k) row[(long)k * ld] = x[(long)k * ld + c];
    row[(long)d * ld] = lp[c];
    accepted[slot * ld + c] = last_acc[c];
}

// JIT entry points: hiprtc compiles this header with the specialisation macros defined
#ifdef MHX_JIT_RWMH_REG
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_rwmh_reg(const mhx_rwmh_args a, const float* __restrict__ tparams, const float* __restrict__ pvec)
{
    mhx_rwmh_reg_body<MHX_JIT_DIM, MHX_JIT_TK, MHX_JIT_PK>(a, tparams, pvec);
}
#endif
#ifdef MHX_JIT_RWMH_GENERIC
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_rwmh_generic(const mhx_rwmh_args a, const float* __restrict__ tparams, const float* __restrict__ pvec)
{
    mhx_rwmh_generic_body<MHX_JIT_TK>(a, tparams, pvec);
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_rwmh_init(const mhx_rwmh_args a, const float* __restrict__ tparams, const float* __restrict__ pvec,
                  const int draw)
{
    mhx_rwmh_init_body<MHX_JIT_TK>(a, tparams, pvec, draw);
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_target_eval(const float* __restrict__ x, float* __restrict__ lp, const int n, const int d,
                    const int kind, const float* __restrict__ tparams, const int ntparams, const float tconst)
{
    mhx_target_eval_body<MHX_JIT_TK>(x, lp, n, d, kind, tparams, ntparams, tconst);
}
#endif
