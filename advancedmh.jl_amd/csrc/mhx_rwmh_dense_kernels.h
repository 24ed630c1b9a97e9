// mhx_rwmh_dense_kernels.h -- random-walk Metropolis on the dense Gaussian target (CORR_GAUSS), L lanes per
// chain, for the dimensions whose fully unrolled lane-per-chain kernel no longer fits (d > 64) and for
// chain counts that leave a lane-per-chain grid at one wave per SIMD.
//
// Same step as mhx_rwmh_reg_body (src/mh-core.jl:92-117) and the same mat-vec machinery as the cooperative
// stretch move (mhx_emcee_kernels.h): the packed factor A = inv(chol(Sigma)) sits in LDS as a zero-padded
// float4 image, lane l of a chain's group owns rows l, l+L, ... of A y and the float4 slices l, l+L, ... of
// the state.  A float4 slice IS a Philox block (normals 4b..4b+3 of the step), so the proposal noise of a
// slice is drawn by the lane that owns it.  The kernel is persistent over the steps of a launch: the state
// stays in registers, the factor image is built once, and a chain's candidate row belongs to its own wave --
// no block barrier inside the step loop.  The reduction shape L is part of the arithmetic spec (the oracle's
// reduce_lanes), exactly as for the stretch move.
#pragma once
#include "mhx_rwmh_kernels.h"
#include "mhx_emcee_kernels.h"

template <int D, int L, int PK>
MHX_DEV void mhx_rwmh_dense_coop_body(const mhx_rwmh_args& a, const float* __restrict__ A,
                                      const float* __restrict__ pvec, float* ysh_all, mhx_e4* Ash4)
{
    typedef mhx_emcee_geom<D, L> GEO;
    constexpr int CPW = 64 / L;                  // chains per wave
    constexpr int NK = GEO::NK, NQ = GEO::NQ, NQL = GEO::NQL, DP4 = GEO::DP4;
    // ---- the factor image, once per launch
    {
        mhx_e4 areg[NK][GEO::maxit()];
        mhx_dense_image_load<D, L>(A, areg);
        mhx_dense_image_store<D, L>(areg, Ash4);
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int cw = lane & (CPW - 1);
    const int l = lane / CPW;
    const long c_raw = ((long)blockIdx.x * MHX_EMCEE_COOP_WAVES + wave) * CPW + cw;
    const bool valid = c_raw < a.nchains;
    const long c = valid ? c_raw : (long)a.nchains - 1;      // idle groups shadow the last chain (loads only)
    const long ld = a.ld;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    float* yrow = ysh_all + (wave * CPW + cw) * DP4;

    // ---- state: float4 slices l, l+L, ... of x (ABI layout [dim][ld], touched once per launch)
    mhx_e4 xs[NQL];
    float sc[NQL][4];                            // proposal scales of the owned dimensions (0 in the pad)
#pragma unroll
    for (int m = 0; m < NQL; ++m) {
        const int q4 = l + L * m;
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q4 + j;
            e[j] = (q4 < NQ && k < D) ? a.x[(long)k * ld + c] : 0.0f;
            sc[m][j] = (q4 < NQ && k < D) ? (PK == MHX_PROP_ISO ? a.pscale : pvec[k]) : 0.0f;
        }
        xs[m].x = e[0]; xs[m].y = e[1]; xs[m].z = e[2]; xs[m].w = e[3];
    }
    float lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        // ---- candidate: y = x + sigma z (src/proposal.jl:49-56); the pad of the row stays zero (scale 0, x 0)
        mhx_e4 ys[NQL];
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int q4 = l + L * m;
            const mhx_e4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
            ys[m] = zero4;
            if (q4 < NQ) {                                               // wave-uniform per m except the last slice
                float n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)q4, n);
                ys[m].x = mhx_fma(sc[m][0], n[0], xs[m].x);
                ys[m].y = mhx_fma(sc[m][1], n[1], xs[m].y);
                ys[m].z = mhx_fma(sc[m][2], n[2], xs[m].z);
                ys[m].w = mhx_fma(sc[m][3], n[3], xs[m].w);
            }
            if (q4 < DP4 / 4) ((mhx_e4*)yrow)[q4] = ys[m];
        }
        MHX_WAVE_SYNC();
        // ---- lp' = -1/2 |A y|^2 + const: rows l, l+L, ... by this lane, butterfly over the chain's lanes
        float q = mhx_dense_rows_sq<D, L>(Ash4, (const mhx_e4*)yrow, l);
#pragma unroll
        for (int off = 1; off < L; off <<= 1) q = q + __shfl_xor(q, off * CPW, 64);
        const float lpy = mhx_fma(-0.5f, q, a.tconst);
        MHX_WAVE_SYNC();                                             // the row is free for the next candidate
        // ---- accept (src/mh-core.jl:104-114); a zero-mean random walk has no Hastings term
        const float logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);
#pragma unroll
        for (int m = 0; m < NQL; ++m) xs[m] = acc ? ys[m] : xs[m];
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && l == 0));
        if (step == save_next) {                                         // wave-uniform
            if (valid) {
                float* row = a.samples + slot * (long)(D + 1) * ld + c;
#pragma unroll
                for (int m = 0; m < NQL; ++m) {
                    const int k = 4 * (l + L * m);
                    if (k + 0 < D) row[(long)(k + 0) * ld] = xs[m].x;
                    if (k + 1 < D) row[(long)(k + 1) * ld] = xs[m].y;
                    if (k + 2 < D) row[(long)(k + 2) * ld] = xs[m].z;
                    if (k + 3 < D) row[(long)(k + 3) * ld] = xs[m].w;
                }
                if (l == 0) {
                    row[(long)D * ld] = lp;
                    a.accepted[slot * ld + c] = acc ? 1 : 0;
                }
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    if (valid) {
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int k = 4 * (l + L * m);
            if (k + 0 < D) a.x[(long)(k + 0) * ld + c] = xs[m].x;
            if (k + 1 < D) a.x[(long)(k + 1) * ld + c] = xs[m].y;
            if (k + 2 < D) a.x[(long)(k + 2) * ld + c] = xs[m].z;
            if (k + 3 < D) a.x[(long)(k + 3) * ld + c] = xs[m].w;
        }
        if (l == 0) {
            a.lp[c] = lp;
            a.acc_count[c] = nacc;
            a.last_acc[c] = last ? 1 : 0;
        }
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

#ifdef MHX_JIT_RWMH_DENSE
extern "C" __global__ void __launch_bounds__(64 * MHX_EMCEE_COOP_WAVES)
mhx_jit_rwmh_dense(const mhx_rwmh_args a, const float* __restrict__ tparams, const float* __restrict__ pvec)
{
    __shared__ mhx_e4 ysh4[MHX_EMCEE_COOP_WAVES * (64 / MHX_JIT_L) * mhx_emcee_geom<MHX_JIT_DIM, MHX_JIT_L>::DP4 / 4];
    __shared__ mhx_e4 Ash4[mhx_emcee_geom<MHX_JIT_DIM, MHX_JIT_L>::TOTAL4];
    mhx_rwmh_dense_coop_body<MHX_JIT_DIM, MHX_JIT_L, MHX_JIT_PK>(a, tparams, pvec, (float*)ysh4, Ash4);
}
#endif
