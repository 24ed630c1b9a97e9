// mhx_mala_mfma_kernels.h -- Metropolis-adjusted Langevin on the dense Gaussian target, on the matrix cores.
//
// The target's value and gradient are two products with the ONE factor all chains share: w = A y (rows of the lower
// triangle) and grad = -A^T w (rows of the upper triangle) -- over the 16 chains of a wave two triangular GEMMs on
// v_mfma_*_16x16x4, each accumulator an exact k-ordered fma chain = the arithmetic spec's sums (ascending index, from 0).
// Geometry of mhx_rwmh_mfma_kernels.h: 4 lanes per chain, lane g owns the dimensions 4s + g; the fragment of A y is
// already the B operand of A^T w, and the fragment of A^T w lands on the lanes that own its dimensions.  State x, grad(x),
// candidate, its gradient, the noise and w live in registers for the steps of a launch.  The three sums of a step take
// the reduction shape 4: |A y|^2 by rows g, g+4, ...; |z|^2 and |z + (sigma/2)(grad x + grad y)|^2 by Philox blocks
// g, g+4, ... (the shape of the cooperative kernel), each followed by the butterfly over the chain's lanes.
//
// Same step as mhx_mala_reg_body (src/MALA.jl:54-93).
#pragma once
#include "mhx_mala_kernels.h"
#include "mhx_rwmh_mfma_kernels.h"

MHX_NS_BEGIN

template <int D>
MHX_DEV void mhx_mala_mfma_body(const mhx_mala_args& a, const mhx_real* __restrict__ A, mhx_real* Aimg, mhx_real* ATimg)
{
    typedef mhx_mfma_geom<D> GEO;
    constexpr int NS = GEO::NS;
    constexpr int NQD = (NS + 3) / 4;
    mhx_mfma_image_fill<D>(A, Aimg);
    mhx_mfma_image_fill_T<D>(A, ATimg);
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;
    const long c_raw = ((long)blockIdx.x * MHX_MFMA_WAVES + wave) * 16 + j;
    const bool valid = c_raw < a.nchains;
    const long c = valid ? c_raw : (long)a.nchains - 1;
    const long ld = a.ld;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_u32 lane_off = ((mhx_u32)g * (mhx_u32)ld + (mhx_u32)c) * MHX_RB;
    const mhx_u32 ldb = (mhx_u32)ld * MHX_RB;

    mhx_real xs[NS], gx[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool in = 4 * s + 3 < D || 4 * s + g < D;
        xs[s] = in ? mhx_ld_off(a.x + (long)(4 * s) * ld, lane_off) : MHX_R(0.0);
        gx[s] = in ? mhx_ld_off(a.gx + (long)(4 * s) * ld, lane_off) : MHX_R(0.0);
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

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        asm volatile("" ::: "memory");        // the factor images are re-read from LDS every step, not kept in registers
        // ---- noise and candidate (src/MALA.jl:70): lane g draws the blocks 4qd + g and sums their squares (|z|^2 in the
        // block shape), the transpose hands z of dimension 4(4qd + e) + g to n[e]
        mhx_real ys[NS], zs[NS];
        mhx_real fq = MHX_R(0.0);
#pragma unroll
        for (int qd = 0; qd < NQD; ++qd) {
            mhx_real n[4];
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)(4 * qd + g), n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k0 = 4 * (4 * qd) + e;                               // dimension 4 (4qd + g) + e
                if (k0 + 12 >= D) n[e] = (k0 + 4 * g < D) ? n[e] : MHX_R(0.0);     // the pad draws nothing
                fq = mhx_fma(n[e], n[e], fq);
            }
            mhx_lanes4_transpose(n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 4 * qd + e;
                if (s < NS) {
                    zs[s] = n[e];
                    ys[s] = mhx_fma(a.sigma, n[e], mhx_fma(a.h, gx[s], xs[s]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        fq = fq + __shfl_xor(fq, 16, 64);
        fq = fq + __shfl_xor(fq, 32, 64);
        // ---- value and gradient at the candidate (:73-75): w = A y, lp' = -1/2 |w|^2 + const, grad = -A^T w
        mhx_real w[NS], gy[NS];
        mhx_real q = MHX_R(0.0);
        mhx_mfma_rows<D, 3>(Aimg, lane, ys, q, w);
        q = q + __shfl_xor(q, 16, 64);
        q = q + __shfl_xor(q, 32, 64);
        const mhx_real lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        mhx_mfma_rows_T<D>(ATimg, lane, w, gy);
#pragma unroll
        for (int s = 0; s < NS; ++s) gy[s] = -gy[s];
        // ---- log ratio of the proposal densities (:78-80): |z + (sigma/2)(grad x + grad y)|^2 in the block shape
        mhx_real bq = MHX_R(0.0);
#pragma unroll
        for (int qd = 0; qd < NQD; ++qd) {
            mhx_real n[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 4 * qd + e;
                n[e] = s < NS ? mhx_fma(a.hs, gx[s] + gy[s], zs[s]) : MHX_R(0.0);      // the pad is zero throughout
            }
            mhx_lanes4_transpose(n);                                                  // block 4qd + g, element e
#pragma unroll
            for (int e = 0; e < 4; ++e) bq = mhx_fma(n[e], n[e], bq);
        }
        bq = bq + __shfl_xor(bq, 16, 64);
        bq = bq + __shfl_xor(bq, 32, 64);
        const mhx_real loga = (lpy - lp) + MHX_R(0.5) * (fq - bq);                    // :83
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;                                                 // :86 (strict)
#pragma unroll
        for (int s = 0; s < NS; ++s) { xs[s] = acc ? ys[s] : xs[s]; gx[s] = acc ? gy[s] : gx[s]; }
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && g == 0));
        if (step == save_next) {                                                      // wave-uniform
            if (valid) {
                mhx_real* slotp = a.samples + slot * (long)(D + 1) * ld;
                const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(D + 1) * (mhx_u32)ld * MHX_RB);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (4 * s + 3 < D || 4 * s + g < D) mhx_srd_store(srd, lane_off, (mhx_u32)(4 * s) * ldb, xs[s]);
                if (g == 0) {
                    slotp[(long)D * ld + c] = lp;
                    a.accepted[slot * ld + c] = acc ? 1 : 0;
                }
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    if (valid) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (4 * s + 3 < D || 4 * s + g < D) {
                mhx_st_off(a.x + (long)(4 * s) * ld, lane_off, xs[s]);
                mhx_st_off(a.gx + (long)(4 * s) * ld, lane_off, gx[s]);
            }
        if (g == 0) {
            a.lp[c] = lp;
            a.acc_count[c] = nacc;
            a.last_acc[c] = last ? 1 : 0;
        }
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

#ifdef MHX_JIT_MALA_MFMA
#ifndef MHX_JIT_WAVES
#define MHX_JIT_WAVES 1
#endif
// dynamic LDS: [image of A][image of A^T]
extern "C" __global__ void __launch_bounds__(64 * MHX_MFMA_WAVES, MHX_JIT_WAVES)
mhx_jit_mala_mfma(const mhx_mala_args a, const mhx_real* __restrict__ tparams)
{
    typedef mhx_mfma_geom<MHX_JIT_DIM> GEO;
    extern __shared__ mhx_acc4 mhx_mala_mfma_lds[];
    mhx_real* Aimg = (mhx_real*)mhx_mala_mfma_lds;
    mhx_mala_mfma_body<MHX_JIT_DIM>(a, tparams, Aimg, Aimg + GEO::REALS);
}
#endif
MHX_NS_END
