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

// STREAM: both images are pre-built in global memory and walked through the LDS ring `ring` (mhx_rwmh_mfma_kernels.h), and x and
// grad(x) are not held in registers but re-read from their slabs when the candidate / the backward sum is formed and written back
// on accept: candidate, noise, w and the candidate's gradient are what a lane keeps.
template <int D, bool STREAM = false>
MHX_DEV void mhx_mala_mfma_body(const mhx_mala_args& a, const mhx_real* __restrict__ A, mhx_real* Aimg, mhx_real* ATimg,
                                mhx_real* ring = nullptr)
{
    typedef mhx_mfma_geom<D> GEO;
    constexpr int NS = GEO::NS;
    constexpr int NQD = (NS + 3) / 4;
    if constexpr (!STREAM) {
        mhx_mfma_image_fill<D>(A, Aimg);
        mhx_mfma_image_fill_T<D>(A, ATimg);
        __syncthreads();
    }
    typedef mhx_mfma_stream_geom<D, true, false> SGA;
    typedef mhx_mfma_stream_geom<D, true, true> SGT;
    constexpr int PFM = SGA::PF > SGT::PF ? SGA::PF : SGT::PF;             // the two images share prefetch registers and ring
    constexpr long BUFM = SGA::BUF_BYTES > SGT::BUF_BYTES ? SGA::BUF_BYTES : SGT::BUF_BYTES;
    mhx_piece16 pf[STREAM ? PFM : 1];
    int parity = 0;
    const mhx_srd sA = mhx_make_srd(STREAM ? Aimg : nullptr, (mhx_u32)(GEO::REALS * (long)sizeof(mhx_real)));
    const mhx_srd sAT = mhx_make_srd(STREAM ? ATimg : nullptr, (mhx_u32)(GEO::REALS_T * (long)sizeof(mhx_real)));
    if constexpr (STREAM) mhx_mfma_chunk_load<D, true, false>(sA, 0, pf);

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

    mhx_real xs[STREAM ? 1 : NS], gx[STREAM ? 1 : NS];
    if constexpr (!STREAM) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool in = 4 * s + 3 < D || 4 * s + g < D;
            xs[s] = in ? mhx_ld_off(a.x + (long)(4 * s) * ld, lane_off) : MHX_R(0.0);
            gx[s] = in ? mhx_ld_off(a.gx + (long)(4 * s) * ld, lane_off) : MHX_R(0.0);
        }
    }
    const mhx_srd xsrd = mhx_make_srd(a.x, (mhx_u32)D * ldb), gsrd = mhx_make_srd(a.gx, (mhx_u32)D * ldb);
    auto xat = [&](const int s) -> mhx_real {
        if constexpr (STREAM) return (4 * s + 3 < D || 4 * s + g < D) ? mhx_srd_load(xsrd, lane_off, (mhx_u32)(4 * s) * ldb) : MHX_R(0.0);
        else return xs[s];
    };
    auto gat = [&](const int s) -> mhx_real {
        if constexpr (STREAM) return (4 * s + 3 < D || 4 * s + g < D) ? mhx_srd_load(gsrd, lane_off, (mhx_u32)(4 * s) * ldb) : MHX_R(0.0);
        else return gx[s];
    };
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
                    ys[s] = mhx_fma(a.sigma, n[e], mhx_fma(a.h, gat(s), xat(s)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        fq = fq + __shfl_xor(fq, 16, 64);
        fq = fq + __shfl_xor(fq, 32, 64);
        // ---- value and gradient at the candidate (:73-75): w = A y, lp' = -1/2 |w|^2 + const, grad = -A^T w
        mhx_real w[NS], gy[NS];
        mhx_real q = MHX_R(0.0);
        if constexpr (STREAM) mhx_mfma_rows_stream<D, 3, true, true, BUFM>(sA, sAT, ring, lane, ys, q, w, pf, parity);
        else mhx_mfma_rows<D, 3>(Aimg, lane, ys, q, w);
        q = q + __shfl_xor(q, 16, 64);
        q = q + __shfl_xor(q, 32, 64);
        const mhx_real lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        if constexpr (STREAM) mhx_mfma_rows_T_stream<D, BUFM>(sAT, sA, ring, lane, w, gy, pf, parity);
        else mhx_mfma_rows_T<D>(ATimg, lane, w, gy);
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
                n[e] = s < NS ? mhx_fma(a.hs, gat(s) + gy[s], zs[s]) : MHX_R(0.0);      // the pad is zero throughout
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
        if constexpr (STREAM) {
            if (acc && valid) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (4 * s + 3 < D || 4 * s + g < D) {
                        mhx_srd_store(xsrd, lane_off, (mhx_u32)(4 * s) * ldb, ys[s]);
                        mhx_srd_store(gsrd, lane_off, (mhx_u32)(4 * s) * ldb, gy[s]);
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the slabs are current before anything reads them again
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) { xs[s] = acc ? ys[s] : xs[s]; gx[s] = acc ? gy[s] : gx[s]; }
        }
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && g == 0));
        if (step == save_next) {                                                      // wave-uniform
            if (valid) {
                mhx_real* slotp = a.samples + slot * (long)(D + 1) * ld;
                const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(D + 1) * (mhx_u32)ld * MHX_RB);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (4 * s + 3 < D || 4 * s + g < D) mhx_srd_store(srd, lane_off, (mhx_u32)(4 * s) * ldb, xat(s));
                    if (STREAM && (s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
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
        if constexpr (!STREAM) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (4 * s + 3 < D || 4 * s + g < D) {
                    mhx_st_off(a.x + (long)(4 * s) * ld, lane_off, xs[s]);
                    mhx_st_off(a.gx + (long)(4 * s) * ld, lane_off, gx[s]);
                }
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
#ifdef MHX_JIT_MALA_MFMA_STREAM
// the image builders (one block each, once per run) and the streamed kernel: dynamic LDS = the two-buffer ring
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_mfma_image(const mhx_real* __restrict__ packed, mhx_real* img) { mhx_mfma_image_fill<MHX_JIT_DIM>(packed, img); }
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_mfma_image_T(const mhx_real* __restrict__ packed, mhx_real* img) { mhx_mfma_image_fill_T<MHX_JIT_DIM>(packed, img); }
extern "C" __global__ void __launch_bounds__(64 * MHX_MFMA_WAVES, 1)
mhx_jit_mala_mfma_stream(const mhx_mala_args a, const mhx_real* __restrict__ tparams, mhx_real* gAimg, mhx_real* gATimg)
{
    extern __shared__ mhx_acc4 mhx_mala_mfma_ring[];
    mhx_mala_mfma_body<MHX_JIT_DIM, true>(a, tparams, gAimg, gATimg, (mhx_real*)mhx_mala_mfma_ring);
}
#endif
MHX_NS_END
