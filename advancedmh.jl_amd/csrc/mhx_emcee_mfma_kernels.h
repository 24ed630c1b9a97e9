// mhx_emcee_mfma_kernels.h -- the stretch move on a DENSE precision factor with the row products on the matrix cores.
//
// The factor A = inv(chol Sigma) is the same for every walker, so A y over the 16 walkers of a wave is a GEMM (the geometry of
// mhx_rwmh_mfma_kernels.h): walker j = lane & 15 is spread over the four lanes g = lane >> 4, lane g owns the coordinates
// k = 4s + g -- exactly the B operand of k-step s of v_mfma_f64_16x16x4 / v_mfma_f32_16x16x4, so the candidate is FORMED in the
// operand layout and never goes through LDS -- and the C/D fragment leaves rows {16t + 4r + g} with lane g: the oracle's
// reduction shape 4 (lane partial sums in ascending row order, butterfly over lanes ^16, ^32).  Each MFMA is an exact fma chain
// in k order, so accumulator r of tile t after its k-steps IS the spec's w_i (see mhx_rwmh_mfma_kernels.h).
//
// What is different from the random-walk kernel: a half-step (or sweep) is ONE short launch, so an LDS image of the factor
// rebuilt per launch behind a block barrier is exactly the cost DESIGN 6.2 describes for the lane-group form.  Here the operand
// image is built ONCE per run in global memory (host side, mhx_emcee_create) and every lane fetches ITS operands -- TOTAL reals,
// contiguous per 4 k-steps -- straight into registers at the top of the kernel, behind the walker rows: no LDS at all, no barrier,
// no dependence between the waves of a block.  v_fmac_f64_dpp (the scalar-factor form) issues one 64-lane product per 8 cycles;
// a 16x16x4 MFMA does 1024 products in 32.
//
// SWEEP = false: one half-step per launch (in place, slices for a sharded ensemble); SWEEP = true: the whole sweep in one launch
// (mhx_emcee_coop_sweep_body has the idea: the second half's lanes re-do their partner's move from the old state, then move
// against its result; both products use the lane's A operands), state double-buffered.
#pragma once
#include "mhx_emcee_kernels.h"
#include "mhx_rwmh_mfma_kernels.h"

MHX_NS_BEGIN

#ifndef MHX_EMCEE_MFMA_WAVES
#define MHX_EMCEE_MFMA_WAVES 1                    // waves per block (16 walkers each)
#endif
// Round 5: what the per-launch fetch of the operands costs was measured (tools/c3_mfma_probe.py, profiles/r05e_c3_mfma_probe.log: every
// lane fetching its groups from the same 2 KB instead runs the C3 sweep in 8.47 us against 10.20) -- 19 KB per wave x 1024 waves =
// 19 MB out of the L2s per sweep, 1.7 us of a 10-us launch.  With more than one wave per block the waves therefore SHARE the fetch:
// wave w brings groups w, w + WAVES, ... into the block's LDS (plain loads issued behind the walker rows), one barrier, and every lane
// takes its operands from there -- the L2 traffic of the image falls by the number of waves per block.  (One wave per block: straight
// into registers as before, no LDS, no barrier.)

// rows of (factor operands in registers) x b, two tiles at a time; q = sum of squares of this lane's rows in ascending order
template <int D>
MHX_DEV mhx_real mhx_mfma_rows_sq_areg(const mhx_acc4 (&areg)[mhx_mfma_geom<D>::TOTAL / 4], const mhx_real (&b)[mhx_mfma_geom<D>::NS])
{
    typedef mhx_mfma_geom<D> GEO;
    mhx_real q = MHX_R(0.0);
#pragma unroll
    for (int t0 = 0; t0 < GEO::NT; t0 += 2) {
        constexpr mhx_acc4 zero = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
        mhx_acc4 c[2] = {zero, zero};
        const int t1 = t0 + 1 < GEO::NT ? t0 + 1 : t0;
#pragma unroll
        for (int grp = 0; grp < GEO::groups(t1); ++grp) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = t0 + h;
                if (t < GEO::NT && grp < GEO::groups(t)) {
                    const mhx_acc4 a4 = areg[GEO::first(t) / 4 + grp];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * grp + u < GEO::steps(t)) c[h] = MHX_MFMA16(a4[u], b[4 * grp + u], c[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + h;
            if (t < GEO::NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) q = mhx_fma(c[h][r], c[h][r], q);       // rows >= D are zero rows of the image
            }
        }
    }
    return q;
}

template <int D, bool SWEEP>
MHX_DEV void mhx_emcee_mfma_body(const mhx_emcee_args& a, const mhx_real* __restrict__ img)
{
    typedef mhx_mfma_geom<D> GEO;
    constexpr int NS = GEO::NS;                       // coordinates per lane (k = 4s + g)
    constexpr int NG = GEO::TOTAL / 4;                // operand groups per lane
    constexpr int WPB = 16 * MHX_EMCEE_MFMA_WAVES;    // walkers per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jw = lane & 15, g = lane >> 4;
    const int W = a.nwalkers;
    const int halfW = W / 2, cntB = W - halfW;
    bool second;
    int t_raw, cnt;
    if constexpr (SWEEP) {
        const int nbB = (cntB + WPB - 1) / WPB;
        second = (int)blockIdx.x < nbB;                                          // (block-uniform) blocks [0, nbB): the second half
        const int blk = second ? (int)blockIdx.x : (int)blockIdx.x - nbB;
        cnt = second ? cntB : halfW;
        t_raw = blk * WPB + wave * 16 + jw;
    } else {
        second = a.half != 0;
        cnt = second ? cntB : halfW;
        t_raw = a.t_begin + (int)blockIdx.x * WPB + wave * 16 + jw;
    }
    const bool valid = t_raw < cnt && (SWEEP || t_raw < a.t_begin + a.t_count);
    const int lo = second ? halfW : 0;
    const int i = lo + (valid ? t_raw : cnt - 1);
    const long ld = a.ld;
    constexpr int PITCH = mhx_xw_pitch(D);
    // ---- the chain of latencies first: own row, draws, partner row(s); then this lane's operands of the factor
    mhx_real xs[NS], xj[NS], y0[NS];
    const mhx_real* xrow_i = a.xw + (long)i * PITCH + g;
#pragma unroll
    for (int s = 0; s < NS; ++s) xs[s] = xrow_i[4 * s];                          // (the zero pad of a row is part of it)
    const mhx_real lpi = a.lp[i];
    const mhx_u32 acc_i = a.acc_count[i];
    bool moved_before = true;
    if constexpr (SWEEP) moved_before = a.all_rows != 0 || a.last_acc[i] != 0;   // xw_out does not hold this walker's row
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const mhx_emcee_draws dr = mhx_emcee_draw(ks, (mhx_u32)i, (mhx_u32)a.ensemble_id, a.sweep);
    // the partner: the second half draws from the first ([0, halfW)), the first from the second
    const int j = (second ? 0 : halfW) + (int)(((mhx_u64)dr.partner * (mhx_u64)(mhx_u32)(second ? halfW : cntB)) >> 32);
    const mhx_real* xrow_j = a.xw + (long)j * PITCH + g;
#pragma unroll
    for (int s = 0; s < NS; ++s) xj[s] = xrow_j[4 * s];
    const bool redo = SWEEP && second;                                           // (block-uniform) the partner's own move first
    mhx_emcee_draws da = dr;
    mhx_real lpa = MHX_R(0.0);
    mhx_real xb[SWEEP ? NS : 1];
    if (redo) {
        da = mhx_emcee_draw(ks, (mhx_u32)j, (mhx_u32)a.ensemble_id, a.sweep);    // j is of the first half: its draws of this sweep
        const int jb = halfW + (int)(((mhx_u64)da.partner * (mhx_u64)(mhx_u32)cntB) >> 32);
        const mhx_real* xrow_b = a.xw + (long)jb * PITCH + g;
#pragma unroll
        for (int s = 0; s < NS; ++s) xb[SWEEP ? s : 0] = xrow_b[4 * s];
        lpa = a.lp[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    mhx_acc4 areg[NG];
    if constexpr (MHX_EMCEE_MFMA_WAVES > 1) {
        extern __shared__ mhx_acc4 mhx_emcee_mfma_opnd[];                        // [NG][64] acc4: the operand image, once per block
        constexpr int NQ = (NG + MHX_EMCEE_MFMA_WAVES - 1) / MHX_EMCEE_MFMA_WAVES;
        mhx_acc4 part[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = wave + u * MHX_EMCEE_MFMA_WAVES;
            part[u] = ((const mhx_acc4*)img)[(q < NG ? q : NG - 1) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = wave + u * MHX_EMCEE_MFMA_WAVES;
            if (q < NG) mhx_emcee_mfma_opnd[q * 64 + lane] = part[u];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NG; ++q) areg[q] = mhx_emcee_mfma_opnd[q * 64 + lane];
    } else {
#pragma unroll
        for (int q = 0; q < NG; ++q) areg[q] = ((const mhx_acc4*)img)[q * 64 + lane];
    }
#ifdef MHX_TOOLS_BUILD
#if defined(MHX_EMCEE_MFMA_PROBE) && MHX_EMCEE_MFMA_PROBE == 1     // timing probe (tools build, JIT_DEFS): every group from the SAME 2 KB -- what the operand fetch costs
#pragma unroll
    for (int q = 0; q < NG; ++q) areg[q] = ((const mhx_acc4*)img)[lane];
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
    const mhx_real tt = mhx_fma(a.stretch - MHX_R(1.0), dr.u, MHX_R(1.0));
    const mhx_real z = (tt * tt) / a.stretch;                                    // src/emcee.jl:81
    const mhx_real alphamult = (mhx_real)(D - 1) * mhx_log(z);                   // :82
    mhx_real lpy;
    if (!redo) {
#pragma unroll
        for (int s = 0; s < NS; ++s) y0[s] = mhx_fma(z, xs[s] - xj[s], xj[s]);   // :85
        const mhx_real q = mhx_butterfly_add<32>(mhx_butterfly_add<16>(mhx_mfma_rows_sq_areg<D>(areg, y0)));
        lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
    } else {
        const mhx_real ta = mhx_fma(a.stretch - MHX_R(1.0), da.u, MHX_R(1.0));
        const mhx_real za = (ta * ta) / a.stretch;
        const mhx_real alphamult_a = (mhx_real)(D - 1) * mhx_log(za);
        mhx_real ya[NS], y1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ya[s] = mhx_fma(za, xj[s] - xb[SWEEP ? s : 0], xb[SWEEP ? s : 0]);   // the partner's candidate
            y0[s] = mhx_fma(z, xs[s] - xj[s], xj[s]);                            // this walker's, if the partner stays
            y1[s] = mhx_fma(z, xs[s] - ya[s], ya[s]);                            //                if it moves
        }
        // (no speculation here, unlike the lane-group form: one wave carries the whole product of its walkers, so a third product is
        // a third of a microsecond-long MFMA chain -- the partner's test first, then the one candidate it selects)
        const mhx_real qa = mhx_butterfly_add<32>(mhx_butterfly_add<16>(mhx_mfma_rows_sq_areg<D>(areg, ya)));
        const mhx_real lpya = mhx_fma(-MHX_R(0.5), qa, a.tconst);
        const bool acc_a = da.logu <= (alphamult_a + lpya) - lpa;                // the partner's accept test, as its own lanes run it
#pragma unroll
        for (int s = 0; s < NS; ++s) y0[s] = acc_a ? y1[s] : y0[s];
        const mhx_real qb = mhx_butterfly_add<32>(mhx_butterfly_add<16>(mhx_mfma_rows_sq_areg<D>(areg, y0)));
        lpy = mhx_fma(-MHX_R(0.5), qb, a.tconst);
    }
    const mhx_real alpha = (alphamult + lpy) - lpi;                              // :91
    const bool acc = dr.logu <= alpha;                                           // :93
    if (!valid) return;
    if constexpr (SWEEP) {
#pragma unroll
        for (int s = 0; s < NS; ++s) y0[s] = acc ? y0[s] : xs[s];
        if (acc || moved_before) {
            mhx_real* xrow_o = a.xw_out + (long)i * PITCH + g;
#pragma unroll
            for (int s = 0; s < NS; ++s) xrow_o[4 * s] = y0[s];
        }
        if (g == 0) {
            a.lp_out[i] = acc ? lpy : lpi;
            if (acc) a.acc_count[i] = acc_i + 1u;
            a.last_acc[i] = acc ? 1 : 0;
        }
    } else {
        if (acc) {
            mhx_real* xrow_o = a.xw + (long)i * PITCH + g;
#pragma unroll
            for (int s = 0; s < NS; ++s) xrow_o[4 * s] = y0[s];
            if (g == 0) { a.lp[i] = lpy; a.acc_count[i] = acc_i + 1u; }
        }
        if (g == 0) a.last_acc[i] = acc ? 1 : 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) y0[s] = acc ? y0[s] : xs[s];
    }
    if (a.save_slot >= 0) {
        // row k = 4s + g of the [dim+1][W] record for the wave's 16 consecutive walkers: 16 x sizeof(real) contiguous bytes per store
        mhx_real* row = a.samples + a.save_slot * (long)(D + 1) * ld + i;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (4 * s + 3 < D || 4 * s + g < D) MHX_REC_ST(&row[(long)(4 * s + g) * ld], y0[s]);
        if (g == 0) {
            MHX_REC_ST(&row[(long)D * ld], acc ? lpy : lpi);
            a.accepted[a.save_slot * ld + i] = acc ? 1 : 0;
        }
    }
}

#ifdef MHX_JIT_EMCEE_MFMA
extern "C" __global__ void __launch_bounds__(64 * MHX_EMCEE_MFMA_WAVES)
mhx_jit_emcee_mfma_half(const mhx_emcee_args a_, const mhx_real* __restrict__ img)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    mhx_emcee_mfma_body<MHX_JIT_DIM, false>(a, img);
}
extern "C" __global__ void __launch_bounds__(64 * MHX_EMCEE_MFMA_WAVES)
mhx_jit_emcee_mfma_sweep(const mhx_emcee_args a_, const mhx_real* __restrict__ img)
{
    const mhx_emcee_args a = mhx_emcee_pick(a_);
    mhx_emcee_mfma_body<MHX_JIT_DIM, true>(a, img);
}
#endif
MHX_NS_END
