// mhx_rwmh_dense_kernels.h -- random-walk Metropolis with a dense factor in play -- the dense Gaussian target
// (CORR_GAUSS), a dense (Cholesky) proposal, or both -- L lanes per chain: for the dimensions whose fully unrolled
// lane-per-chain kernel no longer fits and for chain counts that leave a lane-per-chain grid at one wave per SIMD.
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

MHX_NS_BEGIN

// TK: MHX_TARGET_CORR_GAUSS (factor image A) or MHX_TARGET_ISO_GAUSS; PK: ISO / DIAG scales, or DENSE -- the
// proposal's Cholesky factor as a second image: xi = L z is a row product like A y, and y = x + xi.
template <int D, int L, int PK, int TK>
MHX_DEV void mhx_rwmh_dense_coop_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ A,
                                      const mhx_real* __restrict__ pvec, mhx_real* ysh_all, mhx_e4* Ash4, mhx_e4* Lsh4)
{
    typedef mhx_emcee_geom<D, L> GEO;
    constexpr bool CORR = TK == MHX_TARGET_CORR_GAUSS;
    constexpr bool DENSEP = PK == MHX_PROP_DENSE;
    constexpr int CPW = 64 / L;                  // chains per wave
    constexpr int NK = GEO::NK, NQ = GEO::NQ, NQL = GEO::NQL, DP4 = GEO::DP4;
    // ---- the factor images, once per launch
    if (CORR) mhx_dense_image_fill<D, L>(A, Ash4);
    if (DENSEP) mhx_dense_image_fill<D, L>(pvec, Lsh4);
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
    mhx_real* yrow = ysh_all + (wave * CPW + cw) * DP4;
    mhx_e4* yrow4 = (mhx_e4*)yrow;

    // ---- state: float4 slices l, l+L, ... of x (ABI layout [dim][ld], touched once per launch)
    mhx_e4 xs[NQL];
    mhx_real sc[NQL][4];                            // ISO / DIAG: scales of the owned dimensions; DENSE: 1 (0 in the pad)
#pragma unroll
    for (int m = 0; m < NQL; ++m) {
        const int q4 = l + L * m;
        mhx_real e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q4 + j;
            const bool in = q4 < NQ && k < D;
            e[j] = in ? a.x[(long)k * ld + c] : MHX_R(0.0);
            sc[m][j] = in ? (PK == MHX_PROP_ISO ? a.pscale : (DENSEP ? MHX_R(1.0) : pvec[k])) : MHX_R(0.0);
        }
        xs[m].x = e[0]; xs[m].y = e[1]; xs[m].z = e[2]; xs[m].w = e[3];
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
        // ---- candidate (src/proposal.jl:49-56).  ISO / DIAG: y = fma(sigma, z, x) on the owned slices.  DENSE: the
        // slices of z go to the chain's LDS row, xi = L z comes back by rows, is handed over through the same row
        // and y = x + xi.  The pad of every row stays zero (scale 0, x 0).
        mhx_e4 ys[NQL];
#pragma unroll
        for (int m = 0; m < NQL; ++m) {
            const int q4 = l + L * m;
            const mhx_e4 zero4 = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
            ys[m] = zero4;
            if (q4 < NQ) {
                mhx_real n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)q4, n);
                if (DENSEP) {
                    ys[m].x = sc[m][0] * n[0]; ys[m].y = sc[m][1] * n[1]; ys[m].z = sc[m][2] * n[2]; ys[m].w = sc[m][3] * n[3];
                } else {
                    ys[m].x = mhx_fma(sc[m][0], n[0], xs[m].x);
                    ys[m].y = mhx_fma(sc[m][1], n[1], xs[m].y);
                    ys[m].z = mhx_fma(sc[m][2], n[2], xs[m].z);
                    ys[m].w = mhx_fma(sc[m][3], n[3], xs[m].w);
                }
            }
            if (q4 < DP4 / 4) yrow4[q4] = ys[m];
        }
        MHX_WAVE_SYNC();
        if (DENSEP) {
            mhx_real xi[NK];
            mhx_dense_rows<D, L>(Lsh4, yrow4, l, xi);                    // xi_r = sum_{j<=r} L_rj z_j, ascending j
            MHX_WAVE_SYNC();                                             // every lane has read z
#pragma unroll
            for (int m = 0; m < NK; ++m) { const int r = l + L * m; if (r < D) yrow[r] = xi[m]; }
            MHX_WAVE_SYNC();
#pragma unroll
            for (int m = 0; m < NQL; ++m) {
                const int q4 = l + L * m;
                if (q4 < NQ) {
                    const mhx_e4 w4 = yrow4[q4];                          // the pad still holds the zeros of z
                    ys[m].x = xs[m].x + w4.x; ys[m].y = xs[m].y + w4.y; ys[m].z = xs[m].z + w4.z; ys[m].w = xs[m].w + w4.w;
                }
            }
            MHX_WAVE_SYNC();
#pragma unroll
            for (int m = 0; m < NQL; ++m) { const int q4 = l + L * m; if (q4 < NQ) yrow4[q4] = ys[m]; }
            MHX_WAVE_SYNC();
        }
        // ---- lp': dense Gaussian -1/2 |A y|^2 + const (rows l, l+L, ... by this lane); isotropic -1/2 |y|^2 + const
        // (the lane's slices in ascending order); butterfly over the chain's lanes
        mhx_real q = MHX_R(0.0);
        if (CORR) {
            q = mhx_dense_rows_sq<D, L>(Ash4, yrow4, l);
        } else {
#pragma unroll
            for (int m = 0; m < NQL; ++m) {
                const int k = 4 * (l + L * m);
                if (k + 0 < D) q = mhx_fma(ys[m].x, ys[m].x, q);
                if (k + 1 < D) q = mhx_fma(ys[m].y, ys[m].y, q);
                if (k + 2 < D) q = mhx_fma(ys[m].z, ys[m].z, q);
                if (k + 3 < D) q = mhx_fma(ys[m].w, ys[m].w, q);
            }
        }
        q = mhx_butterfly<L>(q);
        const mhx_real lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        MHX_WAVE_SYNC();                                                 // the row is free for the next candidate
        // ---- accept (src/mh-core.jl:104-114); a zero-mean random walk has no Hastings term
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);
#pragma unroll
        for (int m = 0; m < NQL; ++m) xs[m] = acc ? ys[m] : xs[m];
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && l == 0));
        if (step == save_next) {                                         // wave-uniform
            if (valid) {
                mhx_real* row = a.samples + slot * (long)(D + 1) * ld + c;
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
// dynamic LDS (up to 160 KB per block on gfx950): [candidate rows][target image][proposal image]
extern "C" __global__ void __launch_bounds__(64 * MHX_EMCEE_COOP_WAVES)
mhx_jit_rwmh_dense(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    typedef mhx_emcee_geom<MHX_JIT_DIM, MHX_JIT_L> GEO;
    extern __shared__ mhx_e4 mhx_dense_lds[];
    constexpr int YS4 = MHX_EMCEE_COOP_WAVES * (64 / MHX_JIT_L) * GEO::DP4 / 4;
    mhx_e4* Ash4 = mhx_dense_lds + YS4;
    mhx_e4* Lsh4 = Ash4 + (MHX_JIT_TK == MHX_TARGET_CORR_GAUSS ? GEO::TOTAL4 : 0);
    mhx_rwmh_dense_coop_body<MHX_JIT_DIM, MHX_JIT_L, MHX_JIT_PK, MHX_JIT_TK>(a, tparams, pvec, (mhx_real*)mhx_dense_lds, Ash4, Lsh4);
}
#endif
MHX_NS_END
