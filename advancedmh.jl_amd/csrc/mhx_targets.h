// mhx_targets.h -- log-density catalogue evaluated per lane (one lane = one chain / walker).
//
// A target is a function of an indexable `x` (x[k] -> mhx_real): a register array in the
// compile-time-dimension kernels, a strided view of the [dim][nchains] HBM state in the generic
// ones.  Parameters (`p`) are wave-uniform and read through the scalar cache.  The summation
// order of every target is part of the arithmetic spec: sequential fmaf
// accumulation in ascending index order.
//
// Replaces DensityModel(f) / logdensity(model, x) of the reference (src/AdvancedMH.jl:52-54, :74).
#pragma once
#include "mhx_device_math.h"

#define MHX_TARGET_ISO_GAUSS  0
#define MHX_TARGET_CORR_GAUSS 1
#define MHX_TARGET_IID_NORMAL 2
#define MHX_TARGET_BANANA     3
#define MHX_TARGET_FUNNEL     4
#define MHX_TARGET_USER       100

#if MHX_REAL64
#define MHX_HALF_LOG_2PI 0x1.d67f1c864beb5p-1
#define MHX_ONE_18 0x1.c71c71c71c71cp-5
#define MHX_ONE_9  0x1.c71c71c71c71cp-4
#else
#define MHX_HALF_LOG_2PI 0x1.d67f1cp-1f
#define MHX_ONE_18 0x1.c71c72p-5f
#define MHX_ONE_9  0x1.c71c72p-4f
#endif

MHX_NS_BEGIN

// strided view of one chain's parameters inside a [dim][ld] array
struct mhx_strided_x {
    const mhx_real* base;   // &array[0][chain]
    long ld;
    MHX_DEV mhx_real operator[](int k) const { return base[(long)k * ld]; }
};

// A user log-density supplied as HIP source (mhx_target_from_hip_source) is placed by the JIT
// between mhx_device_math.h and this header; it defines mhx_user_logdensity via MHX_LOGDENSITY
// and the JIT defines MHX_HAVE_USER_TARGET.

// KIND is a compile-time constant in specialised kernels and MHX_TARGET_DYNAMIC (-1) in the
// generic pre-built kernel, where `kind` is a wave-uniform run-time switch.
#define MHX_TARGET_DYNAMIC (-1)

template <int KIND, class X>
MHX_DEV mhx_real mhx_target_eval(int kind, const X& x, const int d, const mhx_real* __restrict__ p,
                              const int np, const mhx_real cst)
{
    const int k_ = (KIND == MHX_TARGET_DYNAMIC) ? kind : KIND;
    switch (k_) {
    case MHX_TARGET_ISO_GAUSS: {
        mhx_real q = MHX_R(0.0);
#pragma unroll
        for (int k = 0; k < d; ++k) { const mhx_real v = x[k]; q = mhx_fma(v, v, q); }
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    case MHX_TARGET_CORR_GAUSS: {
        // w_i = sum_{j<=i} A_ij x_j (ascending j, fma from 0), q = sum_i w_i^2 (ascending i): rows are walked EIGHT at a time
        // so that one read of x_j (a strided HBM / L2 access in the run-time-dimension kernels) feeds eight row chains --
        // every chain and the order of the squares are those of the row-by-row loop, bit for bit
        mhx_real q = MHX_R(0.0);
        constexpr int RB8 = 8;
#pragma unroll
        for (int i0 = 0; i0 < d; i0 += RB8) {
            mhx_real w[RB8];
            int off[RB8];
#pragma unroll
            for (int r = 0; r < RB8; ++r) { w[r] = MHX_R(0.0); off[r] = (i0 + r) * (i0 + r + 1) / 2; }
            if (i0 + RB8 <= d) {
                // a whole block of rows: columns 0 .. i0 feed all eight chains, the triangle behind them row by row
#pragma unroll
                for (int j = 0; j <= i0; ++j) {
                    const mhx_real xj = x[j];
#pragma unroll
                    for (int r = 0; r < RB8; ++r) w[r] = mhx_fma(p[off[r] + j], xj, w[r]);
                }
#pragma unroll
                for (int jj = 1; jj < RB8; ++jj) {
                    const mhx_real xj = x[i0 + jj];
#pragma unroll
                    for (int r = jj; r < RB8; ++r) w[r] = mhx_fma(p[off[r] + i0 + jj], xj, w[r]);
                }
            } else {
                const int jmax = d - 1;
#pragma unroll
                for (int j = 0; j <= jmax; ++j) {
                    const mhx_real xj = x[j];
#pragma unroll
                    for (int r = 0; r < RB8; ++r)
                        if (j <= i0 + r && i0 + r < d) w[r] = mhx_fma(p[off[r] + j], xj, w[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < RB8; ++r)
                if (i0 + r < d) q = mhx_fma(w[r], w[r], q);
        }
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    case MHX_TARGET_IID_NORMAL: {
        const mhx_real mu = x[0], sigma = x[1];
        if (!(sigma > MHX_R(0.0))) return -MHX_INF;         // theta[2] >= 0 support, and logpdf = -Inf at sigma == 0
        mhx_real acc = MHX_R(0.0);
        for (int i = 0; i < np; ++i) {
            const mhx_real z = (p[i] - mu) / sigma;
            acc = mhx_fma(z, z, acc);
        }
        const mhx_real tt = mhx_log(sigma) + MHX_HALF_LOG_2PI;
        return mhx_fma(-MHX_R(0.5), acc, -((mhx_real)np * tt));
    }
    case MHX_TARGET_BANANA: {
        const mhx_real b = p[0];
        const mhx_real x0 = x[0];
        mhx_real q = (x0 * x0) * MHX_R(0.01);
        const mhx_real u = mhx_fma(b, mhx_fma(x0, x0, -MHX_R(100.0)), x[1]);
        q = mhx_fma(u, u, q);
#pragma unroll
        for (int k = 2; k < d; ++k) { const mhx_real v = x[k]; q = mhx_fma(v, v, q); }
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    case MHX_TARGET_FUNNEL: {
        const mhx_real v = x[0];
        mhx_real q = MHX_R(0.0);
#pragma unroll
        for (int k = 1; k < d; ++k) { const mhx_real xk = x[k]; q = mhx_fma(xk, xk, q); }
        const mhx_real ev = mhx_exp(-v);
        mhx_real r = (v * v) * MHX_ONE_18;
        r = mhx_fma(MHX_R(0.5) * (mhx_real)(d - 1), v, r);
        r = mhx_fma(MHX_R(0.5) * ev, q, r);
        return cst - r;
    }
#ifdef MHX_HAVE_USER_TARGET
    case MHX_TARGET_USER:
        return mhx_user_logdensity(x, d, p, np);
#endif
    default:
        return MHX_NAN;
    }
}

// sum of squares of a separable target (ISO_GAUSS, BANANA, FUNNEL) in the L-lane reduction shape, evaluated by ONE lane:
// lane l of the shape owns the Philox blocks b = l, l+L, ... (4 dimensions each); the L partial sums meet in an
// xor-butterfly with offsets 1, 2, 4, ...
template <class X>
MHX_DEV mhx_real mhx_separable_q_lanes(const int k_, const X& x, const int d, const mhx_real* __restrict__ p, const int L)
{
    mhx_real part[64];
    const int nblk = (d + 3) >> 2;
    for (int l = 0; l < L; ++l) {
        mhx_real q = MHX_R(0.0);
        for (int b = l; b < nblk; b += L)
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k >= d) break;
                const mhx_real v = x[k];
                if (k_ == MHX_TARGET_BANANA && k == 0) q = (v * v) * MHX_R(0.01);
                else if (k_ == MHX_TARGET_BANANA && k == 1) {
                    const mhx_real x0 = x[0];
                    const mhx_real u = mhx_fma(p[0], mhx_fma(x0, x0, -MHX_R(100.0)), v);
                    q = mhx_fma(u, u, q);
                } else if (k_ == MHX_TARGET_FUNNEL && k == 0) {
                } else q = mhx_fma(v, v, q);
            }
        part[l] = q;
    }
    for (int off = 1; off < L; off <<= 1) {
        mhx_real nxt[64];
        for (int l = 0; l < L; ++l) nxt[l] = part[l] + part[l ^ off];
        for (int l = 0; l < L; ++l) part[l] = nxt[l];
    }
    return part[0];
}

// The separable targets evaluated with the L-lane reduction shape of the cooperative kernels, by ONE
// lane (initial state, setparams): lane l of the shape owns the Philox blocks b = l, l+L, ...; the L
// partial sums meet in an xor-butterfly with offsets 1, 2, 4, ...  Same arithmetic as
// mhx_rwmh_coop_body, serialised.
template <int KIND, class X>
MHX_DEV mhx_real mhx_target_eval_lanes(int kind, const X& x, const int d, const mhx_real* __restrict__ p,
                                    const int np, const mhx_real cst, const int L)
{
    const int k_ = (KIND == MHX_TARGET_DYNAMIC) ? kind : KIND;
    const bool separable = k_ == MHX_TARGET_ISO_GAUSS || k_ == MHX_TARGET_BANANA || k_ == MHX_TARGET_FUNNEL;
    if (L <= 1 || !(separable || k_ == MHX_TARGET_CORR_GAUSS || k_ == MHX_TARGET_IID_NORMAL)) return mhx_target_eval<KIND>(kind, x, d, p, np, cst);
    mhx_real part[64];
    if (k_ == MHX_TARGET_IID_NORMAL) {
        // shape of the wave-per-chain kernel (mhx_rwmh_wave_body): lane l owns the data terms i = l, l+L, ...
        const mhx_real mu = x[0], sigma = x[1];
        if (!(sigma > MHX_R(0.0))) return -MHX_INF;
        for (int l = 0; l < L; ++l) {
            mhx_real q = MHX_R(0.0);
            for (int i = l; i < np; i += L) { const mhx_real z = (p[i] - mu) / sigma; q = mhx_fma(z, z, q); }
            part[l] = q;
        }
        for (int off = 1; off < L; off <<= 1) {
            mhx_real nxt[64];
            for (int l = 0; l < L; ++l) nxt[l] = part[l] + part[l ^ off];
            for (int l = 0; l < L; ++l) part[l] = nxt[l];
        }
        const mhx_real tt = mhx_log(sigma) + MHX_HALF_LOG_2PI;
        return mhx_fma(-MHX_R(0.5), part[0], -((mhx_real)np * tt));
    }
    if (k_ == MHX_TARGET_CORR_GAUSS) {
        // shape of the cooperative ensemble kernel: lane l owns rows i = l, l+L, ... of A x
        for (int l = 0; l < L; ++l) {
            mhx_real q = MHX_R(0.0);
            for (int i = l; i < d; i += L) {
                const mhx_real* Ar = p + (long)i * (i + 1) / 2;
                mhx_real w = MHX_R(0.0);
                for (int j = 0; j <= i; ++j) w = mhx_fma(Ar[j], x[j], w);
                q = mhx_fma(w, w, q);
            }
            part[l] = q;
        }
        for (int off = 1; off < L; off <<= 1) {
            mhx_real nxt[64];
            for (int l = 0; l < L; ++l) nxt[l] = part[l] + part[l ^ off];
            for (int l = 0; l < L; ++l) part[l] = nxt[l];
        }
        return mhx_fma(-MHX_R(0.5), part[0], cst);
    }
    part[0] = mhx_separable_q_lanes(k_, x, d, p, L);
    const mhx_real q = part[0];
    if (k_ != MHX_TARGET_FUNNEL) return mhx_fma(-MHX_R(0.5), q, cst);
    const mhx_real v = x[0];
    const mhx_real ev = mhx_exp(-v);
    mhx_real r = (v * v) * MHX_ONE_18;
    r = mhx_fma(MHX_R(0.5) * (mhx_real)(d - 1), v, r);
    r = mhx_fma(MHX_R(0.5) * ev, q, r);
    return cst - r;
}
MHX_NS_END
