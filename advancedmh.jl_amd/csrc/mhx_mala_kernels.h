// mhx_mala_kernels.h -- Metropolis-adjusted Langevin, one wavefront lane per chain.
//
// Replaces MALA's step (src/MALA.jl:54-93) with the standard Langevin proposal
// g -> MvNormal((sigma2/2) g, sigma2 I) used by the reference's tests (test/runtests.jl:291,352):
//   y = x + (sigma2/2) grad(x) + sigma z
//   logratio = q(prop(grad y), x, y) - q(prop(grad x), y, x) = 1/2 |z|^2 - 1/2 |z + (sigma/2)(grad x + grad y)|^2
//   accept iff -randexp < lp(y) - lp(x) + logratio
// The gradient the reference gets from ForwardDiff / LogDensityProblems (ext/AdvancedMHForwardDiffExt.jl,
// src/MALA.jl:100-105) is analytic here for the catalogue targets, and supplied as HIP source
// (MHX_LOGDENSITY_AND_GRADIENT) for user models.  Run-time dimension: state x and grad(x) live in HBM as
// [dim][nchains] (chain fastest, coalesced); candidate, its gradient and the noise go through scratch slabs.
// Compile-time dimension (hiprtc, d <= 48): all five vectors are registers (mhx_mala_reg_body).
#pragma once
#include "mhx_targets.h"
#include "mhx_rwmh_kernels.h"      // mhx_reg_zig_fill: the register-array ziggurat of MHX_FLAG_ZIGGURAT

MHX_NS_BEGIN

// read/write view of one chain's column inside a [dim][ld] slab
struct mhx_strided_rw {
    mhx_real* base;
    long ld;
    MHX_DEV mhx_real operator[](int k) const { return base[(long)k * ld]; }
    MHX_DEV void set(int k, mhx_real v) const { base[(long)k * ld] = v; }
};

// value and gradient of the catalogue targets; same accumulation order as mhx_target_eval
template <int KIND, class X, class GO>
MHX_DEV mhx_real mhx_target_grad(int kind, const X& x, const GO& g, const int d, const mhx_real* __restrict__ p,
                              const int np, const mhx_real cst)
{
    const int k_ = (KIND == MHX_TARGET_DYNAMIC) ? kind : KIND;
    switch (k_) {
    case MHX_TARGET_ISO_GAUSS: {
        mhx_real q = MHX_R(0.0);
        for (int k = 0; k < d; ++k) { const mhx_real v = x[k]; q = mhx_fma(v, v, q); g.set(k, -v); }
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    case MHX_TARGET_CORR_GAUSS: {                      // grad = -A^T (A x)
        mhx_real q = MHX_R(0.0);
        int off = 0;
        for (int i = 0; i < d; ++i) {                  // w = A x, parked in g
            mhx_real w = MHX_R(0.0);
            for (int j = 0; j <= i; ++j) w = mhx_fma(p[off + j], x[j], w);
            g.set(i, w);
            q = mhx_fma(w, w, q);
            off += i + 1;
        }
        for (int j = 0; j < d; ++j) {                  // g_j = -sum_{i>=j} A_ij w_i (ascending i), in place
            mhx_real acc = MHX_R(0.0);
            for (int i = j; i < d; ++i) acc = mhx_fma(p[(long)i * (i + 1) / 2 + j], g[i], acc);
            g.set(j, -acc);
        }
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    case MHX_TARGET_IID_NORMAL: {
        const mhx_real mu = x[0], sigma = x[1];
        if (!(sigma > MHX_R(0.0))) { g.set(0, MHX_R(0.0)); g.set(1, MHX_R(0.0)); return -MHX_INF; }
        const mhx_real inv = MHX_R(1.0) / sigma;
        mhx_real acc = MHX_R(0.0), s1 = MHX_R(0.0);
        for (int i = 0; i < np; ++i) {
            const mhx_real z = (p[i] - mu) / sigma;
            acc = mhx_fma(z, z, acc);
            s1 = s1 + z;
        }
        const mhx_real nf = (mhx_real)np;
        g.set(0, s1 * inv);
        g.set(1, (acc - nf) * inv);
        const mhx_real tt = mhx_log(sigma) + MHX_HALF_LOG_2PI;
        return mhx_fma(-MHX_R(0.5), acc, -(nf * tt));
    }
    case MHX_TARGET_BANANA: {
        const mhx_real b = p[0];
        const mhx_real x0 = x[0];
        mhx_real q = (x0 * x0) * MHX_R(0.01);
        const mhx_real u = mhx_fma(b, mhx_fma(x0, x0, -MHX_R(100.0)), x[1]);
        q = mhx_fma(u, u, q);
        g.set(0, -(mhx_fma(x0, MHX_R(0.01), (MHX_R(2.0) * b) * (u * x0))));
        g.set(1, -u);
        for (int k = 2; k < d; ++k) { const mhx_real v = x[k]; q = mhx_fma(v, v, q); g.set(k, -v); }
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    case MHX_TARGET_FUNNEL: {
        const mhx_real v = x[0];
        mhx_real q = MHX_R(0.0);
        for (int k = 1; k < d; ++k) { const mhx_real xk = x[k]; q = mhx_fma(xk, xk, q); }
        const mhx_real ev = mhx_exp(-v);
        mhx_real r = (v * v) * MHX_ONE_18;
        r = mhx_fma(MHX_R(0.5) * (mhx_real)(d - 1), v, r);
        r = mhx_fma(MHX_R(0.5) * ev, q, r);
        g.set(0, mhx_fma(MHX_R(0.5) * ev, q, -(mhx_fma(v, MHX_ONE_9, MHX_R(0.5) * (mhx_real)(d - 1)))));
        for (int k = 1; k < d; ++k) g.set(k, -(ev * x[k]));
        return cst - r;
    }
#ifdef MHX_HAVE_USER_TARGET
    case MHX_TARGET_USER:
        return mhx_user_logdensity_and_gradient(x, g, d, p, np);
#endif
    default:
        return MHX_NAN;
    }
}

// value and gradient with the L-lane reduction shape of the cooperative kernel, by ONE lane (initial state, setparams):
// the separable targets take their sum of squares from mhx_separable_q_lanes; everything else is mhx_target_grad
template <int KIND, class X, class GO>
MHX_DEV mhx_real mhx_target_grad_lanes(int kind, const X& x, const GO& g, const int d, const mhx_real* __restrict__ p,
                                       const int np, const mhx_real cst, const int L)
{
    const int k_ = (KIND == MHX_TARGET_DYNAMIC) ? kind : KIND;
    const bool separable = k_ == MHX_TARGET_ISO_GAUSS || k_ == MHX_TARGET_BANANA || k_ == MHX_TARGET_FUNNEL;
    if (L > 1 && k_ == MHX_TARGET_CORR_GAUSS) {
        // the matrix-core kernel's shape: w = A x by rows (parked in g), the squares of rows l, l+L, ... on lane l,
        // butterfly; the gradient -A^T w as in mhx_target_grad
        mhx_real part[64];
        for (int l = 0; l < L; ++l) part[l] = MHX_R(0.0);
        for (int i = 0; i < d; ++i) {
            const mhx_real* Ar = p + (long)i * (i + 1) / 2;
            mhx_real w = MHX_R(0.0);
            for (int j = 0; j <= i; ++j) w = mhx_fma(Ar[j], x[j], w);
            g.set(i, w);
            part[i % L] = mhx_fma(w, w, part[i % L]);
        }
        for (int off = 1; off < L; off <<= 1) {
            mhx_real nxt[64];
            for (int l = 0; l < L; ++l) nxt[l] = part[l] + part[l ^ off];
            for (int l = 0; l < L; ++l) part[l] = nxt[l];
        }
        for (int j = 0; j < d; ++j) {
            mhx_real acc = MHX_R(0.0);
            for (int i = j; i < d; ++i) acc = mhx_fma(p[(long)i * (i + 1) / 2 + j], g[i], acc);
            g.set(j, -acc);
        }
        return mhx_fma(-MHX_R(0.5), part[0], cst);
    }
    if (L <= 1 || !separable) return mhx_target_grad<KIND>(kind, x, g, d, p, np, cst);
    const mhx_real q = mhx_separable_q_lanes(k_, x, d, p, L);
    if (k_ == MHX_TARGET_ISO_GAUSS) {
        for (int k = 0; k < d; ++k) g.set(k, -x[k]);
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    if (k_ == MHX_TARGET_BANANA) {
        const mhx_real b = p[0], x0 = x[0];
        const mhx_real u = mhx_fma(b, mhx_fma(x0, x0, -MHX_R(100.0)), x[1]);
        g.set(0, -(mhx_fma(x0, MHX_R(0.01), (MHX_R(2.0) * b) * (u * x0))));
        g.set(1, -u);
        for (int k = 2; k < d; ++k) g.set(k, -x[k]);
        return mhx_fma(-MHX_R(0.5), q, cst);
    }
    const mhx_real v = x[0];
    const mhx_real ev = mhx_exp(-v);
    mhx_real r = (v * v) * MHX_ONE_18;
    r = mhx_fma(MHX_R(0.5) * (mhx_real)(d - 1), v, r);
    r = mhx_fma(MHX_R(0.5) * ev, q, r);
    g.set(0, mhx_fma(MHX_R(0.5) * ev, q, -(mhx_fma(v, MHX_ONE_9, MHX_R(0.5) * (mhx_real)(d - 1)))));
    for (int k = 1; k < d; ++k) g.set(k, -(ev * x[k]));
    return cst - r;
}

struct mhx_mala_args {
    mhx_real* x;                 // [dim][ld]
    mhx_real* gx;                // [dim][ld] gradient at x (GradientTransition.gradient, src/MALA.jl:14-19)
    mhx_real* lp;
    mhx_u32* acc_count;
    mhx_u64* acc_total;
    mhx_real* samples;
    unsigned char* accepted;
    unsigned char* last_acc;
    mhx_real* ybuf;              // [dim][ld] candidate
    mhx_real* gybuf;             // [dim][ld] gradient at the candidate
    mhx_real* zbuf;              // [dim][ld] proposal noise
    mhx_u64 seed;
    mhx_u64 first_chain;
    int nchains;
    int ld;
    int dim;
    int target_kind;
    int ntparams;
    mhx_real tconst;
    mhx_real sigma;              // sqrt(sigma2)
    mhx_real h;                  // sigma2 / 2
    mhx_real hs;                 // sigma / 2
    mhx_u32 step0;
    int nsteps;
    mhx_u32 save_next;
    int save_slot;
    int thinning;
    int reduce_lanes;         // lanes per chain of the cooperative kernel (reduction shape of the sums), >= 1
};

// ---------------------------------------------------------------------------------------------
// Cooperative MALA for the separable catalogue targets (iso-Gaussian, banana, funnel): L lanes share one chain exactly as
// in mhx_rwmh_coop_body -- lane l owns the Philox blocks b = l, l+L, ... -- with state x, grad(x) and per step candidate,
// its gradient and the noise in VGPRs (5 NBL float4 per lane).  The three sums of a step (the target's sum of squares,
// |z|^2, |z + (sigma/2)(grad x + grad y)|^2) are lane partial sums in block order + the xor-butterfly: reduction shape L,
// which the oracle takes too.  The gradients of these targets are element-wise once the few shared scalars are known
// (banana: x1, u in lane 0's first block; funnel: v = x1 and the total sum of squares).
template <int L, int NBL, int TK>
MHX_DEV void mhx_mala_coop_body(const mhx_mala_args& a, const mhx_real* __restrict__ tparams)
{
    constexpr int CPW = 64 / L;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int cw = lane & (CPW - 1);
    const int l = lane / CPW;
    const long c_raw = wave * CPW + cw;
    const bool valid = c_raw < a.nchains;
    const long c = valid ? c_raw : (long)a.nchains - 1;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;
    const int d = a.dim;
    const int k_last = 4 * (l + L * (NBL - 1));

    mhx_real x[NBL][4], gx[NBL][4];
#pragma unroll
    for (int i = 0; i < NBL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * (l + L * i) + j;
            const bool in = i < NBL - 1 || k < d;
            x[i][j] = in ? a.x[(long)k * ld + c] : MHX_R(0.0);
            gx[i][j] = in ? a.gx[(long)k * ld + c] : MHX_R(0.0);
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
        mhx_real y[NBL][4], gy[NBL][4], z[NBL][4];
        mhx_real q = MHX_R(0.0), fwd = MHX_R(0.0);
        // ---- propose (src/MALA.jl:70): y = x + (sigma2/2) grad(x) + sigma z; the target's sum of squares on the way
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            mhx_real n[4];
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)(l + L * i), n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = i < NBL - 1 || k_last + j < d;
                const mhx_real zk = in ? n[j] : MHX_R(0.0);
                const mhx_real yk = in ? mhx_fma(a.sigma, zk, mhx_fma(a.h, gx[i][j], x[i][j])) : MHX_R(0.0);
                z[i][j] = zk;
                y[i][j] = yk;
                fwd = mhx_fma(zk, zk, fwd);
                const mhx_real sq = mhx_fma(yk, yk, q);
                if (TK == MHX_TARGET_BANANA && i == 0 && j == 0) {
                    q = l == 0 ? (yk * yk) * MHX_R(0.01) : sq;
                } else if (TK == MHX_TARGET_BANANA && i == 0 && j == 1) {
                    const mhx_real y0 = y[0][0];
                    const mhx_real u = mhx_fma(tparams[0], mhx_fma(y0, y0, -MHX_R(100.0)), yk);
                    q = l == 0 ? mhx_fma(u, u, q) : sq;
                } else if (TK == MHX_TARGET_FUNNEL && i == 0 && j == 0) {
                    q = l == 0 ? q : sq;
                } else {
                    q = sq;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        q = mhx_butterfly<L>(q);
        fwd = mhx_butterfly<L>(fwd);
        // ---- value and gradient at the candidate (:73-75): same expressions as mhx_target_grad
        mhx_real lpy;
        if (TK == MHX_TARGET_FUNNEL) {
            const mhx_real v = __shfl(y[0][0], cw, 64);               // x1 lives in lane l == 0 of the chain
            const mhx_real ev = mhx_exp(-v);
            mhx_real r = (v * v) * MHX_ONE_18;
            r = mhx_fma(MHX_R(0.5) * (mhx_real)(d - 1), v, r);
            r = mhx_fma(MHX_R(0.5) * ev, q, r);
            lpy = a.tconst - r;
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) gy[i][j] = -(ev * y[i][j]);
            if (l == 0) gy[0][0] = mhx_fma(MHX_R(0.5) * ev, q, -(mhx_fma(v, MHX_ONE_9, MHX_R(0.5) * (mhx_real)(d - 1))));
        } else {
            lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) gy[i][j] = -y[i][j];
            if (TK == MHX_TARGET_BANANA && l == 0) {
                const mhx_real b = tparams[0], y0 = y[0][0];
                const mhx_real u = mhx_fma(b, mhx_fma(y0, y0, -MHX_R(100.0)), y[0][1]);
                gy[0][0] = -(mhx_fma(y0, MHX_R(0.01), (MHX_R(2.0) * b) * (u * y0)));
                gy[0][1] = -u;
            }
        }
        // ---- log ratio of the proposal densities (:78-80)
        mhx_real bwd = MHX_R(0.0);
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = i < NBL - 1 || k_last + j < d;
                const mhx_real tk = in ? mhx_fma(a.hs, gx[i][j] + gy[i][j], z[i][j]) : MHX_R(0.0);
                bwd = mhx_fma(tk, tk, bwd);
            }
        bwd = mhx_butterfly<L>(bwd);
        const mhx_real loga = (lpy - lp) + MHX_R(0.5) * (fwd - bwd);                 // :83
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;                                               // :86 (strict)
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[i][j] = acc ? y[i][j] : x[i][j];
                gx[i][j] = acc ? gy[i][j] : gx[i][j];
            }
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && l == 0));
        if (step == save_next) {
            if (valid) {
                mhx_real* row = a.samples + slot * (long)(d + 1) * ld + c;
                // (buffer descriptor + running scalar row offset behind an opaque asm: MHX_COOP_REC_RUN in mhx_rwmh_kernels.h)
                const mhx_srd srd = mhx_make_srd(a.samples + slot * (long)(d + 1) * ld, (mhx_u32)(d + 1) * (mhx_u32)ld * MHX_RB);
                const mhx_u32 ldb = (mhx_u32)ld * MHX_RB;
                const mhx_u32 loff = ((mhx_u32)(4 * l) * (mhx_u32)ld + (mhx_u32)c) * MHX_RB;
                mhx_u32 roff = 0u;
                asm volatile("" : "+s"(roff));
#pragma unroll
                for (int i = 0; i < NBL; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * (l + L * i) + j;
                        if (i < NBL - 1 || k < d) mhx_srd_store<MHX_REC_STORE_AUX>(srd, loff, roff, x[i][j]);
                        roff += (j < 3 ? 1u : (mhx_u32)(4 * L - 3)) * ldb;
                    }
                if (l == 0) {
                    row[(long)d * ld] = lp;
                    a.accepted[slot * ld + c] = acc ? 1 : 0;
                }
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    if (valid) {
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * (l + L * i) + j;
                if (i < NBL - 1 || k < d) { a.x[(long)k * ld + c] = x[i][j]; a.gx[(long)k * ld + c] = gx[i][j]; }
            }
        if (l == 0) {
            a.lp[c] = lp;
            a.acc_count[c] = nacc;
            a.last_acc[c] = last ? 1 : 0;
        }
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

template <int TK>
MHX_DEV void mhx_mala_body(const mhx_mala_args& a, const mhx_real* __restrict__ tparams)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;
    const int d = a.dim;
    mhx_real* xs = a.x + c;
    mhx_real* gs = a.gx + c;
    mhx_real* ys = a.ybuf + c;
    mhx_real* zs = a.zbuf + c;
    mhx_strided_rw gy;
    gy.base = a.gybuf + c;
    gy.ld = ld;
    mhx_strided_x yv;
    yv.base = ys;
    yv.ld = ld;

    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    const int nblk = (d + 3) >> 2;

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        // ---- propose (src/MALA.jl:70): y = x + (sigma2/2) grad(x) + sigma z
        mhx_real fwd = MHX_R(0.0);
        for (int b = 0; b < nblk; ++b) {
            mhx_real n[4];
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k < d) {
                    const mhx_real z = n[j];
                    ys[(long)k * ld] = mhx_fma(a.sigma, z, mhx_fma(a.h, gs[(long)k * ld], xs[(long)k * ld]));
                    zs[(long)k * ld] = z;
                    fwd = mhx_fma(z, z, fwd);
                }
            }
        }
        // ---- value and gradient at the candidate (:73-75)
        const mhx_real lpy = mhx_target_grad<TK>(a.target_kind, yv, gy, d, tparams, a.ntparams, a.tconst);
        // ---- log ratio of the proposal densities (:78-80)
        mhx_real bwd = MHX_R(0.0);
        for (int k = 0; k < d; ++k) {
            const mhx_real tk = mhx_fma(a.hs, gs[(long)k * ld] + gy[k], zs[(long)k * ld]);
            bwd = mhx_fma(tk, tk, bwd);
        }
        const mhx_real loga = (lpy - lp) + MHX_R(0.5) * (fwd - bwd);              // :83
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;                                    // :86 (strict)
        if (acc) {
            for (int k = 0; k < d; ++k) { xs[(long)k * ld] = ys[(long)k * ld]; gs[(long)k * ld] = gy[k]; }
            lp = lpy;
        }
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc));
        if (step == save_next) {
            mhx_real* row = a.samples + slot * (long)(d + 1) * ld + c;
            for (int k = 0; k < d; ++k) row[(long)k * ld] = xs[(long)k * ld];
            row[(long)d * ld] = lp;
            a.accepted[slot * ld + c] = acc ? 1 : 0;
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    a.lp[c] = lp;
    a.acc_count[c] = nacc;
    a.last_acc[c] = last ? 1 : 0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// ---------------------------------------------------------------------------------------------
// Compile-time dimension D (hiprtc): x, grad(x), the candidate, its gradient and the noise all live in
// registers (5 D floats per lane) for the steps of a launch; HBM sees the state once per launch and the
// recorded samples.  Same arithmetic as mhx_mala_body, statement for statement.
template <int D>
struct mhx_reg_rw {
    mhx_real* v;                                        // a lane-private array that full unrolling turns into registers
    MHX_DEV mhx_real operator[](int k) const { return v[k]; }
    MHX_DEV void set(int k, mhx_real x) const { v[k] = x; }
};

// XR < D (round 4): only the candidate and its gradient have to be whole in a lane's registers while the user's function runs; the
// state, its gradient and the step's noise are touched once per step each -- their first XR coordinates stay in registers, the tails
// live in the block's LDS as [3][D - XR][lane] (one wave per block).  Carries the kernel from d = 24 / 48 (fp64 / fp32) to 64 / 128;
// the run-time-dimension kernel it replaces there is 10-20 x slower (tools/bench_mala_user.py).  Same arithmetic, same chains.
// ZIG (round 5; fp32 since round 6; MHX_FLAG_ZIGGURAT on a MALA run): the step's noise by the table ziggurat instead of Box-Muller -- the same
// register-array fill as the RWMH register kernel's (mhx_reg_zig_fill: fast path into the candidate's registers, the wave-step's
// failures queued, refined side by side, handed back), so every lane of the one-wave block stays alive (idle lanes shadow the last
// chain, stores guarded) and `zlds` holds [layer table][queue][results] in front of the tails.  The oracle's orc_mala(normal_gen = 1).
template <int D, int TK, int XR = D, bool ZIG = false>
MHX_DEV void mhx_mala_reg_body(const mhx_mala_args& a, const mhx_real* __restrict__ tparams, mhx_real* tails = nullptr,
                               double* zlds = nullptr)
{
    const int c_raw = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int NT = D - XR;                                       // coordinates per vector in LDS
    mhx_real* xl = tails + threadIdx.x;                              // [NT][64], then g, then z
    mhx_real* gl = xl + NT * 64;
    mhx_real* zl = gl + NT * 64;
    const bool valid = c_raw < a.nchains;
    if (!ZIG && !valid) return;
    const int c = valid ? c_raw : a.nchains - 1;                     // (ZIG: the queue and the hand-back are wave-wide)
    [[maybe_unused]] mhx_real* const zt = (mhx_real*)zlds;
    [[maybe_unused]] unsigned short* zq = (unsigned short*)((char*)zlds + MHX_ZIG_TABLE_BYTES);
    [[maybe_unused]] mhx_real* zres = (mhx_real*)((char*)zlds + MHX_ZIG_TABLE_BYTES + 128);
    [[maybe_unused]] mhx_u32 zsign = 0x80000000u;
    if constexpr (ZIG) {
#if MHX_REAL64
        for (int e = (int)threadIdx.x; e <= MHX_ZIG_N; e += 64) zt[e] = mhx_zig_x[e];
#else
        if ((mhx_u32)(mhx_u64)zlds != 0u) __builtin_trap();       // (MHX_ZIG_PAIR_OF addresses the table at LDS byte 0)
        for (int e = (int)threadIdx.x; e < MHX_ZIG_PAIR_FLOATS; e += 64) zt[e] = mhx_zig_pair_entry(e);
#endif
        __syncthreads();
        asm volatile("" : "+s"(zsign));
    }
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const long ld = a.ld;
    constexpr int XA = XR > 0 ? XR : 1;
    mhx_real x[XA], g[XA], y[D], gyv[D], z[XA];
    auto getx = [&](const int k) -> mhx_real { return k < XR ? x[k < XR ? k : 0] : xl[(k - XR) * 64]; };
    auto getg = [&](const int k) -> mhx_real { return k < XR ? g[k < XR ? k : 0] : gl[(k - XR) * 64]; };
    auto getz = [&](const int k) -> mhx_real { return k < XR ? z[k < XR ? k : 0] : zl[(k - XR) * 64]; };
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const mhx_real xv = a.x[(long)k * ld + c], gv = a.gx[(long)k * ld + c];
        if (k < XR) { x[k < XR ? k : 0] = xv; g[k < XR ? k : 0] = gv; } else { xl[(k - XR) * 64] = xv; gl[(k - XR) * 64] = gv; }
    }
    mhx_reg_rw<D> gy;
    gy.v = gyv;
    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    constexpr int nblk = (D + 3) >> 2;

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        mhx_real fwd = MHX_R(0.0);
        if constexpr (ZIG) {
            mhx_reg_zig_fill<D>(y, ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, zt, zq, zres, (int)threadIdx.x, (long)blockIdx.x * 64,
                                a.first_chain, a.nchains, zsign);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const mhx_real nk = y[k];
                if (k < XR) z[k < XR ? k : 0] = nk; else zl[(k - XR) * 64] = nk;
                y[k] = mhx_fma(a.sigma, nk, mhx_fma(a.h, getg(k), getx(k)));            // src/MALA.jl:70
                fwd = mhx_fma(nk, nk, fwd);
            }
        } else
        {
#pragma unroll
        for (int b = 0; b < nblk; ++b) {
            mhx_real n[4];
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k < D) {
                    if (k < XR) z[k < XR ? k : 0] = n[j]; else zl[(k - XR) * 64] = n[j];
                    y[k] = mhx_fma(a.sigma, n[j], mhx_fma(a.h, getg(k), getx(k)));      // src/MALA.jl:70
                    fwd = mhx_fma(n[j], n[j], fwd);
                }
            }
        }
        }
        const mhx_real lpy = mhx_target_grad<TK>(TK, y, gy, D, tparams, a.ntparams, a.tconst);   // :73-75
        mhx_real bwd = MHX_R(0.0);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const mhx_real tk = mhx_fma(a.hs, getg(k) + gyv[k], getz(k));
            bwd = mhx_fma(tk, tk, bwd);
        }
        const mhx_real loga = (lpy - lp) + MHX_R(0.5) * (fwd - bwd);              // :78-83
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;                                    // :86 (strict)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < XR) { x[k < XR ? k : 0] = acc ? y[k] : x[k < XR ? k : 0]; g[k < XR ? k : 0] = acc ? gyv[k] : g[k < XR ? k : 0]; }
            else if (acc) { xl[(k - XR) * 64] = y[k]; gl[(k - XR) * 64] = gyv[k]; }
        }
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid));
        if (step == save_next) {
            mhx_real* row = a.samples + slot * (long)(D + 1) * ld + c;
            if (valid) {   // (buffer descriptor + running scalar row offset behind an opaque asm: MHX_COOP_REC_RUN in mhx_rwmh_kernels.h)
                const mhx_srd srd = mhx_make_srd(a.samples + slot * (long)(D + 1) * ld, (mhx_u32)(D + 1) * (mhx_u32)ld * MHX_RB);
                const mhx_u32 ldb = (mhx_u32)ld * MHX_RB, loff = (mhx_u32)c * MHX_RB;
                mhx_u32 roff = 0u;
                asm volatile("" : "+s"(roff));
#pragma unroll
                for (int k = 0; k < D; ++k) { mhx_srd_store<MHX_REC_STORE_AUX>(srd, loff, roff, getx(k)); roff += ldb; }
                row[(long)D * ld] = lp;
                a.accepted[slot * ld + c] = acc ? 1 : 0;
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < D; ++k) { a.x[(long)k * ld + c] = getx(k); a.gx[(long)k * ld + c] = getg(k); }
        a.lp[c] = lp;
        a.acc_count[c] = nacc;
        a.last_acc[c] = last ? 1 : 0;
    }
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

#define MHX_MALA_ZIG_LDS_BYTES(D, XR) ((size_t)3 * ((D) - (XR)) * 64 * sizeof(mhx_real) + MHX_REG_ZIG_HDR_BYTES)

// initial GradientTransition (src/MALA.jl:38-40): lp and gradient at the given initial_params
template <int TK>
MHX_DEV void mhx_mala_init_body(const mhx_mala_args& a, const mhx_real* __restrict__ tparams, const int reset_counts)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    mhx_strided_x xv;
    xv.base = a.x + c;
    xv.ld = a.ld;
    mhx_strided_rw g;
    g.base = a.gx + c;
    g.ld = a.ld;
    a.lp[c] = mhx_target_grad_lanes<TK>(a.target_kind, xv, g, a.dim, tparams, a.ntparams, a.tconst, a.reduce_lanes);
    if (reset_counts) { a.acc_count[c] = 0u; a.last_acc[c] = 0; }
}

#ifdef MHX_JIT_MALA_COOP
extern "C" __global__ void __launch_bounds__(256, MHX_JIT_WAVES)
mhx_jit_mala_coop(const mhx_mala_args a, const mhx_real* __restrict__ tparams)
{
    mhx_mala_coop_body<MHX_JIT_L, MHX_JIT_NBL, MHX_JIT_TK>(a, tparams);
}
#endif
#ifdef MHX_JIT_MALA
#if defined(MHX_JIT_XR) && MHX_JIT_DIM > 0 && MHX_JIT_XR < MHX_JIT_DIM
// the register kernel with the tails of x, grad(x) and the noise in LDS: one wave per block
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_mala_split(const mhx_mala_args a, const mhx_real* __restrict__ tparams)
{
    extern __shared__ mhx_real mhx_mala_tails[];               // [3][MHX_JIT_DIM - MHX_JIT_XR][64]
    mhx_mala_reg_body<MHX_JIT_DIM, MHX_JIT_TK, MHX_JIT_XR>(a, tparams, mhx_mala_tails);
}
#endif
#if defined(MHX_JIT_GEN) && MHX_JIT_GEN == 1 && MHX_JIT_DIM > 0
// MHX_FLAG_ZIGGURAT: the register kernel with the ziggurat noise, one wave per block; LDS = [table][queue][results][tails]
#ifndef MHX_JIT_XR
#define MHX_JIT_XR MHX_JIT_DIM
#endif
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_mala_zig(const mhx_mala_args a, const mhx_real* __restrict__ tparams)
{
    extern __shared__ double mhx_mala_zig_lds[];               // MHX_MALA_ZIG_LDS_BYTES(MHX_JIT_DIM, MHX_JIT_XR)
    mhx_mala_reg_body<MHX_JIT_DIM, MHX_JIT_TK, MHX_JIT_XR, true>(a, tparams, (mhx_real*)((char*)mhx_mala_zig_lds + MHX_REG_ZIG_HDR_BYTES), mhx_mala_zig_lds);
}
#endif
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_mala(const mhx_mala_args a, const mhx_real* __restrict__ tparams)
{
#if MHX_JIT_DIM > 0 && !(defined(MHX_JIT_XR) && MHX_JIT_XR < MHX_JIT_DIM)
    mhx_mala_reg_body<MHX_JIT_DIM, MHX_JIT_TK>(a, tparams);
#else
    mhx_mala_body<MHX_JIT_TK>(a, tparams);
#endif
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_mala_init(const mhx_mala_args a, const mhx_real* __restrict__ tparams, const int reset_counts)
{
    mhx_mala_init_body<MHX_JIT_TK>(a, tparams, reset_counts);
}
#endif
MHX_NS_END
