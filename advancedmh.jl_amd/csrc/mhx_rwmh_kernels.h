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
    int reduce_lanes;         // reduction shape of the separable targets (lanes per chain), >= 1
    // running moments instead of a sample tensor (runs too large to store): per chain and parameter the
    // Welford mean / M2 over the states the schedule would have recorded
    float* mom_mean;          // [dim+1][ld] or null
    float* mom_m2;            // [dim+1][ld]
    mhx_u32 mom_n0;           // states already folded in before this launch
    // drifting random walk (non-zero proposal mean; generic kernel only): mu[dim] followed by 2 L^-1 mu [dim]
    const float* pmean;       // null = zero mean (the Hastings ratio is then exactly 0 and is not computed)
    // static (independence) proposal, generic kernel only: q(x) = -1/2 |L^-1 (x - mu)|^2 of each chain's state
    float* qx;                // [ld] or null = random walk
};

// one Welford step with the wave-uniform 1/n
MHX_DEV void mhx_welford(float x, float rn, float& mean, float& m2)
{
    const float delta = x - mean;
    mean = mhx_fma(delta, rn, mean);
    m2 = mhx_fma(delta, x - mean, m2);
}

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
    const mhx_u32 cu = (mhx_u32)c * 4u;  // row pointers are wave-uniform (scalar), the lane adds c * 4 bytes
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = mhx_ld_off(a.x + (long)k * ld, cu);
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
            float* slotp = a.samples + slot * (long)(D + 1) * ld;
            const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(D + 1) * (mhx_u32)ld * 4u);
            const mhx_u32 ldb = (mhx_u32)ld * 4u;
#pragma unroll
            for (int k = 0; k < D; ++k) mhx_srd_store(srd, cu, (mhx_u32)k * ldb, x[k]);
            mhx_srd_store(srd, cu, (mhx_u32)D * ldb, lp);
            a.accepted[slot * ld + c] = acc ? 1 : 0;
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) mhx_st_off(a.x + (long)k * ld, cu, x[k]);
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
    mhx_u32 mom_n = a.mom_n0;
    const int nblk = (d + 3) >> 2;
    float qxc = a.qx ? a.qx[c] : 0.0f;

    for (int i = 0; i < a.nsteps; ++i) {
        const mhx_u32 step = a.step0 + (mhx_u32)i;
        // a drifting walk (proposal mean mu != 0) keeps |z|^2 and |z + 2 L^-1 mu|^2 for its Hastings ratio
        const float* mu = a.pmean;
        const bool stat = a.qx != nullptr;             // StaticProposal: the candidate ignores x (src/proposal.jl:66-72)
        float fwd = 0.0f, bwd = 0.0f;
        if (a.prop_kind == MHX_PROP_DENSE) {
            for (int b = 0; b < nblk; ++b) {
                float n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * b + j;
                    if (k < d) {
                        ys[(long)k * ld] = n[j];
                        if (mu || stat) fwd = mhx_fma(n[j], n[j], fwd);
                        if (mu && !stat) { const float tk = n[j] + mu[d + k]; bwd = mhx_fma(tk, tk, bwd); }
                    }
                }
            }
            // y_r = x_r + sum_{j<=r} L_rj z_j ; rows in descending order so z can be overwritten in place
            for (int r = d - 1; r >= 0; --r) {
                const float* Lr = pvec + (long)r * (r + 1) / 2;
                float w = 0.0f;
                for (int j = 0; j <= r; ++j) w = mhx_fma(Lr[j], ys[(long)j * ld], w);
                const float xr = stat ? 0.0f : xs[(long)r * ld];
                ys[(long)r * ld] = mu ? xr + (mu[r] + w) : xr + w;
            }
        } else {
            for (int b = 0; b < nblk; ++b) {
                float n[4];
                mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * b + j;
                    if (k < d) {
                        const float s = a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k];
                        const float xk = stat ? 0.0f : xs[(long)k * ld];
                        if (mu) {
                            ys[(long)k * ld] = xk + mhx_fma(s, n[j], mu[k]);
                            fwd = mhx_fma(n[j], n[j], fwd);
                            const float tk = n[j] + mu[d + k];
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
        const float lpy = mhx_target_eval<TK>(a.target_kind, yv, d, tparams, a.ntparams, a.tconst);
        const float logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        // src/mh-core.jl:104-105 with logratio_proposal_density (src/proposal.jl:190-192) when the walk drifts
        // static proposal: logpdf(p, x) - logpdf(p, y) (src/proposal.jl:74-83), q(y) = -1/2 |z|^2
        const float qy = -0.5f * fwd;
        const float loga = stat ? (lpy - lp) + (qxc - qy) : (mu ? (lpy - lp) + 0.5f * (fwd - bwd) : (lpy - lp));
        const bool acc = logu < loga;
        lp = acc ? lpy : lp;
        qxc = (stat && acc) ? qy : qxc;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc));
        if (a.mom_mean && step == save_next) {
            // running moments kept in HBM (this kernel's state lives there anyway)
            ++mom_n;
            const float rn = 1.0f / (float)mom_n;
            const bool first = mom_n == 1u;
            for (int k = 0; k <= d; ++k) {
                float v;
                if (k < d) {
                    v = acc ? ys[(long)k * ld] : xs[(long)k * ld];
                    if (acc) xs[(long)k * ld] = v;
                } else {
                    v = lp;
                }
                float mean = first ? 0.0f : a.mom_mean[(long)k * ld + c];
                float m2v = first ? 0.0f : a.mom_m2[(long)k * ld + c];
                mhx_welford(v, rn, mean, m2v);
                a.mom_mean[(long)k * ld + c] = mean;
                a.mom_m2[(long)k * ld + c] = m2v;
            }
            save_next += (mhx_u32)a.thinning;
        } else if (step == save_next) {
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
    if (a.qx) a.qx[c] = qxc;
    a.acc_count[c] = nacc;
    a.last_acc[c] = last ? 1 : 0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// q(x) = -1/2 |L^-1 (x - mu)|^2 of every chain's current state: the static proposal's logpdf up to its
// constant (src/proposal.jl:31-35); forward substitution with the whitened vector in the scratch slab
MHX_DEV void mhx_rwmh_whiten_body(const mhx_rwmh_args& a, const float* __restrict__ pvec)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const long ld = a.ld;
    const int d = a.dim;
    const float* xs = a.x + c;
    float* ts = a.ybuf + c;
    const float* mu = a.pmean;
    float q = 0.0f;
    for (int i = 0; i < d; ++i) {
        const float r = mu ? xs[(long)i * ld] - mu[i] : xs[(long)i * ld];
        float t;
        if (a.prop_kind == MHX_PROP_ISO) t = r / a.pscale;
        else if (a.prop_kind == MHX_PROP_DIAG) t = r / pvec[i];
        else {
            const float* Li = pvec + (long)i * (i + 1) / 2;
            float acc = 0.0f;
            for (int j = 0; j < i; ++j) acc = mhx_fma(Li[j], ts[(long)j * ld], acc);
            t = (r - acc) / Li[i];
        }
        ts[(long)i * ld] = t;
        q = mhx_fma(t, t, q);
    }
    a.qx[c] = -0.5f * q;
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
template <int L, int NBL, int TK, int PK, bool MOM>
MHX_DEV void mhx_rwmh_coop_body(const mhx_rwmh_args& a, const float* __restrict__ tparams,
                                const float* __restrict__ pvec)
{
    constexpr int CPW = 64 / L;                    // chains per wave
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
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
    const mhx_u32 lane_off = ((mhx_u32)(4 * l) * (mhx_u32)ld + (mhx_u32)c) * 4u;     // bytes
    const int k_last = 4 * (l + L * (NBL - 1));               // first dimension of the last block

    float x[NBL][4], y[NBL][4];
#pragma unroll
    for (int i = 0; i < NBL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* col = a.x + (long)(4 * L * i + j) * ld;
            if (i < NBL - 1) x[i][j] = mhx_ld_off(col, lane_off);
            else x[i][j] = (k_last + j < d) ? mhx_ld_off(col, lane_off) : 0.0f;
            y[i][j] = 0.0f;
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
    // running moments (MOM): continue the Welford recursion of the previous launches
    float mm[MOM ? NBL : 1][4], m2[MOM ? NBL : 1][4], lpm = 0.0f, lpm2 = 0.0f;
    mhx_u32 mom_n = a.mom_n0;
    if (MOM) {
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = (i < NBL - 1) || (k_last + j < d);
                const long e = (long)(4 * L * i + j) * ld;
                mm[i][j] = (in && mom_n) ? mhx_ld_off(a.mom_mean + e, lane_off) : 0.0f;
                m2[i][j] = (in && mom_n) ? mhx_ld_off(a.mom_m2 + e, lane_off) : 0.0f;
            }
        if (mom_n) { lpm = a.mom_mean[(long)d * ld + c]; lpm2 = a.mom_m2[(long)d * ld + c]; }
    }

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        // Branch-free over the lane's blocks: a last block past the end of the vector (and the padding
        // dimensions of the final block) computes on zeros -- y = 0 there, and fma(0, 0, q) == q bit for
        // bit, so the partial sums need no predication.
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            const int b = l + L * i;
            float n[4];
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sc = a.pscale;
                if (PK != MHX_PROP_ISO) {
                    const int k = 4 * b + j;
                    sc = pvec[(i < NBL - 1 || k < d) ? k : 0];
                }
                float yk = mhx_fma(sc, n[j], x[i][j]);
                if (i == NBL - 1) yk = (k_last + j < d) ? yk : 0.0f;
                y[i][j] = yk;
                const float sq = mhx_fma(yk, yk, q);
                if (TK == MHX_TARGET_BANANA && i == 0 && j == 0) {
                    q = l == 0 ? (yk * yk) * 0.01f : sq;           // x1 ~ N(0, 100)
                } else if (TK == MHX_TARGET_BANANA && i == 0 && j == 1) {
                    const float y0 = y[0][0];
                    const float u = mhx_fma(tparams[0], mhx_fma(y0, y0, -100.0f), yk);
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
#pragma unroll
        for (int off = 1; off < L; off <<= 1) q = q + __shfl_xor(q, off * CPW, 64);
        float lpy;
        if (TK == MHX_TARGET_FUNNEL) {
            const float v = __shfl(y[0][0], cw, 64);           // x1 lives in lane l == 0 of the chain
            const float ev = mhx_exp(-v);
            float r = (v * v) * 0x1.c71c72p-5f;
            r = mhx_fma(0.5f * (float)(d - 1), v, r);
            r = mhx_fma(0.5f * ev, q, r);
            lpy = a.tconst - r;
        } else {
            lpy = mhx_fma(-0.5f, q, a.tconst);
        }
        const float logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = acc ? y[i][j] : x[i][j];
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && l == 0));
        if (MOM && step == save_next) {
            ++mom_n;
            const float rn = 1.0f / (float)mom_n;
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mhx_welford(x[i][j], rn, mm[i][j], m2[i][j]);
            mhx_welford(lp, rn, lpm, lpm2);
            save_next += (mhx_u32)a.thinning;
        } else if (step == save_next) {
            if (valid) {
                float* slotp = a.samples + slot * (long)(d + 1) * ld;
                const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(d + 1) * (mhx_u32)ld * 4u);
                const mhx_u32 ldb = (mhx_u32)ld * 4u;
#pragma unroll
                for (int i = 0; i < NBL; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const mhx_u32 rowb = (mhx_u32)(4 * L * i + j) * ldb;     // wave-uniform -> soffset
                        if (i < NBL - 1) mhx_srd_store(srd, lane_off, rowb, x[i][j]);
                        else if (k_last + j < d) mhx_srd_store(srd, lane_off, rowb, x[i][j]);
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
    if (valid) {
#pragma unroll
        for (int i = 0; i < NBL; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* col = a.x + (long)(4 * L * i + j) * ld;
                if (i < NBL - 1) mhx_st_off(col, lane_off, x[i][j]);
                else if (k_last + j < d) mhx_st_off(col, lane_off, x[i][j]);
            }
        if (l == 0) {
            a.lp[c] = lp;
            a.acc_count[c] = nacc;
            a.last_acc[c] = last ? 1 : 0;
        }
        if (MOM) {
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = (i < NBL - 1) || (k_last + j < d);
                    const long e = (long)(4 * L * i + j) * ld;
                    if (in) { mhx_st_off(a.mom_mean + e, lane_off, mm[i][j]); mhx_st_off(a.mom_m2 + e, lane_off, m2[i][j]); }
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
                    else if (a.pmean) xs[(long)k * ld] = 0.0f + mhx_fma(a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k], n[j], a.pmean[k]);
                    else xs[(long)k * ld] = mhx_fma(a.prop_kind == MHX_PROP_ISO ? a.pscale : pvec[k], n[j], 0.0f);
                }
            }
        }
        if (a.prop_kind == MHX_PROP_DENSE) {
            for (int r = d - 1; r >= 0; --r) {
                const float* Lr = pvec + (long)r * (r + 1) / 2;
                float w = 0.0f;
                for (int j = 0; j <= r; ++j) w = mhx_fma(Lr[j], xs[(long)j * ld], w);
                xs[(long)r * ld] = a.pmean ? 0.0f + (a.pmean[r] + w) : 0.0f + w;
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
MHX_DEV void mhx_target_eval_body(const float* __restrict__ x, float* __restrict__ lp, const int n,
                                  const int d, const int kind, const float* __restrict__ tparams,
                                  const int ntparams, const float tconst, const int lanes)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    mhx_strided_x xv;
    xv.base = x + c;
    xv.ld = n;
    lp[c] = mhx_target_eval_lanes<TK>(kind, xv, d, tparams, ntparams, tconst, lanes);
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

// first folded sample = the current state (discard_initial == 0): mean = state, M2 = 0
MHX_DEV void mhx_moments_first_body(const float* __restrict__ x, const float* __restrict__ lp, float* mean, float* m2,
                                    const int n, const long ld, const int d)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    for (int k = 0; k < d; ++k) { mean[(long)k * ld + c] = x[(long)k * ld + c]; m2[(long)k * ld + c] = 0.0f; }
    mean[(long)d * ld + c] = lp[c];
    m2[(long)d * ld + c] = 0.0f;
}

// JIT entry points: hiprtc compiles this header with the specialisation macros defined
#ifdef MHX_JIT_RWMH_REG
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_rwmh_reg(const mhx_rwmh_args a, const float* __restrict__ tparams, const float* __restrict__ pvec)
{
    mhx_rwmh_reg_body<MHX_JIT_DIM, MHX_JIT_TK, MHX_JIT_PK>(a, tparams, pvec);
}
#endif
#ifdef MHX_JIT_RWMH_COOP
extern "C" __global__ void __launch_bounds__(256, (MHX_JIT_NBL) <= 5 ? 4 : ((MHX_JIT_NBL) <= 13 ? 2 : 1))
mhx_jit_rwmh_coop(const mhx_rwmh_args a, const float* __restrict__ tparams, const float* __restrict__ pvec)
{
    mhx_rwmh_coop_body<MHX_JIT_L, MHX_JIT_NBL, MHX_JIT_TK, MHX_JIT_PK, (MHX_JIT_MOM != 0)>(a, tparams, pvec);
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
                    const int kind, const float* __restrict__ tparams, const int ntparams, const float tconst,
                    const int lanes)
{
    mhx_target_eval_body<MHX_JIT_TK>(x, lp, n, d, kind, tparams, ntparams, tconst, lanes);
}
#endif
