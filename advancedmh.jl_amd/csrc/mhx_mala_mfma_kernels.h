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
        fq = mhx_butterfly_add<16>(fq);
        fq = mhx_butterfly_add<32>(fq);
        // ---- value and gradient at the candidate (:73-75): w = A y, lp' = -1/2 |w|^2 + const, grad = -A^T w
        mhx_real w[NS], gy[NS];
        mhx_real q = MHX_R(0.0);
        if constexpr (STREAM) mhx_mfma_rows_stream<D, 3, true, true, BUFM>(sA, sAT, ring, lane, ys, q, w, pf, parity);
        else mhx_mfma_rows<D, 3>(Aimg, lane, ys, q, w);
        q = mhx_butterfly_add<16>(q);
        q = mhx_butterfly_add<32>(q);
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
        bq = mhx_butterfly_add<16>(bq);
        bq = mhx_butterfly_add<32>(bq);
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
                mhx_u32 roff = 0u;                               // (running row offset behind an opaque asm: MHX_COOP_REC_RUN, mhx_rwmh_kernels.h)
                asm volatile("" : "+s"(roff));
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (4 * s + 3 < D || 4 * s + g < D) mhx_srd_store<MHX_REC_STORE_AUX>(srd, lane_off, roff, xat(s));
                    roff += 4u * ldb;
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


// ---- LEAN mode: ONE vector per lane.  The largest dimensions (fp64 d = 201 ... 512, fp32 d = 401 ... 1000) leave a lane room
// for one vector of ceil(d/4) reals beside the streaming registers, so the step is arranged around a single array v:
//   v = y                       the candidate (its noise goes to the z slab as it is drawn)
//   v = A y    IN PLACE         row tiles in DESCENDING order: tile t reads y[0 .. 16t+15], so once it is done the y of its own
//                               rows is dead and the fragment takes its place (an accumulator is the same ascending fma chain
//                               whatever order the tiles are visited in)
//   q = |v|^2                   afterwards, in ascending row order -- the spec's order
//   v = A^T v  IN PLACE         tiles ascending: tile t reads w[16t ..]: dead below, replaced by the fragment
//   grad y = -v, the backward sum with z re-read from its slab; on accept y is RE-FORMED from x, grad x and z (same expression,
//   same operands: same bits) and x, grad x are rewritten.
// Both images stream through the ring one tile (or tile pair) per chunk, A from its LAST chunk down, A^T from its first up.
template <int D, bool PAIR, long BUFB, int NPF>
MHX_DEV void mhx_mfma_rows_stream_desc(const mhx_srd gimg, const mhx_srd gnextT, mhx_real* ring, const int lane,
                                       mhx_real (&v)[mhx_mfma_geom<D>::NS], mhx_piece16 (&pf)[NPF], int& parity)
{
    typedef mhx_mfma_geom<D> GEO;
    typedef mhx_mfma_stream_geom<D, PAIR> SG;
#pragma unroll
    for (int p = SG::NP - 1; p >= 0; --p) {
        mhx_acc4* buf = (mhx_acc4*)((char*)ring + (parity ? BUFB : 0));
#pragma unroll
        for (int i = 0; i < SG::PF; ++i)
            if (i < SG::pieces(p)) ((mhx_piece16*)buf)[(int)threadIdx.x + i * SG::THREADS] = pf[i];
        __syncthreads();
        if (p > 0) mhx_mfma_chunk_load<D, PAIR>(gimg, p - 1, pf);
        else mhx_mfma_chunk_load<D, PAIR, true>(gnextT, 0, pf);                   // next: A^T from its first chunk
        constexpr mhx_acc4 zero = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
        mhx_acc4 c[2] = {zero, zero};
        const int t0 = SG::TPC * p;
        const int t1 = PAIR && t0 + 1 < GEO::NT ? t0 + 1 : t0;
#pragma unroll
        for (int grp = 0; grp < GEO::groups(t1); ++grp) {
#pragma unroll
            for (int h = 0; h < SG::TPC; ++h) {
                const int t = t0 + h;
                if (t < GEO::NT && grp < GEO::groups(t)) {
                    const mhx_acc4 a4 = buf[(GEO::first(t) / 4 + grp - SG::first(p)) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * grp + u < GEO::steps(t)) c[h] = MHX_MFMA16(a4[u], v[4 * grp + u], c[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < SG::TPC; ++h) {
            const int t = t0 + h;
            if (t < GEO::NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < GEO::NS) v[4 * t + r] = c[h][r];               // rows 16t.. of A y replace y[16t..]: no lower tile reads them
            }
        }
        parity ^= 1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// A^T w over the streamed image of A^T, in place (tiles ascending), one tile or a pair per chunk; on return `pf` holds the LAST
// chunk of the lower image `gnext` (where the next step's descending A y starts)
template <int D, bool PAIR, long BUFB, int NPF>
MHX_DEV void mhx_mfma_rows_T_stream_inplace(const mhx_srd gimgT, const mhx_srd gnext, mhx_real* ring, const int lane,
                                            mhx_real (&v)[mhx_mfma_geom<D>::NS], mhx_piece16 (&pf)[NPF], int& parity)
{
    typedef mhx_mfma_geom<D> GEO;
    typedef mhx_mfma_stream_geom<D, PAIR, true> SG;
    typedef mhx_mfma_stream_geom<D, PAIR, false> SGA;
#pragma unroll
    for (int p = 0; p < SG::NP; ++p) {
        mhx_acc4* buf = (mhx_acc4*)((char*)ring + (parity ? BUFB : 0));
#pragma unroll
        for (int i = 0; i < SG::PF; ++i)
            if (i < SG::pieces(p)) ((mhx_piece16*)buf)[(int)threadIdx.x + i * SG::THREADS] = pf[i];
        __syncthreads();
        if (p + 1 < SG::NP) mhx_mfma_chunk_load<D, PAIR, true>(gimgT, p + 1, pf);
        else mhx_mfma_chunk_load<D, PAIR, false>(gnext, SGA::NP - 1, pf);
        constexpr mhx_acc4 zero = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
        mhx_acc4 c[2] = {zero, zero};
        const int t0 = SG::TPC * p;
#pragma unroll
        for (int grp = t0; grp < GEO::NT; ++grp) {
#pragma unroll
            for (int h = 0; h < SG::TPC; ++h) {
                const int t = t0 + h;
                if (t < GEO::NT && grp >= t) {
                    const mhx_acc4 a4 = buf[(GEO::firstT(t) + grp - t - SG::first(p)) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * grp + u < GEO::NS) c[h] = MHX_MFMA16(a4[u], v[4 * grp + u], c[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < SG::TPC; ++h) {
            const int t = t0 + h;
            if (t < GEO::NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < GEO::NS) v[4 * t + r] = c[h][r];               // no later tile reads w below its own rows
            }
        }
        parity ^= 1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int D, bool PAIR>
MHX_DEV void mhx_mala_mfma_lean_body(const mhx_mala_args& a, mhx_real* Aimg, mhx_real* ATimg, mhx_real* ring)
{
    typedef mhx_mfma_geom<D> GEO;
    constexpr int NS = GEO::NS;
    constexpr int NQD = (NS + 3) / 4;
    typedef mhx_mfma_stream_geom<D, PAIR, false> SGA;
    typedef mhx_mfma_stream_geom<D, PAIR, true> SGT;
    constexpr int PFM = SGA::PF > SGT::PF ? SGA::PF : SGT::PF;
    constexpr long BUFM = SGA::BUF_BYTES > SGT::BUF_BYTES ? SGA::BUF_BYTES : SGT::BUF_BYTES;
    mhx_piece16 pf[PFM];
    int parity = 0;
    const mhx_srd sA = mhx_make_srd(Aimg, (mhx_u32)(GEO::REALS * (long)sizeof(mhx_real)));
    const mhx_srd sAT = mhx_make_srd(ATimg, (mhx_u32)(GEO::REALS_T * (long)sizeof(mhx_real)));
    mhx_mfma_chunk_load<D, PAIR, false>(sA, SGA::NP - 1, pf);

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
    const mhx_srd xsrd = mhx_make_srd(a.x, (mhx_u32)D * ldb), gsrd = mhx_make_srd(a.gx, (mhx_u32)D * ldb);
    const mhx_srd zsrd = mhx_make_srd(a.zbuf, (mhx_u32)D * ldb);
    auto own = [&](const int s) -> bool { return 4 * s + 3 < D || 4 * s + g < D; };
    auto xat = [&](const int s) -> mhx_real { return own(s) ? mhx_srd_load(xsrd, lane_off, (mhx_u32)(4 * s) * ldb) : MHX_R(0.0); };
    auto gat = [&](const int s) -> mhx_real { return own(s) ? mhx_srd_load(gsrd, lane_off, (mhx_u32)(4 * s) * ldb) : MHX_R(0.0); };
    auto zat = [&](const int s) -> mhx_real { return own(s) ? mhx_srd_load(zsrd, lane_off, (mhx_u32)(4 * s) * ldb) : MHX_R(0.0); };
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
        asm volatile("" ::: "memory");
        mhx_real v[NS];
        mhx_real fq = MHX_R(0.0);
        // ---- noise and candidate (src/MALA.jl:70); the noise of dimension 4s + g goes to the z slab
#pragma unroll
        for (int qd = 0; qd < NQD; ++qd) {
            mhx_real n[4];
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)(4 * qd + g), n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k0 = 4 * (4 * qd) + e;
                if (k0 + 12 >= D) n[e] = (k0 + 4 * g < D) ? n[e] : MHX_R(0.0);
                fq = mhx_fma(n[e], n[e], fq);
            }
            mhx_lanes4_transpose(n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 4 * qd + e;
                if (s < NS) {
                    if (own(s) && valid) mhx_srd_store(zsrd, lane_off, (mhx_u32)(4 * s) * ldb, n[e]);
                    v[s] = mhx_fma(a.sigma, n[e], mhx_fma(a.h, gat(s), xat(s)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        fq = mhx_butterfly_add<16>(fq);
        fq = mhx_butterfly_add<32>(fq);
        // ---- w = A y (in place, tiles descending), lp' = -1/2 |w|^2 + const in ascending row order
        mhx_mfma_rows_stream_desc<D, PAIR, BUFM>(sA, sAT, ring, lane, v, pf, parity);
        mhx_real q = MHX_R(0.0);
#pragma unroll
        for (int s = 0; s < NS; ++s) q = mhx_fma(v[s], v[s], q);
        q = mhx_butterfly_add<16>(q);
        q = mhx_butterfly_add<32>(q);
        const mhx_real lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        // ---- grad = -A^T w (in place, tiles ascending)
        mhx_mfma_rows_T_stream_inplace<D, PAIR, BUFM>(sAT, sA, ring, lane, v, pf, parity);
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = -v[s];
        // ---- log ratio of the proposal densities (:78-80)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (the z slab written above is read back here by the same lane)
        mhx_real bq = MHX_R(0.0);
#pragma unroll
        for (int qd = 0; qd < NQD; ++qd) {
            mhx_real n[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 4 * qd + e;
                n[e] = s < NS ? mhx_fma(a.hs, gat(s) + v[s], zat(s)) : MHX_R(0.0);
            }
            mhx_lanes4_transpose(n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bq = mhx_fma(n[e], n[e], bq);
        }
        bq = mhx_butterfly_add<16>(bq);
        bq = mhx_butterfly_add<32>(bq);
        const mhx_real loga = (lpy - lp) + MHX_R(0.5) * (fq - bq);
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;
        if (acc && valid) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (own(s)) {
                    const mhx_real ynew = mhx_fma(a.sigma, zat(s), mhx_fma(a.h, gat(s), xat(s)));   // the candidate again: same operands, same bits
                    mhx_srd_store(xsrd, lane_off, (mhx_u32)(4 * s) * ldb, ynew);
                    mhx_srd_store(gsrd, lane_off, (mhx_u32)(4 * s) * ldb, v[s]);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the slabs are current before anything reads them again
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && g == 0));
        if (step == save_next) {
            if (valid) {
                mhx_real* slotp = a.samples + slot * (long)(D + 1) * ld;
                const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(D + 1) * (mhx_u32)ld * MHX_RB);
                mhx_u32 roff = 0u;                               // (running row offset behind an opaque asm: MHX_COOP_REC_RUN, mhx_rwmh_kernels.h)
                asm volatile("" : "+s"(roff));
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (own(s)) mhx_srd_store<MHX_REC_STORE_AUX>(srd, lane_off, roff, xat(s));
                    roff += 4u * ldb;
                    if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
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
    if (valid && g == 0) {
        a.lp[c] = lp;
        a.acc_count[c] = nacc;
        a.last_acc[c] = last ? 1 : 0;
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
#ifdef MHX_JIT_MALA_MFMA_LEAN
#ifndef MHX_JIT_PAIR
#define MHX_JIT_PAIR 0
#endif
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_mfma_image(const mhx_real* __restrict__ packed, mhx_real* img) { mhx_mfma_image_fill<MHX_JIT_DIM>(packed, img); }
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_mfma_image_T(const mhx_real* __restrict__ packed, mhx_real* img) { mhx_mfma_image_fill_T<MHX_JIT_DIM>(packed, img); }
extern "C" __global__ void __launch_bounds__(64 * MHX_MFMA_WAVES, 1)
mhx_jit_mala_mfma_lean(const mhx_mala_args a, const mhx_real* __restrict__ tparams, mhx_real* gAimg, mhx_real* gATimg)
{
    extern __shared__ mhx_acc4 mhx_mala_mfma_ring[];
    mhx_mala_mfma_lean_body<MHX_JIT_DIM, (MHX_JIT_PAIR != 0)>(a, gAimg, gATimg, (mhx_real*)mhx_mala_mfma_ring);
}
#endif
MHX_NS_END
