// mhx_rwmh_mfma_kernels.h -- random-walk Metropolis with a dense factor shared by all chains, on the matrix cores.
//
// With ONE factor for every chain -- the dense Gaussian target's A = inv(chol(Sigma)), a dense proposal's Cholesky
// factor, or both -- the row products A y of the chains of a wave are a real GEMM: [16 rows] x [16 chains] tiles of
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64.  The f32-input (and the f64) MFMA is an exact fma chain in k order --
// D = fma(a_k3, b_k3, fma(a_k2, b_k2, fma(a_k1, b_k1, fma(a_k0, b_k0, C)))), one rounding per product -- so the
// accumulator of row i after the k-steps 0 .. i/4 IS the arithmetic spec's w_i = sum_{j<=i} A_ij y_j (ascending j, fma from
// 0; the zeros of the upper triangle add nothing): bit for bit the chain of mhx_rwmh_dense_kernels.h and of the oracle.
// What the matrix instruction buys is not flops -- its f32 / f64 rate equals the vector rate, and on gfx950 it does not even
// run beside the VALU (tools/ubench/mfma_overlap.hip: MFMAs + v_fma take the SUM of their times) -- but operands: one A
// operand per lane feeds 4 x 16 x 16 multiply-adds where the vector kernel reads one LDS value per lane per fma.
//
// Geometry.  A wave holds 16 chains; chain j = lane & 15 is spread over the four lanes g = lane >> 4:
//   * lane g owns the dimensions k = 4s + g (s = 0 .. NS-1): exactly the B operand of k-step s (B[k = lane>>4][j]),
//     so the candidate feeds the MFMAs from registers -- no LDS round trip for y;
//   * the C/D fragment gives lane g the rows {4g + r} (f32) or {g + 4r} (f64) of a 16-row tile; the A image permutes
//     the f32 rows so that both widths hold rows i = 16t + 4r + g in accumulator r -- the oracle's reduction shape
//     L = 4 (lane g owns rows g, g+4, ...; partial sums meet in the xor butterfly 1, 2 = lanes ^16, ^32);
//   * the same fragment is the dense proposal's xi = L z: row 16t + 4r + g lands on the lane that owns dimension
//     4(4t+r) + g.
// Proposal noise: a Philox "normal4" block is 4 consecutive dimensions = one per lane of a chain; lane g draws the
// blocks b = g, g+4, ... and a 4x4 transpose over the chain's lanes (v_permlane32_swap + v_permlane16_swap) hands every
// normal to its owner -- each block is drawn once.
//
// Same step as mhx_rwmh_reg_body (src/mh-core.jl:92-117; proposal src/proposal.jl:49-56).
#pragma once
#include "mhx_rwmh_kernels.h"

MHX_NS_BEGIN

#define MHX_MFMA_WAVES 4                          // waves per block (64 chains)

#if MHX_REAL64
typedef double mhx_acc4 __attribute__((ext_vector_type(4)));
#define MHX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#else
typedef float mhx_acc4 __attribute__((ext_vector_type(4)));
#define MHX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// The operand image of a packed lower-triangular factor: for tile t (rows 16t .. 16t+15) the k-steps s = 0 .. S(t)-1
// with S(t) = min(4(t+1), NS), padded to whole groups of 4 steps; a group holds, per lane, the 4 A operands of its
// steps contiguously (one 16-byte LDS read in fp32, two in fp64).  Tile t starts at step 2t(t+1).
template <int D>
struct mhx_mfma_geom {
    static constexpr int NS = (D + 3) / 4;                        // k-steps = reals of state per lane
    static constexpr int NT = (D + 15) / 16;                      // row tiles
    static constexpr int steps(int t) { return 4 * (t + 1) < NS ? 4 * (t + 1) : NS; }
    static constexpr int groups(int t) { return (steps(t) + 3) / 4; }
    static constexpr int first(int t) { return 2 * t * (t + 1); }
    static constexpr int TOTAL = first(NT - 1) + 4 * groups(NT - 1);    // steps in the image
    static constexpr long REALS = (long)TOTAL * 64;
    // the transposed factor (A^T w, upper triangular): tile t needs the k-groups t .. NT-1 (k = row of A >= 16t)
    static constexpr int firstT(int t) { return t * NT - t * (t - 1) / 2; }      // in groups of 4 steps
    static constexpr int GROUPS_T = NT * (NT + 1) / 2;
    static constexpr long REALS_T = (long)GROUPS_T * 256;
};

// matrix row behind MFMA row m of tile t: the permutation that puts row 16t + 4r + g into accumulator r of lane group g
MHX_DEV int mhx_mfma_row(int t, int m)
{
#if MHX_REAL64
    return 16 * t + m;                              // C/D row = g + 4r
#else
    return 16 * t + 4 * (m & 3) + (m >> 2);        // C/D row = 4g + r
#endif
}

template <int D>
MHX_DEV void mhx_mfma_image_fill(const mhx_real* __restrict__ A, mhx_real* img)
{
    typedef mhx_mfma_geom<D> GEO;
    for (int e = threadIdx.x; e < GEO::TOTAL * 64; e += blockDim.x) {
        const int sg = e >> 6, ln = e & 63;
        int t = 0;
        while (t + 1 < GEO::NT && GEO::first(t + 1) <= sg) ++t;
        const int s = sg - GEO::first(t);
        const int i = mhx_mfma_row(t, ln & 15);
        const int k = 4 * s + (ln >> 4);
        const bool in = i < D && k <= i;
        const mhx_real v = A[in ? (long)i * (i + 1) / 2 + k : 0];
        img[(GEO::first(t) + (s & ~3)) * 64 + ln * 4 + (s & 3)] = in ? v : MHX_R(0.0);
    }
}

// image of A^T: operand (MFMA row m of tile t, k) = A[k][row(t, m)] for k >= row, groups of 4 k-steps from 4t on
template <int D>
MHX_DEV void mhx_mfma_image_fill_T(const mhx_real* __restrict__ A, mhx_real* img)
{
    typedef mhx_mfma_geom<D> GEO;
    for (int e = threadIdx.x; e < GEO::GROUPS_T * 256; e += blockDim.x) {
        const int gi = e >> 8, ln = (e >> 2) & 63, u = e & 3;
        int t = 0;
        while (t + 1 < GEO::NT && GEO::firstT(t + 1) <= gi) ++t;
        const int grp = t + (gi - GEO::firstT(t));
        const int i = 4 * (4 * grp + u) + (ln >> 4);             // k: the row of A
        const int j = mhx_mfma_row(t, ln & 15);                  // the output row: the column of A
        const bool in = i < D && j < D && i >= j;
        const mhx_real v = A[in ? (long)i * (i + 1) / 2 + j : 0];
        img[e] = in ? v : MHX_R(0.0);
    }
}

// rows of `factor image` x `b`, two tiles at a time (two independent accumulator chains keep the matrix pipe issuing:
// a dependent 16x16x4 MFMA waits 40 cycles, the issue interval is 32).  Component r of tile t's fragment is row
// 16t + 4r + g.  MODE & 1: fold the rows into q = fma(w, w, q) in ascending row order; MODE & 2: store them in out[4t + r].
template <int D, int MODE>
MHX_DEV void mhx_mfma_rows(const mhx_real* img, const int lane, const mhx_real (&b)[mhx_mfma_geom<D>::NS],
                           mhx_real& q, mhx_real (&out)[mhx_mfma_geom<D>::NS])
{
    typedef mhx_mfma_geom<D> GEO;
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
                    const mhx_acc4 a4 = ((const mhx_acc4*)img)[(GEO::first(t) / 4 + grp) * 64 + lane];
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
                for (int r = 0; r < 4; ++r) {
                    if (MODE & 1) q = mhx_fma(c[h][r], c[h][r], q);
                    if ((MODE & 2) && 4 * t + r < GEO::NS) out[4 * t + r] = c[h][r];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);            // keep the operand loads of later tiles out of this pair's registers
    }
}

// rows of (image of A^T) x `b`: out[4t + r] = sum_{k >= row} A[k][row] b_k, ascending k from 0 (the zeros below `row` add nothing)
template <int D>
MHX_DEV void mhx_mfma_rows_T(const mhx_real* imgT, const int lane, const mhx_real (&b)[mhx_mfma_geom<D>::NS],
                             mhx_real (&out)[mhx_mfma_geom<D>::NS])
{
    typedef mhx_mfma_geom<D> GEO;
#pragma unroll
    for (int t0 = 0; t0 < GEO::NT; t0 += 2) {
        constexpr mhx_acc4 zero = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
        mhx_acc4 c[2] = {zero, zero};
#pragma unroll
        for (int grp = t0; grp < GEO::NT; ++grp) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = t0 + h;
                if (t < GEO::NT && grp >= t) {
                    const mhx_acc4 a4 = ((const mhx_acc4*)imgT)[(GEO::firstT(t) + grp - t) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * grp + u < GEO::NS) c[h] = MHX_MFMA16(a4[u], b[4 * grp + u], c[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + h;
            if (t < GEO::NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < GEO::NS) out[4 * t + r] = c[h][r];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- images larger than a block's LDS: streamed.  The image lives in HBM / L2 (built once per run); a block walks it in
// chunks of one tile PAIR (adjacent in the image) through a two-buffer LDS ring: every thread holds its share of the NEXT
// chunk in registers while the MFMAs of the current one run, drops it into the other buffer, one block barrier per chunk.
// PAIR: a chunk is a tile pair (two independent accumulator chains); otherwise one tile (half the ring: the largest images).
// TRANS: the image of A^T (tile t holds the k-groups t .. NT-1).
template <int D, bool PAIR = true, bool TRANS = false>
struct mhx_mfma_stream_geom {
    typedef mhx_mfma_geom<D> GEO;
    static constexpr int THREADS = 64 * MHX_MFMA_WAVES;
    static constexpr int TPC = PAIR ? 2 : 1;                                      // tiles per chunk
    static constexpr int NP = (GEO::NT + TPC - 1) / TPC;                          // chunks per image
    static constexpr int tgroups(int t) { return TRANS ? GEO::NT - t : GEO::groups(t); }
    static constexpr int tfirst(int t) { return TRANS ? GEO::firstT(t) : GEO::first(t) / 4; }         // in groups
    static constexpr int first(int p) { return tfirst(TPC * p); }
    static constexpr int groups(int p) { return tgroups(TPC * p) + (PAIR && 2 * p + 1 < GEO::NT ? tgroups(2 * p + 1) : 0); }
    static constexpr int maxg() { int m = 0; for (int p = 0; p < NP; ++p) m = groups(p) > m ? groups(p) : m; return m; }
    static constexpr int MAXG = maxg();
    static constexpr int P16 = (int)sizeof(mhx_acc4) / 16;                        // 16-byte pieces per lane and group
    static constexpr int PF = (MAXG * 64 * P16 + THREADS - 1) / THREADS;          // 16-byte pieces per thread and chunk
    static constexpr int pieces(int p) { return (groups(p) * 64 * P16 + THREADS - 1) / THREADS; }   // per thread, whole rounds
    static constexpr long BUF_BYTES = (long)PF * THREADS * 16;                    // one ring buffer
    static constexpr long RING_BYTES = 2 * BUF_BYTES;
};

// this thread's pieces of chunk p of the image behind the descriptor `img`: wave-uniform chunk offset (SGPR), one 32-bit lane
// offset -- no 64-bit vector address per piece
typedef mhx_u32 mhx_piece16 __attribute__((ext_vector_type(4)));
template <int D, bool PAIR, bool TRANS = false, int NPF>
MHX_DEV void mhx_mfma_chunk_load(const mhx_srd img, const int p, mhx_piece16 (&pf)[NPF])
{
    typedef mhx_mfma_stream_geom<D, PAIR, TRANS> SG;
    static_assert(NPF >= SG::PF, "prefetch registers for the largest chunk");
    // whole rounds of the block, no per-thread predicate: the last round may run into the next chunk (ignored) or past the
    // image (the descriptor's range check returns zeros)
    const mhx_u32 voff = (mhx_u32)threadIdx.x * 16u;
#pragma unroll
    for (int i = 0; i < SG::PF; ++i) {
        const mhx_u32 soff = (mhx_u32)(SG::first(p) * 64 * (int)sizeof(mhx_acc4) + i * SG::THREADS * 16);
        if (i < SG::pieces(p)) pf[i] = __builtin_amdgcn_raw_buffer_load_b128(img, (int)voff, (int)soff, 0);
    }
}

// mhx_mfma_rows over a streamed image.  `pf` holds chunk 0 of this image on entry and, on return, chunk 0 of `gnext` (the image
// the step walks next -- itself if it is the only one); `parity` is the ring buffer the next chunk goes to.
// BUFB: bytes of one ring buffer (the caller's, when images of different chunk sizes share the ring)
template <int D, int MODE, bool PAIR, bool NEXT_TRANS = false, long BUFB = mhx_mfma_stream_geom<D, PAIR>::BUF_BYTES, int NPF>
MHX_DEV void mhx_mfma_rows_stream(const mhx_srd gimg, const mhx_srd gnext, mhx_real* ring, const int lane,
                                  const mhx_real (&b)[mhx_mfma_geom<D>::NS], mhx_real& q, mhx_real (&out)[mhx_mfma_geom<D>::NS],
                                  mhx_piece16 (&pf)[NPF], int& parity)
{
    typedef mhx_mfma_geom<D> GEO;
    typedef mhx_mfma_stream_geom<D, PAIR> SG;
#pragma unroll
    for (int p = 0; p < SG::NP; ++p) {
        mhx_acc4* buf = (mhx_acc4*)((char*)ring + (parity ? BUFB : 0));
#pragma unroll
        for (int i = 0; i < SG::PF; ++i)
            if (i < SG::pieces(p)) ((mhx_piece16*)buf)[(int)threadIdx.x + i * SG::THREADS] = pf[i];
        __syncthreads();                                  // the chunk is complete; the other buffer is free (see above)
        if (p + 1 < SG::NP) mhx_mfma_chunk_load<D, PAIR>(gimg, p + 1, pf);
        else mhx_mfma_chunk_load<D, PAIR, NEXT_TRANS>(gnext, 0, pf);
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
                        if (4 * grp + u < GEO::steps(t)) c[h] = MHX_MFMA16(a4[u], b[4 * grp + u], c[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < SG::TPC; ++h) {
            const int t = t0 + h;
            if (t < GEO::NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (MODE & 1) q = mhx_fma(c[h][r], c[h][r], q);
                    if ((MODE & 2) && 4 * t + r < GEO::NS) out[4 * t + r] = c[h][r];
                }
            }
        }
        parity ^= 1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// mhx_mfma_rows_T over a streamed image of A^T (tile pairs).  `pf` holds chunk 0 of this image on entry and chunk 0 of the
// LOWER image `gnext` on return (the next step's A y); NEXT_TRANS = false says so to the loader.
template <int D, long BUFB, int NPF>
MHX_DEV void mhx_mfma_rows_T_stream(const mhx_srd gimgT, const mhx_srd gnext, mhx_real* ring, const int lane,
                                    const mhx_real (&b)[mhx_mfma_geom<D>::NS], mhx_real (&out)[mhx_mfma_geom<D>::NS],
                                    mhx_piece16 (&pf)[NPF], int& parity)
{
    typedef mhx_mfma_geom<D> GEO;
    typedef mhx_mfma_stream_geom<D, true, true> SG;
#pragma unroll
    for (int p = 0; p < SG::NP; ++p) {
        mhx_acc4* buf = (mhx_acc4*)((char*)ring + (parity ? BUFB : 0));
#pragma unroll
        for (int i = 0; i < SG::PF; ++i)
            if (i < SG::pieces(p)) ((mhx_piece16*)buf)[(int)threadIdx.x + i * SG::THREADS] = pf[i];
        __syncthreads();
        if (p + 1 < SG::NP) mhx_mfma_chunk_load<D, true, true>(gimgT, p + 1, pf);
        else mhx_mfma_chunk_load<D, true, false>(gnext, 0, pf);
        constexpr mhx_acc4 zero = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
        mhx_acc4 c[2] = {zero, zero};
        const int t0 = 2 * p;
#pragma unroll
        for (int grp = t0; grp < GEO::NT; ++grp) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = t0 + h;
                if (t < GEO::NT && grp >= t) {
                    const mhx_acc4 a4 = buf[(GEO::firstT(t) + grp - t - SG::first(p)) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * grp + u < GEO::NS) c[h] = MHX_MFMA16(a4[u], b[4 * grp + u], c[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + h;
            if (t < GEO::NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * t + r < GEO::NS) out[4 * t + r] = c[h][r];
            }
        }
        parity ^= 1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// 4x4 transpose over the four lanes of a chain (lanes j, j+16, j+32, j+48): lane g gives n[e] to lane e and receives
// that lane's n[g].  v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of the second,
// v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second.
MHX_DEV void mhx_swap32(mhx_u32& a, mhx_u32& b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
MHX_DEV void mhx_swap16(mhx_u32& a, mhx_u32& b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
MHX_DEV void mhx_lanes4_transpose(mhx_real (&n)[4])
{
#if MHX_REAL64
    mhx_u32 lo[4], hi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const mhx_u64 w = (mhx_u64)__double_as_longlong(n[e]); lo[e] = (mhx_u32)w; hi[e] = (mhx_u32)(w >> 32); }
    mhx_swap32(lo[0], lo[2]); mhx_swap32(hi[0], hi[2]);
    mhx_swap32(lo[1], lo[3]); mhx_swap32(hi[1], hi[3]);
    mhx_swap16(lo[0], lo[1]); mhx_swap16(hi[0], hi[1]);
    mhx_swap16(lo[2], lo[3]); mhx_swap16(hi[2], hi[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) n[e] = __longlong_as_double((long long)(((mhx_u64)hi[e] << 32) | lo[e]));
#else
    mhx_u32 w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(n[e]);
    mhx_swap32(w[0], w[2]);
    mhx_swap32(w[1], w[3]);
    mhx_swap16(w[0], w[1]);
    mhx_swap16(w[2], w[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) n[e] = __uint_as_float(w[e]);
#endif
}

// TK: MHX_TARGET_CORR_GAUSS (factor image A) or MHX_TARGET_ISO_GAUSS (with a dense proposal); PK: ISO / DIAG scales, or
// DENSE -- the proposal's Cholesky factor as a second image.
// STREAM: Aimg / Limg are the pre-built images in global memory and `ring` the LDS ring they are walked through; otherwise
// they are LDS and filled here.
// PAIR: see mhx_mfma_stream_geom.  XMEM: the chain state x is NOT held in registers -- the candidate alone is (the B operands) --
// but re-read from its [dim][ld] slab when the candidate is formed and written back on accept: the largest dimensions.
template <int D, int PK, int TK, bool STREAM = false, bool PAIR = true, bool XMEM = false>
MHX_DEV void mhx_rwmh_mfma_body(const mhx_rwmh_args& a, const mhx_real* __restrict__ A, const mhx_real* __restrict__ pvec,
                                mhx_real* Aimg, mhx_real* Limg, mhx_real* ring = nullptr)
{
    typedef mhx_mfma_geom<D> GEO;
    constexpr bool CORR = TK == MHX_TARGET_CORR_GAUSS;
    constexpr bool DENSEP = PK == MHX_PROP_DENSE;
    constexpr int NS = GEO::NS;
    constexpr int NQD = (NS + 3) / 4;                 // quads of normal4 blocks (block b = dimensions 4b .. 4b+3)
    if (!STREAM) {
        if (CORR) mhx_mfma_image_fill<D>(A, Aimg);
        if (DENSEP) mhx_mfma_image_fill<D>(pvec, Limg);
        __syncthreads();
    }
    // streamed images: the first chunk of the step's first image is in flight before the loop
    mhx_piece16 pf[STREAM ? mhx_mfma_stream_geom<D, PAIR>::PF : 1];
    int parity = 0;
    const mhx_u32 img_bytes = (mhx_u32)(GEO::REALS * (long)sizeof(mhx_real));
    const mhx_srd sA = mhx_make_srd(STREAM ? (CORR ? Aimg : Limg) : nullptr, img_bytes), sL = mhx_make_srd(STREAM ? (DENSEP ? Limg : Aimg) : nullptr, img_bytes);
    if constexpr (STREAM) mhx_mfma_chunk_load<D, PAIR>(DENSEP ? sL : sA, 0, pf);

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;
    const long c_raw = ((long)blockIdx.x * MHX_MFMA_WAVES + wave) * 16 + j;
    const bool valid = c_raw < a.nchains;
    const long c = valid ? c_raw : (long)a.nchains - 1;      // idle groups shadow the last chain (loads only)
    const long ld = a.ld;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);

    // rows 4s of a [dim][ld] slab are wave-uniform (SGPR offsets); the lane adds (g ld + c) reals -- the register /
    // cooperative kernels only run on slabs below 4 GB
    const mhx_u32 lane_off = ((mhx_u32)g * (mhx_u32)ld + (mhx_u32)c) * MHX_RB;
    const mhx_u32 ldb = (mhx_u32)ld * MHX_RB;
    // ---- state: dimensions 4s + g (ABI layout [dim][ld], touched once per launch); the pad stays zero
    mhx_real xs[XMEM ? 1 : NS], sc[PK == MHX_PROP_DIAG && !XMEM ? NS : 1];
    if constexpr (!XMEM) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = 4 * s + g;
            const bool in = 4 * s + 3 < D || k < D;              // only the last slot can fall into the pad
            xs[s] = in ? mhx_ld_off(a.x + (long)(4 * s) * ld, lane_off) : MHX_R(0.0);
            if (PK == MHX_PROP_DIAG) sc[s] = in ? pvec[k] : MHX_R(0.0);
        }
    }
    // slot s of the state / of the diagonal scales: a register, or (XMEM) a read of the slab / the scale vector
    // (through a buffer descriptor: SGPR row offset + the lane offset -- 64-bit vector addresses per slot would be hoisted out of
    // the step loop and cost two registers each)
    const mhx_srd xsrd = mhx_make_srd(a.x, (mhx_u32)D * ldb);
    auto xat = [&](const int s) -> mhx_real {
        if constexpr (XMEM) return (4 * s + 3 < D || 4 * s + g < D) ? mhx_srd_load(xsrd, lane_off, (mhx_u32)(4 * s) * ldb) : MHX_R(0.0);
        else return xs[s];
    };
    auto scat = [&](const int s) -> mhx_real {
        if constexpr (XMEM) return (4 * s + 3 < D || 4 * s + g < D) ? pvec[4 * s + g] : MHX_R(0.0);
        else return sc[s];
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
        // ---- proposal noise: lane g draws the blocks 4qd + g, the transpose leaves z of dimension 4(4qd + e) + g in n[e]
        mhx_real ys[NS];
#pragma unroll
        for (int qd = 0; qd < NQD; ++qd) {
            mhx_real n[4];
            mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)(4 * qd + g), n);
            mhx_lanes4_transpose(n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int s = 4 * qd + e;
                if (s < NS) {                                                              // src/proposal.jl:49-56
                    const bool in = 4 * s + 3 < D || 4 * s + g < D;
                    if (DENSEP) ys[s] = in ? n[e] : MHX_R(0.0);
                    else if (PK == MHX_PROP_DIAG) ys[s] = mhx_fma(scat(s), n[e], xat(s));
                    else ys[s] = in ? mhx_fma(a.pscale, n[e], xat(s)) : MHX_R(0.0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);        // one block of normals in flight at a time (register pressure)
        }
        if (DENSEP) {
            // xi = L z by rows (ascending j, fma from 0); row 16t + 4r + g = dimension 4(4t + r) + g: this lane's
            mhx_real xi[NS], unused = MHX_R(0.0);
            if constexpr (STREAM) mhx_mfma_rows_stream<D, 2, PAIR>(sL, CORR ? sA : sL, ring, lane, ys, unused, xi, pf, parity);
            else mhx_mfma_rows<D, 2>(Limg, lane, ys, unused, xi);
#pragma unroll
            for (int s = 0; s < NS; ++s) ys[s] = (4 * s + 3 < D || 4 * s + g < D) ? xat(s) + xi[s] : MHX_R(0.0);
        }
        // ---- lp': dense Gaussian -1/2 |A y|^2 + const, rows g, g+4, ... by this lane; butterfly over the chain's lanes
        mhx_real q = MHX_R(0.0);
        if (CORR) {
            mhx_real unused[NS];
            if constexpr (STREAM) mhx_mfma_rows_stream<D, 1, PAIR>(sA, DENSEP ? sL : sA, ring, lane, ys, q, unused, pf, parity);
            else mhx_mfma_rows<D, 1>(Aimg, lane, ys, q, unused);
        } else {
            // isotropic target in the reduction shape L = 4 of the cooperative kernels: lane g owns the BLOCKS g, g+4, ...
            // -- transpose the candidate back to blocks
#pragma unroll
            for (int qd = 0; qd < NQD; ++qd) {
                mhx_real n[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) n[e] = 4 * qd + e < NS ? ys[4 * qd + e] : MHX_R(0.0);
                mhx_lanes4_transpose(n);                       // n[e] = y of dimension 4(4qd + g) + e
#pragma unroll
                for (int e = 0; e < 4; ++e) q = mhx_fma(n[e], n[e], q);      // the pad is zero
            }
        }
        q = mhx_butterfly_add<16>(q);
        q = mhx_butterfly_add<32>(q);
        const mhx_real lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        // ---- accept (src/mh-core.jl:104-114); a zero-mean random walk has no Hastings term
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < (lpy - lp);
        if constexpr (XMEM) {
            if (acc && valid) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (4 * s + 3 < D || 4 * s + g < D) mhx_srd_store(xsrd, lane_off, (mhx_u32)(4 * s) * ldb, ys[s]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the slab is current before anything reads it again
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) xs[s] = acc ? ys[s] : xs[s];
        }
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && g == 0));
        if (step == save_next) {                                         // wave-uniform
            if (valid) {
                mhx_real* slotp = a.samples + slot * (long)(D + 1) * ld;
                const mhx_srd srd = mhx_make_srd(slotp, (mhx_u32)(D + 1) * (mhx_u32)ld * MHX_RB);
                mhx_u32 roff = 0u;                               // (running row offset behind an opaque asm: MHX_COOP_REC_RUN, mhx_rwmh_kernels.h)
                asm volatile("" : "+s"(roff));
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (4 * s + 3 < D || 4 * s + g < D) mhx_srd_store<MHX_REC_STORE_AUX>(srd, lane_off, roff, xat(s));
                    roff += 4u * ldb;
                    if (XMEM && (s & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // a few re-reads of the slab in flight, not all
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
        if constexpr (!XMEM) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (4 * s + 3 < D || 4 * s + g < D) mhx_st_off(a.x + (long)(4 * s) * ld, lane_off, xs[s]);
        }
        if (g == 0) {
            a.lp[c] = lp;
            a.acc_count[c] = nacc;
            a.last_acc[c] = last ? 1 : 0;
        }
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

#ifdef MHX_JIT_RWMH_MFMA
// dynamic LDS: [target image][proposal image]
#ifndef MHX_JIT_WAVES
#define MHX_JIT_WAVES 1
#endif
extern "C" __global__ void __launch_bounds__(64 * MHX_MFMA_WAVES, MHX_JIT_WAVES)
mhx_jit_rwmh_mfma(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    typedef mhx_mfma_geom<MHX_JIT_DIM> GEO;
    extern __shared__ mhx_acc4 mhx_mfma_lds[];
    mhx_real* Aimg = (mhx_real*)mhx_mfma_lds;
    mhx_real* Limg = Aimg + (MHX_JIT_TK == MHX_TARGET_CORR_GAUSS ? GEO::REALS : 0);
    mhx_rwmh_mfma_body<MHX_JIT_DIM, MHX_JIT_PK, MHX_JIT_TK>(a, tparams, pvec, Aimg, Limg);
}
#endif
#ifdef MHX_JIT_RWMH_MFMA_STREAM
#ifndef MHX_JIT_WAVES
#define MHX_JIT_WAVES 1
#endif
// the image builder (one block, once per run) and the streamed kernel: dynamic LDS = the two-buffer ring
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_mfma_image(const mhx_real* __restrict__ packed, mhx_real* img)
{
    mhx_mfma_image_fill<MHX_JIT_DIM>(packed, img);
}
extern "C" __global__ void __launch_bounds__(64 * MHX_MFMA_WAVES, MHX_JIT_WAVES)
mhx_jit_rwmh_mfma_stream(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec,
                         mhx_real* gAimg, mhx_real* gLimg)
{
    extern __shared__ mhx_acc4 mhx_mfma_ring[];
#ifndef MHX_JIT_PAIR
#define MHX_JIT_PAIR 1
#endif
#ifndef MHX_JIT_XMEM
#define MHX_JIT_XMEM 0
#endif
    mhx_rwmh_mfma_body<MHX_JIT_DIM, MHX_JIT_PK, MHX_JIT_TK, true, MHX_JIT_PAIR != 0, MHX_JIT_XMEM != 0>(
        a, tparams, pvec, gAimg, gLimg, (mhx_real*)mhx_mfma_ring);
}
#endif
MHX_NS_END
