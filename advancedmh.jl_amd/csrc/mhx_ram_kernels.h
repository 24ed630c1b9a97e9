// mhx_ram_kernels.h -- Robust Adaptive Metropolis (Vihola 2012), a GROUP of G lanes per chain.
//
// Replaces ram_step_inner (src/RobustAdaptiveMetropolis.jl:123-151), ram_adapt (:153-173),
// valid_eigenvalues (:239-245) and the step / step_warmup methods (:216-278).
//
// Every chain owns a lower-triangular factor S (d(d+1)/2 floats: 80 KB at d = 200), so a lane
// cannot own a chain.  A group of G lanes (16, 32 or 64; 64/G chains per wavefront = per block) owns
// one: lane tg of the group holds rows tg, tg+G, ... of x, U, v, w.  The kernel is bound by
// instruction issue, and what a column costs besides its elements -- the rotation (a division, a
// square root, a reciprocal), loop control, addressing -- is paid once per 64/G chains.  S is stored
// per chain as a PACKED COLUMN-MAJOR lower triangle (one contiguous array) and is streamed through a
// two-chunk LDS ring in full-width 16-byte loads; both passes over it are column sweeps:
//   pass A  x' = S U + x      v_j += S_ji U_i, columns i ascending  == row-dot in ascending order
//   pass B  rank-1 sweep      for column i: (s, c) from (S_ii, w_i), then every row j > i independently
// The sweep is the sign-unified textbook form (sigma = +1 update, -1 downdate), so
// chains that update and chains that downdate share one instruction stream.  The new factor goes to
// the chain's SECOND buffer and a per-chain selector flips only if it is valid (downdate stayed
// positive definite, diagonal inside the eigenvalue bounds -- RAM.jl:259-264 keeps the old S
// otherwise).  The NEXT step's mat-vec is fused into the sweep (its noise is recomputable from the
// counter RNG): an adapting step moves 1 read + 1 write of S, a fixed-S step 1 read.
#pragma once
#include "mhx_targets.h"

MHX_NS_BEGIN

struct mhx_ram_args {
    mhx_real* x;                 // [dim][ld]  (ABI layout; touched once per launch)
    mhx_real* lp;                // [ld]
    mhx_u32* acc_count;
    mhx_u64* acc_total;
    mhx_real* samples;           // [slots][dim+1][ld] or null
    unsigned char* accepted;
    unsigned char* last_acc;
    mhx_real* S;                 // [nchains][2][tri_pad]: both buffers of a chain side by side, packed column-major lower
    unsigned char* sel;       // [nchains] which buffer is current
    unsigned char* status;    // [nchains] bit0: a downdate left the PD cone, bit1: NaN log-ratio
    mhx_real* dmin;              // [nchains][dim] running min of diag(S)
    mhx_real* dmax;
    mhx_real* loga;              // [nchains] log acceptance ratio min(lp' - lp, 0) of each chain's latest transition
                              // (RobustAdaptiveMetropolisState.logα, RAM.jl:99-114, :141-147)
    mhx_real* rec_loga;          // [slots][ld] or null: the log acceptance ratio of every RECORDED transition (the state.logα a
                              // callback would have seen after that step, test/RobustAdaptiveMetropolis.jl:11-28)
    const mhx_real* eta;         // [nsteps] adaptation step sizes iteration^-gamma of this launch
    const mhx_real* acol;        // CORR_GAUSS target: inv(chol(Sigma)) packed column-major lower
    mhx_real* defer;             // deferred-factor form only: [nchains][MHX_RAM_DEFER_REALS] pending updates of each chain
    int prof;                    // tools build, deferred-factor form: block 0 prints its cycles per phase
    mhx_u64 seed;
    mhx_u64 first_chain;
    int nchains;
    int ld;
    int dim;
    int target_kind;
    int ntparams;
    mhx_real tconst;
    mhx_real alpha;
    mhx_real eig_lo, eig_hi;
    int default_bounds;
    mhx_u32 step0;
    int nsteps;
    int n_adapt;              // the first n_adapt steps of this launch adapt S (step_warmup)
    mhx_u32 save_next;
    int save_slot;
    int thinning;
};

// x accessor over LDS (broadcast reads: every lane of a group evaluates the target redundantly)
struct mhx_lds_x {
    const mhx_real* lds;      // wave-uniform base
    int off;               // per lane: mhx_real offset of its chain's vector
    MHX_DEV mhx_real operator[](int k) const { return lds[off + k]; }
};

MHX_DEV long mhx_ram_col_off(int i, int d) { return (long)i * d - ((long)i * (i - 1)) / 2; }

// floats per factor buffer: the triangle, 16-byte aligned, plus slack.  The sweep stores whole row
// slots: rows >= d of a column land on the cells that FOLLOW the column -- later columns, rewritten
// afterwards in program order, or this slack behind the last one.
#define MHX_RAM_SLACK 256
MHX_HD long mhx_ram_tri_pad(int d) { return ((((long)d * (d + 1)) / 2 + 3) & ~3L) + MHX_RAM_SLACK; }

typedef mhx_real mhx_f4 __attribute__((ext_vector_type(4)));

// LDS plan of one wave (= one block), in floats:
//   [ chain 0: ring | mirror ][ chain 1: ring | mirror ] ... [ chain 0: vector ][ chain 1: vector ] ...
// ring = 2 chunks of CHF floats, a chunk = NV x (G lanes x 16 B) >= the longest column; the mirror
// repeats the first d floats (rounded up to 4) of the ring behind it, so every column is CONTIGUOUS in LDS even when
// it runs off the end of the ring: a column read is one per-column address plus compile-time offsets (row slots past
// the column read whatever follows -- the vectors, still inside the allocation -- and nothing uses them).
// ONE d-vector per chain serves, in turn, as the noise U of a step that has to form S U itself, the candidate the
// target reads, the target's scratch and the NEXT step's noise during the sweep: each dies before the next is written
// (LDS, not registers, bounds the resident waves: 11.4 KB per wave at d = 200 in fp64, 15 KB with three vectors).
// The factor of a dense Gaussian target is the same for every chain: the wave streams it ONCE through
// the same LDS seen as a single ring (64 lanes x 16 B pieces) and all its chains read it from there.
#ifndef MHX_RAM_NV4_R
#define MHX_RAM_NV4_R 99  // tuning knob (fp64): rows per lane from which a chunk is 4 pieces deep instead of 2. Measured at C4 (R = 4):
                         // 8 KB in flight per wave but 22.8 KB of LDS (7 waves per CU instead of 8) -> 1.12e7 against 1.29e7 steps/s: off
#endif
#define MHX_RAM_NV(R) ((R) <= 8 ? ((MHX_REAL64 && (R) >= MHX_RAM_NV4_R) ? 4 : 2) : 4)
#define MHX_RAM_CHF(G, R) (MHX_RAM_NV(R) * (G) * 4)
#define MHX_RAM_RING(G, R) (2 * MHX_RAM_CHF(G, R))
#define MHX_RAM_MIRF(G, R) ((G) * (R))      // upper bound of the mirror (compile time); mhx_ram_mirror(d) reals are kept
MHX_HD int mhx_ram_mirror(int d) { return (d + 3) & ~3; }
MHX_HD int mhx_ram_rings(int ring_f, int d) { return ring_f + mhx_ram_mirror(d); }
#define MHX_RAM_LDS_FLOATS(G, R, d) ((64 / (G)) * (mhx_ram_rings(MHX_RAM_RING(G, R), d) + mhx_ram_mirror(d)))


// Streaming a packed factor: GS lanes pull one contiguous array in full-width pieces -- NV x (GS lanes
// x 16 B) = one chunk per round, every lane active -- and park it in a two-chunk LDS ring; the column
// logic reads its row segments from LDS while the next chunk's loads are already in flight in VGPRs.
// `require(end)` commits chunks until the linear range [0, end) has arrived; a column is at most one
// chunk long, so the chunk being overwritten is always dead.  All chains of a wave see the same
// column / chunk boundaries (same d): the control flow is wave-uniform.
template <int GS, int NV, int MIRF>
struct mhx_ram_stream {
    static constexpr int CHF = NV * GS * 4;
    static constexpr int RINGF = 2 * CHF;
    mhx_srd srd;                       // the factors of the wave's chains (or the shared one)
    mhx_u32 vsrc;                      // per lane: byte offset of the first 16 bytes it loads
    mhx_f4* lds4;                      // wave-uniform LDS base
    int r4;                            // per lane: float4 index of its slot in the ring of its chain
    int tri, avail;
    int committed, nchunks;
    mhx_f4 regs[NV];

    // reads past the triangle hit the slack / the next buffer (harmless) or the descriptor's range check (zeros)
    // a piece = 4 reals per lane: one 16-byte load in fp32, two in fp64
    MHX_DEV void load(const int k)
    {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#if MHX_REAL64
            typedef double mhx_d2 __attribute__((ext_vector_type(2)));
            const mhx_d2 p0 = __builtin_bit_cast(mhx_d2, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vsrc, (k * NV + v) * (GS * 32), 0));
            const mhx_d2 p1 = __builtin_bit_cast(mhx_d2, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vsrc, (k * NV + v) * (GS * 32) + 16, 0));
            mhx_f4 t;
            t.x = p0.x; t.y = p0.y; t.z = p1.x; t.w = p1.y;
            regs[v] = t;
#else
            regs[v] = __builtin_bit_cast(mhx_f4, __builtin_amdgcn_raw_buffer_load_b128(srd, (int)vsrc, (k * NV + v) * (GS * 16), 0));
#endif
        }
    }
    MHX_DEV void commit()
    {
        const int sb = (committed & 1) * (NV * GS);
#pragma unroll
        for (int v = 0; v < NV; ++v) lds4[r4 + sb + v * GS] = regs[v];
        if (!(committed & 1)) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (4 * v * GS < MIRF && mirror_lane(v)) lds4[r4 + 2 * NV * GS + v * GS] = regs[v];
        }
        ++committed;
        avail = committed * CHF < tri ? committed * CHF : tri;
        if (committed < nchunks) load(committed);
    }
    bool mlane[NV];
    MHX_DEV bool mirror_lane(const int v) const { return mlane[v]; }
    // ring_f: mhx_real offset of the ring in LDS; tg: this lane's index among the GS lanes of the stream
    MHX_DEV void begin(const mhx_srd srd_, const mhx_u32 chain_byte_off, const int d, const int tg, mhx_real* lds, const int ring_f)
    {
        srd = srd_;
        vsrc = chain_byte_off + (4u * MHX_RB) * (mhx_u32)tg;
        lds4 = (mhx_f4*)lds;
        r4 = (ring_f >> 2) + tg;
#pragma unroll
        for (int v = 0; v < NV; ++v) mlane[v] = 4 * (v * GS + tg) < mhx_ram_mirror(d);
        tri = d * (d + 1) / 2;
        nchunks = (tri + CHF - 1) / CHF;
        committed = 0;
        avail = 0;
        load(0);
        MHX_WAVE_SYNC();                                   // the previous pass is done with the ring
        commit();
        MHX_WAVE_SYNC();
    }
    MHX_DEV void require(const int end)
    {
        while (end > avail) {                              // wave-uniform
            MHX_WAVE_SYNC();
            commit();
            MHX_WAVE_SYNC();
        }
    }
};

// columns G IR + il in ascending order; the row slot IR that holds the diagonal is a compile-time
// constant inside the functor, so slots above the diagonal vanish at compile time and only slot IR
// carries masks.
template <int G, int R, int IR>
struct mhx_ram_cols {
    template <class ST, class F>
    static MHX_DEV void run(ST& st, const int d, int off, F& f)
    {
        for (int il = 0; il < G; ++il) {
            const int i = G * IR + il;
            if (i >= d) return;
            st.require(off + (d - i));
            f.template col<IR>(il, i, off & (ST::RINGF - 1), off);
            off += d - i;
        }
        mhx_ram_cols<G, R, IR + 1>::run(st, d, off, f);
    }
};
template <int G, int R>
struct mhx_ram_cols<G, R, R> {
    template <class ST, class F>
    static MHX_DEV void run(ST&, const int, int, F&) {}
};

// v = S u: column sweep (columns ascending == the row-dot's ascending j order).  Rows >= d of the last
// slot accumulate whatever follows the column in the ring; nothing ever reads them.
template <int G, int R>
struct mhx_ram_matvec_f {
    const mhx_real* lds;      // wave-uniform
    int ring;              // per lane: mhx_real offset of the ring the column is read from, plus tg
    int ush;               // per lane: mhx_real offset of the vector of its chain
    int tg;
    mhx_real v[R];
    template <int IR>
    MHX_DEV void col(const int il, const int i, const int roff, const int)
    {
        const mhx_real* cp = lds + (ring + (roff - il));                   // row tg + G IR of this column
        const mhx_real ui = lds[ush + i];
        const mhx_real c0 = cp[0];
        v[IR] = mhx_fma(tg >= il ? c0 : MHX_R(0.0), ui, v[IR]);               // rows above the diagonal: fma(0, u, v) == v
#pragma unroll
        for (int r = IR + 1; r < R; ++r) v[r] = mhx_fma(cp[G * (r - IR)], ui, v[r]);
    }
};
// SHARED = the factor is the same for all chains of the wave (streamed once by all 64 lanes)
template <int G, int R, bool SHARED>
MHX_DEV void mhx_ram_matvec(const mhx_srd srd, const mhx_u32 chain_byte_off, const int ush, const int d, const int lane,
                            mhx_real* lds, mhx_real (&v)[R])
{
    constexpr int GS = SHARED ? 64 : G;
    const int g = lane / G, tg = lane % G;
    const int ring = SHARED ? 0 : g * mhx_ram_rings(MHX_RAM_RING(G, R), d);
    mhx_ram_matvec_f<G, R> f;
    f.lds = lds; f.ring = ring + tg; f.ush = ush; f.tg = tg;
#pragma unroll
    for (int r = 0; r < R; ++r) f.v[r] = MHX_R(0.0);
    mhx_ram_stream<GS, MHX_RAM_NV(R), MHX_RAM_MIRF(G, R)> st;
    st.begin(srd, chain_byte_off, d, SHARED ? lane : tg, lds, ring);
    mhx_ram_cols<G, R, 0>::run(st, d, 0, f);
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = f.v[r];
}

// draw U = randn(d) of `step` into the chain's LDS slot (lane tg draws Philox blocks tg, tg+G, ...) and
// return |U|^2 (ascending order, every lane of the group)
template <int G>
MHX_DEV mhx_real mhx_ram_draw(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step, const int d,
                           const int tg, mhx_real* lds, const int uoff)
{
    mhx_real* ush = lds + uoff;
    const int nblk = (d + 3) >> 2;
    for (int b0 = 0; b0 < nblk; b0 += G) {                 // wave-uniform trip count
        const int b = b0 + tg;
        mhx_real n[4];
        mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * b + j < d) ush[4 * b + j] = n[j];
    }
    MHX_WAVE_SYNC();
    mhx_real nn = MHX_R(0.0);
    for (int j = 0; j < d; ++j) { const mhx_real u = ush[j]; nn = mhx_fma(u, u, nn); }
    return nn;
}

// value of x in lane il of the caller's own group
template <int G>
MHX_DEV mhx_real mhx_ram_group_bcast(const mhx_real x, const int il, const int g)
{
    mhx_real out = mhx_readlane(x, il);
#pragma unroll
    for (int k = 1; k < 64 / G; ++k) {
        const mhx_real o = mhx_readlane(x, k * G + il);
        out = g == k ? o : out;
    }
    return out;
}

// The sign-unified rank-1 sweep (sg = +1 update, -1 downdate per chain), one column per
// call, written to the chain's other buffer through one buffer descriptor per wave; the NEXT step's
// proposal mat-vec is accumulated for both the old and the new factor on the way.
// Invariant: w is exactly 0 in rows above the current column, so those rows need no masks:
// fma(ss, 0, 0) * rc == 0 leaves their accumulators alone.  A chain that does not adapt in this step, or
// whose downdate has left the positive-definite cone, carries w == 0 everywhere: its rotation is the
// identity and the sweep copies its columns (the selector does not flip, so nobody reads the copy).
template <int G, int R>
struct mhx_ram_sweep_f {
    const mhx_real* lds;      // wave-uniform
    int ring;              // per lane: mhx_real offset of the ring of its chain
    int unext;             // per lane: mhx_real offset of next step's noise of its chain
    mhx_srd srd;           // both buffers of every chain of the wave
    mhx_u32 vbase;         // per lane: byte offset of row tg of the chain's new buffer (out of range: idle group)
    int tg, g;
    mhx_real sg;              // per chain
    bool ok;               // per chain: the downdate is still positive definite
    mhx_real w[R];            // the rank-1 vector, rotated column by column
    mhx_real nd[R];           // new diagonal entries of the rows this lane owns
    mhx_real vo[R];           // next step's S_old U'   (fused mat-vec)
    mhx_real vn[R];           // next step's S_new U'
    template <int IR>
    MHX_DEV void col(const int il, const int i, const int roff, const int off)
    {
        const mhx_real* cp = lds + (ring + tg + (roff - il));
        const bool lo = tg >= il;                                      // row >= i inside slot IR
        const bool ondiag = tg == il;
        const mhx_real c0 = lo ? cp[0] : MHX_R(0.0);
        const mhx_real aii = lds[ring + roff];                            // the diagonal entry (broadcast read)
        const mhx_real bi = mhx_ram_group_bcast<G>(w[IR], il, g);
        mhx_real sn = bi / aii;
        const bool bad = sg < MHX_R(0.0) && sn * sn > MHX_R(1.0);                  // PosDefException upstream
        if (__ballot(bad) != 0ull) {                                   // rare: that chain coasts from here on
            ok = ok && !bad;
            sn = bad ? MHX_R(0.0) : sn;
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = bad ? MHX_R(0.0) : w[r];
        }
        const mhx_real ss = sg * sn;
        const mhx_real cs = mhx_sqrt(mhx_fma(ss, sn, MHX_R(1.0)));
        const mhx_real rcs = MHX_R(1.0) / cs;                                   // one reciprocal per column
        const mhx_real diag = cs * aii;
        const mhx_real un = lds[unext + i];
        const mhx_u32 voff = vbase + MHX_RB * (mhx_u32)(off - i);          // row tg of column i
        {
            const mhx_real vj = w[IR];
            const mhx_real oe = mhx_fma(ss, vj, c0) * rcs;
            const mhx_real wn = mhx_fma(cs, vj, -(sn * oe));
            const mhx_real out = ondiag ? diag : oe;
            w[IR] = ondiag ? MHX_R(0.0) : wn;
            nd[IR] = ondiag ? diag : nd[IR];
            if (lo) mhx_srd_store(srd, voff + MHX_RB * G * IR, 0u, out);
            vo[IR] = mhx_fma(c0, un, vo[IR]);
            vn[IR] = mhx_fma(out, un, vn[IR]);
        }
#pragma unroll
        for (int r = IR + 1; r < R; ++r) {
            const mhx_real Aji = cp[G * (r - IR)], vj = w[r];
            const mhx_real oe = mhx_fma(ss, vj, Aji) * rcs;
            w[r] = mhx_fma(cs, vj, -(sn * oe));
            mhx_srd_store(srd, voff + MHX_RB * G * r, 0u, oe);
            vo[r] = mhx_fma(Aji, un, vo[r]);
            vn[r] = mhx_fma(oe, un, vn[r]);
        }
    }
};

// G = lanes per chain, R = rows per lane (dim <= G R)
template <int G, int R, int TK>
MHX_DEV void mhx_ram_body(const mhx_ram_args& a, const mhx_real* __restrict__ tparams, mhx_real* lds)
{
    constexpr int CPW = 64 / G;                  // chains per wave (= per block)
    const int RINGS = mhx_ram_rings(MHX_RAM_RING(G, R), a.dim);
    // XCD-aware mapping: blocks b, b+8, ... run on one XCD; give them consecutive chain groups
    const int nb = gridDim.x;
    const int per = (nb + 7) >> 3;
    const long c0 = ((long)(blockIdx.x & 7u) * per + (long)(blockIdx.x >> 3)) * CPW;
    if (c0 >= a.nchains) return;                 // whole wave past the end
    const int lane = threadIdx.x;
    const int g = lane / G, tg = lane % G;
    const bool valid = c0 + g < a.nchains;
    const long c = valid ? c0 + g : (long)a.nchains - 1;      // idle groups shadow the last chain (no stores)
    const int d = a.dim;
    const long ld = a.ld;
    const long tri_pad = mhx_ram_tri_pad(d);
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    // per-lane LDS mhx_real offsets (the LDS base stays wave-uniform)
    const int ring = g * RINGS;                                         // 16-byte aligned
    const int ucur = CPW * RINGS + g * mhx_ram_mirror(d);   // [d] noise of a step that forms S U itself (dead after that mat-vec), then target scratch
    const int unxt = ucur;                // [d] noise of the next step (fused mat-vec): drawn after the target is done with ysh / ucur
    const int ysh = ucur;                 // [d] candidate (written after S U has consumed the noise)
    // both buffers of every chain of this wave through one descriptor; idle groups read chain c0
    const mhx_srd srd = mhx_make_srd(a.S + c0 * 2 * tri_pad, (mhx_u32)(CPW * 2 * tri_pad * MHX_RB));
    const mhx_srd srd_a = mhx_make_srd(a.acol, (mhx_u32)(tri_pad * MHX_RB));
    const int gv = valid ? g : 0;

    mhx_real x[R], dmn[R], dmx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = tg + G * r;
        x[r] = row < d ? a.x[(long)row * ld + c] : MHX_R(0.0);
        dmn[r] = row < d ? a.dmin[(long)c * d + row] : MHX_R(0.0);
        dmx[r] = row < d ? a.dmax[(long)c * d + row] : MHX_R(0.0);
    }
    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    int sel = a.sel[c];
    unsigned st = a.status[c];
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    mhx_real loga_last = a.loga[c];
    bool have_v = false;         // wave-uniform: v = S U of this step (and nn) came out of the previous sweep
    mhx_real v[R], nn = MHX_R(0.0);
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = MHX_R(0.0);

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        const mhx_u32 scur = (mhx_u32)((2 * gv + sel) * tri_pad * MHX_RB);      // byte offset of the current factor

        // ---- U = randn(d), v = S U, x' = v + x   (RAM.jl:135-136)
        if (!have_v) {
            MHX_WAVE_SYNC();
            nn = mhx_ram_draw<G>(ks, id_lo, id_hi, step, d, tg, lds, ucur);
            mhx_ram_matvec<G, R, false>(srd, scur, ucur, d, lane, lds, v);
        }
        mhx_real y[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = tg + G * r;
            y[r] = v[r] + x[r];
            if (row < d) lds[ysh + row] = y[r];
        }
        MHX_WAVE_SYNC();

        // ---- lp' = logdensity(x')  (RAM.jl:140)
        mhx_real lpy;
        const int kind = (TK == MHX_TARGET_DYNAMIC) ? a.target_kind : TK;
        if (kind == MHX_TARGET_CORR_GAUSS) {
            // cooperative column sweep over A = inv(chol(Sigma)): w_j += A_ji y_i, i ascending
            mhx_real wv[R];
            mhx_ram_matvec<G, R, true>(srd_a, 0u, ysh, d, lane, lds, wv);
            MHX_WAVE_SYNC();
#pragma unroll
            for (int r = 0; r < R; ++r) if (tg + G * r < d) lds[ucur + tg + G * r] = wv[r];
            MHX_WAVE_SYNC();
            mhx_real q = MHX_R(0.0);
            for (int j = 0; j < d; ++j) { const mhx_real w = lds[ucur + j]; q = mhx_fma(w, w, q); }
            lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        } else {
            mhx_lds_x yv;
            yv.lds = lds; yv.off = ysh;
            lpy = mhx_target_eval<TK>(kind, yv, d, tparams, a.ntparams, a.tconst);
        }

        // ---- accept (RAM.jl:147-148): loga = min(lp' - lp, 0); accept iff randexp > -loga
        const mhx_real diff = lpy - lp;
        const mhx_real loga = (diff != diff) ? diff : (diff < MHX_R(0.0) ? diff : MHX_R(0.0));
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;
        loga_last = loga;

        // ---- adapt (RAM.jl:153-173, :259-264) during warm-up; the step index is wave-uniform
        have_v = false;
        if (it < a.n_adapt) {
            const mhx_real da = mhx_exp(loga) - a.alpha;                    // :159
            const bool adapt = da == da;                                 // a NaN log-ratio skips the adaptation
            if (!adapt) st |= 2u;
            const mhx_real eta = a.eta[it];                                 // :162 iteration^-gamma
            const mhx_real coef = mhx_sqrt(eta * mhx_abs(da)) / mhx_sqrt(nn);   // :163
            const bool fuse = it + 1 < a.nsteps;                         // a next step exists in this launch
            mhx_ram_sweep_f<G, R> sw;
            sw.lds = lds; sw.ring = ring; sw.unext = unxt; sw.srd = srd; sw.tg = tg; sw.g = g;
            sw.vbase = valid ? (mhx_u32)(((2 * g + (sel ^ 1)) * tri_pad + tg) * MHX_RB) : 0x80000000u;
            sw.sg = da > MHX_R(0.0) ? MHX_R(1.0) : -MHX_R(1.0);                            // :165 sign(da) == 1 ? update : downdate
            sw.ok = true;
#pragma unroll
            for (int r = 0; r < R; ++r) { sw.w[r] = adapt ? v[r] * coef : MHX_R(0.0); sw.nd[r] = MHX_R(0.0); sw.vo[r] = MHX_R(0.0); sw.vn[r] = MHX_R(0.0); }
            mhx_real nn_next = MHX_R(0.0);
            MHX_WAVE_SYNC();                                             // every lane is done with ucur / ysh
            if (fuse) nn_next = mhx_ram_draw<G>(ks, id_lo, id_hi, step + 1u, d, tg, lds, unxt);
            mhx_ram_stream<G, MHX_RAM_NV(R), MHX_RAM_MIRF(G, R)> stream;
            stream.begin(srd, scur, d, tg, lds, ring);
            mhx_ram_cols<G, R, 0>::run(stream, d, 0, sw);
            if (adapt && !sw.ok) st |= 1u;
            bool ok = adapt && sw.ok;
            // valid_eigenvalues (RAM.jl:239-245): every diagonal entry of the chain inside [lo, hi]
            if (!a.default_bounds) {
                bool bad = false;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (tg + G * r < d && !(a.eig_lo <= sw.nd[r] && sw.nd[r] <= a.eig_hi)) bad = true;
                const mhx_u64 bm = __ballot(bad);
                const mhx_u64 gm = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << (g * G));
                if (bm & gm) ok = false;
            }
            if (ok) {
                sel ^= 1;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    dmn[r] = sw.nd[r] < dmn[r] ? sw.nd[r] : dmn[r];
                    dmx[r] = sw.nd[r] > dmx[r] ? sw.nd[r] : dmx[r];
                }
            }
            if (fuse) {
                // the next step's mat-vec is done: S_{t+1} = the new factor if it was kept, else the old one
                // (vo only reads the old columns: exact whatever happened to the sweep)
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = ok ? sw.vn[r] : sw.vo[r];
                nn = nn_next;
                have_v = true;
            }
        }

        // ---- state select (RAM.jl:267-277)
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = acc ? y[r] : x[r];
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += (mhx_u32)__popcll(__ballot(acc && valid && tg == 0));
        if (step == save_next) {
            if (valid) {
                mhx_real* rowp = a.samples + slot * (long)(d + 1) * ld + c;
#pragma unroll
                for (int r = 0; r < R; ++r) if (tg + G * r < d) rowp[(long)(tg + G * r) * ld] = x[r];
                if (tg == 0) {
                    rowp[(long)d * ld] = lp;
                    a.accepted[slot * ld + c] = acc ? 1 : 0;
                    if (a.rec_loga) a.rec_loga[slot * ld + c] = loga;
                }
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
        MHX_WAVE_SYNC();
    }
    if (valid) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = tg + G * r;
            if (row < d) {
                a.x[(long)row * ld + c] = x[r];
                a.dmin[(long)c * d + row] = dmn[r];
                a.dmax[(long)c * d + row] = dmx[r];
            }
        }
        if (tg == 0) {
            a.lp[c] = lp;
            a.acc_count[c] = nacc;
            a.last_acc[c] = last ? 1 : 0;
            a.sel[c] = (unsigned char)sel;
            a.status[c] = (unsigned char)st;
            a.loga[c] = loga_last;
        }
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
}

// initial state (RAM.jl:175-214): x0 = initial_params or randn(d); lp0; accepted = true (:213)
template <int TK>
MHX_DEV void mhx_ram_init_body(const mhx_ram_args& a, const mhx_real* __restrict__ tparams, const int draw)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const long ld = a.ld;
    const int d = a.dim;
    mhx_real* xs = a.x + c;
    if (draw) {
        const mhx_u64 id = a.first_chain + (mhx_u64)c;
        const mhx_philox_key ks = mhx_philox_schedule(a.seed);
        const int nblk = (d + 3) >> 2;
        for (int b = 0; b < nblk; ++b) {
            mhx_real n[4];
            mhx_normal4(ks, (mhx_u32)id, (mhx_u32)(id >> 32), 0u, MHX_STREAM_INIT, (mhx_u32)b, n);
            for (int j = 0; j < 4; ++j) if (4 * b + j < d) xs[(long)(4 * b + j) * ld] = n[j];
        }
    }
    mhx_strided_x xv;
    xv.base = xs;
    xv.ld = ld;
    a.lp[c] = mhx_target_eval<TK>(a.target_kind, xv, d, tparams, a.ntparams, a.tconst);
    a.acc_count[c] = 0u;
    a.last_acc[c] = 1;
    a.status[c] = 0;
    a.loga[c] = MHX_R(0.0);         // RAM.jl:211: RobustAdaptiveMetropolisState(x, lp, S, zero(T), 0, 1, true)
}

// ------------------------------------------------------------------------------------------------------------------
// The DEFERRED-FACTOR form (MHX_FLAG_RAM_DEFERRED; arithmetic spec DESIGN.md 3.12, which also names the CPU twin the tests compare it with).
// ram_adapt (RAM.jl:153-173) moves  S S' <- S S' + sigma w w',  w = c S U,  c^2 = eta |dalpha| / |U|^2:  S_new = S M  with
// M = chol(I + sigma c^2 U U'), a lower-triangular factor known from U and two scalars alone
//     M_jj = a_j = sqrt(T_{j+1} / T_j),   M_ij = U_i g_j (i > j),   g_j = sigma c^2 U_j / (T_j a_j),   T_j = 1 + sigma c^2 sum_{m<j} U_m^2.
// Up to K accepted updates stay PENDING as (a, g, u) triples in a per-chain scratch (L2 / MALL resident):
//     proposal   v = S_0 (M_1 (... (M_m U)))     ONE read of the stored factor, m prefix scans over d elements
//     flush      S_0 <- S_0 M_1 ... M_m           one read + one write per K steps -- columns ascending through the same LDS ring,
//                the K stages pipelined per column:  (S M)_rj = a_j S_rj + g_j (t_r - sum_{i<=j} S_ri u_i),  t = S u = the v of that
//                step's proposal; the next step's mat-vec rides on the new columns as in the sweep above.
// An adapting step moves (K + 1) / K reads + 1 / K writes of S instead of 1 + 1.  One chain per wave (G = 64), dim <= 256.
// Element layout of the O(d) vectors ("layout 4"): lane b owns elements 4b .. 4b+3 = Philox block b of the step's noise; prefix sums
// are a running fma inside the lane + a Kogge-Stone scan over the 64 lane totals.
#ifndef MHX_RAM_DEFER_K
#define MHX_RAM_DEFER_K 8
#endif
// per-chain scratch, in reals: [K slots][256 elements] (a, g, u, -) -- a slot is written in one coalesced piece (lane b: its four
// elements, 128 B), read back the same way by the prefix applications, and gathered K x 64/K columns at a time by the fold --
// then [K][R][64] the suffix vectors t
#define MHX_RAM_DEFER_CF(K) ((K) * 256 * 4)
#define MHX_RAM_DEFER_REALS(K, R) (MHX_RAM_DEFER_CF(K) + (K) * (R) * 64)
// LDS of the fold's coefficient staging: two groups of 64 (a, g, u, -) entries = 64 / K columns each, then what is kept of a pending
// update between its step and the fold: the step's number and the signed scale sigma c^2 (2 x 8 reals; round 6: the update's
// O(d) coefficients are REGENERATED from these two -- U from the Philox counter, T_j from its prefix sums -- instead of being
// written to the per-chain scratch at every step and read back at every proposal: 39 -> 17 KB of scratch traffic per step at d = 200)
#define MHX_RAM_DEFER_STAGE (512 + 16)

MHX_DEV mhx_real mhx_shfl_up(const mhx_real x, const int s) { return __shfl_up(x, s, 64); }

// exclusive prefix of p_j q_j at the lane's four elements, and the lane's inclusive total (== the exclusive prefix of element 4b+4)
MHX_DEV void mhx_scan_dot4(const mhx_real (&p)[4], const mhx_real (&q)[4], const int lane, mhx_real (&ex)[4], mhx_real& incl)
{
    const mhx_real r0 = p[0] * q[0];
    const mhx_real r1 = mhx_fma(p[1], q[1], r0);
    const mhx_real r2 = mhx_fma(p[2], q[2], r1);
    const mhx_real r3 = mhx_fma(p[3], q[3], r2);
    mhx_real t = r3;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const mhx_real o = mhx_shfl_up(t, sft);
        t = lane >= sft ? o + t : t;
    }
    mhx_real E = mhx_shfl_up(t, 1);
    E = lane ? E : MHX_R(0.0);
    incl = t;
    ex[0] = E; ex[1] = E + r0; ex[2] = E + r1; ex[3] = E + r2;
}

// the K-stage column pipeline of a flush; s[k] = the running suffix vector of stage k (rows this lane owns)
template <int R, int K>
struct mhx_ram_flush_f {
    const mhx_real* lds;      // wave-uniform
    int ring;                 // mhx_real offset of the ring
    int unext;                // mhx_real offset of next step's noise
    mhx_srd srd;
    mhx_u32 vbase;            // byte offset of row tg of the chain's new buffer
    int tg;
    int m;                    // pending updates (wave-uniform)
    const mhx_f4* coef;       // [K][256] (a, g, u, -) of this chain
    mhx_f4* stage;            // LDS, wave-uniform: [2][64] entries -- the coefficients of 64 / K columns per half, loaded a group ahead
    mhx_f4 pre;               // this lane's entry of the next group
    mhx_real s[K][R];
    mhx_real vn[R];           // next step's S_new U'
    static constexpr int CG = 64 / K;
    MHX_DEV void prefetch(const int i) { pre = coef[(tg / CG) * 256 + i + (tg & (CG - 1))]; }   // entry tg of the group: slot tg / CG, column i + tg % CG
    template <int IR>
    MHX_DEV void col(const int il, const int i, const int roff, const int off)
    {
        if ((il & (CG - 1)) == 0) {                                  // wave-uniform: a new group of columns
            // slots that hold no pending update are the identity (a = 1, g = u = 0: an exact no-op on the column and on s = 0)
            if (tg / CG >= m) { pre.x = MHX_R(1.0); pre.y = MHX_R(0.0); pre.z = MHX_R(0.0); }
            stage[((i / CG) & 1) * 64 + tg] = pre;
            if (i + CG < 256) prefetch(i + CG);
            MHX_WAVE_SYNC();
        }
        const mhx_real* cp = lds + (ring + tg + (roff - il));
        const bool lo = tg >= il;
        const bool ondiag = tg == il;
        mhx_real c[R];
        c[IR] = lo ? cp[0] : MHX_R(0.0);
#pragma unroll
        for (int r = IR + 1; r < R; ++r) c[r] = cp[64 * (r - IR)];
        const mhx_f4* cf = stage + (((i / CG) & 1) * 64 + (i & (CG - 1)));                 // slot k of this column: cf[k CG]
#pragma unroll
        for (int k = 0; k < K; ++k) {
            {
                const mhx_f4 q = cf[k * CG];
                const mhx_real ak = q.x, gk = q.y, nu = -q.z;
                {
                    const mhx_real ac = ak * c[IR];
                    s[k][IR] = mhx_fma(nu, c[IR], s[k][IR]);
                    const mhx_real t = mhx_fma(gk, s[k][IR], ac);
                    c[IR] = ondiag ? ac : t;
                }
#pragma unroll
                for (int r = IR + 1; r < R; ++r) {
                    const mhx_real ac = ak * c[r];
                    s[k][r] = mhx_fma(nu, c[r], s[k][r]);
                    c[r] = mhx_fma(gk, s[k][r], ac);
                }
            }
        }
        const mhx_real un = lds[unext + i];
        const mhx_u32 voff = vbase + MHX_RB * (mhx_u32)(off - i);
        if (lo) mhx_srd_store(srd, voff + MHX_RB * 64 * IR, 0u, c[IR]);
        vn[IR] = mhx_fma(lo ? c[IR] : MHX_R(0.0), un, vn[IR]);
#pragma unroll
        for (int r = IR + 1; r < R; ++r) {
            mhx_srd_store(srd, voff + MHX_RB * 64 * r, 0u, c[r]);
            vn[r] = mhx_fma(c[r], un, vn[r]);
        }
    }
};

// the noise of `step` in layout 4 (lane b: Philox block b), its prefix sums of squares, and a copy in the chain's LDS vector
struct mhx_ram_noise4 {
    mhx_real u[4], pu[4], incl, nn;
};
MHX_DEV void mhx_ram_draw4(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step, const int d, const int lane,
                           mhx_ram_noise4& o)
{
    const int nblk = (d + 3) >> 2;
    mhx_real n[4] = {MHX_R(0.0), MHX_R(0.0), MHX_R(0.0), MHX_R(0.0)};
    if (lane < nblk) mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)lane, n);
#pragma unroll
    for (int e = 0; e < 4; ++e) o.u[e] = 4 * lane + e < d ? n[e] : MHX_R(0.0);
    mhx_scan_dot4(o.u, o.u, lane, o.pu, o.incl);
    o.nn = mhx_readlane(o.incl, 63);
}

// The coefficients of a pending update at this lane's four elements, from the step that made it and its signed scale alone:
// U = the step's noise again (counter-based: the same bits), T_j = 1 + sc sum_{m<j} U_m^2 from the same scan, then
// a_j = sqrt(T_{j+1} / T_j), g_j = sc U_j / (T_j a_j) -- the very expressions of the adapt phase below, so the values are the ones
// that were stored until round 5.
MHX_DEV void mhx_ram_defer_regen(const mhx_philox_key& ks, const mhx_u32 id_lo, const mhx_u32 id_hi, const mhx_u32 step, const mhx_real sc,
                                 const int d, const int lane, mhx_real (&aa)[4], mhx_real (&gg)[4], mhx_real (&uu)[4])
{
    mhx_ram_noise4 q;
    mhx_ram_draw4(ks, id_lo, id_hi, step, d, lane, q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const mhx_real T0 = mhx_fma(sc, q.pu[e], MHX_R(1.0));
        const mhx_real T1 = mhx_fma(sc, e < 3 ? q.pu[(e + 1) & 3] : q.incl, MHX_R(1.0));
        const bool in = 4 * lane + e < d;
        const mhx_real ae = mhx_sqrt(T1 / T0);
        aa[e] = in ? ae : MHX_R(1.0);
        gg[e] = in ? (sc * q.u[e]) / (T0 * ae) : MHX_R(0.0);
        uu[e] = q.u[e];
    }
}

template <int R, int K, int TK>
#ifdef MHX_TOOLS_BUILD                                             // (tools build: block 0 prints its cycles per phase)
#define MHX_RAM_DEFER_PROF 1
#define MHX_PROF_T(k) do { const mhx_u64 now_ = __builtin_readcyclecounter(); prof_[k] += now_ - prof_t_; prof_t_ = now_; } while (0)
#else
#define MHX_PROF_T(k) do { } while (0)
#endif
MHX_DEV void mhx_ram_defer_body(const mhx_ram_args& a, const mhx_real* __restrict__ tparams, mhx_real* lds)
{
    constexpr int G = 64;
    const int RINGS = mhx_ram_rings(MHX_RAM_RING(G, R), a.dim);
    const int nb = gridDim.x;
    const int per = (nb + 7) >> 3;
    const long c = (long)(blockIdx.x & 7u) * per + (long)(blockIdx.x >> 3);
    if (c >= a.nchains) return;
    const int lane = threadIdx.x;
    const int tg = lane;
    const int d = a.dim;
    const long ld = a.ld;
    const long tri_pad = mhx_ram_tri_pad(d);
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    const int ring = 0;
    const int vec = RINGS;                       // the chain's one LDS vector: z, then the candidate / target scratch, then a, then U'
    const mhx_srd srd = mhx_make_srd(a.S + c * 2 * tri_pad, (mhx_u32)(2 * tri_pad * MHX_RB));
    const mhx_srd srd_a = mhx_make_srd(a.acol, (mhx_u32)(tri_pad * MHX_RB));
    mhx_real* scratch = a.defer + c * (long)MHX_RAM_DEFER_REALS(K, R);
    mhx_f4* coef = (mhx_f4*)scratch;                                   // [K][256] (a, g, u, -)
    mhx_real* svec = scratch + MHX_RAM_DEFER_CF(K);                    // [K][R][64]
    mhx_f4* stage = (mhx_f4*)(lds + RINGS + mhx_ram_mirror(d));
    mhx_real* const psc = lds + RINGS + mhx_ram_mirror(d) + 512;        // [K] signed scale of pending update k
    mhx_u32* const pstep = (mhx_u32*)(psc + 8);                         // [K] the step that made it
    const bool has_blk = lane < ((d + 3) >> 2);                        // this lane's four elements hold part of the vector
    const bool in_vec = 4 * lane < mhx_ram_mirror(d);                  // this lane's four elements lie inside the LDS vector

    int sel = a.sel[c];
    mhx_real x[R], dmn[R], dmx[R], dg[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = tg + G * r;
        x[r] = row < d ? a.x[(long)row * ld + c] : MHX_R(0.0);
        dmn[r] = row < d ? a.dmin[(long)c * d + row] : MHX_R(0.0);
        dmx[r] = row < d ? a.dmax[(long)c * d + row] : MHX_R(0.0);
        dg[r] = row < d ? a.S[(2 * c + sel) * tri_pad + mhx_ram_col_off(row, d)] : MHX_R(1.0);
    }
    mhx_real lp = a.lp[c];
    mhx_u32 nacc = a.acc_count[c];
    mhx_u32 wave_acc = 0;
    bool last = a.last_acc[c] != 0;
    unsigned st = a.status[c];
    mhx_accept_cache ac;
    ac.group = 0xffffffffu;
    ac.w.x = ac.w.y = ac.w.z = ac.w.w = 0u;
    mhx_u32 save_next = a.save_next;
    long slot = a.save_slot;
    mhx_real loga_last = a.loga[c];
    bool have_v = false;
    int m = 0;                                   // pending updates
    mhx_real v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = MHX_R(0.0);
    mhx_ram_noise4 nz;
#pragma unroll
    for (int e = 0; e < 4; ++e) nz.u[e] = nz.pu[e] = MHX_R(0.0);
    nz.incl = nz.nn = MHX_R(0.0);

#ifdef MHX_TOOLS_BUILD
    mhx_u64 prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mhx_u64 prof_t_ = __builtin_readcyclecounter();
#endif
    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        const mhx_u32 scur = (mhx_u32)(sel * tri_pad * MHX_RB);
        MHX_PROF_T(7);

        // ---- U = randn(d), z = M_1 (... (M_m U)), v = S_0 z, x' = v + x
        if (!have_v) {
            // the pending factors, newest first, each REGENERATED at this lane's four elements (mhx_ram_defer_regen): a Philox
            // block, a scan and a dozen slow operations per update -- a few hundred cycles of a step that waits ~10^5 on HBM
            mhx_ram_draw4(ks, id_lo, id_hi, step, d, lane, nz);
            mhx_real z[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = nz.u[e];
            for (int k = m - 1; k >= 0; --k) {                          // wave-uniform
                mhx_real aa[4], gg[4], uu[4];
                mhx_ram_defer_regen(ks, id_lo, id_hi, pstep[k], psc[k], d, lane, aa, gg, uu);
                mhx_real P[4], tot;
                mhx_scan_dot4(gg, z, lane, P, tot);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = mhx_fma(uu[e], P[e], aa[e] * z[e]);
            }
            MHX_WAVE_SYNC();
            if (in_vec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) lds[vec + 4 * lane + e] = z[e];
            }
            MHX_PROF_T(0);
            mhx_ram_matvec<G, R, false>(srd, scur, vec, d, lane, lds, v);
            MHX_PROF_T(1);
        }
        const mhx_real eta = it < a.n_adapt ? a.eta[it] : MHX_R(0.0);      // asked for a whole log-density early
        mhx_real y[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = tg + G * r;
            y[r] = v[r] + x[r];
            if (row < d) lds[vec + row] = y[r];
        }
        MHX_WAVE_SYNC();

        // ---- lp' = logdensity(x')
        mhx_real lpy;
        const int kind = (TK == MHX_TARGET_DYNAMIC) ? a.target_kind : TK;
        if (kind == MHX_TARGET_CORR_GAUSS) {
            mhx_real wv[R];
            mhx_ram_matvec<G, R, true>(srd_a, 0u, vec, d, lane, lds, wv);
            MHX_WAVE_SYNC();
#pragma unroll
            for (int r = 0; r < R; ++r) if (tg + G * r < d) lds[vec + tg + G * r] = wv[r];
            MHX_WAVE_SYNC();
            mhx_real q = MHX_R(0.0);
            for (int j = 0; j < d; ++j) { const mhx_real w = lds[vec + j]; q = mhx_fma(w, w, q); }
            lpy = mhx_fma(-MHX_R(0.5), q, a.tconst);
        } else {
            mhx_lds_x yv;
            yv.lds = lds; yv.off = vec;
            lpy = mhx_target_eval<TK>(kind, yv, d, tparams, a.ntparams, a.tconst);
        }

        MHX_PROF_T(2);
        // ---- accept
        const mhx_real diff = lpy - lp;
        const mhx_real loga = (diff != diff) ? diff : (diff < MHX_R(0.0) ? diff : MHX_R(0.0));
        const mhx_real logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;
        loga_last = loga;
        // ---- state select (before the update: the candidate dies here, not after the fold)
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = acc ? y[r] : x[r];
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += acc && tg == 0 ? 1u : 0u;

        // ---- adapt: the update becomes pending
        have_v = false;
        if (it < a.n_adapt) {
            const mhx_real da = mhx_exp(loga) - a.alpha;
            const bool adapt = da == da;
            if (!adapt) st |= 2u;
            const mhx_real c2 = (eta * mhx_abs(da)) / nz.nn;
            const mhx_real sc = da > MHX_R(0.0) ? c2 : -c2;
            mhx_real aa[4], gg[4];
            bool bad = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const mhx_real T0 = mhx_fma(sc, nz.pu[e], MHX_R(1.0));
                const mhx_real T1 = mhx_fma(sc, e < 3 ? nz.pu[(e + 1) & 3] : nz.incl, MHX_R(1.0));    // the prefix of the NEXT element (lane's own total after its last)
                const bool in = 4 * lane + e < d;
                if (in && !(T1 > MHX_R(0.0))) bad = true;
                const mhx_real ae = mhx_sqrt(T1 / T0);
                aa[e] = in ? ae : MHX_R(1.0);
                gg[e] = in ? (sc * nz.u[e]) / (T0 * ae) : MHX_R(0.0);
            }
            bool ok = adapt && __ballot(bad) == 0ull;
            if (adapt && !ok) st |= 1u;
            // a in the row layout: through the LDS vector (the target is done with it)
            MHX_WAVE_SYNC();
            if (in_vec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) lds[vec + 4 * lane + e] = aa[e];
            }
            MHX_WAVE_SYNC();
            mhx_real nd[R];
#pragma unroll
            for (int r = 0; r < R; ++r) nd[r] = tg + G * r < d ? dg[r] * lds[vec + tg + G * r] : dg[r];
            if (!a.default_bounds) {
                bool out = false;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (tg + G * r < d && !(a.eig_lo <= nd[r] && nd[r] <= a.eig_hi)) out = true;
                if (__ballot(out) != 0ull) ok = false;
            }
            if (ok) {                                                    // wave-uniform
                if (lane == 0) { psc[m] = sc; pstep[m] = step; }         // all that is kept of the update (read after the wave syncs below)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    svec[(m * R + r) * 64 + tg] = v[r];
                    dg[r] = nd[r];
                    dmn[r] = nd[r] < dmn[r] ? nd[r] : dmn[r];
                    dmx[r] = nd[r] > dmx[r] ? nd[r] : dmx[r];
                }
                ++m;                                                     // (the scratch is re-read by other lanes of this wave: fences at the reads)
            }
            MHX_PROF_T(3);
            const bool at_end = it + 1 == a.n_adapt || it + 1 == a.nsteps;
            if (m == K || (at_end && m > 0)) {
                // ---- flush: S_0 <- S_0 M_1 ... M_m into the other buffer, the next step's mat-vec on the way
                const bool fuse = it + 1 < a.nsteps;
                MHX_WAVE_SYNC();
                // the fold reads the coefficients TRANSPOSED (lane = (update, column of the group)): they go through the chain's
                // scratch once per fold -- each lane writes its four elements of every pending update, regenerated
                for (int k = 0; k < m; ++k) {                            // wave-uniform
                    mhx_real ra[4], rg[4], ru[4];
                    mhx_ram_defer_regen(ks, id_lo, id_hi, pstep[k], psc[k], d, lane, ra, rg, ru);
                    if (has_blk) {
                        mhx_f4* q = coef + (k * 256 + 4 * lane);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            mhx_f4 c4;
                            c4.x = ra[e]; c4.y = rg[e]; c4.z = ru[e]; c4.w = MHX_R(0.0);
                            q[e] = c4;
                        }
                    }
                }
                MHX_WAVE_SYNC();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                mhx_ram_noise4 nx;
#pragma unroll
                for (int e = 0; e < 4; ++e) nx.u[e] = nx.pu[e] = MHX_R(0.0);
                nx.incl = nx.nn = MHX_R(0.0);
                if (fuse) mhx_ram_draw4(ks, id_lo, id_hi, step + 1u, d, lane, nx);
                if (in_vec) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) lds[vec + 4 * lane + e] = nx.u[e];
                }
                mhx_ram_flush_f<R, K> fl;
                fl.lds = lds; fl.ring = ring; fl.unext = vec; fl.srd = srd; fl.tg = tg; fl.m = m; fl.coef = coef; fl.stage = stage;
                fl.prefetch(0);
                fl.vbase = (mhx_u32)(((sel ^ 1) * tri_pad + tg) * MHX_RB);
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int r = 0; r < R; ++r) fl.s[k][r] = k < m ? svec[(k * R + r) * 64 + tg] : MHX_R(0.0);
#pragma unroll
                for (int r = 0; r < R; ++r) fl.vn[r] = MHX_R(0.0);
                mhx_ram_stream<G, MHX_RAM_NV(R), MHX_RAM_MIRF(G, R)> stream;
                stream.begin(srd, scur, d, tg, lds, ring);
                mhx_ram_cols<G, R, 0>::run(stream, d, 0, fl);
                sel ^= 1;
                m = 0;
                MHX_PROF_T(4);
                if (fuse) {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = fl.vn[r];
                    nz = nx;
                    have_v = true;
                }
            }
        }

        if (step == save_next) {
            mhx_real* rowp = a.samples + slot * (long)(d + 1) * ld + c;
#pragma unroll
            for (int r = 0; r < R; ++r) if (tg + G * r < d) rowp[(long)(tg + G * r) * ld] = x[r];
            if (tg == 0) {
                rowp[(long)d * ld] = lp;
                a.accepted[slot * ld + c] = acc ? 1 : 0;
                if (a.rec_loga) a.rec_loga[slot * ld + c] = loga;
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
        MHX_WAVE_SYNC();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = tg + G * r;
        if (row < d) {
            a.x[(long)row * ld + c] = x[r];
            a.dmin[(long)c * d + row] = dmn[r];
            a.dmax[(long)c * d + row] = dmx[r];
        }
    }
    if (tg == 0) {
        a.lp[c] = lp;
        a.acc_count[c] = nacc;
        a.last_acc[c] = last ? 1 : 0;
        a.sel[c] = (unsigned char)sel;
        a.status[c] = (unsigned char)st;
        a.loga[c] = loga_last;
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
    }
#ifdef MHX_TOOLS_BUILD
    MHX_PROF_T(7);
    if (blockIdx.x == 0 && lane == 0 && a.prof)
        printf("defer prof (cycles, block 0, %d steps): draw+apply %llu  matvec %llu  target %llu  adapt %llu  fold %llu  other %llu\n", a.nsteps,
               prof_[0], prof_[1], prof_[2], prof_[3], prof_[4], prof_[7]);
#endif
}

#ifdef MHX_JIT_RAM
#ifdef MHX_JIT_DEFER_K
#define MHX_JIT_RAM_BOUNDS __launch_bounds__(64, 2)      // the deferred-factor form is built for two waves per SIMD, like its pre-built kernels
#else
#define MHX_JIT_RAM_BOUNDS __launch_bounds__(64)
#endif
extern "C" __global__ void MHX_JIT_RAM_BOUNDS
mhx_jit_ram(const mhx_ram_args a, const mhx_real* __restrict__ tparams)
{
    extern __shared__ mhx_real mhx_ram_lds[];
#ifdef MHX_JIT_DEFER_K
    mhx_ram_defer_body<MHX_JIT_R, MHX_JIT_DEFER_K, MHX_JIT_TK>(a, tparams, mhx_ram_lds);
#else
    mhx_ram_body<MHX_JIT_G, MHX_JIT_R, MHX_JIT_TK>(a, tparams, mhx_ram_lds);
#endif
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_ram_init(const mhx_ram_args a, const mhx_real* __restrict__ tparams, const int draw)
{
    mhx_ram_init_body<MHX_JIT_TK>(a, tparams, draw);
}
#endif
MHX_NS_END
