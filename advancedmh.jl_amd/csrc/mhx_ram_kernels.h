// mhx_ram_kernels.h -- Robust Adaptive Metropolis (Vihola 2012), a GROUP of G lanes per chain.
//
// Replaces ram_step_inner (src/RobustAdaptiveMetropolis.jl:123-151), ram_adapt (:153-173),
// valid_eigenvalues (:239-245) and the step / step_warmup methods (:216-278).
//
// Every chain owns a lower-triangular factor S (d(d+1)/2 floats: 80 KB at d = 200), so a lane
// cannot own a chain.  A group of G lanes (16, 32 or 64; 64/G chains per wavefront) owns one: lane
// tg of the group holds rows tg, tg+G, ... of x, U, v, w.  All per-column work that is not per
// element -- the rotation (a division, a square root, a reciprocal), loop control, addressing --
// is then paid once per 64/G chains.  S is stored per chain as a PACKED COLUMN-MAJOR lower triangle
// (one contiguous array, padded to a multiple of 4 floats) and is streamed through a two-chunk LDS
// ring in full-width 16-byte loads; both passes over it are column sweeps:
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

struct mhx_ram_args {
    float* x;                 // [dim][ld]  (ABI layout; touched once per launch)
    float* lp;                // [ld]
    mhx_u32* acc_count;
    mhx_u64* acc_total;
    float* samples;           // [slots][dim+1][ld] or null
    unsigned char* accepted;
    unsigned char* last_acc;
    float* S0;                // [nchains][tri_pad] packed column-major lower (tri padded to a multiple of 4)
    float* S1;                // second buffer
    unsigned char* sel;       // [nchains] which buffer is current
    unsigned char* status;    // [nchains] bit0: a downdate left the PD cone, bit1: NaN log-ratio
    float* dmin;              // [nchains][dim] running min of diag(S)
    float* dmax;
    const float* eta;         // [nsteps] adaptation step sizes iteration^-gamma of this launch
    const float* acol;        // CORR_GAUSS target: inv(chol(Sigma)) packed column-major lower
    mhx_u64 seed;
    mhx_u64 first_chain;
    int nchains;
    int ld;
    int dim;
    int target_kind;
    int ntparams;
    float tconst;
    float alpha;
    float eig_lo, eig_hi;
    int default_bounds;
    mhx_u32 step0;
    int nsteps;
    int n_adapt;              // the first n_adapt steps of this launch adapt S (step_warmup)
    mhx_u32 save_next;
    int save_slot;
    int thinning;
};

// x accessor over LDS (broadcast reads: every lane evaluates the target redundantly)
struct mhx_lds_x {
    const float* p;
    MHX_DEV float operator[](int k) const { return p[k]; }
};

MHX_DEV long mhx_ram_col_off(int i, int d) { return (long)i * d - ((long)i * (i - 1)) / 2; }

typedef float mhx_f4 __attribute__((ext_vector_type(4)));

// LDS plan of one chain (= one wave = one block), in floats:
//   [ ring: NS chunks of CHF | mirror: the first MIRF floats of the ring again | marks | noise | next noise | candidate ]
// A chunk is NP pieces of 1 KB (64 lanes x 16 B), the unit of one LDS-DMA instruction.  The mirror makes
// every column CONTIGUOUS in LDS even when it runs off the end of the ring, so a column read is one
// per-column address plus compile-time offsets.
#ifndef MHX_RAM_NS
#define MHX_RAM_NS 4                                       // ring slots (a power of two)
#endif
#define MHX_RAM_NP(R) ((R) <= 4 ? 1 : ((R) <= 8 ? 2 : 4))   // pieces per chunk: a chunk holds the longest column
#define MHX_RAM_CHF(R) (MHX_RAM_NP(R) * 256)
#define MHX_RAM_RING(R) (MHX_RAM_NS * MHX_RAM_CHF(R))
#define MHX_RAM_MIRF(R) (256 * (((R) + 3) / 4))          // whole 1 KB pieces covering the longest column
#define MHX_RAM_FIXED_FLOATS(R) (MHX_RAM_RING(R) + MHX_RAM_MIRF(R) + 16)
#define MHX_RAM_LDS_FLOATS(R, d) (MHX_RAM_FIXED_FLOATS(R) + 3 * (d))

// ordering of one wave's LDS traffic (the block is one wave: no s_barrier needed)
#define MHX_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

MHX_DEV unsigned mhx_lds_addr(const void* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// one LDS-DMA piece: lane l copies 16 bytes from its own global address to LDS byte lds_dst + 16 l.
// The compiler does not see this load (no s_waitcnt bookkeeping): completion is counted by hand below.
MHX_DEV void mhx_glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// wait until at most n vector-memory operations of this wave are outstanding (n wave-uniform)
MHX_DEV void mhx_wait_vmcnt(int n)
{
#define MHX_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        MHX_VMW(0) MHX_VMW(1) MHX_VMW(2) MHX_VMW(3) MHX_VMW(4) MHX_VMW(5) MHX_VMW(6) MHX_VMW(7)
        MHX_VMW(8) MHX_VMW(9) MHX_VMW(10) MHX_VMW(11) MHX_VMW(12) MHX_VMW(13) MHX_VMW(14) MHX_VMW(15)
        MHX_VMW(16) MHX_VMW(17) MHX_VMW(18) MHX_VMW(19) MHX_VMW(20) MHX_VMW(21) MHX_VMW(22) MHX_VMW(23)
        MHX_VMW(24) MHX_VMW(25) MHX_VMW(26) MHX_VMW(27) MHX_VMW(28) MHX_VMW(29) MHX_VMW(30) MHX_VMW(31)
        default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    }
#undef MHX_VMW
}

// Streaming a packed factor.  A chain's factor is ONE contiguous array, so its wave pulls it in 1 KB
// pieces straight into an LDS ring (LDS-DMA: no staging registers), up to NS chunks ahead of the
// column being read; the column logic reads its row segments from LDS.  vmcnt counts loads AND stores
// in issue order, so the wave keeps its own count of vector-memory operations (`issued`; the sweep adds
// its stores) and waits for a chunk with s_waitcnt vmcnt(issued - mark of that chunk): the stores issued
// after it stay in flight.
template <int R>
struct mhx_ram_stream {
    static constexpr int NP = MHX_RAM_NP(R);
    static constexpr int CHF = MHX_RAM_CHF(R);
    static constexpr int NS = MHX_RAM_NS;
    const char* src;          // this lane's first 16 bytes
    float* ring;
    int* marks;               // [NS] value of `issued` after the last piece of the chunk in the slot
    unsigned ring_lds;        // LDS byte address of the ring
    int tg, nvec, nchunks;
    int issued;               // vector-memory operations issued by this wave since begin()
    int next_issue;           // first chunk not yet requested
    int landed;               // last chunk known to be in LDS

    MHX_DEV void issue(const int k)
    {
        const int slot = k & (NS - 1);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int q = k * NP + p;                                  // 1 KB piece index in the factor
            if (q * 64 < nvec) {                                       // wave-uniform
                const int fo = (slot * NP + p) * 256;                  // float offset inside the ring
                const bool on = q * 64 + tg < nvec;
                if (on) mhx_glds16(src + (long)q * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + 4u * fo)));
                ++issued;
                if (fo < MHX_RAM_MIRF(R)) {
                    if (on) mhx_glds16(src + (long)q * 1024,
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + 4u * (MHX_RAM_RING(R) + fo))));
                    ++issued;
                }
            }
        }
        marks[slot] = issued;
    }
    MHX_DEV void begin(const float* __restrict__ S, const int d, const int lane, float* ring_, int* marks_)
    {
        tg = lane;
        src = (const char*)S + 16 * lane;
        ring = ring_;
        marks = marks_;
        ring_lds = mhx_lds_addr(ring_);
        const int tri = d * (d + 1) / 2;
        nvec = (tri + 3) >> 2;
        nchunks = (nvec + NP * 64 - 1) / (NP * 64);
        issued = 0;
        next_issue = 0;
        landed = -1;
        // everything this wave issued so far (samples, the previous pass) is out of the count
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // the linear float range [start, end) of the column about to be read
    MHX_DEV void require(const int start, const int end)
    {
        const int c_lo = start / CHF, c_hi = (end - 1) / CHF;
        if (next_issue < nchunks && next_issue < c_lo + NS) {          // wave-uniform
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // reads of the slots being recycled are done
            do issue(next_issue++); while (next_issue < nchunks && next_issue < c_lo + NS);
        }
        if (c_hi > landed) {
            MHX_WAVE_SYNC();
            const int m = __builtin_amdgcn_readfirstlane(marks[c_hi & (NS - 1)]);
            mhx_wait_vmcnt(issued - m);
            landed = c_hi;
        }
    }
};

// columns 64 IR + il in ascending order; the row slot IR that holds the diagonal is a compile-time
// constant inside the functor, so slots above the diagonal vanish at compile time and only slot IR
// carries masks.  The functor returns false to abandon the sweep.
template <int R, int IR>
struct mhx_ram_cols {
    template <class F>
    static MHX_DEV bool run(mhx_ram_stream<R>& st, const int d, int off, F& f)
    {
        for (int il = 0; il < 64; ++il) {
            const int i = 64 * IR + il;
            if (i >= d) return true;
            st.require(off, off + (d - i));
            if (!f.template col<IR>(st, il, i, off)) return false;
            off += d - i;
        }
        return mhx_ram_cols<R, IR + 1>::run(st, d, off, f);
    }
};
template <int R>
struct mhx_ram_cols<R, R> {
    template <class F>
    static MHX_DEV bool run(mhx_ram_stream<R>&, const int, int, F&) { return true; }
};

// v = S u: column sweep (columns ascending == the row-dot's ascending j order).  Rows >= d of the last
// slot accumulate whatever follows the column in the ring; nothing ever reads them.
template <int R>
struct mhx_ram_matvec_f {
    const float* ring;
    const float* ush;
    int tg;
    float v[R];
    template <int IR>
    MHX_DEV bool col(mhx_ram_stream<R>&, const int il, const int i, const int off)
    {
        constexpr int RM = MHX_RAM_RING(R) - 1;
        const float* cp = ring + ((off & RM) - il) + tg;                // row tg + 64 IR of this column
        const float ui = ush[i];
        const float c0 = cp[0];
        v[IR] = mhx_fma(tg >= il ? c0 : 0.0f, ui, v[IR]);               // rows above the diagonal: fma(0, u, v) == v
#pragma unroll
        for (int r = IR + 1; r < R; ++r) v[r] = mhx_fma(cp[64 * (r - IR)], ui, v[r]);
        return true;
    }
};
template <int R>
MHX_DEV void mhx_ram_matvec(const float* __restrict__ S, const float* ush, const int d, const int tg, float* ring,
                            int* marks, float (&v)[R])
{
    mhx_ram_matvec_f<R> f;
    f.ring = ring; f.ush = ush; f.tg = tg;
#pragma unroll
    for (int r = 0; r < R; ++r) f.v[r] = 0.0f;
    mhx_ram_stream<R> st;
    st.begin(S, d, tg, ring, marks);
    mhx_ram_cols<R, 0>::run(st, d, 0, f);
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = f.v[r];
}

// draw U = randn(d) of `step` into LDS (lane tg draws Philox blocks tg, tg+64, ...) and return |U|^2
// (ascending order, every lane)
MHX_DEV float mhx_ram_draw(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step, const int d,
                           const int tg, float* ush)
{
    const int nblk = (d + 3) >> 2;
    for (int b0 = 0; b0 < nblk; b0 += 64) {                // wave-uniform trip count
        const int b = b0 + tg;
        float n[4];
        mhx_normal4(ks, id_lo, id_hi, step, MHX_STREAM_PROPOSAL, (mhx_u32)b, n);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * b + j < d) ush[4 * b + j] = n[j];
    }
    __syncthreads();
    float nn = 0.0f;
    for (int j = 0; j < d; ++j) { const float u = ush[j]; nn = mhx_fma(u, u, nn); }
    return nn;
}

// The sign-unified rank-1 sweep (sg = +1 update, -1 downdate), one column per call,
// written to Snew through a buffer descriptor whose range check drops rows >= d; the NEXT step's
// proposal mat-vec is accumulated for both the old and the new factor on the way.
// Invariant: w is exactly 0 in rows above the current column, so those rows need no masks:
// fma(ss, 0, 0) * rc == 0 leaves their accumulators alone.
template <int R>
struct mhx_ram_sweep_f {
    const float* ring;
    const float* unext;
    float* Snew;
    int tg, d;
    float sg;
    float w[R];        // the rank-1 vector, rotated column by column
    float nd[R];       // new diagonal entries of the rows this lane owns
    float vo[R];       // next step's S_old U'   (fused mat-vec)
    float vn[R];       // next step's S_new U'
    template <int IR>
    MHX_DEV bool col(mhx_ram_stream<R>& st, const int il, const int i, const int off)
    {
        constexpr int RM = MHX_RAM_RING(R) - 1;
        const float* cp = ring + ((off & RM) - il) + tg;
        const bool lo = tg >= il;                                      // row >= i inside slot IR
        const bool ondiag = tg == il;
        const float c0 = lo ? cp[0] : 0.0f;
        const float aii = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c0), il));
        const float bi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w[IR]), il));
        const float sn = bi / aii;
        if (sg < 0.0f && sn * sn > 1.0f) return false;                 // PosDefException upstream
        const float ss = sg * sn;
        const float cs = mhx_sqrt(mhx_fma(ss, sn, 1.0f));
        const float rcs = 1.0f / cs;                                   // one reciprocal per column
        const float diag = cs * aii;
        const float un = unext[i];
        // rows of column i as seen from row 0: base + 4 row, range d rows
        const mhx_srd srd = mhx_make_srd(Snew + (off - i), (mhx_u32)d * 4u);
        const mhx_u32 lane_off = (mhx_u32)tg * 4u;
        {
            const float vj = w[IR];
            const float oe = mhx_fma(ss, vj, c0) * rcs;
            const float wn = mhx_fma(cs, vj, -(sn * oe));
            const float out = ondiag ? diag : oe;
            w[IR] = ondiag ? 0.0f : wn;
            nd[IR] = ondiag ? diag : nd[IR];
#ifndef MHX_RAM_NOSTORE
            if (lo) mhx_srd_store(srd, lane_off + 256u * IR, 0u, out);
#endif
            vo[IR] = mhx_fma(c0, un, vo[IR]);
            vn[IR] = mhx_fma(out, un, vn[IR]);
        }
#pragma unroll
        for (int r = IR + 1; r < R; ++r) {
            const float Aji = cp[64 * (r - IR)], vj = w[r];
            const float oe = mhx_fma(ss, vj, Aji) * rcs;
            w[r] = mhx_fma(cs, vj, -(sn * oe));
#ifndef MHX_RAM_NOSTORE
            mhx_srd_store(srd, lane_off + 256u * r, 0u, oe);
#endif
            vo[r] = mhx_fma(Aji, un, vo[r]);
            vn[r] = mhx_fma(oe, un, vn[r]);
        }
#ifndef MHX_RAM_NOSTORE
        st.issued += R - IR;                                           // the stores of this column
#endif
        return true;
    }
};

// one wave per chain; R = rows per lane (dim <= 64 R)
template <int R, int TK>
MHX_DEV void mhx_ram_body(const mhx_ram_args& a, const float* __restrict__ tparams, float* lds)
{
    // XCD-aware mapping: blocks b, b+8, ... run on one XCD; give them consecutive chains
    const int nb = gridDim.x;
    const int per = (nb + 7) >> 3;
    const long c = (long)(blockIdx.x & 7u) * per + (long)(blockIdx.x >> 3);
    if (c >= a.nchains) return;
    const int tg = threadIdx.x;
    const int d = a.dim;
    const long ld = a.ld;
    const long tri = (long)d * (d + 1) / 2;
    const long tri_pad = (tri + 3) & ~3L;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    float* ring = lds;                                                  // 16-byte aligned
    int* marks = (int*)(lds + MHX_RAM_RING(R) + MHX_RAM_MIRF(R));
    float* vecs = lds + MHX_RAM_FIXED_FLOATS(R);
    float* ucur = vecs;          // [d] noise of the current step (dead after its mat-vec: target scratch)
    float* unxt = vecs + d;      // [d] noise of the next step (fused mat-vec)
    float* ysh = vecs + 2 * d;   // [d] candidate

    float x[R], dmn[R], dmx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = tg + 64 * r;
        x[r] = row < d ? a.x[(long)row * ld + c] : 0.0f;
        dmn[r] = row < d ? a.dmin[(long)c * d + row] : 0.0f;
        dmx[r] = row < d ? a.dmax[(long)c * d + row] : 0.0f;
    }
    float lp = a.lp[c];
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
    bool have_v = false;         // v = S U of this step (and nn) came out of the previous sweep
    float v[R], nn = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0.0f;

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        const float* Scur = (sel ? a.S1 : a.S0) + c * tri_pad;
        float* Snew = (sel ? a.S0 : a.S1) + c * tri_pad;

        // ---- U = randn(d), v = S U, x' = v + x   (RAM.jl:135-136)
        if (!have_v) {
            __syncthreads();
            nn = mhx_ram_draw(ks, id_lo, id_hi, step, d, tg, ucur);
            mhx_ram_matvec<R>(Scur, ucur, d, tg, ring, marks, v);
        }
        float y[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = tg + 64 * r;
            y[r] = v[r] + x[r];
            if (row < d) ysh[row] = y[r];
        }
        __syncthreads();

        // ---- lp' = logdensity(x')  (RAM.jl:140)
        float lpy;
        const int kind = (TK == MHX_TARGET_DYNAMIC) ? a.target_kind : TK;
        if (kind == MHX_TARGET_CORR_GAUSS) {
            // cooperative column sweep over A = inv(chol(Sigma)): w_j += A_ji y_i, i ascending
            float wv[R];
            mhx_ram_matvec<R>(a.acol, ysh, d, tg, ring, marks, wv);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) if (tg + 64 * r < d) ucur[tg + 64 * r] = wv[r];
            __syncthreads();
            float q = 0.0f;
            for (int j = 0; j < d; ++j) { const float w = ucur[j]; q = mhx_fma(w, w, q); }
            lpy = mhx_fma(-0.5f, q, a.tconst);
        } else {
            mhx_lds_x yv;
            yv.p = ysh;
            lpy = mhx_target_eval<TK>(kind, yv, d, tparams, a.ntparams, a.tconst);
        }

        // ---- accept (RAM.jl:147-148): loga = min(lp' - lp, 0); accept iff randexp > -loga
        const float diff = lpy - lp;
        const float loga = (diff != diff) ? diff : (diff < 0.0f ? diff : 0.0f);
        const float logu = mhx_accept_logu(ks, id_lo, id_hi, step, ac);
        const bool acc = logu < loga;

        // ---- adapt (RAM.jl:153-173, :259-264) during warm-up; everything here is wave-uniform
        have_v = false;
        if (it < a.n_adapt) {
            const float da = mhx_exp(loga) - a.alpha;                    // :159
            const bool adapt = da == da;                                 // a NaN log-ratio skips the adaptation
            if (!adapt) st |= 2u;
            if (adapt) {
                const float eta = a.eta[it];                             // :162 iteration^-gamma
                const float coef = mhx_sqrt(eta * __builtin_fabsf(da)) / mhx_sqrt(nn);   // :163
                const bool fuse = it + 1 < a.nsteps;                     // a next step exists in this launch
                mhx_ram_sweep_f<R> sw;
                sw.ring = ring; sw.unext = unxt; sw.Snew = Snew; sw.tg = tg; sw.d = d;
                sw.sg = da > 0.0f ? 1.0f : -1.0f;                        // :165 sign(da) == 1 ? update : downdate
#pragma unroll
                for (int r = 0; r < R; ++r) { sw.w[r] = v[r] * coef; sw.nd[r] = 0.0f; sw.vo[r] = 0.0f; sw.vn[r] = 0.0f; }
                float nn_next = 0.0f;
                __syncthreads();                                         // every lane is done with ucur / ysh
                if (fuse) nn_next = mhx_ram_draw(ks, id_lo, id_hi, step + 1u, d, tg, unxt);
                mhx_ram_stream<R> stream;
                stream.begin(Scur, d, tg, ring, marks);
                bool ok = mhx_ram_cols<R, 0>::run(stream, d, 0, sw);
                const bool swept = ok;                                   // the fused mat-vec saw every column
                if (!ok) st |= 1u;
                // valid_eigenvalues (RAM.jl:239-245): every diagonal entry of the chain inside [lo, hi]
                if (ok && !a.default_bounds) {
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (tg + 64 * r < d && !(a.eig_lo <= sw.nd[r] && sw.nd[r] <= a.eig_hi)) bad = true;
                    if (__ballot(bad) != 0ull) ok = false;
                }
                if (ok) {
                    sel ^= 1;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        dmn[r] = sw.nd[r] < dmn[r] ? sw.nd[r] : dmn[r];
                        dmx[r] = sw.nd[r] > dmx[r] ? sw.nd[r] : dmx[r];
                    }
                }
                if (fuse && swept) {
                    // the next step's mat-vec is done: S_{t+1} = the new factor if it was kept, else the old one
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = ok ? sw.vn[r] : sw.vo[r];
                    nn = nn_next;
                    have_v = true;
                    float* sp = ucur; ucur = unxt; unxt = sp;
                }
            }
        }

        // ---- state select (RAM.jl:267-277)
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = acc ? y[r] : x[r];
        lp = acc ? lpy : lp;
        nacc += acc ? 1u : 0u;
        last = acc;
        wave_acc += acc ? 1u : 0u;
        if (step == save_next) {
            float* rowp = a.samples + slot * (long)(d + 1) * ld + c;
#pragma unroll
            for (int r = 0; r < R; ++r) if (tg + 64 * r < d) rowp[(long)(tg + 64 * r) * ld] = x[r];
            if (tg == 0) {
                rowp[(long)d * ld] = lp;
                a.accepted[slot * ld + c] = acc ? 1 : 0;
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = tg + 64 * r;
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
        atomicAdd(a.acc_total, (mhx_u64)wave_acc);
    }
}

// initial state (RAM.jl:175-214): x0 = initial_params or randn(d); lp0; accepted = true (:213)
template <int TK>
MHX_DEV void mhx_ram_init_body(const mhx_ram_args& a, const float* __restrict__ tparams, const int draw)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.nchains) return;
    const long ld = a.ld;
    const int d = a.dim;
    float* xs = a.x + c;
    if (draw) {
        const mhx_u64 id = a.first_chain + (mhx_u64)c;
        const mhx_philox_key ks = mhx_philox_schedule(a.seed);
        const int nblk = (d + 3) >> 2;
        for (int b = 0; b < nblk; ++b) {
            float n[4];
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
}

#ifdef MHX_JIT_RAM
extern "C" __global__ void __launch_bounds__(64)
mhx_jit_ram(const mhx_ram_args a, const float* __restrict__ tparams)
{
    extern __shared__ float mhx_ram_lds[];
    mhx_ram_body<MHX_JIT_R, MHX_JIT_TK>(a, tparams, mhx_ram_lds);
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_ram_init(const mhx_ram_args a, const float* __restrict__ tparams, const int draw)
{
    mhx_ram_init_body<MHX_JIT_TK>(a, tparams, draw);
}
#endif
