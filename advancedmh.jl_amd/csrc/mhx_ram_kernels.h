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

#ifndef MHX_RAM_NV
#define MHX_RAM_NV 2           // 16-byte loads in flight per lane and chunk
#endif

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

// x accessor over LDS (broadcast reads: every lane of a group evaluates the target redundantly)
struct mhx_lds_x {
    const float* p;
    MHX_DEV float operator[](int k) const { return p[k]; }
};

MHX_DEV long mhx_ram_col_off(int i, int d) { return (long)i * d - ((long)i * (i - 1)) / 2; }

typedef float mhx_f4 __attribute__((ext_vector_type(4)));

// floats per streaming chunk and per ring of ONE chain (ring = 2 chunks, a power of two)
#define MHX_RAM_CHF(G) (MHX_RAM_NV * (G) * 4)
#define MHX_RAM_RING(G) (2 * MHX_RAM_CHF(G))
// LDS floats per chain: ring + current noise + next noise + candidate
#define MHX_RAM_LDS_PER_CHAIN(G, d) (MHX_RAM_RING(G) + 3 * (d))

// Streaming a packed factor.  A chain's factor is ONE contiguous array, so its group pulls it in
// full-width pieces -- MHX_RAM_NV x (G lanes x 16 B) = one chunk per round, every lane active -- and
// parks it in the chain's two-chunk LDS ring; the column logic reads its row segments from LDS while
// the next chunk's loads are already in flight in VGPRs.  A column that straddles two chunks is
// processed once its tail has arrived.  All groups of a wave see the same column / chunk boundaries
// (same d), so the control flow is wave-uniform.
template <int G, class F>
MHX_DEV void mhx_ram_stream_columns(const float* __restrict__ S, const int d, const int tg, float* ring, F&& f)
{
    constexpr int CHF = MHX_RAM_CHF(G);
    const long tri = (long)d * (d + 1) / 2;
    const long nvec = (tri + 3) >> 2;
    const mhx_f4* __restrict__ src = (const mhx_f4*)S;
    mhx_f4* ring4 = (mhx_f4*)ring;
    const int nchunks = (int)((tri + CHF - 1) / CHF);
    mhx_f4 regs[MHX_RAM_NV];
    const mhx_f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int v = 0; v < MHX_RAM_NV; ++v) {
        const long i4 = (long)v * G + tg;
        regs[v] = i4 < nvec ? src[i4] : zero4;
    }
#pragma unroll
    for (int v = 0; v < MHX_RAM_NV; ++v) ring4[v * G + tg] = regs[v];
    __syncthreads();
    int col = 0;
    long off = 0;                                           // linear offset of column `col`
    for (int k = 0; k < nchunks; ++k) {
        const bool more = k + 1 < nchunks;
        if (more) {
#pragma unroll
            for (int v = 0; v < MHX_RAM_NV; ++v) {
                const long i4 = (long)(k + 1) * (MHX_RAM_NV * G) + v * G + tg;
                regs[v] = i4 < nvec ? src[i4] : zero4;
            }
        }
        const long avail = (long)(k + 1) * CHF < tri ? (long)(k + 1) * CHF : tri;
        while (col < d && off + (d - col) <= avail) {
            f(col, off);
            off += d - col;
            ++col;
        }
        if (more) {
            __syncthreads();
#pragma unroll
            for (int v = 0; v < MHX_RAM_NV; ++v) ring4[(((k + 1) & 1) * (MHX_RAM_NV * G)) + v * G + tg] = regs[v];
            __syncthreads();
        }
    }
}

// rows of column i owned by this lane, from its chain's ring.  Row slots entirely above the diagonal
// (G (r+1) <= i) are skipped by a wave-uniform test; inside a slot the read is unconditional and the
// rows above the diagonal are zeroed by a select -- no divergent branches in the column loops.
template <int G, int R>
MHX_DEV void mhx_ram_ring_col(const float* ring, const int i, const long off, const int d, const int tg, float (&col)[R])
{
    constexpr int RM = MHX_RAM_RING(G) - 1;
    const int base = (int)(off & RM) + (tg - i);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        col[r] = 0.0f;
        if (G * (r + 1) > i) {                                         // wave-uniform
            const int row = tg + G * r;
            const float v = ring[(base + G * r) & RM];
            col[r] = (row >= i && row < d) ? v : 0.0f;
        }
    }
}

// v = S u: column sweep (columns ascending == the row-dot's ascending j order)
template <int G, int R>
MHX_DEV void mhx_ram_matvec(const float* __restrict__ S, const float* ush, const int d, const int tg, float* ring,
                            float (&v)[R])
{
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0.0f;
    mhx_ram_stream_columns<G>(S, d, tg, ring, [&](const int i, const long off) {
        float col[R];
        mhx_ram_ring_col<G, R>(ring, i, off, d, tg, col);
        const float ui = ush[i];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (G * (r + 1) > i) {                                     // wave-uniform
                // rows above the diagonal carry col == 0: fma(0, u, v) == v bit for bit
                const float nv = mhx_fma(col[r], ui, v[r]);
                v[r] = (tg + G * r >= i) ? nv : v[r];
            }
    });
}

// draw U = randn(d) of `step` into the chain's LDS slot (lane tg draws Philox blocks tg, tg+G, ...) and
// return |U|^2 (ascending order, every lane of the group)
template <int G>
MHX_DEV float mhx_ram_draw(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step, const int d,
                           const int tg, float* ush)
{
    const int nblk = (d + 3) >> 2;
    for (int b0 = 0; b0 < nblk; b0 += G) {                 // wave-uniform trip count
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

// state of the rank-1 sweep that is carried across columns
template <int R>
struct mhx_ram_sweep {
    float w[R];        // the rank-1 vector, rotated column by column
    float nd[R];       // new diagonal entries of the rows this lane owns
    float vo[R];       // next step's S_old U'   (fused mat-vec)
    float vn[R];       // next step's S_new U'
    bool ok;           // per chain: the downdate is still positive definite
};

// one column of the sign-unified rank-1 sweep (sg = +1 update, -1 downdate, per chain), written to Snew;
// if `fuse`, the NEXT step's proposal mat-vec is accumulated for both the old and the new factor.
// `active` masks chains that do not adapt in this step (NaN log-ratio) or whose downdate failed.
template <int G, int R>
MHX_DEV void mhx_ram_sweep_col(const float (&col)[R], float* __restrict__ Snew, const float* unext, const int i,
                               const long off, const int d, const int tg, const int gbase, const float sg,
                               const bool adapt, const bool fuse, mhx_ram_sweep<R>& sw)
{
    const int il = i % G, ir = i / G;
    float aii = 0.0f, bi = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r == ir) {                                                 // wave-uniform
            if (G == 64) {                                             // one chain per wave: scalar broadcast
                aii = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, col[r]), il));
                bi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sw.w[r]), il));
            } else {                                                   // the group's lane holding row i
                aii = __shfl(col[r], gbase + il, 64);
                bi = __shfl(sw.w[r], gbase + il, 64);
            }
        }
    const float sn = bi / aii;
    if (sg < 0.0f && sn * sn > 1.0f) sw.ok = false;                    // PosDefException upstream (per chain)
    const float ss = sg * sn;
    const float cs = mhx_sqrt(mhx_fma(ss, sn, 1.0f));
    const float rcs = 1.0f / cs;                                       // one reciprocal per column
    const float diag = cs * aii;
    const float un = fuse ? unext[i] : 0.0f;
    const bool live = adapt && sw.ok;
    float* dst = Snew + (off - i);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (G * (r + 1) > i) {                                         // wave-uniform: slot has rows >= i
            const int row = tg + G * r;
            const bool below = row > i && row < d;
            const bool ondiag = row == i;
            const float Aji = col[r], vj = sw.w[r];
            const float oe = mhx_fma(ss, vj, Aji) * rcs;
            const float wn = mhx_fma(cs, vj, -(sn * oe));
            const float out = ondiag ? diag : (below ? oe : 0.0f);
            sw.w[r] = (below && live) ? wn : vj;
            sw.nd[r] = (ondiag && live) ? diag : sw.nd[r];
            if ((below || ondiag) && live) dst[row] = out;
            if (fuse) {
                // rows above the diagonal contribute fma(0, un, v) == v
                sw.vo[r] = mhx_fma(Aji, un, sw.vo[r]);
                sw.vn[r] = mhx_fma(out, un, sw.vn[r]);
            }
        }
    }
}

// G = lanes per chain, R = rows per lane (dim <= G R)
template <int G, int R, int TK>
MHX_DEV void mhx_ram_body(const mhx_ram_args& a, const float* __restrict__ tparams, float* lds)
{
    constexpr int CPW = 64 / G;                  // chains per wave (= per block)
    // XCD-aware mapping: blocks b, b+8, ... run on one XCD; give them consecutive chain groups
    const int nb = gridDim.x;
    const int per = (nb + 7) >> 3;
    const int wslot = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    const int lane = threadIdx.x;
    const int g = lane / G, tg = lane % G, gbase = g * G;
    const long c_raw = (long)wslot * CPW + g;
    const bool valid = c_raw < a.nchains;
    if (__ballot(valid) == 0ull) return;         // whole wave past the end
    const long c = valid ? c_raw : (long)a.nchains - 1;      // idle groups shadow the last chain (no stores)
    const int d = a.dim;
    const long ld = a.ld;
    const long tri = (long)d * (d + 1) / 2;
    const long tri_pad = (tri + 3) & ~3L;
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    float* ring = lds + (long)g * MHX_RAM_RING(G);                       // rings first: 16-byte aligned
    float* vecs = lds + (long)CPW * MHX_RAM_RING(G) + (long)g * 3 * d;
    float* ucur = vecs;          // [d] noise of the current step (dead after its mat-vec: target scratch)
    float* unxt = vecs + d;      // [d] noise of the next step (fused mat-vec)
    float* ysh = vecs + 2 * d;   // [d] candidate

    float x[R], dmn[R], dmx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = tg + G * r;
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
    bool have_v = false;         // wave-uniform: v = S U of this step (and nn) came out of the previous sweep
    float v[R], nn = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0.0f;

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        const float* Scur = (sel ? a.S1 : a.S0) + c * tri_pad;
        float* Snew = (sel ? a.S0 : a.S1) + c * tri_pad;

        // ---- U = randn(d), v = S U, x' = v + x   (RAM.jl:135-136)
        if (!have_v) {
            nn = mhx_ram_draw<G>(ks, id_lo, id_hi, step, d, tg, ucur);
            mhx_ram_matvec<G, R>(Scur, ucur, d, tg, ring, v);
        }
        float y[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = tg + G * r;
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
            mhx_ram_matvec<G, R>(a.acol, ysh, d, tg, ring, wv);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) if (tg + G * r < d) ucur[tg + G * r] = wv[r];
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

        // ---- adapt (RAM.jl:153-173, :259-264) during warm-up; the step index is wave-uniform
        have_v = false;
        if (it < a.n_adapt) {
            const float da = mhx_exp(loga) - a.alpha;                    // :159
            const bool adapt = da == da;                                 // a NaN log-ratio skips the adaptation
            if (!adapt) st |= 2u;
            const float eta = a.eta[it];                                 // :162 iteration^-gamma
            const float coef = mhx_sqrt(eta * __builtin_fabsf(da)) / mhx_sqrt(nn);   // :163
            mhx_ram_sweep<R> sw;
#pragma unroll
            for (int r = 0; r < R; ++r) { sw.w[r] = v[r] * coef; sw.nd[r] = 0.0f; sw.vo[r] = 0.0f; sw.vn[r] = 0.0f; }
            sw.ok = true;
            const float sg = da > 0.0f ? 1.0f : -1.0f;                   // :165 sign(da) == 1 ? update : downdate
            const bool fuse = it + 1 < a.nsteps;                         // a next step exists in this launch
            float nn_next = 0.0f;
            __syncthreads();                                             // every lane is done with ucur / ysh
            if (fuse) nn_next = mhx_ram_draw<G>(ks, id_lo, id_hi, step + 1u, d, tg, unxt);
            mhx_ram_stream_columns<G>(Scur, d, tg, ring, [&](const int i, const long off) {
                float col[R];
                mhx_ram_ring_col<G, R>(ring, i, off, d, tg, col);
                mhx_ram_sweep_col<G, R>(col, Snew, unxt, i, off, d, tg, gbase, sg, adapt && valid, fuse, sw);
            });
            if (adapt && !sw.ok) st |= 1u;
            bool ok = adapt && sw.ok;
            // valid_eigenvalues (RAM.jl:239-245): every diagonal entry of the chain inside [lo, hi]
            if (!a.default_bounds) {
                bool bad = false;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (tg + G * r < d && !(a.eig_lo <= sw.nd[r] && sw.nd[r] <= a.eig_hi)) bad = true;
                const mhx_u64 bm = __ballot(bad);
                const mhx_u64 gm = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << gbase);
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
                // the next step's mat-vec is done: S_{t+1} = the new factor if it was kept, else the old one.
                // (vo is exact whatever happened to the sweep: it only reads the old columns.)
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = ok ? sw.vn[r] : sw.vo[r];
                nn = nn_next;
                have_v = true;
                float* sp = ucur; ucur = unxt; unxt = sp;
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
                float* rowp = a.samples + slot * (long)(d + 1) * ld + c;
#pragma unroll
                for (int r = 0; r < R; ++r) if (tg + G * r < d) rowp[(long)(tg + G * r) * ld] = x[r];
                if (tg == 0) {
                    rowp[(long)d * ld] = lp;
                    a.accepted[slot * ld + c] = acc ? 1 : 0;
                }
            }
            save_next += (mhx_u32)a.thinning;
            ++slot;
        }
        __syncthreads();
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
        }
    }
    if (lane == 0) atomicAdd(a.acc_total, (mhx_u64)wave_acc);
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
    mhx_ram_body<MHX_JIT_G, MHX_JIT_R, MHX_JIT_TK>(a, tparams, mhx_ram_lds);
}
extern "C" __global__ void __launch_bounds__(256)
mhx_jit_ram_init(const mhx_ram_args a, const float* __restrict__ tparams, const int draw)
{
    mhx_ram_init_body<MHX_JIT_TK>(a, tparams, draw);
}
#endif
