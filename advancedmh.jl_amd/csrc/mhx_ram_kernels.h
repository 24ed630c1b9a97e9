// mhx_ram_kernels.h -- Robust Adaptive Metropolis (Vihola 2012), one WAVEFRONT per chain.
//
// Replaces ram_step_inner (src/RobustAdaptiveMetropolis.jl:123-151), ram_adapt (:153-173),
// valid_eigenvalues (:239-245) and the step / step_warmup methods (:216-278).
//
// Every chain owns a lower-triangular factor S (d(d+1)/2 floats: 80 KB at d = 200), so a lane
// cannot own a chain.  A 64-lane wave owns one: lane t holds rows t, t+64, ... of x, U, v, w.  S is
// stored per chain as a PACKED COLUMN-MAJOR lower triangle (column i = rows i..d-1, contiguous), so
// that both passes over it are column sweeps with coalesced row-segment loads:
//   pass A  x' = S U + x      v_j += S_ji U_i, columns i ascending  == row-dot in ascending order
//   pass B  rank-1 update     for column i: rotation (c,s) from (S_ii, w_i), then every row j > i
//                             independently -- the upstream lowrankupdate/lowrankdowndate sweep
// The update is written to the chain's SECOND factor buffer and a per-chain selector flips only if
// the new factor is valid (downdate stayed positive definite, diagonal inside the eigenvalue bounds
// -- RAM.jl:259-264 keeps the old S otherwise), so a rejected update costs no copy.
// HBM traffic per adapting step: 1 read + 1 write of S -- the read pass of the NEXT step's mat-vec is
// fused into the update sweep (its noise is recomputable from the counter RNG); 1 read when S is fixed.
#pragma once
#include "mhx_targets.h"

#ifndef MHX_RAM_NV
#define MHX_RAM_NV 2           // 16-byte loads in flight per lane and chunk (chunk = NV KiB per wave); measured best of 2/4/8
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

// x accessor over LDS (broadcast reads: every lane evaluates the target redundantly)
struct mhx_lds_x {
    const float* p;
    MHX_DEV float operator[](int k) const { return p[k]; }
};

MHX_DEV long mhx_ram_col_off(int i, int d) { return (long)i * d - ((long)i * (i - 1)) / 2; }

// ---------------------------------------------------------------------------------------------
// Streaming a packed factor.  A chain's factor is ONE contiguous array, so it is pulled through the
// wave in full-width pieces -- MHX_RAM_NV x (64 lanes x 16 B) = one "chunk" per round, every lane
// active, 16-byte aligned -- and parked in a two-chunk LDS ring; the column logic then reads its row
// segments from LDS.  While the columns of chunk k are processed, the loads of chunk k+1 are already
// in flight in VGPRs: 4 VGPRs hold 1 KiB in flight (per-column masked dword loads held ~0.4 KiB).
// A column that straddles two chunks is processed once its tail has arrived.
#define MHX_RAM_CHF (MHX_RAM_NV * 256)          // floats per chunk
#define MHX_RAM_RING (2 * MHX_RAM_CHF)          // floats in the ring (power of two)

typedef float mhx_f4 __attribute__((ext_vector_type(4)));

template <class F>
MHX_DEV void mhx_ram_stream_columns(const float* __restrict__ S, const int d, const int t, float* ring, F&& f)
{
    const long tri = (long)d * (d + 1) / 2;
    const long nvec = (tri + 3) >> 2;                       // the host pads every factor to a multiple of 4 floats
    const mhx_f4* __restrict__ src = (const mhx_f4*)S;
    mhx_f4* ring4 = (mhx_f4*)ring;
    const int nchunks = (int)((tri + MHX_RAM_CHF - 1) / MHX_RAM_CHF);
    mhx_f4 regs[MHX_RAM_NV];
    const mhx_f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int v = 0; v < MHX_RAM_NV; ++v) {
        const long i4 = (long)v * 64 + t;
        regs[v] = i4 < nvec ? src[i4] : zero4;
    }
#pragma unroll
    for (int v = 0; v < MHX_RAM_NV; ++v) ring4[v * 64 + t] = regs[v];
    __syncthreads();
    int col = 0;
    long off = 0;                                           // linear offset of column `col`
    for (int k = 0; k < nchunks; ++k) {
        const bool more = k + 1 < nchunks;
        if (more) {
#pragma unroll
            for (int v = 0; v < MHX_RAM_NV; ++v) {
                const long i4 = (long)(k + 1) * (MHX_RAM_NV * 64) + v * 64 + t;
                regs[v] = i4 < nvec ? src[i4] : zero4;
            }
        }
        const long avail = (long)(k + 1) * MHX_RAM_CHF < tri ? (long)(k + 1) * MHX_RAM_CHF : tri;
        while (col < d && off + (d - col) <= avail) {
            if (!f(col, off)) return;
            off += d - col;
            ++col;
        }
        if (more) {
            __syncthreads();
#pragma unroll
            for (int v = 0; v < MHX_RAM_NV; ++v)
                ring4[(((k + 1) & 1) * (MHX_RAM_NV * 64)) + v * 64 + t] = regs[v];
            __syncthreads();
        }
    }
}

// rows of column i owned by this lane, from the ring.  Row slots entirely above the diagonal
// (64 (r+1) <= i) are skipped by a wave-uniform test; inside a slot the read is unconditional and the
// rows above the diagonal are zeroed by a select -- no divergent branches in the column loops.
template <int R>
MHX_DEV void mhx_ram_ring_col(const float* ring, const int i, const long off, const int d, const int t, float (&col)[R])
{
    const int base = (int)(off & (MHX_RAM_RING - 1)) + (t - i);        // ring index of row t (may be negative: masked)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        col[r] = 0.0f;
        if (64 * (r + 1) > i) {                                        // wave-uniform
            const int row = t + 64 * r;
            const float v = ring[(base + 64 * r) & (MHX_RAM_RING - 1)];
            col[r] = (row >= i && row < d) ? v : 0.0f;
        }
    }
}

// v = S u: column sweep (columns ascending == the row-dot's ascending j order)
template <int R>
MHX_DEV void mhx_ram_matvec(const float* __restrict__ S, const float* ush, const int d, const int t, float* ring,
                            float (&v)[R])
{
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0.0f;
    mhx_ram_stream_columns(S, d, t, ring, [&](const int i, const long off) {
        float col[R];
        mhx_ram_ring_col<R>(ring, i, off, d, t, col);
        const float ui = ush[i];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (64 * (r + 1) > i) {                                    // wave-uniform
                // rows above the diagonal carry col == 0: fma(0, u, v) == v bit for bit (v is finite or NaN)
                const float nv = mhx_fma(col[r], ui, v[r]);
                v[r] = (t + 64 * r >= i) ? nv : v[r];
            }
        return true;
    });
}

// draw U = randn(d) of `step` into LDS (lane b draws Philox block b) and return |U|^2 (ascending order)
MHX_DEV float mhx_ram_draw(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step, const int d,
                           const int t, float* ush)
{
    const int nblk = (d + 3) >> 2;
    for (int b = t; b < nblk; b += 64) {
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
    float vo[R];       // next step's S_old U'   (fused mat-vec, see below)
    float vn[R];       // next step's S_new U'
    bool ok;
    unsigned st;
};

// rank-1 update (UP) / downdate of one column, written to Snew; if `fuse`, the NEXT step's proposal
// mat-vec is accumulated on the fly for both the old and the new factor (the next U is recomputable
// from the counter RNG), which removes that step's separate read pass over S.
template <int R, bool UP>
MHX_DEV bool mhx_ram_sweep_col(const float (&col)[R], float* __restrict__ Snew, const float* unext, const int i,
                               const long off, const int d, const int t, const bool fuse, mhx_ram_sweep<R>& sw)
{
    const int il = i & 63, ir = i >> 6;
    float aii = 0.0f, bi = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r == ir) {                                                 // wave-uniform
            aii = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, col[r]), il));
            bi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sw.w[r]), il));
        }
    // one reciprocal per column; the per-element divisions of the upstream sweep become multiplications
    float cs, sn, diag, rcs = 0.0f;
    if (UP) {
        const float rr = mhx_sqrt(mhx_fma(bi, bi, aii * aii));
        const float rinv = 1.0f / rr;
        cs = aii * rinv;
        sn = bi * rinv;
        diag = rr;
    } else {
        sn = bi / aii;
        const float s2 = sn * sn;
        if (s2 > 1.0f) { sw.ok = false; sw.st |= 1u; return false; }      // PosDefException upstream (wave-uniform)
        cs = mhx_sqrt(1.0f - s2);
        rcs = 1.0f / cs;
        diag = cs * aii;
    }
    const float un = fuse ? unext[i] : 0.0f;
    float* dst = Snew + (off - i);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (64 * (r + 1) > i) {                                        // wave-uniform: slot has rows >= i
            const int row = t + 64 * r;
            const bool below = row > i && row < d;
            const bool ondiag = row == i;
            const float Aji = col[r], vj = sw.w[r];
            float oe, wn;
            if (UP) {
                oe = mhx_fma(cs, Aji, sn * vj);
                wn = mhx_fma(cs, vj, -(sn * Aji));
            } else {
                oe = (Aji - sn * vj) * rcs;
                wn = mhx_fma(cs, vj, -(sn * oe));
            }
            const float out = ondiag ? diag : (below ? oe : 0.0f);
            sw.w[r] = below ? wn : vj;
            sw.nd[r] = ondiag ? diag : sw.nd[r];
            if (below || ondiag) dst[row] = out;
            if (fuse) {
                // rows above the diagonal contribute fma(0, un, v) == v
                sw.vo[r] = mhx_fma(Aji, un, sw.vo[r]);
                sw.vn[r] = mhx_fma(out, un, sw.vn[r]);
            }
        }
    }
    return true;
}

// R = rows per lane (dim <= 64 R)
template <int R, int TK>
MHX_DEV void mhx_ram_body(const mhx_ram_args& a, const float* __restrict__ tparams, float* lds)
{
    // XCD-aware chain mapping: blocks b, b+8, ... run on one XCD; give them consecutive chains
    const int nb = gridDim.x;
    const int per = (nb + 7) >> 3;
    const int c = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (c >= a.nchains) return;
    const int t = threadIdx.x;
    const int d = a.dim;
    const long ld = a.ld;
    const long tri = (long)d * (d + 1) / 2;
    const long tri_pad = (tri + 3) & ~3L;                    // per-chain stride of the factor buffers
    const mhx_u64 id = a.first_chain + (mhx_u64)c;
    const mhx_u32 id_lo = (mhx_u32)id, id_hi = (mhx_u32)(id >> 32);
    const mhx_philox_key ks = mhx_philox_schedule(a.seed);
    float* ring = lds;                          // [MHX_RAM_RING] streaming ring (16-byte aligned: first in LDS)
    float* ucur = lds + MHX_RAM_RING;           // [d] noise of the current step (dead after its mat-vec: target scratch)
    float* unxt = ucur + d;                     // [d] noise of the next step (fused mat-vec)
    float* ysh = unxt + d;                      // [d] candidate

    float x[R], dmn[R], dmx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = t + 64 * r;
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
    bool have_v = false;         // v = S U of this step (and nn = |U|^2) already produced by the previous sweep
    float v[R], nn = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0.0f;

    for (int it = 0; it < a.nsteps; ++it) {
        const mhx_u32 step = a.step0 + (mhx_u32)it;
        const float* Scur = (sel ? a.S1 : a.S0) + (long)c * tri_pad;
        float* Snew = (sel ? a.S0 : a.S1) + (long)c * tri_pad;

        // ---- U = randn(d), v = S U, x' = v + x   (RAM.jl:135-136)
        if (!have_v) {
            nn = mhx_ram_draw(ks, id_lo, id_hi, step, d, t, ucur);
            mhx_ram_matvec<R>(Scur, ucur, d, t, ring, v);
        }
        float y[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = t + 64 * r;
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
            mhx_ram_matvec<R>(a.acol, ysh, d, t, ring, wv);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) if (t + 64 * r < d) ucur[t + 64 * r] = wv[r];
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

        // ---- adapt (RAM.jl:153-173, :259-264) during warm-up
        have_v = false;
        if (it < a.n_adapt) {
            const float da = mhx_exp(loga) - a.alpha;                    // :159
            if (da == da) {
                const float eta = a.eta[it];                             // :162 iteration^-gamma
                const float coef = mhx_sqrt(eta * __builtin_fabsf(da)) / mhx_sqrt(nn);   // :163
                mhx_ram_sweep<R> sw;
#pragma unroll
                for (int r = 0; r < R; ++r) { sw.w[r] = v[r] * coef; sw.nd[r] = 0.0f; sw.vo[r] = 0.0f; sw.vn[r] = 0.0f; }
                sw.ok = true;
                sw.st = 0u;
                const bool up = da > 0.0f;                               // :165 sign(da) == 1
                const bool fuse = it + 1 < a.nsteps;                     // a next step exists in this launch
                float nn_next = 0.0f;
                __syncthreads();                                         // every lane is done with ucur / ysh
                if (fuse) nn_next = mhx_ram_draw(ks, id_lo, id_hi, step + 1u, d, t, unxt);
                if (up) {
                    mhx_ram_stream_columns(Scur, d, t, ring, [&](const int i, const long off) {
                        float col[R];
                        mhx_ram_ring_col<R>(ring, i, off, d, t, col);
                        return mhx_ram_sweep_col<R, true>(col, Snew, unxt, i, off, d, t, fuse, sw);
                    });
                } else {
                    mhx_ram_stream_columns(Scur, d, t, ring, [&](const int i, const long off) {
                        float col[R];
                        mhx_ram_ring_col<R>(ring, i, off, d, t, col);
                        return mhx_ram_sweep_col<R, false>(col, Snew, unxt, i, off, d, t, fuse, sw);
                    });
                }
                st |= sw.st;
                bool ok = sw.ok;
                const bool swept = sw.ok;                                // the sweep reached the last column
                // valid_eigenvalues (RAM.jl:239-245): every diagonal entry inside [lo, hi]
                if (ok && !a.default_bounds) {
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (t + 64 * r < d && !(a.eig_lo <= sw.nd[r] && sw.nd[r] <= a.eig_hi)) bad = true;
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
                    // the next step's mat-vec is done: S_{t+1} = new factor if it was kept, else the old one
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = ok ? sw.vn[r] : sw.vo[r];
                    nn = nn_next;
                    have_v = true;
                    float* sp = ucur; ucur = unxt; unxt = sp;
                }
            } else {
                st |= 2u;
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
            for (int r = 0; r < R; ++r) if (t + 64 * r < d) rowp[(long)(t + 64 * r) * ld] = x[r];
            if (t == 0) {
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
        const int row = t + 64 * r;
        if (row < d) {
            a.x[(long)row * ld + c] = x[r];
            a.dmin[(long)c * d + row] = dmn[r];
            a.dmax[(long)c * d + row] = dmx[r];
        }
    }
    if (t == 0) {
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
