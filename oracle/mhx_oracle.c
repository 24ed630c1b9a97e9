/*
 * mhx_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See mhx_oracle.h.
 *
 * Scalar, one chain at a time, fp32, written against the arithmetic spec of DESIGN.md section 3
 * so that it is bit-comparable with the HIP kernels.  Build: oracle/Makefile (gcc, -ffp-contract=off:
 * every fused multiply-add below is an explicit fmaf, every other operation rounds separately).
 */
#include "mhx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* bit casts                                                                                  */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al. 2011; constants as in rocrand_philox4x32_10.h:62-65)          */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_at(uint64_t seed, uint64_t id, uint32_t step, uint32_t stream, uint32_t block,
                      uint32_t out[4])
{
    uint32_t ctr[4] = { (uint32_t)id, (uint32_t)(id >> 32), step, (stream << 28) | block };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(ctr, key, out);
}

/* ------------------------------------------------------------------------------------------ */
/* transcendental spec (coefficients: tools/fit_coeffs.py)                                    */
#define LN2_HI 0x1.62e4p-1f           /* 16 significant bits: e*LN2_HI is exact for |e| < 256 */
#define LN2_LO 0x1.7f7d1cp-20f        /* ln2 - LN2_HI */
#define LOG2E  0x1.715476p+0f

float orc_logf(float x)
{
    uint32_t ix = f2u(x);
    int eadj = 0;
    if ((ix << 1) == 0) return -INFINITY;              /* log(+-0) */
    if (ix >> 31) return NAN;                          /* negative */
    if (ix >= 0x7f800000u) return x;                   /* +inf, nan */
    if (ix < 0x00800000u) {                            /* subnormal */
        x = x * 0x1p23f; ix = f2u(x); eadj = -23;
    }
    uint32_t t = ix - 0x3f2aaaabu;                     /* bits(2/3) */
    int32_t e = (int32_t)t >> 23;
    float m = u2f(ix - ((uint32_t)e << 23));           /* [2/3, 4/3) */
    float f = m - 1.0f;
    float ef = (float)(e + eadj);
    float q = -0x1.04cba2p-3f;
    q = fmaf(q, f, 0x1.19bbe2p-3f);
    q = fmaf(q, f, -0x1.f483fap-4f);
    q = fmaf(q, f, 0x1.1fd494p-3f);
    q = fmaf(q, f, -0x1.55913ep-3f);
    q = fmaf(q, f, 0x1.99bffep-3f);
    q = fmaf(q, f, -0x1.ffff28p-3f);
    q = fmaf(q, f, 0x1.55552cp-2f);
    q = fmaf(q, f, -0.5f);
    float f2 = f * f;
    float r = fmaf(f2, q, f);
    r = fmaf(ef, LN2_LO, r);
    r = fmaf(ef, LN2_HI, r);
    return r;
}

float orc_expf(float x)
{
    if (x != x) return x;
    if (x > 0x1.62e42ep+6f) return INFINITY;           /* > 88.72283 overflows */
    if (x < -0x1.9fe368p+6f) return 0.0f;              /* < -103.97 rounds to 0 */
    float n = rintf(x * LOG2E);
    float r = fmaf(n, -LN2_HI, x);
    r = fmaf(n, -LN2_LO, r);
    float p = 0x1.a1517cp-13f;
    p = fmaf(p, r, 0x1.6d4328p-10f);
    p = fmaf(p, r, 0x1.1110c6p-7f);
    p = fmaf(p, r, 0x1.5554eap-5f);
    p = fmaf(p, r, 0x1.555556p-3f);
    p = fmaf(p, r, 0.5f);
    float r2 = r * r;
    float y = fmaf(r2, p, r) + 1.0f;
    int ni = (int)n;
    int n1 = ni / 2;                                    /* truncation toward zero */
    int n2 = ni - n1;
    y = y * u2f((uint32_t)(n1 + 127) << 23);
    y = y * u2f((uint32_t)(n2 + 127) << 23);
    return y;
}

/* angle = 2*pi*k/2^32 ; quadrant reduction is pure integer arithmetic */
void orc_sincos2pi_u32(uint32_t k, float *s, float *c)
{
    uint32_t kk = k + 0x20000000u;                     /* wraps mod 2^32 */
    uint32_t q = kk >> 30;                             /* rounded quadrant, 0..3 */
    int32_t ri = (int32_t)(kk & 0x3fffffffu) - 0x20000000;   /* [-2^29, 2^29) */
    float r = (float)ri * 0x1p-32f;                    /* turns, [-1/8, 1/8) */
    float r2 = r * r;
    float sp = 0x1.4bc87cp+5f;
    sp = fmaf(sp, r2, -0x1.32ca9ep+6f);
    sp = fmaf(sp, r2, 0x1.466bbap+6f);
    sp = fmaf(sp, r2, -0x1.4abbcep+5f);
    sp = fmaf(sp, r2, 0x1.921fb6p+2f);
    sp = sp * r;
    float cp = 0x1.d9c326p+5f;
    cp = fmaf(cp, r2, -0x1.55c57ap+6f);
    cp = fmaf(cp, r2, 0x1.03c1dcp+6f);
    cp = fmaf(cp, r2, -0x1.3bd3ccp+4f);
    cp = fmaf(cp, r2, 1.0f);
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    if (q == 2 || q == 3) ss = -ss;
    if (q == 1 || q == 2) cc = -cc;
    *s = ss; *c = cc;
}

float orc_u01_open(uint32_t k) { return fmaf((float)k, 0x1p-32f, 0x1p-33f); }
float orc_u01_half(uint32_t k) { return (float)(k >> 8) * 0x1p-24f; }

/* Box-Muller: radius from k0, angle from k1 */
void orc_normal_pair(uint32_t k0, uint32_t k1, float *n0, float *n1)
{
    float l = orc_logf(orc_u01_open(k0));              /* <= 0 */
    float rad = sqrtf(-2.0f * l);
    float s, c;
    orc_sincos2pi_u32(k1, &s, &c);
    *n0 = rad * c;
    *n1 = rad * s;
}

void orc_normals(uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, int d, float *out)
{
    for (int b = 0; 4 * b < d; ++b) {
        uint32_t w[4]; float n[4];
        philox_at(seed, chain, step, stream, (uint32_t)b, w);
        orc_normal_pair(w[0], w[1], &n[0], &n[1]);
        orc_normal_pair(w[2], w[3], &n[2], &n[3]);
        for (int j = 0; j < 4 && 4 * b + j < d; ++j) out[4 * b + j] = n[j];
    }
}

/* log of the accept uniform: the block is shared by 4 consecutive steps */
float orc_accept_logu(uint64_t seed, uint64_t chain, uint32_t step)
{
    uint32_t w[4];
    philox_at(seed, chain, step >> 2, ORC_STREAM_ACCEPT, 0, w);
    return orc_logf(orc_u01_open(w[step & 3]));
}

/* ------------------------------------------------------------------------------------------ */
/* targets.  reference: logdensity(model, x) = model.logdensity(x), src/AdvancedMH.jl:74      */
#define LOG_2PI_D 1.8378770664093454835606594728112
#define HALF_LOG_2PI_F 0x1.d67f1cp-1f

static float target_const(const orc_target *t)
{
    const int d = t->dim;
    double c = -0.5 * (double)d * LOG_2PI_D;
    switch (t->kind) {
    case ORC_TARGET_CORR_GAUSS: {                      /* + log det A = -1/2 log det Sigma */
        size_t off = 0;
        for (int i = 0; i < d; ++i) { c += log((double)t->params[off + i]); off += (size_t)i + 1; }
        break;
    }
    case ORC_TARGET_BANANA: c -= 0.5 * log(100.0); break;
    case ORC_TARGET_FUNNEL: c -= log(3.0); break;
    default: break;
    }
    return (float)c;
}

/* sum of squares of a separable target with the L-lane reduction shape:
 * `first` elements are handled by the caller-supplied head (lane 0, block 0). */
static float butterfly(float *p, int L)
{
    float tmp[64];
    for (int off = 1; off < L; off <<= 1) {
        for (int l = 0; l < L; ++l) tmp[l] = p[l] + p[l ^ off];
        memcpy(p, tmp, sizeof(float) * (size_t)L);
    }
    return p[0];
}

static float split_sum_squares(const orc_target *t, const float *x)
{
    const int d = t->dim, L = t->reduce_lanes, nblk = (d + 3) / 4;
    float p[64];
    for (int l = 0; l < L; ++l) {
        float q = 0.0f;
        for (int b = l; b < nblk; b += L)
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k >= d) break;
                if (t->kind == ORC_TARGET_BANANA && k == 0) { q = (x[0] * x[0]) * 0.01f; continue; }
                if (t->kind == ORC_TARGET_BANANA && k == 1) {
                    float u = fmaf(t->params[0], fmaf(x[0], x[0], -100.0f), x[1]);
                    q = fmaf(u, u, q);
                    continue;
                }
                if (t->kind == ORC_TARGET_FUNNEL && k == 0) continue;
                q = fmaf(x[k], x[k], q);
            }
        p[l] = q;
    }
    return butterfly(p, L);
}

float orc_target_eval(const orc_target *t, const float *x)
{
    const int d = t->dim;
    if (t->reduce_lanes > 1 && t->kind == ORC_TARGET_CORR_GAUSS) {
        /* rows i = l, l+L, ... on lane l (each row-dot sequential), partial sums of squares in a butterfly */
        const float *A = t->params;
        const int L = t->reduce_lanes;
        float p[64];
        for (int l = 0; l < L; ++l) {
            float q = 0.0f;
            for (int i = l; i < d; i += L) {
                const size_t off = (size_t)i * ((size_t)i + 1) / 2;
                float w = 0.0f;
                for (int j = 0; j <= i; ++j) w = fmaf(A[off + j], x[j], w);
                q = fmaf(w, w, q);
            }
            p[l] = q;
        }
        return fmaf(-0.5f, butterfly(p, L), target_const(t));
    }
    if (t->reduce_lanes > 1 && (t->kind == ORC_TARGET_ISO_GAUSS || t->kind == ORC_TARGET_BANANA ||
                                t->kind == ORC_TARGET_FUNNEL)) {
        const float q = split_sum_squares(t, x);
        if (t->kind != ORC_TARGET_FUNNEL) return fmaf(-0.5f, q, target_const(t));
        const float v = x[0];
        float ev = orc_expf(-v);
        float r = (v * v) * 0x1.c71c72p-5f;
        r = fmaf(0.5f * (float)(d - 1), v, r);
        r = fmaf(0.5f * ev, q, r);
        return target_const(t) - r;
    }
    switch (t->kind) {
    case ORC_TARGET_ISO_GAUSS: {                       /* logpdf(MvNormal(zeros(d), I), x) */
        float q = 0.0f;
        for (int k = 0; k < d; ++k) q = fmaf(x[k], x[k], q);
        return fmaf(-0.5f, q, target_const(t));
    }
    case ORC_TARGET_CORR_GAUSS: {                      /* -1/2 |A x|^2 + const, A = inv(chol(Sigma)) */
        const float *A = t->params;
        float q = 0.0f;
        size_t off = 0;
        for (int i = 0; i < d; ++i) {
            float w = 0.0f;
            for (int j = 0; j <= i; ++j) w = fmaf(A[off + j], x[j], w);
            q = fmaf(w, w, q);
            off += (size_t)i + 1;
        }
        return fmaf(-0.5f, q, target_const(t));
    }
    case ORC_TARGET_IID_NORMAL: {                      /* README.md:29-31 / test/runtests.jl:26-28 */
        const float mu = x[0], sigma = x[1];
        if (!(sigma >= 0.0f)) return -INFINITY;        /* insupport(theta) = theta[2] >= 0 */
        if (sigma == 0.0f) return -INFINITY;
        float acc = 0.0f;
        for (int i = 0; i < t->nparams; ++i) {
            float z = (t->params[i] - mu) / sigma;
            acc = fmaf(z, z, acc);
        }
        float nf = (float)t->nparams;
        float tt = orc_logf(sigma) + HALF_LOG_2PI_F;
        return fmaf(-0.5f, acc, -(nf * tt));
    }
    case ORC_TARGET_BANANA: {
        const float b = t->params[0];
        float q = (x[0] * x[0]) * 0.01f;
        float u = fmaf(b, fmaf(x[0], x[0], -100.0f), x[1]);
        q = fmaf(u, u, q);
        for (int k = 2; k < d; ++k) q = fmaf(x[k], x[k], q);
        return fmaf(-0.5f, q, target_const(t));
    }
    case ORC_TARGET_FUNNEL: {
        const float v = x[0];
        float q = 0.0f;
        for (int k = 1; k < d; ++k) q = fmaf(x[k], x[k], q);
        float ev = orc_expf(-v);
        float r = (v * v) * 0x1.c71c72p-5f;             /* 1/18 */
        r = fmaf(0.5f * (float)(d - 1), v, r);
        r = fmaf(0.5f * ev, q, r);
        return target_const(t) - r;
    }
    case ORC_TARGET_CALLBACK:
        return t->fn(x, d, t->fn_data);
    default:
        return NAN;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* proposal draw: xi ~ MvNormal(0, Sigma) = L z   [upstream Distributions rand(MvNormal), restated]
 * reference call sites: src/proposal.jl:24-25 (rand), :41-47 (initial), :49-56 (t + rand).   */
static void propose_from(const orc_proposal *p, int d, const float *z, const float *x, float *y)
{
    const float *mu = p->mean;
    switch (p->kind) {
    case ORC_PROP_ISO:
        for (int k = 0; k < d; ++k) y[k] = mu ? x[k] + fmaf(p->scale, z[k], mu[k]) : fmaf(p->scale, z[k], x[k]);
        break;
    case ORC_PROP_DIAG:
        for (int k = 0; k < d; ++k) y[k] = mu ? x[k] + fmaf(p->vec[k], z[k], mu[k]) : fmaf(p->vec[k], z[k], x[k]);
        break;
    default: {
        size_t off = 0;
        for (int i = 0; i < d; ++i) {
            float w = 0.0f;
            for (int j = 0; j <= i; ++j) w = fmaf(p->vec[off + j], z[j], w);
            y[i] = mu ? x[i] + (mu[i] + w) : x[i] + w;
            off += (size_t)i + 1;
        }
    }
    }
}

/* q(x) = -1/2 |L^-1 (x - mu)|^2: logpdf of the proposal at x up to its constant (src/proposal.jl:31-35);
 * forward substitution, every sum in ascending order */
static float static_logq(const orc_proposal *p, int d, const float *x, float *t)
{
    const float *mu = p->mean;
    size_t off = 0;
    float q = 0.0f;
    for (int i = 0; i < d; ++i) {
        const float r = mu ? x[i] - mu[i] : x[i];
        if (p->kind == ORC_PROP_ISO) t[i] = r / p->scale;
        else if (p->kind == ORC_PROP_DIAG) t[i] = r / p->vec[i];
        else {
            float acc = 0.0f;
            for (int j = 0; j < i; ++j) acc = fmaf(p->vec[off + j], t[j], acc);
            t[i] = (r - acc) / p->vec[off + i];
            off += (size_t)i + 1;
        }
        q = fmaf(t[i], t[i], q);
    }
    return -0.5f * q;
}

/* twice the whitened mean 2 L^-1 mu (host arithmetic in double, rounded once): with it the Hastings ratio of a
 * drifting random walk is  logq(x|y) - logq(y|x) = 1/2 |z|^2 - 1/2 |z + 2 L^-1 mu|^2   (src/proposal.jl:58-64,190-192) */
static void whitened_mean2(const orc_proposal *p, int d, float *tm)
{
    double *m = malloc(sizeof(double) * (size_t)d);
    size_t off = 0;
    for (int i = 0; i < d; ++i) {
        double acc = (double)p->mean[i];
        if (p->kind == ORC_PROP_ISO) m[i] = acc / (double)p->scale;
        else if (p->kind == ORC_PROP_DIAG) m[i] = acc / (double)p->vec[i];
        else {
            for (int j = 0; j < i; ++j) acc -= (double)p->vec[off + j] * m[j];
            m[i] = acc / (double)p->vec[off + i];
            off += (size_t)i + 1;
        }
        tm[i] = (float)(2.0 * m[i]);
    }
    free(m);
}

/* ------------------------------------------------------------------------------------------ */
/* [upstream AbstractMCMC.mcmcsample, restated from memory -- unverifiable here]
 * iteration 1 = initial state; `discard_initial` transitions are dropped (transition j uses
 * step_warmup iff j <= num_warmup); sample i >= 2 is reached after `thinning` transitions, all
 * of which use step_warmup iff i <= num_warmup - min(num_warmup, discard_initial).            */
void orc_schedule_counts(const orc_schedule *s, int64_t *n_transitions, int64_t *n_adapt)
{
    int64_t N = s->n_samples, di = s->discard_initial, th = s->thinning, nw = s->num_warmup;
    int64_t dfw = nw < di ? nw : di;
    int64_t kfw = nw - dfw;
    int64_t k = kfw < N ? kfw : N;
    *n_transitions = di + (N - 1) * th;
    *n_adapt = dfw + (k >= 2 ? (k - 1) * th : 0);
}

/* is transition tau (1-based) the one that produces a saved sample?  returns slot or -1 */
static int64_t save_slot(const orc_schedule *s, int64_t tau)
{
    int64_t r = tau - s->discard_initial;
    if (r < 0 || r % s->thinning) return -1;
    return r / s->thinning;                            /* tau = discard_initial -> slot 0 */
}

static void record(float *samples, uint8_t *accepted, int64_t slot, int d, int C, int c,
                   const float *x, float lp, int acc)
{
    if (samples) {
        float *row = samples + (size_t)slot * (size_t)(d + 1) * (size_t)C;
        for (int k = 0; k < d; ++k) row[(size_t)k * C + c] = x[k];
        row[(size_t)d * C + c] = lp;
    }
    if (accepted) accepted[(size_t)slot * C + c] = (uint8_t)acc;
}

/* ------------------------------------------------------------------------------------------ */
/* RWMH: src/mh-core.jl:76-86 (initial step) and :92-117 (step)                               */
int orc_rwmh(const orc_target *t, const orc_proposal *p, const orc_schedule *s,
             uint64_t seed, uint64_t first_chain, int nchains,
             const float *init, float *samples, uint8_t *accepted,
             float *final_x, float *final_lp, uint32_t *accept_counts)
{
    const int d = t->dim, C = nchains;
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    float *x = malloc(sizeof(float) * (size_t)d * 5);
    float *y = x + d, *z = y + d, *tm = z + d, *zero = tm + d;
    if (p->mean && !p->is_static) whitened_mean2(p, d, tm);
    for (int k = 0; k < d; ++k) zero[k] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const uint64_t id = first_chain + (uint64_t)c;
        /* mh-core.jl:83  params = initial_params === nothing ? propose(rng, sampler, model) : initial_params
         * proposal.jl:41-47: the initial propose is a bare draw from the proposal (x = 0 + xi). */
        if (init) {
            for (int k = 0; k < d; ++k) x[k] = init[(size_t)k * C + c];
        } else {
            orc_normals(seed, id, 0, ORC_STREAM_INIT, d, z);
            for (int k = 0; k < d; ++k) y[k] = 0.0f;
            propose_from(p, d, z, y, x);
        }
        float lp = orc_target_eval(t, x);               /* mh-core.jl:84 transition(..., false) */
        float qx = p->is_static ? static_logq(p, d, x, y) : 0.0f;
        uint32_t nacc = 0;
        int64_t slot = save_slot(s, 0);
        if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, 0);
        for (int64_t tau = 1; tau <= nT; ++tau) {
            const uint32_t step = (uint32_t)tau;
            orc_normals(seed, id, step, ORC_STREAM_PROPOSAL, d, z);
            propose_from(p, d, z, p->is_static ? zero : x, y);   /* mh-core.jl:100; static: proposal.jl:66-72 */
            float lpy = orc_target_eval(t, y);          /* :103 */
            float loga = lpy - lp;                      /* :104-105, Hastings ratio of a zero-mean RW == 0 */
            float qy = 0.0f;
            if (p->is_static) {                         /* proposal.jl:74-83: q = logpdf(proposal, t) */
                float fwd = 0.0f;
                for (int k = 0; k < d; ++k) fwd = fmaf(z[k], z[k], fwd);
                qy = -0.5f * fwd;
                loga = (lpy - lp) + (qx - qy);
            } else if (p->mean) {                       /* :105,119-123 -> proposal.jl:190-192 */
                float fwd = 0.0f, bwd = 0.0f;
                for (int k = 0; k < d; ++k) {
                    fwd = fmaf(z[k], z[k], fwd);
                    const float tk = z[k] + tm[k];
                    bwd = fmaf(tk, tk, bwd);
                }
                loga = (lpy - lp) + 0.5f * (fwd - bwd);
            }
            float logu = orc_accept_logu(seed, id, step);
            int acc = logu < loga;                      /* :108  -randexp(rng) < loga (strict; NaN -> reject) */
            if (acc) { memcpy(x, y, sizeof(float) * (size_t)d); lp = lpy; qx = qy; ++nacc; }
            slot = save_slot(s, tau);
            if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, acc);
        }
        if (final_x) for (int k = 0; k < d; ++k) final_x[(size_t)k * C + c] = x[k];
        if (final_lp) final_lp[c] = lp;
        if (accept_counts) accept_counts[c] = nacc;
    }
    free(x);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* emcee: src/emcee.jl:14-24 (step), :39-58 (sweep), :70-102 (stretch move)                   */
static int stretch_move(const orc_target *t, float a, uint64_t seed, uint64_t ens, int i,
                        uint32_t sweep, int other_start, int other_size, int wrap_W,
                        const float *cur, const float *oth_new, const float *oth_old, int use_seq,
                        int W, float *xi, float *lpi, float *y, float *xj)
{
    const int d = t->dim;
    uint32_t w[4];
    uint32_t ctr[4] = { (uint32_t)i, (uint32_t)ens, sweep, (uint32_t)ORC_STREAM_EMCEE << 28 };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(ctr, key, w);
    int j;
    if (use_seq) {
        /* emcee.jl:48,52  idx = mod1(i + rand(1:W-1), W)  (never i) */
        uint32_t r = 1u + (uint32_t)(((uint64_t)w[0] * (uint64_t)(W - 1)) >> 32);
        j = (int)(((uint64_t)i + r) % (uint64_t)wrap_W);
        /* emcee.jl:53  other = idx < i ? new_walkers[idx] : walkers[idx] */
        const float *src = (j < i) ? oth_new : oth_old;
        for (int k = 0; k < d; ++k) xj[k] = src[(size_t)k * W + j];
    } else {
        j = other_start + (int)(((uint64_t)w[0] * (uint64_t)other_size) >> 32);
        for (int k = 0; k < d; ++k) xj[k] = cur[(size_t)k * W + j];
    }
    /* emcee.jl:81  z = ((a - 1) * rand(rng) + 1)^2 / a */
    float u = orc_u01_half(w[1]);
    float tt = fmaf(a - 1.0f, u, 1.0f);
    float z = (tt * tt) / a;
    float alphamult = (float)(d - 1) * orc_logf(z);     /* :82 */
    for (int k = 0; k < d; ++k) y[k] = fmaf(z, xi[k] - xj[k], xj[k]);   /* :85 */
    float lpy = orc_target_eval(t, y);                  /* :88 */
    float alpha = (alphamult + lpy) - *lpi;             /* :91 */
    float logu = orc_logf(orc_u01_open(w[2]));
    int acc = logu <= alpha;                            /* :93  -randexp <= alpha (non-strict) */
    if (acc) { memcpy(xi, y, sizeof(float) * (size_t)d); *lpi = lpy; }
    return acc;
}

int orc_emcee(const orc_target *t, float a, int mode, const orc_schedule *s,
              uint64_t seed, uint64_t ensemble_id, int nwalkers,
              const float *init, float *samples, uint8_t *accepted,
              float *final_x, float *final_lp, uint32_t *accept_counts)
{
    const int d = t->dim, W = nwalkers;
    if (!init || W < 2) return -1;
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    float *cur = malloc(sizeof(float) * (size_t)d * W);
    float *nxt = malloc(sizeof(float) * (size_t)d * W);
    float *lp = malloc(sizeof(float) * (size_t)W);
    float *lpn = malloc(sizeof(float) * (size_t)W);
    uint8_t *acc = malloc((size_t)W);
    float *tmp = malloc(sizeof(float) * (size_t)d * 3);
    float *xi = tmp, *y = tmp + d, *xj = tmp + 2 * d;
    memcpy(cur, init, sizeof(float) * (size_t)d * W);
    for (int i = 0; i < W; ++i) {                       /* emcee.jl:6-8: W log-density evaluations */
        for (int k = 0; k < d; ++k) xi[k] = cur[(size_t)k * W + i];
        lp[i] = orc_target_eval(t, xi);
        if (accept_counts) accept_counts[i] = 0;
    }
    int64_t slot = save_slot(s, 0);
    if (slot >= 0)
        for (int i = 0; i < W; ++i) {
            for (int k = 0; k < d; ++k) xi[k] = cur[(size_t)k * W + i];
            record(samples, accepted, slot, d, W, i, xi, lp[i], 0);
        }
    const int half = W / 2;
    for (int64_t tau = 1; tau <= nT; ++tau) {
        const uint32_t sweep = (uint32_t)tau;
        if (mode == 0) {
            /* reference-faithful Gauss-Seidel sweep, emcee.jl:50-55 */
            for (int i = 0; i < W; ++i) {
                for (int k = 0; k < d; ++k) xi[k] = cur[(size_t)k * W + i];
                float l = lp[i];
                acc[i] = (uint8_t)stretch_move(t, a, seed, ensemble_id, i, sweep, 0, 0, W, cur, nxt, cur,
                                               1, W, xi, &l, y, xj);
                for (int k = 0; k < d; ++k) nxt[(size_t)k * W + i] = xi[k];
                lpn[i] = l;
            }
            float *sw = cur; cur = nxt; nxt = sw;
            sw = lp; lp = lpn; lpn = sw;
        } else {
            /* parallel split: half 0 = [0, W/2) moves against half 1, then half 1 against updated half 0 */
            for (int h = 0; h < 2; ++h) {
                const int lo = h ? half : 0, hi = h ? W : half;
                const int ostart = h ? 0 : half, osize = h ? half : W - half;
                for (int i = lo; i < hi; ++i) {
                    for (int k = 0; k < d; ++k) xi[k] = cur[(size_t)k * W + i];
                    float l = lp[i];
                    acc[i] = (uint8_t)stretch_move(t, a, seed, ensemble_id, i, sweep, ostart, osize, W, cur,
                                                   NULL, NULL, 0, W, xi, &l, y, xj);
                    /* partners come from the other half only, so in-place update is race-free */
                    for (int k = 0; k < d; ++k) cur[(size_t)k * W + i] = xi[k];
                    lp[i] = l;
                }
            }
        }
        for (int i = 0; i < W; ++i) if (acc[i] && accept_counts) accept_counts[i]++;
        slot = save_slot(s, tau);
        if (slot >= 0)
            for (int i = 0; i < W; ++i) {
                for (int k = 0; k < d; ++k) xi[k] = cur[(size_t)k * W + i];
                record(samples, accepted, slot, d, W, i, xi, lp[i], acc[i]);
            }
    }
    if (final_x) memcpy(final_x, cur, sizeof(float) * (size_t)d * W);
    if (final_lp) memcpy(final_lp, lp, sizeof(float) * (size_t)W);
    free(cur); free(nxt); free(lp); free(lpn); free(acc); free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* rank-1 Cholesky update / downdate of a packed lower factor
 * [upstream LinearAlgebra.lowrankupdate!/lowrankdowndate!, restated]; call site
 * src/RobustAdaptiveMetropolis.jl:165-171.  Column sweep: for each column i a Givens-type
 * rotation (c, s) from (S_ii, w_i) is applied to the sub-diagonal column and to w[i+1:].     */
#define SIDX(i, j) ((size_t)(i) * ((size_t)(i) + 1) / 2 + (size_t)(j))
int orc_chol_rank1(float *S, float *w, int d, int sign)
{
    /* One sweep for both signs (the textbook rank-one modification; for sigma = -1 it is the upstream
     * lowrankdowndate! loop, for sigma = +1 it equals the upstream Givens form algebraically):
     *   s = w_i / S_ii,  c = sqrt(1 + sigma s^2),  S_ii <- c S_ii,
     *   S_ji <- (S_ji + sigma s w_j) / c,  w_j <- c w_j - s S_ji(new)            (spec 3.9) */
    const float sg = sign > 0 ? 1.0f : -1.0f;
    for (int i = 0; i < d; ++i) {
        const float a = S[SIDX(i, i)], b = w[i];
        const float sn = b / a;
        if (sign < 0 && sn * sn > 1.0f) return i + 1;   /* PosDefException(i) upstream */
        const float ss = sg * sn;
        const float c = sqrtf(fmaf(ss, sn, 1.0f));
        const float rc = 1.0f / c;                        /* one reciprocal per column */
        S[SIDX(i, i)] = c * a;
        for (int j = i + 1; j < d; ++j) {
            const float vj = w[j];
            const float Aji = fmaf(ss, vj, S[SIDX(j, i)]) * rc;
            S[SIDX(j, i)] = Aji;
            w[j] = fmaf(c, vj, -(sn * Aji));
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* RAM: src/RobustAdaptiveMetropolis.jl:175-214 (initial step), :123-151 (ram_step_inner),
 * :153-173 (ram_adapt), :216-237 (step), :239-245 (valid_eigenvalues), :247-278 (step_warmup) */
int orc_ram(const orc_target *t, const orc_ram_cfg *cfg, const orc_schedule *s,
            uint64_t seed, uint64_t first_chain, int nchains,
            const float *init, const float *S_in, float *S_out,
            float *samples, uint8_t *accepted, float *final_x, float *final_lp,
            uint32_t *accept_counts, uint8_t *status, float *diag_min, float *diag_max)
{
    const int d = t->dim, C = nchains;
    const size_t nS = (size_t)d * ((size_t)d + 1) / 2;
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    float *x = malloc(sizeof(float) * (size_t)d * 5);
    float *y = x + d, *U = y + d, *v = U + d, *w = v + d;
    float *S = malloc(sizeof(float) * nS), *Sn = malloc(sizeof(float) * nS);
    const int default_bounds = (cfg->eig_lo == 0.0f && isinf(cfg->eig_hi) && cfg->eig_hi > 0);
    for (int c = 0; c < C; ++c) {
        const uint64_t id = first_chain + (uint64_t)c;
        uint8_t st = 0;
        if (init) for (int k = 0; k < d; ++k) x[k] = init[(size_t)k * C + c];
        else orc_normals(seed, id, 0, ORC_STREAM_INIT, d, x);      /* :193 randn(rng, T, d) */
        if (S_in) memcpy(S, S_in + (size_t)c * nS, sizeof(float) * nS);
        else { memset(S, 0, sizeof(float) * nS); for (int i = 0; i < d; ++i) S[SIDX(i, i)] = 1.0f; }
        float lp = orc_target_eval(t, x);                          /* :210 */
        uint32_t nacc = 0;
        if (diag_min) for (int k = 0; k < d; ++k) {
            diag_min[(size_t)k * C + c] = S[SIDX(k, k)];
            diag_max[(size_t)k * C + c] = S[SIDX(k, k)];
        }
        int64_t slot = save_slot(s, 0);
        if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, 1);   /* :213 Transition(x, lp, true) */
        for (int64_t tau = 1; tau <= nT; ++tau) {
            const uint32_t step = (uint32_t)tau;                   /* == state.iteration */
            orc_normals(seed, id, step, ORC_STREAM_PROPOSAL, d, U);            /* :135 */
            for (int i = 0; i < d; ++i) {                                       /* :136 muladd(S, U, x) */
                float acc = 0.0f;
                for (int j = 0; j <= i; ++j) acc = fmaf(S[SIDX(i, j)], U[j], acc);
                v[i] = acc;
                y[i] = acc + x[i];
            }
            float lpy = orc_target_eval(t, y);                     /* :140 */
            float diff = lpy - lp;
            float loga = (diff != diff) ? diff : (diff < 0.0f ? diff : 0.0f);   /* :147 min(lp_new - lp, 0) */
            float logu = orc_accept_logu(seed, id, step);
            int acc = logu < loga;                                  /* :148 randexp(rng) > -loga */
            if (tau <= nA) {                                        /* step_warmup: adapt, :153-173 */
                float da = orc_expf(loga) - cfg->alpha;             /* :159 */
                if (da == da) {
                    float eta = (float)pow((double)step, -(double)cfg->gamma);   /* :162 */
                    float nn = 0.0f;
                    for (int j = 0; j < d; ++j) nn = fmaf(U[j], U[j], nn);
                    float coef = sqrtf(eta * fabsf(da)) / sqrtf(nn);             /* :163 */
                    for (int j = 0; j < d; ++j) w[j] = v[j] * coef;
                    memcpy(Sn, S, sizeof(float) * nS);
                    int fail = orc_chol_rank1(Sn, w, d, da > 0.0f ? +1 : -1);   /* :165-171 */
                    int ok = !fail;
                    if (fail) st |= 1;
                    if (ok && !default_bounds)                                  /* :239-245, :259-264 */
                        for (int k = 0; k < d; ++k) {
                            float e = Sn[SIDX(k, k)];
                            if (!(cfg->eig_lo <= e && e <= cfg->eig_hi)) { ok = 0; break; }
                        }
                    if (ok) { float *sw = S; S = Sn; Sn = sw; }
                } else {
                    st |= 2;                                        /* NaN log-ratio: adaptation skipped */
                }
                if (diag_min) for (int k = 0; k < d; ++k) {
                    float e = S[SIDX(k, k)];
                    if (e < diag_min[(size_t)k * C + c]) diag_min[(size_t)k * C + c] = e;
                    if (e > diag_max[(size_t)k * C + c]) diag_max[(size_t)k * C + c] = e;
                }
            }
            if (acc) { memcpy(x, y, sizeof(float) * (size_t)d); lp = lpy; ++nacc; }   /* :267-277 */
            slot = save_slot(s, tau);
            if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, acc);
        }
        if (final_x) for (int k = 0; k < d; ++k) final_x[(size_t)k * C + c] = x[k];
        if (final_lp) final_lp[c] = lp;
        if (accept_counts) accept_counts[c] = nacc;
        if (status) status[c] = st;
        if (S_out) memcpy(S_out + (size_t)c * nS, S, sizeof(float) * nS);
    }
    free(x); free(S); free(Sn);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* gradients of the catalogue targets (what ForwardDiff / LogDensityProblems.logdensity_and_gradient
 * supply to the reference's MALA, src/MALA.jl:73-75, ext/AdvancedMHForwardDiffExt.jl:13-17)     */
float orc_target_grad(const orc_target *t, const float *x, float *g, orc_logdensity_grad_fn user)
{
    const int d = t->dim;
    switch (t->kind) {
    case ORC_TARGET_ISO_GAUSS:
        for (int k = 0; k < d; ++k) g[k] = -x[k];
        return orc_target_eval(t, x);
    case ORC_TARGET_CORR_GAUSS: {                      /* grad = -A^T (A x) */
        const float *A = t->params;
        float q = 0.0f;
        size_t off = 0;
        for (int i = 0; i < d; ++i) {                  /* w = A x, kept in g */
            float w = 0.0f;
            for (int j = 0; j <= i; ++j) w = fmaf(A[off + j], x[j], w);
            g[i] = w;
            q = fmaf(w, w, q);
            off += (size_t)i + 1;
        }
        for (int j = 0; j < d; ++j) {                  /* g_j = -sum_{i>=j} A_ij w_i, ascending i, in place */
            float acc = 0.0f;
            for (int i = j; i < d; ++i) acc = fmaf(A[SIDX(i, j)], g[i], acc);
            g[j] = -acc;
        }
        return fmaf(-0.5f, q, target_const(t));
    }
    case ORC_TARGET_IID_NORMAL: {
        const float mu = x[0], sigma = x[1];
        if (!(sigma > 0.0f)) { g[0] = 0.0f; g[1] = 0.0f; return -INFINITY; }
        const float inv = 1.0f / sigma;
        float acc = 0.0f, s1 = 0.0f;
        for (int i = 0; i < t->nparams; ++i) {
            const float z = (t->params[i] - mu) / sigma;
            acc = fmaf(z, z, acc);
            s1 = s1 + z;
        }
        const float nf = (float)t->nparams;
        g[0] = s1 * inv;                               /* sum (y-mu)/sigma^2 */
        g[1] = (acc - nf) * inv;                       /* -n/sigma + sum (y-mu)^2/sigma^3 */
        const float tt = orc_logf(sigma) + HALF_LOG_2PI_F;
        return fmaf(-0.5f, acc, -(nf * tt));
    }
    case ORC_TARGET_BANANA: {
        const float b = t->params[0];
        const float x0 = x[0];
        float q = (x0 * x0) * 0.01f;
        const float u = fmaf(b, fmaf(x0, x0, -100.0f), x[1]);
        q = fmaf(u, u, q);
        g[0] = -(fmaf(x0, 0.01f, (2.0f * b) * (u * x0)));
        g[1] = -u;
        for (int k = 2; k < d; ++k) { q = fmaf(x[k], x[k], q); g[k] = -x[k]; }
        return fmaf(-0.5f, q, target_const(t));
    }
    case ORC_TARGET_FUNNEL: {
        const float v = x[0];
        float q = 0.0f;
        for (int k = 1; k < d; ++k) q = fmaf(x[k], x[k], q);
        const float ev = orc_expf(-v);
        float r = (v * v) * 0x1.c71c72p-5f;
        r = fmaf(0.5f * (float)(d - 1), v, r);
        r = fmaf(0.5f * ev, q, r);
        g[0] = fmaf(0.5f * ev, q, -(fmaf(v, 0x1.c71c72p-4f, 0.5f * (float)(d - 1))));   /* -v/9 - (d-1)/2 + e^-v q/2 */
        for (int k = 1; k < d; ++k) g[k] = -(ev * x[k]);
        return target_const(t) - r;
    }
    case ORC_TARGET_CALLBACK:
        return user ? user(x, g, d, t->fn_data) : NAN;
    default:
        return NAN;
    }
}

/* MALA: src/MALA.jl:54-93 with the standard Langevin proposal g -> MvNormal((sigma2/2) g, sigma2 I)
 * (the form of the reference's tests, test/runtests.jl:291,352):
 *   y = x + (sigma2/2) grad(x) + sigma z                                     (:70, proposal.jl:49-56)
 *   logratio = q(prop(grad y), x, y) - q(prop(grad x), y, x)                 (:78-80)
 *            = 1/2 |z|^2 - 1/2 |z + (sigma/2)(grad x + grad y)|^2
 *   accept iff -randexp < lp(y) - lp(x) + logratio                           (:83-86)            */
int orc_mala(const orc_target *t, orc_logdensity_grad_fn user, float sigma2, const orc_schedule *s,
             uint64_t seed, uint64_t first_chain, int nchains, const float *init,
             float *samples, uint8_t *accepted, float *final_x, float *final_lp, uint32_t *accept_counts)
{
    const int d = t->dim, C = nchains;
    if (!init) return -1;                                /* :37 "please specify initial parameters" */
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    const float sigma = sqrtf(sigma2);
    const float h = (sigma * sigma) * 0.5f;              /* drift step sigma2/2 */
    const float hs = 0.5f * sigma;
    float *x = malloc(sizeof(float) * (size_t)d * 5);
    float *gx = x + d, *y = gx + d, *gy = y + d, *z = gy + d;
    for (int c = 0; c < C; ++c) {
        const uint64_t id = first_chain + (uint64_t)c;
        for (int k = 0; k < d; ++k) x[k] = init[(size_t)k * C + c];
        float lp = orc_target_grad(t, x, gx, user);      /* :38-40 GradientTransition(params, lp, grad, false) */
        uint32_t nacc = 0;
        int64_t slot = save_slot(s, 0);
        if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, 0);
        for (int64_t tau = 1; tau <= nT; ++tau) {
            const uint32_t step = (uint32_t)tau;
            orc_normals(seed, id, step, ORC_STREAM_PROPOSAL, d, z);
            float fwd = 0.0f;
            for (int k = 0; k < d; ++k) {
                y[k] = fmaf(sigma, z[k], fmaf(h, gx[k], x[k]));
                fwd = fmaf(z[k], z[k], fwd);
            }
            const float lpy = orc_target_grad(t, y, gy, user);
            float bwd = 0.0f;
            for (int k = 0; k < d; ++k) {
                const float tk = fmaf(hs, gx[k] + gy[k], z[k]);
                bwd = fmaf(tk, tk, bwd);
            }
            const float loga = (lpy - lp) + 0.5f * (fwd - bwd);
            const float logu = orc_accept_logu(seed, id, step);
            const int acc = logu < loga;
            if (acc) { memcpy(x, y, sizeof(float) * (size_t)d); memcpy(gx, gy, sizeof(float) * (size_t)d); lp = lpy; ++nacc; }
            slot = save_slot(s, tau);
            if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, acc);
        }
        if (final_x) for (int k = 0; k < d; ++k) final_x[(size_t)k * C + c] = x[k];
        if (final_lp) final_lp[c] = lp;
        if (accept_counts) accept_counts[c] = nacc;
    }
    free(x);
    return 0;
}
