/*
 * mhx_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See mhx_oracle.h.
 *
 * Scalar, one chain at a time, written against the arithmetic spec of DESIGN.md section 3 so that it is
 * bit-comparable with the HIP kernels.  One source, two builds (oracle/Makefile): ORC_F64=0 -> libmhx_oracle.so
 * (`real` = float, the fp32 engine) and ORC_F64=1 -> libmhx_oracle64.so (`real` = double: the reference computes in
 * Float64 end to end -- Distributions' rand / logpdf, src/RobustAdaptiveMetropolis.jl:187-196 `T = eltype(sampler.gamma)`).
 * gcc, -ffp-contract=off: every fused multiply-add below is an explicit FMA, every other operation rounds separately.
 */
#include "mhx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#if ORC_F64
#define R(x) x
#define FMA(a, b, c) fma((a), (b), (c))
#define SQRT(x) sqrt(x)
#define FABS(x) fabs(x)
#define RINT(x) rint(x)
#else
#define R(x) x##f
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(x) sqrtf(x)
#define FABS(x) fabsf(x)
#define RINT(x) rintf(x)
#endif

/* ------------------------------------------------------------------------------------------ */
/* bit casts                                                                                  */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double   u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

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

#if !ORC_F64
/* ------------------------------------------------------------------------------------------ */
/* fp32 transcendental spec (coefficients: tools/fit_coeffs.py)                               */
#define LN2_HI 0x1.62e4p-1f           /* 16 significant bits: e*LN2_HI is exact for |e| < 256 */
#define LN2_LO 0x1.7f7d1cp-20f        /* ln2 - LN2_HI */
#define LOG2E  0x1.715476p+0f
#define HALF_LOG_2PI 0x1.d67f1cp-1f
#define ONE_18 0x1.c71c72p-5f
#define ONE_9  0x1.c71c72p-4f

float orc_log(float x)
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

float orc_exp(float x)
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
    float l = orc_log(orc_u01_open(k0));               /* <= 0 */
    float rad = sqrtf(-2.0f * l);
    float s, c;
    orc_sincos2pi_u32(k1, &s, &c);
    *n0 = rad * c;
    *n1 = rad * s;
}

/* normals 4b .. 4b+3 of (seed, chain, step, stream): one Philox block */
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
    return orc_log(orc_u01_open(w[step & 3]));
}

/* the draws of one stretch move (src/emcee.jl:48,52 partner, :81 stretch uniform, :93 accept): one Philox block */
static void emcee_draws(uint64_t seed, uint64_t ens, int i, uint32_t sweep, uint32_t *partner_word, float *u, float *logu)
{
    uint32_t w[4];
    philox_at(seed, ((uint64_t)ens << 32) | (uint32_t)i, sweep, ORC_STREAM_EMCEE, 0, w);
    *partner_word = w[0];
    *u = orc_u01_half(w[1]);
    *logu = orc_log(orc_u01_open(w[2]));
}

#else /* ORC_F64 */
/* ------------------------------------------------------------------------------------------ */
/* fp64 transcendental spec (coefficients: tools/fit_coeffs64.py)                             */
#define LN2_HI 0x1.62e42feep-1                  /* 32 significant bits: e*LN2_HI is exact for |e| < 2^20 */
#define LN2_LO 0x1.a39ef35793c76p-33            /* ln2 - LN2_HI */
#define LOG2E  0x1.71547652b82fep+0
#define HALF_LOG_2PI 0x1.d67f1c864beb5p-1
#define ONE_18 0x1.c71c71c71c71cp-5
#define ONE_9  0x1.c71c71c71c71cp-4

/* m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), z = s^2:
 *   log(1 + f) = log((1+s)/(1-s)) = 2 s + s z P(z) = f - hfsq + s (hfsq + z P(z)),  hfsq = f^2 / 2
 * (the classical argument reduction; one correctly rounded division) */
double orc_log(double x)
{
    uint64_t ix = d2u(x);
    int eadj = 0;
    if ((ix << 1) == 0) return -INFINITY;
    if (ix >> 63) return NAN;
    if (ix >= 0x7ff0000000000000ull) return x;
    if (ix < 0x0010000000000000ull) { x = x * 0x1p54; ix = d2u(x); eadj = -54; }
    uint64_t t = ix - 0x3fe6a09e667f3bcdull;           /* bits(sqrt(1/2)) */
    int64_t e = (int64_t)t >> 52;
    double m = u2d(ix - ((uint64_t)e << 52));
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 0x1.2b59b70eb76c6p-3;
    p = fma(p, z, 0x1.39fe42e9d4a8ap-3);
    p = fma(p, z, 0x1.7462b58e4403ap-3);
    p = fma(p, z, 0x1.c71c62e26212ep-3);
    p = fma(p, z, 0x1.2492492df3ba9p-2);
    p = fma(p, z, 0x1.99999999952ccp-2);
    p = fma(p, z, 0x1.5555555555558p-1);
    double hfsq = (0.5 * f) * f;
    double ef = (double)(e + eadj);
    double t1 = s * fma(z, p, hfsq);
    double t2 = fma(ef, LN2_LO, t1);
    double t3 = hfsq - t2;
    double t4 = f - t3;
    return fma(ef, LN2_HI, t4);
}

double orc_exp(double x)
{
    if (x != x) return x;
    if (x > 0x1.62e42fefa39efp+9) return INFINITY;      /* > 709.78 overflows */
    if (x < -0x1.74910d52d3052p+9) return 0.0;          /* < -745.13 rounds to 0 */
    double n = rint(x * LOG2E);
    double r = fma(n, -LN2_HI, x);
    r = fma(n, -LN2_LO, r);
    double p = 0x1.61bfaa228dde5p-33;
    p = fma(p, r, 0x1.1f7f2776cfaf2p-29);
    p = fma(p, r, 0x1.ae642c82e33d5p-26);
    p = fma(p, r, 0x1.27e4d41966f2fp-22);
    p = fma(p, r, 0x1.71de3a5aa7bb7p-19);
    p = fma(p, r, 0x1.a01a01a9e991bp-16);
    p = fma(p, r, 0x1.a01a01a0196acp-13);
    p = fma(p, r, 0x1.6c16c16c15a68p-10);
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555557p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0.5);
    double r2 = r * r;
    double y = fma(r2, p, r) + 1.0;
    int ni = (int)n;
    int n1 = ni / 2;
    int n2 = ni - n1;
    y = y * u2d((uint64_t)(n1 + 1023) << 52);
    y = y * u2d((uint64_t)(n2 + 1023) << 52);
    return y;
}

/* angle = 2 pi a / 2^64, a = hi:lo.  Integer quadrant reduction; the residual keeps 52 bits so that it is exact in
 * a double: r = ((ri >> 10) as double) * 2^-54 turns, [-1/8, 1/8) */
void orc_sincos2pi_u64(uint32_t hi, uint32_t lo, double *s, double *c)
{
    uint64_t a = ((uint64_t)hi << 32) | lo;
    uint64_t kk = a + 0x2000000000000000ull;           /* wraps mod 2^64 */
    unsigned q = (unsigned)(kk >> 62);
    int64_t ri = (int64_t)(kk & 0x3fffffffffffffffull) - 0x2000000000000000ll;   /* [-2^61, 2^61) */
    int64_t ti = ri >> 10;                             /* arithmetic shift: [-2^51, 2^51) */
    double r = (double)ti * 0x1p-54;
    double u = r * r;
    /* sin(2 pi r) = 2 pi r + r u S1(u), 2 pi = HI + LO;  cos(2 pi r) = 1 - 2 pi^2 u + u^2 C2(u), -2 pi^2 = HI + LO */
    double s1 = -0x1.6cc577dadd922p-1;
    s1 = fma(s1, u, 0x1.e8f036bcd3237p+1);
    s1 = fma(s1, u, -0x1.e3074d2614b2dp+3);
    s1 = fma(s1, u, 0x1.50783486facaap+5);
    s1 = fma(s1, u, -0x1.32d2cce62b872p+6);
    s1 = fma(s1, u, 0x1.466bc6775aae1p+6);
    s1 = fma(s1, u, -0x1.4abbce625be53p+5);
    double ts = (r * u) * s1;
    double sp = fma(r, 0x1.921fb54442d18p+2, fma(r, 0x1.1a62633145c07p-52, ts));
    double c2 = 0x1.1ebe62242e9d8p-2;
    c2 = fma(c2, u, -0x1.b6df855cc99ffp+0);
    c2 = fma(c2, u, 0x1.f9d38850e5eedp+2);
    c2 = fma(c2, u, -0x1.a6d1f2a15a701p+4);
    c2 = fma(c2, u, 0x1.e1f506891b72fp+5);
    c2 = fma(c2, u, -0x1.55d3c7e3cbffap+6);
    c2 = fma(c2, u, 0x1.03c1f081b5ac4p+6);
    double wc = (u * u) * c2;
    double vc = fma(u, -0x1.692b71366cc04p-50, wc);
    double ac = fma(u, -0x1.3bd3cc9be45dep+4, 1.0);             /* in [0.69, 1]: 1 - ac is exact */
    double ec = fma(u, -0x1.3bd3cc9be45dep+4, 1.0 - ac);        /* what the rounding of ac dropped */
    double cp = ac + (vc + ec);
    double ss = (q & 1) ? cp : sp;
    double cc = (q & 1) ? sp : cp;
    if (q == 2 || q == 3) ss = -ss;
    if (q == 1 || q == 2) cc = -cc;
    *s = ss; *c = cc;
}

/* 52-bit uniforms from two Philox words: k = hi:lo >> 12 */
double orc_u01_open(uint32_t hi, uint32_t lo)          /* (0,1): (k + 1/2) 2^-52, exact */
{
    uint64_t k = ((uint64_t)hi << 20) | (lo >> 12);
    return fma((double)k, 0x1p-52, 0x1p-53);
}
double orc_u01_half(uint32_t hi, uint32_t lo)          /* [0,1): k 2^-52 */
{
    uint64_t k = ((uint64_t)hi << 20) | (lo >> 12);
    return (double)k * 0x1p-52;
}

/* Box-Muller from one Philox block: radius from (w0, w1), angle from (w2, w3) */
void orc_normal_pair(const uint32_t w[4], double *n0, double *n1)
{
    double l = orc_log(orc_u01_open(w[0], w[1]));      /* < 0 */
    double rad = sqrt(-2.0 * l);
    double s, c;
    orc_sincos2pi_u64(w[2], w[3], &s, &c);
    *n0 = rad * c;
    *n1 = rad * s;
}

/* normals 4b .. 4b+3 of (seed, chain, step, stream): Philox blocks 2b (normals 4b, 4b+1) and 2b+1 (4b+2, 4b+3) */
void orc_normals(uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, int d, double *out)
{
    for (int b = 0; 4 * b < d; ++b) {
        uint32_t w[4]; double n[4];
        philox_at(seed, chain, step, stream, (uint32_t)(2 * b), w);
        orc_normal_pair(w, &n[0], &n[1]);
        philox_at(seed, chain, step, stream, (uint32_t)(2 * b + 1), w);
        orc_normal_pair(w, &n[2], &n[3]);
        for (int j = 0; j < 4 && 4 * b + j < d; ++j) out[4 * b + j] = n[j];
    }
}

/* log of the accept uniform: the block is shared by 2 consecutive steps */
double orc_accept_logu(uint64_t seed, uint64_t chain, uint32_t step)
{
    uint32_t w[4];
    philox_at(seed, chain, step >> 1, ORC_STREAM_ACCEPT, 0, w);
    return (step & 1) ? orc_log(orc_u01_open(w[2], w[3])) : orc_log(orc_u01_open(w[0], w[1]));
}

/* the draws of one stretch move: block 0 = partner word, stretch uniform (w1, w2); block 1 = accept uniform (w0, w1) */
static void emcee_draws(uint64_t seed, uint64_t ens, int i, uint32_t sweep, uint32_t *partner_word, double *u, double *logu)
{
    uint32_t w[4];
    philox_at(seed, ((uint64_t)ens << 32) | (uint32_t)i, sweep, ORC_STREAM_EMCEE, 0, w);
    *partner_word = w[0];
    *u = orc_u01_half(w[1], w[2]);
    philox_at(seed, ((uint64_t)ens << 32) | (uint32_t)i, sweep, ORC_STREAM_EMCEE, 1, w);
    *logu = orc_log(orc_u01_open(w[0], w[1]));
}
#endif

/* ------------------------------------------------------------------------------------------ */
/* normal generators.  gen 0: Box-Muller (orc_normals).  gen 1: the table ZIGGURAT of spec 3.11 (fp64 here, the fp32 form below) --
 * ORC_ZIG_N equal-area layers under exp(-x^2/2) (mhx_zig_table.h, generated by tools/gen_zig_table.py), 64 bits per normal:
 * Philox block p of (chain, step, stream) serves normals 2p (words hi = 0, lo = 1) and 2p+1 (words 2, 3); layer = lo mod N,
 * sign = bit 31 of lo, u = k 2^-52 with k = (bits 11..30 of lo) : hi, |x| = u x[layer], accepted at once iff |x| < x[layer+1]; otherwise rejection
 * attempts t = 1, 2, ... from block (n << 8 | t) of stream | 4 (n = index of the normal in its step): layer 0 = Marsaglia's
 * tail beyond r, else the wedge test with the next candidate from the same block on rejection.
 * (What Julia's randn does with its own 256-layer table and Xoshiro bits, Random/src/normal.jl -- restated, not copied.) */
#if ORC_F64
#include "mhx_zig_table.h"
static const double zig_x[MHX_ZIG_N + 1] = MHX_ZIG_TABLE;

static int zig_try(uint32_t hi, uint32_t lo, double *x, uint32_t *layer)
{
    *layer = lo & (uint32_t)(MHX_ZIG_N - 1);                          /* bits 0..9 */
    const uint64_t k = ((uint64_t)((lo >> 11) & 0xfffffu) << 32) | hi;   /* 52 bits: (bits 11..30 of lo) : hi */
    const double u = (double)k * 0x1p-52;                              /* [0, 1), exact */
    const double ax = u * zig_x[*layer];
    *x = (lo >> 31) ? -ax : ax;                                         /* bit 31 of lo is the sign */
    return ax < zig_x[*layer + 1];
}

double orc_zig_normal(uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, uint32_t n)
{
    uint32_t w[4], layer;
    double x;
    philox_at(seed, chain, step, stream, n >> 1, w);
    if (zig_try((n & 1u) ? w[2] : w[0], (n & 1u) ? w[3] : w[1], &x, &layer)) return x;
    for (uint32_t t = 1;; ++t) {
        philox_at(seed, chain, step, stream | 4u, (n << 8) | (t & 255u), w);
        if (layer == 0u) {
            const double xx = orc_log(orc_u01_open(w[0], w[1])) * MHX_ZIG_NEG_RINV;
            const double yy = -orc_log(orc_u01_open(w[2], w[3]));
            if (yy + yy >= xx * xx) return signbit(x) ? -(MHX_ZIG_R + xx) : (MHX_ZIG_R + xx);
        } else {
            const double xl = zig_x[layer], xl1 = zig_x[layer + 1], xsq = x * x;
            const double f0 = orc_exp(-0.5 * (xl * xl - xsq)), f1 = orc_exp(-0.5 * (xl1 * xl1 - xsq));
            if (fma(orc_u01_half(w[2], w[3]), f0 - f1, f1) < 1.0) return x;
            if (zig_try(w[0], w[1], &x, &layer)) return x;
        }
    }
}
#endif

#if !ORC_F64
/* The fp32 ziggurat (round 6; spec 3.11, fp32 form): MHX_ZIG32_N = 256 layers, ONE 32-bit word per normal -- Philox block p of
 * (chain, step, stream) serves normals 4p .. 4p+3 from its words 0 .. 3; u = k 2^-23 with the 23-bit k = bits 0..22 (a float's
 * mantissa where it lies), layer = bits 23..30, sign = bit 31, |x| = u x[layer], accepted at once iff |x| < x[layer+1] (98.5 %);
 * otherwise rejection attempts t = 1, 2, ... from block
 * (n << 8 | t) of stream | 4: layer 0 = the tail beyond r (uniforms from words 0 and 2), else the wedge test (uniform from word 2)
 * with the next candidate from word 0 of the same block on rejection. */
#include "mhx_zig_table.h"
static const float zig_x[MHX_ZIG32_N + 1] = MHX_ZIG32_TABLE;

static int zig_try(uint32_t w, float *x, uint32_t *layer)
{
    *layer = (w >> 23) & (uint32_t)(MHX_ZIG32_N - 1);                  /* bits 23..30 */
    const float u = (float)(w & 0x7fffffu) * 0x1p-23f;                  /* bits 0..22: [0, 1), exact */
    const float ax = u * zig_x[*layer];
    *x = (w >> 31) ? -ax : ax;                                          /* bit 31 is the sign */
    return ax < zig_x[*layer + 1];
}

float orc_zig_normal(uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, uint32_t n)
{
    uint32_t w[4], layer;
    float x;
    philox_at(seed, chain, step, stream, n >> 2, w);
    if (zig_try(w[n & 3u], &x, &layer)) return x;
    for (uint32_t t = 1;; ++t) {
        philox_at(seed, chain, step, stream | 4u, (n << 8) | (t & 255u), w);
        if (layer == 0u) {
            const float xx = orc_log(orc_u01_open(w[0])) * MHX_ZIG32_NEG_RINV;
            const float yy = -orc_log(orc_u01_open(w[2]));
            if (yy + yy >= xx * xx) return signbit(x) ? -(MHX_ZIG32_R + xx) : (MHX_ZIG32_R + xx);
        } else {
            const float xl = zig_x[layer], xl1 = zig_x[layer + 1], xsq = x * x;
            const float f0 = orc_exp(-0.5f * (xl * xl - xsq)), f1 = orc_exp(-0.5f * (xl1 * xl1 - xsq));
            if (fmaf(orc_u01_half(w[2]), f0 - f1, f1) < 1.0f) return x;
            if (zig_try(w[0], &x, &layer)) return x;
        }
    }
}
#endif

void orc_normals_gen(int gen, uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, int d, real *out);
static void normals_gen(int gen, uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, int d, real *out)
{
    if (gen == 1) {
        for (int k = 0; k < d; ++k) out[k] = orc_zig_normal(seed, chain, step, stream, (uint32_t)k);
        return;
    }
    orc_normals(seed, chain, step, stream, d, out);
}
void orc_normals_gen(int gen, uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, int d, real *out)
{
    normals_gen(gen, seed, chain, step, stream, d, out);
}

#define LOG_2PI_D 1.8378770664093454835606594728112

/* ------------------------------------------------------------------------------------------ */
/* targets.  reference: logdensity(model, x) = model.logdensity(x), src/AdvancedMH.jl:74      */

static real target_const(const orc_target *t)
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
    return (real)c;
}

/* sum of squares of a separable target with the L-lane reduction shape:
 * `first` elements are handled by the caller-supplied head (lane 0, block 0). */
static real butterfly(real *p, int L)
{
    real tmp[64];
    for (int off = 1; off < L; off <<= 1) {
        for (int l = 0; l < L; ++l) tmp[l] = p[l] + p[l ^ off];
        memcpy(p, tmp, sizeof(real) * (size_t)L);
    }
    return p[0];
}

static real split_sum_squares(const orc_target *t, const real *x)
{
    const int d = t->dim, L = t->reduce_lanes, nblk = (d + 3) / 4;
    real p[64];
    for (int l = 0; l < L; ++l) {
        real q = R(0.0);
        for (int b = l; b < nblk; b += L)
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * b + j;
                if (k >= d) break;
                if (t->kind == ORC_TARGET_BANANA && k == 0) { q = (x[0] * x[0]) * R(0.01); continue; }
                if (t->kind == ORC_TARGET_BANANA && k == 1) {
                    real u = FMA(t->params[0], FMA(x[0], x[0], -R(100.0)), x[1]);
                    q = FMA(u, u, q);
                    continue;
                }
                if (t->kind == ORC_TARGET_FUNNEL && k == 0) continue;
                q = FMA(x[k], x[k], q);
            }
        p[l] = q;
    }
    return butterfly(p, L);
}

real orc_target_eval(const orc_target *t, const real *x)
{
    const int d = t->dim;
    if (t->reduce_lanes > 1 && t->kind == ORC_TARGET_CORR_GAUSS) {
        /* rows i = l, l+L, ... on lane l (each row-dot sequential), partial sums of squares in a butterfly */
        const real *A = t->params;
        const int L = t->reduce_lanes;
        real p[64];
        for (int l = 0; l < L; ++l) {
            real q = R(0.0);
            for (int i = l; i < d; i += L) {
                const size_t off = (size_t)i * ((size_t)i + 1) / 2;
                real w = R(0.0);
                for (int j = 0; j <= i; ++j) w = FMA(A[off + j], x[j], w);
                q = FMA(w, w, q);
            }
            p[l] = q;
        }
        return FMA(-R(0.5), butterfly(p, L), target_const(t));
    }
    if (t->reduce_lanes > 1 && t->kind == ORC_TARGET_IID_NORMAL) {
        /* README.md:29-31 with the sum over the data in reduction shape L: lane l owns the terms i = l, l+L, ... in ascending order,
         * the L partial sums meet in the butterfly (the wave-per-chain kernel: L = 64) */
        const real mu = x[0], sigma = x[1];
        if (!(sigma > R(0.0))) return -INFINITY;
        const int L = t->reduce_lanes;
        real p[64];
        for (int l = 0; l < L; ++l) {
            real q = R(0.0);
            for (int i = l; i < t->nparams; i += L) { const real z = (t->params[i] - mu) / sigma; q = FMA(z, z, q); }
            p[l] = q;
        }
        const real acc = butterfly(p, L);
        const real tt = orc_log(sigma) + HALF_LOG_2PI;
        return FMA(-R(0.5), acc, -((real)t->nparams * tt));
    }
    if (t->reduce_lanes > 1 && (t->kind == ORC_TARGET_ISO_GAUSS || t->kind == ORC_TARGET_BANANA ||
                                t->kind == ORC_TARGET_FUNNEL)) {
        const real q = split_sum_squares(t, x);
        if (t->kind != ORC_TARGET_FUNNEL) return FMA(-R(0.5), q, target_const(t));
        const real v = x[0];
        real ev = orc_exp(-v);
        real r = (v * v) * ONE_18;
        r = FMA(R(0.5) * (real)(d - 1), v, r);
        r = FMA(R(0.5) * ev, q, r);
        return target_const(t) - r;
    }
    switch (t->kind) {
    case ORC_TARGET_ISO_GAUSS: {                       /* logpdf(MvNormal(zeros(d), I), x) */
        real q = R(0.0);
        for (int k = 0; k < d; ++k) q = FMA(x[k], x[k], q);
        return FMA(-R(0.5), q, target_const(t));
    }
    case ORC_TARGET_CORR_GAUSS: {                      /* -1/2 |A x|^2 + const, A = inv(chol(Sigma)) */
        const real *A = t->params;
        real q = R(0.0);
        size_t off = 0;
        for (int i = 0; i < d; ++i) {
            real w = R(0.0);
            for (int j = 0; j <= i; ++j) w = FMA(A[off + j], x[j], w);
            q = FMA(w, w, q);
            off += (size_t)i + 1;
        }
        return FMA(-R(0.5), q, target_const(t));
    }
    case ORC_TARGET_IID_NORMAL: {                      /* README.md:29-31 / test/runtests.jl:26-28 */
        const real mu = x[0], sigma = x[1];
        if (!(sigma >= R(0.0))) return -INFINITY;        /* insupport(theta) = theta[2] >= 0 */
        if (sigma == R(0.0)) return -INFINITY;
        real acc = R(0.0);
        for (int i = 0; i < t->nparams; ++i) {
            real z = (t->params[i] - mu) / sigma;
            acc = FMA(z, z, acc);
        }
        real nf = (real)t->nparams;
        real tt = orc_log(sigma) + HALF_LOG_2PI;
        return FMA(-R(0.5), acc, -(nf * tt));
    }
    case ORC_TARGET_BANANA: {
        const real b = t->params[0];
        real q = (x[0] * x[0]) * R(0.01);
        real u = FMA(b, FMA(x[0], x[0], -R(100.0)), x[1]);
        q = FMA(u, u, q);
        for (int k = 2; k < d; ++k) q = FMA(x[k], x[k], q);
        return FMA(-R(0.5), q, target_const(t));
    }
    case ORC_TARGET_FUNNEL: {
        const real v = x[0];
        real q = R(0.0);
        for (int k = 1; k < d; ++k) q = FMA(x[k], x[k], q);
        real ev = orc_exp(-v);
        real r = (v * v) * ONE_18;             /* 1/18 */
        r = FMA(R(0.5) * (real)(d - 1), v, r);
        r = FMA(R(0.5) * ev, q, r);
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
static void propose_from(const orc_proposal *p, int d, const real *z, const real *x, real *y)
{
    const real *mu = p->mean;
    switch (p->kind) {
    case ORC_PROP_ISO:
        for (int k = 0; k < d; ++k) y[k] = mu ? x[k] + FMA(p->scale, z[k], mu[k]) : FMA(p->scale, z[k], x[k]);
        break;
    case ORC_PROP_DIAG:
        for (int k = 0; k < d; ++k) y[k] = mu ? x[k] + FMA(p->vec[k], z[k], mu[k]) : FMA(p->vec[k], z[k], x[k]);
        break;
    default: {
        size_t off = 0;
        for (int i = 0; i < d; ++i) {
            real w = R(0.0);
            for (int j = 0; j <= i; ++j) w = FMA(p->vec[off + j], z[j], w);
            y[i] = mu ? x[i] + (mu[i] + w) : x[i] + w;
            off += (size_t)i + 1;
        }
    }
    }
}

/* q(x) = -1/2 |L^-1 (x - mu)|^2: logpdf of the proposal at x up to its constant (src/proposal.jl:31-35);
 * forward substitution, every sum in ascending order */
static real static_logq(const orc_proposal *p, int d, const real *x, real *t)
{
    const real *mu = p->mean;
    size_t off = 0;
    real q = R(0.0);
    for (int i = 0; i < d; ++i) {
        const real r = mu ? x[i] - mu[i] : x[i];
        if (p->kind == ORC_PROP_ISO) t[i] = r / p->scale;
        else if (p->kind == ORC_PROP_DIAG) t[i] = r / p->vec[i];
        else {
            real acc = R(0.0);
            for (int j = 0; j < i; ++j) acc = FMA(p->vec[off + j], t[j], acc);
            t[i] = (r - acc) / p->vec[off + i];
            off += (size_t)i + 1;
        }
        q = FMA(t[i], t[i], q);
    }
    return -R(0.5) * q;
}

/* sum_k (z_k + add_k)^2 (add == NULL: sum z_k^2) with the reduction shape of the cooperative kernel: lane l accumulates the
 * blocks b = l, l+L, ... of 4 dimensions in order, the L partial sums meet in the butterfly; L <= 1: plain ascending order */
static real lanes_sumsq(const real *z, const real *add, int d, int L)
{
    if (L <= 1) {
        real q = R(0.0);
        for (int k = 0; k < d; ++k) { const real tk = add ? z[k] + add[k] : z[k]; q = FMA(tk, tk, q); }
        return q;
    }
    const int nblk = (d + 3) / 4;
    real p[64];
    for (int l = 0; l < L; ++l) {
        real q = R(0.0);
        for (int b = l; b < nblk; b += L)
            for (int j = 0; j < 4 && 4 * b + j < d; ++j) {
                const int k = 4 * b + j;
                const real tk = add ? z[k] + add[k] : z[k];
                q = FMA(tk, tk, q);
            }
        p[l] = q;
    }
    return butterfly(p, L);
}

/* twice the whitened mean 2 L^-1 mu (host arithmetic in double, rounded once): with it the Hastings ratio of a
 * drifting random walk is  logq(x|y) - logq(y|x) = 1/2 |z|^2 - 1/2 |z + 2 L^-1 mu|^2   (src/proposal.jl:58-64,190-192) */
static void whitened_mean2(const orc_proposal *p, int d, real *tm)
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
        tm[i] = (real)(2.0 * m[i]);
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

static void record(real *samples, uint8_t *accepted, int64_t slot, int d, int C, int c,
                   const real *x, real lp, int acc)
{
    if (samples) {
        real *row = samples + (size_t)slot * (size_t)(d + 1) * (size_t)C;
        for (int k = 0; k < d; ++k) row[(size_t)k * C + c] = x[k];
        row[(size_t)d * C + c] = lp;
    }
    if (accepted) accepted[(size_t)slot * C + c] = (uint8_t)acc;
}

/* Trace sink (tests only): per saved slot and chain [n_samples][C]
 *   margin   -- the smallest |logu - logalpha| over the transitions since the previous saved slot: how far the closest
 *               accept decision was from flipping (tests/test_julia_reference_traces.py compares decisions only where
 *               the margin exceeds the rounding differences between this spec's fused sums and Julia's unfused ones);
 *   logalpha, eta -- RAM: state.logalpha and state.eta after the saved transition (...RAM.jl:99-114).
 * Thread-local: a sink set by one host thread is seen by the calls of that thread only. */
static _Thread_local real *g_margin, *g_logalpha, *g_eta;
void orc_set_trace(real *margin, real *logalpha, real *eta) { g_margin = margin; g_logalpha = logalpha; g_eta = eta; }

static inline void margin_note(real *cur, real logu, real loga)
{
    real m = FABS(logu - loga);
    if (m != m) m = (real)INFINITY;                    /* NaN ratio: rejected on both sides whatever the rounding */
    if (m < *cur) *cur = m;
}
static inline void margin_flush(real *cur, int64_t slot, int C, int c)
{
    if (g_margin && slot >= 0) { g_margin[(size_t)slot * C + c] = *cur; *cur = (real)INFINITY; }
}

/* ------------------------------------------------------------------------------------------ */
/* RWMH: src/mh-core.jl:76-86 (initial step) and :92-117 (step)                               */
int orc_rwmh(const orc_target *t, const orc_proposal *p, const orc_schedule *s,
             uint64_t seed, uint64_t first_chain, int nchains,
             const real *init, real *samples, uint8_t *accepted,
             real *final_x, real *final_lp, uint32_t *accept_counts)
{
    const int d = t->dim, C = nchains;
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    real *x = malloc(sizeof(real) * (size_t)d * 5);
    real *y = x + d, *z = y + d, *tm = z + d, *zero = tm + d;
    if (p->mean && !p->is_static) whitened_mean2(p, d, tm);
    for (int k = 0; k < d; ++k) zero[k] = R(0.0);
    for (int c = 0; c < C; ++c) {
        const uint64_t id = first_chain + (uint64_t)c;
        /* mh-core.jl:83  params = initial_params === nothing ? propose(rng, sampler, model) : initial_params
         * proposal.jl:41-47: the initial propose is a bare draw from the proposal (x = 0 + xi). */
        if (init) {
            /* a caller-supplied -0.0 enters the chain as +0.0 (x + 0.0; every other value unchanged; equal under ==): the
             * engine does the same in mhx_run_init / mhx_run_set_state, see rwmh_canonical_zero in csrc/mhx_api.hip */
            for (int k = 0; k < d; ++k) x[k] = init[(size_t)k * C + c] + R(0.0);
        } else {
            normals_gen(p->normal_gen, seed, id, 0, ORC_STREAM_INIT, d, z);
            for (int k = 0; k < d; ++k) y[k] = R(0.0);
            propose_from(p, d, z, y, x);
        }
        real lp = orc_target_eval(t, x);               /* mh-core.jl:84 transition(..., false) */
        real qx = p->is_static ? static_logq(p, d, x, y) : R(0.0);
        uint32_t nacc = 0;
        real mg = (real)INFINITY;
        int64_t slot = save_slot(s, 0);
        if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, 0);
        margin_flush(&mg, slot, C, c);
        for (int64_t tau = 1; tau <= nT; ++tau) {
            const uint32_t step = (uint32_t)tau;
            normals_gen(p->normal_gen, seed, id, step, ORC_STREAM_PROPOSAL, d, z);
            propose_from(p, d, z, p->is_static ? zero : x, y);   /* mh-core.jl:100; static: proposal.jl:66-72 */
            real lpy = orc_target_eval(t, y);          /* :103 */
            real loga = lpy - lp;                      /* :104-105, Hastings ratio of a zero-mean RW == 0 */
            real qy = R(0.0);
            /* the sums of the ratio take the target's reduction shape when the cooperative kernel runs the walk */
            const int Lw = (p->kind != ORC_PROP_DENSE) ? t->reduce_lanes : 1;
            if (p->is_static) {                         /* proposal.jl:74-83: q = logpdf(proposal, t) */
                const real fwd = lanes_sumsq(z, NULL, d, Lw);
                qy = -R(0.5) * fwd;
                loga = (lpy - lp) + (qx - qy);
            } else if (p->mean) {                       /* :105,119-123 -> proposal.jl:190-192 */
                const real fwd = lanes_sumsq(z, NULL, d, Lw), bwd = lanes_sumsq(z, tm, d, Lw);
                loga = (lpy - lp) + R(0.5) * (fwd - bwd);
            }
            real logu = orc_accept_logu(seed, id, step);
            int acc = logu < loga;                      /* :108  -randexp(rng) < loga (strict; NaN -> reject) */
            margin_note(&mg, logu, loga);
            if (acc) { memcpy(x, y, sizeof(real) * (size_t)d); lp = lpy; qx = qy; ++nacc; }
            slot = save_slot(s, tau);
            if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, acc);
            margin_flush(&mg, slot, C, c);
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
static int stretch_move(const orc_target *t, real a, uint64_t seed, uint64_t ens, int i,
                        uint32_t sweep, int other_start, int other_size, int wrap_W,
                        const real *cur, const real *oth_new, const real *oth_old, int use_seq,
                        int W, real *xi, real *lpi, real *y, real *xj, real *mg)
{
    const int d = t->dim;
    uint32_t w0;
    real u, logu;
    emcee_draws(seed, ens, i, sweep, &w0, &u, &logu);
    int j;
    if (use_seq) {
        /* emcee.jl:48,52  idx = mod1(i + rand(1:W-1), W)  (never i) */
        uint32_t r = 1u + (uint32_t)(((uint64_t)w0 * (uint64_t)(W - 1)) >> 32);
        j = (int)(((uint64_t)i + r) % (uint64_t)wrap_W);
        /* emcee.jl:53  other = idx < i ? new_walkers[idx] : walkers[idx] */
        const real *src = (j < i) ? oth_new : oth_old;
        for (int k = 0; k < d; ++k) xj[k] = src[(size_t)k * W + j];
    } else {
        j = other_start + (int)(((uint64_t)w0 * (uint64_t)other_size) >> 32);
        for (int k = 0; k < d; ++k) xj[k] = cur[(size_t)k * W + j];
    }
    /* emcee.jl:81  z = ((a - 1) * rand(rng) + 1)^2 / a */
    real tt = FMA(a - R(1.0), u, R(1.0));
    real z = (tt * tt) / a;
    real alphamult = (real)(d - 1) * orc_log(z);     /* :82 */
    for (int k = 0; k < d; ++k) y[k] = FMA(z, xi[k] - xj[k], xj[k]);   /* :85 */
    real lpy = orc_target_eval(t, y);                  /* :88 */
    real alpha = (alphamult + lpy) - *lpi;             /* :91 */
    int acc = logu <= alpha;                            /* :93  -randexp <= alpha (non-strict) */
    margin_note(mg, logu, alpha);
    if (acc) { memcpy(xi, y, sizeof(real) * (size_t)d); *lpi = lpy; }
    return acc;
}

int orc_emcee(const orc_target *t, real a, int mode, const orc_schedule *s,
              uint64_t seed, uint64_t ensemble_id, int nwalkers,
              const real *init, const orc_proposal *prior, real *samples, uint8_t *accepted,
              real *final_x, real *final_lp, uint32_t *accept_counts)
{
    const int d = t->dim, W = nwalkers;
    if ((!init && !prior) || W < 2) return -1;
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    real *cur = malloc(sizeof(real) * (size_t)d * W);
    real *nxt = malloc(sizeof(real) * (size_t)d * W);
    real *lp = malloc(sizeof(real) * (size_t)W);
    real *lpn = malloc(sizeof(real) * (size_t)W);
    uint8_t *acc = malloc((size_t)W);
    real *mgw = malloc(sizeof(real) * (size_t)W);
    for (int i = 0; i < W; ++i) mgw[i] = (real)INFINITY;
    real *tmp = malloc(sizeof(real) * (size_t)d * 3);
    real *xi = tmp, *y = tmp + d, *xj = tmp + 2 * d;
    if (init) memcpy(cur, init, sizeof(real) * (size_t)d * W);
    else {
        /* emcee.jl:29-34: the initial walkers are W draws from the wrapped prior -- here a (Mv)Normal mu + L z
         * [upstream Distributions rand(MvNormal), restated], z from stream INIT of (ensemble, walker) */
        const real *mu = prior->mean;
        for (int i = 0; i < W; ++i) {
            orc_normals(seed, (ensemble_id << 32) | (uint32_t)i, 0, ORC_STREAM_INIT, d, y);
            size_t off = 0;
            for (int k = 0; k < d; ++k) {
                if (prior->kind == ORC_PROP_DENSE) {
                    real w = R(0.0);
                    for (int j = 0; j <= k; ++j) w = FMA(prior->vec[off + j], y[j], w);
                    off += (size_t)k + 1;
                    xi[k] = mu ? mu[k] + w : w;
                } else {
                    const real sc = prior->kind == ORC_PROP_ISO ? prior->scale : prior->vec[k];
                    xi[k] = mu ? FMA(sc, y[k], mu[k]) : FMA(sc, y[k], R(0.0));
                }
            }
            for (int k = 0; k < d; ++k) cur[(size_t)k * W + i] = xi[k];
        }
    }
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
            margin_flush(&mgw[i], slot, W, i);
        }
    const int half = W / 2;
    for (int64_t tau = 1; tau <= nT; ++tau) {
        const uint32_t sweep = (uint32_t)tau;
        if (mode == 0) {
            /* reference-faithful Gauss-Seidel sweep, emcee.jl:50-55 */
            for (int i = 0; i < W; ++i) {
                for (int k = 0; k < d; ++k) xi[k] = cur[(size_t)k * W + i];
                real l = lp[i];
                acc[i] = (uint8_t)stretch_move(t, a, seed, ensemble_id, i, sweep, 0, 0, W, cur, nxt, cur,
                                               1, W, xi, &l, y, xj, &mgw[i]);
                for (int k = 0; k < d; ++k) nxt[(size_t)k * W + i] = xi[k];
                lpn[i] = l;
            }
            real *sw = cur; cur = nxt; nxt = sw;
            sw = lp; lp = lpn; lpn = sw;
        } else {
            /* parallel split: half 0 = [0, W/2) moves against half 1, then half 1 against updated half 0 */
            for (int h = 0; h < 2; ++h) {
                const int lo = h ? half : 0, hi = h ? W : half;
                const int ostart = h ? 0 : half, osize = h ? half : W - half;
                for (int i = lo; i < hi; ++i) {
                    for (int k = 0; k < d; ++k) xi[k] = cur[(size_t)k * W + i];
                    real l = lp[i];
                    acc[i] = (uint8_t)stretch_move(t, a, seed, ensemble_id, i, sweep, ostart, osize, W, cur,
                                                   NULL, NULL, 0, W, xi, &l, y, xj, &mgw[i]);
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
                margin_flush(&mgw[i], slot, W, i);
            }
    }
    if (final_x) memcpy(final_x, cur, sizeof(real) * (size_t)d * W);
    if (final_lp) memcpy(final_lp, lp, sizeof(real) * (size_t)W);
    free(cur); free(nxt); free(lp); free(lpn); free(acc); free(tmp); free(mgw);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* rank-1 Cholesky update / downdate of a packed lower factor
 * [upstream LinearAlgebra.lowrankupdate!/lowrankdowndate!, restated]; call site
 * src/RobustAdaptiveMetropolis.jl:165-171.  Column sweep: for each column i a Givens-type
 * rotation (c, s) from (S_ii, w_i) is applied to the sub-diagonal column and to w[i+1:].     */
#define SIDX(i, j) ((size_t)(i) * ((size_t)(i) + 1) / 2 + (size_t)(j))
int orc_chol_rank1(real *S, real *w, int d, int sign)
{
    /* One sweep for both signs (the textbook rank-one modification; for sigma = -1 it is the upstream
     * lowrankdowndate! loop, for sigma = +1 it equals the upstream Givens form algebraically):
     *   s = w_i / S_ii,  c = sqrt(1 + sigma s^2),  S_ii <- c S_ii,
     *   S_ji <- (S_ji + sigma s w_j) / c,  w_j <- c w_j - s S_ji(new)            (spec 3.9) */
    const real sg = sign > 0 ? R(1.0) : -R(1.0);
    for (int i = 0; i < d; ++i) {
        const real a = S[SIDX(i, i)], b = w[i];
        const real sn = b / a;
        if (sign < 0 && sn * sn > R(1.0)) return i + 1;   /* PosDefException(i) upstream */
        const real ss = sg * sn;
        const real c = SQRT(FMA(ss, sn, R(1.0)));
        const real rc = R(1.0) / c;                        /* one reciprocal per column */
        S[SIDX(i, i)] = c * a;
        for (int j = i + 1; j < d; ++j) {
            const real vj = w[j];
            const real Aji = FMA(ss, vj, S[SIDX(j, i)]) * rc;
            S[SIDX(j, i)] = Aji;
            w[j] = FMA(c, vj, -(sn * Aji));
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* RAM: src/RobustAdaptiveMetropolis.jl:175-214 (initial step), :123-151 (ram_step_inner),
 * :153-173 (ram_adapt), :216-237 (step), :239-245 (valid_eigenvalues), :247-278 (step_warmup) */
int orc_ram(const orc_target *t, const orc_ram_cfg *cfg, const orc_schedule *s,
            uint64_t seed, uint64_t first_chain, int nchains,
            const real *init, const real *S_in, real *S_out,
            real *samples, uint8_t *accepted, real *final_x, real *final_lp,
            uint32_t *accept_counts, uint8_t *status, real *diag_min, real *diag_max)
{
    const int d = t->dim, C = nchains;
    const size_t nS = (size_t)d * ((size_t)d + 1) / 2;
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    real *x = malloc(sizeof(real) * (size_t)d * 5);
    real *y = x + d, *U = y + d, *v = U + d, *w = v + d;
    real *S = malloc(sizeof(real) * nS), *Sn = malloc(sizeof(real) * nS);
    const int default_bounds = (cfg->eig_lo == R(0.0) && isinf(cfg->eig_hi) && cfg->eig_hi > 0);
    for (int c = 0; c < C; ++c) {
        const uint64_t id = first_chain + (uint64_t)c;
        uint8_t st = 0;
        if (init) for (int k = 0; k < d; ++k) x[k] = init[(size_t)k * C + c];
        else orc_normals(seed, id, 0, ORC_STREAM_INIT, d, x);      /* :193 randn(rng, T, d) */
        if (S_in) memcpy(S, S_in + (size_t)c * nS, sizeof(real) * nS);
        else { memset(S, 0, sizeof(real) * nS); for (int i = 0; i < d; ++i) S[SIDX(i, i)] = R(1.0); }
        real lp = orc_target_eval(t, x);                          /* :210 */
        uint32_t nacc = 0;
        if (diag_min) for (int k = 0; k < d; ++k) {
            diag_min[(size_t)k * C + c] = S[SIDX(k, k)];
            diag_max[(size_t)k * C + c] = S[SIDX(k, k)];
        }
        real mg = (real)INFINITY;
        real st_loga = R(0.0), st_eta = R(0.0);                   /* :211 State(x, lp, S, zero(T), 0, 1, true) */
        int64_t slot = save_slot(s, 0);
        if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, 1);   /* :213 Transition(x, lp, true) */
        margin_flush(&mg, slot, C, c);
        if (slot >= 0 && g_logalpha) g_logalpha[(size_t)slot * C + c] = st_loga;
        if (slot >= 0 && g_eta) g_eta[(size_t)slot * C + c] = st_eta;
        for (int64_t tau = 1; tau <= nT; ++tau) {
            const uint32_t step = (uint32_t)tau;                   /* == state.iteration */
            orc_normals(seed, id, step, ORC_STREAM_PROPOSAL, d, U);            /* :135 */
            for (int i = 0; i < d; ++i) {                                       /* :136 muladd(S, U, x) */
                real acc = R(0.0);
                for (int j = 0; j <= i; ++j) acc = FMA(S[SIDX(i, j)], U[j], acc);
                v[i] = acc;
                y[i] = acc + x[i];
            }
            real lpy = orc_target_eval(t, y);                     /* :140 */
            real diff = lpy - lp;
            real loga = (diff != diff) ? diff : (diff < R(0.0) ? diff : R(0.0));   /* :147 min(lp_new - lp, 0) */
            real logu = orc_accept_logu(seed, id, step);
            int acc = logu < loga;                                  /* :148 randexp(rng) > -loga */
            margin_note(&mg, logu, loga);
            st_loga = loga;                                         /* :231,:272 the new state's logalpha */
            if (tau <= nA) {                                        /* step_warmup: adapt, :153-173 */
                real da = orc_exp(loga) - cfg->alpha;             /* :159 */
                const real eta = (real)pow((double)step, -(double)cfg->gamma);   /* :162 */
                st_eta = eta;                                       /* :273 (a fixed-S step keeps state.eta, :232) */
                if (da == da) {
                    real nn = R(0.0);
                    for (int j = 0; j < d; ++j) nn = FMA(U[j], U[j], nn);
                    real coef = SQRT(eta * FABS(da)) / SQRT(nn);             /* :163 */
                    for (int j = 0; j < d; ++j) w[j] = v[j] * coef;
                    memcpy(Sn, S, sizeof(real) * nS);
                    int fail = orc_chol_rank1(Sn, w, d, da > R(0.0) ? +1 : -1);   /* :165-171 */
                    int ok = !fail;
                    if (fail) st |= 1;
                    if (ok && !default_bounds)                                  /* :239-245, :259-264 */
                        for (int k = 0; k < d; ++k) {
                            real e = Sn[SIDX(k, k)];
                            if (!(cfg->eig_lo <= e && e <= cfg->eig_hi)) { ok = 0; break; }
                        }
                    if (ok) { real *sw = S; S = Sn; Sn = sw; }
                } else {
                    st |= 2;                                        /* NaN log-ratio: adaptation skipped */
                }
                if (diag_min) for (int k = 0; k < d; ++k) {
                    real e = S[SIDX(k, k)];
                    if (e < diag_min[(size_t)k * C + c]) diag_min[(size_t)k * C + c] = e;
                    if (e > diag_max[(size_t)k * C + c]) diag_max[(size_t)k * C + c] = e;
                }
            }
            if (acc) { memcpy(x, y, sizeof(real) * (size_t)d); lp = lpy; ++nacc; }   /* :267-277 */
            slot = save_slot(s, tau);
            if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, acc);
            margin_flush(&mg, slot, C, c);
            if (slot >= 0 && g_logalpha) g_logalpha[(size_t)slot * C + c] = st_loga;
            if (slot >= 0 && g_eta) g_eta[(size_t)slot * C + c] = st_eta;
        }
        if (final_x) for (int k = 0; k < d; ++k) final_x[(size_t)k * C + c] = x[k];
        if (final_lp) final_lp[c] = lp;
        if (accept_counts) accept_counts[c] = nacc;
        if (status) status[c] = st;
        if (S_out) memcpy(S_out + (size_t)c * nS, S, sizeof(real) * nS);
    }
    free(x); free(S); free(Sn);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* RAM with a DEFERRED factor: the same chain as orc_ram in exact arithmetic, another rounding.
 * ram_adapt (src/RobustAdaptiveMetropolis.jl:153-173) updates  S S' <- S S' + sigma w w'  with  w = c S U,
 * c^2 = eta |dalpha| / |U|^2, so  S_new = S M  with  M = chol(I + sigma c^2 U U')  -- lower triangular,
 *     M_jj = a_j = sqrt(T_{j+1} / T_j),   M_ij = U_i g_j  (i > j),   g_j = sigma c^2 U_j / (T_j a_j),
 *     T_j = 1 + sigma c^2 sum_{m<j} U_m^2
 * and known from U and two scalars alone.  Up to K accepted updates stay PENDING as (a, g, u) triples:
 *     proposal   x' = x + S_0 (M_1 (M_2 ... (M_m U)))          one read of the stored factor S_0, m prefix scans
 *     flush      S_0 <- S_0 M_1 ... M_m                        one read + one write per K steps, column by column:
 *                (S M)_rj = a_j S_rj + g_j sum_{i>j} S_ri u_i,  the suffix sum carried as  t_r - sum_{i<=j} S_ri u_i
 *                with t = S u = the v of that step's proposal.
 * Prefix sums over the <= 256 elements have the shape of the kernel's scan: 64 blocks of 4 consecutive elements
 * (running fma inside a block), a Kogge-Stone scan over the 64 block totals.
 * A flush happens when K updates are pending and after every transition listed in flush_at (ascending; the launch ends
 * of the engine: the factor is whole in memory between launches).                                                      */
#define DEFER_NB 64
static void defer_scan_dot(const real *p, const real *q, int d, real *excl /* [257] */)
{
    real run[DEFER_NB][4], tot[DEFER_NB], nxt[DEFER_NB];
    for (int b = 0; b < DEFER_NB; ++b) {
        real r = R(0.0);
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * b + e;
            const real pj = j < d ? p[j] : R(0.0), qj = j < d ? q[j] : R(0.0);
            r = e == 0 ? pj * qj : FMA(pj, qj, r);
            run[b][e] = r;
        }
        tot[b] = r;
    }
    for (int s = 1; s < DEFER_NB; s <<= 1) {                 /* inclusive Kogge-Stone */
        for (int b = 0; b < DEFER_NB; ++b) nxt[b] = b >= s ? tot[b - s] + tot[b] : tot[b];
        memcpy(tot, nxt, sizeof tot);
    }
    for (int b = 0; b < DEFER_NB; ++b) {
        const real E = b ? tot[b - 1] : R(0.0);
        excl[4 * b] = E;
        for (int e = 1; e < 4; ++e) excl[4 * b + e] = E + run[b][e - 1];
    }
    excl[4 * DEFER_NB] = tot[DEFER_NB - 1];
}

int orc_ram_deferred(const orc_target *t, const orc_ram_cfg *cfg, const orc_schedule *s,
                     uint64_t seed, uint64_t first_chain, int nchains,
                     const real *init, const real *S_in, real *S_out,
                     real *samples, uint8_t *accepted, real *final_x, real *final_lp,
                     uint32_t *accept_counts, uint8_t *status, real *diag_min, real *diag_max,
                     int K, const int64_t *flush_at, int nflush)
{
    const int d = t->dim, C = nchains;
    if (d > 4 * DEFER_NB || K < 1) return -1;
    const size_t nS = (size_t)d * ((size_t)d + 1) / 2;
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    real *x = malloc(sizeof(real) * (size_t)d * 6);
    real *y = x + d, *U = y + d, *v = U + d, *z = v + d, *dg = z + d;
    real *S = malloc(sizeof(real) * nS);
    real *pa = malloc(sizeof(real) * (size_t)K * d * 4);      /* pending a, g, u, t */
    real *pg = pa + (size_t)K * d, *pu = pg + (size_t)K * d, *pt = pu + (size_t)K * d;
    real PU[4 * DEFER_NB + 1], P[4 * DEFER_NB + 1], T[4 * DEFER_NB + 1];
    const int default_bounds = (cfg->eig_lo == R(0.0) && isinf(cfg->eig_hi) && cfg->eig_hi > 0);
    for (int c = 0; c < C; ++c) {
        const uint64_t id = first_chain + (uint64_t)c;
        uint8_t st = 0;
        if (init) for (int k = 0; k < d; ++k) x[k] = init[(size_t)k * C + c];
        else orc_normals(seed, id, 0, ORC_STREAM_INIT, d, x);
        if (S_in) memcpy(S, S_in + (size_t)c * nS, sizeof(real) * nS);
        else { memset(S, 0, sizeof(real) * nS); for (int i = 0; i < d; ++i) S[SIDX(i, i)] = R(1.0); }
        for (int k = 0; k < d; ++k) dg[k] = S[SIDX(k, k)];        /* diag(S_0 M_1 ... M_m): the product of the diagonals */
        real lp = orc_target_eval(t, x);
        uint32_t nacc = 0;
        if (diag_min) for (int k = 0; k < d; ++k) { diag_min[(size_t)k * C + c] = dg[k]; diag_max[(size_t)k * C + c] = dg[k]; }
        real mg = (real)INFINITY;
        real st_loga = R(0.0), st_eta = R(0.0);
        int64_t slot = save_slot(s, 0);
        if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, 1);
        margin_flush(&mg, slot, C, c);
        if (slot >= 0 && g_logalpha) g_logalpha[(size_t)slot * C + c] = st_loga;
        if (slot >= 0 && g_eta) g_eta[(size_t)slot * C + c] = st_eta;
        int m = 0, fi = 0;
        for (int64_t tau = 1; tau <= nT; ++tau) {
            const uint32_t step = (uint32_t)tau;
            orc_normals(seed, id, step, ORC_STREAM_PROPOSAL, d, U);
            defer_scan_dot(U, U, d, PU);
            const real nn = PU[4 * DEFER_NB];
            memcpy(z, U, sizeof(real) * (size_t)d);
            for (int i = m - 1; i >= 0; --i) {                      /* z = M_1 (... (M_m U)) */
                const real *a = pa + (size_t)i * d, *g = pg + (size_t)i * d, *u = pu + (size_t)i * d;
                defer_scan_dot(g, z, d, P);
                for (int j = 0; j < d; ++j) z[j] = FMA(u[j], P[j], a[j] * z[j]);
            }
            for (int i = 0; i < d; ++i) {
                real acc = R(0.0);
                for (int j = 0; j <= i; ++j) acc = FMA(S[SIDX(i, j)], z[j], acc);
                v[i] = acc;
                y[i] = acc + x[i];
            }
            real lpy = orc_target_eval(t, y);
            real diff = lpy - lp;
            real loga = (diff != diff) ? diff : (diff < R(0.0) ? diff : R(0.0));
            real logu = orc_accept_logu(seed, id, step);
            int acc = logu < loga;
            margin_note(&mg, logu, loga);
            st_loga = loga;
            if (tau <= nA) {
                real da = orc_exp(loga) - cfg->alpha;
                const real eta = (real)pow((double)step, -(double)cfg->gamma);
                st_eta = eta;
                if (da == da) {
                    const real c2 = (eta * FABS(da)) / nn;
                    const real sc = da > R(0.0) ? c2 : -c2;
                    real *a = pa + (size_t)m * d, *g = pg + (size_t)m * d, *u = pu + (size_t)m * d, *tt = pt + (size_t)m * d;
                    int ok = 1;
                    for (int j = 0; j <= d; ++j) T[j] = FMA(sc, PU[j], R(1.0));
                    for (int j = 0; j < d; ++j) if (!(T[j + 1] > R(0.0))) ok = 0;      /* the downdate left the PD cone */
                    if (!ok) st |= 1;
                    if (ok) {
                        for (int j = 0; j < d; ++j) {
                            a[j] = SQRT(T[j + 1] / T[j]);
                            g[j] = (sc * U[j]) / (T[j] * a[j]);
                            u[j] = U[j];
                            tt[j] = v[j];
                        }
                        if (!default_bounds)
                            for (int k = 0; k < d; ++k) {
                                real e = dg[k] * a[k];
                                if (!(cfg->eig_lo <= e && e <= cfg->eig_hi)) { ok = 0; break; }
                            }
                    }
                    if (ok) { for (int k = 0; k < d; ++k) dg[k] = dg[k] * a[k]; ++m; }
                } else {
                    st |= 2;
                }
                if (diag_min) for (int k = 0; k < d; ++k) {
                    real e = dg[k];
                    if (e < diag_min[(size_t)k * C + c]) diag_min[(size_t)k * C + c] = e;
                    if (e > diag_max[(size_t)k * C + c]) diag_max[(size_t)k * C + c] = e;
                }
            }
            int forced = 0;
            while (fi < nflush && flush_at[fi] < tau) ++fi;
            if (fi < nflush && flush_at[fi] == tau) forced = 1;
            if (tau == nT || tau == nA) forced = 1;
            if (m == K || (forced && m > 0)) {                      /* S_0 <- S_0 M_1 ... M_m */
                for (int r = 0; r < d; ++r)
                    for (int j = 0; j <= r; ++j) {
                        real cc = S[SIDX(r, j)];
                        for (int i = 0; i < m; ++i) {
                            const real a = pa[(size_t)i * d + j], g = pg[(size_t)i * d + j], u = pu[(size_t)i * d + j];
                            real *tt = pt + (size_t)i * d;
                            if (r == j) cc = a * cc;
                            else { tt[r] = FMA(-u, cc, tt[r]); cc = FMA(g, tt[r], a * cc); }
                        }
                        S[SIDX(r, j)] = cc;
                    }
                m = 0;
            }
            if (acc) { memcpy(x, y, sizeof(real) * (size_t)d); lp = lpy; ++nacc; }
            slot = save_slot(s, tau);
            if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, acc);
            margin_flush(&mg, slot, C, c);
            if (slot >= 0 && g_logalpha) g_logalpha[(size_t)slot * C + c] = st_loga;
            if (slot >= 0 && g_eta) g_eta[(size_t)slot * C + c] = st_eta;
        }
        if (final_x) for (int k = 0; k < d; ++k) final_x[(size_t)k * C + c] = x[k];
        if (final_lp) final_lp[c] = lp;
        if (accept_counts) accept_counts[c] = nacc;
        if (status) status[c] = st;
        if (S_out) memcpy(S_out + (size_t)c * nS, S, sizeof(real) * nS);
    }
    free(x); free(S); free(pa);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* gradients of the catalogue targets (what ForwardDiff / LogDensityProblems.logdensity_and_gradient
 * supply to the reference's MALA, src/MALA.jl:73-75, ext/AdvancedMHForwardDiffExt.jl:13-17)     */
real orc_target_grad(const orc_target *t, const real *x, real *g, orc_logdensity_grad_fn user)
{
    const int d = t->dim;
    if (t->reduce_lanes > 1 && (t->kind == ORC_TARGET_ISO_GAUSS || t->kind == ORC_TARGET_BANANA || t->kind == ORC_TARGET_FUNNEL)) {
        /* the cooperative kernel's reduction shape: the sum of squares comes from the lane partial sums + butterfly, the
         * gradient is element-wise from it */
        const real q = split_sum_squares(t, x);
        if (t->kind == ORC_TARGET_ISO_GAUSS) {
            for (int k = 0; k < d; ++k) g[k] = -x[k];
            return FMA(-R(0.5), q, target_const(t));
        }
        if (t->kind == ORC_TARGET_BANANA) {
            const real b = t->params[0], x0 = x[0];
            const real u = FMA(b, FMA(x0, x0, -R(100.0)), x[1]);
            g[0] = -(FMA(x0, R(0.01), (R(2.0) * b) * (u * x0)));
            g[1] = -u;
            for (int k = 2; k < d; ++k) g[k] = -x[k];
            return FMA(-R(0.5), q, target_const(t));
        }
        const real v = x[0];
        const real ev = orc_exp(-v);
        real r = (v * v) * ONE_18;
        r = FMA(R(0.5) * (real)(d - 1), v, r);
        r = FMA(R(0.5) * ev, q, r);
        g[0] = FMA(R(0.5) * ev, q, -(FMA(v, ONE_9, R(0.5) * (real)(d - 1))));
        for (int k = 1; k < d; ++k) g[k] = -(ev * x[k]);
        return target_const(t) - r;
    }
    switch (t->kind) {
    case ORC_TARGET_ISO_GAUSS:
        for (int k = 0; k < d; ++k) g[k] = -x[k];
        return orc_target_eval(t, x);
    case ORC_TARGET_CORR_GAUSS: {                      /* grad = -A^T (A x) */
        const real *A = t->params;
        real q = R(0.0);
        size_t off = 0;
        for (int i = 0; i < d; ++i) {                  /* w = A x, kept in g */
            real w = R(0.0);
            for (int j = 0; j <= i; ++j) w = FMA(A[off + j], x[j], w);
            g[i] = w;
            q = FMA(w, w, q);
            off += (size_t)i + 1;
        }
        if (t->reduce_lanes > 1) {                     /* the matrix-core kernel's shape: the squares of rows l, l+L, ... on lane l */
            real p[64];
            for (int l = 0; l < t->reduce_lanes; ++l) {
                real ql = R(0.0);
                for (int i = l; i < d; i += t->reduce_lanes) ql = FMA(g[i], g[i], ql);
                p[l] = ql;
            }
            q = butterfly(p, t->reduce_lanes);
        }
        for (int j = 0; j < d; ++j) {                  /* g_j = -sum_{i>=j} A_ij w_i, ascending i, in place */
            real acc = R(0.0);
            for (int i = j; i < d; ++i) acc = FMA(A[SIDX(i, j)], g[i], acc);
            g[j] = -acc;
        }
        return FMA(-R(0.5), q, target_const(t));
    }
    case ORC_TARGET_IID_NORMAL: {
        const real mu = x[0], sigma = x[1];
        if (!(sigma > R(0.0))) { g[0] = R(0.0); g[1] = R(0.0); return -INFINITY; }
        const real inv = R(1.0) / sigma;
        real acc = R(0.0), s1 = R(0.0);
        for (int i = 0; i < t->nparams; ++i) {
            const real z = (t->params[i] - mu) / sigma;
            acc = FMA(z, z, acc);
            s1 = s1 + z;
        }
        const real nf = (real)t->nparams;
        g[0] = s1 * inv;                               /* sum (y-mu)/sigma^2 */
        g[1] = (acc - nf) * inv;                       /* -n/sigma + sum (y-mu)^2/sigma^3 */
        const real tt = orc_log(sigma) + HALF_LOG_2PI;
        return FMA(-R(0.5), acc, -(nf * tt));
    }
    case ORC_TARGET_BANANA: {
        const real b = t->params[0];
        const real x0 = x[0];
        real q = (x0 * x0) * R(0.01);
        const real u = FMA(b, FMA(x0, x0, -R(100.0)), x[1]);
        q = FMA(u, u, q);
        g[0] = -(FMA(x0, R(0.01), (R(2.0) * b) * (u * x0)));
        g[1] = -u;
        for (int k = 2; k < d; ++k) { q = FMA(x[k], x[k], q); g[k] = -x[k]; }
        return FMA(-R(0.5), q, target_const(t));
    }
    case ORC_TARGET_FUNNEL: {
        const real v = x[0];
        real q = R(0.0);
        for (int k = 1; k < d; ++k) q = FMA(x[k], x[k], q);
        const real ev = orc_exp(-v);
        real r = (v * v) * ONE_18;
        r = FMA(R(0.5) * (real)(d - 1), v, r);
        r = FMA(R(0.5) * ev, q, r);
        g[0] = FMA(R(0.5) * ev, q, -(FMA(v, ONE_9, R(0.5) * (real)(d - 1))));   /* -v/9 - (d-1)/2 + e^-v q/2 */
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
int orc_mala(const orc_target *t, orc_logdensity_grad_fn user, real sigma2, const orc_schedule *s,
             uint64_t seed, uint64_t first_chain, int nchains, const real *init,
             real *samples, uint8_t *accepted, real *final_x, real *final_lp, uint32_t *accept_counts, int normal_gen)
{
    const int d = t->dim, C = nchains;
    if (!init) return -1;                                /* :37 "please specify initial parameters" */
    /* normal_gen: 0 Box-Muller, 1 the table ziggurat (fp64) -- which standard normals the noise z is */
    int64_t nT, nA;
    orc_schedule_counts(s, &nT, &nA);
    const real sigma = SQRT(sigma2);
    const real h = (sigma * sigma) * R(0.5);              /* drift step sigma2/2 */
    const real hs = R(0.5) * sigma;
    real *x = malloc(sizeof(real) * (size_t)d * 6);
    real *gx = x + d, *y = gx + d, *gy = y + d, *z = gy + d, *tk = z + d;
    const int Lw = t->reduce_lanes;                      /* reduction shape of the three sums of a step */
    for (int c = 0; c < C; ++c) {
        const uint64_t id = first_chain + (uint64_t)c;
        for (int k = 0; k < d; ++k) x[k] = init[(size_t)k * C + c];
        real lp = orc_target_grad(t, x, gx, user);      /* :38-40 GradientTransition(params, lp, grad, false) */
        uint32_t nacc = 0;
        real mg = (real)INFINITY;
        int64_t slot = save_slot(s, 0);
        if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, 0);
        margin_flush(&mg, slot, C, c);
        for (int64_t tau = 1; tau <= nT; ++tau) {
            const uint32_t step = (uint32_t)tau;
            normals_gen(normal_gen, seed, id, step, ORC_STREAM_PROPOSAL, d, z);
            for (int k = 0; k < d; ++k) y[k] = FMA(sigma, z[k], FMA(h, gx[k], x[k]));
            const real fwd = lanes_sumsq(z, NULL, d, Lw);
            const real lpy = orc_target_grad(t, y, gy, user);
            for (int k = 0; k < d; ++k) tk[k] = FMA(hs, gx[k] + gy[k], z[k]);
            const real bwd = lanes_sumsq(tk, NULL, d, Lw);
            const real loga = (lpy - lp) + R(0.5) * (fwd - bwd);
            const real logu = orc_accept_logu(seed, id, step);
            const int acc = logu < loga;
            margin_note(&mg, logu, loga);
            if (acc) { memcpy(x, y, sizeof(real) * (size_t)d); memcpy(gx, gy, sizeof(real) * (size_t)d); lp = lpy; ++nacc; }
            slot = save_slot(s, tau);
            if (slot >= 0) record(samples, accepted, slot, d, C, c, x, lp, acc);
            margin_flush(&mg, slot, C, c);
        }
        if (final_x) for (int k = 0; k < d; ++k) final_x[(size_t)k * C + c] = x[k];
        if (final_lp) final_lp[c] = lp;
        if (accept_counts) accept_counts[c] = nacc;
    }
    free(x);
    return 0;
}
