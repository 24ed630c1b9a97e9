// mhx_device_math.h -- device-side arithmetic of the MHX engine (gfx950, wave64).
//
// Everything a lane needs to turn (seed, chain id, step) into proposal noise and an accept
// threshold: Philox4x32-10, uniform conversions, and fp32 log / exp / sincos(2*pi*k/2^32) written
// as explicit fmaf polynomials.  The translation unit is compiled with
// -ffp-contract=off, so the only fused operations are the __builtin_fmaf calls below and every
// other +,*,/ and sqrt rounds once (IEEE, correctly rounded) -- that is what makes a chain's
// trajectory a pure function of (seed, chain id) that a host restatement can reproduce bit for bit.
//
// This header is compiled twice: by hipcc into libmhx.so (pre-built kernels) and by hiprtc at run
// time (specialised kernels, user log-densities); keep it free of host headers.
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#define MHX_DEV __device__ __forceinline__
#define MHX_HD __host__ __device__ __forceinline__

// The engine is compiled twice from these headers: mhx_real = float (MHX_REAL64 = 0) and mhx_real = double (MHX_REAL64 = 1:
// the reference computes in Float64 end to end -- Distributions' rand / logpdf, src/RobustAdaptiveMetropolis.jl:187-196).
// The two pre-built instantiations live in their own namespaces inside libmhx.so; a run-time build gets -DMHX_REAL64=... and no
// namespace (one module per specialisation; a user's source sits in front of the kernels' header, outside any namespace).
#ifndef MHX_REAL64
#define MHX_REAL64 0
#endif
#if MHX_REAL64
typedef double mhx_real;
#define MHX_R(x) x
#define MHX_NS mhx_f64
#else
typedef float mhx_real;
#define MHX_R(x) x##f
#define MHX_NS mhx_f32
#endif
#define MHX_RB ((unsigned)sizeof(mhx_real))              // bytes per real
#if defined(__HIPCC_RTC__) || defined(MHX_JIT_BUILD)      // (MHX_JIT_BUILD: the same run-time source through the installation's clang++)
#define MHX_NS_BEGIN
#define MHX_NS_END
#else
#define MHX_NS_BEGIN namespace MHX_NS {
#define MHX_NS_END }
#endif

// ordering of ONE wave's LDS traffic (data that only the lanes of a wave exchange: no s_barrier needed)
#define MHX_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

typedef unsigned int mhx_u32;
typedef unsigned long long mhx_u64;
#include "mhx_zig_table.h"     // generated (tools/gen_zig_table.py): the layer tables of the ziggurat normal generator (fp64: MHX_ZIG_*, fp32: MHX_ZIG32_*)
MHX_NS_BEGIN

// signature of a user log-density in HIP source form (see include/mhx.h, mhx_target_from_hip_source)
#define MHX_LOGDENSITY(x, d, data, ndata)                                                          \
    template <class MHX_X>                                                                          \
    MHX_DEV mhx_real mhx_user_logdensity(const MHX_X& x, const int d, const mhx_real* __restrict__ data, \
                                         const int ndata)

// a user gradient in HIP source form: writes g.set(k, dlp/dx_k) and returns lp
#define MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata)                                                       \
    template <class MHX_X, class MHX_G>                                                                         \
    MHX_DEV mhx_real mhx_user_logdensity_and_gradient(const MHX_X& x, const MHX_G& g, const int d,              \
                                                      const mhx_real* __restrict__ data, const int ndata)

// wave-uniform base pointer + 32-bit per-lane BYTE offset: lowers to the scalar-base addressing mode
// (global_load/store v_off, ..., s[base:base+1]) instead of a 64-bit vector address per access
MHX_DEV mhx_real mhx_ld_off(const mhx_real* base, mhx_u32 byte_off)
{
    return *(const mhx_real*)((const char*)base + byte_off);
}
MHX_DEV void mhx_st_off(mhx_real* base, mhx_u32 byte_off, mhx_real v) { *(mhx_real*)((char*)base + byte_off) = v; }

// A [rows][ld] slab of reals addressed through a buffer descriptor: wave-uniform base in the SRD, the row
// offset in an SGPR (soffset), the lane's column offset in one VGPR (voffset) -- no per-access 64-bit
// vector address arithmetic.  `bytes` bounds the slab (hardware range check).
typedef __amdgpu_buffer_rsrc_t mhx_srd;
MHX_DEV mhx_srd mhx_make_srd(const void* base, mhx_u32 bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
#if MHX_REAL64
typedef mhx_u32 mhx_u32v2 __attribute__((ext_vector_type(2)));
#ifndef MHX_SRD_STORE_AUX
#define MHX_SRD_STORE_AUX 0     // cache-policy bits of the slab stores (tuning knob: 1 sc0, 2 nt, 16 sc1)
#endif
template <int AUX = MHX_SRD_STORE_AUX>
MHX_DEV void mhx_srd_store(mhx_srd srd, mhx_u32 lane_byte_off, mhx_u32 row_byte_off, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(mhx_u32v2, v), srd, (int)lane_byte_off, (int)row_byte_off, AUX);
}
#else
template <int AUX = 0>
MHX_DEV void mhx_srd_store(mhx_srd srd, mhx_u32 lane_byte_off, mhx_u32 row_byte_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(mhx_u32, v), srd, (int)lane_byte_off, (int)row_byte_off, AUX);
}
#endif
// The sample record is written once and not read again by the kernel that writes it: non-temporal stores (aux bit 1) -- C2 3.00 ->
// 2.95 ms per launch (profiles/r04x_record_ab.log).  (The RAM factor stores are the opposite case: MHX_SRD_STORE_AUX)
#ifndef MHX_REC_STORE_AUX
#define MHX_REC_STORE_AUX 2
#endif

#if MHX_REAL64
MHX_DEV double mhx_srd_load(mhx_srd srd, mhx_u32 lane_byte_off, mhx_u32 row_byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(srd, (int)lane_byte_off, (int)row_byte_off, 0));
}
#else
MHX_DEV float mhx_srd_load(mhx_srd srd, mhx_u32 lane_byte_off, mhx_u32 row_byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, (int)lane_byte_off, (int)row_byte_off, 0));
}
#endif

// v_readlane of a real (two dwords in fp64)
MHX_DEV mhx_real mhx_readlane(const mhx_real x, const int lane)
{
#if MHX_REAL64
    const mhx_u64 b = __builtin_bit_cast(mhx_u64, x);
    const mhx_u32 lo = (mhx_u32)__builtin_amdgcn_readlane((int)(mhx_u32)b, lane);
    const mhx_u32 hi = (mhx_u32)__builtin_amdgcn_readlane((int)(mhx_u32)(b >> 32), lane);
    return __builtin_bit_cast(double, ((mhx_u64)hi << 32) | lo);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
#endif
}

// RNG stream tags: counter word 3 = tag << 28 | block
#define MHX_STREAM_PROPOSAL 0u
#define MHX_STREAM_ACCEPT   1u
#define MHX_STREAM_INIT     2u
#define MHX_STREAM_EMCEE    3u

struct mhx_u32x4 { mhx_u32 x, y, z, w; };

// ---------------------------------------------------------------------------------------------
// Philox4x32-10.  The key schedule (seed + r * Weyl) is wave-uniform and lives in SGPRs; the two
// 32x32->64 products per round are the only slow-rate VALU ops.
struct mhx_philox_key { mhx_u32 k0[10], k1[10]; };

MHX_DEV mhx_philox_key mhx_philox_schedule(mhx_u64 seed)
{
    mhx_philox_key ks;
    mhx_u32 a = (mhx_u32)seed, b = (mhx_u32)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { ks.k0[r] = a; ks.k1[r] = b; a += 0x9E3779B9u; b += 0xBB67AE85u; }
    return ks;
}

// a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96); hipcc leaves it as two v_xor_b32
MHX_DEV mhx_u32 mhx_xor3(mhx_u32 a, mhx_u32 b, mhx_u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

MHX_DEV mhx_u32x4 mhx_philox(const mhx_philox_key& ks, mhx_u32 c0, mhx_u32 c1, mhx_u32 c2, mhx_u32 c3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const mhx_u64 p0 = (mhx_u64)0xD2511F53u * c0;
        const mhx_u64 p1 = (mhx_u64)0xCD9E8D57u * c2;
        const mhx_u32 n0 = mhx_xor3((mhx_u32)(p1 >> 32), c1, ks.k0[r]);
        const mhx_u32 n2 = mhx_xor3((mhx_u32)(p0 >> 32), c3, ks.k1[r]);
        c1 = (mhx_u32)p1; c3 = (mhx_u32)p0; c0 = n0; c2 = n2;
    }
    mhx_u32x4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

#if MHX_REAL64
// ---------------------------------------------------------------------------------------------
// fp64 arithmetic spec (coefficients: tools/fit_coeffs64.py; the CPU checker restates the same literals)
MHX_DEV mhx_u64 mhx_d2u(double f) { return __builtin_bit_cast(mhx_u64, f); }
MHX_DEV double  mhx_u2d(mhx_u64 u) { return __builtin_bit_cast(double, u); }
MHX_DEV double  mhx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

#define MHX_LN2_HI 0x1.62e42feep-1
#define MHX_LN2_LO 0x1.a39ef35793c76p-33
#define MHX_LOG2E  0x1.71547652b82fep+0
#define MHX_INF    __builtin_inf()
#define MHX_NAN    __builtin_nan("")

// a / b for finite normal operands whose quotient is normal (or a == 0): the correctly rounded sequence hipcc emits for an
// fp64 division -- reciprocal estimate, two Newton steps, quotient, residual, final fma -- without its range scaling
// (v_div_scale x 2) and special-case fix-up (v_div_fixup), which are the identity in that range: 8 instead of 11 instructions
MHX_DEV double mhx_div_normal(const double a, const double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = mhx_fma(-b, y, 1.0);
    y = mhx_fma(y, e, y);
    e = mhx_fma(-b, y, 1.0);
    y = mhx_fma(y, e, y);
    const double q = a * y;
    const double r = mhx_fma(-b, q, a);
    return mhx_fma(r, y, q);
}

// m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), z = s^2:
//   log(1 + f) = 2 s + s z P(z) = f - hfsq + s (hfsq + z P(z)),  hfsq = f^2 / 2       (one correctly rounded division)
MHX_DEV double mhx_log_core(mhx_u64 ix, const int eadj)
{
    const mhx_u64 t = ix - 0x3fe6a09e667f3bcdull;
    const long long e = (long long)t >> 52;
    const double m = mhx_u2d(ix - ((mhx_u64)e << 52));
    const double f = m - 1.0;
    const double s = mhx_div_normal(f, 2.0 + f);       // 2 + f in [1.7, 2.42]; f == 0 or |f| >= 2^-53
    const double z = s * s;
    double p = 0x1.2b59b70eb76c6p-3;
    p = mhx_fma(p, z, 0x1.39fe42e9d4a8ap-3);
    p = mhx_fma(p, z, 0x1.7462b58e4403ap-3);
    p = mhx_fma(p, z, 0x1.c71c62e26212ep-3);
    p = mhx_fma(p, z, 0x1.2492492df3ba9p-2);
    p = mhx_fma(p, z, 0x1.99999999952ccp-2);
    p = mhx_fma(p, z, 0x1.5555555555558p-1);
    const double hfsq = (0.5 * f) * f;
    const double ef = (double)((int)e + eadj);
    const double t1 = s * mhx_fma(z, p, hfsq);
    const double t2 = mhx_fma(ef, MHX_LN2_LO, t1);
    const double t3 = hfsq - t2;
    const double t4 = f - t3;
    return mhx_fma(ef, MHX_LN2_HI, t4);
}
// log for a positive, normal, finite argument (every uniform we draw): no special cases
MHX_DEV double mhx_log_pos(double x) { return mhx_log_core(mhx_d2u(x), 0); }
// full-range log (user log-densities, emcee's log z)
MHX_DEV double mhx_log(double x)
{
    mhx_u64 ix = mhx_d2u(x);
    if ((ix << 1) == 0ull) return -MHX_INF;
    if (ix >> 63) return MHX_NAN;
    if (ix >= 0x7ff0000000000000ull) return x;
    int eadj = 0;
    if (ix < 0x0010000000000000ull) { x = x * 0x1p54; ix = mhx_d2u(x); eadj = -54; }
    return mhx_log_core(ix, eadj);
}
// the same function without a branch -- special cases by selects AFTER the arithmetic ran on (possibly meaningless) bits -- for kernels
// that evaluate several logarithms side by side (mhx_rwmh_wave_body): early returns would serialise them.  Same value for every input.
MHX_DEV double mhx_log_sel(const double x)
{
    const mhx_u64 ix0 = mhx_d2u(x);
    const bool sub = ix0 < 0x0010000000000000ull;
    const double xs = sub ? x * 0x1p54 : x;
    double r = mhx_log_core(mhx_d2u(xs), sub ? -54 : 0);
    r = ix0 >= 0x7ff0000000000000ull ? x : r;
    r = (ix0 >> 63) ? MHX_NAN : r;
    r = (ix0 << 1) == 0ull ? -MHX_INF : r;
    return r;
}

MHX_DEV double mhx_exp(double x)
{
    if (x != x) return x;
    if (x > 0x1.62e42fefa39efp+9) return MHX_INF;
    if (x < -0x1.74910d52d3052p+9) return 0.0;
    const double n = __builtin_rint(x * MHX_LOG2E);
    double r = mhx_fma(n, -MHX_LN2_HI, x);
    r = mhx_fma(n, -MHX_LN2_LO, r);
    double p = 0x1.61bfaa228dde5p-33;
    p = mhx_fma(p, r, 0x1.1f7f2776cfaf2p-29);
    p = mhx_fma(p, r, 0x1.ae642c82e33d5p-26);
    p = mhx_fma(p, r, 0x1.27e4d41966f2fp-22);
    p = mhx_fma(p, r, 0x1.71de3a5aa7bb7p-19);
    p = mhx_fma(p, r, 0x1.a01a01a9e991bp-16);
    p = mhx_fma(p, r, 0x1.a01a01a0196acp-13);
    p = mhx_fma(p, r, 0x1.6c16c16c15a68p-10);
    p = mhx_fma(p, r, 0x1.1111111111111p-7);
    p = mhx_fma(p, r, 0x1.5555555555557p-5);
    p = mhx_fma(p, r, 0x1.5555555555555p-3);
    p = mhx_fma(p, r, 0.5);
    const double r2 = r * r;
    double y = mhx_fma(r2, p, r) + 1.0;
    const int ni = (int)n;
    const int n1 = ni / 2;
    const int n2 = ni - n1;
    y = y * mhx_u2d((mhx_u64)(n1 + 1023) << 52);
    y = y * mhx_u2d((mhx_u64)(n2 + 1023) << 52);
    return y;
}

MHX_DEV double mhx_sqrt(double x) { return __builtin_sqrt(x); }   // correctly rounded (default HIP lowering)
MHX_DEV double mhx_abs(double x) { return __builtin_fabs(x); }
// Correctly rounded sqrt for a NORMAL positive x away from the ends of the exponent range (the Box-Muller radius argument
// -2 ln u lies in [2.2e-16, 74]): the sequence hipcc emits -- v_rsq_f64, one coupled Goldschmidt step, two residual
// corrections -- without its pre/post scaling for x < 2^-767 and its class test for 0 / inf: 10 instead of 17 instructions
MHX_DEV double mhx_sqrt_normal(const double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = mhx_fma(-h, g, 0.5);
    g = mhx_fma(g, r, g);
    h = mhx_fma(h, r, h);
    double d = mhx_fma(-g, g, x);
    g = mhx_fma(d, h, g);
    d = mhx_fma(-g, g, x);
    return mhx_fma(d, h, g);
}

// sin/cos of 2 pi a / 2^64, a = hi:lo: integer quadrant reduction; the residual keeps 52 bits so that it is exact in
// a double (r = (ri >> 10) 2^-54 turns, [-1/8, 1/8)); split leading constants, compensated cos: < 0.7 ulp each
MHX_DEV void mhx_sincos2pi_u64(mhx_u32 hi, mhx_u32 lo, double& s, double& c)
{
    const mhx_u64 a = ((mhx_u64)hi << 32) | lo;
    const mhx_u64 kk = a + 0x2000000000000000ull;
    const mhx_u32 q = (mhx_u32)(kk >> 62);
    const long long ri = (long long)(kk & 0x3fffffffffffffffull) - 0x2000000000000000ll;
    const long long ti = ri >> 10;                                        // [-2^51, 2^51)
    // exact integer -> double: 2^52 + 2^51 + ti is representable, subtract the bias again
    const double r = (mhx_u2d(0x4338000000000000ull + (mhx_u64)ti) - 0x1.8p+52) * 0x1p-54;
    const double u = r * r;
    double s1 = -0x1.6cc577dadd922p-1;
    s1 = mhx_fma(s1, u, 0x1.e8f036bcd3237p+1);
    s1 = mhx_fma(s1, u, -0x1.e3074d2614b2dp+3);
    s1 = mhx_fma(s1, u, 0x1.50783486facaap+5);
    s1 = mhx_fma(s1, u, -0x1.32d2cce62b872p+6);
    s1 = mhx_fma(s1, u, 0x1.466bc6775aae1p+6);
    s1 = mhx_fma(s1, u, -0x1.4abbce625be53p+5);
    const double ts = (r * u) * s1;
    const double sp = mhx_fma(r, 0x1.921fb54442d18p+2, mhx_fma(r, 0x1.1a62633145c07p-52, ts));
    double c2 = 0x1.1ebe62242e9d8p-2;
    c2 = mhx_fma(c2, u, -0x1.b6df855cc99ffp+0);
    c2 = mhx_fma(c2, u, 0x1.f9d38850e5eedp+2);
    c2 = mhx_fma(c2, u, -0x1.a6d1f2a15a701p+4);
    c2 = mhx_fma(c2, u, 0x1.e1f506891b72fp+5);
    c2 = mhx_fma(c2, u, -0x1.55d3c7e3cbffap+6);
    c2 = mhx_fma(c2, u, 0x1.03c1f081b5ac4p+6);
    const double wc = (u * u) * c2;
    const double vc = mhx_fma(u, -0x1.692b71366cc04p-50, wc);
    const double ac = mhx_fma(u, -0x1.3bd3cc9be45dep+4, 1.0);
    const double ec = mhx_fma(u, -0x1.3bd3cc9be45dep+4, 1.0 - ac);
    const double cp = ac + (vc + ec);
    const bool odd = (q & 1u) != 0u;
    const double ss = odd ? cp : sp;
    const double cc = odd ? sp : cp;
    const mhx_u64 sneg = (mhx_u64)(q & 2u) << 62;
    const mhx_u64 cneg = (mhx_u64)((q + 1u) & 2u) << 62;
    s = mhx_u2d(mhx_d2u(ss) ^ sneg);
    c = mhx_u2d(mhx_d2u(cc) ^ cneg);
}

// 52-bit uniforms from two Philox words, k = hi:lo >> 12: (k + 1/2) 2^-52 in (0,1) and k 2^-52 in [0,1), both exact
MHX_DEV double mhx_u01_open(mhx_u32 hi, mhx_u32 lo)
{
    const mhx_u64 k = ((mhx_u64)hi << 20) | (lo >> 12);
    return mhx_u2d(0x3ff0000000000000ull | k) - 0x1.fffffffffffffp-1;       // (1 + k 2^-52) - (1 - 2^-53)
}
MHX_DEV double mhx_u01_half(mhx_u32 hi, mhx_u32 lo)
{
    const mhx_u64 k = ((mhx_u64)hi << 20) | (lo >> 12);
    return mhx_u2d(0x3ff0000000000000ull | k) - 1.0;
}

// Box-Muller from one Philox block: radius from (x, y), angle from (z, w)
MHX_DEV void mhx_normal_pair(const mhx_u32x4& w, double& n0, double& n1)
{
    const double l = mhx_log_pos(mhx_u01_open(w.x, w.y));
    const double rad = mhx_sqrt_normal(-2.0 * l);
    double s, c;
    mhx_sincos2pi_u64(w.z, w.w, s, c);
    n0 = rad * c;
    n1 = rad * s;
}

// the 4 standard normals 4b..4b+3 of (id, step, stream): Philox blocks 2b and 2b+1
MHX_DEV void mhx_normal4(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step,
                         mhx_u32 stream, mhx_u32 block, double n[4])
{
    const mhx_u32x4 w0 = mhx_philox(ks, id_lo, id_hi, step, (stream << 28) | (2u * block));
    mhx_normal_pair(w0, n[0], n[1]);
    const mhx_u32x4 w1 = mhx_philox(ks, id_lo, id_hi, step, (stream << 28) | (2u * block + 1u));
    mhx_normal_pair(w1, n[2], n[3]);
}

// log of the accept uniform of `step`: one Philox block serves 2 consecutive steps.
struct mhx_accept_cache { mhx_u32x4 w; mhx_u32 group; };
#define MHX_ACCEPT_GROUP_SHIFT 1

MHX_DEV double mhx_accept_logu(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step,
                               mhx_accept_cache& cache)
{
    const mhx_u32 g = step >> 1;
    if (g != cache.group) {                           // wave-uniform
        cache.w = mhx_philox(ks, id_lo, id_hi, g, MHX_STREAM_ACCEPT << 28);
        cache.group = g;
    }
    const bool odd = (step & 1u) != 0u;               // wave-uniform select
    return mhx_log_pos(mhx_u01_open(odd ? cache.w.z : cache.w.x, odd ? cache.w.w : cache.w.y));
}

// ---------------------------------------------------------------------------------------------
// The ZIGGURAT normal generator of the fp64 spec (Marsaglia & Tsang 2000 in Doornik's ZIGNOR
// form): MHX_ZIG_N equal-area layers under exp(-x^2/2), table x[0..N] (mhx_zig_table.h: x[0] = v/f(r), x[1] = r, x[N] = 0).
// A normal takes 64 bits like a Box-Muller normal does -- Philox block p of (id, step, stream) serves normals 2p (words
// x, y) and 2p + 1 (words z, w):
//   layer = lo mod N (bits 0..9);  sign = bit 31 of lo;  u = k 2^-52 in [0, 1) with the 52-bit k = (bits 11..30 of lo) : hi
//   (laid out so that the double 1 + u is {0x3ff00000 | bits 11..30 of lo, hi}: no 64-bit shift);  |x| = u x[layer];
//   accept at once when |x| < x[layer + 1]  (99.6 % of the draws: a table look-up, a multiply and a compare);
// otherwise rejection attempts t = 1, 2, ... from Philox block (n << 8 | t) of stream | 4, n = index of the normal in
// its step:  layer 0: the tail beyond r (Marsaglia) -- xx = -log(U1)/r, yy = -log(U2), accept r + xx iff 2 yy >= xx^2;
// else the wedge -- accept x iff f1 + U (f0 - f1) < 1 with f0 = exp(-(x_l^2 - x^2)/2), f1 = exp(-(x_{l+1}^2 - x^2)/2),
// U from words (z, w); on rejection words (x, y) of the same block are the next candidate.
#define MHX_STREAM_RETRY 4u
#define MHX_GEN_BOX_MULLER 0
#define MHX_GEN_ZIGGURAT 1
__device__ const double mhx_zig_x[MHX_ZIG_N + 1] = MHX_ZIG_TABLE;

// the candidate's uniform: k 2^-52, k = (bits 11..30 of lo) : hi
MHX_DEV double mhx_zig_u(const mhx_u32 hi, const mhx_u32 lo)
{
    const mhx_u32 top = 0x3ff00000u | ((lo >> 11) & 0xfffffu);
    return mhx_u2d(((mhx_u64)top << 32) | hi) - 1.0;
}
// u x_l in ONE operation: with m = 1 + u (exact) the product m x_l - x_l == u x_l exactly, and the fma rounds it once --
// the same bits as the rounded product mhx_zig_u(hi, lo) * xl of the spec, without the subtraction
MHX_DEV double mhx_zig_ax(const mhx_u32 hi, const mhx_u32 lo, const double xl)
{
    const mhx_u32 top = 0x3ff00000u | ((lo >> 11) & 0xfffffu);
    return mhx_fma(mhx_u2d(((mhx_u64)top << 32) | hi), xl, -xl);
}
// |x| with the candidate's sign (bit 31 of lo); ax >= 0.  `sign` = 0x80000000: handed in by the hot loop from a scalar register
// that hipcc cannot see through, because gfx950 has the three-operand (lo & sign) | hi -- v_and_or_b32 -- but no literal
// constants in that encoding: with the constant in the source the pair stays v_and_b32 + v_or_b32
MHX_DEV double mhx_zig_signed(const double ax, const mhx_u32 lo, const mhx_u32 sign = 0x80000000u)
{
    const mhx_u64 b = mhx_d2u(ax);
    return mhx_u2d(((mhx_u64)((mhx_u32)(b >> 32) | (lo & sign)) << 32) | (mhx_u32)b);
}
MHX_DEV bool mhx_zig_try(const double* __restrict__ zt, const mhx_u32 hi, const mhx_u32 lo, double& x, mhx_u32& layer)
{
    layer = lo & (mhx_u32)(MHX_ZIG_N - 1);
    const double ax = mhx_zig_ax(hi, lo, zt[layer]);
    x = mhx_zig_signed(ax, lo);
    return ax < zt[layer + 1];
}

// the normal behind a candidate (x, layer) that left its rectangle: rejection attempts t0, t0 + 1, ...
MHX_DEV double mhx_zig_slow(const mhx_philox_key& ks, const double* __restrict__ zt, const mhx_u32 id_lo, const mhx_u32 id_hi,
                            const mhx_u32 step, const mhx_u32 stream, const mhx_u32 n, double x, mhx_u32 layer, const mhx_u32 t0 = 1u)
{
    for (mhx_u32 t = t0;; ++t) {
        const mhx_u32x4 v = mhx_philox(ks, id_lo, id_hi, step, ((stream | MHX_STREAM_RETRY) << 28) | ((n << 8) | (t & 255u)));
        if (layer == 0u) {
            const double xx = mhx_log_pos(mhx_u01_open(v.x, v.y)) * MHX_ZIG_NEG_RINV;
            const double yy = -mhx_log_pos(mhx_u01_open(v.z, v.w));
            if (yy + yy >= xx * xx) return mhx_u2d(mhx_d2u(MHX_ZIG_R + xx) | (mhx_d2u(x) & 0x8000000000000000ull));
        } else {
            const double xl = zt[layer], xl1 = zt[layer + 1], xsq = x * x;
            const double f0 = mhx_exp(-0.5 * (xl * xl - xsq)), f1 = mhx_exp(-0.5 * (xl1 * xl1 - xsq));
            if (mhx_fma(mhx_u01_half(v.z, v.w), f0 - f1, f1) < 1.0) return x;
            if (mhx_zig_try(zt, v.x, v.y, x, layer)) return x;
        }
    }
}

// The same normal, laid out for latency (the fix-up pass of the cooperative kernel runs a dozen lanes of a wave through this while
// the other lanes wait): the Philox block of the failed candidate and the block of rejection attempt 1 are independent of each
// other and of the table -- both are drawn up front, the table entries of the failed candidate AND of attempt 1's fresh candidate
// are fetched together, and the common case (a wedge, settled by attempt 1) runs straight through; anything else (the tail beyond
// r, a second rejection) continues in mhx_zig_slow from the attempt it has reached.  Same values as mhx_zig_try + mhx_zig_slow.
MHX_DEV double mhx_zig_refine(const mhx_philox_key& ks, const double* __restrict__ zt, const mhx_u32 id_lo, const mhx_u32 id_hi,
                              const mhx_u32 step, const mhx_u32 stream, const mhx_u32 n)
{
    const mhx_u32x4 w = mhx_philox(ks, id_lo, id_hi, step, (stream << 28) | (n >> 1));
    const mhx_u32x4 v = mhx_philox(ks, id_lo, id_hi, step, ((stream | MHX_STREAM_RETRY) << 28) | ((n << 8) | 1u));
    const mhx_u32 hi = (n & 1u) ? w.z : w.x, lo = (n & 1u) ? w.w : w.y;
    mhx_u32 layer = lo & (mhx_u32)(MHX_ZIG_N - 1);
    const mhx_u32 layer2 = v.y & (mhx_u32)(MHX_ZIG_N - 1);
    const double xl = zt[layer], xl1 = zt[layer + 1];
    const double xn = zt[layer2], xn1 = zt[layer2 + 1];
    const double ax = mhx_zig_ax(hi, lo, xl);
    double x = mhx_zig_signed(ax, lo);
    if (ax < xl1) return x;                                        // (not a failed candidate after all: callers only send failures)
    if (layer == 0u) return mhx_zig_slow(ks, zt, id_lo, id_hi, step, stream, n, x, layer, 1u);
    const double xsq = x * x;
    const double f0 = mhx_exp(-0.5 * (xl * xl - xsq)), f1 = mhx_exp(-0.5 * (xl1 * xl1 - xsq));
    if (mhx_fma(mhx_u01_half(v.z, v.w), f0 - f1, f1) < 1.0) return x;
    const double ax2 = mhx_zig_ax(v.x, v.y, xn);
    x = mhx_zig_signed(ax2, v.y);
    if (ax2 < xn1) return x;
    return mhx_zig_slow(ks, zt, id_lo, id_hi, step, stream, n, x, layer2, 2u);
}

// normal number n (0-based) of (id, step, stream), straight from the table in global memory: the kernels off the hot path
MHX_DEV double mhx_zig_normal(const mhx_philox_key& ks, const mhx_u32 id_lo, const mhx_u32 id_hi, const mhx_u32 step,
                              const mhx_u32 stream, const mhx_u32 n)
{
    const mhx_u32x4 w = mhx_philox(ks, id_lo, id_hi, step, (stream << 28) | (n >> 1));
    double x; mhx_u32 layer;
    if (mhx_zig_try(mhx_zig_x, (n & 1u) ? w.z : w.x, (n & 1u) ? w.w : w.y, x, layer)) return x;
    return mhx_zig_slow(ks, mhx_zig_x, id_lo, id_hi, step, stream, n, x, layer);
}
// the 4 normals 4b..4b+3 by either generator (lane-per-chain kernels: initial draws, generic paths)
MHX_DEV void mhx_normal4_gen(const int gen, const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step,
                             mhx_u32 stream, mhx_u32 block, double n[4])
{
    if (gen == MHX_GEN_ZIGGURAT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) n[j] = mhx_zig_normal(ks, id_lo, id_hi, step, stream, 4u * block + (mhx_u32)j);
    } else {
        mhx_normal4(ks, id_lo, id_hi, step, stream, block, n);
    }
}

// the draws of one stretch move (src/emcee.jl:48,52 partner, :81 stretch uniform, :93 accept)
struct mhx_emcee_draws { mhx_u32 partner; double u, logu; };
MHX_DEV mhx_emcee_draws mhx_emcee_draw(const mhx_philox_key& ks, mhx_u32 walker, mhx_u32 ens, mhx_u32 sweep)
{
    const mhx_u32x4 w = mhx_philox(ks, walker, ens, sweep, MHX_STREAM_EMCEE << 28);
    const mhx_u32x4 v = mhx_philox(ks, walker, ens, sweep, (MHX_STREAM_EMCEE << 28) | 1u);
    mhx_emcee_draws o;
    o.partner = w.x;
    o.u = mhx_u01_half(w.y, w.z);
    o.logu = mhx_log_pos(mhx_u01_open(v.x, v.y));
    return o;
}
#else
// ---------------------------------------------------------------------------------------------
MHX_DEV mhx_u32 mhx_f2u(float f) { return __builtin_bit_cast(mhx_u32, f); }
MHX_DEV float   mhx_u2f(mhx_u32 u) { return __builtin_bit_cast(float, u); }
MHX_DEV float   mhx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

#define MHX_LN2_HI 0x1.62e4p-1f
#define MHX_LN2_LO 0x1.7f7d1cp-20f
#define MHX_LOG2E  0x1.715476p+0f
#define MHX_INF    __builtin_inff()
#define MHX_NAN    __builtin_nanf("")

// log for a positive, normal, finite argument (every uniform we draw): no special cases
MHX_DEV float mhx_log_pos(float x)
{
    const mhx_u32 ix = mhx_f2u(x);
    const mhx_u32 t = ix - 0x3f2aaaabu;
    const int e = (int)t >> 23;
    const float m = mhx_u2f(ix - ((mhx_u32)e << 23));
    const float f = m - 1.0f;
    const float ef = (float)e;
    float q = -0x1.04cba2p-3f;
    q = mhx_fma(q, f, 0x1.19bbe2p-3f);
    q = mhx_fma(q, f, -0x1.f483fap-4f);
    q = mhx_fma(q, f, 0x1.1fd494p-3f);
    q = mhx_fma(q, f, -0x1.55913ep-3f);
    q = mhx_fma(q, f, 0x1.99bffep-3f);
    q = mhx_fma(q, f, -0x1.ffff28p-3f);
    q = mhx_fma(q, f, 0x1.55552cp-2f);
    q = mhx_fma(q, f, -0.5f);
    const float f2 = f * f;
    float r = mhx_fma(f2, q, f);
    r = mhx_fma(ef, MHX_LN2_LO, r);
    r = mhx_fma(ef, MHX_LN2_HI, r);
    return r;
}

// full-range log (user log-densities, emcee's log z)
MHX_DEV float mhx_log_core(const mhx_u32 ix, const int eadj)
{
    // same polynomial as mhx_log_pos; the exponent correction is folded in before the conversion
    const mhx_u32 t = ix - 0x3f2aaaabu;
    const int e = (int)t >> 23;
    const float m = mhx_u2f(ix - ((mhx_u32)e << 23));
    const float f = m - 1.0f;
    const float ef = (float)(e + eadj);
    float q = -0x1.04cba2p-3f;
    q = mhx_fma(q, f, 0x1.19bbe2p-3f);
    q = mhx_fma(q, f, -0x1.f483fap-4f);
    q = mhx_fma(q, f, 0x1.1fd494p-3f);
    q = mhx_fma(q, f, -0x1.55913ep-3f);
    q = mhx_fma(q, f, 0x1.99bffep-3f);
    q = mhx_fma(q, f, -0x1.ffff28p-3f);
    q = mhx_fma(q, f, 0x1.55552cp-2f);
    q = mhx_fma(q, f, -0.5f);
    const float f2 = f * f;
    float r = mhx_fma(f2, q, f);
    r = mhx_fma(ef, MHX_LN2_LO, r);
    r = mhx_fma(ef, MHX_LN2_HI, r);
    return r;
}
MHX_DEV float mhx_log(float x)
{
    mhx_u32 ix = mhx_f2u(x);
    if ((ix << 1) == 0u) return -MHX_INF;
    if (ix >> 31) return MHX_NAN;
    if (ix >= 0x7f800000u) return x;
    int eadj = 0;
    if (ix < 0x00800000u) { x = x * 0x1p23f; eadj = -23; }
    return mhx_log_core(mhx_f2u(x), eadj);
}
// the same function without a branch (see the fp64 twin): special cases by selects after the arithmetic
MHX_DEV float mhx_log_sel(const float x)
{
    const mhx_u32 ix0 = mhx_f2u(x);
    const bool sub = ix0 < 0x00800000u;
    const float xs = sub ? x * 0x1p23f : x;
    float r = mhx_log_core(mhx_f2u(xs), sub ? -23 : 0);
    r = ix0 >= 0x7f800000u ? x : r;
    r = (ix0 >> 31) ? MHX_NAN : r;
    r = (ix0 << 1) == 0u ? -MHX_INF : r;
    return r;
}

MHX_DEV float mhx_exp(float x)
{
    if (x != x) return x;
    if (x > 0x1.62e42ep+6f) return MHX_INF;
    if (x < -0x1.9fe368p+6f) return 0.0f;
    const float n = __builtin_rintf(x * MHX_LOG2E);
    float r = mhx_fma(n, -MHX_LN2_HI, x);
    r = mhx_fma(n, -MHX_LN2_LO, r);
    float p = 0x1.a1517cp-13f;
    p = mhx_fma(p, r, 0x1.6d4328p-10f);
    p = mhx_fma(p, r, 0x1.1110c6p-7f);
    p = mhx_fma(p, r, 0x1.5554eap-5f);
    p = mhx_fma(p, r, 0x1.555556p-3f);
    p = mhx_fma(p, r, 0.5f);
    const float r2 = r * r;
    float y = mhx_fma(r2, p, r) + 1.0f;
    const int ni = (int)n;
    const int n1 = ni / 2;
    const int n2 = ni - n1;
    y = y * mhx_u2f((mhx_u32)(n1 + 127) << 23);
    y = y * mhx_u2f((mhx_u32)(n2 + 127) << 23);
    return y;
}

MHX_DEV float mhx_sqrt(float x) { return __builtin_sqrtf(x); }   // correctly rounded (default HIP lowering)
MHX_DEV float mhx_abs(float x) { return __builtin_fabsf(x); }

// Correctly rounded sqrt for x that is +-0 or a NORMAL positive number: the hardware estimate (1 ulp) and
// the same two-residual fix-up hipcc emits, without its denormal pre-scaling and class test (7 of 17
// instructions).  The Box-Muller radius argument -2 ln u is 0 or >= 1.19e-7.
MHX_DEV float mhx_sqrt_normal(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = mhx_u2f(mhx_f2u(s) - 1u), sp = mhx_u2f(mhx_f2u(s) + 1u);
    const float rm = mhx_fma(-sm, s, x), rp = mhx_fma(-sp, s, x);
    float r = (0.0f >= rm) ? sm : s;
    r = (0.0f < rp) ? sp : r;
    return r;
}

// sin/cos of 2*pi*k/2^32: integer quadrant reduction, degree-4 polynomials in r^2 on |r| <= 1/8 turn
MHX_DEV void mhx_sincos2pi_u32(mhx_u32 k, float& s, float& c)
{
    const mhx_u32 kk = k + 0x20000000u;
    const mhx_u32 q = kk >> 30;
    const int ri = (int)(kk & 0x3fffffffu) - 0x20000000;
    const float r = (float)ri * 0x1p-32f;
    const float r2 = r * r;
    float sp = 0x1.4bc87cp+5f;
    sp = mhx_fma(sp, r2, -0x1.32ca9ep+6f);
    sp = mhx_fma(sp, r2, 0x1.466bbap+6f);
    sp = mhx_fma(sp, r2, -0x1.4abbcep+5f);
    sp = mhx_fma(sp, r2, 0x1.921fb6p+2f);
    sp = sp * r;
    float cp = 0x1.d9c326p+5f;
    cp = mhx_fma(cp, r2, -0x1.55c57ap+6f);
    cp = mhx_fma(cp, r2, 0x1.03c1dcp+6f);
    cp = mhx_fma(cp, r2, -0x1.3bd3ccp+4f);
    cp = mhx_fma(cp, r2, 1.0f);
    const bool odd = (q & 1u) != 0u;
    float ss = odd ? cp : sp;
    float cc = odd ? sp : cp;
    // sign flips as integer xors on the sign bit: q in {2,3} negates sin, q in {1,2} negates cos
    const mhx_u32 sneg = (q & 2u) << 30;
    const mhx_u32 cneg = ((q + 1u) & 2u) << 30;
    s = mhx_u2f(mhx_f2u(ss) ^ sneg);
    c = mhx_u2f(mhx_f2u(cc) ^ cneg);
}

MHX_DEV float mhx_u01_open(mhx_u32 k) { return mhx_fma((float)k, 0x1p-32f, 0x1p-33f); }   // (0,1]
MHX_DEV float mhx_u01_half(mhx_u32 k) { return (float)(k >> 8) * 0x1p-24f; }               // [0,1)

// Box-Muller: radius from k0, angle from k1
MHX_DEV void mhx_normal_pair(mhx_u32 k0, mhx_u32 k1, float& n0, float& n1)
{
    const float l = mhx_log_pos(mhx_u01_open(k0));
    const float rad = mhx_sqrt_normal(-2.0f * l);
    float s, c;
    mhx_sincos2pi_u32(k1, s, c);
    n0 = rad * c;
    n1 = rad * s;
}

// the 4 standard normals 4b..4b+3 of (id, step, stream)
MHX_DEV void mhx_normal4(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step,
                         mhx_u32 stream, mhx_u32 block, float n[4])
{
    const mhx_u32x4 w = mhx_philox(ks, id_lo, id_hi, step, (stream << 28) | block);
    mhx_normal_pair(w.x, w.y, n[0], n[1]);
    mhx_normal_pair(w.z, w.w, n[2], n[3]);
}
// ---------------------------------------------------------------------------------------------
// The ZIGGURAT normal generator in fp32 (round 6; DESIGN.md section 3.11, fp32 form; the fp64 form is above): 256 equal-area layers,
// table x[0..256] (mhx_zig_table.h: MHX_ZIG32_*), ONE 32-bit word per normal -- Philox block p of (id, step, stream) serves normals
// 4p .. 4p+3 from its words x, y, z, w (the block a Box-Muller step spends on the same four normals).  The word, laid out so that
// every field is used where it lies:
//   k = bits 0..22: u = k 2^-23 in [0, 1), the float 1 + u is 0x3f800000 | k -- the mantissa in place, no shift;
//   layer = bits 23..30;  sign = bit 31: the nine bits (w >> 23) index a table of SIGNED pairs (+-x[layer], x[layer + 1]), so the
//   candidate x = u (+-x[layer]) comes out of one fma with its sign and the test |x| < x[layer + 1] (98.5 % of the draws pass) takes
//   the magnitude as an operand modifier: and_or, shift, and, fma, compare per normal;
// otherwise rejection attempts t = 1, 2, ... from Philox block (n << 8 | t) of stream | 4: layer 0: the tail beyond r -- xx =
// -log(U(word x))/r, yy = -log(U(word z)), accept r + xx iff 2 yy >= xx^2; else the wedge -- accept x iff f1 + U(word z) (f0 - f1) < 1;
// on rejection word x of the same block is the next candidate.  Inside the kernels the macros MHX_ZIG_N / _R / _NEG_RINV mean the
// table of the engine's own width.
#define MHX_STREAM_RETRY 4u
#define MHX_GEN_BOX_MULLER 0
#define MHX_GEN_ZIGGURAT 1
#undef MHX_ZIG_N
#undef MHX_ZIG_LOG2N
#undef MHX_ZIG_R
#undef MHX_ZIG_NEG_RINV
#define MHX_ZIG_N MHX_ZIG32_N
#define MHX_ZIG_LOG2N 8
#define MHX_ZIG_R MHX_ZIG32_R
#define MHX_ZIG_NEG_RINV MHX_ZIG32_NEG_RINV
__device__ const float mhx_zig_x[MHX_ZIG_N + 1] = MHX_ZIG32_TABLE;
// The table as the kernels keep it in LDS: 2 x 256 pairs (sign s, layer l) -> { s ? -x[l] : x[l], x[l + 1] }, 8 bytes each, the pair of
// a candidate at byte (w >> 20) & 0xff8.  Entry e of the flat float array:
#define MHX_ZIG_PAIR_FLOATS (4 * MHX_ZIG_N)
MHX_DEV float mhx_zig_pair_entry(const int e)
{
    const int l = (e >> 1) & (MHX_ZIG_N - 1);
    return (e & 1) ? mhx_zig_x[l + 1] : ((e >> 1) >= MHX_ZIG_N ? -mhx_zig_x[l] : mhx_zig_x[l]);
}
// table access of the helpers below: TS = 1 the plain table x[0..N] (global memory), TS = 2 the positive half of the pair table (LDS)
template <int TS> MHX_DEV float mhx_zig_lo(const float* __restrict__ zt, const mhx_u32 layer) { return zt[TS * layer]; }
template <int TS> MHX_DEV float mhx_zig_hi(const float* __restrict__ zt, const mhx_u32 layer) { return zt[TS * layer + 1]; }

MHX_DEV mhx_u32 mhx_zig_layer(const mhx_u32 w) { return (w >> 23) & (mhx_u32)(MHX_ZIG_N - 1); }
// u x_l in ONE operation: with m = 1 + u (exact) the fma m x_l - x_l rounds u x_l once -- the rounded product of the spec
// (x_l with either sign: the fma is odd in it)
MHX_DEV float mhx_zig_ax(const mhx_u32 w, const float xl)
{
    return mhx_fma(mhx_u2f(0x3f800000u | (w & 0x7fffffu)), xl, -xl);
}
// |x| with the candidate's sign (bit 31 of its word)
MHX_DEV float mhx_zig_signed(const float ax, const mhx_u32 w, const mhx_u32 sign = 0x80000000u)
{
    return mhx_u2f(mhx_f2u(ax) | (w & sign));
}
template <int TS>
MHX_DEV bool mhx_zig_try(const float* __restrict__ zt, const mhx_u32 w, float& x, mhx_u32& layer)
{
    layer = mhx_zig_layer(w);
    const float ax = mhx_zig_ax(w, mhx_zig_lo<TS>(zt, layer));
    x = mhx_zig_signed(ax, w);
    return ax < mhx_zig_hi<TS>(zt, layer);
}
// the normal behind a candidate (x, layer) that left its rectangle: rejection attempts t0, t0 + 1, ...
template <int TS>
MHX_DEV float mhx_zig_slow(const mhx_philox_key& ks, const float* __restrict__ zt, const mhx_u32 id_lo, const mhx_u32 id_hi,
                           const mhx_u32 step, const mhx_u32 stream, const mhx_u32 n, float x, mhx_u32 layer, const mhx_u32 t0 = 1u)
{
    for (mhx_u32 t = t0;; ++t) {
        const mhx_u32x4 v = mhx_philox(ks, id_lo, id_hi, step, ((stream | MHX_STREAM_RETRY) << 28) | ((n << 8) | (t & 255u)));
        if (layer == 0u) {
            const float xx = mhx_log_pos(mhx_u01_open(v.x)) * MHX_ZIG_NEG_RINV;
            const float yy = -mhx_log_pos(mhx_u01_open(v.z));
            if (yy + yy >= xx * xx) return mhx_u2f(mhx_f2u(MHX_ZIG_R + xx) | (mhx_f2u(x) & 0x80000000u));
        } else {
            const float xl = mhx_zig_lo<TS>(zt, layer), xl1 = mhx_zig_hi<TS>(zt, layer), xsq = x * x;
            const float f0 = mhx_exp(-0.5f * (xl * xl - xsq)), f1 = mhx_exp(-0.5f * (xl1 * xl1 - xsq));
            if (mhx_fma(mhx_u01_half(v.z), f0 - f1, f1) < 1.0f) return x;
            if (mhx_zig_try<TS>(zt, v.x, x, layer)) return x;
        }
    }
}
// the same normal laid out for latency (the fix-up pass; zt = the pair table in LDS): the failed candidate's block and the block of
// attempt 1 drawn together, the table entries of both candidates fetched together, the common case -- a wedge settled by attempt 1 --
// straight through
MHX_DEV float mhx_zig_refine(const mhx_philox_key& ks, const float* __restrict__ zt, const mhx_u32 id_lo, const mhx_u32 id_hi,
                             const mhx_u32 step, const mhx_u32 stream, const mhx_u32 n)
{
    const mhx_u32x4 w4 = mhx_philox(ks, id_lo, id_hi, step, (stream << 28) | (n >> 2));
    const mhx_u32x4 v = mhx_philox(ks, id_lo, id_hi, step, ((stream | MHX_STREAM_RETRY) << 28) | ((n << 8) | 1u));
    const mhx_u32 j = n & 3u;
    const mhx_u32 w = j == 0u ? w4.x : (j == 1u ? w4.y : (j == 2u ? w4.z : w4.w));
    mhx_u32 layer = mhx_zig_layer(w);
    const mhx_u32 layer2 = mhx_zig_layer(v.x);
    const float xl = mhx_zig_lo<2>(zt, layer), xl1 = mhx_zig_hi<2>(zt, layer);
    const float xn = mhx_zig_lo<2>(zt, layer2), xn1 = mhx_zig_hi<2>(zt, layer2);
    const float ax = mhx_zig_ax(w, xl);
    float x = mhx_zig_signed(ax, w);
    if (ax < xl1) return x;                                        // (not a failed candidate after all: callers only send failures)
    if (layer == 0u) return mhx_zig_slow<2>(ks, zt, id_lo, id_hi, step, stream, n, x, layer, 1u);
    const float xsq = x * x;
    const float f0 = mhx_exp(-0.5f * (xl * xl - xsq)), f1 = mhx_exp(-0.5f * (xl1 * xl1 - xsq));
    if (mhx_fma(mhx_u01_half(v.z), f0 - f1, f1) < 1.0f) return x;
    const float ax2 = mhx_zig_ax(v.x, xn);
    x = mhx_zig_signed(ax2, v.x);
    if (ax2 < xn1) return x;
    return mhx_zig_slow<2>(ks, zt, id_lo, id_hi, step, stream, n, x, layer2, 2u);
}
// normal number n (0-based) of (id, step, stream), straight from the table in global memory: the kernels off the hot path
MHX_DEV float mhx_zig_normal(const mhx_philox_key& ks, const mhx_u32 id_lo, const mhx_u32 id_hi, const mhx_u32 step,
                             const mhx_u32 stream, const mhx_u32 n)
{
    const mhx_u32x4 w4 = mhx_philox(ks, id_lo, id_hi, step, (stream << 28) | (n >> 2));
    const mhx_u32 j = n & 3u;
    float x; mhx_u32 layer;
    if (mhx_zig_try<1>(mhx_zig_x, j == 0u ? w4.x : (j == 1u ? w4.y : (j == 2u ? w4.z : w4.w)), x, layer)) return x;
    return mhx_zig_slow<1>(ks, mhx_zig_x, id_lo, id_hi, step, stream, n, x, layer);
}
// the 4 normals 4b..4b+3 by either generator (lane-per-chain kernels: initial draws, generic paths)
MHX_DEV void mhx_normal4_gen(const int gen, const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step,
                             mhx_u32 stream, mhx_u32 block, float n[4])
{
    if (gen == MHX_GEN_ZIGGURAT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) n[j] = mhx_zig_normal(ks, id_lo, id_hi, step, stream, 4u * block + (mhx_u32)j);
    } else {
        mhx_normal4(ks, id_lo, id_hi, step, stream, block, n);
    }
}

// log of the accept uniform of `step`: one Philox block serves 4 consecutive steps.
// `cache` holds the block of step>>2; refresh it when a new group starts.
struct mhx_accept_cache { mhx_u32x4 w; mhx_u32 group; };

MHX_DEV float mhx_accept_logu(const mhx_philox_key& ks, mhx_u32 id_lo, mhx_u32 id_hi, mhx_u32 step,
                              mhx_accept_cache& cache)
{
    const mhx_u32 g = step >> 2;
    if (g != cache.group) {                           // wave-uniform
        cache.w = mhx_philox(ks, id_lo, id_hi, g, MHX_STREAM_ACCEPT << 28);
        cache.group = g;
    }
    const mhx_u32 j = step & 3u;                      // wave-uniform select
    const mhx_u32 k = j == 0u ? cache.w.x : (j == 1u ? cache.w.y : (j == 2u ? cache.w.z : cache.w.w));
    return mhx_log_pos(mhx_u01_open(k));
}

#define MHX_ACCEPT_GROUP_SHIFT 2

// the draws of one stretch move (src/emcee.jl:48,52 partner, :81 stretch uniform, :93 accept): one Philox block
struct mhx_emcee_draws { mhx_u32 partner; float u, logu; };
MHX_DEV mhx_emcee_draws mhx_emcee_draw(const mhx_philox_key& ks, mhx_u32 walker, mhx_u32 ens, mhx_u32 sweep)
{
    const mhx_u32x4 w = mhx_philox(ks, walker, ens, sweep, MHX_STREAM_EMCEE << 28);
    mhx_emcee_draws o;
    o.partner = w.x;
    o.u = mhx_u01_half(w.y);
    o.logu = mhx_log_pos(mhx_u01_open(w.z));
    return o;
}
#endif

// ---------------------------------------------------------------------------------------------
// One step of the xor-butterfly of the reduction shapes (spec 3.7): q + (the q of lane ^ DIST), DIST a power of two below 64.
// `__shfl_xor` of a double compiles to two ds_bpermute_b32 and a wait -- an LDS round trip (~120 cycles) per step, six of them in a
// row for a wave-per-chain sum, on the critical path of every step.  Round 4: the same sum from the VALU's own cross-lane paths --
// DPP moves inside a row of 16 lanes (quad_perm for 1 and 2, row_shl / row_shr under bank masks for 4, row_ror:8 for 8) and the
// gfx950 row / half swaps for 16 and 32 (v_permlane16_swap, v_permlane32_swap: with both operands a copy of q they leave the even /
// lower part in one register and the odd / upper part in the other, and their sum IS q + partner for every lane -- in the upper
// lanes as partner + q, the same bits).  Bit-identical to the shuffle form.  All 64 lanes must be active.
#ifndef MHX_BUTTERFLY_DPP
#define MHX_BUTTERFLY_DPP 1
#endif
template <int CTRL, int BANKS>
MHX_DEV mhx_u32 mhx_dpp_mov(const mhx_u32 old, const mhx_u32 v)
{
    return (mhx_u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, BANKS, false);
}
template <int DIST>
MHX_DEV mhx_u32 mhx_lane_xor_u32(const mhx_u32 v)
{
    static_assert(DIST == 1 || DIST == 2 || DIST == 4 || DIST == 8, "DPP distances");
    if constexpr (DIST == 1) return mhx_dpp_mov<0xB1, 0xf>(v, v);            // quad_perm:[1,0,3,2]
    else if constexpr (DIST == 2) return mhx_dpp_mov<0x4E, 0xf>(v, v);       // quad_perm:[2,3,0,1]
    else if constexpr (DIST == 4) {
        const mhx_u32 t = mhx_dpp_mov<0x104, 0x5>(v, v);                     // row_shl:4 -> banks 0, 2 read lane + 4
        return mhx_dpp_mov<0x114, 0xa>(t, v);                                // row_shr:4 -> banks 1, 3 read lane - 4
    } else return mhx_dpp_mov<0x128, 0xf>(v, v);                             // row_ror:8 == lane ^ 8 within a row of 16
}
template <int DIST>
MHX_DEV mhx_real mhx_butterfly_add(const mhx_real q)
{
#if MHX_BUTTERFLY_DPP
    if constexpr (DIST == 16 || DIST == 32) {
#if MHX_REAL64
        const mhx_u64 b = __builtin_bit_cast(mhx_u64, q);
        const mhx_u32 lo = (mhx_u32)b, hi = (mhx_u32)(b >> 32);
        mhx_u32 alo, blo, ahi, bhi;
        if constexpr (DIST == 16) {
            const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); alo = r0[0]; blo = r0[1];
            const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false); ahi = r1[0]; bhi = r1[1];
        } else {
            const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); alo = r0[0]; blo = r0[1];
            const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false); ahi = r1[0]; bhi = r1[1];
        }
        const double a = __builtin_bit_cast(double, (mhx_u64)alo | ((mhx_u64)ahi << 32));
        const double c = __builtin_bit_cast(double, (mhx_u64)blo | ((mhx_u64)bhi << 32));
        return a + c;
#else
        const mhx_u32 w = __builtin_bit_cast(mhx_u32, q);
        mhx_u32 aw, bw;
        if constexpr (DIST == 16) { const auto r = __builtin_amdgcn_permlane16_swap(w, w, false, false); aw = r[0]; bw = r[1]; }
        else { const auto r = __builtin_amdgcn_permlane32_swap(w, w, false, false); aw = r[0]; bw = r[1]; }
        return __builtin_bit_cast(float, aw) + __builtin_bit_cast(float, bw);
#endif
    } else {
#if MHX_REAL64
        const mhx_u64 b = __builtin_bit_cast(mhx_u64, q);
        const mhx_u32 lo = mhx_lane_xor_u32<DIST>((mhx_u32)b), hi = mhx_lane_xor_u32<DIST>((mhx_u32)(b >> 32));
        return q + __builtin_bit_cast(double, (mhx_u64)lo | ((mhx_u64)hi << 32));
#else
        return q + __builtin_bit_cast(float, mhx_lane_xor_u32<DIST>(__builtin_bit_cast(mhx_u32, q)));
#endif
    }
#else
    return q + __shfl_xor(q, DIST, 64);
#endif
}
// The value of lane ^ DIST for DIST = 1, 2, 4, 8 (DPP inside a row of 16 lanes); all lanes of the group must be active.
template <int DIST>
MHX_DEV mhx_real mhx_lane_xor(const mhx_real v)
{
#if MHX_REAL64
    const mhx_u64 b = __builtin_bit_cast(mhx_u64, v);
    const mhx_u32 lo = mhx_lane_xor_u32<DIST>((mhx_u32)b), hi = mhx_lane_xor_u32<DIST>((mhx_u32)(b >> 32));
    return __builtin_bit_cast(double, (mhx_u64)lo | ((mhx_u64)hi << 32));
#else
    return __builtin_bit_cast(float, mhx_lane_xor_u32<DIST>(__builtin_bit_cast(mhx_u32, v)));
#endif
}
// The value of lane ^ 1 (quad_perm:[1,0,3,2]); both lanes of a pair must be active.
MHX_DEV mhx_real mhx_lane_swap1(const mhx_real v)
{
#if MHX_REAL64
    const mhx_u64 b = __builtin_bit_cast(mhx_u64, v);
    const mhx_u32 lo = mhx_lane_xor_u32<1>((mhx_u32)b), hi = mhx_lane_xor_u32<1>((mhx_u32)(b >> 32));
    return __builtin_bit_cast(double, (mhx_u64)lo | ((mhx_u64)hi << 32));
#else
    return __builtin_bit_cast(float, mhx_lane_xor_u32<1>(__builtin_bit_cast(mhx_u32, v)));
#endif
}
// Two rows x two neighbouring columns of a [rows][ld] slab as ONE store per lane (round 4, tools/ubench/store_width.hip: at one
// wave per SIMD the record of a wave-step costs by the store INSTRUCTION, not by the byte -- 52 x 8 bytes per lane add 34 % to a
// stand-in wave-step, 26 x 16 bytes of the same rows 13 %, the transposes included 14 %).  Lanes 2m and 2m + 1 hold columns
// 2m and 2m + 1; both hold the values (a, b) of rows (r, r + 1).  After a 2 x 2 transpose between the two lanes the even lane
// writes row r of both columns and the odd lane row r + 1: the same bytes at the same addresses, half the stores.  Returns the
// pair this lane writes; its address is (row r + odd) of column 2m.
MHX_DEV void mhx_pair_rows(const mhx_real a, const mhx_real b, mhx_real& v0, mhx_real& v1)
{
    // v0 = odd ? (b of lane ^ 1) : a;  v1 = odd ? b : (a of lane ^ 1) -- the select and the lane swap in ONE instruction per
    // 32-bit word (v_cndmask_b32 takes its lane mask from vcc and a DPP control on src0; hipcc emits move + DPP move + select)
    const mhx_u64 even = 0x5555555555555555ull;
#if MHX_REAL64
    const mhx_u64 ab = __builtin_bit_cast(mhx_u64, a), bb = __builtin_bit_cast(mhx_u64, b);
    const mhx_u32 alo = (mhx_u32)ab, ahi = (mhx_u32)(ab >> 32), blo = (mhx_u32)bb, bhi = (mhx_u32)(bb >> 32);
    mhx_u32 p0, p1, q0, q1;
    // (s_nop: a DPP operand written by the VALU instruction just before needs two wait states, and nothing checks inside an asm)
    asm("s_mov_b64 vcc, %8\n\ts_nop 0\n\t"
        "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(p0), "=&v"(p1), "=&v"(q0), "=&v"(q1) : "v"(alo), "v"(ahi), "v"(blo), "v"(bhi), "s"(even) : "vcc", "scc");
    v0 = __builtin_bit_cast(double, (mhx_u64)p0 | ((mhx_u64)p1 << 32));
    v1 = __builtin_bit_cast(double, (mhx_u64)q0 | ((mhx_u64)q1 << 32));
#else
    const mhx_u32 aw = __builtin_bit_cast(mhx_u32, a), bw = __builtin_bit_cast(mhx_u32, b);
    mhx_u32 p, q;
    asm("s_mov_b64 vcc, %4\n\ts_nop 0\n\t"
        "v_cndmask_b32_dpp %0, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %1, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(p), "=&v"(q) : "v"(aw), "v"(bw), "s"(even) : "vcc", "scc");
    v0 = __builtin_bit_cast(float, p);
    v1 = __builtin_bit_cast(float, q);
#endif
}
MHX_DEV void mhx_srd_store2(mhx_srd srd, mhx_u32 lane_byte_off, mhx_u32 row_byte_off, const mhx_real v0, const mhx_real v1)
{
#if MHX_REAL64
    typedef mhx_u32 mhx_u32v4 __attribute__((ext_vector_type(4)));
    const mhx_u64 b0 = __builtin_bit_cast(mhx_u64, v0), b1 = __builtin_bit_cast(mhx_u64, v1);
    mhx_u32v4 w;
    w.x = (mhx_u32)b0; w.y = (mhx_u32)(b0 >> 32); w.z = (mhx_u32)b1; w.w = (mhx_u32)(b1 >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(w, srd, (int)lane_byte_off, (int)row_byte_off, MHX_REC_STORE_AUX);
#else
    typedef mhx_u32 mhx_u32w2 __attribute__((ext_vector_type(2)));
    mhx_u32w2 w;
    w.x = __builtin_bit_cast(mhx_u32, v0); w.y = __builtin_bit_cast(mhx_u32, v1);
    __builtin_amdgcn_raw_buffer_store_b64(w, srd, (int)lane_byte_off, (int)row_byte_off, MHX_REC_STORE_AUX);
#endif
}

// the whole butterfly of reduction shape L for 64 / L chains per wave (lane = l * CPW + chain): offsets CPW, 2 CPW, ... below 64
template <int L, int OFF = 1>
MHX_DEV mhx_real mhx_butterfly(mhx_real q)
{
    if constexpr (OFF < L) {
        q = mhx_butterfly_add<OFF * (64 / L)>(q);
        return mhx_butterfly<L, 2 * OFF>(q);
    } else {
        return q;
    }
}

MHX_NS_END
