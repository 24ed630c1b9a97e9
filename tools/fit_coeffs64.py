#!/usr/bin/env python3
"""Derive the fp64 polynomial coefficients of the MHX arithmetic spec (Float64 engine).

Own derivation: Chebyshev-series approximation in 60-digit arithmetic (mpmath.chebyfit), coefficients rounded to
binary64.  The fma evaluation of the rounded polynomials is measured against mpmath in tests/test_oracle_primitives.py
through the oracle (log / exp / sincos stay below 1 ulp).  The output is pasted into csrc/mhx_device_math.h and
oracle/mhx_oracle.c.
"""
import mpmath as mp

mp.mp.dps = 60


def show(name, coeffs):
    print("// %s" % name)
    # chebyfit returns the highest power first
    for i, c in enumerate(coeffs[::-1]):
        print("  c%-2d = %s   /* %.20e */" % (i, float(c).hex(), float(c)))


# ---- log: m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), z = s^2
#      log(1 + f) = log((1 + s) / (1 - s)) = 2 s + s z P(z),  P(z) = 2/3 + 2 z / 5 + ...
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
zmax = smax * smax * mp.mpf("1.0002")


def P(z):
    if z < mp.mpf("1e-30"):
        return mp.mpf(2) / 3
    s = mp.sqrt(z)
    return (mp.log((1 + s) / (1 - s)) / s - 2) / z


show("LOG P(z), deg 6: log((1+s)/(1-s)) = 2 s + s z P(z), z = s^2 <= %.6f" % float(zmax), mp.chebyfit(P, [0, zmax], 7))


# ---- exp: exp(r) = 1 + r + r^2 E(r), |r| <= ln2 / 2
def E(r):
    if abs(r) < mp.mpf("1e-20"):
        return mp.mpf(1) / 2 + r / 6
    return (mp.expm1(r) - r) / (r * r)


h = mp.log(2) / 2 * mp.mpf("1.0001")
show("EXP E(r), deg 11: exp(r) = 1 + r + r^2 E(r)", mp.chebyfit(E, [-h, h], 12))


# ---- sin(2 pi r) = 2 pi r + r u S1(u),  cos(2 pi r) = 1 - 2 pi^2 u + u^2 C2(u),  u = r^2, |r| <= 1/8.
#      The leading constants are split (hi + lo) so that their rounding does not reach the result.
c1 = -2 * mp.pi ** 2


def S1(u):
    if u < mp.mpf("1e-30"):
        return -(2 * mp.pi) ** 3 / 6
    r = mp.sqrt(u)
    return (mp.sin(2 * mp.pi * r) / r - 2 * mp.pi) / u


def C2(u):
    if u < mp.mpf("1e-20"):
        return (2 * mp.pi) ** 4 / 24
    return (mp.cos(2 * mp.pi * mp.sqrt(u)) - 1 - c1 * u) / (u * u)


show("SIN S1(u), deg 6", mp.chebyfit(S1, [0, mp.mpf(1) / 64], 7))
show("COS C2(u), deg 6", mp.chebyfit(C2, [0, mp.mpf(1) / 64], 7))
for name, v in (("2PI", 2 * mp.pi), ("-2PI^2", c1)):
    hi_ = float(v)
    print("  %s_HI = %s   %s_LO = %s" % (name, hi_.hex(), name, float(v - mp.mpf(hi_)).hex()))

print("// ln2 split: hi = 32 significant bits (n * hi exact for |n| < 2^20), lo = ln2 - hi")
ln2 = mp.log(2)
hi = mp.floor(ln2 * 2 ** 32) / 2 ** 32
print("  LN2_HI = %s\n  LN2_LO = %s\n  LOG2E = %s" % (float(hi).hex(), float(ln2 - hi).hex(), float(1 / ln2).hex()))
print("  HALF_LOG_2PI = %s" % float(mp.log(2 * mp.pi) / 2).hex())
print("  1/18 = %s" % float(mp.mpf(1) / 18).hex())
print("  exp overflow above %s ; zero below %s" % (float(mp.log(mp.mpf(2) ** 1024)).hex(), float(mp.log(mp.mpf(2) ** -1075)).hex()))
