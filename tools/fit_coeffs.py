#!/usr/bin/env python3
"""Derive the fp32 polynomial coefficients of the MHX arithmetic spec.

Own derivation (Chebyshev-node interpolation in float64, coefficients rounded to fp32,
then the *fp32 fmaf evaluation* is measured exhaustively / densely against float64 libm).
The output of this script is pasted into csrc/mhx_math.h and oracle/mhx_oracle.c.
"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P

f32 = np.float32

def cheb_fit(fun, a, b, deg, npts=None):
    n = deg + 1 if npts is None else npts
    k = np.arange(n)
    t = np.cos(np.pi * (2 * k + 1) / (2 * n))          # Chebyshev nodes on [-1,1]
    x = 0.5 * (b - a) * t + 0.5 * (b + a)
    c = C.chebfit(t, fun(x), deg)
    # convert to monomials in x
    p_t = C.cheb2poly(c)                                # poly in t
    # t = (2x - (a+b))/(b-a)
    s = 2.0 / (b - a); o = -(a + b) / (b - a)
    px = np.zeros(1)
    tpow = np.ones(1)
    lin = np.array([o, s])
    for ck in p_t:
        px = P.polyadd(px, ck * tpow)
        tpow = P.polymul(tpow, lin)
    return px

def hexf(v):
    return float(f32(v)).hex()

def show(name, coeffs):
    print(f"// {name}")
    for i, c in enumerate(coeffs):
        print(f"  c{i} = {float(f32(c))!r:>22}f  /* {hexf(c)} */")

# ---- log1p(f) = f + f^2 * Q(f),  f in [-1/3, 1/3]
def Q(f):
    f = np.asarray(f, dtype=np.float64)
    out = np.empty_like(f)
    small = np.abs(f) < 1e-4
    fs = f[small]
    out[small] = -0.5 + fs / 3 - fs * fs / 4 + fs ** 3 / 5
    fl = f[~small]
    out[~small] = (np.log1p(fl) - fl) / (fl * fl)
    return out

q = cheb_fit(Q, -1.0 / 3, 1.0 / 3, 8, 64)
show("LOG Q(f), deg 8: log1p(f) = f + f*f*Q(f)", q)

# ---- sin(2*pi*r)/r = S(r^2), cos(2*pi*r) = Cc(r^2), r in [-1/8,1/8]
def Sfun(u):
    r = np.sqrt(u)
    out = np.where(r < 1e-9, 2 * np.pi, np.sin(2 * np.pi * r) / np.where(r == 0, 1, r))
    return out
def Cfun(u):
    r = np.sqrt(u)
    return np.cos(2 * np.pi * r)
s = cheb_fit(Sfun, 0.0, 1.0 / 64, 4, 32)
c = cheb_fit(Cfun, 0.0, 1.0 / 64, 4, 32)
show("SIN: sin(2 pi r) = r * S(r*r), deg 4 in r^2", s)
show("COS: cos(2 pi r) = C(r*r), deg 4 in r^2", c)

# ---- exp(r) for r in [-ln2/2, ln2/2]: exp(r) = 1 + r + r^2 * E(r)
def Efun(r):
    r = np.asarray(r, dtype=np.float64)
    out = np.empty_like(r)
    small = np.abs(r) < 1e-4
    rs = r[small]
    out[small] = 0.5 + rs / 6 + rs * rs / 24
    rl = r[~small]
    out[~small] = (np.expm1(rl) - rl) / (rl * rl)
    return out
e = cheb_fit(Efun, -0.5 * np.log(2) * 1.0001, 0.5 * np.log(2) * 1.0001, 5, 32)
show("EXP: exp(r) = 1 + r + r*r*E(r), deg 5", e)
