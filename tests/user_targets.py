"""User log-densities in HIP-source form (the `DensityModel(f)` of the reference, lowered by hiprtc)
plus a helper that compiles the SAME source for the host so the oracle can evaluate it.  The sources are written
against `mhx_real` / `MHX_R(literal)`, so one text serves the fp32 and the fp64 engine.

The models are the reference's own test models:
  NIG_UNTRANSFORMED / NIG_TRANSFORMED  -- test/emcee.jl:5-14 and :46-56 (known answers E[s]=49/24, E[m]=7/6)
"""
import ctypes as C
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# logpdf(InverseGamma(2,3), s) = 2 log 3 - lgamma(2) - 3 log s - 3/s
# logpdf(Normal(mu, sqrt(s)), y) = -1/2 (log 2pi + log s) - (y-mu)^2 / (2 s)
NIG_UNTRANSFORMED = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    const mhx_real s = x[0], m = x[1];
    if (!(s > MHX_R(0.0))) return -MHX_INF;                    // s > 0 || return -Inf
    const mhx_real ls = mhx_log(s);
    const mhx_real inv = MHX_R(1.0) / s;
    mhx_real lp = MHX_R(2.1972245773362196) - MHX_R(3.0) * ls - MHX_R(3.0) * inv;   // InverseGamma(2,3) at s   (2 log 3)
    const mhx_real c = -MHX_R(0.5) * (MHX_R(1.8378770664093453) + ls);        // -1/2 (log 2pi + log s)
    lp += c - MHX_R(0.5) * (m * m) * inv;                       // Normal(0, sqrt s) at m
    const mhx_real r1 = MHX_R(1.5) - m, r2 = MHX_R(2.0) - m;
    lp += c - MHX_R(0.5) * (r1 * r1) * inv;                     // Normal(m, sqrt s) at 1.5
    lp += c - MHX_R(0.5) * (r2 * r2) * inv;                     //                   at 2.0
    return lp;
}
"""

NIG_TRANSFORMED = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    const mhx_real ls = x[0], m = x[1];
    const mhx_real s = mhx_exp(ls);
    const mhx_real inv = MHX_R(1.0) / s;
    mhx_real lp = MHX_R(2.1972245773362196) - MHX_R(3.0) * ls - MHX_R(3.0) * inv;
    const mhx_real c = -MHX_R(0.5) * (MHX_R(1.8378770664093453) + ls);
    lp += c - MHX_R(0.5) * (m * m) * inv;
    const mhx_real r1 = MHX_R(1.5) - m, r2 = MHX_R(2.0) - m;
    lp += c - MHX_R(0.5) * (r1 * r1) * inv;
    lp += c - MHX_R(0.5) * (r2 * r2) * inv;
    return lp + ls;                                       // + log-Jacobian
}
"""

# a data-dependent one: independent Gaussians with per-dimension mean/std passed as data
SHIFTED_GAUSS = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    mhx_real q = MHX_R(0.0);
    for (int k = 0; k < d; ++k) {
        const mhx_real z = (x[k] - data[k]) / data[d + k];
        q = mhx_fma(z, z, q);
    }
    return -MHX_R(0.5) * q;
}
"""

# independent shifted Gaussians with their gradient (data = [shift[d], 1/scale[d]]): a cheap MALA model for any dimension
SHIFTED_GAUSS_WITH_GRADIENT = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    mhx_real q = MHX_R(0.0);
    for (int k = 0; k < d; ++k) { const mhx_real z = (x[k] - data[k]) * data[d + k]; q = mhx_fma(z, z, q); }
    return -MHX_R(0.5) * q;
}
MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata)
{
    mhx_real q = MHX_R(0.0);
    for (int k = 0; k < d; ++k) {
        const mhx_real z = (x[k] - data[k]) * data[d + k];
        q = mhx_fma(z, z, q);
        g.set(k, -z * data[d + k]);
    }
    return -MHX_R(0.5) * q;
}
"""

# test/runtests.jl:334-365 (issue #95): TheNormalLogDensity(A): lp = -x'Ax/2, gradient = -Ax; data = A row-major
QUADRATIC_WITH_GRADIENT = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    mhx_real q = MHX_R(0.0);
    for (int i = 0; i < d; ++i) {
        mhx_real r = MHX_R(0.0);
        for (int j = 0; j < d; ++j) r = mhx_fma(data[i * d + j], x[j], r);
        q = mhx_fma(x[i], r, q);
    }
    return -MHX_R(0.5) * q;
}
MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata)
{
    mhx_real q = MHX_R(0.0);
    for (int i = 0; i < d; ++i) {
        mhx_real r = MHX_R(0.0);
        for (int j = 0; j < d; ++j) r = mhx_fma(data[i * d + j], x[j], r);
        q = mhx_fma(x[i], r, q);
        g.set(i, -r);
    }
    return -MHX_R(0.5) * q;
}
"""

_PRELUDE = r"""
#include <math.h>
#include <string.h>
#if MHX_REAL64
typedef double mhx_real;
#define MHX_R(x) x
static inline double mhx_fma(double a, double b, double c) { return fma(a, b, c); }
static inline double mhx_sqrt(double x) { return sqrt(x); }
static inline double mhx_abs(double x) { return fabs(x); }
#else
typedef float mhx_real;
#define MHX_R(x) x##f
static inline float mhx_fma(float a, float b, float c) { return fmaf(a, b, c); }
static inline float mhx_sqrt(float x) { return sqrtf(x); }
static inline float mhx_abs(float x) { return fabsf(x); }
#endif
extern "C" { mhx_real orc_log(mhx_real); mhx_real orc_exp(mhx_real); }
static inline mhx_real mhx_log(mhx_real x) { return orc_log(x); }
static inline mhx_real mhx_exp(mhx_real x) { return orc_exp(x); }
#define MHX_INF INFINITY
#define MHX_NAN NAN
#define MHX_LOGDENSITY(x, d, data, ndata) \
    template <class MHX_X> static inline mhx_real mhx_user_logdensity(const MHX_X& x, const int d, const mhx_real* data, const int ndata)
#define MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata) \
    template <class MHX_X, class MHX_G> static inline mhx_real mhx_user_logdensity_and_gradient(const MHX_X& x, const MHX_G& g, const int d, const mhx_real* data, const int ndata)
struct host_grad_out { mhx_real* p; void set(int k, mhx_real v) const { p[k] = v; } mhx_real operator[](int k) const { return p[k]; } };
"""

_EPILOGUE_GRAD = r"""
extern "C" mhx_real user_logdensity_and_gradient(const mhx_real* x, mhx_real* g, int d, const void* data)
{
    const user_data* D = (const user_data*)data;
    host_grad_out go = { g };
    return mhx_user_logdensity_and_gradient(x, go, d, D ? D->p : (const mhx_real*)0, D ? D->n : 0);
}
"""

_EPILOGUE = r"""
struct user_data { const mhx_real* p; int n; };
extern "C" mhx_real user_logdensity(const mhx_real* x, int d, const void* data)
{
    const user_data* D = (const user_data*)data;
    return mhx_user_logdensity(x, d, D ? D->p : (const mhx_real*)0, D ? D->n : 0);
}
"""


class _UserData(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_int)]


def host_target(oracle, source, dim, data=None, cache_dir="/tmp/mhx_user_targets"):
    """Compile `source` with g++ (same -ffp-contract=off discipline) and wrap it as an oracle Target."""
    import numpy as np
    oracle.build()
    os.makedirs(cache_dir, exist_ok=True)
    has_grad = "MHX_LOGDENSITY_AND_GRADIENT" in source
    epilogue = _EPILOGUE + (_EPILOGUE_GRAD if has_grad else "")
    f64 = oracle.get_dtype() == "f64"
    tag = hashlib.sha1((_PRELUDE + source + epilogue).encode()).hexdigest()[:16] + ("_f64" if f64 else "_f32")
    so = os.path.join(cache_dir, "user_%s.so" % tag)
    if not os.path.exists(so):
        cpp = os.path.join(cache_dir, "user_%s.cpp" % tag)
        open(cpp, "w").write(_PRELUDE + source + epilogue)
        odir = os.path.join(ROOT, "oracle")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                               "-mfma", "-mavx2", "-DMHX_REAL64=%d" % (1 if f64 else 0), "-o", so, cpp, "-L" + odir,
                               "-lmhx_oracle64" if f64 else "-lmhx_oracle", "-Wl,-rpath," + odir])
    lib = C.CDLL(so)
    addr = C.cast(lib.user_logdensity, C.c_void_p).value
    ud = None
    arr = None
    if data is not None:
        arr = np.ascontiguousarray(data, dtype=oracle.real()).ravel()
        ud = _UserData(arr.ctypes.data_as(C.c_void_p), arr.size)
    t = oracle.Target(oracle.TARGET_CALLBACK, dim, fn=addr, fn_data=ud)
    t._keep = (lib, arr, ud)
    t.grad_addr = C.cast(lib.user_logdensity_and_gradient, C.c_void_p).value if has_grad else None
    return t
