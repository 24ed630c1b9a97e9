"""User log-densities in HIP-source form (the `DensityModel(f)` of the reference, lowered by hiprtc)
plus a helper that compiles the SAME source for the host so the oracle can evaluate it.

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
    const float s = x[0], m = x[1];
    if (!(s > 0.0f)) return -MHX_INF;                    // s > 0 || return -Inf
    const float ls = mhx_log(s);
    const float inv = 1.0f / s;
    float lp = 0x1.193ea8p+1f - 3.0f * ls - 3.0f * inv;   // InverseGamma(2,3) at s   (2 log 3)
    const float c = -0.5f * (0x1.d67f1cp+0f + ls);        // -1/2 (log 2pi + log s)
    lp += c - 0.5f * (m * m) * inv;                       // Normal(0, sqrt s) at m
    const float r1 = 1.5f - m, r2 = 2.0f - m;
    lp += c - 0.5f * (r1 * r1) * inv;                     // Normal(m, sqrt s) at 1.5
    lp += c - 0.5f * (r2 * r2) * inv;                     //                   at 2.0
    return lp;
}
"""

NIG_TRANSFORMED = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    const float ls = x[0], m = x[1];
    const float s = mhx_exp(ls);
    const float inv = 1.0f / s;
    float lp = 0x1.193ea8p+1f - 3.0f * ls - 3.0f * inv;
    const float c = -0.5f * (0x1.d67f1cp+0f + ls);
    lp += c - 0.5f * (m * m) * inv;
    const float r1 = 1.5f - m, r2 = 2.0f - m;
    lp += c - 0.5f * (r1 * r1) * inv;
    lp += c - 0.5f * (r2 * r2) * inv;
    return lp + ls;                                       // + log-Jacobian
}
"""

# a data-dependent one: independent Gaussians with per-dimension mean/std passed as data
SHIFTED_GAUSS = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    float q = 0.0f;
    for (int k = 0; k < d; ++k) {
        const float z = (x[k] - data[k]) / data[d + k];
        q = mhx_fma(z, z, q);
    }
    return -0.5f * q;
}
"""

# test/runtests.jl:334-365 (issue #95): TheNormalLogDensity(A): lp = -x'Ax/2, gradient = -Ax; data = A row-major
QUADRATIC_WITH_GRADIENT = r"""
MHX_LOGDENSITY(x, d, data, ndata)
{
    float q = 0.0f;
    for (int i = 0; i < d; ++i) {
        float r = 0.0f;
        for (int j = 0; j < d; ++j) r = mhx_fma(data[i * d + j], x[j], r);
        q = mhx_fma(x[i], r, q);
    }
    return -0.5f * q;
}
MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata)
{
    float q = 0.0f;
    for (int i = 0; i < d; ++i) {
        float r = 0.0f;
        for (int j = 0; j < d; ++j) r = mhx_fma(data[i * d + j], x[j], r);
        q = mhx_fma(x[i], r, q);
        g.set(i, -r);
    }
    return -0.5f * q;
}
"""

_PRELUDE = r"""
#include <math.h>
#include <string.h>
extern "C" { float orc_logf(float); float orc_expf(float); }
static inline float mhx_fma(float a, float b, float c) { return fmaf(a, b, c); }
static inline float mhx_log(float x) { return orc_logf(x); }
static inline float mhx_exp(float x) { return orc_expf(x); }
static inline float mhx_sqrt(float x) { return sqrtf(x); }
#define MHX_INF INFINITY
#define MHX_NAN NAN
#define MHX_LOGDENSITY(x, d, data, ndata) \
    template <class MHX_X> static inline float mhx_user_logdensity(const MHX_X& x, const int d, const float* data, const int ndata)
#define MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata) \
    template <class MHX_X, class MHX_G> static inline float mhx_user_logdensity_and_gradient(const MHX_X& x, const MHX_G& g, const int d, const float* data, const int ndata)
struct host_grad_out { float* p; void set(int k, float v) const { p[k] = v; } float operator[](int k) const { return p[k]; } };
"""

_EPILOGUE_GRAD = r"""
extern "C" float user_logdensity_and_gradient(const float* x, float* g, int d, const void* data)
{
    const user_data* D = (const user_data*)data;
    host_grad_out go = { g };
    return mhx_user_logdensity_and_gradient(x, go, d, D ? D->p : (const float*)0, D ? D->n : 0);
}
"""

_EPILOGUE = r"""
struct user_data { const float* p; int n; };
extern "C" float user_logdensity(const float* x, int d, const void* data)
{
    const user_data* D = (const user_data*)data;
    return mhx_user_logdensity(x, d, D ? D->p : (const float*)0, D ? D->n : 0);
}
"""


class _UserData(C.Structure):
    _fields_ = [("p", C.POINTER(C.c_float)), ("n", C.c_int)]


def host_target(oracle, source, dim, data=None, cache_dir="/tmp/mhx_user_targets"):
    """Compile `source` with g++ (same -ffp-contract=off discipline) and wrap it as an oracle Target."""
    import numpy as np
    oracle.build()
    os.makedirs(cache_dir, exist_ok=True)
    has_grad = "MHX_LOGDENSITY_AND_GRADIENT" in source
    epilogue = _EPILOGUE + (_EPILOGUE_GRAD if has_grad else "")
    tag = hashlib.sha1((_PRELUDE + source + epilogue).encode()).hexdigest()[:16]
    so = os.path.join(cache_dir, "user_%s.so" % tag)
    if not os.path.exists(so):
        cpp = os.path.join(cache_dir, "user_%s.cpp" % tag)
        open(cpp, "w").write(_PRELUDE + source + epilogue)
        odir = os.path.join(ROOT, "oracle")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                               "-mfma", "-mavx2", "-o", so, cpp, "-L" + odir, "-lmhx_oracle", "-Wl,-rpath," + odir])
    lib = C.CDLL(so)
    addr = C.cast(lib.user_logdensity, C.c_void_p).value
    ud = None
    arr = None
    if data is not None:
        arr = np.ascontiguousarray(data, dtype=np.float32).ravel()
        ud = _UserData(arr.ctypes.data_as(C.POINTER(C.c_float)), arr.size)
    t = oracle.Target(oracle.TARGET_CALLBACK, dim, fn=addr, fn_data=ud)
    t._keep = (lib, arr, ud)
    t.grad_addr = C.cast(lib.user_logdensity_and_gradient, C.c_void_p).value if has_grad else None
    return t
