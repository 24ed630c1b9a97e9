"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE -- never imported by the product path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmhx_oracle.so")

STREAM_PROPOSAL, STREAM_ACCEPT, STREAM_INIT, STREAM_EMCEE = 0, 1, 2, 3
TARGET_ISO_GAUSS, TARGET_CORR_GAUSS, TARGET_IID_NORMAL, TARGET_BANANA, TARGET_FUNNEL = 0, 1, 2, 3, 4
TARGET_CALLBACK = 100
PROP_ISO, PROP_DIAG, PROP_DENSE = 0, 1, 2

LOGDENSITY_FN = C.CFUNCTYPE(C.c_float, C.POINTER(C.c_float), C.c_int, C.c_void_p)


class _Target(C.Structure):
    _fields_ = [("kind", C.c_int), ("dim", C.c_int), ("params", C.POINTER(C.c_float)),
                ("nparams", C.c_int), ("fn", LOGDENSITY_FN), ("fn_data", C.c_void_p), ("reduce_lanes", C.c_int)]


class _Proposal(C.Structure):
    _fields_ = [("kind", C.c_int), ("scale", C.c_float), ("vec", C.POINTER(C.c_float)), ("mean", C.POINTER(C.c_float)),
                ("is_static", C.c_int)]


class _Schedule(C.Structure):
    _fields_ = [("n_samples", C.c_int), ("discard_initial", C.c_int), ("thinning", C.c_int),
                ("num_warmup", C.c_int)]


class _RamCfg(C.Structure):
    _fields_ = [("alpha", C.c_float), ("gamma", C.c_float), ("eig_lo", C.c_float), ("eig_hi", C.c_float)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mhx_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_logf.restype = C.c_float
        L.orc_logf.argtypes = [C.c_float]
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_u01_open.restype = C.c_float
        L.orc_u01_open.argtypes = [C.c_uint32]
        L.orc_u01_half.restype = C.c_float
        L.orc_u01_half.argtypes = [C.c_uint32]
        L.orc_accept_logu.restype = C.c_float
        L.orc_accept_logu.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_target_eval.restype = C.c_float
        L.orc_target_eval.argtypes = [C.POINTER(_Target), C.POINTER(C.c_float)]
        L.orc_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_float)]
        L.orc_chol_rank1.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def logf(x):
    L = lib()
    return np.array([L.orc_logf(float(v)) for v in np.atleast_1d(x)], dtype=np.float32)


def expf(x):
    L = lib()
    return np.array([L.orc_expf(float(v)) for v in np.atleast_1d(x)], dtype=np.float32)


def sincos2pi_u32(k):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos2pi_u32(C.c_uint32(int(k)), C.byref(s), C.byref(c))
    return s.value, c.value


def normals(seed, chain, step, stream, d):
    out = np.empty(d, dtype=np.float32)
    lib().orc_normals(seed, chain, step, stream, d, _fp(out))
    return out


def accept_logu(seed, chain, step):
    return lib().orc_accept_logu(seed, chain, step)


class Target:
    """kind + params, or a Python callable f(x: np.ndarray) -> float (DensityModel(f))."""

    def __init__(self, kind, dim, params=None, fn=None, fn_data=None, reduce_lanes=0):
        self.kind, self.dim = kind, dim
        self._fn_data = fn_data             # keep-alive for a ctypes object passed as void*
        self.params = None if params is None else np.ascontiguousarray(params, dtype=np.float32)
        self._cb = None
        self._fn_addr = None
        if fn is not None:
            if isinstance(fn, int):             # raw C function pointer (e.g. from a gcc-built user source)
                self._fn_addr = fn
            else:
                def _tramp(xp, d, _data, fn=fn):
                    return float(fn(np.ctypeslib.as_array(xp, shape=(d,)).copy()))
                self._cb = LOGDENSITY_FN(_tramp)
        self.c = _Target()
        self.c.kind = kind
        self.c.dim = dim
        self.c.params = _fp(self.params)
        self.c.nparams = 0 if self.params is None else int(self.params.size)
        if self._cb is not None:
            self.c.fn = self._cb
        elif self._fn_addr is not None:
            self.c.fn = C.cast(self._fn_addr, LOGDENSITY_FN)
        self.c.fn_data = None if fn_data is None else C.cast(C.pointer(fn_data), C.c_void_p)
        self.c.reduce_lanes = int(reduce_lanes)

    def with_lanes(self, L):
        """the same target with the L-lane reduction shape of the cooperative kernels"""
        self.c.reduce_lanes = int(L)
        return self

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        return lib().orc_target_eval(C.byref(self.c), _fp(x))


def iso_gauss(d, reduce_lanes=0):
    return Target(TARGET_ISO_GAUSS, d, reduce_lanes=reduce_lanes)


def corr_gauss_from_cov(Sigma, reduce_lanes=0):
    """params = inv(chol(Sigma)) packed lower row-major, computed in float64 then rounded."""
    Sigma = np.asarray(Sigma, dtype=np.float64)
    A = np.linalg.inv(np.linalg.cholesky(Sigma))
    return Target(TARGET_CORR_GAUSS, Sigma.shape[0], pack_lower(A), reduce_lanes=reduce_lanes)


def pack_lower(M):
    M = np.asarray(M)
    d = M.shape[0]
    return np.concatenate([M[i, :i + 1] for i in range(d)]).astype(np.float32)


def unpack_lower(p, d):
    M = np.zeros((d, d), dtype=np.float32)
    o = 0
    for i in range(d):
        M[i, :i + 1] = p[o:o + i + 1]
        o += i + 1
    return M


class Proposal:
    def __init__(self, kind, scale=1.0, vec=None, mean=None, static=False):
        self.vec = None if vec is None else np.ascontiguousarray(vec, dtype=np.float32)
        self.mean = None if mean is None else np.ascontiguousarray(mean, dtype=np.float32)
        self.c = _Proposal(kind, float(scale), _fp(self.vec), _fp(self.mean), 1 if static else 0)


def schedule(n_samples, discard_initial=0, thinning=1, num_warmup=0):
    return _Schedule(n_samples, discard_initial, thinning, num_warmup)


def schedule_counts(s):
    a, b = C.c_int64(), C.c_int64()
    lib().orc_schedule_counts(C.byref(s), C.byref(a), C.byref(b))
    return a.value, b.value


def rwmh(target, prop, sched, seed, first_chain, nchains, init=None, save=True):
    d, N, Cn = target.dim, sched.n_samples, nchains
    samples = np.empty((N, d + 1, Cn), dtype=np.float32) if save else None
    accepted = np.empty((N, Cn), dtype=np.uint8) if save else None
    fx = np.empty((d, Cn), dtype=np.float32)
    flp = np.empty(Cn, dtype=np.float32)
    cnt = np.empty(Cn, dtype=np.uint32)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.float32)
        assert init.shape == (d, Cn)
    rc = lib().orc_rwmh(C.byref(target.c), C.byref(prop.c), C.byref(sched), C.c_uint64(seed),
                        C.c_uint64(first_chain), Cn, _fp(init), _fp(samples), _u8p(accepted), _fp(fx),
                        _fp(flp), _u32p(cnt))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt)


def emcee(target, a, mode, sched, seed, ensemble_id, nwalkers, init, save=True):
    d, N, W = target.dim, sched.n_samples, nwalkers
    samples = np.empty((N, d + 1, W), dtype=np.float32) if save else None
    accepted = np.empty((N, W), dtype=np.uint8) if save else None
    fx = np.empty((d, W), dtype=np.float32)
    flp = np.empty(W, dtype=np.float32)
    cnt = np.empty(W, dtype=np.uint32)
    init = np.ascontiguousarray(init, dtype=np.float32)
    assert init.shape == (d, W)
    rc = lib().orc_emcee(C.byref(target.c), C.c_float(a), mode, C.byref(sched), C.c_uint64(seed),
                         C.c_uint64(ensemble_id), W, _fp(init), _fp(samples), _u8p(accepted), _fp(fx),
                         _fp(flp), _u32p(cnt))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt)


def ram(target, sched, seed, first_chain, nchains, init=None, S_in=None, alpha=0.234, gamma=0.6,
        eig_lo=0.0, eig_hi=float("inf"), save=True):
    d, N, Cn = target.dim, sched.n_samples, nchains
    nS = d * (d + 1) // 2
    cfg = _RamCfg(alpha, gamma, eig_lo, eig_hi)
    samples = np.empty((N, d + 1, Cn), dtype=np.float32) if save else None
    accepted = np.empty((N, Cn), dtype=np.uint8) if save else None
    fx = np.empty((d, Cn), dtype=np.float32)
    flp = np.empty(Cn, dtype=np.float32)
    cnt = np.empty(Cn, dtype=np.uint32)
    status = np.empty(Cn, dtype=np.uint8)
    S_out = np.empty((Cn, nS), dtype=np.float32)
    dmin = np.empty((d, Cn), dtype=np.float32)
    dmax = np.empty((d, Cn), dtype=np.float32)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.float32)
    if S_in is not None:
        S_in = np.ascontiguousarray(S_in, dtype=np.float32)
        assert S_in.shape == (Cn, nS)
    rc = lib().orc_ram(C.byref(target.c), C.byref(cfg), C.byref(sched), C.c_uint64(seed),
                       C.c_uint64(first_chain), Cn, _fp(init), _fp(S_in), _fp(S_out), _fp(samples),
                       _u8p(accepted), _fp(fx), _fp(flp), _u32p(cnt), _u8p(status), _fp(dmin), _fp(dmax))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt,
                status=status, S=S_out, diag_min=dmin, diag_max=dmax)


GRAD_FN = C.CFUNCTYPE(C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_void_p)


def target_grad(target, x, user_grad_addr=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    g = np.empty_like(x)
    L = lib()
    L.orc_target_grad.restype = C.c_float
    ug = C.cast(user_grad_addr, GRAD_FN) if user_grad_addr else C.cast(None, GRAD_FN)
    lp = L.orc_target_grad(C.byref(target.c), _fp(x), _fp(g), ug)
    return lp, g


def mala(target, sigma2, sched, seed, first_chain, nchains, init, user_grad_addr=None, save=True):
    d, N, Cn = target.dim, sched.n_samples, nchains
    samples = np.empty((N, d + 1, Cn), dtype=np.float32) if save else None
    accepted = np.empty((N, Cn), dtype=np.uint8) if save else None
    fx = np.empty((d, Cn), dtype=np.float32)
    flp = np.empty(Cn, dtype=np.float32)
    cnt = np.empty(Cn, dtype=np.uint32)
    init = np.ascontiguousarray(init, dtype=np.float32)
    assert init.shape == (d, Cn)
    ug = C.cast(user_grad_addr, GRAD_FN) if user_grad_addr else C.cast(None, GRAD_FN)
    rc = lib().orc_mala(C.byref(target.c), ug, C.c_float(sigma2), C.byref(sched), C.c_uint64(seed),
                        C.c_uint64(first_chain), Cn, _fp(init), _fp(samples), _u8p(accepted), _fp(fx), _fp(flp), _u32p(cnt))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt)


def chol_rank1(S_packed, w, sign):
    S = np.ascontiguousarray(S_packed, dtype=np.float32).copy()
    w = np.ascontiguousarray(w, dtype=np.float32).copy()
    d = w.size
    rc = lib().orc_chol_rank1(_fp(S), _fp(w), d, sign)
    return rc, S
