"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE -- never imported by the product path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Two builds of one source (oracle/Makefile): libmhx_oracle.so computes in float (the fp32 engine), libmhx_oracle64.so in
double (the reference's Float64).  `set_dtype("f32" | "f64")` selects which one the module-level functions bind.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATHS = {"f32": os.path.join(_HERE, "libmhx_oracle.so"), "f64": os.path.join(_HERE, "libmhx_oracle64.so")}

STREAM_PROPOSAL, STREAM_ACCEPT, STREAM_INIT, STREAM_EMCEE = 0, 1, 2, 3
TARGET_ISO_GAUSS, TARGET_CORR_GAUSS, TARGET_IID_NORMAL, TARGET_BANANA, TARGET_FUNNEL = 0, 1, 2, 3, 4
TARGET_CALLBACK = 100
PROP_ISO, PROP_DIAG, PROP_DENSE = 0, 1, 2

_DT = "f32"


def set_dtype(dt):
    """Select the arithmetic of every call below: "f32" or "f64"."""
    global _DT
    if dt not in _LIB_PATHS:
        raise ValueError("dtype must be 'f32' or 'f64'")
    _DT = dt


def get_dtype():
    return _DT


def real():
    """numpy dtype of the current build"""
    return np.float64 if _DT == "f64" else np.float32


def _creal():
    return C.c_double if _DT == "f64" else C.c_float


def _mk_types(cr):
    fn = C.CFUNCTYPE(cr, C.c_void_p, C.c_int, C.c_void_p)
    gfn = C.CFUNCTYPE(cr, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

    class Target_(C.Structure):
        _fields_ = [("kind", C.c_int), ("dim", C.c_int), ("params", C.c_void_p),
                    ("nparams", C.c_int), ("fn", fn), ("fn_data", C.c_void_p), ("reduce_lanes", C.c_int)]

    class Proposal_(C.Structure):
        _fields_ = [("kind", C.c_int), ("scale", cr), ("vec", C.c_void_p), ("mean", C.c_void_p), ("is_static", C.c_int), ("normal_gen", C.c_int)]

    class RamCfg_(C.Structure):
        _fields_ = [("alpha", cr), ("gamma", cr), ("eig_lo", cr), ("eig_hi", cr)]

    return dict(fn=fn, gfn=gfn, Target=Target_, Proposal=Proposal_, RamCfg=RamCfg_)


_TYPES = {"f32": _mk_types(C.c_float), "f64": _mk_types(C.c_double)}


def _T(name):
    return _TYPES[_DT][name]


class _Schedule(C.Structure):
    _fields_ = [("n_samples", C.c_int), ("discard_initial", C.c_int), ("thinning", C.c_int),
                ("num_warmup", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "mhx_oracle.c")
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src) for p in _LIB_PATHS.values()):
        if os.path.exists(src):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATHS[_DT]


_libs = {}


def use_native():
    """bench.py's cpu_baseline leg: rebuild the oracle with -O3 -march=native ON THIS HOST (oracle/Makefile `native`, into
    oracle/_native/) and bind the module to it.  Returns the flags string in effect; on any failure the portable build
    (-O3 -mavx2 -mfma) stays bound.  Bit-identical results either way (-ffp-contract=off in both)."""
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nat = {"f32": os.path.join(_HERE, "_native", "libmhx_oracle.so"), "f64": os.path.join(_HERE, "_native", "libmhx_oracle64.so")}
        for p in nat.values():
            C.CDLL(p)                                   # loads on this host?
        _LIB_PATHS.update(nat)
        _libs.clear()
        return "gcc -O3 -march=native -ffp-contract=off"
    except Exception:
        return "gcc -O3 -mavx2 -mfma -ffp-contract=off (portable build; the -march=native rebuild failed on this host)"


def lib():
    if _DT not in _libs:
        if not os.path.exists(_LIB_PATHS[_DT]):
            build()
        L = C.CDLL(_LIB_PATHS[_DT])
        cr = _creal()
        L.orc_log.restype = cr
        L.orc_log.argtypes = [cr]
        L.orc_exp.restype = cr
        L.orc_exp.argtypes = [cr]
        L.orc_u01_open.restype = cr
        L.orc_u01_half.restype = cr
        if _DT == "f64":
            L.orc_u01_open.argtypes = [C.c_uint32, C.c_uint32]
            L.orc_u01_half.argtypes = [C.c_uint32, C.c_uint32]
        else:
            L.orc_u01_open.argtypes = [C.c_uint32]
            L.orc_u01_half.argtypes = [C.c_uint32]
        L.orc_accept_logu.restype = cr
        L.orc_accept_logu.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_target_eval.restype = cr
        L.orc_target_eval.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_target_grad.restype = cr
        L.orc_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_set_trace.restype = None
        L.orc_set_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_chol_rank1.restype = C.c_int
        L.orc_chol_rank1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        for f in (L.orc_mt_rwmh, L.orc_mt_ram):
            f.restype = C.c_double
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _libs[_DT] = L
    return _libs[_DT]


def rarr(a):
    """contiguous array in the current real type"""
    return np.ascontiguousarray(a, dtype=real())


def _fp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


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


def log(x):
    L = lib()
    return np.array([L.orc_log(float(v)) for v in np.atleast_1d(x)], dtype=real())


def exp(x):
    L = lib()
    return np.array([L.orc_exp(float(v)) for v in np.atleast_1d(x)], dtype=real())


logf, expf = log, exp          # the fp32 names of round 1


def sincos2pi_u32(k):
    """f32 build: sin / cos of 2 pi k / 2^32"""
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos2pi_u32(C.c_uint32(int(k)), C.byref(s), C.byref(c))
    return s.value, c.value


def sincos2pi_u64(k):
    """f64 build: sin / cos of 2 pi k / 2^64"""
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos2pi_u64(C.c_uint32(int(k) >> 32), C.c_uint32(int(k) & 0xffffffff), C.byref(s), C.byref(c))
    return s.value, c.value


def u01_open(*words):
    return lib().orc_u01_open(*[C.c_uint32(int(w)) for w in words])


def u01_half(*words):
    return lib().orc_u01_half(*[C.c_uint32(int(w)) for w in words])


def normals(seed, chain, step, stream, d):
    out = np.empty(d, dtype=real())
    lib().orc_normals(seed, chain, step, stream, d, _fp(out))
    return out


def zig_normals(seed, chain, step, stream, d):
    """d standard normals of (seed, chain, step, stream) by the ziggurat generator (the build's own form: 64 / 32 bits per normal)"""
    out = np.empty(d, dtype=real())
    lib().orc_normals_gen(1, C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(step), C.c_uint32(stream), d, _fp(out))
    return out


def accept_logu(seed, chain, step):
    return lib().orc_accept_logu(seed, chain, step)


class Target:
    """kind + params, or a Python callable f(x: np.ndarray) -> float (DensityModel(f))."""

    def __init__(self, kind, dim, params=None, fn=None, fn_data=None, reduce_lanes=0):
        self.kind, self.dim = kind, dim
        self._fn_data = fn_data             # keep-alive for a ctypes object passed as void*
        self.params = None if params is None else np.ascontiguousarray(params, dtype=real())
        self._cb = None
        self._fn_addr = None
        if fn is not None:
            if isinstance(fn, int):             # raw C function pointer (e.g. from a gcc-built user source)
                self._fn_addr = fn
            else:
                cr, npr = _creal(), real()

                def _tramp(xp, d, _data, fn=fn):
                    return float(fn(np.ctypeslib.as_array(C.cast(xp, C.POINTER(cr)), shape=(d,)).astype(npr)))
                self._cb = _T("fn")(_tramp)
        self.c = _T("Target")()
        self.c.kind = kind
        self.c.dim = dim
        self.c.params = _fp(self.params)
        self.c.nparams = 0 if self.params is None else int(self.params.size)
        if self._cb is not None:
            self.c.fn = self._cb
        elif self._fn_addr is not None:
            self.c.fn = C.cast(self._fn_addr, _T("fn"))
        self.c.fn_data = None if fn_data is None else C.cast(C.pointer(fn_data), C.c_void_p)
        self.c.reduce_lanes = int(reduce_lanes)

    def with_lanes(self, L):
        """the same target with the L-lane reduction shape of the cooperative kernels"""
        self.c.reduce_lanes = int(L)
        return self

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=real())
        return lib().orc_target_eval(C.byref(self.c), _fp(x))


def iso_gauss(d, reduce_lanes=0):
    return Target(TARGET_ISO_GAUSS, d, reduce_lanes=reduce_lanes)


def corr_gauss_from_cov(Sigma, reduce_lanes=0):
    """params = inv(chol(Sigma)) packed lower row-major, computed in float64 then rounded.  The checker must be handed the same
    numbers as the device, so this restates the host mirror's rule for a banded factor (mhx.precision_factor): off-diagonal
    entries at the round-off level of the inversion, |A_ij| <= 256 eps |A_jj|, become exact zeros IF the cleaned factor
    is then banded (bandwidth <= 8); otherwise the raw inverse is used."""
    Sigma = np.asarray(Sigma, dtype=np.float64)
    A = np.tril(np.linalg.inv(np.linalg.cholesky(Sigma)))
    d = A.shape[0]
    dg = np.abs(np.diag(A))
    noise = np.abs(A) <= 256.0 * np.finfo(np.float64).eps * dg[None, :]
    noise[np.arange(d), np.arange(d)] = False
    B = np.where(noise, 0.0, A)
    i, j = np.nonzero(B)
    if d > 1 and int((i - j).max()) <= min(8, d - 2):
        A = B
    return Target(TARGET_CORR_GAUSS, Sigma.shape[0], pack_lower(A), reduce_lanes=reduce_lanes)


def pack_lower(M):
    M = np.asarray(M)
    d = M.shape[0]
    return np.concatenate([M[i, :i + 1] for i in range(d)]).astype(real())


def unpack_lower(p, d):
    M = np.zeros((d, d), dtype=real())
    o = 0
    for i in range(d):
        M[i, :i + 1] = p[o:o + i + 1]
        o += i + 1
    return M


class Proposal:
    def __init__(self, kind, scale=1.0, vec=None, mean=None, static=False, normal_gen=0):
        """normal_gen: 0 Box-Muller, 1 the table ziggurat (what MHX_FLAG_ZIGGURAT selects on the device; 64 bits per normal in the
        fp64 build, 32 in the fp32 build)"""
        self.vec = None if vec is None else np.ascontiguousarray(vec, dtype=real())
        self.mean = None if mean is None else np.ascontiguousarray(mean, dtype=real())
        self.c = _T("Proposal")(kind, float(scale), _fp(self.vec), _fp(self.mean), 1 if static else 0, int(normal_gen))


def schedule(n_samples, discard_initial=0, thinning=1, num_warmup=0):
    return _Schedule(n_samples, discard_initial, thinning, num_warmup)


def schedule_counts(s):
    a, b = C.c_int64(), C.c_int64()
    lib().orc_schedule_counts(C.byref(s), C.byref(a), C.byref(b))
    return a.value, b.value


def traced(fn, *args, **kw):
    """Run a sampler of this module (rwmh / emcee / ram / mala) with the trace sink on.  The result gains `margin`
    [n_samples][C] (the smallest |logu - logalpha| over the transitions that led to each saved slot: how far the closest
    accept decision was from flipping) and, for RAM, `logalpha` / `eta` (state.logalpha, state.eta after the saved transition)."""
    sched = next(a for a in args if isinstance(a, _Schedule))
    ncol = [a for a in args if isinstance(a, (int, np.integer))][-1]          # nchains / nwalkers: the last integer argument
    bufs = [np.full((sched.n_samples, ncol), np.nan, dtype=real()) for _ in range(3)]
    lib().orc_set_trace(_fp(bufs[0]), _fp(bufs[1]), _fp(bufs[2]))
    try:
        r = fn(*args, **kw)
    finally:
        lib().orc_set_trace(None, None, None)
    r["margin"] = bufs[0]
    if fn is ram:
        r["logalpha"], r["eta"] = bufs[1], bufs[2]
    return r


def rwmh(target, prop, sched, seed, first_chain, nchains, init=None, save=True):
    d, N, Cn = target.dim, sched.n_samples, nchains
    samples = np.empty((N, d + 1, Cn), dtype=real()) if save else None
    accepted = np.empty((N, Cn), dtype=np.uint8) if save else None
    fx = np.empty((d, Cn), dtype=real())
    flp = np.empty(Cn, dtype=real())
    cnt = np.empty(Cn, dtype=np.uint32)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=real())
        assert init.shape == (d, Cn)
    rc = lib().orc_rwmh(C.byref(target.c), C.byref(prop.c), C.byref(sched), C.c_uint64(seed),
                        C.c_uint64(first_chain), Cn, _fp(init), _fp(samples), _u8p(accepted), _fp(fx),
                        _fp(flp), _u32p(cnt))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt)


def emcee(target, a, mode, sched, seed, ensemble_id, nwalkers, init, save=True, prior=None):
    """init [d][W], or None with `prior` (a Proposal: the (Mv)Normal the StretchProposal wraps): W draws from it"""
    d, N, W = target.dim, sched.n_samples, nwalkers
    samples = np.empty((N, d + 1, W), dtype=real()) if save else None
    accepted = np.empty((N, W), dtype=np.uint8) if save else None
    fx = np.empty((d, W), dtype=real())
    flp = np.empty(W, dtype=real())
    cnt = np.empty(W, dtype=np.uint32)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=real())
        assert init.shape == (d, W)
    rc = lib().orc_emcee(C.byref(target.c), _creal()(a), mode, C.byref(sched), C.c_uint64(seed),
                         C.c_uint64(ensemble_id), W, _fp(init), C.byref(prior.c) if prior is not None else None, _fp(samples),
                         _u8p(accepted), _fp(fx), _fp(flp), _u32p(cnt))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt)


def ram(target, sched, seed, first_chain, nchains, init=None, S_in=None, alpha=0.234, gamma=0.6,
        eig_lo=0.0, eig_hi=float("inf"), save=True):
    d, N, Cn = target.dim, sched.n_samples, nchains
    nS = d * (d + 1) // 2
    cfg = _T("RamCfg")(alpha, gamma, eig_lo, eig_hi)
    samples = np.empty((N, d + 1, Cn), dtype=real()) if save else None
    accepted = np.empty((N, Cn), dtype=np.uint8) if save else None
    fx = np.empty((d, Cn), dtype=real())
    flp = np.empty(Cn, dtype=real())
    cnt = np.empty(Cn, dtype=np.uint32)
    status = np.empty(Cn, dtype=np.uint8)
    S_out = np.empty((Cn, nS), dtype=real())
    dmin = np.empty((d, Cn), dtype=real())
    dmax = np.empty((d, Cn), dtype=real())
    if init is not None:
        init = np.ascontiguousarray(init, dtype=real())
    if S_in is not None:
        S_in = np.ascontiguousarray(S_in, dtype=real())
        assert S_in.shape == (Cn, nS)
    rc = lib().orc_ram(C.byref(target.c), C.byref(cfg), C.byref(sched), C.c_uint64(seed),
                       C.c_uint64(first_chain), Cn, _fp(init), _fp(S_in), _fp(S_out), _fp(samples),
                       _u8p(accepted), _fp(fx), _fp(flp), _u32p(cnt), _u8p(status), _fp(dmin), _fp(dmax))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt,
                status=status, S=S_out, diag_min=dmin, diag_max=dmax)




def ram_flush_points(n_transitions, n_adapt=None, max_launch=4096):
    """the transitions after which the engine's deferred-factor RAM kernel folds its pending updates into the stored factor
    besides `K pending`: the ends of its launches (a sampling call is cut into launches of `max_launch` transitions,
    csrc/mhx_api_ram.inc MHX_RAM_MAX_STEPS_PER_LAUNCH) -- the end of the warm-up and of the call are flush points by themselves"""
    return list(range(max_launch, n_transitions, max_launch))


def ram_deferred(target, sched, seed, first_chain, nchains, init=None, S_in=None, alpha=0.234, gamma=0.6,
                 eig_lo=0.0, eig_hi=float("inf"), save=True, K=8, flush_at=None):
    """the twin of MHX_FLAG_RAM_DEFERRED (orc_ram_deferred): `ram` in exact arithmetic, its own rounding"""
    d, N, Cn = target.dim, sched.n_samples, nchains
    nS = d * (d + 1) // 2
    cfg = _T("RamCfg")(alpha, gamma, eig_lo, eig_hi)
    samples = np.empty((N, d + 1, Cn), dtype=real()) if save else None
    accepted = np.empty((N, Cn), dtype=np.uint8) if save else None
    fx = np.empty((d, Cn), dtype=real())
    flp = np.empty(Cn, dtype=real())
    cnt = np.empty(Cn, dtype=np.uint32)
    status = np.empty(Cn, dtype=np.uint8)
    S_out = np.empty((Cn, nS), dtype=real())
    dmin = np.empty((d, Cn), dtype=real())
    dmax = np.empty((d, Cn), dtype=real())
    if init is not None:
        init = np.ascontiguousarray(init, dtype=real())
    if S_in is not None:
        S_in = np.ascontiguousarray(S_in, dtype=real())
        assert S_in.shape == (Cn, nS)
    fl = np.ascontiguousarray(flush_at if flush_at is not None else [], dtype=np.int64)
    f = lib().orc_ram_deferred
    f.restype = C.c_int
    rc = f(C.byref(target.c), C.byref(cfg), C.byref(sched), C.c_uint64(seed),
           C.c_uint64(first_chain), C.c_int(Cn), _fp(init), _fp(S_in), _fp(S_out), _fp(samples),
           _u8p(accepted), _fp(fx), _fp(flp), _u32p(cnt), _u8p(status), _fp(dmin), _fp(dmax),
           C.c_int(K), fl.ctypes.data_as(C.c_void_p), C.c_int(len(fl)))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt,
                status=status, S=S_out, diag_min=dmin, diag_max=dmax)


def mt_rwmh(target, prop, sched, seed, first_chain, nchains, nthreads, save=True, init1=None):
    """bench.py's CPU baseline (oracle/mhx_oracle_mt.c): `nchains` independent chains on `nthreads` POSIX threads, each chain
    one orc_rwmh call with its own contiguous [N][d+1] record.  Returns (wall seconds, per-thread CPU seconds)."""
    busy = np.zeros(max(1, nthreads), dtype=np.float64)
    init1 = None if init1 is None else np.ascontiguousarray(init1, dtype=real())
    w = lib().orc_mt_rwmh(C.addressof(target.c), C.addressof(prop.c), C.addressof(sched), seed, first_chain, nchains, nthreads,
                          1 if save else 0, _fp(init1), _fp(busy))
    return w, busy


def mt_ram(target, sched, seed, first_chain, nchains, nthreads, save=False, init1=None, alpha=0.234, gamma=0.6):
    """the same for RobustAdaptiveMetropolis chains (identity start factor each)"""
    busy = np.zeros(max(1, nthreads), dtype=np.float64)
    cfg = _T("RamCfg")(alpha, gamma, 0.0, float("inf"))
    init1 = None if init1 is None else np.ascontiguousarray(init1, dtype=real())
    w = lib().orc_mt_ram(C.addressof(target.c), C.addressof(cfg), C.addressof(sched), seed, first_chain, nchains, nthreads,
                         1 if save else 0, _fp(init1), _fp(busy))
    return w, busy


def target_grad(target, x, user_grad_addr=None):
    x = np.ascontiguousarray(x, dtype=real())
    g = np.empty_like(x)
    L = lib()
    ug = C.cast(user_grad_addr, _T("gfn")) if user_grad_addr else C.cast(None, _T("gfn"))
    lp = L.orc_target_grad(C.byref(target.c), _fp(x), _fp(g), ug)
    return lp, g


def mala(target, sigma2, sched, seed, first_chain, nchains, init, user_grad_addr=None, save=True, normal_gen=0):
    d, N, Cn = target.dim, sched.n_samples, nchains
    samples = np.empty((N, d + 1, Cn), dtype=real()) if save else None
    accepted = np.empty((N, Cn), dtype=np.uint8) if save else None
    fx = np.empty((d, Cn), dtype=real())
    flp = np.empty(Cn, dtype=real())
    cnt = np.empty(Cn, dtype=np.uint32)
    init = np.ascontiguousarray(init, dtype=real())
    assert init.shape == (d, Cn)
    ug = C.cast(user_grad_addr, _T("gfn")) if user_grad_addr else C.cast(None, _T("gfn"))
    rc = lib().orc_mala(C.byref(target.c), ug, _creal()(sigma2), C.byref(sched), C.c_uint64(seed),
                        C.c_uint64(first_chain), Cn, _fp(init), _fp(samples), _u8p(accepted), _fp(fx), _fp(flp), _u32p(cnt),
                        int(normal_gen))
    assert rc == 0
    return dict(samples=samples, accepted=accepted, final_x=fx, final_lp=flp, accept_counts=cnt)


def chol_rank1(S_packed, w, sign):
    S = np.ascontiguousarray(S_packed, dtype=real()).copy()
    w = np.ascontiguousarray(w, dtype=real()).copy()
    d = w.size
    rc = lib().orc_chol_rank1(_fp(S), _fp(w), d, sign)
    return rc, S
