"""ctypes binding of libmhx.so (include/mhx.h).  This is the only way the Python host mirror reaches
the device: there is no CPU fallback -- if the HIP library is missing or fails, calls raise."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MHX_LIB") or os.path.join(os.path.dirname(_PKG), "libmhx.so")
# the tools build (`make -C csrc tools`): the same engine + timing probes and fault injection; never loaded unless asked for
TOOLS_LIB_PATH = os.path.join(os.path.dirname(_PKG), "libmhx_tools.so")

MHX_OK, MHX_EINVAL, MHX_ENOMEM, MHX_EHIP, MHX_EJIT, MHX_ENOTPD, MHX_ESTATE = 0, -1, -2, -3, -4, -5, -6

TARGET_ISO_GAUSS, TARGET_CORR_GAUSS, TARGET_IID_NORMAL, TARGET_BANANA, TARGET_FUNNEL = 0, 1, 2, 3, 4
TARGET_USER = 100
PROP_ISO, PROP_DIAG, PROP_DENSE = 0, 1, 2
FLAG_NO_JIT, FLAG_GENERIC = 1, 2
MHX_FLAG_STATIC_PROPOSAL = 4
FLAG_EMCEE_SEQUENTIAL = 8
FLAG_ZIGGURAT = 16
FLAG_DENSE_FACTOR = 32
FLAG_RAM_DEFERRED = 64


class MhxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmhx error %d: %s" % (code, msg))
        self.code = code


class ArgumentError(MhxError, ValueError):
    """MHX_EINVAL -- where the reference throws ArgumentError / MethodError."""


class PosDefException(MhxError):
    """MHX_ENOTPD -- LinearAlgebra.PosDefException of the reference."""


class Schedule(C.Structure):
    _fields_ = [("n_samples", C.c_int32), ("discard_initial", C.c_int32), ("thinning", C.c_int32),
                ("num_warmup", C.c_int32)]


# scalars travel as double and are rounded once to the context's dtype; real buffers are void* (include/mhx.h)
class RwmhCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("nchains", C.c_int32), ("seed", C.c_uint64), ("first_chain", C.c_uint64),
                ("proposal_kind", C.c_int32), ("proposal_scale", C.c_double),
                ("proposal_vec", C.c_void_p), ("flags", C.c_int32),
                ("proposal_mean", C.c_void_p), ("reduce_lanes", C.c_int32)]


class EmceeCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("nwalkers", C.c_int32), ("seed", C.c_uint64), ("ensemble_id", C.c_uint64),
                ("stretch", C.c_double), ("flags", C.c_int32), ("reduce_lanes", C.c_int32),
                ("init_kind", C.c_int32), ("init_scale", C.c_double), ("init_vec", C.c_void_p), ("init_mean", C.c_void_p),
                ("n_ensembles", C.c_int32)]


class RamCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("nchains", C.c_int32), ("seed", C.c_uint64), ("first_chain", C.c_uint64),
                ("alpha", C.c_double), ("gamma", C.c_double), ("eig_lo", C.c_double), ("eig_hi", C.c_double),
                ("flags", C.c_int32)]


class MalaCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("nchains", C.c_int32), ("seed", C.c_uint64), ("first_chain", C.c_uint64),
                ("sigma2", C.c_double), ("flags", C.c_int32), ("reduce_lanes", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("transitions", C.c_uint64), ("accepted", C.c_uint64), ("kernel_ms", C.c_double),
                ("wall_ms", C.c_double), ("kernel_variant", C.c_int32), ("launches", C.c_int32),
                ("reduce_lanes", C.c_int32), ("dtype", C.c_int32), ("normal_gen", C.c_int32), ("factor_band", C.c_int32),
                ("tainted", C.c_int32), ("reserved_", C.c_int32)]


class HostStats(C.Structure):
    """mhx_host_stats: what the last mhx_run_sample_to_host moved"""
    _fields_ = [("tensor_bytes", C.c_uint64), ("wire_bytes", C.c_uint64), ("link_ms", C.c_double), ("expand_ms", C.c_double),
                ("compact", C.c_int32), ("threads", C.c_int32), ("slabs", C.c_int32), ("ring", C.c_int32)]


class CompactHdr(C.Structure):
    """mhx_compact_hdr: the 64-byte header of an accept-compacted block"""
    _fields_ = [("magic", C.c_uint32), ("elem_bytes", C.c_uint32), ("dim1", C.c_uint32), ("nchains", C.c_uint32),
                ("first_sample", C.c_uint64), ("count", C.c_uint32), ("words", C.c_uint32), ("total_changed", C.c_uint64),
                ("payload_offset", C.c_uint64), ("block_bytes", C.c_uint64), ("reserved_", C.c_uint64)]


COMPACT_MAGIC = 0x4358484d


class DiagCfg(C.Structure):
    _fields_ = [("max_lag", C.c_int32), ("ess_chains", C.c_int32), ("split", C.c_int32)]


EXPORTS = [
    "mhx_version", "mhx_last_error", "mhx_ctx_create", "mhx_ctx_destroy", "mhx_target_builtin",
    "mhx_target_from_hip_source", "mhx_target_destroy", "mhx_target_eval", "mhx_rwmh_create",
    "mhx_emcee_create", "mhx_ram_create", "mhx_mala_create", "mhx_ram_set_factor", "mhx_ram_get_factor",
    "mhx_ram_get_diag_range", "mhx_run_init", "mhx_run_sample", "mhx_run_get_samples",
    "mhx_run_get_state", "mhx_run_set_state", "mhx_run_stats", "mhx_run_device_samples",
    "mhx_run_destroy", "mhx_run_diagnostics", "mhx_run_ess_bulk_tail", "mhx_emcee_half_step", "mhx_emcee_end_sweep",
    "mhx_emcee_device_state", "mhx_run_state_size", "mhx_run_save_state", "mhx_run_load_state",
    "mhx_ctx_dtype", "mhx_ctx_device", "mhx_ram_set_factor_all", "mhx_ram_get_adapt_state", "mhx_emcee_exchange_plan", "mhx_emcee_exchange_pack",
    "mhx_emcee_exchange_unpack", "mhx_comm_unique_id", "mhx_comm_init", "mhx_comm_destroy", "mhx_comm_rank",
    "mhx_comm_allreduce_sum", "mhx_comm_slice", "mhx_comm_allgather_walkers",
    "mhx_ram_get_step_stats", "mhx_ram_watch_factors", "mhx_ram_get_watched_factors", "mhx_run_sample_to_host", "mhx_host_alloc", "mhx_host_free", "mhx_ctx_jit_counts", "mhx_ctx_jit_compiler", "mhx_ctx_host_pin_counts",
    "mhx_ctx_set_option", "mhx_ctx_get_option", "mhx_ctx_pci_bus_id", "mhx_run_shape", "mhx_comm_init_timed", "mhx_comm_set_timeout",
    "mhx_group_create", "mhx_group_destroy", "mhx_group_size", "mhx_group_ctx", "mhx_group_shard", "mhx_group_attach", "mhx_group_run",
    "mhx_group_init", "mhx_group_sample", "mhx_group_sample_to_host", "mhx_group_stats", "mhx_group_diagnostics", "mhx_group_ess_bulk_tail",
    "mhx_compact_expand", "mhx_run_host_stats",
]

MHX_F32, MHX_F64 = 0, 1
DTYPES = {"f32": MHX_F32, "f64": MHX_F64}
NP_DTYPES = {"f32": np.float32, "f64": np.float64}
# the reference computes in Float64 (Distributions, src/RobustAdaptiveMetropolis.jl:187-196): that is the default
_default_dtype = os.environ.get("MHX_DTYPE", "f64")


def set_default_dtype(dt):
    """"f64" (the reference's arithmetic, default) or "f32" (same engine, half the bytes, ~3x the rate)."""
    global _default_dtype
    if dt not in DTYPES:
        raise ValueError("dtype must be 'f32' or 'f64'")
    _default_dtype = dt


def get_default_dtype():
    return _default_dtype

_lib = None
_lib_path = LIB_PATH
_loaded = {}


def use_library(path=None):
    """Bind the host mirror to another build of the library -- `use_library(TOOLS_LIB_PATH)` for the tools build (timing probes,
    fault injection: tests and A/B scripts), `use_library()` back to libmhx.so.  Contexts, models and runs of the previous
    binding must not be used afterwards (the default contexts are forgotten here)."""
    global _lib, _lib_path
    _lib_path = path or LIB_PATH
    _lib = None
    Context._default = {}
    Context._default_options = {}
    return lib()


def lib():
    """Load libmhx.so (built in-tree by __graft_entry__.build()).  Fails loudly when absent."""
    global _lib
    if _lib is None and _lib_path in _loaded:
        _lib = _loaded[_lib_path]
    if _lib is None:
        if not os.path.exists(_lib_path):
            raise ImportError("%s not found -- run `python __graft_entry__.py` (build()) first; "
                              "there is no CPU fallback" % _lib_path)
        L = C.CDLL(_lib_path, mode=C.RTLD_GLOBAL)
        L.mhx_last_error.restype = C.c_char_p
        vp, u8p, u32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        rp = C.c_void_p                                   # a buffer of reals of the context's dtype
        dp = C.POINTER(C.c_double)
        L.mhx_ctx_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
        L.mhx_ctx_destroy.argtypes = [vp]
        L.mhx_ctx_dtype.argtypes = [vp]
        L.mhx_ctx_device.argtypes = [vp, C.POINTER(C.c_int)]
        L.mhx_target_builtin.argtypes = [vp, C.c_int, C.c_int, rp, C.c_size_t, C.POINTER(vp)]
        L.mhx_target_from_hip_source.argtypes = [vp, C.c_char_p, C.c_int, rp, C.c_size_t, C.POINTER(vp)]
        L.mhx_target_destroy.argtypes = [vp]
        L.mhx_target_eval.argtypes = [vp, vp, rp, C.c_int, rp]
        L.mhx_rwmh_create.argtypes = [vp, vp, C.POINTER(RwmhCfg), C.POINTER(vp)]
        L.mhx_emcee_create.argtypes = [vp, vp, C.POINTER(EmceeCfg), C.POINTER(vp)]
        L.mhx_ram_create.argtypes = [vp, vp, C.POINTER(RamCfg), C.POINTER(vp)]
        L.mhx_mala_create.argtypes = [vp, vp, C.POINTER(MalaCfg), C.POINTER(vp)]
        L.mhx_ram_set_factor.argtypes = [vp, rp]
        L.mhx_ram_set_factor_all.argtypes = [vp, rp]
        L.mhx_ram_get_factor.argtypes = [vp, rp, u8p]
        L.mhx_ram_get_diag_range.argtypes = [vp, rp, rp]
        L.mhx_ram_get_adapt_state.argtypes = [vp, rp, dp, u8p, C.POINTER(C.c_uint64)]
        L.mhx_ram_get_step_stats.argtypes = [vp, rp, dp, C.c_int64, C.POINTER(C.c_int64)]
        L.mhx_ram_watch_factors.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32]
        L.mhx_ram_get_watched_factors.argtypes = [vp, rp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.mhx_run_init.argtypes = [vp, rp]
        L.mhx_run_sample.argtypes = [vp, C.POINTER(Schedule), C.c_int]
        L.mhx_run_get_samples.argtypes = [vp, rp, u8p]
        L.mhx_run_sample_to_host.argtypes = [vp, C.POINTER(Schedule), rp, u8p, C.c_int32]
        L.mhx_compact_expand.argtypes = [vp, C.c_size_t, rp, u8p, C.c_int64, C.c_int32]
        L.mhx_run_host_stats.argtypes = [vp, C.POINTER(HostStats)]
        L.mhx_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
        L.mhx_host_free.argtypes = [vp]
        L.mhx_ctx_jit_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mhx_ctx_jit_compiler.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64)]
        L.mhx_ctx_host_pin_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mhx_run_get_state.argtypes = [vp, rp, rp, u32p]
        L.mhx_run_set_state.argtypes = [vp, rp]
        L.mhx_run_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mhx_run_device_samples.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]
        L.mhx_run_destroy.argtypes = [vp]
        L.mhx_run_diagnostics.argtypes = [vp, C.POINTER(DiagCfg), dp, dp, dp, dp]
        L.mhx_run_state_size.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.mhx_run_save_state.argtypes = [vp, C.c_void_p, C.c_size_t]
        L.mhx_run_load_state.argtypes = [vp, C.c_void_p, C.c_size_t]
        L.mhx_emcee_half_step.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.mhx_emcee_end_sweep.argtypes = [vp]
        L.mhx_emcee_device_state.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(u32p), C.POINTER(u8p)]
        L.mhx_emcee_exchange_plan.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(vp)]
        L.mhx_emcee_exchange_pack.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.mhx_emcee_exchange_unpack.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
        L.mhx_run_ess_bulk_tail.argtypes = [vp, C.POINTER(DiagCfg), C.POINTER(C.c_int32), C.c_int32, dp, dp]
        L.mhx_comm_unique_id.argtypes = [vp]
        L.mhx_comm_init.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.mhx_comm_destroy.argtypes = [vp]
        L.mhx_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mhx_comm_allreduce_sum.argtypes = [vp, dp, C.c_size_t]
        L.mhx_comm_slice.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mhx_comm_allgather_walkers.argtypes = [vp, vp, C.c_int]
        L.mhx_comm_init_timed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_double, C.POINTER(vp)]
        L.mhx_comm_set_timeout.argtypes = [vp, C.c_double]
        L.mhx_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.mhx_ctx_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
        L.mhx_ctx_pci_bus_id.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.mhx_run_shape.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        pvp, i32p, i64p = C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.mhx_group_create.argtypes = [i32p, C.c_int32, C.c_int, pvp]
        L.mhx_group_destroy.argtypes = [vp]
        L.mhx_group_size.argtypes = [vp, i32p]
        L.mhx_group_ctx.argtypes = [vp, C.c_int32, pvp]
        L.mhx_group_shard.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_uint64), i32p]
        L.mhx_group_attach.argtypes = [vp, pvp]
        L.mhx_group_run.argtypes = [vp, C.c_int32, pvp]
        L.mhx_group_init.argtypes = [vp, pvp]
        L.mhx_group_sample.argtypes = [vp, C.POINTER(Schedule), C.c_int]
        L.mhx_group_sample_to_host.argtypes = [vp, C.POINTER(Schedule), pvp, pvp, C.c_int32]
        L.mhx_group_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mhx_group_diagnostics.argtypes = [vp, C.POINTER(DiagCfg), dp, dp, dp, dp, i64p]
        L.mhx_group_ess_bulk_tail.argtypes = [vp, C.POINTER(DiagCfg), i32p, C.c_int32, dp, dp]
        _lib = _loaded[_lib_path] = L
    return _lib


def check(rc):
    if rc == MHX_OK:
        return
    msg = lib().mhx_last_error().decode("utf-8", "replace")
    if rc == MHX_EINVAL:
        raise ArgumentError(rc, msg)
    if rc == MHX_ENOTPD:
        raise PosDefException(rc, msg)
    raise MhxError(rc, msg)


def rptr(a):
    """pointer to a contiguous array of reals (float32 or float64, the caller made it the context's dtype)"""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


fptr = rptr


def u8ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))


def u32ptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint32))


class _Pinned:
    """owner of one mhx_host_alloc block; freed when the last array viewing it is gone"""

    def __init__(self, nbytes):
        self.p = C.c_void_p()
        check(lib().mhx_host_alloc(nbytes, C.byref(self.p)))

    def __del__(self):
        try:
            if self.p:
                lib().mhx_host_free(self.p)
                self.p = C.c_void_p()
        except Exception:
            pass


def host_array(shape, dtype):
    """numpy array in page-locked host memory (mhx_host_alloc): device-to-host copies into it run at the link rate.
    The block is released when the array (and every view of it) has been garbage-collected."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64))
    nbytes = n * dtype.itemsize
    if nbytes == 0:
        return np.empty(shape, dtype=dtype)
    owner = _Pinned(nbytes)
    buf = (C.c_ubyte * nbytes).from_address(owner.p.value)
    buf._mhx_owner = owner                                 # the ctypes array is the numpy array's base: it keeps the block alive
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


class Context:
    """mhx_ctx: one per GPU and dtype."""

    _default = {}
    _default_options = {}      # set_default_option: applied to the default contexts, present and future

    def __init__(self, device=0, dtype=None, handle=None):
        """`handle`: wrap a context somebody else owns (a member of a Group) instead of creating one."""
        dtype = dtype or _default_dtype
        if dtype not in DTYPES:
            raise ValueError("dtype must be 'f32' or 'f64'")
        self.h = C.c_void_p()
        self.borrowed = handle is not None
        if handle is not None:
            self.h = handle
        else:
            check(lib().mhx_ctx_create(device, DTYPES[dtype], C.byref(self.h)))
        self.device = device
        self.dtype = dtype
        self.real = NP_DTYPES[dtype]

    def set_option(self, name, value):
        """mhx_ctx_set_option: an explicit engine option (kernel form / tuning; include/mhx.h lists them) for the runs created on
        this context from now on.  value None unsets.  The library reads no such thing from the environment."""
        check(lib().mhx_ctx_set_option(self.h, str(name).encode(), None if value is None else str(value).encode()))

    def get_option(self, name):
        buf = C.create_string_buffer(256)
        check(lib().mhx_ctx_get_option(self.h, str(name).encode(), buf, len(buf)))
        return buf.value.decode()

    def pci_bus_id(self):
        """"dddd:bb:dd.f" of the device (independent of HIP_VISIBLE_DEVICES ordinals): distinct ids = distinct GPUs"""
        buf = C.create_string_buffer(32)
        check(lib().mhx_ctx_pci_bus_id(self.h, buf, len(buf)))
        return buf.value.decode()

    @classmethod
    def set_default_option(cls, name, value):
        """the option on every default context (Context.default) that exists or will be created; None unsets"""
        if value is None:
            cls._default_options.pop(name, None)
        else:
            cls._default_options[name] = str(value)
        for c in cls._default.values():
            c.set_option(name, value)

    @classmethod
    def clear_default_options(cls):
        for name in list(cls._default_options):
            cls.set_default_option(name, None)

    def arr(self, a):
        """contiguous array in this context's real type"""
        return np.ascontiguousarray(a, dtype=self.real)

    def jit_counts(self):
        """(hiprtc compilations, code objects taken from the on-disk cache) of this context so far"""
        a, b = C.c_int64(), C.c_int64()
        check(lib().mhx_ctx_jit_counts(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def jit_compiler(self):
        """(identity of the installation's clang++ that builds the run-time kernels, "" = hiprtc only; compilations it did so far)"""
        buf, n = C.create_string_buffer(512), C.c_int64()
        check(lib().mhx_ctx_jit_compiler(self.h, buf, 512, C.byref(n)))
        return buf.value.decode(), int(n.value)

    def host_pin_counts(self):
        """(caller buffers page-locked for a mhx_run_sample_to_host call, released again) so far: equal between calls"""
        a, b = C.c_int64(), C.c_int64()
        check(lib().mhx_ctx_host_pin_counts(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    @classmethod
    def default(cls, device=None, dtype=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("MHX_DEVICE") is None else int(
                os.environ["MHX_DEVICE"])
        dtype = dtype or _default_dtype
        if (device, dtype) not in cls._default:
            c = cls._default[(device, dtype)] = cls(device, dtype)
            for name, value in cls._default_options.items():
                c.set_option(name, value)
        return cls._default[(device, dtype)]

    def close(self):
        if self.h and not self.borrowed:
            lib().mhx_ctx_destroy(self.h)
        self.h = C.c_void_p()
