"""Multi-GPU plumbing.  One process per GPU; chains shard by GLOBAL chain id (no data-path collective); the only
collective of a sharded run is ONE all-reduce of 3(dim+1)+3 doubles per reporting interval -- acceptance totals and the
R-hat / ESS sums.  ONE ensemble sharded over the GPUs adds ONE all-gather per half-step.

On the device the collectives go through the C ABI (`mhx_comm_*` of include/mhx.h: RCCL over xGMI, the same entry points
the Julia glue binds): class `Comm`.  The torch.distributed forms below (`group=` arguments, backend gloo) exist for the
world_size-2 CPU tests of the combining / slicing logic, where there is no device."""
import ctypes as C
import os

import numpy as np


def shard_chains(nchains_total, rank, world):
    """Contiguous block of global chain ids for `rank`: (first_chain, count)."""
    base, rem = divmod(int(nchains_total), int(world))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def pack_stats(diag, accepted, transitions):
    return np.concatenate([diag["sum_m"], diag["sum_m2"], diag["sum_v"],
                           [float(accepted), float(transitions), float(diag["n_chains"])]]).astype(np.float64)


class Group:
    """mhx_group: many chains over many GPUs as ONE call from ONE process -- the `sample(model, spl, MCMCThreads(), N, nchains)` of
    the reference (README.md:135-148) with one host thread per GPU behind the C ABI instead of one task per chain.

        g = Group([0, 1, 2, 3])                       # member i on device devices[i]; entries may repeat ([0, 0]: two members on GPU 0)
        g.create(model, sampler, nchains=65536, seed=1)    # member runs over contiguous blocks of global chain ids, attached
        g.init(None); g.sample(250, 1, 1, 0)          # all members side by side
        g.stats(); g.diagnostics()                    # totals / R-hat over ALL chains (host sum of what ranks all-reduce over RCCL)

    The union of the members' chains is the unsharded run bit for bit (tests/test_gpu_group.py)."""

    def __init__(self, devices, dtype=None):
        from . import _lib as L
        self.dtype = dtype or L.get_default_dtype()
        self.devices = [int(d) for d in devices]
        dv = (C.c_int32 * len(self.devices))(*self.devices)
        self.h = C.c_void_p()
        L.check(L.lib().mhx_group_create(dv, len(self.devices), L.DTYPES[self.dtype], C.byref(self.h)))
        self.ctxs = []
        for i, dev in enumerate(self.devices):
            ch = C.c_void_p()
            L.check(L.lib().mhx_group_ctx(self.h, i, C.byref(ch)))
            self.ctxs.append(L.Context(dev, self.dtype, handle=ch))
        self.runs = []

    def __len__(self):
        return len(self.devices)

    def shard(self, nchains_total, i):
        from . import _lib as L
        first, cnt = C.c_uint64(), C.c_int32()
        L.check(L.lib().mhx_group_shard(self.h, int(nchains_total), i, C.byref(first), C.byref(cnt)))
        return int(first.value), int(cnt.value)

    def set_option(self, name, value):
        for c in self.ctxs:
            c.set_option(name, value)

    def pci_bus_ids(self):
        return [c.pci_bus_id() for c in self.ctxs]

    def create(self, model, sampler, nchains=1, seed=0, first_chain=0, **kw):
        """One Run per member: chains [first_chain + shard) by global id (an Ensemble: whole ensembles by global id, at least one per member)."""
        from .api import Ensemble, Run
        self.close_runs()
        for i, ctx in enumerate(self.ctxs):
            if isinstance(sampler, Ensemble):
                # nchains ENSEMBLES (README.md:135-148) dealt to the members by global id, at least one each; a member runs its
                # ensembles in one launch (mhx_emcee_cfg.n_ensembles)
                f, n = self.shard(max(nchains, len(self.ctxs)), i)
                self.runs.append(Run(model, sampler, nchains=n, seed=seed, first_chain=first_chain + f, ctx=ctx, **kw))
            else:
                f, n = self.shard(nchains, i)
                self.runs.append(Run(model, sampler, nchains=n, seed=seed, first_chain=first_chain + f, ctx=ctx, **kw))
        return self.attach(self.runs)

    def attach(self, runs):
        from . import _lib as L
        self.runs = list(runs)
        hs = (C.c_void_p * len(self.runs))(*[r.h for r in self.runs])
        L.check(L.lib().mhx_group_attach(self.h, hs))
        self.dim = self.runs[0].dim
        self.n = sum(r.n for r in self.runs)
        return self

    def init(self, initial_params=None):
        """None: every member draws its chains' initial states on the device.  (dim,) for all chains, or (dim, n_total): the columns are
        dealt to the members in chain order."""
        from . import _lib as L
        if initial_params is None:
            L.check(L.lib().mhx_group_init(self.h, None))
            return
        ip = np.asarray(initial_params)
        parts, o = [], 0
        for r in self.runs:
            if ip.ndim == 1:
                parts.append(r.ctx.arr(np.repeat(ip.reshape(-1, 1), r.n, axis=1)))
            else:
                parts.append(r.ctx.arr(ip[:, o:o + r.n]))
                o += r.n
        ptrs = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
        L.check(L.lib().mhx_group_init(self.h, ptrs))

    def sample(self, n_samples, discard_initial=0, thinning=1, num_warmup=0, save=True):
        from . import _lib as L
        s = L.Schedule(n_samples, discard_initial, thinning, num_warmup)
        mode = 2 if (isinstance(save, str) and save == "moments") or (save is not True and save == 2) else (1 if save else 0)
        L.check(L.lib().mhx_group_sample(self.h, C.byref(s), mode))

    def sample_to_host(self, n_samples, discard_initial=0, thinning=1, num_warmup=0, want_accepted=True, slab_samples=0, out=None,
                       out_accepted=None):
        """mhx_group_sample_to_host: returns (list of per-member tensors [N][dim+1][n_i], list of accepted [N][n_i] or None);
        `np.concatenate(values, axis=2)` is the tensor of the unsharded run."""
        from . import _lib as L
        s = L.Schedule(n_samples, discard_initial, thinning, num_warmup)
        vals = out or [L.host_array((n_samples, self.dim + 1, r.n), r.real) for r in self.runs]
        accs = out_accepted or ([L.host_array((n_samples, r.n), np.uint8) for r in self.runs] if want_accepted else None)
        vp = (C.c_void_p * len(vals))(*[v.ctypes.data for v in vals])
        ap = (C.c_void_p * len(vals))(*[a.ctypes.data for a in accs]) if accs is not None else None
        L.check(L.lib().mhx_group_sample_to_host(self.h, C.byref(s), vp, ap, slab_samples))
        return vals, accs

    def stats(self):
        from . import _lib as L
        st = L.Stats()
        L.check(L.lib().mhx_group_stats(self.h, C.byref(st)))
        return dict(transitions=st.transitions, accepted=st.accepted, kernel_ms=st.kernel_ms, wall_ms=st.wall_ms,
                    kernel_variant=st.kernel_variant, launches=st.launches, reduce_lanes=st.reduce_lanes,
                    dtype="f64" if st.dtype == L.MHX_F64 else "f32", normal_gen=st.normal_gen, factor_band=st.factor_band,
                    tainted=st.tainted)

    def diagnostics(self, max_lag=0, ess_chains=256, split=False):
        """R-hat / between-chain ESS over ALL chains of the group from the host-summed statistics (mhx_group_diagnostics)"""
        from . import _lib as L
        from .api import combine_diagnostics
        d1 = self.dim + 1
        arrs = [np.zeros(d1, dtype=np.float64) for _ in range(4)]
        ptrs = [a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs]
        if max_lag <= 0:
            ptrs[3] = None
        nch = C.c_int64()
        cfg = L.DiagCfg(max_lag, ess_chains, 1 if split else 0)
        L.check(L.lib().mhx_group_diagnostics(self.h, C.byref(cfg), *ptrs, C.byref(nch)))
        n_saved = C.c_int64()
        L.check(L.lib().mhx_run_device_samples(self.runs[0].h, None, None, C.byref(n_saved)))
        ns = int(n_saved.value) // (2 if split else 1)
        if ns == 0 and getattr(self.runs[0], "_moments_n", 0):
            ns = self.runs[0]._moments_n
        out = dict(sum_m=arrs[0], sum_m2=arrs[1], sum_v=arrs[2], n_chains=int(nch.value), n_samples=ns)
        if max_lag > 0:
            out["ess_geyer"] = np.abs(arrs[3])
            out["ess_geyer_truncated"] = arrs[3] < 0
        out.update(combine_diagnostics(out["sum_m"], out["sum_m2"], out["sum_v"], out["n_chains"], max(ns, 1)))
        return out

    def close_runs(self):
        for r in self.runs:
            r.close()
        self.runs = []

    def close(self):
        from . import _lib as L
        self.close_runs()
        if self.h:
            L.lib().mhx_group_destroy(self.h)
            self.h = C.c_void_p()
            for c in self.ctxs:
                c.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """mhx_comm: an RCCL communicator behind the C ABI.  `Comm(ctx, rank, world, unique_id)`; `Comm.from_env(ctx)` takes
    rank / world / rendezvous from the torchrun environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) and hands the
    128-byte id from rank 0 to the others through a TCP key-value store (no process group is created)."""

    def __init__(self, ctx, rank, world, unique_id):
        from . import _lib as L
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        L.check(L.lib().mhx_comm_init(ctx.h, self.rank, self.world, C.cast(buf, C.c_void_p), C.byref(self.h)))

    @staticmethod
    def unique_id():
        from . import _lib as L
        buf = (C.c_char * 128)()
        L.check(L.lib().mhx_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @classmethod
    def from_env(cls, ctx, key="mhx_comm_id"):
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return cls(ctx, 0, 1, cls.unique_id())
        from torch.distributed import TCPStore
        # next to torchrun's own store (MASTER_PORT), not on it
        port = int(os.environ.get("MHX_STORE_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, world, rank == 0)
        if rank == 0:
            store.set(key, cls.unique_id())
        uid = store.get(key)
        comm = cls(ctx, rank, world, uid)
        comm._store = store                      # keep the server alive until every rank has read the id
        return comm

    def allreduce_sum(self, v):
        """in-place sum over the ranks of a float64 host array"""
        from . import _lib as L
        v = np.ascontiguousarray(v, dtype=np.float64)
        L.check(L.lib().mhx_comm_allreduce_sum(self.h, v.ctypes.data_as(C.POINTER(C.c_double)), v.size))
        return v

    def rank_world(self):
        """(rank, world) as the communicator itself reports them (mhx_comm_rank: ncclCommUserRank / ncclCommCount)"""
        from . import _lib as L
        r, w = C.c_int(), C.c_int()
        L.check(L.lib().mhx_comm_rank(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def slice(self, cnt):
        from . import _lib as L
        b, c = C.c_int(), C.c_int()
        L.check(L.lib().mhx_comm_slice(self.h, int(cnt), C.byref(b), C.byref(c)))
        return b.value, c.value

    def allgather_walkers(self, run, half):
        from . import _lib as L
        L.check(L.lib().mhx_comm_allgather_walkers(self.h, run.h, int(half)))

    def close(self):
        if self.h:
            from . import _lib as L
            L.lib().mhx_comm_destroy(self.h)
            self.h = C.c_void_p()


def allreduce_stats(diag, accepted, transitions, comm=None, group=None):
    """All-reduce the per-shard sums and return the global diagnostics (same on every rank).  `comm`: a Comm (RCCL
    through the C ABI); without one: torch.distributed on host tensors (gloo, the CPU tests)."""
    from .api import combine_diagnostics
    v = pack_stats(diag, accepted, transitions)
    if comm is not None:
        v = comm.allreduce_sum(v)
    else:
        import torch
        import torch.distributed as dist
        buf = torch.from_numpy(v)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        v = buf.numpy()
    d1 = (v.size - 3) // 3
    sm, sm2, sv = v[:d1], v[d1:2 * d1], v[2 * d1:3 * d1]
    acc, tr, nch = v[3 * d1], v[3 * d1 + 1], int(round(v[3 * d1 + 2]))
    out = dict(sum_m=sm, sum_m2=sm2, sum_v=sv, n_chains=nch, n_samples=diag["n_samples"],
               accepted=acc, transitions=tr, acceptance_rate=acc / tr if tr else float("nan"))
    out.update(combine_diagnostics(sm, sm2, sv, nch, diag["n_samples"]))
    return out


class ShardedEnsemble:
    """ONE stretch-move ensemble moved by several GPUs (src/emcee.jl:14-24 in its parallel
    half-split form).  Every rank holds the whole ensemble in the walker-major device state of its own Run (same
    seed, same ensemble id, same initial walkers on every rank); per half-step each rank moves its contiguous slice
    of the moving half (mhx_emcee_half_step) and the slices are exchanged with ONE all-gather of a packed staging buffer
    (mhx_comm_allgather_walkers: walker rows, lp, accept bookkeeping; RCCL over xGMI; stream-ordered, no host
    synchronisation inside a sweep).  Walkers carry their global index in the RNG counter, so the sharded run is the
    single-GPU run bit for bit.

    exchange="rccl" (needs `comm`), or None: the same slicing with no collective (all slices on this device, one after
    the other) -- the single-GPU emulation the parity test uses.  exchange="torch" drives `_all_gather` below with
    torch.distributed on the tensors handed to `.tensors(...)` (the gloo CPU test of the slicing / padding logic; sweep()
    refuses to run in this mode without them)."""

    def __init__(self, run, rank=0, world=1, group=None, exchange="rccl", comm=None):
        self.run, self.group = run, group
        self.exchange, self.comm = exchange, comm
        if comm is not None:
            rank, world = comm.rank, comm.world
        self.rank, self.world = int(rank), int(world)
        if exchange not in (None, "rccl", "torch"):
            raise ValueError("exchange must be None (emulation on one device), 'rccl' or 'torch'")
        if exchange == "rccl" and comm is None:
            raise ValueError("exchange='rccl' needs a Comm")
        self.W = run.n
        self._t = None

    def tensors(self, xw, lp, acc, last):
        """exchange='torch': the whole-ensemble tensors (walker-major rows, lp, accept counts, last accept flags) that
        `_all_gather` exchanges in place -- views of the run's device state, or CPU stand-ins in the gloo test."""
        self._t = dict(xw=xw, lp=lp, acc=acc, last=last)
        return self

    @staticmethod
    def slices(count, world):
        """The slice of rank q of a half with `count` walkers is [count q / world, count (q+1) / world) -- the rule of
        mhx_comm_slice: contiguous, sizes differ by at most one.  [(begin, count)] per rank."""
        return [(count * q // world, count * (q + 1) // world - count * q // world) for q in range(world)]

    def sweep(self, n=1):
        from . import _lib as L
        lib = L.lib()
        W, halfW = self.W, self.W // 2
        for _ in range(int(n)):
            for h in (0, 1):
                lo, cnt = (halfW, W - halfW) if h else (0, halfW)
                sl = self.slices(cnt, self.world)
                if self.exchange is None:                    # emulation: every slice on this device, in turn
                    for b, c in sl:
                        L.check(lib.mhx_emcee_half_step(self.run.h, h, b, c))
                    continue
                b, c = sl[self.rank]
                L.check(lib.mhx_emcee_half_step(self.run.h, h, b, c))
                if self.exchange == "rccl":
                    self.comm.allgather_walkers(self.run, h)
                else:                                        # "torch": torch.distributed collectives on the caller's tensors
                    if self._t is None:
                        raise RuntimeError("ShardedEnsemble(exchange='torch'): set .tensors(xw, lp, acc, last) first -- torch views "
                                           "of the arrays mhx_emcee_device_state exposes (the half-step must have completed on them)")
                    self._all_gather(lo, cnt, sl)
            L.check(lib.mhx_emcee_end_sweep(self.run.h))

    def _all_gather(self, lo, cnt, sl):
        """torch.distributed form of the exchange on the tensors in self._t (CPU / gloo test of the slicing logic):
        every slice padded to the longest, one all_gather per array, copied back slice by slice."""
        import torch
        import torch.distributed as dist
        per = max(c for _, c in sl)
        t = self._t
        for name in ("xw", "lp", "acc", "last"):
            full = t[name][lo:lo + cnt]
            b, c = sl[self.rank]
            piece = torch.zeros((per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
            piece[:c] = full[b:b + c]
            out = torch.empty((per * self.world,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
            dist.all_gather_into_tensor(out, piece, group=self.group)
            for q, (bq, cq) in enumerate(sl):
                full[bq:bq + cq] = out[q * per:q * per + cq]
