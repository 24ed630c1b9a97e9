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
