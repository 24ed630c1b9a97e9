"""Multi-GPU plumbing: chains shard by GLOBAL chain id (no data-path collective); the only
collective is ONE all-reduce of 3(dim+1)+2 doubles per reporting interval -- acceptance totals and
the R-hat / ESS sums.  torch.distributed is used as the transport: backend
"nccl" is RCCL over xGMI on MI355X, "gloo" on CPU for the tests."""
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


def allreduce_stats(diag, accepted, transitions, device=None, group=None):
    """All-reduce the per-shard sums and return the global diagnostics (same on every rank)."""
    import torch
    import torch.distributed as dist
    from .api import combine_diagnostics
    buf = torch.from_numpy(pack_stats(diag, accepted, transitions))
    if device is not None:
        buf = buf.to(device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    v = buf.cpu().numpy()
    d1 = (v.size - 3) // 3
    sm, sm2, sv = v[:d1], v[d1:2 * d1], v[2 * d1:3 * d1]
    acc, tr, nch = v[3 * d1], v[3 * d1 + 1], int(round(v[3 * d1 + 2]))
    out = dict(sum_m=sm, sum_m2=sm2, sum_v=sv, n_chains=nch, n_samples=diag["n_samples"],
               accepted=acc, transitions=tr, acceptance_rate=acc / tr if tr else float("nan"))
    out.update(combine_diagnostics(sm, sm2, sv, nch, diag["n_samples"]))
    return out


class _DevArray:
    """A view of device memory owned by libmhx that torch can wrap without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(int(v) for v in shape), typestr=typestr,
                                             data=(int(ptr), False), version=2)


class ShardedEnsemble:
    """ONE stretch-move ensemble moved by several GPUs (src/emcee.jl:14-24 in its parallel
    half-split form).  Every rank holds the whole ensemble in the walker-major device state of its own Run (same
    seed, same ensemble id, same initial walkers on every rank); per half-step each rank moves its contiguous slice
    of the moving half (mhx_emcee_half_step) and the slices are exchanged with an all-gather -- RCCL over xGMI with
    backend "nccl" -- of the walker rows, lp and the accept bookkeeping.  Walkers carry their global index in the
    RNG counter, so the sharded run is the single-GPU run bit for bit.

    `exchange=None` runs the same slicing with no collective (all slices on this device, one after the other):
    the single-GPU emulation the parity test uses."""

    def __init__(self, run, rank=0, world=1, group=None, exchange="torch"):
        import ctypes as C
        from . import _lib as L
        self.run, self.rank, self.world, self.group = run, int(rank), int(world), group
        self.exchange = exchange
        fp, u32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        xw, lp, acc, last, pitch = fp(), fp(), u32p(), u8p(), C.c_int32()
        L.check(L.lib().mhx_emcee_device_state(run.h, C.byref(xw), C.byref(pitch), C.byref(lp), C.byref(acc), C.byref(last)))
        self.W, self.pitch = run.n, int(pitch.value)
        self._ptr = dict(xw=C.cast(xw, C.c_void_p).value, lp=C.cast(lp, C.c_void_p).value,
                         acc=C.cast(acc, C.c_void_p).value, last=C.cast(last, C.c_void_p).value)
        self._t = None

    def _tensors(self):
        if self._t is None:
            import torch
            W, P = self.W, self.pitch
            self._t = dict(xw=torch.as_tensor(_DevArray(self._ptr["xw"], (W, P), "<f4"), device="cuda"),
                           lp=torch.as_tensor(_DevArray(self._ptr["lp"], (W,), "<f4"), device="cuda"),
                           acc=torch.as_tensor(_DevArray(self._ptr["acc"], (W,), "<i4"), device="cuda"),
                           last=torch.as_tensor(_DevArray(self._ptr["last"], (W,), "|u1"), device="cuda"))
        return self._t

    @staticmethod
    def slices(count, world):
        """Equal slices of one half (the last ranks may idle on a remainder): [(begin, count)] per rank."""
        per = (count + world - 1) // world
        return [(min(r * per, count), max(0, min(per, count - r * per))) for r in range(world)]

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
                self._all_gather(lo, cnt, sl)
            L.check(lib.mhx_emcee_end_sweep(self.run.h))

    def _all_gather(self, lo, cnt, sl):
        import torch
        import torch.distributed as dist
        per = sl[0][1]
        t = self._tensors()
        on_gpu = t["xw"].is_cuda
        if on_gpu:
            torch.cuda.synchronize()
        for name in ("xw", "lp", "acc", "last"):
            full = t[name][lo:lo + cnt]
            if per * self.world == cnt:                      # equal slices: gather straight into the state
                b, c = sl[self.rank]
                dist.all_gather_into_tensor(full, full[b:b + c].clone(), group=self.group)
            else:                                            # ragged tail: pad to equal pieces, copy back
                b, c = sl[self.rank]
                piece = torch.zeros((per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
                piece[:c] = full[b:b + c]
                out = torch.empty((per * self.world,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
                dist.all_gather_into_tensor(out, piece, group=self.group)
                full.copy_(out[:cnt])
        if on_gpu:
            torch.cuda.synchronize()
