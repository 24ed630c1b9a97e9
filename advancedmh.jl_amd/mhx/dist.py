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
