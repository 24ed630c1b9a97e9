#!/usr/bin/env python3
"""BASELINE config 5: 1000-dim Neal's funnel / banana, RWMH, 32 768 chains PER GPU (262 144 over 8), chains sharded
by global id, no sample tensor (running moments, every 10th state), ONE RCCL all-reduce for acceptance + R-hat.

    python tools/bench_c5.py                                   # one GPU (one shard of the 8)
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_c5.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))

rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
dist = None
if world > 1 or os.environ.get("MHX_BENCH_FORCE_DIST"):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

import mhx  # noqa: E402
from mhx.dist import allreduce_stats, shard_chains  # noqa: E402

d, total = 1000, 32768 * world
first, C = shard_chains(total, rank, world)
s = float(np.float32(2.38 / d ** 0.5))
ctx = mhx.Context(local_rank)
for name, spec in (("funnel", mhx.Funnel(d)), ("banana", mhx.Banana(d, 0.03))):
    run = mhx.Run(mhx.DensityModel(spec), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=5,
                  first_chain=first, ctx=ctx)
    run.init(None)
    run.sample(1, 200, 1, 0, save=False)                     # burn-in
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
    t0 = time.perf_counter()
    run.sample(100, 10, 10, 0, save="moments")               # 1000 transitions, every 10th state folded into the moments
    st = run.stats()
    diag = run.diagnostics()
    if dist is not None:
        diag = allreduce_stats(diag, st["accepted"], st["transitions"], device=torch.device("cuda", local_rank))
        torch.cuda.synchronize()
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        acc = diag["acceptance_rate"] if dist is not None else st["accepted"] / st["transitions"]
        print(json.dumps({"config": "C5 %s d=1000, %d chains on %d GPU(s)" % (name, total, world),
                          "steps_per_s": 1000.0 * total / dt, "kernel_steps_per_s_rank0": st["transitions"] / (st["kernel_ms"] * 1e-3),
                          "acceptance": acc, "rhat_max": float(np.nanmax(diag["rhat"][:d])),
                          "rhat_median": float(np.nanmedian(diag["rhat"][:d])),
                          "lanes_per_chain": st["reduce_lanes"], "variant": st["kernel_variant"]}), flush=True)
    run.close()
if dist is not None:
    dist.destroy_process_group()
