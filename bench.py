#!/usr/bin/env python3
"""bench.py -- headline benchmark: MH steps/sec (all chains) + ESS/sec on the isotropic 100-dim
Gaussian, RWMH, 65 536 chains per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: ONE mhx_run_sample launch that advances all
65 536 chains of this rank by `--inner` (default 250) Metropolis-Hastings transitions and records
every state (the save-all semantics of the reference's `sample`) into the HBM-resident sample
tensor [inner][d+1][chains].  Inputs (chain state) and outputs (samples) stay in HBM; nothing
crosses PCIe inside the timed region.  Chains are sharded over ranks by global chain id
(first_chain = rank * chains), no data-path collective; scaling is weak.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against HBM with the
algorithmic bytes of DESIGN.md section 7; `cpu_baseline` is the CPU oracle (a port of the
reference algorithm, oracle/) timed on this host on a bounded sample -- rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
VALU_PEAK = 256 * 4 * 2.4e9 / 4.0   # wave64 VALU instructions per second: 16-lane SIMDs, 4 cycles per instruction
D = 100
CHAINS = 65536
VARIANTS = {0: "generic", 1: "prebuilt-register", 2: "hiprtc-register", 3: "prebuilt-cooperative",
            4: "hiprtc-cooperative", 5: "hiprtc-dense-cooperative"}


def algorithmic_bytes_per_launch(d, chains, inner):
    """sample record 4(d+1)+1 B per chain-step (save-all) + one state
    round trip per launch (read x, lp, accept count, last flag; write them back)."""
    record = 4 * (d + 1) + 1
    state = 2 * (4 * d + 4 + 4 + 1)
    return chains * (inner * record + state)


def cpu_baseline(d, inner, seed, target_seconds=10.0):
    """The oracle (same algorithm, same Philox streams, scalar loop per chain) on all host cores:
    chains statically partitioned over threads (the MCMCThreads analogue).  Bounded sample."""
    import concurrent.futures as cf
    import numpy as np
    from oracle import oracle as O
    O.build()
    s = float(np.float32(2.38 / d ** 0.5))
    cores = os.cpu_count() or 1
    tgt = O.iso_gauss(d)

    def work(first, n, steps):
        O.rwmh(tgt, O.Proposal(O.PROP_ISO, s), O.schedule(steps + 1), seed, first, n, save=True)

    def pool(per_thread, steps):
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda i: work(i * per_thread, per_thread, steps), range(cores)))
        return time.perf_counter() - t0

    t0 = time.perf_counter()
    work(0, 8, 100)                                   # single-thread rate (the `sample(model, spl, N)` analogue)
    rate1 = 800 / (time.perf_counter() - t0)
    rate_all = cores * 8 * 100 / pool(8, 100)         # calibration on all cores
    per_thread = max(1, int(rate_all * target_seconds / (inner * cores)))
    dt = pool(per_thread, inner)
    total = cores * per_thread * inner
    return {"value": total / dt, "unit": "MH steps/s", "cores": cores, "kind": "port",
            "sample": "%d chains x %d transitions of the same d=%d workload (oracle/mhx_oracle.c, %d threads, %.1f s); "
                      "single-thread %.3g steps/s" % (cores * per_thread, inner, d, cores, dt, rate1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5, help="untimed launches right before the timed ones")
    ap.add_argument("--inner", type=int, default=250, help="MH transitions per chain per step (launch)")
    ap.add_argument("--chains", type=int, default=CHAINS, help="chains per GPU")
    ap.add_argument("--dim", type=int, default=D)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per chain (0 = engine's choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("MHX_BENCH_FORCE_DIST"):      # the env knob exercises the RCCL path on 1 GPU
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import numpy as np
    import mhx

    d, C, inner = args.dim, args.chains, args.inner
    s = float(np.float32(2.38 / d ** 0.5))
    ctx = mhx.Context(local_rank)
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
    run = mhx.Run(model, spl, nchains=C, seed=0xC0FFEE, first_chain=rank * C, ctx=ctx, reduce_lanes=args.lanes)
    run.init(None)                                    # x0 ~ proposal draw (src/mh-core.jl:83), on the device

    def step():
        # N=inner saved samples, the first one being the state after 1 transition: inner transitions
        run.sample(inner, 1, 1, 0, save=True)
        return run.stats()

    def sync():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    # device spin-up (setup, like init): the first ~20 launches after an idle period run below the steady clock
    # (8.0e9 vs 9.2e9 steps/s); bring the GPU there whatever --warmup the caller picked
    for _ in range(max(0, 30 - args.warmup)):
        step()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    kernel_ms, accepted, transitions = 0.0, 0, 0
    for _ in range(args.steps):
        st = step()                                   # blocking: the stream is synchronised on return
        kernel_ms += st["kernel_ms"]
        accepted += st["accepted"]
        transitions += st["transitions"]
    sync()
    dt = time.perf_counter() - t0
    variant = st["kernel_variant"]

    # Diagnostics of the LAST launch's sample tensor (outside the timed region): per-shard sums for R-hat
    # and the between-chain ESS; across GPUs they combine with the one small all-reduce the design has.
    diag = run.diagnostics(max_lag=0)
    bulk = None
    if world == 1:
        try:                                          # rank-normalised bulk ESS of three parameters (device sort), for reference
            b = run.ess_bulk_tail(params=[0, d // 2, d - 1], split=True)
            bulk = {"params": [int(v) for v in b["params"]], "ess_bulk": [float(v) for v in b["ess_bulk"]],
                    "upper_bound_only": [bool(v) for v in b["bulk_truncated"]],
                    "note": "split chains, Geyer truncation; 250 draws per chain are shorter than the autocorrelation time, "
                            "so the sequence is still positive at the last lag and the value is an upper bound -- "
                            "ess_per_sec uses the between-chain estimator"}
        except Exception as e:                        # never let a diagnostic break the bench line
            bulk = {"error": str(e)}
    if dist is not None:
        import torch
        from mhx.dist import allreduce_stats
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        diag = allreduce_stats(diag, accepted, transitions, device=torch.device("cuda", local_rank))   # RCCL over xGMI
        acc_rate = diag["acceptance_rate"]
    else:
        acc_rate = accepted / float(transitions)

    if rank == 0:
        total_steps = float(C) * inner * args.steps * world
        value = total_steps / dt
        launch_s = kernel_ms * 1e-3 / args.steps
        bytes_launch = algorithmic_bytes_per_launch(d, C, inner)
        achieved = bytes_launch / launch_s / 1e9
        traffic, valu = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get("rwmh_d%d_c%d_inner%d" % (d, C, inner), {})
                traffic = tj.get("hbm_bytes_per_launch")
                if tj.get("valu_insts_per_launch"):
                    # what actually bounds the kernel: wave-instructions (PMC SQ_INSTS_VALU) per second against
                    # 256 CUs x 4 SIMDs x one wave64 VALU instruction per 4 cycles at 2.4 GHz
                    rate = tj["valu_insts_per_launch"] / launch_s
                    valu = {"wave_insts_per_launch": tj["valu_insts_per_launch"], "achieved_per_s": rate,
                            "peak_per_s": VALU_PEAK, "frac": rate / VALU_PEAK}
            except Exception:
                traffic, valu = None, None
        essb = np.asarray(diag["ess_between"][:d], dtype=np.float64)
        out = {
            "metric": "MH steps/sec (all chains) + ESS/sec", "value": value, "unit": "MH steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RWMH, isotropic %d-dim standard MvNormal target, %d chains per GPU, proposal "
                                   "N(0,(2.38/sqrt(d))^2 I), %d transitions per launch, every state recorded" % (d, C, inner),
                       "chains_per_gpu": C, "dim": d, "transitions_per_step": inner,
                       "kernel_variant": VARIANTS[variant], "lanes_per_chain": st["reduce_lanes"],
                       "sharding": "chains by global id, no data-path collective"},
            "acceptance_rate": acc_rate,
            # ESS/sec: total effective sample size of ONE step's draws (median over the d parameters; between-chain
            # estimator C * var+ / Var_c(chain means), include/mhx.h) divided by the wall time of one step
            "ess_per_sec": float(np.median(essb)) / (dt / args.steps),
            "ess": {"estimator": "between-chain, last step's %d draws x %d chains" % (inner, C * world),
                    "median": float(np.median(essb)), "min": float(essb.min()),
                    "rhat_max": float(np.max(diag["rhat"][:d])), "bulk": bulk},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_rwmh_coop<2,13,iso,iso>" if variant == 3 else "rwmh variant %d" % variant,
                         "avg_launch_ms": launch_s * 1e3, "algorithmic_bytes_per_launch": bytes_launch,
                         "valu": valu,
                         "note": "VALU-issue-bound kernel (Philox4x32-10 + Box-Muller polynomials, 64 wave-instructions "
                                 "per chain-step, PMC SQ_INSTS_VALU); HBM sees only the sample records"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, inner, 0xC0FFEE)
        # RCCL writes its version banner to the C stdout buffer: push it out first so the JSON line is the last one
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
