#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs[2..4]) on one GPU -- one JSON line each.
Not the contract bench (bench.py is); used for DESIGN.md numbers and profiles/."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mhx  # noqa: E402

which = sys.argv[1:] or ["c3", "c4", "c5"]


def sigma_ar1(d, rho):
    i = np.arange(d)
    return rho ** np.abs(i[:, None] - i[None, :])


def report(name, st, extra):
    steps = st["transitions"]
    out = dict(config=name, steps_per_s=steps / (st["kernel_ms"] * 1e-3), kernel_ms=st["kernel_ms"], wall_ms=st["wall_ms"],
               launches=st["launches"], acceptance=st["accepted"] / steps, variant=st["kernel_variant"])
    out.update(extra)
    print(json.dumps(out), flush=True)


if "c3" in which:
    d, W, sweeps = 50, 16384, 500
    model = mhx.DensityModel(mhx.CorrGaussian(sigma_ar1(d, 0.9)))
    run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3,
                  reduce_lanes=int(os.environ.get("C3_LANES", 0)))
    run.init(None)
    run.sample(20, 1, 1, 0, save=True)
    for save in (True, False):
        if save:
            run.sample(sweeps, 1, 1, 0, save=True)
        else:
            run.sample(1, sweeps, 1, 0, save=False)
        st = run.stats()
        report("C3 emcee d=50 W=16384 save=%s lanes=%d" % (save, st["reduce_lanes"]), st,
               dict(bytes_per_move=813, achieved_GBs=813 * st["transitions"] / (st["kernel_ms"] * 1e-3) / 1e9))
    run.close()

if "c4" in which:
    d, C = 200, int(os.environ.get("C4_CHAINS", 32768))
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.normal(size=(d, d)))
    lam = 1e3 ** (np.arange(d) / (d - 1.0))
    Sig = (Q * lam) @ Q.T
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(), nchains=C, seed=4)
    run.init(np.zeros(d))
    phases = (("warm-up (adapting)", (200, 200)), ("fixed S", (100, 0)))
    if os.environ.get("C4_FULL"):                   # 2 000 warm-up + 500 fixed steps
        phases = (("warm-up (adapting), 2000 steps", (2000, 2000)), ("fixed S, 500 steps", (500, 0)))
    if os.environ.get("C4_ONLY") == "adapt":
        phases = phases[:1]
    for nm, (n, warm) in phases:
        run.sample(1, n, 1, warm, save=False)
        st = run.stats()
        tri = d * (d + 1) // 2
        bps = (2 if warm else 1) * 4 * tri + 8 * d + 8      # 1R+1W of S when adapting (fused next mat-vec), 1R otherwise
        report("C4 RAM d=200 C=%d %s" % (C, nm), st,
               dict(algorithmic_bytes_per_step=bps, achieved_GBs=bps * st["transitions"] / (st["kernel_ms"] * 1e-3) / 1e9))
    S, status = run.factor()
    print(json.dumps(dict(config="C4 status", downdate_failures=int((status & 1).sum()), nan=int((status & 2).sum()))))
    run.close()

if "c5" in which:
    d, C, N = 1000, 32768, 40
    s = float(np.float32(2.38 / d ** 0.5))
    for nm, spec in (("funnel", mhx.Funnel(d)), ("banana", mhx.Banana(d, 0.03))):
        run = mhx.Run(mhx.DensityModel(spec), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=5)
        run.init(None)
        run.sample(1, 5, 1, 0, save=False)
        run.sample(1, N, 1, 0, save=False)
        st = run.stats()
        report("C5 RWMH d=1000 %s C=32768 (1 of 8 GPUs)" % nm, st,
               dict(min_bytes_per_step=8 * d + 8, achieved_GBs=(8 * d + 8) * st["transitions"] / (st["kernel_ms"] * 1e-3) / 1e9))
        run.close()
