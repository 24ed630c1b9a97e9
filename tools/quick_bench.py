#!/usr/bin/env python3
"""Quick GPU throughput probe of the RWMH kernels (not the contract bench -- see bench.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
import mhx  # noqa: E402

d = int(os.environ.get("D", 100))
C = int(os.environ.get("C", 65536))
N = int(os.environ.get("N", 250))
lanes = [int(v) for v in os.environ.get("LANES", "1,2,4,0").split(",")]
s = float(np.float32(2.38 / d ** 0.5))
model = mhx.DensityModel(mhx.IsoGaussian(d))
spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
for L in lanes:
    run = mhx.Run(model, spl, nchains=C, seed=1, reduce_lanes=L)
    run.init(None)
    run.sample(20, 0, 1, 0, save=True)          # warm-up
    for save in (True, False):
        best = None
        for rep in range(3):
            if save:
                run.sample(N, 1, 1, 0, save=True)
            else:
                run.sample(1, N, 1, 0, save=False)
            st = run.stats()
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best = st
        steps = best["transitions"]
        print("d=%d C=%d lanes=%d(req %d) save=%-5s variant=%d  %.3e steps/s (kernel %.2f ms) acc=%.3f" % (
            d, C, best["reduce_lanes"], L, save, best["kernel_variant"], steps / (best["kernel_ms"] * 1e-3),
            best["kernel_ms"], best["accepted"] / steps), flush=True)
    run.close()
