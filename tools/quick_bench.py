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
N = int(os.environ.get("N", 200))
s = float(np.float32(2.38 / d ** 0.5))
model = mhx.DensityModel(mhx.IsoGaussian(d))
spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
for name, flags in (("auto", 0), ("generic", mhx.FLAG_GENERIC)):
    run = mhx.Run(model, spl, nchains=C, seed=1, flags=flags)
    run.init(None)
    run.sample(20, 0, 1, 0, save=True)          # warm-up
    for save, thin in ((True, 1), (False, 1)):
        t0 = time.time()
        if save:
            run.sample(N, 1, 1, 0, save=True)
        else:
            run.sample(1, N, 1, 0, save=False)
        st = run.stats()
        steps = st["transitions"]
        print("%-8s save=%-5s variant=%d  %.3e steps/s (kernel %.2f ms, wall %.2f ms) acc=%.3f" % (
            name, save, st["kernel_variant"], steps / (st["kernel_ms"] * 1e-3), st["kernel_ms"],
            (time.time() - t0) * 1e3, st["accepted"] / steps))
    run.close()
