#!/usr/bin/env python3
"""Throughput of RWMH with ONE dense factor for all chains (dense Gaussian target and / or dense proposal): the
matrix-core kernel (default) against the vector cooperative kernel (MHX_NO_MFMA=1 in the environment).
DIMS="32 50 64 100 128 200", C chains, MHX_DTYPE=f32|f64 picks the engine; PROPS="iso dense"."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mhx  # noqa: E402
import _opts
_opts.bridge(mhx)                  # MHX_* variables of the command line -> explicit engine options (tools only)
import cases  # noqa: E402

C = int(os.environ.get("C", 65536))
for d in [int(v) for v in os.environ.get("DIMS", "32 50 64 100 128 200").split()]:
    Sig = cases.sigma_ar1(d, 0.6)
    for prop in os.environ.get("PROPS", "iso").split():
        model = mhx.DensityModel(mhx.CorrGaussian(Sig))
        s = float(np.float32(1.2 / d ** 0.5))
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I if prop == "iso" else (s * s) * cases.sigma_ar1(d, 0.4)))
        run = mhx.Run(model, spl, nchains=C, seed=1, reduce_lanes=int(os.environ.get("LANES", 0)))
        run.init(np.zeros(d))
        run.sample(1, 20, 1, 0, save=False)
        run.sample(1, 100, 1, 0, save=False)
        st = run.stats()
        rate = st["transitions"] / (st["kernel_ms"] * 1e-3)
        print(json.dumps(dict(config="dense target d=%d C=%d %s proposal %s" % (d, C, prop, st["dtype"]), variant=st["kernel_variant"],
                              lanes=st["reduce_lanes"], steps_per_s=rate, tflops=rate * d * (d + 1) * (2 if prop == "dense" else 1) / 1e12,
                              acc=st["accepted"] / st["transitions"])), flush=True)
        run.close()
    if os.environ.get("MALA", "1") != "0":
        run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.MALA(float(np.float32(0.5 / d ** (1 / 3)))), nchains=C, seed=1,
                      reduce_lanes=int(os.environ.get("LANES", 0)))
        run.init(np.zeros(d))
        run.sample(1, 10, 1, 0, save=False)
        run.sample(1, 50, 1, 0, save=False)
        st = run.stats()
        rate = st["transitions"] / (st["kernel_ms"] * 1e-3)
        print(json.dumps(dict(config="MALA dense target d=%d C=%d %s" % (d, C, st["dtype"]), variant=st["kernel_variant"],
                              lanes=st["reduce_lanes"], steps_per_s=rate, tflops=rate * 2 * d * (d + 1) / 1e12,
                              acc=st["accepted"] / st["transitions"])), flush=True)
        run.close()
