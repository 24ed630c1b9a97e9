#!/usr/bin/env python3
"""Throughput of the walks with a Hastings ratio and of the state-in-HBM kernels: d = 100, 65 536 chains, isotropic target.
MHX_DTYPE=f32|f64 picks the engine."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
import mhx  # noqa: E402

d, C = int(os.environ.get("D", 100)), int(os.environ.get("C", 65536))
model = mhx.DensityModel(mhx.IsoGaussian(d))
s = float(np.float32(2.38 / d ** 0.5))
mu = np.full(d, 0.01)
for name, spl, flags in (("generic RWMH", mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), mhx.FLAG_GENERIC),
                         ("drifting RWMH (cooperative)", mhx.RWMH(mhx.MvNormal(mu, s * s * mhx.I)), 0),
                         ("drifting RWMH (state in HBM)", mhx.RWMH(mhx.MvNormal(mu, s * s * mhx.I)), mhx.FLAG_GENERIC),
                         ("static MH (cooperative)", mhx.StaticMH(mhx.MvNormal(mhx.zeros(d), mhx.I)), 0),
                         ("static MH (state in HBM)", mhx.StaticMH(mhx.MvNormal(mhx.zeros(d), mhx.I)), mhx.FLAG_GENERIC),
                         ("MALA", mhx.MALA(0.3), 0)):
    run = mhx.Run(model, spl, nchains=C, seed=1, flags=flags)
    run.init(np.zeros(d))
    run.sample(1, 20, 1, 0, save=False)
    run.sample(1, 100, 1, 0, save=False)
    st = run.stats()
    print(json.dumps(dict(config="%s d=%d C=%d iso %s" % (name, d, C, st["dtype"]), lanes=st["reduce_lanes"], steps_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3),
                          acc=st["accepted"] / st["transitions"], variant=st["kernel_variant"])), flush=True)
    run.close()
