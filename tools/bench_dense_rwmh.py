#!/usr/bin/env python3
"""RWMH on a dense (correlated) Gaussian target, 65 536 chains: the register kernel (d <= 64) and the cooperative
kernel of mhx_rwmh_dense_kernels.h (L lanes per chain, factor image in LDS)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
import mhx  # noqa: E402

C = int(os.environ.get("C", 65536))
for d, L in [tuple(int(v) for v in c.split(':')) for c in os.environ.get('CASES', '50:0,50:16,64:0,100:0,100:16,128:0').split(',')]:
    i = np.arange(d)
    model = mhx.DensityModel(mhx.CorrGaussian(0.5 ** np.abs(i[:, None] - i[None, :])))
    s = float(np.float32(1.7 / d ** 0.5))
    run = mhx.Run(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=1, reduce_lanes=L)
    run.init(np.zeros(d))
    run.sample(1, 20, 1, 0, save=False)
    for save in (False, True):
        if save:
            run.sample(100, 1, 1, 0, save=True)
        else:
            run.sample(1, 100, 1, 0, save=False)
        st = run.stats()
        print(json.dumps(dict(config="RWMH dense Gaussian d=%d C=%d save=%s" % (d, C, save), lanes=st["reduce_lanes"],
                              variant=st["kernel_variant"], steps_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3),
                              acc=st["accepted"] / st["transitions"])), flush=True)
    run.close()
