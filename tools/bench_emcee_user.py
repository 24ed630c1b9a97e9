#!/usr/bin/env python3
"""emcee with a user log-density in HIP source (lane-per-walker kernel): independent Gaussians.  DIMS="10 50 100", SIZES="16384 262144",
MHX_DTYPE=f32|f64, MHX_EMCEE_FUSED=0|1."""
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
import user_targets  # noqa: E402

for d in [int(x) for x in os.environ.get("DIMS", os.environ.get("D", "50")).split()]:
    for W in [int(x) for x in os.environ.get("SIZES", "16384 262144").split()]:
        data = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
        model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
        run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3)
        run.init(None)
        run.sample(1, 20, 1, 0, save=False)
        run.sample(1, 200, 1, 0, save=False)
        st = run.stats()
        print(json.dumps(dict(config="emcee user target d=%d W=%d %s" % (d, W, st["dtype"]), moves_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3),
                              us_per_sweep=st["kernel_ms"] * 1e3 / 200, launches_per_sweep=st["launches"] / 200.0, variant=st["kernel_variant"])), flush=True)
        run.close()
