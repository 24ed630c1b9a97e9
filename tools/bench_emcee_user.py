#!/usr/bin/env python3
"""emcee with a user log-density in HIP source (lane-per-walker kernel): independent Gaussians, d = 50."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mhx  # noqa: E402
import user_targets  # noqa: E402

d = int(os.environ.get("D", 50))
for W in (16384, 262144):
    data = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3)
    run.init(None)
    run.sample(1, 20, 1, 0, save=False)
    run.sample(1, 200, 1, 0, save=False)
    st = run.stats()
    print(json.dumps(dict(config="emcee user target d=%d W=%d" % (d, W), moves_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3),
                          us_per_half_step=st["kernel_ms"] * 1e3 / 400, variant=st["kernel_variant"])), flush=True)
    run.close()
