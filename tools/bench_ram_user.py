#!/usr/bin/env python3
"""RobustAdaptiveMetropolis with a user log-density in HIP source vs the same target from the catalogue: independent Gaussians,
32 768 chains, adapting.  DIMS="2 20 50 100 200", MHX_DTYPE."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mhx  # noqa: E402
import user_targets  # noqa: E402

C = int(os.environ.get("C", 32768))
for d in [int(x) for x in os.environ.get("DIMS", "2 20 50 100 200").split()]:
    data = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
    for name, model in (("user", mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))),
                        ("catalogue", mhx.DensityModel(mhx.IsoGaussian(d)))):
        run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(), nchains=C, seed=1)
        run.init(np.zeros(d))
        n = max(10, 4000 // d)
        run.sample(1, n, 1, 2 * n + 2, save=False)
        run.sample(1, n, 1, 2 * n + 2, save=False)
        st = run.stats()
        print(json.dumps(dict(config="RAM %s target d=%d C=%d %s adapting" % (name, d, C, st["dtype"]), steps_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3),
                              acc=st["accepted"] / st["transitions"], variant=st["kernel_variant"], lanes=st["reduce_lanes"])), flush=True)
        run.close()
