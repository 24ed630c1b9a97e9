#!/usr/bin/env python3
"""MALA with a user log-density + gradient in HIP source: independent shifted Gaussians, 65 536 chains.  DIMS="16 24 32 50 100", MHX_DTYPE."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
import mhx  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import user_targets  # noqa: E402

SRC = user_targets.SHIFTED_GAUSS_WITH_GRADIENT
C = int(os.environ.get("C", 65536))
for d in [int(x) for x in os.environ.get("DIMS", "16 24 32 50 100").split()]:
    data = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(SRC, d, data=data))
    run = mhx.Run(model, mhx.MALA(1.0 / d ** (1.0 / 3.0)), nchains=C, seed=1)
    run.init(np.zeros(d))
    run.sample(50, 0, 1, 0, save=True)
    run.sample(50, 0, 1, 0, save=True)
    st = run.stats()
    print(json.dumps(dict(config="MALA user target d=%d C=%d %s save-all" % (d, C, st["dtype"]), steps_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3),
                          acc=st["accepted"] / st["transitions"], variant=st["kernel_variant"])), flush=True)
    run.close()
