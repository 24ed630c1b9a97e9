#!/usr/bin/env python3
"""RWMH with a user log-density in HIP source (JIT-lowered, evaluated per lane): independent shifted Gaussians, 65 536 chains,
save-all launches of 50 transitions.  DIMS="50 80 100 128", MHX_DTYPE=f32|f64, MHX_GEN=ziggurat (fp64: the register kernel's ziggurat form).  Which kernel runs it and how fast."""
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

C = int(os.environ.get("C", 65536))
for d in [int(x) for x in os.environ.get("DIMS", "50 80 100 128").split()]:
    data = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    s = float(np.float32(2.38 / d ** 0.5))
    run = mhx.Run(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=1, normal_gen=os.environ.get("MHX_GEN") or None)
    run.init(np.zeros(d))
    run.sample(50, 0, 1, 0, save=True)
    run.sample(50, 0, 1, 0, save=True)
    st = run.stats()
    print(json.dumps(dict(config="RWMH user target d=%d C=%d %s save-all" % (d, C, st["dtype"]), lanes=st["reduce_lanes"],
                          steps_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3), acc=st["accepted"] / st["transitions"],
                          variant=st["kernel_variant"])), flush=True)
    run.close()
