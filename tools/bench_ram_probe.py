#!/usr/bin/env python3
"""RAM throughput probe: the same d / chain count as C4 with targets of different memory appetite
(dense Gaussian = one extra 80 KB L2 stream per step; isotropic = none)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
import mhx  # noqa: E402

d = int(os.environ.get("PROBE_D", 200))
C = int(os.environ.get("PROBE_CHAINS", 32768))
for name in sys.argv[1:] or ["iso", "corr"]:
    if name == "iso":
        model = mhx.DensityModel(mhx.IsoGaussian(d))
    else:
        i = np.arange(d)
        model = mhx.DensityModel(mhx.CorrGaussian(0.5 ** np.abs(i[:, None] - i[None, :])))
    run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(), nchains=C, seed=4)
    run.init(np.zeros(d))
    for phase, (n, nad) in (("adapting", (100, 100)), ("fixed", (100, 0))):
        run.sample(1, n, 1, nad, save=False)
        st = run.stats()
        tri = d * (d + 1) // 2 * (8 if mhx.get_default_dtype() == "f64" else 4)
        per = (2 * tri if nad else tri)
        print(json.dumps(dict(target=name, phase=phase, steps_per_s=st["transitions"] / (st["kernel_ms"] * 1e-3),
                              kernel_ms=st["kernel_ms"], acceptance=st["accepted"] / st["transitions"],
                              hbm_GBs=per * st["transitions"] / (st["kernel_ms"] * 1e-3) / 1e9)), flush=True)
    run.close()
