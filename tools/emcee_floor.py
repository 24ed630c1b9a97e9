"""emcee half-step time against ensemble size and lanes per walker (C3 target): the launch floor and the throughput regime."""
import sys, os, json
sys.path.insert(0, "advancedmh.jl_amd"); sys.path.insert(0, "tests")
import numpy as np, mhx
d = 50
i = np.arange(d)
Sig = 0.9 ** np.abs(i[:, None] - i[None, :])
LANES = [int(v) for v in os.environ.get("LANES", "0").split(",")]
for W, L in [(w, l) for w in (256, 2048, 16384, 65536, 262144) for l in LANES]:
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3, reduce_lanes=L)
    run.init(None)
    run.sample(1, 50, 1, 0, save=False)
    run.sample(1, 500, 1, 0, save=False)
    st = run.stats()
    print(W, "walkers: %.2f us per half-step, %.3e moves/s, lanes %d" % (st["kernel_ms"] * 1e3 / 1000, st["transitions"] / (st["kernel_ms"] * 1e-3), st["reduce_lanes"]), flush=True)
    run.close()
