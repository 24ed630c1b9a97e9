import sys, time
sys.path.insert(0, "advancedmh.jl_amd"); sys.path.insert(0, "tests")
import numpy as np, mhx, cases
for dt in ("f64", "f32"):
    for d in (256, 330, 512) if dt == "f64" else (512, 1000):
        C = 65536
        Sig = cases.sigma_ar1(d, 0.6)
        t0 = time.time()
        run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.MALA(0.3 / d ** (1 / 3)), nchains=C, seed=1, dtype=dt)
        tc = time.time() - t0
        run.init(np.zeros(d))
        run.sample(1, 5, 1, 0, save=False)
        run.sample(1, 20, 1, 0, save=False)
        st = run.stats()
        print(dt, "d", d, "variant", st["kernel_variant"], "%.3e steps/s" % (st["transitions"] / (st["kernel_ms"] * 1e-3)), "acc %.3f" % (st["accepted"] / st["transitions"]), "create %.1f s" % tc, flush=True)
        run.close()
