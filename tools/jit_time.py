import sys, time, os
sys.path.insert(0, "advancedmh.jl_amd"); sys.path.insert(0, "tests")
import numpy as np, mhx, cases
for dt, d in [(a, int(b)) for a, b in (c.split(":") for c in os.environ.get("CASES", "f32:100 f32:400 f64:100 f64:256").split())]:
    mhx.set_default_dtype(dt)
    Sig = cases.sigma_ar1(d, 0.5)
    t0 = time.time()
    run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.01 * mhx.I)), nchains=4096, seed=1)
    t1 = time.time()
    run.init(np.zeros(d)); run.sample(1, 5, 1, 0, save=False)
    print(dt, d, "create (hiprtc) %.1f s, variant %d" % (t1 - t0, run.stats()["kernel_variant"]), flush=True)
    run.close()
