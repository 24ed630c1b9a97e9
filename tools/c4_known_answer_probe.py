import sys, time
sys.path.insert(0, "advancedmh.jl_amd"); sys.path.insert(0, ".")
import numpy as np, mhx, bench
d, C = 200, 32768
Sig = bench.sigma_illcond(d)
run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RobustAdaptiveMetropolis(S=(2.38 / d ** 0.5) * np.eye(d)), nchains=C, seed=4)
L = np.linalg.cholesky(Sig)
run.init(L @ np.random.default_rng(11).normal(size=(d, C)))
tot = 0
for n in (200, 300, 500, 1000, 2000, 4000):
    t0 = time.time()
    run.sample(1, n, 1, n, save=False)
    tot += n
    st = run.stats(); ad = run.adapt_state()
    la = ad["logα"] if "logα" in ad else ad["logalpha"]
    dmin, dmax = run.diag_range()
    print("after %5d steps: acceptance of the last call %.4f, mean exp(logalpha) of the last step %.4f, eta %.4g, diag(S) in [%.3g, %.3g], %.1f s" % (
        tot, st["accepted"] / st["transitions"], float(np.exp(la.astype(np.float64)).mean()), ad["η"], float(np.min(dmin)), float(np.max(dmax)), time.time() - t0), flush=True)
S, status = run.factor()
print("status ok:", (status == 0).all())
