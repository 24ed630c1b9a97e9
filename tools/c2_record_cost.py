import sys
sys.path.insert(0, "advancedmh.jl_amd"); sys.path.insert(0, ".")
import numpy as np, mhx
d, C, inner = 100, 65536, 250
s = float(np.float32(2.38 / d ** 0.5))
for save in (True, False, "moments"):
    run = mhx.Run(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=0xC0FFEE, normal_gen="ziggurat")
    run.init(None)
    best = 1e9
    for it in range(40):
        if save == "moments":
            run.sample(inner, 1, 1, 0, save="moments")
        else:
            run.sample(inner, 1, 1, 0, save=save)
        if it >= 30:
            best = min(best, run.stats()["kernel_ms"])
    print("save=%s: %.3f ms per %d-step launch, %.4g steps/s, variant %d" % (save, best, inner, C * inner / (best * 1e-3), run.stats()["kernel_variant"]), flush=True)
    run.close()
