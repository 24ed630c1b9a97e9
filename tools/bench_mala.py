import os, sys, json
sys.path.insert(0, "/root/repo/advancedmh.jl_amd")
import numpy as np, mhx
d, C = 100, 65536
model = mhx.DensityModel(mhx.IsoGaussian(d))
run = mhx.Run(model, mhx.MALA(0.3), nchains=C, seed=1)
run.init(np.zeros(d))
run.sample(1, 20, 1, 0, save=False)
run.sample(1, 100, 1, 0, save=False)
st = run.stats()
print(json.dumps(dict(config="MALA d=100 C=65536 iso", steps_per_s=st["transitions"]/(st["kernel_ms"]*1e-3), acc=st["accepted"]/st["transitions"])))
