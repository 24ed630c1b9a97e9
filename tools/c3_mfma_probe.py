#!/usr/bin/env python3
"""Timing probe of the matrix-core stretch move (tools build): what does the per-launch fetch of the factor's MFMA operands cost?
JIT_DEFS MHX_EMCEE_MFMA_PROBE=1 makes every lane fetch all its operand groups from the same 2 KB (wrong chains, right latencies)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "advancedmh.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhx
import bench

mhx.use_library(mhx.TOOLS_LIB_PATH)
cases = [(None, 1), ("MHX_EMCEE_MFMA_PROBE=1", 1)] + [(None, w) for w in (2, 4, 8)]
for defs, waves in cases:
    ctx = mhx.Context(0, "f64")
    if defs:
        ctx.set_option("JIT_DEFS", defs)
    ctx.set_option("EMCEE_MFMA_WAVES", str(waves))
    d, W = 50, 16384
    Sig = bench.sigma_ar1(d, 0.9)
    Q, _ = np.linalg.qr(np.random.default_rng(50).normal(size=(d, d)))
    Sig = Q @ Sig @ Q.T
    run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3, ctx=ctx)
    run.init(None)
    for _ in range(5):
        run.sample(500, 1, 1, 0)
    ts = []
    for _ in range(5):
        run.sample(500, 1, 1, 0)
        ts.append(run.stats()["kernel_ms"])
    st = run.stats()
    print("waves/block=%d defs=%s variant %d launches %d: %.3f us per sweep (kernel_ms %.3f)" % (waves, defs, st["kernel_variant"], st["launches"], 1e3 * np.median(ts) / 500, np.median(ts)))
    run.close()
