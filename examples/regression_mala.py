#!/usr/bin/env python3
"""A closure over a data set, sampled with MALA: Bayesian linear regression y ~ Normal(a + b t, exp(ls)) over 5 000 rows.
The likelihood is ONE loop in the kernel source (mhx.trace.sum_over: the rows travel to the device as the log-density's data
block), its gradient is the trace's own reverse-mode sweep -- what the reference gets from ForwardDiff
(ext/AdvancedMHForwardDiffExt.jl, src/MALA.jl:54-93) -- and 4 096 chains advance together on one MI355X.

    python examples/regression_mala.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
import mhx  # noqa: E402
import mhx.trace as T  # noqa: E402

rng = np.random.default_rng(7)
t = rng.uniform(-1.0, 2.0, size=5000)
rows = np.stack([t, 0.7 - 1.3 * t + 0.3 * rng.normal(size=t.size)], axis=1)          # (t, y)


def logdensity(theta):
    a, b, ls = theta
    s = T.exp(ls)
    loglik = T.sum_over(rows, lambda r: -0.5 * ((r[1] - (a + b * r[0])) / s) ** 2 - ls)
    return loglik - 0.5 * 0.01 * (a * a + b * b)                                      # N(0, 10^2) priors on a and b


model = mhx.DensityModel(logdensity, dim=3)
print("traced: %d operations in the source, data block of %d reals" % (model.traced.n_operations, model.traced.data.size))
chain = mhx.sample(model, mhx.MALA(lambda g: mhx.MvNormal(0.5 * 4e-5 * g, 4e-5 * mhx.I)), 2000, 4096,
                   initial_params=np.zeros(3), discard_initial=1000, param_names=["a", "b", "log_s"], seed=11)
print(chain)
b_ols, a_ols = np.polyfit(rows[:, 0], rows[:, 1], 1)
print("least squares: a = %.4f, b = %.4f;  acceptance rate %.3f;  %.3g MALA steps/s (kernel)" % (
    a_ols, b_ols, chain.accepted[1:].mean(), chain.stats["transitions"] / (chain.stats["kernel_ms"] * 1e-3)))
