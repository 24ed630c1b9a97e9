#!/usr/bin/env python3
"""The reference's README example (README.md:18-63) on the GPU engine: a Normal(mu, sigma) model of 30 data
points, RWMH with an identity-covariance proposal, 100 000 draws -- here from 256 chains of 400 draws after
200 discarded each, all advanced together on one MI355X -- and the MCMCChains-style summary.

    python examples/readme_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
import mhx  # noqa: E402

data = np.random.default_rng(1234).normal(0.0, 1.0, size=30)          # README.md:22  data = rand(Normal(0, 1), 30)

# README.md:25-31: insupport(theta) = theta[2] >= 0; density = sum(logpdf.(Normal(theta[1], theta[2]), data)) or -Inf
model = mhx.DensityModel(mhx.IIDNormal(data))

# README.md:36-37: spl = RWMH(MvNormal(zeros(2), I))
spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(2), mhx.I))

# README.md:40: chain = sample(model, spl, 100000; param_names=["mu", "sigma"], chain_type=Chains)
chain = mhx.sample(model, spl, 400, 256, param_names=["mu", "sigma"], chain_type=mhx.Chains, discard_initial=200,
                   initial_params=np.array([0.0, 1.0]), seed=1234)
print(chain)
print("data: mean %.4f std %.4f;  acceptance rate %.3f;  kernel variant %d" % (
    data.mean(), data.std(), chain.accepted[1:].mean(), chain.stats["kernel_variant"]))

# The same model written the way the README writes it -- a closure over `data` -- traced into kernel source (mhx/trace.py):
import math  # noqa: E402

import mhx.trace as T  # noqa: E402


def density(theta):                                                   # README.md:25-31
    mu, sigma = theta
    lp = sum(-0.5 * ((y - mu) / sigma) ** 2 - T.log(sigma) - 0.5 * math.log(2 * math.pi) for y in data)
    return T.where(sigma >= 0, lp, -math.inf)


closure_model = mhx.DensityModel(density, dim=2)
chain2 = mhx.sample(closure_model, spl, 400, 256, param_names=["mu", "sigma"], chain_type=mhx.Chains, discard_initial=200,
                    initial_params=np.array([0.0, 1.0]), seed=1234)
print("closure model (%d traced operations): mean mu %.4f sigma %.4f   [catalogue model: %.4f %.4f]" % (
    closure_model.traced.n_operations, chain2.mean("mu"), chain2.mean("sigma"), chain.mean("mu"), chain.mean("sigma")))
