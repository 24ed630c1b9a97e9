"""Log-densities written as plain Python callables (what the reference's `DensityModel(f)` takes, src/AdvancedMH.jl:52-54),
shared by the CPU tests of the tracer and the GPU parity tests.  Every function works on floats and on traced parameters."""
import math

import numpy as np

import mhx.trace as T

LOG2PI = math.log(2 * math.pi)

# README.md:26-38 of the reference: 30 draws from Normal(0, 1); density(theta) = insupport ? sum(logpdf.(Normal(mu, sigma), data)) : -Inf
README_DATA = np.random.default_rng(1234).normal(size=30)


def readme_density(theta):
    mu, sigma = theta
    ls = T.log(sigma)
    lp = 0
    for y in README_DATA:
        z = (y - mu) / sigma
        lp = lp + (-0.5 * (z * z) - ls - 0.5 * LOG2PI)
    return T.where(sigma >= 0, lp, -math.inf)


def nig(theta):
    """test/emcee.jl:5-14: s ~ InverseGamma(2, 3), m ~ Normal(0, sqrt s), 1.5 and 2.0 ~ Normal(m, sqrt s)."""
    s, m = theta
    lp = 2 * math.log(3) - 3 * T.log(s) - 3 / s
    for y in (0.0, 1.5, 2.0):
        lp = lp - 0.5 * (LOG2PI + T.log(s)) - 0.5 * (y - m) ** 2 / s
    return T.where(s > 0, lp, -math.inf)


_PREC = np.linalg.inv(np.array([[1.5, 0.35], [0.35, 1.0]]))


def quadratic(x):
    """test/runtests.jl:334-365 (issue #95): lp = -x' A x / 2 with numpy in the closure."""
    return -0.5 * (x @ (_PREC @ x))


def rosenbrock_like(x):
    """abs, sqrt, fma, minimum and a reused subexpression: exercises every traced operation."""
    a = T.sqrt(T.abs(x[0]) + 1.0)
    b = T.fma(x[1], x[1], a)
    c = T.exp(-0.1 * b) + T.minimum(x[2], 3.0) * 0.01
    return -(b * b) * 0.05 - 0.5 * (x[2] - a) ** 2 + T.log(c + 2.0) - T.maximum(x[0], -50.0) ** 2 * 0.125


# the same likelihood over 2000 points as ONE loop in the kernel source (T.sum_over) instead of 2000 copies of the body
BIG_DATA = np.random.default_rng(77).normal(loc=0.4, scale=1.7, size=2000)


def big_data_density(theta):
    mu, sigma = theta
    ls = T.log(sigma)                                       # row-independent: computed once, outside the loop
    lp = T.sum_over(BIG_DATA, lambda y: -0.5 * ((y - mu) / sigma) ** 2 - ls - 0.5 * LOG2PI)
    return T.where(sigma > 0, lp, -math.inf)


# a regression with two columns per row: y ~ Normal(a + b t, exp(ls))
_T = np.linspace(-1.0, 2.0, 150)
REG_DATA = np.stack([_T, 0.7 - 1.3 * _T + 0.3 * np.random.default_rng(5).normal(size=150)], axis=1)


def regression(theta):
    a, b, ls = theta
    s = T.exp(ls)
    return T.sum_over(REG_DATA, lambda row: -0.5 * ((row[1] - (a + b * row[0])) / s) ** 2 - ls) - 0.5 * (a * a + b * b) * 0.01


MODELS = {"big_data": (big_data_density, 2, [0.4, 1.7]), "regression": (regression, 3, [0.5, -1.0, -1.0]),
          "readme": (readme_density, 2, [0.3, 1.2]), "nig": (nig, 2, [1.4, 0.6]), "quadratic": (quadratic, 2, [0.5, -0.8]),
          "rosenbrock_like": (rosenbrock_like, 3, [0.7, -0.4, 1.1])}
