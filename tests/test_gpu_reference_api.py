"""The reference's container / proposal-style tests on the GPU engine with the model written as a closure.
Reference: test/runtests.jl:22-30 (model), :112-201 (MCMCChains, proposal styles), :203-213 (initial parameters)."""
import math
import os

import numpy as np
import pytest

import mhx.trace as T

pytestmark = pytest.mark.gpu
DATA = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_normal_data.npy")).astype(np.float64)
LOG2PI = math.log(2 * math.pi)


def density(theta):                                          # test/runtests.jl:26-28
    mu, sigma = theta
    ls = T.log(sigma)
    lp = sum(-0.5 * ((y - mu) / sigma) ** 2 - ls - 0.5 * LOG2PI for y in DATA)
    return T.where(sigma >= 0, lp, -math.inf)


def test_namedtuple_static_proposals_and_chains(mhx, real):
    """test/runtests.jl:136-160: MetropolisHastings((μ = StaticProposal(Normal(0,1)), σ = StaticProposal(Normal(0,1))))."""
    model = mhx.DensityModel(density, dim=2)
    spl = mhx.MetropolisHastings({"μ": mhx.StaticProposal(mhx.Normal(0, 1)), "σ": mhx.StaticProposal(mhx.Normal(0, 1))})
    chain = mhx.sample(model, spl, 10_000, 8, seed=2)
    assert chain.names == ["μ", "σ", "lp"] and chain.range() == range(1, 10_001)
    assert abs(chain.mean("μ") - DATA.mean()) < 0.1 and abs(chain.mean("σ") - DATA.std()) < 0.1
    chain_b = mhx.sample(model, spl, 10_000, 8, seed=2, discard_initial=25, thinning=4)
    assert chain_b.range() == range(26, 26 + 4 * 10_000, 4)
    assert abs(chain_b.mean("μ") - DATA.mean()) < 0.1 and abs(chain_b.mean("σ") - DATA.std()) < 0.1
    # the same sampler written as a vector of Normals draws the same chain
    chain_v = mhx.sample(model, mhx.StaticMH([mhx.Normal(0, 1), mhx.Normal(0, 1)]), 200, 8, seed=2)
    assert np.array_equal(chain_v.value, chain.value[:200])


def test_proposal_styles_and_containers(mhx, real):
    """test/runtests.jl:181-201: keys of the NamedTuple container for scalar, vector and named parameters."""
    m1 = mhx.DensityModel(lambda x: -0.5 * (1.0 - x[0]) ** 2 - 0.5 * LOG2PI, dim=1)
    c1 = mhx.sample(m1, mhx.MetropolisHastings(mhx.StaticProposal(mhx.Normal(0, 1))), 100, chain_type=dict, seed=1)
    assert len(c1) == 100 and tuple(c1[0].keys()) == ("param_1", "lp")

    def m3f(x):
        z = (1.0 - x.a) / x.b
        return T.where(x.b > 0, -0.5 * z * z - T.log(x.b) - 0.5 * LOG2PI, -math.inf)

    m3 = mhx.DensityModel(m3f, names=("a", "b"))
    p3 = {"a": mhx.StaticProposal(mhx.Normal(0, 1)), "b": mhx.StaticProposal(mhx.Normal(1, 1))}
    c3 = mhx.sample(m3, mhx.MetropolisHastings(p3), 100, chain_type=dict, seed=1, initial_params=np.array([0.0, 1.0]))
    assert tuple(c3[0].keys()) == ("a", "b", "lp") and all(s["b"] > 0 for s in c3)
    sa = mhx.sample(m3, mhx.MetropolisHastings(p3), 100, 4, chain_type=mhx.StructArray, seed=1, initial_params=np.array([0.0, 1.0]))
    assert sa.keys() == ("a", "b", "lp") and sa.a.shape == (100, 4)


def test_initial_parameters_and_transitions(mhx, real):
    """test/runtests.jl:203-213: chain1[1].params == val with the reference's default container (a vector of Transitions)."""
    model = mhx.DensityModel(density, dim=2)
    val = np.array([0.4, 1.2])
    chain = mhx.sample(model, mhx.StaticMH([mhx.Normal(0, 1), mhx.Normal(0, 1)]), 10, initial_params=val, chain_type=mhx.Transition)
    assert len(chain) == 10 and np.array_equal(np.asarray(chain[0].params, dtype=np.float64), val.astype(chain[0].params.dtype))
    assert chain[0].lp == pytest.approx(float(density(list(val))), rel=1e-5 if real == "f32" else 1e-12)
    assert isinstance(chain[3].accepted, bool)


def test_scalar_parameter_closure(mhx, real):
    """test/runtests.jl:162-178: DensityModel(x -> loglikelihood(Normal(x, 1), data)) with StaticMH(Normal(0, 1))."""
    model = mhx.DensityModel(lambda x: sum(-0.5 * (y - x[0]) ** 2 - 0.5 * LOG2PI for y in DATA), dim=1)
    chain = mhx.sample(model, mhx.StaticMH(mhx.Normal(0, 1)), 10_000, 4, param_names=["μ"], seed=5)
    assert chain.range() == range(1, 10_001) and abs(chain.mean("μ") - DATA.mean()) < 0.1
    chain_b = mhx.sample(model, mhx.StaticMH(mhx.Normal(0, 1)), 10_000, 4, param_names=["μ"], seed=5, discard_initial=25, thinning=4)
    assert chain_b.range() == range(26, 26 + 4 * 10_000, 4) and abs(chain_b.mean("μ") - DATA.mean()) < 0.1


@pytest.mark.soak_f32
def test_mala_closure_form_runs_the_same_chain(mhx, real):
    """test/runtests.jl:291: MALA(x -> MvNormal((σ² / 2) .* x, σ² * I)) is MALA(σ²) of the engine."""
    model = mhx.DensityModel(density, dim=2)
    s2 = 1e-3
    init = np.ones(2)
    a = mhx.sample(model, mhx.MALA(lambda g: mhx.MvNormal(0.5 * s2 * g, s2 * mhx.I)), 50, 8, initial_params=init, seed=4)
    b = mhx.sample(model, mhx.MALA(s2), 50, 8, initial_params=init, seed=4)
    assert np.array_equal(a.value, b.value) and a.accepted[1:].mean() > 0.2


class Gaussian:
    """test/RobustAdaptiveMetropolis.jl:1-9: a LogDensityProblems object -- `dimension`, `logdensity` -- for a zero-mean Gaussian with
    Σ = [σ² ρ; ρ σ²].  (The log-density is written out: forward substitution with L = chol(Σ).)"""

    def __init__(self, s2):
        self.l11 = math.sqrt(s2)
        self.l21 = (s2 / 2) / self.l11
        self.l22 = math.sqrt(s2 - self.l21 * self.l21)
        self.c = math.log(self.l11) + math.log(self.l22) + LOG2PI

    def dimension(self):
        return 2

    def logdensity(self, x):
        z1 = x[0] / self.l11
        z2 = (x[1] - self.l21 * z1) / self.l22
        return -(z1 * z1 + z2 * z2) / 2 - self.c


@pytest.mark.parametrize("s2", [10.0, 0.01])
def test_ram_takes_the_logdensityproblems_form(mhx, oracle, real, s2):
    """test/RobustAdaptiveMetropolis.jl:30-69 as the reference writes it: `sample(model, sampler, num_warmup; num_warmup,
    discard_initial = 0, initial_params = zeros(2), callback)` with `model` a LogDensityProblems OBJECT -- the only form the
    reference's RobustAdaptiveMetropolis has `step` methods for (src/RobustAdaptiveMetropolis.jl:175-181).  The dimension comes from
    the problem, the log-density is traced through problem.logdensity; the chains are the oracle's on the emitted source bit for bit,
    every recorded state.S keeps its diagonal (its eigenvalues) inside the bounds and saturates at the bound the target pushes to."""
    import cases
    import user_targets
    model = Gaussian(s2)
    spl = mhx.RobustAdaptiveMetropolis(γ=0.51, eigenvalue_lower_bound=0.9, eigenvalue_upper_bound=1.1)
    num_warmup, C = 1000, 64
    chain = mhx.sample(model, spl, num_warmup, C, num_warmup=num_warmup, discard_initial=0, initial_params=np.zeros(2), seed=11)
    assert chain.value.shape == (num_warmup, 3, C) and chain.names == ["param_1", "param_2", "lp"]
    lo, hi = chain.state.diag_range()
    assert (lo >= 0.9).all() and (hi <= 1.1).all()
    assert np.abs((hi if s2 > 0.5 else lo) - (1.1 if s2 > 0.5 else 0.9)).max() < 0.05
    wrapped = mhx.LogDensityModel(model)                                   # what `sample` made of it (AbstractMCMC wraps the same way)
    assert wrapped.dim == 2
    ut = user_targets.host_target(oracle, wrapped.traced.source, 2, data=wrapped.traced.data)
    N = 120
    ref = oracle.ram(ut, oracle.schedule(N, 0, 1, N), 11, 0, C, init=np.zeros((2, C), dtype=cases.R()), gamma=0.51, eig_lo=0.9, eig_hi=1.1)
    bad = np.argwhere(cases.bits(chain.value[:N]) != cases.bits(ref["samples"]))
    assert len(bad) == 0, bad[:3]
    # the explicit wrapper and a catalogue target (its own problem) go the same way
    again = mhx.sample(wrapped, spl, 50, C, num_warmup=50, discard_initial=0, initial_params=np.zeros(2), seed=11)
    assert np.array_equal(cases.bits(again.value), cases.bits(chain.value[:50]))
    cat = mhx.sample(mhx.CorrGaussian(np.array([[s2, s2 / 2], [s2 / 2, s2]])), spl, 20, 8, num_warmup=20, initial_params=np.zeros(2))
    assert cat.value.shape == (20, 3, 8)
