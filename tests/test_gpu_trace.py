"""GPU parity for `DensityModel(f)` with a Python callable (mhx/trace.py): the traced program runs inside every sampler's
kernel and is bit-identical to the oracle evaluating the SAME emitted source compiled for the host; the reference's known
answers hold.  Reference: src/AdvancedMH.jl:52-54, README.md:26-60, test/emcee.jl:3-42, test/runtests.jl:334-365."""
import numpy as np
import pytest

import cases
import traced_models as M
import user_targets

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, what
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s: %r vs %r" % (what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])])


@pytest.mark.parametrize("name", sorted(M.MODELS))
def test_traced_logdensity_bit_exact(mhx, oracle, name, real):
    f, d, x0 = M.MODELS[name]
    model = mhx.DensityModel(f, dim=d)
    ut = user_targets.host_target(oracle, model.traced.source, d, data=model.traced.data)
    rng = np.random.default_rng(3)
    x = (np.array(x0)[:, None] + rng.normal(size=(d, 500))).astype(cases.R())      # some points outside the support
    lp = mhx.logdensity(model, x)
    _same(lp, np.array([ut(x[:, i]) for i in range(x.shape[1])], dtype=cases.R()), name)
    assert abs(float(lp[0]) - model.traced.evaluate(x[:, 0].astype(np.float64))) < (1e-12 if real == "f64" else 1e-4) * max(1, abs(float(lp[0])))


def test_readme_example_rwmh(mhx, oracle, real):
    """README.md:26-60: RWMH on the closure over 30 data points; posterior means near the data's mean and std."""
    model = mhx.DensityModel(M.readme_density, dim=2)
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.25 * mhx.I))
    C, N = 64, 3000
    init = np.tile(np.array([[0.0], [1.0]], dtype=cases.R()), (1, C))
    chain = mhx.sample(model, spl, N, C, seed=9, initial_params=init, param_names=["μ", "σ"], discard_initial=500)
    assert abs(chain.mean("μ") - M.README_DATA.mean()) < 0.1 and abs(chain.mean("σ") - M.README_DATA.std()) < 0.15
    assert (chain["σ"] > 0).all()
    ut = user_targets.host_target(oracle, model.traced.source, 2, data=model.traced.data)
    ref = oracle.rwmh(ut, oracle.Proposal(oracle.PROP_ISO, 0.5), oracle.schedule(300, 500), 9, 0, C, init=init)
    _same(chain.value[:300], ref["samples"], "samples")
    _same(chain.accepted[:300], ref["accepted"], "accepted")


def test_nig_emcee_known_answer(mhx, oracle, real):
    """test/emcee.jl:3-42 with the model written as a Python function: E[s] = 49/24, E[m] = 7/6 (atol 0.1)."""
    W = 1000
    model = mhx.DensityModel(M.nig, dim=2)
    spl = mhx.Ensemble(W, mhx.StretchProposal([mhx.InverseGamma(2, 3), mhx.Normal(0, 1)]))
    chain = mhx.sample(model, spl, 1000, seed=100, param_names=["s", "m"])
    assert abs(chain.mean("s") - 49 / 24) < 0.1 and abs(chain.mean("m") - 7 / 6) < 0.1
    ut = user_targets.host_target(oracle, model.traced.source, 2, data=model.traced.data)
    ref = oracle.emcee(ut, 2.0, 1, oracle.schedule(200), 100, 0, W, chain.value[0, :2, :])
    _same(chain.value[:200], ref["samples"], "samples")


def test_traced_gradient_drives_mala(mhx, oracle, real):
    """test/runtests.jl:334-365 (issue #95) with the gradient taken from the trace instead of written by hand."""
    Sig = np.array([[1.5, 0.35], [0.35, 1.0]])
    model = mhx.DensityModel(M.quadratic, dim=2)
    C = 32
    init = np.ones((2, C), dtype=cases.R())
    chain = mhx.sample(model, mhx.MALA(0.5), 20000, C, initial_params=init, seed=1)
    v = chain.value[:, :2, :].astype(np.float64)
    assert np.abs(v.mean(axis=(0, 2))).max() < 0.1
    assert np.abs(np.cov(v.transpose(1, 0, 2).reshape(2, -1)) - Sig).max() < 0.2
    ut = user_targets.host_target(oracle, model.traced.source, 2, data=model.traced.data)
    ref = oracle.mala(ut, 0.5, oracle.schedule(200), 1, 0, C, init, user_grad_addr=ut.grad_addr)
    _same(chain.value[:200], ref["samples"], "samples")
    # without the gradient the capability check refuses MALA (src/MALA.jl:42-52)
    with pytest.raises(mhx.MhxError) as ei:
        mhx.sample(mhx.DensityModel(M.quadratic, dim=2, gradient=False), mhx.MALA(0.5), 10, initial_params=np.ones(2))
    assert ei.value.code == -4


def test_traced_model_under_ram(mhx, oracle, real):
    d, C, N, warm = 3, 6, 40, 30
    model = mhx.DensityModel(M.rosenbrock_like, dim=d)
    init = np.zeros((d, C), dtype=cases.R())
    chain = mhx.sample(model, mhx.RobustAdaptiveMetropolis(), N, C, seed=17, initial_params=init, num_warmup=warm, discard_initial=0)
    ut = user_targets.host_target(oracle, model.traced.source, d, data=model.traced.data)
    ref = oracle.ram(ut, oracle.schedule(N, 0, 1, warm), 17, 0, C, init=init)
    _same(chain.value, ref["samples"], "samples")


def test_regression_with_a_data_loop_under_mala(mhx, oracle, real):
    """A closure over 150 rows of data as ONE loop in the kernel (T.sum_over), its gradient from the trace's fused adjoint loop:
    MALA recovers the least-squares line; bit-exact vs the oracle on the emitted source and its data block."""
    model = mhx.DensityModel(M.regression, dim=3)
    assert model.traced.data.size == 300 and model.traced.n_operations < 40
    C = 64
    init = np.tile(np.array([[0.0], [0.0], [0.0]], dtype=cases.R()), (1, C))
    chain = mhx.sample(model, mhx.MALA(2e-3), 6000, C, initial_params=init, seed=3, discard_initial=1500)
    t, y = M.REG_DATA[:, 0], M.REG_DATA[:, 1]
    b_ols, a_ols = np.polyfit(t, y, 1)
    v = chain.value[:, :3, :].astype(np.float64).mean(axis=(0, 2))
    assert abs(v[0] - a_ols) < 0.05 and abs(v[1] - b_ols) < 0.05 and abs(np.exp(v[2]) - 0.3) < 0.06
    ut = user_targets.host_target(oracle, model.traced.source, 3, data=model.traced.data)
    ref = oracle.mala(ut, 2e-3, oracle.schedule(100, 1500), 3, 0, C, init, user_grad_addr=ut.grad_addr)
    _same(chain.value[:100], ref["samples"], "samples")
