"""CPU: the oracle's samplers against the committed golden traces (drift pin) and against every
analytic known answer the reference's own tests use."""
import os

import numpy as np
import pytest

import cases
import user_targets

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(cases.TRACE_CASES))
def test_oracle_reproduces_golden_traces(oracle, real, name):
    traces = np.load(os.path.join(GOLD, "traces64.npz" if real == "f64" else "traces.npz"))
    res = cases.TRACE_CASES[name](oracle)
    for k, v in res.items():
        want = traces["%s/%s" % (name, k)]
        assert v.shape == want.shape
        assert np.array_equal(v.view(np.uint8), want.view(np.uint8)), "%s/%s drifted" % (name, k)


def test_rwmh_normal_model_known_answer(oracle, real):
    """test/runtests.jl:76-94: RWMH(MvNormal(zeros(2), I)) on the Normal(mu, sigma) likelihood of 300
    N(0,1) points; posterior mean mu ~ 0, sigma ~ 1 (atol 0.1).  100 000 draws, one chain, as README.md:40."""
    data = np.load(os.path.join(GOLD, "c1_normal_data.npy"))
    t = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data)
    r = oracle.rwmh(t, oracle.Proposal(oracle.PROP_ISO, 1.0), oracle.schedule(100000), 1234, 0, 1,
                    init=np.array([[0.0], [1.0]], dtype=np.float32))
    mu, sig = r["samples"][:, 0, 0].astype(np.float64), r["samples"][:, 1, 0].astype(np.float64)
    assert abs(mu.mean() - data.mean()) < 0.1 and abs(mu.mean()) < 0.1
    assert abs(sig.mean() - 1.0) < 0.1
    assert (sig >= 0).all()                                   # never leaves the support
    acc = r["accepted"][1:, 0].mean()
    assert 0.001 < acc < 0.2                                   # unit-scale proposal on a sharp posterior (sd ~ 0.06)


def test_rwmh_first_sample_and_schedule(oracle, real):
    init = np.random.default_rng(0).normal(size=(3, 4)).astype(np.float32)
    r = oracle.rwmh(oracle.iso_gauss(3), oracle.Proposal(oracle.PROP_ISO, 0.7), oracle.schedule(6), 9, 0, 4, init=init)
    assert np.array_equal(r["samples"][0, :3, :], init) and not r["accepted"][0].any()   # test/runtests.jl:203-213
    # thinning/discard pick the same states out of the same chain
    full = oracle.rwmh(oracle.iso_gauss(3), oracle.Proposal(oracle.PROP_ISO, 0.7), oracle.schedule(40), 9, 0, 4, init=init)
    thin = oracle.rwmh(oracle.iso_gauss(3), oracle.Proposal(oracle.PROP_ISO, 0.7), oracle.schedule(5, 7, 4), 9, 0, 4, init=init)
    assert np.array_equal(thin["samples"], full["samples"][7:7 + 4 * 5:4])
    # global chain ids: a shard equals the matching slice of the whole
    part = oracle.rwmh(oracle.iso_gauss(3), oracle.Proposal(oracle.PROP_ISO, 0.7), oracle.schedule(40), 9, 2, 2, init=init[:, 2:])
    assert np.array_equal(part["samples"], full["samples"][:, :, 2:])


def test_rwmh_edge_cases(oracle, real):
    # lp = -inf at the start and a finite candidate: +inf log-ratio => accept
    t = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=np.zeros(5, dtype=np.float32))
    r = oracle.rwmh(t, oracle.Proposal(oracle.PROP_ISO, 0.5), oracle.schedule(200), 3, 0, 8,
                    init=np.tile(np.array([[0.0], [-0.2]], dtype=np.float32), (1, 8)))
    assert np.isneginf(r["samples"][0, 2, :]).all()
    later = r["samples"][-1, 2, :]
    fin0 = np.isfinite(r["samples"][:, 2, :])
    assert np.isfinite(later).all()                          # every chain escaped the -inf start
    # candidates outside the support are always rejected
    # once finite, lp never returns to -inf (a -inf candidate is always rejected)
    assert (np.diff(fin0.astype(int), axis=0) >= 0).all()
    fin = np.isfinite(r["samples"][:, 2, :])
    assert (r["samples"][:, 1, :][fin] > 0).all()


@pytest.mark.parametrize("mode", [0, 1])
def test_emcee_nig_known_answer_untransformed(oracle, mode, real):
    """test/emcee.jl:3-42: E[s] = 49/24, E[m] = 7/6 (atol 0.1); Ensemble(1000, StretchProposal(...)), 1000 iterations.
    mode 0 = the reference's sequential sweep, mode 1 = the parallel half-split the GPU runs."""
    t = user_targets.host_target(oracle, user_targets.NIG_UNTRANSFORMED, 2)
    rng = np.random.default_rng(100)
    W = 1000
    init = np.stack([3.0 / rng.gamma(2.0, size=W), rng.normal(size=W)]).astype(np.float32)   # [InverseGamma(2,3), Normal(0,1)]
    r = oracle.emcee(t, 2.0, mode, oracle.schedule(1000), 100, 0, W, init)
    s, m = r["samples"][:, 0, :].astype(np.float64), r["samples"][:, 1, :].astype(np.float64)
    assert abs(s.mean() - 49 / 24) < 0.1
    assert abs(m.mean() - 7 / 6) < 0.1
    assert (s > 0).all()
    r2 = oracle.emcee(t, 2.0, mode, oracle.schedule(200, 25, 4), 100, 0, W, init)
    assert np.array_equal(r2["samples"], r["samples"][25:25 + 4 * 200:4])     # test/emcee.jl:39 index arithmetic


def test_emcee_nig_known_answer_transformed(oracle, real):
    """test/emcee.jl:44-83 (log-transformed space, initial walkers ~ MvNormal(zeros(2), I))."""
    t = user_targets.host_target(oracle, user_targets.NIG_TRANSFORMED, 2)
    W = 1000
    init = np.random.default_rng(100).normal(size=(2, W)).astype(np.float32)
    r = oracle.emcee(t, 2.0, 1, oracle.schedule(1000), 101, 0, W, init)
    logs, m = r["samples"][:, 0, :].astype(np.float64), r["samples"][:, 1, :].astype(np.float64)
    assert abs(np.exp(logs).mean() - 49 / 24) < 0.1
    assert abs(m.mean() - 7 / 6) < 0.1


def test_emcee_partner_is_never_self_and_initial_sample(oracle, real):
    d, W = 2, 6
    init = cases.emcee_init(d, W, 1)
    for mode in (0, 1):
        r = oracle.emcee(oracle.iso_gauss(d), 2.0, mode, oracle.schedule(50), 5, 0, W, init)
        assert np.array_equal(r["samples"][0, :d, :], init) and not r["accepted"][0].any()
        # a stretch move keeps the walker on the line through its partner: accepted moves changed x
        moved = (np.diff(r["samples"][:, 0, :], axis=0) != 0)
        assert np.array_equal(moved, r["accepted"][1:].astype(bool))


@pytest.mark.parametrize("var", [10.0, 0.01])
def test_ram_eigenvalue_bounds(oracle, var, real):
    """test/RobustAdaptiveMetropolis.jl:30-72: diag(S) stays within [0.9, 1.1] and saturates the relevant bound."""
    Sig = np.array([[var, var / 2], [var / 2, var]])
    C = 8
    r = oracle.ram(oracle.corr_gauss_from_cov(Sig), oracle.schedule(1000, 0, 1, 1000), 7, 0, C,
                   init=np.zeros((2, C), dtype=np.float32), gamma=0.51, eig_lo=0.9, eig_hi=1.1)
    assert (r["diag_min"] >= 0.9).all() and (r["diag_max"] <= 1.1).all()
    if var < 0.5:
        assert np.abs(r["diag_min"] - 0.9).max() < 0.05
    else:
        assert np.abs(r["diag_max"] - 1.1).max() < 0.05
    assert r["accepted"][0].all()                             # initial Transition(x, lp, true), RAM.jl:213


def test_ram_doctest_covariance(oracle, real):
    """src/RobustAdaptiveMetropolis.jl:17-70: 2-d Gaussian, correlation 0.5; 10 000 warm-up + 10 000 draws from
    zeros(2): cov(chain) ~ Sigma (rtol 0.2); with bounds [0.1, 2.0]: |cov - Sigma| < 0.2."""
    Sig = np.array([[1.0, 0.5], [0.5, 1.0]])
    C = 4
    for kw in ({}, dict(eig_lo=0.1, eig_hi=2.0)):
        r = oracle.ram(oracle.corr_gauss_from_cov(Sig), oracle.schedule(10000, 10000, 1, 10000), 42, 0, C,
                       init=np.zeros((2, C), dtype=np.float32), **kw)
        for c in range(C):
            cov = np.cov(r["samples"][:, :2, c].astype(np.float64).T)
            assert np.linalg.norm(cov - Sig) < 0.2 * np.linalg.norm(Sig) + 0.05
        # the adapted factor approximates a scaled chol(Sigma): acceptance near the 0.234 target afterwards
        acc = r["accepted"].mean()
        assert 0.12 < acc < 0.4
        assert (r["status"] & 1).sum() == 0


def test_mala_known_answers(oracle, real):
    """test/runtests.jl:288-366.  basic: the Normal(mu, sigma) model with sigma2 = 1e-3 from ones(2): means ~ (0, 1);
    issue #95: 2-d Gaussian Sigma = [1.5 .35; .35 1], sigma2 = 0.5: mean ~ 0 (atol .1), cov ~ Sigma (atol .2)."""
    data = np.load(os.path.join(GOLD, "c1_normal_data.npy"))
    t = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data)
    r = oracle.mala(t, 1e-3, oracle.schedule(1000, 100), 3, 0, 8, np.ones((2, 8), dtype=np.float32))
    v = r["samples"].astype(np.float64)
    assert abs(v[:, 0, :].mean() - data.mean()) < 0.1 and abs(v[:, 1, :].mean() - 1.0) < 0.1
    Sig = np.array([[1.5, 0.35], [0.35, 1.0]])
    r = oracle.mala(oracle.corr_gauss_from_cov(Sig), 0.5, oracle.schedule(100000), 1, 0, 4, np.ones((2, 4), dtype=np.float32))
    v = r["samples"][:, :2, :].astype(np.float64)
    assert np.abs(v.mean(axis=(0, 2))).max() < 0.1
    assert np.abs(np.cov(v.transpose(1, 0, 2).reshape(2, -1)) - Sig).max() < 0.2
    assert np.array_equal(r["samples"][0, :2, :], np.ones((2, 4), dtype=np.float32)) and not r["accepted"][0].any()
    # same model through a user source with its own gradient (TheNormalLogDensity(inv(Sigma)))
    A = np.linalg.inv(Sig).astype(np.float32)
    ut = user_targets.host_target(oracle, user_targets.QUADRATIC_WITH_GRADIENT, 2, data=A.ravel())
    r2 = oracle.mala(ut, 0.5, oracle.schedule(50000), 1, 0, 4, np.ones((2, 4), dtype=np.float32), user_grad_addr=ut.grad_addr)
    v2 = r2["samples"][:, :2, :].astype(np.float64)
    assert np.abs(np.cov(v2.transpose(1, 0, 2).reshape(2, -1)) - Sig).max() < 0.2


def test_catalogue_gradients_against_finite_differences(oracle, real):
    rng = np.random.default_rng(0)
    d = 6
    x = rng.normal(size=d).astype(np.float32)
    Sig = cases.sigma_ar1(d, 0.7)
    for t in (oracle.iso_gauss(d), oracle.corr_gauss_from_cov(Sig), oracle.Target(oracle.TARGET_BANANA, d, params=[0.03]),
              oracle.Target(oracle.TARGET_FUNNEL, d)):
        lp, g = oracle.target_grad(t, x)
        assert lp == t(x)
        for k in range(d):
            e = np.zeros(d, dtype=np.float32)
            e[k] = 1e-3
            assert abs(g[k] - (t(x + e) - t(x - e)) / 2e-3) < 2e-3 * max(1.0, abs(g[k]))


def test_drifting_walk_is_corrected_by_the_hastings_ratio(oracle, real):
    """src/proposal.jl:58-64,190-192: with a non-zero proposal mean the ratio q(x|y) - q(y|x) is what keeps the target invariant."""
    d = 3
    mean = np.array([0.3, -0.2, 0.1], dtype=np.float32)
    r = oracle.rwmh(oracle.iso_gauss(d), oracle.Proposal(oracle.PROP_ISO, 0.8, mean=mean), oracle.schedule(40000, 1000), 5, 0, 8)
    v = r["samples"][:, :d, :].astype(np.float64)
    assert np.abs(v.mean(axis=(0, 2))).max() < 0.03 and np.abs(v.var(axis=(0, 2)) - 1).max() < 0.03


def test_static_proposal_is_an_independence_sampler(oracle, real):
    """StaticProposal (src/proposal.jl:9-11,66-83): candidates ignore the state, the ratio carries the proposal's
    logpdf.  2-D Gaussian with correlation 0.5 under N(0, 2 I): mean 0, covariance recovered; the first sample of a
    run without initial_params is a draw from the proposal (src/mh-core.jl:83)."""
    Sig = np.array([[1.0, 0.5], [0.5, 1.0]])
    s = float(np.float32(np.sqrt(2.0)))
    ref = oracle.rwmh(oracle.corr_gauss_from_cov(Sig), oracle.Proposal(oracle.PROP_ISO, s, static=True),
                      oracle.schedule(1500, 100), 1, 0, 64, init=None)
    flat = ref["samples"][:, :2, :].transpose(1, 0, 2).reshape(2, -1)
    assert np.abs(flat.mean(axis=1)).max() < 0.05
    assert np.abs(np.cov(flat) - Sig).max() < 0.08
    first = oracle.rwmh(oracle.corr_gauss_from_cov(Sig), oracle.Proposal(oracle.PROP_ISO, s, static=True),
                        oracle.schedule(1), 1, 0, 4096, init=None)["samples"][0, :2, :]
    assert abs(first.std() - s) < 0.05
    # with a non-zero mean the same posterior comes out (the mean enters the ratio through the whitening)
    ref = oracle.rwmh(oracle.corr_gauss_from_cov(Sig), oracle.Proposal(oracle.PROP_ISO, s, mean=np.array([0.4, -0.3]), static=True),
                      oracle.schedule(1500, 100), 2, 0, 64, init=None)
    flat = ref["samples"][:, :2, :].transpose(1, 0, 2).reshape(2, -1)
    assert np.abs(flat.mean(axis=1)).max() < 0.05
    assert np.abs(np.cov(flat) - Sig).max() < 0.08


@pytest.mark.parametrize("d,K", [(5, 8), (13, 3), (70, 8), (200, 8), (256, 1)])
def test_ram_deferred_twin_is_the_reference_order_chain_up_to_rounding(oracle, d, K):
    """Arithmetic spec 3.12 (orc_ram_deferred, the twin of MHX_FLAG_RAM_DEFERRED) against item 9 (orc_ram: the reference's sequential
    lowrankupdate! / lowrankdowndate! sweeps, src/RobustAdaptiveMetropolis.jl:153-173) on the same seeds, in fp64: S <- S chol(I +- c^2 U U')
    IS the rank-one update, so the accept decisions are the same, the chains agree to round-off and S S' to a few ulp -- for any block
    length K, with forced folds in the middle of a block."""
    oracle.set_dtype("f64")
    C, N = 3, 160
    Sig = cases.sigma_ar1(d, 0.7)
    rng = np.random.default_rng(d)
    init, S0 = rng.normal(size=(d, C)), np.eye(d) * (2.38 / np.sqrt(d))
    Sin = np.tile(oracle.pack_lower(S0), (C, 1))
    sched = oracle.schedule(N, 0, 1, N - 21)
    a = oracle.ram(oracle.corr_gauss_from_cov(Sig), sched, 31, 2, C, init=init, S_in=Sin)
    b = oracle.ram_deferred(oracle.corr_gauss_from_cov(Sig), sched, 31, 2, C, init=init, S_in=Sin, K=K, flush_at=[5, 37, 38, 100])
    assert 0.05 < a["accepted"][1:].mean() < 0.95
    assert np.array_equal(a["accepted"], b["accepted"])
    assert np.abs(a["samples"] - b["samples"]).max() < 1e-11
    assert np.array_equal(a["status"], b["status"]) and not a["status"].any()
    for c in range(C):
        A, B = oracle.unpack_lower(a["S"][c], d), oracle.unpack_lower(b["S"][c], d)
        assert np.linalg.norm(A @ A.T - B @ B.T) <= 2e-14 * np.linalg.norm(A @ A.T)
        assert np.abs(a["diag_min"][:, c] - b["diag_min"][:, c]).max() <= 1e-13 * np.abs(a["diag_min"][:, c]).max()


def test_ram_deferred_twin_respects_the_eigenvalue_bounds(oracle, real):
    """a refused update is not pending: with bounds that bite the deferred factor never leaves them (test/RobustAdaptiveMetropolis.jl:57-69)"""
    d, C, N = 5, 7, 60
    rng = np.random.default_rng(3)
    L = np.tril(rng.normal(size=(d, d)) * 0.2) + np.eye(d)
    Sin = np.tile(oracle.pack_lower(L), (C, 1))
    r = oracle.ram_deferred(oracle.iso_gauss(d), oracle.schedule(N, 0, 1, N), 77, 0, C, S_in=Sin, gamma=0.7, eig_lo=0.58, eig_hi=1.45)
    free = oracle.ram_deferred(oracle.iso_gauss(d), oracle.schedule(N, 0, 1, N), 77, 0, C, S_in=Sin, gamma=0.7)
    assert (r["diag_min"] >= 0.58).all() and (r["diag_max"] <= 1.45).all()
    assert free["diag_min"].min() < 0.58 or free["diag_max"].max() > 1.45          # the bounds did refuse something
    for c in range(C):
        dg = np.diag(oracle.unpack_lower(r["S"][c], d))
        assert (dg >= 0.58).all() and (dg <= 1.45).all()
