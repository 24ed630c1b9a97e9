"""GPU: catalogue targets, JIT-specialised vs generic kernels, C1 plumbing, large-d streaming path,
shard invariance, diagnostics, error behaviour -- all through the C ABI."""
import os

import numpy as np
import pytest

import cases
from conftest import soak_tail
import user_targets

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    assert a.dtype == b.dtype, "%s: dtypes %s / %s" % (what, a.dtype, b.dtype)
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s: %r vs %r" % (
        what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])])


def test_logdensity_batch_matches_oracle(mhx, oracle, real):
    rng = np.random.default_rng(0)
    d, n = 9, 300
    x = rng.normal(size=(d, n)).astype(np.float32)
    Sig = cases.sigma_ar1(d, 0.6)
    specs = [
        (mhx.IsoGaussian(d), oracle.iso_gauss(d)),
        (mhx.CorrGaussian(Sig), oracle.corr_gauss_from_cov(Sig)),
        (mhx.Banana(d, 0.03), oracle.Target(oracle.TARGET_BANANA, d, params=[0.03])),
        (mhx.Funnel(d), oracle.Target(oracle.TARGET_FUNNEL, d)),
        (mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=np.r_[np.arange(d), 1 + np.arange(d)]),
         user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=np.r_[np.arange(d), 1 + np.arange(d)])),
    ]
    for spec, ot in specs:
        lp = mhx.logdensity(mhx.DensityModel(spec), x)
        want = np.array([ot(x[:, i]) for i in range(n)], dtype=cases.R())
        _same(lp, want, type(spec).__name__)
    data = rng.normal(size=30).astype(np.float32)
    th = np.stack([rng.normal(size=n), rng.normal(size=n) + 1.0]).astype(np.float32)   # some sigma < 0
    lp = mhx.logdensity(mhx.DensityModel(mhx.IIDNormal(data)), th)
    ot = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data)
    _same(lp, np.array([ot(th[:, i]) for i in range(n)], dtype=cases.R()), "IIDNormal")
    assert np.isneginf(lp[th[1] < 0]).all()


@pytest.mark.parametrize("flags_name", ["auto", "nojit", "generic"])   # golden traces use the sequential (1-lane) shape
@pytest.mark.parametrize("name", ["rwmh_iso", "rwmh_dense_corr", "rwmh_funnel", "rwmh_banana"])
def test_rwmh_golden_traces_all_kernel_variants(mhx, name, flags_name, real):
    tr = np.load(os.path.join(GOLD, "traces64.npz" if real == "f64" else "traces.npz"))
    flags = {"auto": 0, "nojit": mhx.FLAG_NO_JIT, "generic": mhx.FLAG_GENERIC}[flags_name]
    if name == "rwmh_iso":
        chain = mhx.sample(mhx.DensityModel(mhx.IsoGaussian(5)), mhx.RWMH(mhx.MvNormal(mhx.zeros(5), 0.25 * mhx.I)), 32, 8,
                           seed=11, first_chain=3, flags=flags, reduce_lanes=1)
    elif name == "rwmh_dense_corr":
        d = 4
        chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.8))),
                           mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.3 * cases.sigma_ar1(d, 0.5))), 20, 6, seed=12,
                           discard_initial=3, thinning=2, flags=flags)
    elif name == "rwmh_funnel":
        chain = mhx.sample(mhx.DensityModel(mhx.Funnel(6)), mhx.RWMH(mhx.MvNormal(mhx.zeros(6), 0.16 * mhx.I)), 24, 7,
                           seed=13, first_chain=100, flags=flags, reduce_lanes=1)
    else:
        chain = mhx.sample(mhx.DensityModel(mhx.Banana(5, 0.03)),
                           mhx.RWMH([mhx.Normal(0, 2.0), mhx.Normal(0, 0.5), mhx.Normal(0, 1), mhx.Normal(0, 1), mhx.Normal(0, 1)]),
                           24, 7, seed=14, flags=flags, reduce_lanes=1)
    _same(chain.value, tr[name + "/samples"], "samples")
    _same(chain.accepted, tr[name + "/accepted"], "accepted")
    want_variant = {"auto": 2, "nojit": 0, "generic": 0}[flags_name]
    assert chain.stats["kernel_variant"] == want_variant


@pytest.mark.parametrize("lanes,variant", [(0, 11), (1, 1)])
def test_c1_readme_plumbing(mhx, oracle, real, lanes, variant):
    """BASELINE config 1 / README.md:25-40: Normal(mu, sigma) DensityModel, RWMH(MvNormal(zeros(2), I)),
    100 000 steps, ONE chain -- GPU result identical to the CPU oracle and close to the data's moments.  The engine's choice for a
    single chain is the wave-per-chain kernel (variant 11, round 5: the lanes split the 30 likelihood terms, reduction shape 64);
    reduce_lanes = 1 keeps the lane-per-chain register kernel (the plain ascending sum)."""
    data = np.load(os.path.join(GOLD, "c1_normal_data.npy"))[:30]
    model = mhx.DensityModel(mhx.IIDNormal(data))
    chain = mhx.sample(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(2), mhx.I)), 100000, 1, seed=1234,
                       initial_params=np.array([0.0, 1.0]), param_names=["μ", "σ"], reduce_lanes=lanes)
    L = chain.stats["reduce_lanes"]
    assert chain.stats["kernel_variant"] == variant and L == (64 if variant == 11 else 1)
    t = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data, reduce_lanes=L)
    ref = oracle.rwmh(t, oracle.Proposal(oracle.PROP_ISO, 1.0), oracle.schedule(100000), 1234, 0, 1,
                      init=np.array([[0.0], [1.0]], dtype=np.float32))
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert chain.names == ["μ", "σ", "lp"]
    assert abs(chain.mean("μ") - data.mean()) < 0.1 and abs(chain.mean("σ") - data.std()) < 0.15
    d = chain.state.diagnostics(max_lag=2000, ess_chains=1)
    ess = d["ess_geyer"][:2]
    assert (ess > 1000).all() and (ess < 30000).all()           # README.md:59-63 shows ESS ~ 3.9e3 of 1e5 draws


def test_large_dim_streaming_kernel(mhx, oracle, real):
    """BASELINE config 5 shape: 1000-dim funnel / banana, state streamed from HBM (generic kernel)."""
    d, C, N = 1000, 96, 6
    s = float(np.float32(2.38 / d ** 0.5))
    for spec, ot in ((mhx.Funnel(d), oracle.Target(oracle.TARGET_FUNNEL, d)),
                     (mhx.Banana(d, 0.03), oracle.Target(oracle.TARGET_BANANA, d, params=[0.03]))):
        for lanes, flags in ((0, 0), (1, 0), (32 if real == "f64" else 16, 0), (0, mhx.FLAG_GENERIC)):
            chain = mhx.sample(mhx.DensityModel(spec), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), N, C, seed=55,
                               first_chain=1 << 33, reduce_lanes=lanes, flags=flags)
            L = chain.stats["reduce_lanes"]
            ref = oracle.rwmh(ot.with_lanes(L), oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(N), 55, 1 << 33, C)
            _same(chain.value, ref["samples"], "samples L=%d" % L)
            if lanes == 0 and flags == 0:
                assert L == 64 and chain.stats["kernel_variant"] in (3, 4)  # wave per chain (few chains), cooperative kernel
            if lanes == 1 or flags:
                assert L == 1 and chain.stats["kernel_variant"] == 0       # state streamed from HBM


def test_shard_invariance_and_resume(mhx, real):
    """chains carry global ids: two shards == one run; a run continued in pieces == one long run."""
    d, C, N = 6, 200, 30
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.3 * mhx.I))
    whole = mhx.sample(model, spl, N, C, seed=9, reduce_lanes=2)
    a = mhx.sample(model, spl, N, 120, seed=9, first_chain=0, reduce_lanes=2)
    b = mhx.sample(model, spl, N, 80, seed=9, first_chain=120, reduce_lanes=2)
    _same(np.concatenate([a.value, b.value], axis=2), whole.value, "shards")
    run = mhx.Run(model, spl, nchains=C, seed=9, reduce_lanes=2)
    run.init(None)
    run.sample(10, 0, 1, 0)
    p1, _ = run.samples()
    run.sample(20, 1, 1, 0)
    p2, _ = run.samples()
    _same(np.concatenate([p1, p2], axis=0), whole.value, "resume")
    # setparams!! (src/AdvancedMH.jl:150-157): replace params, lp re-evaluated
    x, lp, _ = run.state()
    run.set_params(np.zeros_like(x))
    x2, lp2, _ = run.state()
    assert (x2 == 0).all() and np.allclose(lp2, -0.5 * d * np.log(2 * np.pi), atol=1e-5)


def test_diagnostics_match_numpy(mhx, real):
    d, C, N = 3, 300, 400
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    chain = mhx.sample(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 1.0 * mhx.I)), N, C, seed=3, discard_initial=200)
    dg = chain.state.diagnostics(max_lag=100, ess_chains=0)
    v = chain.value.astype(np.float64)
    m, s2 = v.mean(axis=0), v.var(axis=0, ddof=1)               # [d+1][C]
    assert np.allclose(dg["sum_m"], m.sum(axis=1), rtol=1e-9, atol=1e-7)
    assert np.allclose(dg["sum_m2"], (m * m).sum(axis=1), rtol=1e-9, atol=1e-7)
    assert np.allclose(dg["sum_v"], s2.sum(axis=1), rtol=1e-9)
    # Geyer ESS from the multi-chain autocorrelations rho_t = 1 - (W - A_t)/var+, recomputed in numpy
    def ess_numpy(vv, nlag):
        Nn, Cc = vv.shape
        mm = vv.mean(axis=0)
        xc = vv - mm
        A = np.array([(xc[:Nn - k] * xc[k:]).sum() for k in range(nlag)]) / (Cc * (Nn - 1.0))
        W = vv.var(axis=0, ddof=1).mean()
        varp = (Nn - 1.0) / Nn * W + mm.var(ddof=1)
        rho = 1.0 - (A[0] - A) / varp
        tau, prev = -1.0, np.inf
        for j in range(nlag // 2):
            pm = rho[2 * j] + rho[2 * j + 1]
            if pm <= 0:
                break
            pm = min(pm, prev)
            prev = pm
            tau += 2 * pm
        return Cc * Nn / tau, np.sqrt(varp / W)

    for p in range(d):
        want, _ = ess_numpy(v[:, p, :], 100)
        assert abs(dg["ess_geyer"][p] - want) / want < 1e-3
    # split chains (Vehtari et al. 2021): 2C half-chains of N/2 draws
    ds = chain.state.diagnostics(max_lag=100, ess_chains=0, split=True)
    assert ds["n_chains"] == 2 * C and ds["n_samples"] == N // 2
    for p in range(d):
        halves = np.concatenate([v[:N // 2, p, :], v[N // 2:, p, :]], axis=1)
        want, rhat = ess_numpy(halves, 100)
        assert abs(ds["ess_geyer"][p] - want) / want < 1e-3
        assert abs(ds["rhat"][p] - rhat) < 1e-6
    assert (np.abs(dg["rhat"][:d] - 1) < 0.05).all()
    # between-chain ESS agrees with the autocovariance ESS for stationary replicas
    assert (np.abs(dg["ess_between"][:d] / dg["ess_geyer"][:d] - 1) < 0.25).all()


@pytest.mark.parametrize("flags_name,d,C,lanes", [("auto", 1000, 70, 0), ("auto", 12, 300, 2), ("generic", 9, 130, 0)])
def test_running_moments_match_sample_statistics(mhx, flags_name, d, C, lanes, real):
    """save = "moments": per-chain Welford mean / M2 kept on the device instead of the sample tensor (C5-sized runs);
    the R-hat / ESS sums from them equal the ones from the stored samples of the same chains."""
    flags = mhx.FLAG_GENERIC if flags_name == "generic" else 0
    s = float(np.float32(2.38 / d ** 0.5))
    model = mhx.DensityModel(mhx.Funnel(d) if d > 100 else mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
    N, disc, thin = 40, 3, 2
    out = []
    for mode in (True, "moments"):
        run = mhx.Run(model, spl, nchains=C, seed=77, flags=flags, reduce_lanes=lanes)
        run.init(None)
        run.sample(N, disc, thin, 0, save=mode)
        out.append(run.diagnostics(max_lag=0))
        if mode == "moments":
            with pytest.raises(mhx.MhxError):
                run.samples()
    a, b = out
    for k in ("sum_m", "sum_m2", "sum_v"):
        assert np.allclose(a[k], b[k], rtol=2e-4, atol=1e-4 * C), k
    assert np.allclose(a["rhat"], b["rhat"], rtol=1e-3)
    # the register kernel keeps no moments: asking for them is an argument error, not a silent fallback
    if d == 12:
        run = mhx.Run(model, spl, nchains=C, seed=77, reduce_lanes=1)
        run.init(None)
        with pytest.raises(mhx.ArgumentError):
            run.sample(N, disc, thin, 0, save="moments")


def test_error_behaviour(mhx, real):
    with pytest.raises(mhx.ArgumentError):                      # dim mismatch
        mhx.sample(mhx.DensityModel(mhx.IsoGaussian(3)), mhx.RWMH(4), 5)
    with pytest.raises(mhx.ArgumentError):                      # a drifting walk declared symmetric
        mhx.SymmetricRandomWalkProposal(mhx.MvNormal(np.ones(3), mhx.I))
    with pytest.raises(mhx.ArgumentError):                      # a Python closure cannot be lowered
        mhx.DensityModel(lambda x: -0.5 * (x ** 2).sum())
    with pytest.raises(mhx.MhxError) as ei:                     # bad user source -> compile error at model construction
        mhx.logdensity(mhx.DensityModel(mhx.HipLogDensity("MHX_LOGDENSITY(x, d, data, n) { return undefined_symbol; }", 2)),
                       np.zeros(2))
    assert ei.value.code == -4 and "undefined_symbol" in str(ei.value)
    with pytest.raises(mhx.PosDefException):
        mhx.MvNormal(mhx.zeros(2), np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(mhx.ArgumentError):                      # ensemble of one walker
        mhx.sample(mhx.DensityModel(mhx.IsoGaussian(2)), mhx.Ensemble(1, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(2), mhx.I))), 3)
    with pytest.raises(mhx.ArgumentError):                      # bad schedule
        mhx.sample(mhx.DensityModel(mhx.IsoGaussian(2)), mhx.RWMH(2), 5, thinning=0)


@pytest.mark.parametrize("flags_name", ["auto", "generic"])
def test_infinite_and_nan_log_ratios(mhx, oracle, flags_name, real):
    """src/mh-core.jl:104-108 edge cases: lp = -Inf at the start with a finite candidate (+Inf ratio => accept), a
    candidate outside the support (-Inf => reject), -Inf vs -Inf (NaN => reject) -- identical decisions on the GPU."""
    flags = mhx.FLAG_GENERIC if flags_name == "generic" else 0
    data = np.zeros(5, dtype=np.float32)
    C = 70
    init = np.tile(np.array([[0.0], [-0.2]], dtype=np.float32), (1, C))      # sigma < 0: outside the support
    chain = mhx.sample(mhx.DensityModel(mhx.IIDNormal(data)), mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.25 * mhx.I)), 300, C, seed=3,
                       initial_params=init, flags=flags)
    ref = oracle.rwmh(oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data), oracle.Proposal(oracle.PROP_ISO, 0.5),
                      oracle.schedule(300), 3, 0, C, init=init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert np.isneginf(chain.value[0, 2, :]).all() and np.isfinite(chain.value[-1, 2, :]).all()


def test_ram_nan_log_ratio_skips_adaptation(mhx, oracle, real):
    """RAM from a point of zero density towards more zero density: lp' - lp = NaN, the step is rejected and the
    factor is left alone (status bit 1), exactly as the oracle does."""
    data = np.zeros(4, dtype=np.float32)
    C = 6
    init = np.tile(np.array([[0.0], [-50.0]], dtype=np.float32), (1, C))    # far outside the support
    chain = mhx.sample(mhx.DensityModel(mhx.IIDNormal(data)), mhx.RobustAdaptiveMetropolis(), 12, C, seed=2, initial_params=init,
                       num_warmup=12, discard_initial=0)
    S, st = chain.state.factor()
    ref = oracle.ram(oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data), oracle.schedule(12, 0, 1, 12), 2, 0, C, init=init)
    _same(chain.value, ref["samples"], "samples")
    _same(S, ref["S"], "S")
    _same(st, ref["status"], "status")
    assert (st & 2).all()


def test_symmetric_random_walk_on_a_scalar_model(mhx, real):
    """test/runtests.jl:215-259: target Normal(5, 0.7) as a user log-density, SymmetricRandomWalkProposal(Normal(0, 1)),
    100 000 draws: mean and std within 0.05 (the symmetric flag only skips a Hastings ratio that is 0 anyway)."""
    src = """
    MHX_LOGDENSITY(x, d, data, ndata)
    {
        const float z = (x[0] - 5.0f) / 0.7f;
        return -0.5f * (z * z) - 0x1.d67f1cp-1f - mhx_log(0.7f);
    }
    """
    m1 = mhx.DensityModel(mhx.HipLogDensity(src, 1))
    p2 = mhx.SymmetricRandomWalkProposal(mhx.Normal(0, 1))
    assert p2.issymmetric and not mhx.RandomWalkProposal(mhx.Normal(0, 1)).issymmetric
    chain1 = mhx.sample(m1, mhx.MetropolisHastings(p2), 100000, param_names=["x"], seed=11)
    x = chain1["x"].astype(np.float64)
    assert abs(x.mean() - 5.0) < 0.05 and abs(x.std() - 0.7) < 0.05
    assert chain1.stats["kernel_variant"] == 2                    # hiprtc-specialised register kernel, D = 1


@pytest.mark.parametrize("kind", ["iso", "diag", "dense"])
def test_drifting_random_walk_hastings_ratio(mhx, oracle, kind, real):
    """RandomWalkProposal(MvNormal(mu != 0, Sigma)): the walk drifts and logratio_proposal_density
    (src/proposal.jl:58-64,190-192) is no longer 0 -- evaluated on the device, bit-exact against the oracle, and the
    chain still targets the model (the ratio cancels the drift)."""
    d, C, N = 3, 96, 60
    mean = np.array([0.3, -0.2, 0.1])
    if kind == "iso":
        cov, oprop = 0.64 * mhx.I, oracle.Proposal(oracle.PROP_ISO, 0.8, mean=mean)
    elif kind == "diag":
        sd = np.array([0.5, 1.0, 0.25])
        cov, oprop = sd ** 2, oracle.Proposal(oracle.PROP_DIAG, vec=sd, mean=mean)
    else:
        Sg = 0.5 * np.array([[1, .3, 0], [.3, 1, .2], [0, .2, 1.0]])
        cov, oprop = Sg, oracle.Proposal(oracle.PROP_DENSE, vec=oracle.pack_lower(np.linalg.cholesky(Sg)), mean=mean)
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mean, cov))
    chain = mhx.sample(model, spl, N, C, seed=5, first_chain=2)
    # separable target + ISO / DIAG proposal: the cooperative kernel evaluates the ratio (one lane per chain at d = 3); a dense
    # proposal: the state-in-HBM kernel
    assert chain.stats["kernel_variant"] == (0 if kind == "dense" else 4) and chain.stats["reduce_lanes"] == 1
    ref = oracle.rwmh(oracle.iso_gauss(d), oprop, oracle.schedule(N), 5, 2, C)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    long = mhx.sample(model, spl, 4000, 256, seed=6, discard_initial=500)
    v = long.value[:, :d, :].astype(np.float64)
    assert np.abs(v.mean(axis=(0, 2))).max() < 0.05 and np.abs(v.var(axis=(0, 2)) - 1).max() < 0.05


@pytest.mark.parametrize("target", ["iso", "banana", "funnel"])
@pytest.mark.parametrize("walk", soak_tail(["drift_iso", "drift_diag", "static_iso", "static_diag_mean"], 2))
@pytest.mark.parametrize("lanes", [0, 1, 4])
def test_walks_with_a_hastings_ratio_on_the_cooperative_kernel(mhx, oracle, target, walk, lanes, real):
    """Drifting random walks (src/proposal.jl:58-64,190-192) and static proposals (:9-11,66-83) on the separable catalogue
    targets run on the cooperative kernel: L lanes per chain, the two sums of the ratio reduced like the log-density (lane
    partial sums in block order + butterfly; the oracle takes the same shape).  Bit-exact, with discard / thinning, a
    continued call and setparams (q(x) of a static proposal is chain state)."""
    d, C, N = 37, 70, 14
    rng = np.random.default_rng(7)
    mean = (rng.normal(size=d) * (0.02 if walk.startswith("drift") else 0.05)).astype(np.float32).astype(np.float64)
    s0 = float(np.float32(0.3 if walk.startswith("drift") else 1.0))
    sv = (s0 * (0.9 + 0.2 * rng.random(d))).astype(np.float32).astype(np.float64)
    diag = "diag" in walk
    has_mean = walk != "static_iso"
    dist = mhx.MvNormal(mean if has_mean else mhx.zeros(d), sv ** 2 if diag else s0 * s0 * mhx.I)
    op = dict(kind=oracle.PROP_DIAG, vec=sv) if diag else dict(kind=oracle.PROP_ISO, scale=s0)
    static = walk.startswith("static")
    spl = mhx.StaticMH(dist) if static else mhx.RWMH(dist)
    spec, ot = {"iso": (mhx.IsoGaussian(d), oracle.iso_gauss(d)),
                "banana": (mhx.Banana(d, 0.03), oracle.Target(oracle.TARGET_BANANA, d, params=[0.03])),
                "funnel": (mhx.Funnel(d), oracle.Target(oracle.TARGET_FUNNEL, d))}[target]
    init = (rng.normal(size=(d, C)) * 0.5).astype(np.float32)
    run = mhx.Run(mhx.DensityModel(spec), spl, nchains=C, seed=12, first_chain=5, reduce_lanes=lanes)
    run.init(init)
    run.sample(N, 2, 3, 0)
    st = run.stats()
    L = st["reduce_lanes"]
    assert st["kernel_variant"] == 4 and (lanes == 0 or L == lanes)
    oprop = oracle.Proposal(mean=mean if has_mean else None, static=static, **op)
    ref = oracle.rwmh(ot.with_lanes(L), oprop, oracle.schedule(N, 2, 3), 12, 5, C, init=init)
    got, acc = run.samples()
    _same(got, ref["samples"], "samples (%d lanes)" % L)
    _same(acc, ref["accepted"], "accepted")
    # (a static N(0, I) proposal on the N(0, I) target is the exact independence sampler: every draw is accepted)
    assert (target != "iso" or acc[1:].mean() > 0.01) and (walk == "static_iso" or acc[1:].mean() < 1.0)
    run.sample(5, 1, 1, 0)                                   # a continued call: the same chain, 5 more states
    ref2 = oracle.rwmh(ot.with_lanes(L), oprop, oracle.schedule(5, 2 + 3 * (N - 1) + 1, 1), 12, 5, C, init=init)
    _same(run.samples()[0], ref2["samples"], "continued call")
    run.close()


# ---- StaticProposal / StaticMH: the independence sampler of src/proposal.jl:9-11,66-83 --------------------
@pytest.mark.parametrize("kind", ["iso", "diag", "dense", "dense_mean", "iso_mean"])
def test_static_proposal_bit_exact(mhx, oracle, kind, real):
    d, C, N = 5, 7, 40
    rng = np.random.default_rng(11)
    mean = None
    if kind.startswith("iso"):
        s13 = float(np.float32(1.3))                          # representable in both widths: sqrt(s13 * s13) == s13
        prop, op = mhx.MvNormal(mhx.zeros(d), s13 * s13 * mhx.I), dict(kind=oracle.PROP_ISO, scale=s13)
        if kind == "iso_mean":
            mean = rng.normal(size=d) * 0.3
            prop = mhx.MvNormal(mean, s13 * s13 * mhx.I)
    elif kind == "diag":
        s = (0.5 + rng.random(d)).astype(np.float32).astype(np.float64)
        prop, op = [mhx.Normal(0.0, float(v)) for v in s], dict(kind=oracle.PROP_DIAG, vec=s)
    else:
        A = rng.normal(size=(d, d)) * 0.3
        Sig = A @ A.T + np.eye(d)
        if kind == "dense_mean":
            mean = rng.normal(size=d) * 0.3
        prop = mhx.MvNormal(mhx.zeros(d) if mean is None else mean, Sig)
        op = dict(kind=oracle.PROP_DENSE, vec=oracle.pack_lower(np.linalg.cholesky(Sig)))
    Sig_t = cases.sigma_ar1(d, 0.5)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig_t))
    spl = mhx.StaticMH(prop)
    for init in (None, (rng.normal(size=(d, C)) * 0.5).astype(np.float32)):
        chain = mhx.sample(model, spl, N, C, seed=5, first_chain=3, initial_params=init, discard_initial=2, thinning=3)
        ref = oracle.rwmh(oracle.corr_gauss_from_cov(Sig_t), oracle.Proposal(mean=mean, static=True, **op),
                          oracle.schedule(N, 2, 3), 5, 3, C, init=init)
        assert np.array_equal(cases.bits(chain.value), cases.bits(ref["samples"]))
        assert np.array_equal(chain.accepted, ref["accepted"])
        assert 0.02 < chain.accepted[1:].mean() < 0.98


def test_static_mh_posterior_of_the_readme_model(mhx, real):
    """test/runtests.jl:56-74 (StaticMH testset): the Normal(mu, sigma) model of the README under static proposals
    [Normal(0,1), Normal(0,1)] and MvNormal(zeros(2), I); posterior mean of mu ~ mean(data), sigma ~ std(data)."""
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_normal_data.npy"))
    model = mhx.DensityModel(mhx.IIDNormal(data))
    for prop in ([mhx.Normal(0.0, 1.0), mhx.Normal(0.0, 1.0)], mhx.MvNormal(mhx.zeros(2), mhx.I)):
        chain = mhx.sample(model, mhx.StaticMH(prop), 400, 2048, seed=9, discard_initial=200)
        mu, sig = chain.value[:, 0, :].mean(), chain.value[:, 1, :].mean()
        assert abs(mu - data.mean()) < 0.1 and abs(sig - data.std()) < 0.1, (mu, sig)


def test_static_proposal_survives_setparams(mhx, oracle, real):
    """setparams!! replaces the state: the proposal's logpdf at the new state is recomputed with it."""
    d, C = 3, 4
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.StaticMH(mhx.MvNormal(mhx.zeros(d), 4.0 * mhx.I))
    run = mhx.Run(model, spl, nchains=C, seed=2)
    x0 = np.full((d, C), 0.25, dtype=np.float32)
    run.init(np.zeros((d, C), dtype=np.float32))
    run.set_params(x0)
    run.sample(6, 0, 1, 0, save=True)
    got = run.samples()[0]
    ref = oracle.rwmh(oracle.iso_gauss(d), oracle.Proposal(oracle.PROP_ISO, 2.0, static=True), oracle.schedule(6), 2, 0, C, init=x0)
    assert np.array_equal(cases.bits(got), cases.bits(ref["samples"]))
    run.close()


@pytest.mark.soak_f32
def test_runs_release_their_device_memory(mhx, real):
    """create / init / sample / diagnostics / destroy cycles over every sampler leave the free-memory count unchanged."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    d = 20
    ctx = mhx.Context(0)
    models = [mhx.DensityModel(mhx.IsoGaussian(d)), mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.5)))]

    def cycle():
        for m in models:
            for spl in (mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.3 * mhx.I)), mhx.RobustAdaptiveMetropolis(), mhx.MALA(0.1),
                        mhx.StaticMH(mhx.MvNormal(mhx.zeros(d), 2.0 * mhx.I)),
                        mhx.Ensemble(64, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))):
                ch = mhx.sample(m, spl, 20, 512, seed=1, ctx=ctx, initial_params=np.zeros(d), num_warmup=5)
                ch.state.diagnostics(max_lag=5)
                ch.state.close()

    cycle()                                              # JIT modules and the context's caches are allocated once
    before = free_bytes()
    for _ in range(5):
        cycle()
    assert before - free_bytes() < (1 << 20), "device memory leaked: %d bytes" % (before - free_bytes())


def test_bulk_and_tail_ess_match_numpy(mhx, real):
    """Rank-normalised bulk ESS and tail ESS (Vehtari et al. 2021) against a numpy / scipy restatement."""
    from scipy.stats import norm
    d, C, N = 2, 64, 600
    model = mhx.DensityModel(mhx.CorrGaussian(np.array([[1.0, 0.8], [0.8, 1.0]])))
    chain = mhx.sample(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.6 * mhx.I)), N, C, seed=12, discard_initial=300)
    got = chain.state.ess_bulk_tail(max_lag=120, ess_chains=0, split=True)
    v = chain.value.astype(np.float64)

    def ess(series, nlag):                                   # series [N][C]: multi-chain Geyer ESS on split chains
        h = series.shape[0] // 2
        s = np.concatenate([series[:h], series[h:2 * h]], axis=1)
        Nn, Cc = s.shape
        mm = s.mean(axis=0)
        xc = s - mm
        A = np.array([(xc[:Nn - k] * xc[k:]).sum() for k in range(nlag)]) / (Cc * (Nn - 1.0))
        W = s.var(axis=0, ddof=1).mean()
        varp = (Nn - 1.0) / Nn * W + mm.var(ddof=1)
        rho = 1.0 - (A[0] - A) / varp
        tau, prev = -1.0, np.inf
        for j in range(nlag // 2):
            pm = rho[2 * j] + rho[2 * j + 1]
            if pm <= 0:
                break
            pm = min(pm, prev)
            prev = pm
            tau += 2 * pm
        return Cc * Nn / tau

    for p in range(d + 1):
        x = v[:, p, :]
        S = x.size
        from scipy.stats import rankdata
        ranks = rankdata(x.ravel(), method="average")        # tied draws (rejected steps repeat a state) share a rank
        assert np.unique(x).size < S                         # the chain does contain ties
        z = norm.ppf((ranks - 0.375) / (S + 0.25)).reshape(x.shape).astype(np.float32).astype(np.float64)
        srt = np.sort(x.ravel())
        q05, q95 = srt[int(0.05 * (S - 1))], srt[int(0.95 * (S - 1))]
        want_bulk = ess(z, 120)
        want_tail = min(ess((x <= q05).astype(np.float64), 120), ess((x <= q95).astype(np.float64), 120))
        assert abs(got["ess_bulk"][p] - want_bulk) / want_bulk < 2e-3, (p, got["ess_bulk"][p], want_bulk)
        assert abs(got["ess_tail"][p] - want_tail) / want_tail < 2e-3, (p, got["ess_tail"][p], want_tail)
    # a Gaussian target: the bulk ESS agrees with the plain ESS of the same draws
    plain = chain.state.diagnostics(max_lag=120, ess_chains=0, split=True)["ess_geyer"]
    assert (np.abs(got["ess_bulk"][:d] / plain[:d] - 1) < 0.15).all()


def test_chains_summary_statistics(mhx, real):
    """chain.summarystats(): the MCMCChains table (README.md:59-63) from the device diagnostics; README model."""
    data = np.load(os.path.join(GOLD, "c1_normal_data.npy"))
    model = mhx.DensityModel(mhx.IIDNormal(data))
    chain = mhx.sample(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.05 * mhx.I)), 2000, 64, seed=1, discard_initial=500,
                       param_names=["mu", "sigma"], initial_params=np.array([0.0, 1.0]))
    st = chain.summarystats()
    assert st["parameters"] == ["mu", "sigma"]
    assert abs(st["mean"][0] - data.mean()) < 0.05 and abs(st["mean"][1] - data.std()) < 0.05
    v = chain.value.astype(np.float64)
    assert np.allclose(st["std"], v[:, :2, :].std(axis=(0, 2)), rtol=0.02)
    assert (st["rhat"] < 1.05).all() and (st["ess_bulk"] > 1000).all() and (st["ess_tail"] > 500).all()
    text = repr(chain)
    assert "ess_bulk" in text and "mu" in text and "sigma" in text


def test_readme_example_runs():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "readme_model.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ess_bulk" in out.stdout and "sigma" in out.stdout


@pytest.mark.parametrize("kind", ["rwmh_coop", "rwmh_dense", "rwmh_static", "emcee_coop", "emcee_user", "ram", "ram_deferred", "mala"])
def test_checkpoint_and_resume(mhx, kind, real):
    """mhx_run_save_state / mhx_run_load_state: a NEW run (created with another seed) that loads the blob continues the
    saved run bit for bit -- samples, accept flags, final state, RAM factors."""
    d, C = (40, 70) if kind != "emcee_user" else (6, 30)
    Sig = cases.sigma_ar1(d, 0.5)
    init = None
    if kind == "rwmh_coop":
        model, spl = mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.1 * mhx.I))
    elif kind == "rwmh_dense":
        model, spl = mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.05 * Sig))
    elif kind == "rwmh_static":
        model, spl = mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.StaticMH(mhx.MvNormal(mhx.zeros(d), 1.2 * mhx.I))
    elif kind == "emcee_coop":
        model, spl = mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(C, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
        init = cases.emcee_init(d, C, 3)
    elif kind == "emcee_user":
        data = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
        model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
        spl = mhx.Ensemble(C, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
        init = cases.emcee_init(d, C, 3)
    elif kind in ("ram", "ram_deferred"):                    # (the deferred form folds its pending updates at the end of every call: the blob is whole)
        model, spl = mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RobustAdaptiveMetropolis(deferred_factor=kind == "ram_deferred")
        init = np.zeros((d, C), dtype=np.float32)
    else:
        model, spl = mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.MALA(0.05)
        init = np.zeros((d, C), dtype=np.float32)
    warm = 30 if kind.startswith("ram") else 0               # the warm-up spans the checkpoint

    def new_run(seed):
        # (an Ensemble run: nchains counts ENSEMBLES, README.md:135-148 -- one here)
        return mhx.Run(model, spl, nchains=1 if kind.startswith("emcee") else C, seed=seed, first_chain=4)

    a = new_run(11)
    a.init(init)
    a.sample(5, 2, 1, warm)
    blob = a.save_state()
    a.sample(7, 1, 2, max(0, warm - 6))
    want, want_acc = a.samples()
    want_state = a.state()
    b = new_run(999)                                         # another seed: the blob's takes over
    b.load_state(blob)
    b.sample(7, 1, 2, max(0, warm - 6))
    got, got_acc = b.samples()
    _same(got, want, "samples after the checkpoint")
    _same(got_acc, want_acc, "accept flags")
    for u, v, what in zip(b.state(), want_state, ("x", "lp", "accept counts")):
        _same(u, v, what)
    if kind.startswith("ram"):
        _same(b.factor()[0], a.factor()[0], "factors")
        _same(b.diag_range()[0], a.diag_range()[0], "diag min")
        other = mhx.Run(model, mhx.RobustAdaptiveMetropolis(deferred_factor=kind == "ram"), nchains=C, seed=1)
        with pytest.raises(mhx.ArgumentError):               # the other form of RAM rounds differently: its blob is refused
            other.load_state(blob)
        other.close()
    with pytest.raises(mhx.ArgumentError):                   # a blob of another shape is refused
        other = mhx.Run(mhx.DensityModel(mhx.IsoGaussian(3)), mhx.RWMH(3), nchains=2, seed=1)
        other.load_state(blob)
    if kind == "rwmh_coop":                                  # same sizes, another configuration: refused as well
        for other in (mhx.Run(model, spl, nchains=C, seed=1, reduce_lanes=1),
                      mhx.Run(model, mhx.StaticMH(mhx.MvNormal(mhx.zeros(d), mhx.I)), nchains=C, seed=1),
                      mhx.Run(mhx.DensityModel(mhx.Funnel(d)), spl, nchains=C, seed=1)):
            with pytest.raises(mhx.ArgumentError):
                other.load_state(blob)
            other.close()
    a.close()
    b.close()


def test_bench_two_ranks_on_this_box_and_the_line_says_what_ran():
    """`python bench.py --gpus 2` as the driver would issue it for N = 2, on a box with ONE GPU: refused (a 2-GPU line needs two
    devices); with --allow-gloo the launcher starts two ranks that share the device and the line reports exactly that -- n_ranks 2,
    n_gpus = the distinct devices opened, `oversubscribed` -- with both ranks' chains in `value` (the flag used to be
    parsed and ignored, so an 8-GPU scaling run would have recorded eight 1-GPU lines)."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    ndev = torch.cuda.device_count()
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--inner", "20", "--chains", "4096",
            "--no-cpu-baseline"]
    if ndev < 2:
        p = subprocess.run(base + ["--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode != 0 and "refusing" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = subprocess.run(base + ["--gpus", "2", "--allow-gloo"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_ranks"] == 2 and line["n_gpus"] == min(2, ndev) and line["config"]["ranks_reported_by_transport"] == 2
    assert ("oversubscribed" in line) == (ndev < 2)
    assert line["config"]["units_per_step_per_gpu"] == 4096 * 20 and line["value"] > 0
    assert abs(line["value"] - 2 * 4096 * 20 * 2 / (line["ms_per_step"] * 1e-3 * 2)) < 1e-6 * line["value"]   # both ranks' chains over the max-over-ranks time
    assert 0.1 < line["acceptance_rate"] < 0.4
    assert len(lines[0]) < 8000
