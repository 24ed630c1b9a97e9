"""GPU parity: MANY independent ensembles in one run (mhx_emcee_cfg.n_ensembles) -- what the reference's
`sample(model, Ensemble(W, ..), MCMCThreads(), N, nchains)` is (README.md:135-148: nchains ENSEMBLES, one task each; test/emcee.jl:24:
1 000 walkers).  Ensemble e of a run carries id first + e in its RNG counters and must be, bit for bit, the oracle's ensemble of that
id (src/emcee.jl:14-58, :70-102) -- on every kernel form the stretch move has: the persistent block (a CU per ensemble), the
lane-per-walker half-step and sweep launches, the run-time-dimension kernel, the lane-group / scalar-factor / matrix-core forms of a
dense-Gaussian target, and the reference's own sequential sweep.  Layout: ensemble e = columns e W .. e W + W - 1 of the chain axis."""
import numpy as np
import pytest

import cases
import user_targets

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, "%s: %s %s / %s %s" % (what, a.shape, a.dtype, b.shape, b.dtype)
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s" % (what, len(bad), bad[0])


def _rotated(d, rho=0.9):
    q, _ = np.linalg.qr(np.random.default_rng(42).normal(size=(d, d)))
    return q @ cases.sigma_ar1(d, rho) @ q.T


def _model(mhx, oracle, kind, d, lanes):
    if kind == "user":
        data = np.concatenate([np.linspace(-1.0, 1.0, d), np.linspace(0.5, 2.0, d)]).astype(np.float32)
        return (mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data)),
                lambda L: user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data))
    Sig = cases.sigma_ar1(d, 0.9) if kind == "band" else _rotated(d)
    return mhx.DensityModel(mhx.CorrGaussian(Sig)), lambda L: oracle.corr_gauss_from_cov(Sig, reduce_lanes=L)


FORMS = [
    # name, target kind, d, W, reduce_lanes, flags, options, expected variant
    ("persistent_block", "user", 5, 1000, 1, 0, {}, 6),
    ("persistent_block_odd", "user", 2, 37, 1, 0, {}, 6),
    ("half_step_launches", "user", 5, 130, 1, 0, {"EMCEE_PERSIST": "0", "EMCEE_FUSED": "0"}, 2),
    ("sweep_launches", "user", 7, 1500, 1, 0, {}, 2),
    ("run_time_dimension", "band", 6, 70, 1, "generic", {}, 0),
    ("lane_group_band", "band", 50, 130, 0, 0, {}, 4),
    ("lane_group_band_halves", "band", 17, 66, 4, 0, {"EMCEE_FUSED": "0"}, 4),
    ("scalar_factor", "dense", 50, 131, 0, 0, {"EMCEE_MFMA": "0"}, 9),
    ("matrix_core", "dense", 24, 71, 0, 0, {"EMCEE_MFMA": "1"}, 10),
    ("matrix_core_halves", "dense", 33, 40, 0, 0, {"EMCEE_MFMA": "1", "EMCEE_FUSED": "0"}, 10),
    ("reference_sequential", "band", 3, 21, 1, "sequential", {}, 7),
]


@pytest.mark.parametrize("E", [3, 1])
@pytest.mark.parametrize("form", FORMS, ids=[f[0] for f in FORMS])
def test_every_ensemble_of_a_run_is_the_oracles_ensemble_of_its_id(mhx, oracle, real, engine, form, E):
    name, kind, d, W, lanes, fl, opts, variant = form
    for k, v in opts.items():
        engine.set(k, v)
    flags = {0: 0, "generic": mhx.FLAG_GENERIC, "sequential": mhx.FLAG_EMCEE_SEQUENTIAL}[fl]
    model, mk_target = _model(mhx, oracle, kind, d, lanes)
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    init = cases.emcee_init(d, W * E, 5)                                   # [dim][E W]: ensemble e = columns e W ..
    first, seed, N, di, th = 7, 21, 6, 2, 3
    run = mhx.Run(model, spl, nchains=E, seed=seed, first_chain=first, flags=flags, reduce_lanes=lanes)
    assert run.n == E * W
    run.init(init)
    run.sample(N, di, th, 0)
    val, acc = run.samples()
    st = run.stats()
    assert st["kernel_variant"] == variant, (name, st)
    L = st["reduce_lanes"]
    x, lp, cnt = run.state()
    run.sample(3, 0, 1, 0)                                                # a continued call: slot 0 = the state the first call left
    val2, acc2 = run.samples()
    tgt = mk_target(L)
    mode = 0 if fl == "sequential" else 1
    total_acc = 0
    for e in range(E):
        cols = slice(e * W, (e + 1) * W)
        ref = oracle.emcee(tgt, 2.0, mode, oracle.schedule(N, di, th), seed, first + e, W, init[:, cols])
        _same(val[:, :, cols], ref["samples"], "%s: samples of ensemble %d" % (name, e))
        _same(acc[:, cols], ref["accepted"], "accepted of ensemble %d" % e)
        _same(x[:, cols], ref["final_x"], "state of ensemble %d" % e)
        _same(lp[cols], ref["final_lp"], "lp of ensemble %d" % e)
        _same(cnt[cols], ref["accept_counts"], "counters of ensemble %d" % e)
        total_acc += int(ref["accept_counts"].sum())
        nT = di + (N - 1) * th
        cont = oracle.emcee(tgt, 2.0, mode, oracle.schedule(3, nT, 1), seed, first + e, W, init[:, cols])
        _same(val2[:, :, cols], cont["samples"], "continued call, ensemble %d" % e)
    assert st["accepted"] == total_acc and st["transitions"] == (di + (N - 1) * th) * W * E
    run.close()


@pytest.mark.parametrize("compact", ["1", "0"])
def test_many_ensembles_drawn_on_the_device_and_streamed_to_the_host(mhx, oracle, real, engine, compact):
    """the reference's own test size (test/emcee.jl:24: 1000 walkers) x 24 ensembles: initial walkers drawn on the device from the
    prior StretchProposal wraps (every ensemble from ITS Philox stream), the whole schedule in ONE launch of 24 persistent blocks, the
    samples streamed to the host (accept-compacted and plain); ensemble e == the oracle's ensemble first + e"""
    engine.set("HOST_COMPACT", compact)
    d, W, E, N = 2, 1000, 24, 12
    data = np.concatenate([np.linspace(-1.0, 1.0, d), np.linspace(0.5, 2.0, d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    tgt = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    run = mhx.Run(model, spl, nchains=E, seed=3, first_chain=100, reduce_lanes=1)
    run.init(None)
    val, acc = run.sample_to_host(N, 0, 1, 0, slab_samples=5)
    st = run.stats()
    assert st["kernel_variant"] == 6 and st["launches"] == 3              # one launch per slab, 24 blocks each
    assert run.host_stats()["compact"] == int(compact)
    for e in (0, 1, 11, 23):
        cols = slice(e * W, (e + 1) * W)
        ref = oracle.emcee(tgt, 2.0, 1, oracle.schedule(N), 3, 100 + e, W, None, prior=oracle.Proposal(oracle.PROP_ISO, 1.0))
        _same(val[:, :, cols], ref["samples"], "ensemble %d" % e)
        _same(acc[:, cols], ref["accepted"], "accepted %d" % e)
    # distinct ensembles are distinct chains
    assert not np.array_equal(val[:, :, :W], val[:, :, W:2 * W])
    run.close()


def test_sample_with_nchains_runs_that_many_ensembles(mhx, oracle, real):
    """`sample(model, Ensemble(W, proposal), MCMCThreads(), N, nchains)` (README.md:135-148): nchains ensembles, walkers side by side
    in the chain axis -- and the analytic posterior of test/emcee.jl:25-26 holds for every one of them"""
    import traced_models as M
    W, E, N = 1000, 6, 1000
    model = mhx.DensityModel(M.nig, dim=2)
    spl = mhx.Ensemble(W, mhx.StretchProposal([mhx.InverseGamma(2, 3), mhx.Normal(0, 1)]))
    chain = mhx.sample(model, spl, mhx.MCMCThreads(), N, E, seed=100, param_names=["s", "m"])
    assert chain.value.shape == (N, 3, E * W) and chain.stats["kernel_variant"] == 6
    v = chain.value.astype(np.float64)
    for e in range(E):
        s = v[:, 0, e * W:(e + 1) * W].mean()
        m = v[:, 1, e * W:(e + 1) * W].mean()
        assert abs(s - 49 / 24) < 0.1 and abs(m - 7 / 6) < 0.1, (e, s, m)
    one = mhx.sample(model, spl, 50, seed=100, param_names=["s", "m"])
    _same(chain.value[:50, :, :W], one.value, "ensemble 0 of the many == the single-ensemble call")


def test_checkpoint_and_argument_checks(mhx, real):
    d, W, E = 4, 40, 3
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    a = mhx.Run(model, spl, nchains=E, seed=1)
    a.init(None)
    a.sample(5, 0, 1, 0)
    blob = a.save_state()
    a.sample(4, 1, 1, 0)
    want, _ = a.samples()
    b = mhx.Run(model, spl, nchains=E, seed=99)
    b.load_state(blob)
    b.sample(4, 1, 1, 0)
    _same(b.samples()[0], want, "resumed many-ensemble run")
    # 3 x 40 walkers are not 1 x 120: the blob names its configuration
    c = mhx.Run(model, mhx.Ensemble(W * E, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=1)
    with pytest.raises(mhx.ArgumentError):
        c.load_state(blob)
    # the sharded-ensemble building blocks are for ONE ensemble
    assert mhx.lib().mhx_emcee_half_step(a.h, 0, 0, 1) == mhx.MHX_EINVAL
    with pytest.raises(mhx.ArgumentError):
        mhx.Run(model, spl, nchains=70000, seed=1)
    with pytest.raises(mhx.ArgumentError):
        mhx.Run(model, spl, nchains=2, seed=1, first_chain=2 ** 32 - 1)
