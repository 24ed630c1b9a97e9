"""GPU parity of the DEFERRED-FACTOR form of RobustAdaptiveMetropolis (MHX_FLAG_RAM_DEFERRED, kernel variant 12; arithmetic spec
DESIGN.md 3.12) against its own oracle twin (oracle/mhx_oracle.c orc_ram_deferred), bit for bit; and against the reference-order
arithmetic (orc_ram = src/RobustAdaptiveMetropolis.jl:123-278 with the sequential lowrankupdate! sweep) within rounding."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    assert a.dtype == b.dtype, "%s: dtypes %s / %s" % (what, a.dtype, b.dtype)
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s: %r vs %r" % (
        what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])])


def oracle_unpack(p, d):
    from oracle import oracle as O
    return O.unpack_lower(p, d)


def _moving_start(d, C, seed):
    """a start on the target's scale and S0 = 2.38 / sqrt(d) I: the chain moves from the first step (the specified start x0 = 0,
    S0 = I sits at acceptance ~0 for O(1e5) steps at d = 200: only downdates)"""
    rng = np.random.default_rng(seed)
    return rng.normal(size=(d, C)), np.eye(d) * (2.38 / np.sqrt(d))


def _run(mhx, model, spl, N, C, seed, first, init, **kw):
    chain = mhx.sample(model, spl, N, C, seed=seed, first_chain=first, initial_params=init, **kw)
    S, st = chain.state.factor()
    lo, hi = chain.state.diag_range()
    x, lp, cnt = chain.state.state()
    return chain, S, st, lo, hi, x, lp, cnt


def _check(chain, S, st, lo, hi, x, lp, cnt, ref):
    _same(chain.accepted, ref["accepted"], "accepted")
    _same(chain.value, ref["samples"], "samples")
    _same(S, ref["S"], "S")
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
    _same(lo, ref["diag_min"], "diag min")
    _same(hi, ref["diag_max"], "diag max")
    _same(st, ref["status"], "status")


@pytest.mark.parametrize("d,C,N,warm", [(4, 6, 40, 29), (2, 9, 50, 50), (70, 5, 30, 21), (130, 4, 22, 22), (200, 3, 21, 19), (256, 2, 12, 12)])
def test_deferred_bit_exact(mhx, oracle, d, C, N, warm, real):
    """every kernel shape (1 .. 4 rows per lane), blocks of 8 pending updates, short tail blocks, the warm-up's end"""
    Sig = cases.sigma_ar1(d, 0.7)
    init, S0 = _moving_start(d, C, d)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.RobustAdaptiveMetropolis(S=S0, deferred_factor=True)
    out = _run(mhx, model, spl, N, C, 31, 2, init, num_warmup=warm, discard_initial=0)
    assert out[0].stats["kernel_variant"] == 12
    Sin = np.tile(oracle.pack_lower(S0), (C, 1))
    ref = oracle.ram_deferred(oracle.corr_gauss_from_cov(Sig), oracle.schedule(N, 0, 1, warm), 31, 2, C, init=init, S_in=Sin)
    assert 0.05 < ref["accepted"][1:].mean() < 0.95          # updates AND downdates
    _check(*out, ref)


def test_deferred_golden_trace(mhx, real):
    """the committed fixture of the twin (tests/golden/traces*.npz, case `ram_deferred` of tests/cases.py)"""
    import os
    tr = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traces64.npz" if real == "f64" else "traces.npz"))
    d, C = 6, 4
    init, S0 = cases.ram_deferred_setup(d, C)
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.7))), mhx.RobustAdaptiveMetropolis(S=S0, deferred_factor=True),
                       40, C, seed=33, first_chain=1, initial_params=init, num_warmup=33, discard_initial=0)
    _same(chain.value, tr["ram_deferred/samples"], "samples")
    _same(chain.accepted, tr["ram_deferred/accepted"], "accepted")
    S, _ = chain.state.factor()
    _same(S, tr["ram_deferred/S"], "S")


def test_deferred_long_run_spans_several_launches(mhx, oracle, real):
    """9 000 transitions = three launches (a flush at the end of each); adaptation stops inside the second one"""
    d, C = 3, 5
    Sig = cases.sigma_ar1(d, 0.5)
    init = np.zeros((d, C))
    out = _run(mhx, mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RobustAdaptiveMetropolis(deferred_factor=True),
               3000, C, 5, 0, init, num_warmup=6000, discard_initial=6000)
    ref = oracle.ram_deferred(oracle.corr_gauss_from_cov(Sig), oracle.schedule(3000, 6000, 1, 6000), 5, 0, C, init=init,
                              flush_at=oracle.ram_flush_points(9000))
    _check(*out, ref)


@pytest.mark.parametrize("slab", [2, 3, -2])
def test_deferred_through_the_host_return_path(mhx, oracle, real, slab):
    """mhx_run_sample_to_host cuts the schedule into slabs of saved samples and every slab ends a launch -- a forced fold of the
    pending updates (spec 3.12: the rounding depends on where launches end).  With the slab ends as the twin's fold points the record
    that arrives on the host is the twin's, bit for bit; the device-resident call of the same schedule differs in the last bits."""
    d, C = 70, 4
    N, di, th, warm = 9, 5, 3, 26
    Sig = cases.sigma_ar1(d, 0.7)
    init, S0 = _moving_start(d, C, 5)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.RobustAdaptiveMetropolis(S=S0, deferred_factor=True)
    run = mhx.Run(model, spl, nchains=C, seed=19, first_chain=3)
    run.init(init)
    val, acc = run.sample_to_host(N, di, th, warm, slab_samples=slab)
    S, _ = run.factor()
    run.close()
    K = abs(slab)
    ends = [di + (min(i0 + K, N) - 1) * th for i0 in range(0, N, K)]          # the transition behind each slab's last sample
    Sin = np.tile(oracle.pack_lower(S0), (C, 1))
    ref = oracle.ram_deferred(oracle.corr_gauss_from_cov(Sig), oracle.schedule(N, di, th, warm), 19, 3, C, init=init, S_in=Sin, flush_at=ends)
    _same(val, ref["samples"], "samples")
    _same(acc, ref["accepted"], "accepted")
    _same(S, ref["S"], "S")
    one = oracle.ram_deferred(oracle.corr_gauss_from_cov(Sig), oracle.schedule(N, di, th, warm), 19, 3, C, init=init, S_in=Sin)
    assert np.array_equal(one["accepted"], ref["accepted"]) and np.abs(one["S"] - ref["S"]).max() < (1e-4 if real == "f32" else 1e-12)


def test_deferred_iso_target_thinning_and_bounds(mhx, oracle, real):
    """a separable target, thinning, eigenvalue bounds that refuse some updates (a refused update is not pending)"""
    d, C, N = 5, 7, 40
    rng = np.random.default_rng(3)
    L = np.tril(rng.normal(size=(d, d)) * 0.2) + np.eye(d)
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RobustAdaptiveMetropolis(γ=0.7, S=L, eigenvalue_lower_bound=0.58, eigenvalue_upper_bound=1.45, deferred_factor=True)
    out = _run(mhx, model, spl, N, C, 77, 0, None, num_warmup=60, discard_initial=5, thinning=2)
    Sin = np.tile(oracle.pack_lower(L), (C, 1))
    ref = oracle.ram_deferred(oracle.iso_gauss(d), oracle.schedule(N, 5, 2, 60), 77, 0, C, init=None, S_in=Sin, gamma=0.7,
                              eig_lo=0.58, eig_hi=1.45)
    plain = oracle.ram_deferred(oracle.iso_gauss(d), oracle.schedule(N, 5, 2, 60), 77, 0, C, init=None, S_in=Sin, gamma=0.7)
    assert not np.array_equal(plain["S"], ref["S"])            # the bounds did refuse something
    _check(*out, ref)


def test_deferred_agrees_with_the_reference_order_within_rounding(mhx, oracle):
    """the same chain in exact arithmetic: S S' of the deferred form and of the reference's sequential sweeps agree to rounding, and
    so do the chains while no accept decision sits at rounding level (fp64)"""
    import mhx as m
    m.set_default_dtype("f64"); oracle.set_dtype("f64")
    d, C, N = 200, 4, 120
    Sig = cases.sigma_ar1(d, 0.7)
    init, S0 = _moving_start(d, C, 9)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    chain = mhx.sample(model, mhx.RobustAdaptiveMetropolis(S=S0, deferred_factor=True), N, C, seed=8, initial_params=init, num_warmup=N, discard_initial=0)
    S, _ = chain.state.factor()
    Sin = np.tile(oracle.pack_lower(S0), (C, 1))
    ref = oracle.ram(oracle.corr_gauss_from_cov(Sig), oracle.schedule(N, 0, 1, N), 8, 0, C, init=init, S_in=Sin)
    assert np.array_equal(chain.accepted, ref["accepted"])
    assert np.abs(chain.value - ref["samples"]).max() < 1e-11
    for c in range(C):
        A, B = oracle_unpack(S[c], d), oracle_unpack(ref["S"][c], d)
        assert np.linalg.norm(A @ A.T - B @ B.T) <= 1e-13 * np.linalg.norm(B @ B.T)


def test_deferred_user_log_density(mhx, oracle, real):
    """the hiprtc build of the same body around a user log-density"""
    import user_targets
    d, C, N = 40, 5, 30
    rng = np.random.default_rng(d)
    data = np.concatenate([rng.normal(size=d), 0.5 + rng.random(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    ut = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    init = np.zeros((d, C))
    S0 = np.eye(d) * (2.38 / np.sqrt(d))
    out = _run(mhx, model, mhx.RobustAdaptiveMetropolis(S=S0, deferred_factor=True), N, C, 12, 0, init, num_warmup=N, discard_initial=0)
    ref = oracle.ram_deferred(ut, oracle.schedule(N, 0, 1, N), 12, 0, C, init=init, S_in=np.tile(oracle.pack_lower(S0), (C, 1)))
    _check(*out, ref)


def test_deferred_refusals(mhx, real):
    with pytest.raises(mhx.ArgumentError):                       # one chain per wave
        mhx.sample(mhx.DensityModel(mhx.IsoGaussian(257)), mhx.RobustAdaptiveMetropolis(deferred_factor=True), 3, 2,
                   initial_params=np.zeros((257, 2)))
    run = mhx.Run(mhx.DensityModel(mhx.IsoGaussian(4)), mhx.RobustAdaptiveMetropolis(deferred_factor=True), nchains=3, seed=1)
    with pytest.raises(mhx.ArgumentError):                       # the factor is whole only between launches
        run.watch_factors([0])
    run.close()


def test_deferred_adaptation_reaches_alpha(mhx, real):
    """the reference's own property (test/RobustAdaptiveMetropolis.jl:11-28,36-55: the mean acceptance probability goes to alpha)
    holds for the deferred form -- the same model and bounds as test_gpu_ram's check of the sweep form; statistics, not bits"""
    d, C, N = 8, 256, 4000
    Sig = cases.sigma_ar1(d, 0.6) * np.linspace(0.5, 3.0, d)[:, None] * np.linspace(0.5, 3.0, d)[None, :]
    run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RobustAdaptiveMetropolis(deferred_factor=True), nchains=C, seed=12)
    run.init(np.zeros(d))
    run.sample(N, 0, 1, N)
    st = run.step_stats()
    val, acc = run.samples()
    p = np.exp(st["logα"][N // 2:].astype(np.float64)).mean()
    assert abs(p - 0.234) < 0.02, p
    assert abs(acc[N // 2:].mean() - 0.234) < 0.02
    S, status = run.factor()
    assert (status == 0).all() and np.isfinite(S).all()
    run.close()
