"""GPU parity of the ZIGGURAT normal generator (MHX_FLAG_ZIGGURAT, arithmetic spec 3.11; fp64 engine, and since round 6 the fp32
engine's cooperative kernel): the cooperative RWMH
kernel -- fast path for every lane, the rare candidates that leave their rectangles gathered per wave-step and finished by as many
lanes side by side -- against the oracle's one-normal-at-a-time restatement (oracle.Proposal(normal_gen=1)), bit for bit.
Reference behaviour under test: src/mh-core.jl:76-117 with `randn` replaced by the spec's generator (Julia's own randn is a
ziggurat as well)."""
import numpy as np
import pytest

import cases
from conftest import soak_tail
import user_targets

pytestmark = pytest.mark.gpu

S = float(np.float32(0.238))


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, what
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s: %r vs %r" % (what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])])


@pytest.fixture
def f64(mhx, oracle):
    old_m, old_o = mhx.get_default_dtype(), oracle.get_dtype()
    mhx.set_default_dtype("f64")
    oracle.set_dtype("f64")
    yield
    mhx.set_default_dtype(old_m)
    oracle.set_dtype(old_o)


@pytest.fixture(params=["f64", "f32"])
def width(request, mhx, oracle):
    """the ziggurat exists in both widths since round 6 (fp32: 256 layers, one Philox word per normal)"""
    old_m, old_o = mhx.get_default_dtype(), oracle.get_dtype()
    mhx.set_default_dtype(request.param)
    oracle.set_dtype(request.param)
    yield request.param
    mhx.set_default_dtype(old_m)
    oracle.set_dtype(old_o)


@pytest.mark.parametrize("lanes", [0, 1, 2, 4, 8])
@pytest.mark.parametrize("d,C,N", [(100, 130, 60), (7, 64, 40), (33, 257, 25), (2, 5, 64), (52, 1000, 12)])
def test_iso_gauss_ziggurat_bit_exact(mhx, oracle, width, d, C, N, lanes):
    if lanes > 1 and lanes > (d + 3) // 4:
        pytest.skip("more lanes than Philox blocks")
    if lanes and -(-((d + 3) // 4) // lanes) > 13:
        pytest.skip("more blocks per lane than the cooperative kernel holds")
    seed = 0xABCD + d
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), S * S * mhx.I))
    chain = mhx.sample(model, spl, N, C, seed=seed, first_chain=11, reduce_lanes=lanes, normal_gen="ziggurat")
    st = chain.stats
    assert st["normal_gen"] == 1 and st["kernel_variant"] in (3, 4)
    L = st["reduce_lanes"]
    ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, S, normal_gen=1), oracle.schedule(N), seed, 11, C)
    _same(chain.value, ref["samples"], "samples")                 # incl. sample 1: the initial draw comes from the ziggurat too
    _same(chain.accepted, ref["accepted"], "accepted")
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], "final x")
    _same(cnt, ref["accept_counts"], "accept counts")
    # and it is a different chain from the Box-Muller one of the same seed
    bm = mhx.sample(model, spl, 3, C, seed=seed, first_chain=11, reduce_lanes=L)
    assert bm.stats["normal_gen"] == 0 and not np.array_equal(bm.value[:3], chain.value[:3])


def test_c2_shape_hits_the_prebuilt_ziggurat_kernel_and_the_slow_paths(mhx, oracle, width):
    """65 536-chain shape of the headline on a subset of chains, long enough that every branch of the generator is taken many
    times (wedges: 0.4 % of the draws; tails beyond r = 4.04: 5e-5): 256 chains x 400 transitions x 100 normals = 1e7 draws."""
    d, C, N = 100, 256, 401
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), S * S * mhx.I))
    chain = mhx.sample(model, spl, N, C, seed=0xC0FFEE, reduce_lanes=2, normal_gen="ziggurat")
    assert chain.stats["kernel_variant"] == 3 and chain.stats["normal_gen"] == 1
    ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=2), oracle.Proposal(oracle.PROP_ISO, S, normal_gen=1), oracle.schedule(N), 0xC0FFEE, 0, C)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert 0.15 < chain.accepted[1:].mean() < 0.35


@pytest.mark.parametrize("target", ["funnel", "banana"])
def test_separable_targets_diag_proposal_and_moments(mhx, oracle, width, target):
    d, C, N = 1000, 96, 9
    tm = mhx.Funnel(d) if target == "funnel" else mhx.Banana(d, 0.03)
    ot = oracle.Target(oracle.TARGET_FUNNEL, d) if target == "funnel" else oracle.Target(oracle.TARGET_BANANA, d, params=[0.03])
    sig = np.linspace(0.05, 0.1, d)
    run = mhx.Run(mhx.DensityModel(tm), mhx.RWMH([mhx.Normal(0, s) for s in sig]), nchains=C, seed=5, first_chain=3, normal_gen="ziggurat")
    run.init(None)
    run.sample(N, 2, 3, 0)
    val, acc = run.samples()
    L = run.stats()["reduce_lanes"]
    ot.c.reduce_lanes = L
    ref = oracle.rwmh(ot, oracle.Proposal(oracle.PROP_DIAG, vec=sig, normal_gen=1), oracle.schedule(N, 2, 3), 5, 3, C)
    _same(val, ref["samples"], "samples")
    _same(acc, ref["accepted"], "accepted")
    # running moments of the continuation == moments of the oracle's samples of the same transitions
    run.sample(6, 1, 2, 0, save="moments")
    dg = run.diagnostics()
    assert np.isfinite(dg["mean"]).all()


def test_walks_with_a_hastings_ratio(mhx, oracle, width):
    d, C, N = 20, 70, 30
    mu = np.linspace(-0.05, 0.05, d)
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    for static in (False, True):
        prop = mhx.MvNormal(mu, 0.04 * mhx.I) if not static else mhx.MvNormal(mu, 1.2 * mhx.I)
        spl = mhx.MetropolisHastings(mhx.StaticProposal(prop) if static else mhx.RandomWalkProposal(prop))
        chain = mhx.sample(model, spl, N, C, seed=8, normal_gen="ziggurat", initial_params=np.full(d, 0.1))
        L = chain.stats["reduce_lanes"]
        assert chain.stats["normal_gen"] == 1
        sc = 0.2 if not static else float(np.sqrt(1.2))
        ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, sc, mean=mu, static=static, normal_gen=1),
                          oracle.schedule(N), 8, 0, C, init=np.full((d, C), 0.1))
        _same(chain.value, ref["samples"], "samples (static=%s)" % static)
        _same(chain.accepted, ref["accepted"], "accepted")


def test_where_the_ziggurat_does_not_exist(mhx, f64):
    d = 24
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.04 * mhx.I))
    with pytest.raises(mhx.ArgumentError, match="ziggurat"):                 # dense Gaussian target: matrix-core / dense kernels
        mhx.Run(mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.5))), spl, nchains=64, normal_gen="ziggurat")
    # fp32 (round 6): the cooperative kernel and the register kernel (a user's source) have the form as in fp64
    for model in (mhx.DensityModel(mhx.IsoGaussian(d)),
                  mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=np.zeros(2 * d, np.float32) + 1))):
        ok = mhx.Run(model, spl, nchains=64, normal_gen="ziggurat", dtype="f32")
        ok.init(None)
        ok.sample(3)
        assert ok.stats()["normal_gen"] == 1 and ok.stats()["dtype"] == "f32"
    with pytest.raises(mhx.ArgumentError, match="ziggurat"):                 # ... and the dense target is refused there too
        mhx.Run(mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.5))), spl, nchains=64, normal_gen="ziggurat", dtype="f32")
    with pytest.raises(mhx.ArgumentError):                                    # forced generic kernel
        mhx.Run(mhx.DensityModel(mhx.IsoGaussian(d)), spl, nchains=64, normal_gen="ziggurat", flags=mhx.FLAG_GENERIC)


def test_device_normals_pass_distribution_checks(mhx, width):
    """The device's own draws (the initial draw of 4096 chains x d = 1000 from N(0, I): 4.1e6 normals) against the normal law."""
    import scipy.stats as st
    d, C = 1000, 4096
    run = mhx.Run(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), mhx.I)), nchains=C, seed=77, normal_gen="ziggurat")
    run.init(None)
    x = run.state()[0].ravel().astype(np.float64)
    n = x.size
    assert abs(x.mean()) < 5 / np.sqrt(n) and abs(x.var() - 1) < 5 * np.sqrt(2.0 / n)
    assert abs((x ** 4).mean() - 3) < 5 * np.sqrt(96.0 / n)
    assert st.kstest(x[::7], "norm").pvalue > 1e-4
    for t in (2.0, 3.0, 4.0388498461095045 if width == "f64" else 3.65415288536):
        e = n * 2 * st.norm.sf(t)
        assert abs((np.abs(x) > t).sum() - e) < 5 * np.sqrt(e) + 1, t


@pytest.mark.parametrize("every", soak_tail([3, 7], 1))
def test_fixup_queue_windows(mhx, oracle, width, tools_engine, every):
    """More than 64 candidates of one wave-step in the fix-up queue (never seen in practice: a dozen fail) -- forced by the
    ZIG_FORCE_FAIL test hook of the TOOLS build, which sends every n-th slot through the queue although its candidate is inside its
    rectangle; the refinement re-derives the same normal, so the chains must still equal the oracle's bit for bit.  (A hook taints
    its context: mhx_stats.tainted = 1, and `sample` builds a Chains from such a run only when told to.)"""
    tools_engine.set("ZIG_FORCE_FAIL", str(every))
    tools_engine.set("NO_PREBUILT", "1")
    for d, C, N, lanes in [(100, 70, 12, 2), (100, 33, 9, 4), (1000, 5, 6, 64), (13, 200, 10, 1)]:
        model = mhx.DensityModel(mhx.IsoGaussian(d))
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), S * S * mhx.I))
        with pytest.raises(mhx.MhxError, match="tainted"):
            mhx.sample(model, spl, 2, 4, seed=1, reduce_lanes=lanes, normal_gen="ziggurat")
        chain = mhx.sample(model, spl, N, C, seed=77 + d, first_chain=2, reduce_lanes=lanes, normal_gen="ziggurat", allow_tainted=True)
        assert chain.stats["kernel_variant"] == 4 and chain.stats["normal_gen"] == 1 and chain.stats["tainted"] == 1
        L = chain.stats["reduce_lanes"]
        ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, S, normal_gen=1), oracle.schedule(N), 77 + d, 2, C)
        _same(chain.value, ref["samples"], "samples d=%d lanes=%d" % (d, lanes))
        _same(chain.accepted, ref["accepted"], "accepted")


def test_the_release_library_has_no_probes_and_reads_no_tuning_variable(mhx, oracle, f64, monkeypatch):
    """libmhx.so: a probe / hook name is an unknown option, a run is never tainted, and the environment variables that steered the
    kernels in earlier rounds do nothing (here: MHX_NO_PREBUILT=1 would have replaced the pre-built headline kernel, variant 3)."""
    ctx = mhx.Context.default()
    for name in ("ZIG_PROBE", "EMCEE_PROBE", "ZIG_FORCE_FAIL", "FAULT_SLAB", "JIT_DEFS", "EMCEE_STAMPS", "NOT_AN_OPTION"):
        with pytest.raises(mhx.ArgumentError, match="unknown option"):
            ctx.set_option(name, "1")
    monkeypatch.setenv("MHX_NO_PREBUILT", "1")
    monkeypatch.setenv("MHX_ZIG_PROBE", "1")
    d = 100
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), S * S * mhx.I))
    chain = mhx.sample(model, spl, 6, 64, seed=3, reduce_lanes=2, normal_gen="ziggurat")
    assert chain.stats["kernel_variant"] == 3 and chain.stats["tainted"] == 0
    ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=2), oracle.Proposal(oracle.PROP_ISO, S, normal_gen=1), oracle.schedule(6), 3, 0, 64)
    _same(chain.value, ref["samples"], "samples")
    ctx.set_option("NO_PREBUILT", "1")                      # the explicit form of the same wish
    try:
        assert ctx.get_option("NO_PREBUILT") == "1"
        chain = mhx.sample(model, spl, 6, 64, seed=3, reduce_lanes=2, normal_gen="ziggurat")
        assert chain.stats["kernel_variant"] == 4 and chain.stats["tainted"] == 0
        _same(chain.value, ref["samples"], "samples, hiprtc-specialised")
    finally:
        ctx.set_option("NO_PREBUILT", None)


def test_jit_defs_of_the_tools_build_reach_hiprtc_and_change_no_bit(mhx, oracle, f64, tools_engine, tmp_path, monkeypatch):
    """Option JIT_DEFS of the tools build hands extra defines to every hiprtc compile (how the A/B scripts reach the kernels'
    compile-time parameters): plain instead of non-temporal record stores are data movement only -- the oracle's chains bit for bit;
    and a define that cannot compile must fail the run's creation (what makes the first half a test), in a fresh cache directory."""
    tools_engine.set("JIT_DEFS", "MHX_REC_STORE_AUX=0")
    tools_engine.set("NO_PREBUILT", "1")
    for d, C, N, lanes in [(100, 70, 9, 2), (98, 64, 6, 4)]:
        model = mhx.DensityModel(mhx.IsoGaussian(d))
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), S * S * mhx.I))
        chain = mhx.sample(model, spl, N, C, seed=31 + d, first_chain=5, reduce_lanes=lanes, normal_gen="ziggurat", allow_tainted=True)
        assert chain.stats["kernel_variant"] == 4 and chain.stats["normal_gen"] == 1 and chain.stats["tainted"] == 1
        L = chain.stats["reduce_lanes"]
        ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, S, normal_gen=1), oracle.schedule(N), 31 + d, 5, C)
        _same(chain.value, ref["samples"], "samples d=%d lanes=%d" % (d, lanes))
        _same(chain.accepted, ref["accepted"], "accepted")
    monkeypatch.setenv("MHX_CACHE_DIR", str(tmp_path / "jit"))
    tools_engine.set("JIT_DEFS", "MHX_REC_STORE_AUX=)")
    model = mhx.DensityModel(mhx.IsoGaussian(100))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(100), S * S * mhx.I))
    with pytest.raises(Exception) as ei:
        mhx.sample(model, spl, 4, 64, seed=1, reduce_lanes=2, normal_gen="ziggurat", allow_tainted=True)
    assert "hiprtc" in str(ei.value)


@pytest.mark.parametrize("d,C,prop", soak_tail([(5, 70, "iso"), (100, 130, "iso"), (130, 40, "iso"), (64, 33, "diag"), (100, 64, "diag"), (160, 65, "iso")], 3))
def test_ziggurat_on_the_register_kernel_with_a_user_log_density(mhx, oracle, width, d, C, prop):
    """normal_gen="ziggurat" on the lane-per-chain register kernel (any target; here a user's HIP source): fast-path normals straight
    into the candidate's registers, the wave-step's failures queued, refined side by side and handed back to their owners' registers.
    Same chains as the oracle at reduction shape 1 -- chains that do not fill a wave (idle lanes shadow the last chain), the state's
    tail in LDS above 64 dimensions, one / two / three words of failure mask, thinning with a discarded prefix, the state after.
    Both widths since round 6 (fp32: a Philox block serves four normals, the signed pair table, failures noted by add-with-carry)."""
    rng = np.random.default_rng(1000 + d)
    data = np.concatenate([rng.normal(size=d), 0.5 + rng.random(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    ut = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    if prop == "iso":
        s = float(np.float32(2.38 / d ** 0.5))
        spl, op = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), oracle.Proposal(oracle.PROP_ISO, s, normal_gen=1)
    else:
        sv = (np.float32(2.38 / d ** 0.5) * (0.5 + rng.random(d))).astype(np.float32)
        spl, op = mhx.RWMH([mhx.Normal(0.0, float(v)) for v in sv]), oracle.Proposal(oracle.PROP_DIAG, vec=sv, normal_gen=1)
    init = rng.normal(size=(d, C))
    r = mhx.Run(model, spl, nchains=C, seed=91, first_chain=7, normal_gen="ziggurat")
    r.init(init)
    r.sample(40, 4, 3, 0)
    got, got_acc = r.samples()
    st_ = r.stats()
    assert st_["kernel_variant"] == 2 and st_["normal_gen"] == 1
    ref = oracle.rwmh(ut, op, oracle.schedule(40, 4, 3), 91, 7, C, init=init)
    _same(got, ref["samples"], "samples")
    _same(got_acc, ref["accepted"], "accepted")
    x, lp, cnt = r.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")


def test_ziggurat_register_kernel_draws_its_own_start(mhx, oracle, width):
    """(no initial_params: the start is a bare proposal draw by the same generator, src/proposal.jl:41-47)"""
    d, C = 24, 100
    data = np.concatenate([np.zeros(d), np.ones(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    ut = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    s = float(np.float32(0.5))
    chain = mhx.sample(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), 30, C, seed=5, normal_gen="ziggurat")
    assert chain.stats["kernel_variant"] == 2 and chain.stats["normal_gen"] == 1
    ref = oracle.rwmh(ut, oracle.Proposal(oracle.PROP_ISO, s, normal_gen=1), oracle.schedule(30), 5, 0, C)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
