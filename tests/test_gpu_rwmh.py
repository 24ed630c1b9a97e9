"""GPU parity: HIP RWMH kernels (through the C ABI) vs the CPU oracle, bit for bit.
Reference behaviour under test: src/mh-core.jl:76-117, src/proposal.jl:41-56."""
import numpy as np
import pytest

import cases
from conftest import soak_tail
import user_targets

pytestmark = pytest.mark.gpu


def _bits(a):
    return cases.bits(np.ascontiguousarray(a))


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    assert a.dtype == b.dtype, "%s: dtypes %s / %s" % (what, a.dtype, b.dtype)
    bad = np.argwhere(_bits(a) != _bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s: %r vs %r" % (
        what, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])])


S = float(np.float32(0.238))


@pytest.mark.parametrize("flags_name,lanes", [("auto", 1), ("generic", 0), ("auto", 0), ("auto", 2), ("auto", 8)])
@pytest.mark.parametrize("d,C,N", [(100, 130, 40), (2, 5, 64), (7, 64, 33), (33, 257, 17)])
def test_iso_gauss_rwmh_bit_exact(mhx, oracle, d, C, N, flags_name, lanes, real):
    flags = mhx.FLAG_GENERIC if flags_name == "generic" else 0
    if lanes > 1 and lanes > (d + 3) // 4:
        pytest.skip("more lanes than Philox blocks")
    if lanes > 1 and -(-((d + 3) // 4) // lanes) > (13 if real == "f64" else 16):
        pytest.skip("more blocks per lane than the cooperative kernel holds in registers")
    seed = 0xC0FFEE + d
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), S * S * mhx.I))
    chain = mhx.sample(model, spl, N, C, seed=seed, first_chain=7, flags=flags, reduce_lanes=lanes)
    if lanes == 1:
        assert chain.stats["reduce_lanes"] == 1
    L = chain.stats["reduce_lanes"]
    ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, S), oracle.schedule(N), seed, 7, C)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
    assert chain.stats["accepted"] == int(ref["accept_counts"].sum())
    if flags_name == "generic":
        assert chain.stats["kernel_variant"] == 0 and L == 1
    elif lanes == 1:
        # the register kernel holds the candidate in VGPRs and the state in VGPRs + LDS: up to 320 dimensions in fp32, 160 in fp64
        assert chain.stats["kernel_variant"] in ((1, 2) if d <= (160 if real == "f64" else 320) else (0,))
    elif lanes > 1:
        assert chain.stats["kernel_variant"] in (3, 4) and L == lanes
    else:
        assert chain.stats["kernel_variant"] in (1, 2, 3, 4)


def test_schedule_discard_thinning_and_initial_params(mhx, oracle, real):
    d, C, N = 4, 70, 25
    init = np.random.default_rng(1).normal(size=(d, C)).astype(np.float32)
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), np.array([0.5, 1.0, 0.25, 2.0]) ** 2))
    chain = mhx.sample(model, spl, N, C, seed=5, initial_params=init, discard_initial=25, thinning=4, reduce_lanes=1)
    assert chain.range() == range(26, 26 + 4 * N, 4)          # test/runtests.jl:129
    prop = oracle.Proposal(oracle.PROP_DIAG, vec=np.array([0.5, 1.0, 0.25, 2.0], dtype=np.float32))
    ref = oracle.rwmh(oracle.iso_gauss(d), prop, oracle.schedule(N, 25, 4), 5, 0, C, init=init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    # first sample == initial_params when nothing is discarded (test/runtests.jl:203-213)
    chain0 = mhx.sample(model, spl, 3, C, seed=5, initial_params=init, reduce_lanes=1)
    _same(chain0.value[0, :d, :], init.astype(cases.R()), "sample 1")
    assert not chain0.accepted[0].any()


def test_parallel_sampling_call_forms(mhx, real):
    """test/runtests.jl:96-110: sample(model, spl, MCMCThreads(), 10 000, 4) -- the parallel tags are accepted and
    every form runs the chains together on the GPU; same moments check as the reference (atol 0.1)."""
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_normal_data.npy"))
    model = mhx.DensityModel(mhx.IIDNormal(data))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.01 * mhx.I))
    a = mhx.sample(model, spl, mhx.MCMCThreads(), 10000, 4, param_names=["μ", "σ"], seed=2, initial_params=np.array([0.0, 1.0]))
    b = mhx.sample(model, spl, 10000, 4, param_names=["μ", "σ"], seed=2, initial_params=np.array([0.0, 1.0]))
    assert a.value.shape == (10000, 3, 4) and np.array_equal(a.value, b.value)
    assert abs(a.mean("μ") - data.mean()) < 0.1 and abs(a.mean("σ") - 1.0) < 0.1
    c = mhx.sample(model, spl, mhx.MCMCDistributed(), 100, 4, seed=2, initial_params=np.array([0.0, 1.0]))
    assert np.array_equal(c.value, b.value[:100])


@pytest.mark.parametrize("d,C,lanes,prop", soak_tail([(100, 70, 0, "iso"), (128, 33, 16, "diag"), (70, 9, 8, "iso"), (5, 37, 2, "diag"),
                                            (50, 40, 4, "iso"), (99, 17, 32, "diag"), (100, 21, 0, "dense"), (37, 66, 4, "dense"),
                                            (96, 5, 8, "dense"), (100, 13, 0, "dense_iso_target"), (18, 130, 0, "dense_iso_target"),
                                            (200, 9, 0, "iso"), (256, 5, 0, "diag"), (130, 7, 0, "dense"), (160, 6, 0, "dense_iso_target"),
                                            (16, 100, 0, "iso"), (17, 35, 4, "dense"), (31, 64, 0, "diag"), (64, 16, 4, "dense"),
                                            (100, 65, 4, "diag"), (176, 20, 0, "iso"), (100, 300, 8, "iso"), (50, 33, 2, "dense"),
                                            (300, 20, 0, "iso"), (200, 70, 4, "dense"), (250, 17, 0, "diag"), (384, 9, 0, "dense_iso_target"),
                                            (448, 9, 0, "diag"), (300, 9, 0, "dense"), (272, 12, 4, "dense_iso_target")], 12))
def test_dense_gaussian_target_cooperative_kernel(mhx, oracle, d, C, lanes, prop, real):
    """RWMH on the dense Gaussian target with L lanes per chain (mhx_rwmh_dense_kernels.h): the default above 64
    dimensions, on request below; ISO and DIAG proposals, random initial states, a schedule with discard and
    thinning, a second call that resumes the run."""
    import cases
    seed = 77 + d
    Sig = cases.sigma_ar1(d, 0.6)
    iso_target = prop.endswith("iso_target")
    model = mhx.DensityModel(mhx.IsoGaussian(d) if iso_target else mhx.CorrGaussian(Sig))
    if prop == "iso":
        s = float(np.float32(1.7 / d ** 0.5))
        spl, op = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), oracle.Proposal(oracle.PROP_ISO, s)
    elif prop.startswith("dense"):
        Sp = (1.7 ** 2 / d) * cases.sigma_ar1(d, 0.4)
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), Sp))
        op = oracle.Proposal(oracle.PROP_DENSE, vec=oracle.pack_lower(np.linalg.cholesky(Sp)))
    else:
        sv = (np.float32(1.7 / d ** 0.5) * (0.5 + np.random.default_rng(d).random(d))).astype(np.float32)
        spl, op = mhx.RWMH([mhx.Normal(0.0, float(v)) for v in sv]), oracle.Proposal(oracle.PROP_DIAG, vec=sv)
    run = mhx.Run(model, spl, nchains=C, seed=seed, first_chain=11, reduce_lanes=lanes)
    run.init(None)
    run.sample(6, 3, 2, 0)
    p1, a1 = run.samples()
    st = run.stats()
    L = st["reduce_lanes"]
    want_L = lanes if lanes else next(v for v in (2, 4, 8, 16, 32, 64) if (4 if real == "f64" else 2) * d <= 25 * v or v == 64)   # <= 12.5 (fp64: 6.25) rows per lane
    nimg = (0 if iso_target else 1) + (1 if prop.startswith("dense") else 0)
    if cases.mfma_fits(d, lanes, nimg, real):
        # one factor for all chains: the matrix-core kernel (mhx_rwmh_mfma_kernels.h), reduction shape 4
        assert st["kernel_variant"] == 8 and L == 4
    elif real == "f64" and st["kernel_variant"] == 0:
        # fp64 factor images are twice the size: past ~128 dimensions they no longer fit the 160 KB of LDS of a block and
        # the run falls back to the state-in-HBM kernel (same chain, sequential reduction shape)
        assert d >= 128 and L == 1 and lanes == 0
    else:
        assert st["kernel_variant"] == 5 and L == want_L
    run.sample(5, 1, 1, 0)
    p2, a2 = run.samples()
    x, lp, cnt = run.state()
    # the same chain in one go: 3 discarded, 6 samples 2 apart (13 transitions), then 5 more 1 apart
    ot = oracle.iso_gauss(d, reduce_lanes=L) if iso_target else oracle.corr_gauss_from_cov(Sig, reduce_lanes=L)
    ref = oracle.rwmh(ot, op, oracle.schedule(19, 0, 1), seed, 11, C)
    idx1 = [3 + 2 * i for i in range(6)]
    idx2 = [idx1[-1] + 1 + i for i in range(5)]
    _same(p1, ref["samples"][idx1], "first call")
    _same(p2, ref["samples"][idx2], "resumed call")
    _same(a2, ref["accepted"][idx2], "accept flags")
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
    assert (0.02 if d <= 128 else 0.0) <= ref["accepted"][1:].mean() < 0.9
    run.close()


@pytest.mark.parametrize("lanes", [1, 2])
def test_negative_zero_initial_coordinates(mhx, oracle, real, lanes):
    """the fp64 plain-walk kernel re-forms a rejected state as fma(0, n, x), which is x for every x but -0.0.  A
    caller-supplied -0.0 therefore enters the chain as +0.0 on both sides (mhx_run_init / mhx_run_set_state; orc_rwmh): equal
    under ==, so sample 1 == initial_params (test/runtests.jl:203-213) still holds, and the chain is the oracle's bit for bit."""
    d, C, N = 100, 64, 24
    s = float(np.float32(2.38 / d ** 0.5))
    init = np.random.default_rng(3).normal(size=(d, C))
    init[::3] = -0.0
    init[1::7] = 0.0
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
    chain = mhx.sample(model, spl, N, C, seed=8, initial_params=init, reduce_lanes=lanes)
    ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=chain.stats["reduce_lanes"]), oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(N),
                      8, 0, C, init=init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert (chain.value[0, :d] == init.astype(cases.R())).all()             # == holds, -0.0 == 0.0
    assert not np.signbit(chain.value[:, :d][chain.value[:, :d] == 0]).any()  # no -0.0 anywhere in the chain
    # the same through setparams!!
    run = mhx.Run(model, spl, nchains=C, seed=8, reduce_lanes=lanes)
    run.init(np.ones((d, C)))
    run.set_params(init)
    assert not np.signbit(run.state()[0][init == 0]).any()


@pytest.mark.parametrize("d,C,prop", soak_tail([(65, 130, "iso"), (100, 70, "diag"), (128, 64, "iso"), (160, 33, "diag"), (200, 65, "iso"), (320, 10, "diag"),
                                      (64, 66, "iso"), (96, 5, "dense"), (40, 9, "dense")], 4))
def test_user_log_density_register_kernel_with_the_state_tail_in_lds(mhx, oracle, real, d, C, prop):
    """A user log-density needs the whole candidate in one lane's registers; above 64 (fp32: 128) dimensions the state keeps only
    its head there and its tail in LDS -- the register kernel then reaches d = 160 in fp64 (320 in fp32) instead of handing an
    fp64 d = 100 model to the state-in-HBM kernel.  Same chain as the oracle: ISO / DIAG proposals, chains that do not fill a wave,
    thinning with a discarded prefix, the state after the call; a dense proposal keeps its own limits."""
    limit = 160 if real == "f64" else 320
    rng = np.random.default_rng(d)
    data = np.concatenate([rng.normal(size=d), 0.5 + rng.random(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    ut = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    if prop == "iso":
        s = float(np.float32(2.38 / d ** 0.5))
        spl, op = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), oracle.Proposal(oracle.PROP_ISO, s)
    elif prop == "diag":
        sv = (np.float32(2.38 / d ** 0.5) * (0.5 + rng.random(d))).astype(np.float32)
        spl, op = mhx.RWMH([mhx.Normal(0.0, float(v)) for v in sv]), oracle.Proposal(oracle.PROP_DIAG, vec=sv)
    else:
        Sp = (2.38 ** 2 / d) * cases.sigma_ar1(d, 0.3)
        spl, op = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), Sp)), oracle.Proposal(oracle.PROP_DENSE, vec=oracle.pack_lower(np.linalg.cholesky(Sp)))
    init = rng.normal(size=(d, C))
    r = mhx.Run(model, spl, nchains=C, seed=77, first_chain=3)
    r.init(init)
    r.sample(6, 3, 2, 0)
    got, got_acc = r.samples()
    var = r.stats()["kernel_variant"]
    in_registers = d <= ((48 if real == "f64" else 96) if prop == "dense" else limit)
    assert var == (2 if in_registers else 0), var
    ref = oracle.rwmh(ut, op, oracle.schedule(6, 3, 2), 77, 3, C, init=init)
    _same(got, ref["samples"], "samples")
    _same(got_acc, ref["accepted"], "accepted")
    x, lp, cnt = r.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
