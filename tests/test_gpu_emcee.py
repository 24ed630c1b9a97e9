"""GPU parity: HIP stretch-move kernels vs the oracle's split-mode ensemble (bit exact), and the
reference's known answers (test/emcee.jl).  Reference behaviour: src/emcee.jl:1-102."""
import os

import numpy as np
import pytest

import cases
from conftest import soak_tail
import user_targets

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    assert a.dtype == b.dtype, "%s: dtypes %s / %s" % (what, a.dtype, b.dtype)
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s" % (what, len(bad), bad[0])


@pytest.mark.parametrize("flags_name,lanes", [("auto", 1), ("generic", 0), ("auto", 0), ("auto", 4), ("auto", 16)])
@pytest.mark.parametrize("d,W,N", [(3, 10, 16), (50, 130, 12), (5, 257, 20), (17, 64, 9)])
def test_emcee_corr_gauss_bit_exact(mhx, oracle, d, W, N, flags_name, lanes, real):
    flags = mhx.FLAG_GENERIC if flags_name == "generic" else 0
    if lanes > d:
        pytest.skip("more lanes than dimensions")
    Sig = cases.sigma_ar1(d, 0.9)
    init = cases.emcee_init(d, W, 5)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    chain = mhx.sample(model, spl, N, seed=21, first_chain=3, initial_params=init, discard_initial=2, thinning=3,
                       flags=flags, reduce_lanes=lanes)
    Lx = chain.stats["reduce_lanes"]
    if lanes >= 1:
        assert Lx == lanes
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=Lx), 2.0, 1, oracle.schedule(N, 2, 3), 21, 3, W, init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
    if flags_name == "generic":
        assert chain.stats["kernel_variant"] == 0 and Lx == 1
    else:
        # the register kernel holds a walker and its candidate in VGPRs: up to 64 dimensions in fp32, 32 in fp64
        assert chain.stats["kernel_variant"] in ((4,) if Lx > 1 else ((2, 6) if d <= 160 else (0,)))       # 6: a small ensemble as one persistent block


@pytest.mark.parametrize("lanes", [1, 0])
def test_emcee_continued_call_records_the_live_state(mhx, oracle, lanes, real):
    """mhx_run_sample continues the chains: sample(5, 0) then sample(5, 0) on the register / cooperative kernels, whose
    live state is the walker-major copy -- slot 0 of the second call is the ensemble as the first call left it."""
    d, W = 20, 48
    Sig = cases.sigma_ar1(d, 0.8)
    init = cases.emcee_init(d, W, 9)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=5, first_chain=1, reduce_lanes=lanes)
    run.init(init)
    run.sample(5, 0)
    first, _ = run.samples()
    run.sample(5, 0)
    second, acc2 = run.samples()
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=run.stats()["reduce_lanes"]), 2.0, 1, oracle.schedule(9), 5, 1, W, init)
    _same(first, ref["samples"][:5], "first call")
    _same(second, ref["samples"][4:], "second call (slot 0 = the state the first call ended in)")
    _same(acc2[1:], ref["accepted"][5:], "accept flags")
    run.close()


@pytest.mark.parametrize("prior", ["iso", "diag_mean", "dense", "normals"])
@pytest.mark.parametrize("mode", ["split", "sequential"])
def test_emcee_initial_walkers_are_drawn_on_the_device(mhx, oracle, prior, mode, real):
    """src/emcee.jl:29-34: without initial_params the walkers are W draws from the distribution StretchProposal wraps.  A
    (Mv)Normal / vector-of-Normals prior is drawn by mhx_run_init(run, NULL) from Philox stream INIT -- the oracle draws the
    same walkers -- and both sweeps continue from them bit for bit."""
    d, W, N = 6, 50, 7
    rng = np.random.default_rng(3)
    Sig = cases.sigma_ar1(d, 0.8)
    if prior == "iso":
        s = float(np.float32(1.5))
        dist, op = mhx.MvNormal(mhx.zeros(d), s * s * mhx.I), oracle.Proposal(oracle.PROP_ISO, s)
    elif prior == "diag_mean":
        sv = (0.5 + rng.random(d)).astype(np.float32).astype(np.float64)
        mu = rng.normal(size=d).astype(np.float32).astype(np.float64)
        dist, op = mhx.MvNormal(mu, sv ** 2), oracle.Proposal(oracle.PROP_DIAG, vec=sv, mean=mu)
    elif prior == "dense":
        Sp = 0.7 * cases.sigma_ar1(d, 0.3)
        dist, op = mhx.MvNormal(mhx.zeros(d), Sp), oracle.Proposal(oracle.PROP_DENSE, vec=oracle.pack_lower(np.linalg.cholesky(Sp)))
    else:
        sv = np.array([2.0, 0.5, 1.0, 1.5, 0.25, 3.0])
        dist, op = [mhx.Normal(0.0, float(v)) for v in sv], oracle.Proposal(oracle.PROP_DIAG, vec=sv)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(dist))
    seq = mode == "sequential"
    chain = mhx.sample(model, spl, N, seed=9, first_chain=4, reduce_lanes=1, flags=mhx.FLAG_EMCEE_SEQUENTIAL if seq else 0)
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig), 2.0, 0 if seq else 1, oracle.schedule(N), 9, 4, W, None, prior=op)
    _same(chain.value, ref["samples"], "samples (slot 0 = the drawn walkers)")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert np.abs(chain.value[0, :d, :]).max() > 0.1


@pytest.mark.parametrize("d,W,N,user", [(3, 10, 16, False), (2, 37, 9, True), (50, 24, 5, False), (7, 130, 6, True)])
def test_emcee_reference_sequential_sweep_bit_exact(mhx, oracle, d, W, N, user, real):
    """MHX_FLAG_EMCEE_SEQUENTIAL: the reference's own sweep (src/emcee.jl:39-58 -- walkers one after another, partner
    mod1(i + r, W), the already-updated position when idx < i) on the device, bit for bit against the oracle's mode 0,
    catalogue and user log-densities, recorded with discard and thinning, continued calls."""
    Sig = cases.sigma_ar1(d, 0.9)
    init = cases.emcee_init(d, W, 5)
    if user:
        rng = np.random.default_rng(d)
        data = np.concatenate([rng.normal(size=d), 0.5 + rng.random(d)]).astype(np.float32)
        model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
        ot = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    else:
        model, ot = mhx.DensityModel(mhx.CorrGaussian(Sig)), oracle.corr_gauss_from_cov(Sig)
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    chain = mhx.sample(model, spl, N, seed=21, first_chain=3, initial_params=init, discard_initial=2, thinning=3,
                       flags=mhx.FLAG_EMCEE_SEQUENTIAL)
    assert chain.stats["kernel_variant"] == 7 and chain.stats["launches"] == 1
    ref = oracle.emcee(ot, 2.0, 0, oracle.schedule(N, 2, 3), 21, 3, W, init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
    # a different Markov kernel from the parallel half-split: the traces differ
    split = mhx.sample(model, spl, N, seed=21, first_chain=3, initial_params=init, discard_initial=2, thinning=3)
    assert not np.array_equal(split.value, chain.value)


def test_emcee_reference_sweep_known_answer(mhx, real):
    """test/emcee.jl:16-42 at the reference's own sizes (1000 walkers, 1000 sweeps) with the reference's own sweep on the
    device: E[s] = 49/24, E[m] = 7/6 (atol 0.1), and `range == 26:4:...` for discard_initial = 25, thinning = 4."""
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.NIG_UNTRANSFORMED, 2))
    spl = mhx.Ensemble(1000, mhx.StretchProposal([mhx.InverseGamma(2, 3), mhx.Normal(0, 1)]))
    chain = mhx.sample(model, spl, 1000, seed=100, param_names=["s", "m"], flags=mhx.FLAG_EMCEE_SEQUENTIAL)
    assert abs(chain.mean("s") - 49 / 24) < 0.1 and abs(chain.mean("m") - 7 / 6) < 0.1
    chain = mhx.sample(model, spl, 50, seed=100, param_names=["s", "m"], discard_initial=25, thinning=4, flags=mhx.FLAG_EMCEE_SEQUENTIAL)
    assert chain.range() == range(26, 26 + 4 * 50, 4)


def test_emcee_golden_trace(mhx, real):
    tr = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traces64.npz" if real == "f64" else "traces.npz"))
    d, W = 3, 10
    model = mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.9)))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    chain = mhx.sample(model, spl, 16, seed=21, first_chain=0, initial_params=cases.emcee_init(d, W, 5), reduce_lanes=1)
    _same(chain.value, tr["emcee_split/samples"], "samples")
    _same(chain.accepted, tr["emcee_split/accepted"], "accepted")


def test_emcee_nig_known_answer_user_source(mhx, oracle, real):
    """test/emcee.jl:3-42 through a hiprtc-compiled user log-density; E[s]=49/24, E[m]=7/6, atol 0.1."""
    W = 1000
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.NIG_UNTRANSFORMED, 2))
    spl = mhx.Ensemble(W, mhx.StretchProposal([mhx.InverseGamma(2, 3), mhx.Normal(0, 1)]))
    chain = mhx.sample(model, spl, 1000, seed=100, param_names=["s", "m"])
    assert chain.range() == range(1, 1001)
    assert abs(chain.mean("s") - 49 / 24) < 0.1
    assert abs(chain.mean("m") - 7 / 6) < 0.1
    chain2 = mhx.sample(model, spl, 1000, seed=100, param_names=["s", "m"], discard_initial=25, thinning=4)
    assert chain2.range() == range(26, 26 + 4 * 1000, 4)                   # test/emcee.jl:39
    assert abs(chain2.mean("s") - 49 / 24) < 0.1
    assert abs(chain2.mean("m") - 7 / 6) < 0.1
    # bit-exact vs the oracle running the same source compiled for the host
    init = chain.value[0, :2, :]
    ref = oracle.emcee(user_targets.host_target(oracle, user_targets.NIG_UNTRANSFORMED, 2), 2.0, 1,
                       oracle.schedule(1000), 100, 0, W, init)
    _same(chain.value, ref["samples"], "samples")


def test_emcee_transformed_space(mhx, real):
    """test/emcee.jl:44-83."""
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.NIG_TRANSFORMED, 2))
    spl = mhx.Ensemble(1000, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(2), mhx.I)))
    chain = mhx.sample(model, spl, 1000, seed=101, param_names=["logs", "m"])
    assert abs(np.exp(chain["logs"].astype(np.float64)).mean() - 49 / 24) < 0.1
    assert abs(chain.mean("m") - 7 / 6) < 0.1


@pytest.mark.parametrize("d,W,N,lanes", [(128, 40, 4, 16), (128, 37, 3, 2), (100, 33, 4, 8), (126, 18, 3, 16), (33, 9, 5, 32),
                                         (200, 21, 3, 16), (256, 10, 3, 32), (190, 12, 3, 64)])
def test_emcee_cooperative_kernel_large_dimensions(mhx, oracle, d, W, N, lanes, real):
    """The cooperative kernel at its largest factor images (several float4 of the factor per thread, several
    float4 of a walker per lane), dimensions that are not multiples of 4, odd ensemble sizes, recorded sweeps."""
    Sig = cases.sigma_ar1(d, 0.8)
    init = cases.emcee_init(d, W, 9)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    try:
        chain = mhx.sample(model, spl, N, seed=4, first_chain=1, initial_params=init, reduce_lanes=lanes)
    except mhx.ArgumentError as e:
        # the factor image and the candidate rows live in the 160 KB of LDS of a block: fp64 images are twice the size
        assert real == "f64" and "exceeds the LDS" in str(e)
        pytest.skip("the fp64 factor image of d = %d does not fit the LDS" % d)
    assert chain.stats["reduce_lanes"] == lanes and chain.stats["kernel_variant"] == 4
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=lanes), 2.0, 1, oracle.schedule(N), 4, 1, W, init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")


@pytest.mark.parametrize("d,W", [(7, 37), (50, 64), (64, 10), (100, 66), (160, 5), (161, 4)])
def test_emcee_user_log_density_walker_major_rows(mhx, oracle, d, W, real):
    """The register kernel (user log-density in HIP source) on walker-major rows: dimensions with and without
    padding, recorded sweeps with thinning, state read back in the ABI layout."""
    rng = np.random.default_rng(d)
    data = np.concatenate([rng.normal(size=d), 0.5 + rng.random(d)]).astype(np.float32)
    model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
    ut = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    init = cases.emcee_init(d, W, 2)
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    chain = mhx.sample(model, spl, 9, seed=6, first_chain=2, initial_params=init, discard_initial=1, thinning=2)
    assert chain.stats["kernel_variant"] in ((2, 6) if d <= 160 else (0,))     # (6: a small ensemble as one persistent block)
    ref = oracle.emcee(ut, 2.0, 1, oracle.schedule(9, 1, 2), 6, 2, W, init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")


@pytest.mark.parametrize("d,W,kind", [(7, 37, "user"), (50, 200, "user"), (64, 10, "user"), (3, 2, "user"), (2, 131, "iid"), (30, 66, "banana")])
def test_lane_per_walker_kernel_one_launch_per_sweep(mhx, oracle, d, W, kind, real, engine):
    """The register kernel (any target) with both halves in one launch: the second half's lanes evaluate the log-density twice --
    their partner's candidate, then their own.  Same tensor as two half-step launches and as the oracle (user source, catalogue
    targets; odd W, W = 2, blocks that hold one half's tail, a continued call)."""
    if kind == "user":
        data = np.concatenate([np.linspace(-1.0, 1.0, d), np.linspace(0.5, 2.0, d)]).astype(np.float32)
        model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
        tgt = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    elif kind == "iid":
        data = np.random.default_rng(5).normal(size=30)
        model = mhx.DensityModel(mhx.IIDNormal(data))
        tgt = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data)
    else:
        model = mhx.DensityModel(mhx.Banana(d, 0.03))
        tgt = oracle.Target(oracle.TARGET_BANANA, d, params=[0.03])
    init = cases.emcee_init(d, W, 2)
    if kind == "iid":
        init[1] = np.abs(init[1]) + 0.5
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))

    engine.setenv("MHX_EMCEE_PERSIST", "0")             # (these sizes would run as one persistent block)

    def go(fused):
        engine.setenv("MHX_EMCEE_FUSED", "1" if fused else "0")
        r = mhx.Run(model, spl, seed=21, reduce_lanes=1)
        r.init(init)
        r.sample(5, 2, 2, 0)
        a = r.samples() + (r.stats()["launches"],)
        r.sample(3, 0, 1, 0)
        return a, r.samples() + (r.state()[2], r.stats()["kernel_variant"])

    f, u = go(True), go(False)
    assert f[1][3] == 2 and u[0][2] == 2 * f[0][2], (f[1][3], f[0][2], u[0][2])
    for k in range(2):
        _same(f[k][0], u[k][0], "samples, call %d" % k)
        _same(f[k][1], u[k][1], "accepted, call %d" % k)
    _same(f[1][2], u[1][2], "acceptance counters")
    ref = oracle.emcee(tgt, 2.0, 1, oracle.schedule(5, 2, 2), 21, 0, W, init)
    _same(f[0][0], ref["samples"], "one launch per sweep vs oracle")
    _same(f[0][1], ref["accepted"], "accepted vs oracle")


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("d,W,user", [(50, 100, False), (7, 33, True), (20, 64, False)])
def test_sharded_ensemble_slices_reproduce_the_single_gpu_run(mhx, oracle, d, W, user, world, real):
    """An ensemble sharded over `world` ranks (mhx.dist.ShardedEnsemble): every rank moves one slice of the moving
    half per half-step.  Emulated on one device -- the slices run one after the other on the same state, which is what
    the all-gather of the multi-GPU run reconstructs on every rank -- the result must be the oracle's sweep."""
    from mhx.dist import ShardedEnsemble
    if user:
        rng = np.random.default_rng(d)
        data = np.concatenate([rng.normal(size=d), 0.5 + rng.random(d)]).astype(np.float32)
        model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
        ot = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    else:
        Sig = cases.sigma_ar1(d, 0.8)
        model = mhx.DensityModel(mhx.CorrGaussian(Sig))
        ot = None
    init = cases.emcee_init(d, W, 4)
    run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=8, first_chain=5)
    run.init(init)
    sh = ShardedEnsemble(run, world=world, exchange=None)
    sh.sweep(6)
    x, lp, cnt = run.state()
    if not user:                                             # the lanes-per-walker choice is the engine's
        ot = oracle.corr_gauss_from_cov(Sig, reduce_lanes=_lanes_of(mhx, model, W))
    ref = oracle.emcee(ot, 2.0, 1, oracle.schedule(7), 8, 5, W, init)
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
    run.close()


def _lanes_of(mhx, model, W):
    """lanes per walker the engine picks for this model / ensemble size (reported by a throw-away run)"""
    d = model.dim
    probe = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=1)
    probe.init(cases.emcee_init(d, W, 1))
    probe.sample(2, 0, 1, 0)
    L = probe.stats()["reduce_lanes"]
    probe.close()
    return L


@pytest.mark.soak_f32
def test_sharded_ensemble_over_rccl_single_rank(mhx, oracle, real):
    """The collective path of ShardedEnsemble through the C ABI (mhx_comm_*: RCCL) with the one rank this box has: the
    moved slice is packed, all-gathered and unpacked on the run's stream after every half-step."""
    from mhx.dist import Comm, ShardedEnsemble
    d, W = 12, 40
    Sig = cases.sigma_ar1(d, 0.7)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    init = cases.emcee_init(d, W, 6)
    run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3, first_chain=1)
    run.init(init)
    comm = Comm(run.ctx, 0, 1, Comm.unique_id())
    try:
        sh = ShardedEnsemble(run, exchange="rccl", comm=comm)
        sh.sweep(4)
        v = comm.allreduce_sum(np.array([1.5, -2.0, 7.0]))          # the statistics all-reduce, one rank: identity
        assert v.tolist() == [1.5, -2.0, 7.0]
    finally:
        comm.close()
    x, lp, cnt = run.state()
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=_lanes_of(mhx, model, W)), 2.0, 1, oracle.schedule(5), 3, 1, W, init)
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    run.close()


def _banded_sigma(d, bw, seed):
    """Sigma = inv(A^T A) for a random lower factor A of bandwidth bw (a Gaussian Markov model of order bw)"""
    rng = np.random.default_rng(seed)
    A = np.zeros((d, d))
    for r in range(d):
        A[r, r] = 1.0 + rng.uniform(0.0, 1.0)
        for c in range(max(0, r - bw), r):
            A[r, c] = rng.normal() * 0.4
    return np.linalg.inv(A.T @ A)


@pytest.mark.parametrize("d,W,bw", [(50, 256, 1), (12, 130, 2), (33, 192, 5), (64, 128, 8), (20, 96, 0)])
def test_banded_precision_factor_is_detected_and_bit_identical(mhx, oracle, real, d, W, bw):
    """A Gaussian Markov target (banded inv(chol Sigma)): the cooperative stretch move skips the structural zeros -- the same
    chain, bit for bit, as the dense form (MHX_FLAG_DENSE_FACTOR) and as the oracle's full row products."""
    Sig = cases.sigma_ar1(d, 0.9) if bw == 1 else (_banded_sigma(d, bw, d) if bw else np.diag(np.linspace(0.5, 2.0, d)))
    N = 7
    init = cases.emcee_init(d, W, 5)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    band = mhx.sample(model, spl, N, seed=9, initial_params=init)
    dense = mhx.sample(model, spl, N, seed=9, initial_params=init, flags=mhx.FLAG_DENSE_FACTOR, reduce_lanes=band.stats["reduce_lanes"])
    assert band.stats["factor_band"] == bw and dense.stats["factor_band"] == -1 and band.stats["kernel_variant"] == 4
    _same(band.value, dense.value, "band form vs dense form")
    _same(band.accepted, dense.accepted, "accepted")
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=band.stats["reduce_lanes"]), 2.0, 1, oracle.schedule(N), 9, 0, W, init)
    _same(band.value, ref["samples"], "band form vs oracle")
    _same(band.accepted, ref["accepted"], "accepted vs oracle")


@pytest.mark.parametrize("d,W,bw,lanes", [(50, 256, 1, 0), (50, 131, 1, 16), (12, 130, 2, 4), (33, 193, 5, 8), (20, 96, 0, 0), (7, 3, 1, 2),
                                          (64, 2, 8, 16)])
def test_one_launch_per_sweep_is_the_same_chain_as_two_half_steps(mhx, oracle, real, d, W, bw, lanes, engine):
    """The banded lane-group form moves BOTH halves in one launch (the second half's groups re-do their partner's move from the
    old state; walker rows double-buffered, only rows the other buffer does not hold are stored): the same tensor, accept flags
    and counters as two half-step launches (MHX_EMCEE_FUSED=0) and as the oracle's split sweep -- odd W, ensembles smaller than a
    block, thinning + a discarded prefix, a continued call, a state handed in from outside in between."""
    Sig = cases.sigma_ar1(d, 0.9) if bw == 1 else (_banded_sigma(d, bw, d) if bw else np.diag(np.linspace(0.5, 2.0, d)))
    init = cases.emcee_init(d, W, 5)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))

    def go(fused):
        engine.setenv("MHX_EMCEE_FUSED", "1" if fused else "0")
        r = mhx.Run(model, spl, seed=9, reduce_lanes=lanes)
        r.init(init)
        l0 = r.stats()["launches"]
        r.sample(6, 3, 2, 0)                                   # 3 discarded, every 2nd of the rest
        a = r.samples() + (r.stats()["launches"] - l0,)
        r.sample(4, 0, 1, 0)                                   # continued: the first launch finds the other buffer stale
        b = r.samples()
        x, lp, cnt = r.state()
        r.set_params(x[:, ::-1].copy())                        # walkers handed in from outside (reversed order)
        r.sample(3, 0, 1, 0)
        c = r.samples() + (r.state()[2], r.stats()["reduce_lanes"])
        return a, b, c

    f, u = go(True), go(False)
    assert f[0][2] in (13, 14) and u[0][2] == 2 * f[0][2], (f[0][2], u[0][2])           # one launch per sweep against two
    for k, (pf, pu) in enumerate(zip(f, u)):
        _same(pf[0], pu[0], "samples, call %d" % k)
        _same(pf[1], pu[1], "accepted, call %d" % k)
    _same(f[2][2], u[2][2], "acceptance counters")
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=f[2][3]), 2.0, 1, oracle.schedule(6, 3, 2), 9, 0, W, init)
    _same(f[0][0], ref["samples"], "one launch per sweep vs oracle")
    _same(f[0][1], ref["accepted"], "accepted vs oracle")


@pytest.mark.parametrize("d,W,knobs", soak_tail([(50, 200, {}), (17, 71, {}), (64, 129, {}), (33, 64, {"MHX_EMCEE_SCALAR": "4"}),
                                       (16, 66, {"MHX_EMCEE_SCALAR": "16"}), (24, 3, {}), (50, 2, {}), (12, 33, {})], 4))
def test_scalar_factor_form_one_launch_per_sweep(mhx, oracle, real, d, W, knobs, engine):
    """The scalar-factor form (dense factor) with both halves in one launch: mixed blocks of 16 walkers of each half, the second
    half's carry three candidate rows each through phase 2 (64 rows: every lane busy).  Same tensor as two half-step launches and
    as the oracle; 4 / 8 / 16 row classes, odd W, ensembles smaller than a block, a continued call."""
    Sig = _rotated(d)
    init = cases.emcee_init(d, W, 5)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    engine.setenv("MHX_EMCEE_MFMA", "0")                 # (the matrix-core form would take these shapes by default)
    for k, v in knobs.items():
        engine.setenv(k, v)

    def go(fused):
        engine.setenv("MHX_EMCEE_FUSED", "1" if fused else "0")
        r = mhx.Run(model, spl, seed=11)
        r.init(init)
        l0 = r.stats()["launches"]
        r.sample(5, 2, 2, 0)
        a = r.samples() + (r.stats()["launches"] - l0,)
        r.sample(3, 0, 1, 0)
        return a, r.samples() + (r.state()[2], r.stats())

    f, u = go(True), go(False)
    assert f[1][3]["kernel_variant"] == 9 and u[0][2] == 2 * f[0][2], (f[1][3], f[0][2], u[0][2])
    for k in range(2):
        _same(f[k][0], u[k][0], "samples, call %d" % k)
        _same(f[k][1], u[k][1], "accepted, call %d" % k)
    _same(f[1][2], u[1][2], "acceptance counters")
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=f[1][3]["reduce_lanes"]), 2.0, 1, oracle.schedule(5, 2, 2), 11, 0, W, init)
    _same(f[0][0], ref["samples"], "one launch per sweep vs oracle")
    _same(f[0][1], ref["accepted"], "accepted vs oracle")


def test_a_dense_factor_keeps_the_dense_form(mhx):
    d, W = 24, 128
    rng = np.random.default_rng(3)
    Q, _ = np.linalg.qr(rng.normal(size=(d, d)))
    Sig = Q @ cases.sigma_ar1(d, 0.9) @ Q.T                       # the rotated variant: no structural zeros
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))),
                       4, seed=1)
    assert chain.stats["factor_band"] == -1


@pytest.mark.parametrize("W", [128, 131])
def test_sweep_kernel_deferred_record_writes_the_same_tensor(mhx, real, W, engine):
    """MHX_EMCEE_SWEEP_DEFER=1 (tuning knob): the one-launch-per-sweep kernel records a sweep at the top of the NEXT launch, the
    call's last one by a small kernel of its own.  Same tensor, through thinning, a discarded prefix and slab-wise calls."""
    d = 10
    Sig = cases.sigma_ar1(d, 0.8)
    init = cases.emcee_init(d, W, 3)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))

    def run_it(slab):
        r = mhx.Run(model, spl, seed=4)
        r.init(init)
        if slab:
            return r.sample_to_host(9, 2, 3, 0, slab_samples=slab)
        r.sample(9, 2, 3, 0)
        return r.samples()

    want = run_it(0)
    engine.setenv("MHX_EMCEE_SWEEP_DEFER", "1")
    for slab in (0, 2, 4):
        got = run_it(slab)
        _same(got[0], want[0], "samples, slab %d" % slab)
        _same(got[1], want[1], "accepted, slab %d" % slab)


@pytest.mark.parametrize("W", [128, 131])
def test_deferred_record_writes_the_same_tensor(mhx, real, W, engine):
    """MHX_EMCEE_DEFER=1 (tuning knob): the record of a half leaves at the start of the NEXT launch (the half is at rest and final
    for its sweep) instead of at the end of the launch that moved it; odd W: the half at rest is one walker larger.  Same tensor,
    through thinning, a discarded prefix and slab-wise calls."""
    d = 10
    Sig = cases.sigma_ar1(d, 0.8)
    init = cases.emcee_init(d, W, 3)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))

    def run_it(slab):
        r = mhx.Run(model, spl, seed=4)
        r.init(init)
        if slab:
            return r.sample_to_host(9, 2, 3, 0, slab_samples=slab)
        r.sample(9, 2, 3, 0)
        return r.samples()

    want = run_it(0)
    engine.setenv("MHX_EMCEE_DEFER", "1")
    for slab in (0, 4, -2):
        got = run_it(slab)
        _same(got[0], want[0], "samples, deferred record, slab %d" % slab)
        _same(got[1], want[1], "accepted")


def _rotated(d, rho=0.9, seed=50):
    Q, _ = np.linalg.qr(np.random.default_rng(seed).normal(size=(d, d)))
    return Q @ cases.sigma_ar1(d, rho) @ Q.T


@pytest.mark.parametrize("d,W", [(50, 200), (17, 70), (64, 129), (33, 64), (8, 66)])
def test_scalar_factor_form_runs_a_dense_factor_bit_exact(mhx, oracle, real, d, W, engine):
    """Round 4: a DENSE precision factor (the rotated C3 target: no band to exploit) runs the scalar-factor form of the cooperative
    stretch move -- variant 9: a lane owns a walker during A y, the wave-uniform factor entry is the DPP-broadcast operand of
    v_fmac, rows split over the 8 waves of a block = the spec's reduction shape 8 -- and is the oracle's chain bit for bit, through a
    discarded prefix, thinning, the initial draw on the device and a slab-streamed call (src/emcee.jl:70-102).  (MHX_EMCEE_MFMA=0:
    the matrix-core form is the default for these shapes since the end of round 4.)"""
    engine.setenv("MHX_EMCEE_MFMA", "0")
    Sig = _rotated(d)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    chain = mhx.sample(model, spl, 7, seed=13, discard_initial=2, thinning=2)
    assert chain.stats["kernel_variant"] == 9 and chain.stats["reduce_lanes"] == 8 and chain.stats["factor_band"] == -1
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=8), 2.0, 1, oracle.schedule(7, 2, 2), 13, 0, W, None,
                       prior=oracle.Proposal(oracle.PROP_ISO, 1.0))
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert chain.stats["accepted"] == int(ref["accept_counts"].sum())
    r = mhx.Run(model, spl, seed=13)
    r.init(None)
    got, got_acc = r.sample_to_host(7, 2, 2, 0, slab_samples=-3)
    _same(got, ref["samples"], "slab-streamed")
    _same(got_acc, ref["accepted"], "accepted, slab-streamed")


@pytest.mark.parametrize("knobs", soak_tail([{"MHX_EMCEE_SCALAR": "4"}, {"MHX_EMCEE_SCAL_MODE": "0"}, {"MHX_EMCEE_SCALAR": "0"}, {"MHX_EMCEE_SCALAR": "16"},
                                             {"MHX_EMCEE_SCAL_WPB": "64"}, {"MHX_EMCEE_SCAL_WPB": "16"}, {"MHX_EMCEE_SCAL_REC": "0"}], 3))
def test_scalar_factor_form_every_shape_and_the_lane_group_form_agree_with_the_oracle(mhx, oracle, real, knobs, engine):
    """The tuning knobs of the scalar-factor form (waves per block = reduction shape 4 / 16, walkers per block, SGPR operands instead
    of the DPP broadcast, the record straight from the move mapping) and MHX_EMCEE_SCALAR=0 (the lane-group form with its LDS image):
    every one of them is the oracle's chain for the reduction shape it reports."""
    engine.setenv("MHX_EMCEE_MFMA", "0")
    for k, v in knobs.items():
        engine.setenv(k, v)
    d, W = 50, 131
    Sig = _rotated(d)
    init = cases.emcee_init(d, W, 7)
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), 6,
                       seed=5, initial_params=init)
    L = chain.stats["reduce_lanes"]
    assert chain.stats["kernel_variant"] == (4 if knobs.get("MHX_EMCEE_SCALAR") == "0" else 9)
    if "MHX_EMCEE_SCALAR" in knobs and knobs["MHX_EMCEE_SCALAR"] != "0":
        assert L == int(knobs["MHX_EMCEE_SCALAR"])
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=L), 2.0, 1, oracle.schedule(6), 5, 0, W, init)
    _same(chain.value, ref["samples"], "samples %r" % knobs)
    _same(chain.accepted, ref["accepted"], "accepted")


def test_a_large_dense_factor_falls_back_to_the_lane_group_form(mhx, oracle, real):
    """past the register budget of the scalar-factor form (the candidate of a walker in the VGPRs of one lane) the lane-group form
    with its LDS image takes over: same answer, reduction shape as reported"""
    d, W = 120 if real == "f64" else 230, 64
    Sig = _rotated(d, 0.5)
    init = cases.emcee_init(d, W, 2)
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), 3,
                       seed=2, initial_params=init)
    assert chain.stats["kernel_variant"] == 4
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=chain.stats["reduce_lanes"]), 2.0, 1, oracle.schedule(3), 2, 0, W, init)
    _same(chain.value, ref["samples"], "samples")


@pytest.mark.parametrize("d,W,waves", soak_tail([(50, 200, 1), (17, 71, 4), (64, 129, 2), (24, 3, 1), (100, 40, 1), (8, 66, 1), (33, 64, 1), (50, 2, 4),
                                                 (12, 33, 8), (128, 19, 1), (50, 300, 4), (40, 130, 8)], 5))
def test_matrix_core_form_of_the_stretch_move(mhx, oracle, real, d, W, waves, engine):
    """A dense precision factor on the matrix cores (variant 10): 4 lanes per walker, the candidate formed in the MFMA's B-operand
    layout, the factor's operand image built once per run and fetched into registers per launch -- no LDS, no barrier; reduction
    shape 4.  As one launch per sweep and as two half-step launches: the oracle's chain bit for bit (odd W, ensembles smaller than
    a wave, thinning with a discarded prefix, a continued call, the initial draw on the device)."""
    limit = 64 if real == "f64" else 128
    engine.setenv("MHX_EMCEE_MFMA", "1")               # (by default only large fp64 ensembles take this form)
    # waves per block > 1 (round 5): the waves of a block share ONE fetch of the factor's operand image through LDS
    engine.setenv("MHX_EMCEE_MFMA_WAVES", str(waves))
    Sig = _rotated(d, 0.9 if d <= 64 else 0.5)
    init = cases.emcee_init(d, W, 5)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))

    def go(fused):
        engine.setenv("MHX_EMCEE_FUSED", "1" if fused else "0")
        r = mhx.Run(model, spl, seed=11)
        r.init(init)
        r.sample(5, 2, 2, 0)
        a = r.samples() + (r.stats()["launches"],)
        r.sample(3, 0, 1, 0)
        return a, r.samples() + (r.state()[2], r.stats())

    f, u = go(True), go(False)
    st = f[1][3]
    if d > limit:
        assert st["kernel_variant"] != 10
        return
    assert st["kernel_variant"] == 10 and st["reduce_lanes"] == 4 and u[0][2] == 2 * f[0][2], (st, f[0][2], u[0][2])
    for k in range(2):
        _same(f[k][0], u[k][0], "samples, call %d" % k)
        _same(f[k][1], u[k][1], "accepted, call %d" % k)
    _same(f[1][2], u[1][2], "acceptance counters")
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=4), 2.0, 1, oracle.schedule(5, 2, 2), 11, 0, W, init)
    _same(f[0][0], ref["samples"], "matrix-core form vs oracle")
    _same(f[0][1], ref["accepted"], "accepted vs oracle")
    # the initial draw on the device, a slab-streamed call
    chain = mhx.sample(model, spl, 7, seed=13, discard_initial=2, thinning=2)
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=4), 2.0, 1, oracle.schedule(7, 2, 2), 13, 0, W, None,
                       prior=oracle.Proposal(oracle.PROP_ISO, 1.0))
    _same(chain.value, ref["samples"], "device-drawn walkers")
    r = mhx.Run(model, spl, seed=13)
    r.init(None)
    got, got_acc = r.sample_to_host(7, 2, 2, 0, slab_samples=-3)
    _same(got, ref["samples"], "slab-streamed")
    _same(got_acc, ref["accepted"], "accepted, slab-streamed")


@pytest.mark.parametrize("d,W,kind", soak_tail([(2, 1000, "user"), (7, 37, "user"), (24, 1024, "user"), (2, 131, "iid"), (10, 1025, "user"), (10, 64, "user"),
                                                (50, 200, "user"), (3, 2, "user"), (30, 66, "banana")], 5))
def test_small_ensemble_as_one_persistent_block(mhx, oracle, d, W, kind, real, engine):
    """An ensemble of at most 1024 walkers on the lane-per-walker kernel runs a whole sampling call as ONE launch of one persistent
    block (variant 6): thread = walker, rows in LDS for the partners, block barriers between the half-steps.  Same tensor, state and
    counters as the sweep launches (MHX_EMCEE_PERSIST=0) and as the oracle -- thinning with a discarded prefix, a continued call, a
    slab-streamed call, the reference's own ensemble size (1000 walkers, d = 2)."""
    if kind == "user":
        data = np.concatenate([np.linspace(-1.0, 1.0, d), np.linspace(0.5, 2.0, d)]).astype(np.float32)
        model = mhx.DensityModel(mhx.HipLogDensity(user_targets.SHIFTED_GAUSS, d, data=data))
        tgt = user_targets.host_target(oracle, user_targets.SHIFTED_GAUSS, d, data=data)
    elif kind == "iid":
        data = np.random.default_rng(5).normal(size=30)
        model = mhx.DensityModel(mhx.IIDNormal(data))
        tgt = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data)
    else:
        model = mhx.DensityModel(mhx.Banana(d, 0.03))
        tgt = oracle.Target(oracle.TARGET_BANANA, d, params=[0.03])
    init = cases.emcee_init(d, W, 2)
    if kind == "iid":
        init[1] = np.abs(init[1]) + 0.5
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))

    def go(persist):
        engine.setenv("MHX_EMCEE_PERSIST", "1" if persist else "0")
        r = mhx.Run(model, spl, seed=21, reduce_lanes=1)
        r.init(init)
        r.sample(5, 2, 2, 0)
        a = r.samples() + (r.stats(),)
        r.sample(3, 0, 1, 0)
        b = r.samples() + r.state()
        r2 = mhx.Run(model, spl, seed=21, reduce_lanes=1)
        r2.init(init)
        c = r2.sample_to_host(5, 2, 2, 0, slab_samples=2)
        return a, b, c

    p, u = go(True), go(False)
    fits = W <= 1024 and (2 if real == "f64" else 1) * 2 * ((d + 3) & ~3) + 48 <= 512 // ((((W + 63) // 64) + 3) // 4)
    assert p[0][2]["kernel_variant"] == (6 if fits else 2) and u[0][2]["kernel_variant"] == 2, (p[0][2], u[0][2])
    if fits:
        assert p[0][2]["launches"] == 1
    for k in range(2):
        _same(p[k][0], u[k][0], "samples, call %d" % k)
        _same(p[k][1], u[k][1], "accepted, call %d" % k)
    for k, what in ((2, "x"), (3, "lp"), (4, "counters")):
        _same(p[1][k], u[1][k], what)
    _same(p[2][0], u[2][0], "slab-streamed samples")
    ref = oracle.emcee(tgt, 2.0, 1, oracle.schedule(5, 2, 2), 21, 0, W, init)
    _same(p[0][0], ref["samples"], "persistent block vs oracle")
    _same(p[0][1], ref["accepted"], "accepted vs oracle")
