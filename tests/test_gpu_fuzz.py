"""GPU parity on randomly drawn (but reproducible) configurations: sampler x target x proposal x dimension x chain
count x schedule, each compared with the oracle bit for bit.  The hand-picked cases elsewhere pin the known edges;
this sweeps the combinations between them (odd sizes, d = 1, one chain, N = 1, thinning, resumed runs)."""
import numpy as np
import pytest

import cases
from conftest import soak_tail

pytestmark = pytest.mark.gpu

import os
SEED_OFFSET = int(os.environ.get("MHX_FUZZ_SEED", "0"))          # another slice of the configuration space per value


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert a.dtype == b.dtype, "%s: dtypes %s / %s" % (what, a.dtype, b.dtype)
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s" % (what, len(bad), bad[0])


def _schedule(rng):
    N = int(rng.integers(1, 9))
    di = int(rng.integers(0, 4))
    th = int(rng.integers(1, 4))
    return N, di, th


@pytest.mark.parametrize("case", soak_tail(range(60), 20))
def test_rwmh_random_configurations(mhx, oracle, case, real):
    rng = np.random.default_rng(1000 + case + 100000 * SEED_OFFSET)
    d = int(rng.choice([1, 2, 3, 5, 8, 13, 17, 31, 40, 66, 97]))
    C = int(rng.choice([1, 2, 7, 63, 64, 65, 130]))
    N, di, th = _schedule(rng)
    tname = str(rng.choice(["iso", "corr", "banana", "funnel"])) if d >= 2 else "iso"
    pname = str(rng.choice(["iso", "diag", "dense"]))
    static = bool(rng.integers(0, 4) == 0)
    drift = (not static) and bool(rng.integers(0, 5) == 0)
    Sig = cases.sigma_ar1(d, 0.5)
    if tname == "iso":
        tgt, ot = mhx.IsoGaussian(d), (lambda L: oracle.iso_gauss(d, reduce_lanes=L))
    elif tname == "corr":
        tgt, ot = mhx.CorrGaussian(Sig), (lambda L: oracle.corr_gauss_from_cov(Sig, reduce_lanes=L))
    elif tname == "banana":
        tgt, ot = mhx.Banana(d, 0.03), (lambda L: oracle.Target(oracle.TARGET_BANANA, d, params=[0.03], reduce_lanes=L))
    else:
        tgt, ot = mhx.Funnel(d), (lambda L: oracle.Target(oracle.TARGET_FUNNEL, d, reduce_lanes=L))
    mean = (rng.normal(size=d) * 0.2).astype(np.float32).astype(np.float64) if (drift or (static and rng.integers(0, 2))) else None
    mu = mhx.zeros(d) if mean is None else mean
    if pname == "iso":
        s = float(np.float32(0.3 + rng.random()))
        dist, op = mhx.MvNormal(mu, s * s * mhx.I), dict(kind=oracle.PROP_ISO, scale=s)
    elif pname == "diag":
        sv = (0.3 + rng.random(d)).astype(np.float32)
        dist, op = mhx.MvNormal(mu, sv.astype(np.float64) ** 2), dict(kind=oracle.PROP_DIAG, vec=sv)   # a vector of variances
    else:
        A = rng.normal(size=(d, d)) * 0.2
        Sp = A @ A.T + 0.3 * np.eye(d)
        dist, op = mhx.MvNormal(mu, Sp), dict(kind=oracle.PROP_DENSE, vec=oracle.pack_lower(np.linalg.cholesky(Sp)))
    spl = mhx.StaticMH(dist) if static else mhx.RWMH(dist)
    init = None if rng.integers(0, 2) else (rng.normal(size=(d, C)) * 0.5).astype(np.float32)
    seed, first = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 1 << 33))
    # the ziggurat generator (separable targets, ISO / DIAG proposals: the cooperative kernel; fp32 too since round 6) on a third of the
    # eligible cases
    zig = tname != "corr" and pname != "dense" and bool(rng.integers(0, 3) == 0)
    chain = mhx.sample(mhx.DensityModel(tgt), spl, N, C, seed=seed, first_chain=first, initial_params=init,
                       discard_initial=di, thinning=th, normal_gen="ziggurat" if zig else None)
    L = chain.stats["reduce_lanes"]
    assert chain.stats["normal_gen"] == (1 if zig else 0)
    ref = oracle.rwmh(ot(L), oracle.Proposal(mean=mean, static=static, normal_gen=1 if zig else 0, **op), oracle.schedule(N, di, th),
                      seed, first, C, init=init)
    what = "case %d: d=%d C=%d %s/%s static=%s mean=%s zig=%s variant=%d L=%d" % (case, d, C, tname, pname, static, mean is not None, zig,
                                                                               chain.stats["kernel_variant"], L)
    _same(chain.value, ref["samples"], what)
    _same(chain.accepted, ref["accepted"], what)
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], what)
    _same(cnt, ref["accept_counts"], what)


@pytest.mark.parametrize("case", soak_tail(range(25), 9))
def test_emcee_random_configurations(mhx, oracle, case, real):
    rng = np.random.default_rng(2000 + case + 100000 * SEED_OFFSET)
    d = int(rng.choice([1, 2, 3, 6, 11, 20, 33, 64, 90]))
    W = int(rng.choice([2, 3, 10, 65, 128, 131]))
    N, di, th = _schedule(rng)
    corr = bool(rng.integers(0, 2)) and d >= 2
    Sig = cases.sigma_ar1(d, 0.8)
    tgt, ot = (mhx.CorrGaussian(Sig), lambda L: oracle.corr_gauss_from_cov(Sig, reduce_lanes=L)) if corr else \
              (mhx.IsoGaussian(d), lambda L: oracle.iso_gauss(d))
    a = float(np.float32(1.5 + rng.random()))
    init = cases.emcee_init(d, W, case)
    seed, ens = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 1000))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I), a))
    chain = mhx.sample(mhx.DensityModel(tgt), spl, N, seed=seed, first_chain=ens, initial_params=init, discard_initial=di, thinning=th)
    L = chain.stats["reduce_lanes"]
    ref = oracle.emcee(ot(L), a, 1, oracle.schedule(N, di, th), seed, ens, W, init)
    what = "case %d: d=%d W=%d corr=%s variant=%d L=%d" % (case, d, W, corr, chain.stats["kernel_variant"], L)
    _same(chain.value, ref["samples"], what)
    _same(chain.accepted, ref["accepted"], what)
    x, lp, cnt = chain.state.state()
    _same(x, ref["final_x"], what)
    _same(lp, ref["final_lp"], what)


@pytest.mark.parametrize("case", soak_tail(range(25), 9))
def test_ram_random_configurations(mhx, oracle, case, real):
    rng = np.random.default_rng(3000 + case + 100000 * SEED_OFFSET)
    d = int(rng.choice([1, 2, 4, 9, 16, 17, 33, 47, 65, 100]))
    C = int(rng.choice([1, 3, 4, 5, 9, 33]))
    N = int(rng.integers(2, 9))
    warm = int(rng.integers(0, N + 2))
    corr = bool(rng.integers(0, 2)) and d >= 2
    Sig = cases.sigma_ar1(d, 0.6)
    tgt, ot = (mhx.CorrGaussian(Sig), oracle.corr_gauss_from_cov(Sig)) if corr else (mhx.IsoGaussian(d), oracle.iso_gauss(d))
    bounds = (0.0, float("inf")) if rng.integers(0, 2) else (0.8, 1.3)
    gamma = float(np.float32(0.51 + 0.4 * rng.random()))
    init = None if rng.integers(0, 2) else (rng.normal(size=(d, C)) * 0.3).astype(np.float32)
    seed, first = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 1 << 20))
    spl = mhx.RobustAdaptiveMetropolis(γ=gamma, eigenvalue_lower_bound=bounds[0], eigenvalue_upper_bound=bounds[1])
    chain = mhx.sample(mhx.DensityModel(tgt), spl, N, C, seed=seed, first_chain=first, initial_params=init, num_warmup=warm,
                       discard_initial=0)
    ref = oracle.ram(ot, oracle.schedule(N, 0, 1, warm), seed, first, C, init=init, gamma=gamma, eig_lo=bounds[0], eig_hi=bounds[1])
    what = "case %d: d=%d C=%d corr=%s warm=%d bounds=%s" % (case, d, C, corr, warm, bounds)
    _same(chain.value, ref["samples"], what)
    S, st = chain.state.factor()
    _same(S, ref["S"], what)
    _same(st, ref["status"], what)


@pytest.mark.parametrize("case", soak_tail(range(30), 10))
def test_ram_deferred_factor_random_configurations(mhx, oracle, case, real):
    """the deferred-factor form (MHX_FLAG_RAM_DEFERRED, spec 3.12) against its own twin: every rows-per-lane shape, blocks cut short by
    the warm-up's end and by the end of the call, thinning and discards, bounds that refuse updates, both targets, a given factor"""
    rng = np.random.default_rng(7000 + case + 100000 * SEED_OFFSET)
    d = int(rng.choice([1, 2, 4, 9, 17, 33, 64, 65, 100, 129, 200, 256]))
    C = int(rng.choice([1, 3, 4, 5, 9]))
    N = int(rng.integers(2, 14))
    th = int(rng.choice([1, 1, 2, 3]))
    di = int(rng.integers(0, 12))
    nT = di + (N - 1) * th
    warm = int(rng.integers(0, nT + 2))
    corr = bool(rng.integers(0, 2)) and d >= 2
    Sig = cases.sigma_ar1(d, 0.6)
    tgt, ot = (mhx.CorrGaussian(Sig), oracle.corr_gauss_from_cov(Sig)) if corr else (mhx.IsoGaussian(d), oracle.iso_gauss(d))
    s0 = 2.38 / np.sqrt(d)
    bounds = (0.0, float("inf")) if rng.integers(0, 2) else (0.9 * s0, 1.2 * s0)
    gamma = float(np.float32(0.51 + 0.4 * rng.random()))
    init = rng.normal(size=(d, C))
    seed, first = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 1 << 20))
    S0 = np.eye(d) * s0
    spl = mhx.RobustAdaptiveMetropolis(γ=gamma, S=S0, eigenvalue_lower_bound=bounds[0], eigenvalue_upper_bound=bounds[1], deferred_factor=True)
    chain = mhx.sample(mhx.DensityModel(tgt), spl, N, C, seed=seed, first_chain=first, initial_params=init, num_warmup=warm,
                       discard_initial=di, thinning=th)
    ref = oracle.ram_deferred(ot, oracle.schedule(N, di, th, warm), seed, first, C, init=init, gamma=gamma, eig_lo=bounds[0], eig_hi=bounds[1],
                              S_in=np.tile(oracle.pack_lower(S0), (C, 1)))
    what = "case %d: d=%d C=%d corr=%s N=%d di=%d th=%d warm=%d bounds=%s" % (case, d, C, corr, N, di, th, warm, bounds)
    _same(chain.value, ref["samples"], what)
    _same(chain.accepted, ref["accepted"], what)
    S, st = chain.state.factor()
    _same(S, ref["S"], what)
    _same(st, ref["status"], what)
    lo, hi = chain.state.diag_range()
    _same(lo, ref["diag_min"], what)
    _same(hi, ref["diag_max"], what)


@pytest.mark.parametrize("case", soak_tail(range(16), 6))
def test_mala_random_configurations(mhx, oracle, case, real):
    rng = np.random.default_rng(4000 + case + 100000 * SEED_OFFSET)
    d = int(rng.choice([1, 2, 3, 7, 16, 33, 70]))
    C = int(rng.choice([1, 3, 64, 65, 200]))
    N, di, th = _schedule(rng)
    tname = str(rng.choice(["iso", "corr", "banana", "funnel"])) if d >= 2 else "iso"
    Sig = cases.sigma_ar1(d, 0.4)
    if tname == "iso":
        tgt, ot = mhx.IsoGaussian(d), oracle.iso_gauss(d)
    elif tname == "corr":
        tgt, ot = mhx.CorrGaussian(Sig), oracle.corr_gauss_from_cov(Sig)
    elif tname == "banana":
        tgt, ot = mhx.Banana(d, 0.03), oracle.Target(oracle.TARGET_BANANA, d, params=[0.03])
    else:
        tgt, ot = mhx.Funnel(d), oracle.Target(oracle.TARGET_FUNNEL, d)
    s2 = float(np.float32(0.05 + 0.3 * rng.random()))
    init = (rng.normal(size=(d, C)) * 0.4).astype(np.float32)
    seed, first = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 1 << 20))
    chain = mhx.sample(mhx.DensityModel(tgt), mhx.MALA(s2), N, C, seed=seed, first_chain=first, initial_params=init,
                       discard_initial=di, thinning=th)
    ref = oracle.mala(ot.with_lanes(chain.stats["reduce_lanes"]), s2, oracle.schedule(N, di, th), seed, first, C, init)
    what = "case %d: d=%d C=%d %s, %d lane(s)" % (case, d, C, tname, chain.stats["reduce_lanes"])
    _same(chain.value, ref["samples"], what)
    _same(chain.accepted, ref["accepted"], what)


@pytest.mark.soak_f32
def test_dimensions_beyond_the_specialised_kernels(mhx, oracle, real):
    """Sizes past every specialised kernel's range fall back to the cooperative / run-time-dimension / generic kernels
    and stay bit-exact: RWMH d = 4000 (64 lanes per chain), emcee d = 300 (isotropic and dense target), RAM d = 1024,
    MALA d = 500, RWMH on a dense target at d = 300."""
    def same(a, b, what):
        _same(a, b, what)
    d, C = 4000, 96
    s = float(np.float32(2.38 / d ** 0.5))
    ch = mhx.sample(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), 4, C, seed=3)
    L = ch.stats["reduce_lanes"]
    assert L == (1 if real == "f64" else 64)          # 1000 Philox blocks: 16 per lane of a wave (fp64 holds 8: state in HBM)
    same(ch.value, oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(4), 3, 0, C)["samples"], "rwmh d=4000")
    d, W = 300, 500
    init = cases.emcee_init(d, W, 1)
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    ch = mhx.sample(mhx.DensityModel(mhx.IsoGaussian(d)), spl, 3, seed=2, initial_params=init)
    same(ch.value, oracle.emcee(oracle.iso_gauss(d), 2.0, 1, oracle.schedule(3), 2, 0, W, init)["samples"], "emcee d=300")
    Sig = cases.sigma_ar1(d, 0.5)
    ch = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), spl, 3, seed=2, initial_params=init)
    assert ch.stats["kernel_variant"] == 0
    same(ch.value, oracle.emcee(oracle.corr_gauss_from_cov(Sig), 2.0, 1, oracle.schedule(3), 2, 0, W, init)["samples"], "emcee dense d=300")
    ch = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), 0.01 * mhx.I)), 4, 40, seed=7)
    # the matrix-core kernel with the factor image streamed through LDS (reduction shape 4); in fp64 with the chain state
    # re-read from its slab (75 reals per lane are past the register-resident budget)
    assert (ch.stats["kernel_variant"], ch.stats["reduce_lanes"]) == (8, 4)
    same(ch.value, oracle.rwmh(oracle.corr_gauss_from_cov(Sig, reduce_lanes=ch.stats["reduce_lanes"]), oracle.Proposal(oracle.PROP_ISO, 0.01 ** 0.5),
                               oracle.schedule(4), 7, 0, 40)["samples"], "rwmh dense d=300")
    Sig6 = cases.sigma_ar1(600, 0.5)                          # fp64: past every matrix-core variant (150 reals per lane), state in HBM
    ch = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig6)), mhx.RWMH(mhx.MvNormal(mhx.zeros(600), 0.01 * mhx.I)), 3, 20, seed=7,
                    flags=0 if real == "f64" else mhx.FLAG_GENERIC)
    assert ch.stats["kernel_variant"] == 0
    same(ch.value, oracle.rwmh(oracle.corr_gauss_from_cov(Sig6), oracle.Proposal(oracle.PROP_ISO, 0.01 ** 0.5),
                               oracle.schedule(3), 7, 0, 20)["samples"], "rwmh dense d=600")
    d, C = 1024, 6
    ch = mhx.sample(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RobustAdaptiveMetropolis(), 4, C, seed=5, num_warmup=4, initial_params=np.zeros(d))
    ref = oracle.ram(oracle.iso_gauss(d), oracle.schedule(4, 4, 1, 4), 5, 0, C, init=np.zeros((d, C), dtype=np.float32))
    same(ch.value, ref["samples"], "ram d=1024")
    same(ch.state.factor()[0], ref["S"], "ram d=1024 factor")
    d, C = 500, 100
    init = (np.random.default_rng(1).normal(size=(d, C)) * 0.1).astype(np.float32)
    ch = mhx.sample(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.MALA(0.01), 4, C, seed=6, initial_params=init)
    same(ch.value, oracle.mala(oracle.iso_gauss(d, reduce_lanes=ch.stats["reduce_lanes"]), 0.01, oracle.schedule(4), 6, 0, C, init)["samples"], "mala d=500")


@pytest.mark.parametrize("case", soak_tail(range(16), 6))
def test_large_dimension_shapes_random_configurations(mhx, oracle, case, real):
    """One or two chains per wave (d = 130 ... 1000): state, moments and the per-step record move through LDS as whole row
    segments when all chains of a block exist, element-wise otherwise -- odd chain counts exercise both in one run; both
    generators, recorded and moments-only runs, a continued call."""
    rng = np.random.default_rng(5000 + case + 100000 * SEED_OFFSET)
    d = int(rng.choice([130, 200, 257, 515, 1000]))
    C = int(rng.choice([1, 3, 4, 5, 8, 9, 37, 64]))
    lanes = int(rng.choice([0, 32, 64]))
    N, di, th = int(rng.integers(1, 6)), int(rng.integers(0, 3)), int(rng.integers(1, 3))
    tname = str(rng.choice(["iso", "funnel", "banana"]))
    zig = real == "f64" and bool(rng.integers(0, 2))
    if lanes and -(-((d + 3) // 4) // lanes) > (13 if real == "f64" else 16):
        pytest.skip("more blocks per lane than the cooperative kernel holds")
    tgt = {"iso": mhx.IsoGaussian(d), "funnel": mhx.Funnel(d), "banana": mhx.Banana(d, 0.03)}[tname]
    s = float(np.float32(0.1))
    seed = 7 + case
    run = mhx.Run(mhx.DensityModel(tgt), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=seed, first_chain=case,
                  reduce_lanes=lanes, normal_gen="ziggurat" if zig else None)
    run.init(None)
    moments = bool(rng.integers(0, 2))
    if moments:
        run.sample(N, di, th, 0, save="moments")
        run.sample(N, th, th, 0, save="moments")
    else:
        run.sample(N, di, th, 0)
    L = run.stats()["reduce_lanes"]
    ot = {"iso": oracle.iso_gauss(d, reduce_lanes=L), "funnel": oracle.Target(oracle.TARGET_FUNNEL, d, reduce_lanes=L),
          "banana": oracle.Target(oracle.TARGET_BANANA, d, params=[0.03], reduce_lanes=L)}[tname]
    prop = oracle.Proposal(oracle.PROP_ISO, s, normal_gen=1 if zig else 0)
    what = "case %d: d=%d C=%d lanes=%d L=%d %s zig=%s moments=%s" % (case, d, C, lanes, L, tname, zig, moments)
    if moments:
        total = di + 1 + (N - 1) * th + N * th                # transitions of both calls
        ref = oracle.rwmh(ot, prop, oracle.schedule(1, total - 1), seed, case, C)
        _same(run.state()[0], ref["final_x"], what)
    else:
        ref = oracle.rwmh(ot, prop, oracle.schedule(N, di, th), seed, case, C)
        v, a = run.samples()
        _same(v, ref["samples"], what)
        _same(a, ref["accepted"], what)
    run.close()


@pytest.mark.parametrize("case", soak_tail(range(24), 8))
def test_dense_factor_ensembles_random_configurations(mhx, oracle, case, real, engine):
    """The scalar-factor form of the cooperative stretch move (round 4) over random dense factors: dimension 8 ... 68 (odd ones,
    multiples of 4 and of 16), odd and tiny ensembles (a single block, ragged last blocks, halves of different size), random
    waves per block / walkers per block / operand mode, thinning and a discarded prefix, a resumed call, both widths."""
    rng = np.random.default_rng(7000 + case + 100000 * SEED_OFFSET)
    d = int(rng.choice([8, 9, 12, 15, 16, 17, 23, 31, 32, 33, 47, 48, 49, 50, 63, 64, 65, 68]))
    W = int(rng.choice([2 * d + 2, 67, 128, 129, 193, 320, 1025]))
    W = max(W, 4)
    N, di, th = _schedule(rng)
    knobs = {}
    if rng.integers(0, 2):
        knobs["MHX_EMCEE_SCALAR"] = str(rng.choice([4, 8, 16]))
    if rng.integers(0, 2):
        knobs["MHX_EMCEE_SCAL_WPB"] = str(rng.choice([16, 32, 64]))
    if rng.integers(0, 4) == 0:
        knobs["MHX_EMCEE_SCAL_MODE"] = "0"
    if rng.integers(0, 4) == 0:
        knobs["MHX_EMCEE_SCAL_REC"] = "0"
    if "MHX_EMCEE_SCALAR" not in knobs:
        knobs["MHX_EMCEE_MFMA"] = str(int(rng.integers(0, 2)))   # the matrix-core form, d <= 64 (fp32: 128), or explicitly not
    for k, v in knobs.items():
        engine.setenv(k, v)
    A = rng.normal(size=(d, d))
    Sig = A @ A.T / d + np.diag(0.2 + rng.random(d))               # a dense SPD matrix: no band
    seed, ens = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 1 << 20))
    a = float(np.float32(1.5 + rng.random()))
    init = None if rng.integers(0, 2) else (rng.normal(size=(d, W)) * 0.5).astype(np.float32)
    prior = mhx.MvNormal(mhx.zeros(d), mhx.I)
    run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(W, mhx.StretchProposal(prior, a)), seed=seed, first_chain=ens)
    run.init(init)
    run.sample(N, di, th, 0)
    st = run.stats()
    L = st["reduce_lanes"]
    mfma = knobs.get("MHX_EMCEE_MFMA") == "1" and d >= 8 and d <= (64 if real == "f64" else 128)
    if mfma:
        assert st["kernel_variant"] == 10 and L == 4, (knobs, st)
    elif d * (2 if real == "f64" else 1) <= 136 and int(knobs.get("MHX_EMCEE_SCALAR", 8)) <= d:
        assert st["kernel_variant"] == 9, (knobs, st)
    val, acc = run.samples()
    what = "case %d: d=%d W=%d N=%d di=%d th=%d a=%g knobs=%r variant=%d L=%d" % (case, d, W, N, di, th, a, knobs, st["kernel_variant"], L)
    ot = oracle.corr_gauss_from_cov(Sig, reduce_lanes=L)
    ref = oracle.emcee(ot, a, 1, oracle.schedule(N, di, th), seed, ens, W, init, prior=oracle.Proposal(oracle.PROP_ISO, 1.0))
    _same(val, ref["samples"], what)
    _same(acc, ref["accepted"], what)
    # a second call continues the same ensemble (the oracle: one longer schedule)
    run.sample(3, 1, 1, 0)
    val2, _ = run.samples()
    nT = di + (N - 1) * th
    ref2 = oracle.emcee(ot, a, 1, oracle.schedule(3, nT + 1, 1), seed, ens, W, init, prior=oracle.Proposal(oracle.PROP_ISO, 1.0))
    _same(val2, ref2["samples"], what + " (resumed)")
    run.close()


def _banded_sigma(d, bw, rng):
    A = np.zeros((d, d))
    for r in range(d):
        A[r, r] = 1.0 + rng.uniform(0.0, 1.0)
        for c in range(max(0, r - bw), r):
            A[r, c] = rng.normal() * 0.4
    return np.linalg.inv(A.T @ A)


@pytest.mark.parametrize("case", soak_tail(range(24), 8))
def test_ensemble_sweep_launches_random_configurations(mhx, oracle, case, real, engine):
    """One launch per sweep (round 4) over random ensembles: banded factors on the lane-group form, dense ones on the scalar-factor or
    the matrix-core form, user-style targets on the lane-per-walker kernel; dimensions with and without padding, odd and tiny
    ensembles, random lanes per walker, thinning with a discarded prefix, a resumed call; and the same run as two half-step launches
    per sweep (MHX_EMCEE_FUSED=0) must give the same tensor."""
    rng = np.random.default_rng(9100 + case + 100000 * SEED_OFFSET)
    kind = ["band", "dense", "mfma", "banana"][case % 4]
    d = int(rng.choice([8, 9, 12, 16, 17, 23, 32, 33, 48, 50, 63, 64]))
    W = int(rng.choice([2, 3, 2 * d + 2, 67, 128, 129, 193, 320, 1025]))
    N, di, th = _schedule(rng)
    lanes = 0
    if kind == "band":
        bw = int(rng.choice([0, 1, 2, 5, 8]))
        bw = min(bw, d - 2)
        Sig = _banded_sigma(d, bw, rng) if bw else np.diag(0.5 + rng.random(d))
        lanes = int(rng.choice([0, 0, 2, 4, 8, 16]))
        lanes = lanes if lanes <= d else 0
        spec = mhx.CorrGaussian(Sig)
    elif kind in ("dense", "mfma"):
        A = rng.normal(size=(d, d))
        Sig = A @ A.T / d + np.diag(0.2 + rng.random(d))
        spec = mhx.CorrGaussian(Sig)
        engine.setenv("MHX_EMCEE_MFMA", "1" if kind == "mfma" else "0")
    else:
        spec = mhx.Banana(d, 0.03)
        lanes = 1
        engine.setenv("MHX_EMCEE_PERSIST", "0")         # (small ensembles on this kernel would run as one persistent block)
    seed, ens = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 1 << 20))
    a = float(np.float32(1.5 + rng.random()))
    init = None if rng.integers(0, 2) else (rng.normal(size=(d, W)) * 0.5).astype(np.float32)
    prior = mhx.MvNormal(mhx.zeros(d), mhx.I)

    def go(fused):
        engine.setenv("MHX_EMCEE_FUSED", "1" if fused else "0")
        run = mhx.Run(mhx.DensityModel(spec), mhx.Ensemble(W, mhx.StretchProposal(prior, a)), seed=seed, first_chain=ens, reduce_lanes=lanes)
        run.init(init)
        run.sample(N, di, th, 0)
        st = run.stats()
        first = run.samples()
        run.sample(3, 1, 1, 0)
        second = run.samples()
        cnt = run.state()[2]
        run.close()
        return st, first, second, cnt

    st, first, second, cnt = go(True)
    st0, first0, second0, cnt0 = go(False)
    L = st["reduce_lanes"]
    nT = di + (N - 1) * th
    what = "case %d: %s d=%d W=%d N=%d di=%d th=%d a=%g lanes=%d variant=%d L=%d launches %d / %d" % (
        case, kind, d, W, N, di, th, a, lanes, st["kernel_variant"], L, st["launches"], st0["launches"])
    # (a shape whose three candidate rows per walker do not fit the block's LDS keeps its half-steps: few lanes per walker)
    assert st0["launches"] in (st["launches"], 2 * st["launches"]) and st["kernel_variant"] == st0["kernel_variant"], what
    if kind != "band" or lanes in (0, 8, 16):
        assert st0["launches"] == 2 * st["launches"], what
    if kind == "mfma":
        assert st["kernel_variant"] == 10 and L == 4, what
    for u, v, w in ((first, first0, "first call"), (second, second0, "resumed call")):
        _same(u[0], v[0], what + ": one launch vs two, " + w)
        _same(u[1], v[1], what + ": accepted, " + w)
    _same(cnt, cnt0, what + ": counters")
    ot = oracle.Target(oracle.TARGET_BANANA, d, params=[0.03]) if kind == "banana" else oracle.corr_gauss_from_cov(Sig, reduce_lanes=L)
    pr = oracle.Proposal(oracle.PROP_ISO, 1.0)
    ref = oracle.emcee(ot, a, 1, oracle.schedule(N, di, th), seed, ens, W, init, prior=pr)
    _same(first[0], ref["samples"], what)
    _same(first[1], ref["accepted"], what)
    ref2 = oracle.emcee(ot, a, 1, oracle.schedule(3, nT + 1, 1), seed, ens, W, init, prior=pr)
    _same(second[0], ref2["samples"], what + " (resumed)")
