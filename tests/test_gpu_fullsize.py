"""GPU, at BASELINE.json's FULL sizes: the HIP path on the whole workload, checked (a) bit for bit against the
oracle on subsets of chains (chains are independent and carry global ids, so any subset is reproducible on the
CPU in seconds) and (b) through size-independent properties (shard concatenation, acceptance statistics,
every walker of an ensemble against the full-ensemble oracle)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _the_products_compiler(product_jit):
    """the bench's own configurations: their run-time kernels by the compiler the product picks (tests/conftest.py)"""


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    assert a.dtype == b.dtype, "%s: dtypes %s / %s" % (what, a.dtype, b.dtype)
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s" % (what, len(bad), bad[0])


def test_c2_full_size_rwmh(mhx, oracle, real):
    """configs[1]: isotropic 100-dim Gaussian, RWMH, 65 536 chains."""
    d, C, N = 100, 65536, 40
    s = float(np.float32(2.38 / d ** 0.5))
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
    chain = mhx.sample(model, spl, N, C, seed=0xC0FFEE)
    L = chain.stats["reduce_lanes"]
    assert chain.stats["kernel_variant"] == 3 and L == 2   # the pre-built cooperative kernel
    for first in (0, 31337, C - 64):                               # three subsets of 64 chains
        ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(N),
                          0xC0FFEE, first, 64)
        _same(chain.value[:, :, first:first + 64], ref["samples"], "samples of chains %d.." % first)
        _same(chain.accepted[:, first:first + 64], ref["accepted"], "accepted")
    # lp column is the log-density of the recorded state for every chain
    x = chain.value[-1, :d, :].astype(np.float64)
    lp = -0.5 * (x * x).sum(axis=0) - 0.5 * d * np.log(2 * np.pi)
    assert np.abs(chain.value[-1, d, :] - lp).max() < 2e-3
    # accepted flag <=> the state changed; the device total equals the sum of the flags
    moved = (np.diff(chain.value[:, 0, :], axis=0) != 0)
    assert np.array_equal(moved, chain.accepted[1:].astype(bool))
    assert chain.stats["accepted"] == int(chain.accepted[1:].sum())
    # two half-size shards reproduce the whole
    half = mhx.sample(model, spl, 8, C // 2, seed=0xC0FFEE, first_chain=C // 2, reduce_lanes=L)
    _same(half.value, chain.value[:8, :, C // 2:], "upper shard")


def test_c3_full_size_emcee(mhx, oracle, real):
    """configs[2]: Ensemble(16 384, StretchProposal), 50-dim correlated Gaussian -- the whole ensemble against the oracle."""
    d, W, N = 50, 16384, 4
    Sig = cases.sigma_ar1(d, 0.9)
    init = cases.emcee_init(d, W, 11)
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))),
                       N, seed=3, initial_params=init)
    L = chain.stats["reduce_lanes"]
    assert L > 1 and chain.stats["kernel_variant"] == 4
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=L), 2.0, 1, oracle.schedule(N), 3, 0, W, init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert chain.stats["accepted"] == int(ref["accept_counts"].sum())


@pytest.mark.parametrize("form", ["matrix-core", "scalar-factor"])
def test_c3_rotated_full_size_emcee(mhx, oracle, real, form, engine):
    """SURVEY 8(d) C3 "also a dense-rotated variant": Sigma = Q (0.9^|i-j|) Q^T, no structural zeros in the factor -- the whole
    16 384-walker ensemble against the oracle on the matrix-core form (variant 10, reduction shape 4: the default) and on the
    scalar-factor form (variant 9, reduction shape 8: MHX_EMCEE_MFMA=0)."""
    want_variant, want_L = (10, 4) if form == "matrix-core" else (9, 8)
    engine.setenv("MHX_EMCEE_MFMA", "1" if form == "matrix-core" else "0")    # (default: matrix-core in fp64 at this size)
    d, W, N = 50, 16384, 4
    Q, _ = np.linalg.qr(np.random.default_rng(50).normal(size=(d, d)))
    Sig = Q @ cases.sigma_ar1(d, 0.9) @ Q.T
    init = cases.emcee_init(d, W, 11)
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))),
                       N, seed=3, initial_params=init)
    assert chain.stats["kernel_variant"] == want_variant and chain.stats["reduce_lanes"] == want_L and chain.stats["factor_band"] == -1
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig, reduce_lanes=want_L), 2.0, 1, oracle.schedule(N), 3, 0, W, init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert chain.stats["accepted"] == int(ref["accept_counts"].sum())


def test_c4_full_size_ram(mhx, oracle, real):
    """configs[3]: RobustAdaptiveMetropolis, 200-dim Gaussian with kappa = 1e3, 32 768 chains (5.3 GB of factors)."""
    d, C, N = 200, 32768, 6
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.normal(size=(d, d)))
    Sig = (Q * 1e3 ** (np.arange(d) / (d - 1.0))) @ Q.T
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(), nchains=C, seed=4)
    run.init(np.zeros(d))
    run.sample(N, 0, 1, 4)                                         # 4 adapting + 1 fixed transition
    val, acc = run.samples()
    S, status = run.factor()
    assert (status == 0).all()
    ot = oracle.corr_gauss_from_cov(Sig)
    for first in (0, C - 16):
        ref = oracle.ram(ot, oracle.schedule(N, 0, 1, 4), 4, first, 16, init=np.zeros((d, 16), dtype=np.float32))
        _same(val[:, :, first:first + 16], ref["samples"], "samples of chains %d.." % first)
        _same(S[first:first + 16], ref["S"], "factors")
    # every chain's factor stays lower-triangular positive: packed diagonal entries > 0
    diag_idx = np.array([i * (i + 1) // 2 + i for i in range(d)])
    assert (S[:, diag_idx] > 0).all()
    run.close()


def test_c4_full_size_is_deterministic_under_load(mhx, real):
    """The sweep stores whole row slots and relies on a wave's stores to one address landing in program order
    (DESIGN.md 6.3); the spot-checked chains above could miss a rare reordering under full load, so the whole
    2.6 GB of factors of two identical runs must agree bit for bit, and every factor must be finite."""
    d, C = 200, 32768
    model = mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.9)))
    out = []
    for rep in range(2):
        run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(), nchains=C, seed=8)
        run.init(np.zeros((d, C), dtype=np.float32))
        run.sample(1, 12, 1, 12, save=False)
        S, st = run.factor()
        out.append((S.copy(), st.copy()))
        run.close()
    assert np.isfinite(out[0][0]).all()
    assert np.array_equal(cases.bits(out[0][0]), cases.bits(out[1][0]))
    assert np.array_equal(out[0][1], out[1][1])


def test_c5_shard_size_rwmh(mhx, oracle, real):
    """configs[4], one GPU's shard: 1000-dim funnel, 32 768 chains with global ids of shard 5 of 8."""
    d, C, N = 1000, 32768, 5
    s = float(np.float32(2.38 / d ** 0.5))
    first = 5 * C
    chain = mhx.sample(mhx.DensityModel(mhx.Funnel(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), N, C, seed=5,
                       first_chain=first)
    L = chain.stats["reduce_lanes"]
    assert chain.stats["kernel_variant"] == 3 and L == (64 if real == "f64" else 32)
    ot = oracle.Target(oracle.TARGET_FUNNEL, d).with_lanes(L)
    for off in (0, C - 32):
        ref = oracle.rwmh(ot, oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(N), 5, first + off, 32)
        _same(chain.value[:, :, off:off + 32], ref["samples"], "samples of chains %d.." % (first + off))


@pytest.mark.parametrize("target", ["funnel", "banana"])
def test_c5_as_specified_eight_shards_on_one_device(mhx, oracle, target, real):
    """configs[4] as written -- 1000 dimensions, 262 144 chains over 8 GPUs, R-hat by one reduction -- with the eight
    32 768-chain shards run one after the other on this device (global chain ids, running moments of the thinned states):
    the per-shard sums the ranks would all-reduce (mhx.dist.pack_stats) add up to the sums of ONE unsharded 262 144-chain
    run, chains of the first, a middle and the last shard are the oracle's bit for bit, and the combined R-hat is finite."""
    from mhx import api, dist
    d, Cs, G, N, thin = 1000, 32768, 8, 3, 2
    s = float(np.float32(2.38 / d ** 0.5))
    tgt = mhx.Funnel(d) if target == "funnel" else mhx.Banana(d, 0.03)
    ot = oracle.Target(oracle.TARGET_FUNNEL, d) if target == "funnel" else oracle.Target(oracle.TARGET_BANANA, d, params=[0.03])
    model, spl = mhx.DensityModel(tgt), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
    packed, lanes = [], None
    for g in range(G):
        run = mhx.Run(model, spl, nchains=Cs, seed=5, first_chain=g * Cs)
        run.init(None)
        run.sample(N, 1, thin, 0, save="moments")
        st = run.stats()
        lanes = st["reduce_lanes"]
        dg = run.diagnostics()
        packed.append(dist.pack_stats(dg, st["accepted"], st["transitions"]))
        if g in (0, 3, 7):                                              # parity of a few chains of this shard
            x, lp, cnt = run.state()
            nt = 1 + thin * (N - 1)
            ref = oracle.rwmh(ot.with_lanes(lanes), oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(nt + 1), 5, g * Cs + Cs - 16, 16)
            _same(x[:, Cs - 16:], ref["final_x"], "final states, shard %d" % g)
            _same(cnt[Cs - 16:], ref["accept_counts"], "accept counts, shard %d" % g)
        run.close()
    v = np.sum(packed, axis=0)                                           # what the all-reduce leaves on every rank
    d1 = (v.size - 3) // 3
    assert int(round(v[3 * d1 + 2])) == G * Cs
    comb = api.combine_diagnostics(v[:d1], v[d1:2 * d1], v[2 * d1:3 * d1], G * Cs, N)
    assert np.isfinite(comb["rhat"]).all() and (comb["rhat"] > 0.9).all()
    # the same run unsharded: 262 144 chains at once
    run = mhx.Run(model, spl, nchains=G * Cs, seed=5, first_chain=0)
    run.init(None)
    run.sample(N, 1, thin, 0, save="moments")
    dg = run.diagnostics()
    st = run.stats()
    whole = dist.pack_stats(dg, st["accepted"], st["transitions"])
    run.close()
    assert st["reduce_lanes"] == lanes
    assert v[3 * d1] == whole[3 * d1] and v[3 * d1 + 1] == whole[3 * d1 + 1]          # accepted, transitions: integers
    assert np.allclose(v[:3 * d1], whole[:3 * d1], rtol=1e-9 if real == "f64" else 2e-4, atol=1e-6 if real == "f64" else 1e-1)


def test_state_slab_beyond_four_gigabytes(mhx, oracle, real):
    """Maximum sizes: the register / cooperative kernels address the [dim+1][nchains] slab with 32-bit byte offsets, so a run
    whose slab reaches 4 GB must take the 64-bit state-in-HBM kernel on its own -- and still be the oracle's chains, first
    and last (the last chains sit behind the 4 GB mark)."""
    d = 100
    rb = 8 if real == "f64" else 4
    C = ((1 << 32) // ((d + 1) * rb) + 4096) // 64 * 64          # just past 2^32 bytes
    s = float(np.float32(2.38 / d ** 0.5))
    run = mhx.Run(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=11)
    run.init(None)
    run.sample(1, 3, 1, 0, save=False)
    st = run.stats()
    assert st["kernel_variant"] == 0 and st["reduce_lanes"] == 1 and st["transitions"] == 3 * C
    x, lp, cnt = run.state()
    run.close()
    assert (d + 1) * rb * C >= 1 << 32
    for first in (0, C - 8):
        ref = oracle.rwmh(oracle.iso_gauss(d), oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(4), 11, first, 8)
        _same(x[:, first:first + 8], ref["final_x"], "final states of chains %d.." % first)
        _same(lp[first:first + 8], ref["final_lp"], "final lp")
        _same(cnt[first:first + 8], ref["accept_counts"], "accept counts")


def test_c2_posterior_moments_at_scale(mhx, real):
    """Known answer at the headline size: 65 536 chains x 40 000 transitions on the 100-dim standard normal, running moments of
    every 40th state after 8 000 discarded -- pooled mean 0, within-chain variance 1, R-hat ~ 1 (north_star: posterior mean / std
    within tolerance).  6.5e7 kept draws per parameter: the pooled mean's Monte Carlo error is ~ 1e-3 at tau ~ 300."""
    d, C = 100, 65536
    s = float(np.float32(2.38 / d ** 0.5))
    run = mhx.Run(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=C, seed=2024)
    run.init(None)
    run.sample(800, 8000, 40, 0, save="moments")
    dg = run.diagnostics()
    st = run.stats()
    run.close()
    assert 0.22 < st["accepted"] / st["transitions"] < 0.26                      # the 0.234 regime
    assert np.abs(dg["mean"][:d]).max() < 6e-3, np.abs(dg["mean"][:d]).max()
    assert np.abs(dg["W"][:d] - 1.0).max() < 0.02, np.abs(dg["W"][:d] - 1.0).max()
    assert (dg["rhat"][:d] < 1.02).all() and (dg["rhat"][:d] > 0.99).all()
    assert abs(dg["mean"][d] + 0.5 * d * (1.0 + np.log(2 * np.pi))) < 0.05         # E[lp] = -d/2 (1 + log 2 pi)


def test_ram_adaptation_reaches_its_target_over_many_chains(mhx, real):
    """RobustAdaptiveMetropolis drives the acceptance towards alpha = 0.234 (src/RobustAdaptiveMetropolis.jl:153-173): 32 768
    chains on an 8-dim Gaussian with kappa = 100 from a random start, S0 = I.  The acceptance of transitions 6 000 - 7 000 (all chains) has moved from its early
    value towards the target, no factor left the positive-definite cone, eta = n^-0.6 with the
    iteration before its increment.  (At d = 200 the same pull takes O(1e5) transitions: a rank-1 step moves one direction.)"""
    d, C = 8, 32768
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.normal(size=(d, d)))
    Sig = (Q * 100.0 ** (np.arange(d) / (d - 1.0))) @ Q.T
    x0 = (np.linalg.cholesky(Sig) @ rng.normal(size=(d, C))).astype(np.float32)
    run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(Sig)), mhx.RobustAdaptiveMetropolis(), nchains=C, seed=6)
    run.init(x0)
    marks = []
    for nsteps in (500, 500, 5000, 1000):
        run.sample(1, nsteps, 1, nsteps + 1, save=False)
        marks.append(run.stats())
    S, status = run.factor()
    ad = run.adapt_state()
    run.close()
    rate = lambda m: m["accepted"] / float(m["transitions"])              # mhx_stats counts the last call
    early, late = rate(marks[1]), rate(marks[3])
    # from S0 = I (too small for variances up to 100) the acceptance starts high and is pulled down towards 0.234; with
    # gamma = 0.6 the pull is slow -- 0.35 after 7 000 transitions (the oracle, bit for bit the same chains, says the same)
    assert 0.234 - 0.02 < late < early - 0.05 and late < 0.40, (early, late)
    assert (status == 0).all() and np.isfinite(S).all()
    assert ad["iteration"] == 7001 and abs(ad["η"] - 7000.0 ** -0.6) < 1e-6 * ad["η"]


@pytest.mark.soak_f32
def test_c4_known_answers_at_dimension_200(mhx, real):
    """configs[3] at ITS dimension (the adaptation's known answer ran at d = 8 only).  Two answers at d = 200, kappa = 1e3,
    from the start that moves (x0 ~ target, S0 = 2.38/sqrt(d) I):
    (1) an exact invariant of ram_adapt (src/RobustAdaptiveMetropolis.jl:153-173): S S^T <- S (I + eta dalpha u u^T / |u|^2) S^T multiplies
        det(S S^T) by (1 + eta dalpha), so for EVERY chain  log det S_N - log det S_0 = 1/2 sum_n log(1 + eta_n (exp(logalpha_n) - alpha))
        over its adapting transitions -- checked per chain from the per-step statistics against the factors the device holds;
    (2) the pull towards alpha = 0.234: from 0.65 the mean acceptance probability over 32 768 chains falls monotonically (at d = 200 and
        gamma = 0.6 a rank-1 step moves one direction in two hundred: 0.647 -> 0.634 in 2 000 transitions, O(1e6) to arrive), every factor
        stays in the positive-definite cone."""
    import bench
    d = 200
    Sig = bench.sigma_illcond(d)
    Lc = np.linalg.cholesky(Sig)
    s0 = 2.38 / d ** 0.5
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    # (1)
    C, N = 1024, 160
    run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(S=s0 * np.eye(d)), nchains=C, seed=4)
    run.init(Lc @ np.random.default_rng(11).normal(size=(d, C)))
    run.sample(N, 0, 1, N)                                      # every transition adapts, every state recorded
    st = run.step_stats()
    S, status = run.factor()
    run.close()
    assert (status == 0).all()
    la = st["logα"].astype(np.float64)[1:]                      # sample n + 1 is behind transition n
    eta = st["η"][1:]
    assert np.allclose(eta, np.arange(1, N, dtype=np.float64) ** -0.6, rtol=1e-6 if real == "f32" else 1e-14)
    want = 0.5 * np.log1p(eta[:, None] * (np.exp(la) - 0.234)).sum(axis=0)          # [C]
    diag_idx = np.cumsum(np.arange(1, d + 1)) - 1               # packed lower, row-major: the diagonal entries
    got = np.log(S.astype(np.float64)[:, diag_idx]).sum(axis=1) - d * np.log(float(run.real(s0)))
    assert np.abs(got - want).max() < (2e-3 if real == "f32" else 1e-9), np.abs(got - want).max()
    assert (want > 0).all()                                     # acceptance above alpha: every factor grew
    # (2)
    C = 32768
    run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(S=s0 * np.eye(d)), nchains=C, seed=4)
    run.init(Lc @ np.random.default_rng(11).normal(size=(d, C)))
    marks = []
    for n in (200, 800, 1000):
        run.sample(1, n, 1, n, save=False)
        marks.append(float(np.exp(run.adapt_state()["logα"].astype(np.float64)).mean()))
    S, status = run.factor()
    ad = run.adapt_state()
    run.close()
    assert (status == 0).all() and np.isfinite(S).all()
    # standard error of a mean over 32 768 chains ~ 0.002; the fall over 2 000 transitions is 0.013
    assert 0.66 > marks[0] > marks[1] > marks[2] > 0.60 and marks[0] - marks[2] > 0.008, marks
    assert ad["iteration"] == 2001


@pytest.mark.soak_f32
def test_c4_deferred_factor_known_answers_and_divergence_at_dimension_200(mhx, real):
    """The deferred-factor form of RAM (MHX_FLAG_RAM_DEFERRED, arithmetic spec 3.12) has its own rounding, so it gets its own known
    answers at configs[3]'s dimension -- the two of test_c4_known_answers_at_dimension_200:
    (1) the log-det invariant of ram_adapt per chain -- exact for this form too: the diagonal of chol(I +- c^2 U U') multiplies to
        sqrt(1 +- eta |dalpha|);
    (2) the pull towards alpha over 32 768 chains;
    and (3) the DIVERGENCE from the sequential form on the same seeds: chains whose accept decisions all agree over 160 adapting
    transitions (a flipped decision sends a chain elsewhere for good), and for those the distance of the states and of S S'."""
    import bench
    import json
    import os
    d = 200
    Sig = bench.sigma_illcond(d)
    Lc = np.linalg.cholesky(Sig)
    s0 = 2.38 / d ** 0.5
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    C, N = 1024, 160
    x0 = Lc @ np.random.default_rng(11).normal(size=(d, C))
    out = {}
    for form in ("deferred", "sequential"):
        run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(S=s0 * np.eye(d), deferred_factor=form == "deferred"), nchains=C, seed=4)
        run.init(x0)
        run.sample(N, 0, 1, N)
        st = run.step_stats()
        val, acc = run.samples()
        S, status = run.factor()
        assert run.stats()["kernel_variant"] == (12 if form == "deferred" else 0)
        run.close()
        assert (status == 0).all()
        out[form] = (st, val, acc, S)
    # (1)
    st, val, acc, S = out["deferred"]
    la = st["logα"].astype(np.float64)[1:]
    eta = st["η"][1:]
    want = 0.5 * np.log1p(eta[:, None] * (np.exp(la) - 0.234)).sum(axis=0)
    diag_idx = np.cumsum(np.arange(1, d + 1)) - 1
    got = np.log(S.astype(np.float64)[:, diag_idx]).sum(axis=1) - d * np.log(float(np.float32(s0) if real == "f32" else s0))
    assert np.abs(got - want).max() < (2e-3 if real == "f32" else 1e-9), np.abs(got - want).max()
    assert (want > 0).all()
    # (3)
    _, val2, acc2, S2 = out["sequential"]
    same = (acc == acc2).all(axis=0)                             # [C]: every decision of the chain agrees
    frac = same.mean()
    il = np.tril_indices(d)
    rel = 0.0
    for c in np.flatnonzero(same)[:64]:
        A = np.zeros((d, d)); A[il] = S[c]
        B = np.zeros((d, d)); B[il] = S2[c]
        rel = max(rel, np.linalg.norm(A @ A.T - B @ B.T) / np.linalg.norm(B @ B.T))
    dx = np.abs(val[:, :d, same].astype(np.float64) - val2[:, :d, same].astype(np.float64)).max()
    report = dict(dtype=real, chains=C, transitions=N - 1, chains_with_identical_decisions=float(frac), max_rel_SSt=float(rel), max_abs_dx=float(dx))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "r05_c4_deferred_divergence_%s.json" % real), "w") as f:
        json.dump(report, f)
    if real == "f64":
        assert frac == 1.0 and rel < 1e-12 and dx < 1e-9, report
    else:
        assert frac > 0.5 and rel < 1e-4, report                 # fp32: a log-density of O(100) leaves margins of 1e-5 to round-off
    # (2)
    C = 32768
    run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(S=s0 * np.eye(d), deferred_factor=True), nchains=C, seed=4)
    run.init(Lc @ np.random.default_rng(11).normal(size=(d, C)))
    marks = []
    for n in (200, 800, 1000):
        run.sample(1, n, 1, n, save=False)
        marks.append(float(np.exp(run.adapt_state()["logα"].astype(np.float64)).mean()))
    S, status = run.factor()
    run.close()
    assert (status == 0).all() and np.isfinite(S).all()
    assert 0.66 > marks[0] > marks[1] > marks[2] > 0.60 and marks[0] - marks[2] > 0.008, marks


def test_c3_ensemble_known_answer_at_scale(mhx, real):
    """configs[2] as a known answer: 16 384 walkers, 50-dim Gaussian with Sigma_ij = 0.9^|i-j|, initial walkers drawn on the
    device from N(0, I); after 20 000 sweeps of burn-in (the stretch move mixes slowly in 50 dimensions) the walkers of 20 sweeps 200 apart reproduce mean 0, unit variances and
    the neighbour correlation 0.9 (the stretch move leaves the target invariant, src/emcee.jl:70-102)."""
    d, W = 50, 16384
    Sig = cases.sigma_ar1(d, 0.9)
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), spl, 20, seed=12, discard_initial=20000, thinning=200)
    v = chain.value[:, :d, :].astype(np.float64)                        # [20][d][W]
    pooled = v.transpose(1, 0, 2).reshape(d, -1)
    assert np.abs(pooled.mean(axis=1)).max() < 0.03
    assert np.abs(pooled.var(axis=1) - 1.0).max() < 0.05
    nb = [np.corrcoef(pooled[k], pooled[k + 1])[0, 1] for k in range(d - 1)]
    assert abs(np.mean(nb) - 0.9) < 0.01 and np.abs(np.array(nb) - 0.9).max() < 0.03
    assert 0.1 < chain.accepted[1:].mean() < 0.5


def test_c3_rotated_ensemble_known_answer_at_scale(mhx, real):
    """The dense-rotated C3 target as a known answer on the dense-factor kernel (matrix-core form, variant 10): Sigma = Q (0.9^|i-j|) Q^T has no structure the
    kernel could exploit; rotated back by Q^T the walkers must show the AR(1) model again -- mean 0, unit variances, neighbour
    correlation 0.9 -- after the same burn-in as the banded test (the stretch move is affine-invariant: identical mixing)."""
    d, W = 50, 16384
    Q, _ = np.linalg.qr(np.random.default_rng(50).normal(size=(d, d)))
    Sig = Q @ cases.sigma_ar1(d, 0.9) @ Q.T
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    chain = mhx.sample(mhx.DensityModel(mhx.CorrGaussian(Sig)), spl, 20, seed=12, discard_initial=20000, thinning=200)
    assert chain.stats["kernel_variant"] in (9, 10)
    v = chain.value[:, :d, :].astype(np.float64)                        # [20][d][W]
    pooled = Q.T @ v.transpose(1, 0, 2).reshape(d, -1)                  # back in the AR(1) coordinates
    assert np.abs(pooled.mean(axis=1)).max() < 0.03
    assert np.abs(pooled.var(axis=1) - 1.0).max() < 0.05
    nb = [np.corrcoef(pooled[k], pooled[k + 1])[0, 1] for k in range(d - 1)]
    assert abs(np.mean(nb) - 0.9) < 0.01 and np.abs(np.array(nb) - 0.9).max() < 0.03
    # log-density of the kept walkers: E[lp] = -d/2 - d/2 log(2 pi) - 1/2 log det Sigma, det Sigma = (1 - 0.81)^(d-1)
    want = -0.5 * d * (1.0 + np.log(2 * np.pi)) - 0.5 * (d - 1) * np.log(1.0 - 0.81)
    assert abs(chain.value[:, d, :].astype(np.float64).mean() - want) < 0.1
    assert 0.1 < chain.accepted[1:].mean() < 0.5


@pytest.mark.parametrize("sampler", ["rwmh", "mala"])
def test_matrix_core_kernels_known_answer(mhx, sampler, real):
    """The dense Gaussian target on the matrix cores as a known answer: 16 384 chains on the 100-dim AR(1) Gaussian with
    Sigma_ij = 0.6^|i-j| from x0 = 0; the states of 12 draws far apart reproduce mean 0, unit variances and the neighbour
    correlation 0.6 -- RWMH (mhx_rwmh_mfma_kernels.h) and MALA (mhx_mala_mfma_kernels.h, src/MALA.jl:54-93)."""
    d, C = 100, 16384
    Sig = cases.sigma_ar1(d, 0.6)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    if sampler == "rwmh":
        s = 2.38 / d ** 0.5
        spl, disc, thin = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), 30000, 3000
    else:
        spl, disc, thin = mhx.MALA(0.3 / d ** (1 / 3)), 3000, 300
    chain = mhx.sample(model, spl, 12, C, seed=21, discard_initial=disc, thinning=thin, initial_params=np.zeros(d))
    assert chain.stats["kernel_variant"] == 8 and chain.stats["reduce_lanes"] == 4
    v = chain.value[:, :d, :].astype(np.float64)
    pooled = v.transpose(1, 0, 2).reshape(d, -1)
    assert np.abs(pooled.mean(axis=1)).max() < 0.03
    assert np.abs(pooled.var(axis=1) - 1.0).max() < 0.04
    nb = np.array([np.corrcoef(pooled[k], pooled[k + 1])[0, 1] for k in range(d - 1)])
    assert abs(nb.mean() - 0.6) < 0.01 and np.abs(nb - 0.6).max() < 0.03


def test_c2_and_c5_full_size_with_ziggurat_normals(mhx, oracle):
    """The two configurations bench.py runs with MHX_FLAG_ZIGGURAT, at their full sizes (fp64): C2 -- 65 536 chains, the pre-built
    kernel, three 64-chain subsets against the oracle and two half-size shards against the whole; C5 -- one GPU's shard of the
    1000-dim funnel with global ids, running moments of a continuation."""
    old_m, old_o = mhx.get_default_dtype(), oracle.get_dtype()
    mhx.set_default_dtype("f64")
    oracle.set_dtype("f64")
    try:
        d, C, N = 100, 65536, 24
        s = float(np.float32(2.38 / d ** 0.5))
        model = mhx.DensityModel(mhx.IsoGaussian(d))
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
        chain = mhx.sample(model, spl, N, C, seed=0xC0FFEE, normal_gen="ziggurat")
        assert chain.stats["kernel_variant"] == 3 and chain.stats["reduce_lanes"] == 2 and chain.stats["normal_gen"] == 1
        for first in (0, 31337, C - 64):
            ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=2), oracle.Proposal(oracle.PROP_ISO, s, normal_gen=1), oracle.schedule(N),
                              0xC0FFEE, first, 64)
            _same(chain.value[:, :, first:first + 64], ref["samples"], "C2 samples of chains %d.." % first)
            _same(chain.accepted[:, first:first + 64], ref["accepted"], "accepted")
        half = mhx.sample(model, spl, 6, C // 2, seed=0xC0FFEE, first_chain=C // 2, reduce_lanes=2, normal_gen="ziggurat")
        _same(half.value, chain.value[:6, :, C // 2:], "upper shard")
        assert chain.stats["accepted"] == int(chain.accepted[1:].sum())
        del chain, half
        d, Cs, N = 1000, 32768, 4
        s = float(np.float32(2.38 / d ** 0.5))
        first = 3 * Cs
        run = mhx.Run(mhx.DensityModel(mhx.Funnel(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), nchains=Cs, seed=5,
                      first_chain=first, normal_gen="ziggurat")
        run.init(None)
        run.sample(N)
        val, acc = run.samples()
        st = run.stats()
        assert st["kernel_variant"] == 3 and st["reduce_lanes"] == 64 and st["normal_gen"] == 1
        ot = oracle.Target(oracle.TARGET_FUNNEL, d).with_lanes(64)
        for off in (0, Cs - 16):
            ref = oracle.rwmh(ot, oracle.Proposal(oracle.PROP_ISO, s, normal_gen=1), oracle.schedule(N + 10), 5, first + off, 16)
            _same(val[:, :, off:off + 16], ref["samples"][:N], "C5 samples of chains %d.." % (first + off))
            if off == 0:
                keep = ref
        run.sample(2, 5, 5, 0, save="moments")                      # the pre-built moments twin: states after 5 and 10 more transitions
        x, lp, cnt = run.state()
        _same(x[:, :16], keep["samples"][N + 9, :d, :], "state after the moments continuation")
        assert np.isfinite(run.diagnostics()["mean"]).all()
    finally:
        mhx.set_default_dtype(old_m)
        oracle.set_dtype(old_o)
