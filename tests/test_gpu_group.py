"""mhx_group_*: many chains over many GPUs as ONE call from ONE process (README.md:135-148, `sample(model, spl, MCMCThreads(), N,
nchains)`).  N member contexts on device 0 are legal, so on a one-GPU box: a group of 4 == 4 sequential shards == the unsharded
run, bit for bit -- chains carry global ids in their RNG counters (src/mh-core.jl:92-117 per chain)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, what
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s" % (what, len(bad), bad[0])


def _sampler(mhx, kind, d):
    s = float(np.float32(2.38 / d ** 0.5))
    if kind == "rwmh":
        return mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
    if kind == "mala":
        return mhx.MALA(0.05)
    return mhx.RobustAdaptiveMetropolis()


@pytest.mark.parametrize("kind,d,C,members", [("rwmh", 100, 259, 4), ("rwmh", 7, 64, 3), ("mala", 12, 130, 2), ("ram", 6, 70, 4)])
def test_group_is_its_shards_is_the_unsharded_run(mhx, real, kind, d, C, members):
    N, disc, thin = 9, 2, 3
    warm = 4 if kind == "ram" else 0
    model = mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.5)) if kind != "rwmh" else mhx.IsoGaussian(d))
    spl = _sampler(mhx, kind, d)
    init = np.random.default_rng(3).normal(size=(d, C))
    # the unsharded run
    whole = mhx.Run(model, spl, nchains=C, seed=11, first_chain=5)
    whole.init(init)
    want, want_acc = whole.sample_to_host(N, disc, thin, warm)
    # the group: `members` contexts on device 0, one host thread each
    g = mhx.Group([0] * members)
    assert len(set(g.pci_bus_ids())) == 1                     # what tells N members on one GPU from N GPUs
    g.create(model, spl, nchains=C, seed=11, first_chain=5)
    assert sum(r.n for r in g.runs) == C and max(r.n for r in g.runs) - min(r.n for r in g.runs) <= 1
    g.init(init)
    vals, accs = g.sample_to_host(N, disc, thin, warm)
    _same(np.concatenate(vals, axis=2), want, "group tensor vs unsharded run")
    _same(np.concatenate(accs, axis=1), want_acc, "accepted")
    st, ws = g.stats(), whole.stats()
    assert st["transitions"] == ws["transitions"] and st["accepted"] == ws["accepted"] and st["tainted"] == 0
    # the same shards one after the other on the default context
    o = 0
    for i, r in enumerate(g.runs):
        f, n = g.shard(C, i)
        assert (f, n) == (o, r.n)
        one = mhx.Run(model, spl, nchains=n, seed=11, first_chain=5 + f)
        one.init(init[:, o:o + n])
        v, a = one.sample_to_host(N, disc, thin, warm)
        _same(v, vals[i], "shard %d alone" % i)
        o += n
    # the statistics ranks would all-reduce: summed on the host over the members == the unsharded run's, up to summation order
    dg, dw = g.diagnostics(split=True), whole.diagnostics(split=True)
    assert dg["n_chains"] == dw["n_chains"] == 2 * C
    for k in ("sum_m", "sum_m2", "sum_v"):
        np.testing.assert_allclose(dg[k], dw[k], rtol=1e-12 if real == "f64" else 1e-5, atol=1e-12)
    np.testing.assert_allclose(dg["rhat"], dw["rhat"], rtol=1e-9 if real == "f64" else 1e-4)
    g.close()
    whole.close()


def test_group_draws_initial_states_and_continues_across_calls(mhx, oracle, real):
    """init(None): every member draws ITS chains' starts on the device; two consecutive group calls continue the chains -- against the
    oracle with global chain ids"""
    d, C, N = 100, 130, 6
    s = float(np.float32(2.38 / d ** 0.5))
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I))
    g = mhx.Group([0, 0, 0])
    g.create(model, spl, nchains=C, seed=21, first_chain=1000, reduce_lanes=2)
    g.init(None)
    g.sample(N, 0, 1, 0)
    first = np.concatenate([r.samples()[0] for r in g.runs], axis=2)
    g.sample(N, 1, 1, 0)
    second = np.concatenate([r.samples()[0] for r in g.runs], axis=2)
    L = g.stats()["reduce_lanes"]
    ref = oracle.rwmh(oracle.iso_gauss(d, reduce_lanes=L), oracle.Proposal(oracle.PROP_ISO, s), oracle.schedule(2 * N), 21, 1000, C)
    _same(first, ref["samples"][:N], "first call")
    _same(second, ref["samples"][N:], "second call")
    g.close()


def test_group_of_ensembles_and_moments(mhx, real):
    """an Ensemble: one ensemble per member (ids first_chain + i); save = "moments": the R-hat sums without a sample tensor"""
    d, W = 5, 64
    model = mhx.DensityModel(mhx.CorrGaussian(cases.sigma_ar1(d, 0.7)))
    spl = mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
    g = mhx.Group([0, 0])
    g.create(model, spl, seed=9, first_chain=40)
    g.init(None)
    vals, _ = g.sample_to_host(8)
    for i in range(2):
        one = mhx.Run(model, spl, seed=9, first_chain=40 + i)
        one.init(None)
        v, _ = one.sample_to_host(8)
        _same(v, vals[i], "ensemble %d" % i)
    g.close()
    dm, C = 40, 96
    s = float(np.float32(2.38 / dm ** 0.5))
    model = mhx.DensityModel(mhx.Funnel(dm))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(dm), s * s * mhx.I))
    init = np.random.default_rng(8).normal(size=(dm, C))
    g = mhx.Group([0, 0, 0])
    g.create(model, spl, nchains=C, seed=2)
    g.init(init)
    g.sample(10, 5, 5, 0, save="moments")
    whole = mhx.Run(model, spl, nchains=C, seed=2)
    whole.init(init)
    whole.sample(10, 5, 5, 0, save="moments")
    dg, dw = g.diagnostics(), whole.diagnostics()
    assert dg["n_chains"] == C and dg["n_samples"] == 10
    for k in ("sum_m", "sum_m2", "sum_v"):
        np.testing.assert_allclose(dg[k], dw[k], rtol=1e-12 if real == "f64" else 2e-5, atol=1e-10)
    g.close()


def test_a_failing_member_is_named(mhx, real):
    d = 4
    model = mhx.DensityModel(mhx.IsoGaussian(d))
    g = mhx.Group([0, 0])
    g.create(model, mhx.MALA(0.1), nchains=10, seed=1)
    with pytest.raises(mhx.MhxError, match=r"mhx_group_init: member 0 \(device 0\)"):
        g.init(None)                                       # MALA requires initial_params (src/MALA.jl:37)
    with pytest.raises(mhx.MhxError, match="member 0"):
        g.sample(3)                                        # ... and a run that was never initialised cannot sample
    g.close()
    with pytest.raises(mhx.MhxError):
        mhx.Group([0, 99])                                 # no such device: nothing leaks, the error names the member


def test_a_shard_takes_the_kernel_form_of_the_whole_run(mhx, real):
    """where the engine picks the kernel form from the chain count (reduce_lanes = 0), a shard must pick what the
    UNSHARDED run picks -- the README's data-sum model runs a wave per chain (reduction shape 64) up to 2048 chains and a lane per
    chain (shape 1) beyond: 4 members x 1024 chains of a 4096-chain run used to take shape 64 and differ from the whole in the last
    bits.  mhx_group_shard tells the member's context the whole run's count (option TOTAL_CHAINS); a lone process does it itself."""
    data = np.random.default_rng(1234).normal(size=30)
    model = mhx.DensityModel(mhx.IIDNormal(data))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.25 * mhx.I))
    C, N = 4096, 12
    init = np.array([0.0, 1.0])
    whole = mhx.Run(model, spl, nchains=C, seed=7)
    whole.init(init)
    want, _ = whole.sample_to_host(N, 0, 1, 0)
    assert whole.stats()["reduce_lanes"] == 1 and whole.stats()["kernel_variant"] != 11
    g = mhx.Group([0] * 4)
    g.create(model, spl, nchains=C, seed=7)
    g.init(init)
    vals, _ = g.sample_to_host(N, 0, 1, 0)
    assert all(r.stats()["reduce_lanes"] == 1 for r in g.runs)
    _same(np.concatenate(vals, axis=2), want, "4 x 1024 chains of the data-sum model vs the 4096-chain run")
    # a process that holds one shard of a multi-process run says so itself
    ctx = mhx.Context(0, real)
    ctx.set_option("TOTAL_CHAINS", C)
    part = mhx.Run(model, spl, nchains=1024, seed=7, first_chain=1024, ctx=ctx)
    part.init(init)
    v, _ = part.sample_to_host(N, 0, 1, 0)
    _same(v, want[:, :, 1024:2048], "shard 1 of 4 with TOTAL_CHAINS set")
    # ... and without the hint the small run keeps its own (faster) form: a wave per chain
    lone = mhx.Run(model, spl, nchains=1024, seed=7, first_chain=1024)
    lone.init(init)
    lone.sample(2)
    assert lone.stats()["kernel_variant"] == 11 and lone.stats()["reduce_lanes"] == 64
    g.close()
