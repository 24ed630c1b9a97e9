"""The few-chain kernel (kernel_variant 11, round 5): a WAVE per chain for the data-sum target of the reference's
own example (README.md:25-40, test/runtests.jl:20-35,76-94: theta = (mu, sigma), sum(logpdf.(Normal(mu, sigma), data))).  The 64
lanes split the likelihood's terms (reduction shape 64: lane l owns terms l, l + 64, ..., xor-butterfly), the draws of 64 steps are
made side by side by the lanes, the records of a batch leave one step per lane.  Bit for bit the oracle at that shape."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, what
    bad = np.argwhere(cases.bits(a) != cases.bits(b))
    assert len(bad) == 0, "%s: %d mismatches, first at %s" % (what, len(bad), bad[0])


@pytest.mark.parametrize("np_,C,N,disc,thin,prop", [(30, 1, 200, 0, 1, "iso"), (8, 3, 70, 5, 3, "diag"), (64, 70, 130, 0, 2, "iso"),
                                                   (100, 5, 65, 1, 1, "diag"), (300, 2, 64, 63, 64, "iso"), (65, 33, 129, 7, 5, "iso")])
@pytest.mark.parametrize("wave_k", [None, 4, 8])
def test_wave_per_chain_kernel_is_the_oracle_at_shape_64(mhx, oracle, real, engine, np_, C, N, disc, thin, prop, wave_k):
    """terms that fill less than a wave, exactly a wave, several per lane; launches that are not multiples of the 64-step batch;
    discard / thinning (a record every 64th step: one lane of a batch writes); ISO and DIAG proposals; the state afterwards and a
    second call that continues the chains; 4 and 8 speculative candidates per round (option WAVE_K; None: the host's choice by the
    previous call's acceptance -- the same chain either way)"""
    if wave_k:
        engine.set("WAVE_K", str(wave_k))
    data = np.load(os.path.join(GOLD, "c1_normal_data.npy"))[:np_]
    model = mhx.DensityModel(mhx.IIDNormal(data))
    if prop == "iso":
        spl, op = mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.04 * mhx.I)), oracle.Proposal(oracle.PROP_ISO, 0.2)
    else:
        sv = np.array([0.25, 0.125])
        spl, op = mhx.RWMH([mhx.Normal(0.0, float(v)) for v in sv]), oracle.Proposal(oracle.PROP_DIAG, vec=sv)
    init = np.stack([np.linspace(-0.5, 0.5, C), np.linspace(0.8, 1.5, C)])
    run = mhx.Run(model, spl, nchains=C, seed=99, first_chain=4)
    run.init(init)
    run.sample(N, disc, thin, 0)
    got, got_acc = run.samples()
    st = run.stats()
    assert st["kernel_variant"] == 11 and st["reduce_lanes"] == 64
    t = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data, reduce_lanes=64)
    nT = disc + (N - 1) * thin
    ref = oracle.rwmh(t, op, oracle.schedule(N, disc, thin), 99, 4, C, init=init)
    _same(got, ref["samples"], "samples")
    _same(got_acc, ref["accepted"], "accepted")
    assert st["transitions"] == nT * C and st["accepted"] == int(ref["accept_counts"].sum())
    x, lp, cnt = run.state()
    _same(x, ref["final_x"], "final x")
    _same(lp, ref["final_lp"], "final lp")
    _same(cnt, ref["accept_counts"], "accept counts")
    # a second call continues the chains (slot 0 = the state the first call left)
    run.sample(10, 3, 1, 0)
    more, _ = run.samples()
    ref2 = oracle.rwmh(t, op, oracle.schedule(10, nT + 3, 1), 99, 4, C, init=init)
    _same(more, ref2["samples"], "continued call")


def test_wave_kernel_support_edge_and_device_drawn_start(mhx, oracle, real):
    """sigma <= 0 is outside the support (lp = -Inf: README.md:30 `insupport`): a start there with a finite candidate accepts (+Inf
    ratio), -Inf against -Inf is NaN and rejects -- the decisions of src/mh-core.jl:104-108; and init(None): the start is a bare
    proposal draw on the device, lp evaluated in the same shape"""
    data = np.load(os.path.join(GOLD, "c1_normal_data.npy"))[:30]
    model = mhx.DensityModel(mhx.IIDNormal(data))
    t = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data, reduce_lanes=64)
    C = 9
    init = np.tile(np.array([[0.0], [-0.2]]), (1, C))
    chain = mhx.sample(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.25 * mhx.I)), 200, C, seed=3, initial_params=init)
    assert chain.stats["kernel_variant"] == 11
    ref = oracle.rwmh(t, oracle.Proposal(oracle.PROP_ISO, 0.5), oracle.schedule(200), 3, 0, C, init=init)
    _same(chain.value, ref["samples"], "samples")
    _same(chain.accepted, ref["accepted"], "accepted")
    assert np.isneginf(chain.value[0, 2, :]).all() and np.isfinite(chain.value[-1, 2, :]).all()
    chain = mhx.sample(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.25 * mhx.I)), 100, 5, seed=8, first_chain=2)
    assert chain.stats["kernel_variant"] == 11
    ref = oracle.rwmh(t, oracle.Proposal(oracle.PROP_ISO, 0.5), oracle.schedule(100), 8, 2, 5)
    _same(chain.value, ref["samples"], "device-drawn start")


def test_wave_kernel_is_the_few_chain_choice_only(mhx, real):
    """many chains keep the lane-per-chain kernels (a wave per chain would queue 64 x the waves); reduce_lanes picks explicitly"""
    data = np.load(os.path.join(GOLD, "c1_normal_data.npy"))[:30]
    model = mhx.DensityModel(mhx.IIDNormal(data))
    spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(2), 0.04 * mhx.I))
    for C, lanes, want in ((2048, 0, 11), (2049, 0, 1), (4096, 64, 11), (16, 1, 1)):
        r = mhx.Run(model, spl, nchains=C, seed=1, reduce_lanes=lanes)
        r.init(np.array([0.0, 1.0]))
        r.sample(3)
        assert r.stats()["kernel_variant"] == want, (C, lanes)
        r.close()
    with pytest.raises(mhx.ArgumentError):
        mhx.Run(model, spl, nchains=4, seed=1, reduce_lanes=8)
