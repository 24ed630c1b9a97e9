"""The reduction shape L (`reduce_lanes`: how many lanes share one chain's log-density sum) is part of the arithmetic spec: a GPU
parity test hands the oracle the shape the kernel chose and then demands bit equality.  That pins kernel == oracle(L); what pins
oracle(L) to the REFERENCE ORDER -- L = 1, the plain ascending sum of src/mh-core.jl:103-108's `logdensity(model, candidate)` -- is
this file, on the CPU, for every shape the kernels use (2, 4, 8, 16, 32, 64) and every catalogue target:

  * the log-density of oracle(L) differs from oracle(1) by rounding only: |lp_L - lp_1| <= BOUND(d, lp), the a-priori bound of
    re-associating a sum of d + O(1) non-negative terms (2 d eps relative to the sum of the terms' magnitudes; the observed
    difference is ~sqrt(d) eps);
  * a chain of oracle(L) makes the SAME accept decisions as the chain of oracle(1) wherever the decision's margin
    |log u - log alpha| exceeds twice the MEASURED envelope of that difference (64 ulp of |lp|: the first test asserts it), and holds bit-identical states up to its first within-margin decision (after which
    the two are different realisations of the same law) -- RWMH, the stretch move and MALA."""
import numpy as np
import pytest

import cases

SHAPES = (2, 4, 8, 16, 32, 64)
EPS = {"f32": float(np.finfo(np.float32).eps), "f64": float(np.finfo(np.float64).eps)}


@pytest.fixture(params=["f64", "f32"])
def width(request, oracle):
    old = oracle.get_dtype()
    oracle.set_dtype(request.param)
    yield request.param
    oracle.set_dtype(old)


def _targets(O):
    """(name, builder(L) -> target, dimension, a draw from roughly the target's own scale)"""
    rng = np.random.default_rng(5)
    d = 100
    S_ar = cases.sigma_ar1(50, 0.9)
    Q, _ = np.linalg.qr(np.random.default_rng(50).normal(size=(50, 50)))
    S_rot = Q @ S_ar @ Q.T

    def funnel_draw(n, dd=1000):
        x = rng.normal(size=(dd, n))
        x[0] *= 3.0
        x[1:] *= np.exp(0.5 * x[0])
        return x

    def banana_draw(n, dd=1000):
        x = rng.normal(size=(dd, n))
        x[0] *= 10.0
        x[1] -= 0.03 * (x[0] ** 2 - 100.0)
        return x
    return [
        ("iso100", lambda L: O.iso_gauss(d, reduce_lanes=L), d, lambda n: rng.normal(size=(d, n))),
        ("iso1000", lambda L: O.iso_gauss(1000, reduce_lanes=L), 1000, lambda n: rng.normal(size=(1000, n))),
        ("funnel1000", lambda L: O.Target(O.TARGET_FUNNEL, 1000, reduce_lanes=L), 1000, funnel_draw),
        ("banana1000", lambda L: O.Target(O.TARGET_BANANA, 1000, params=[0.03], reduce_lanes=L), 1000, banana_draw),
        ("ar1_50", lambda L: O.corr_gauss_from_cov(S_ar, reduce_lanes=L), 50, lambda n: np.linalg.cholesky(S_ar) @ rng.normal(size=(50, n))),
        ("rotated50", lambda L: O.corr_gauss_from_cov(S_rot, reduce_lanes=L), 50, lambda n: np.linalg.cholesky(S_rot) @ rng.normal(size=(50, n))),
    ]


def envelope(lp, width):
    """what the first test measures as the worst |lp_L - lp_1| over every target and shape, with room: 64 ulp of |lp|"""
    return 64.0 * EPS[width] * max(abs(lp), 1.0)


def bound(d, lp, width):
    """re-association of a sum of d + O(1) terms whose magnitudes add up to about |lp| + d (the quadratic form and its constant)"""
    return 2.0 * (d + 8) * EPS[width] * (abs(lp) + d)


@pytest.mark.parametrize("L", SHAPES)
def test_log_density_of_every_shape_is_the_reference_sum_up_to_rounding(oracle, width, L):
    worst = 0.0
    for name, mk, d, draw in _targets(oracle):
        if L > d:
            continue
        t1, tL = mk(1), mk(L)
        x = np.ascontiguousarray(draw(64), dtype=oracle.real())
        for c in range(x.shape[1]):
            a, b = float(t1(x[:, c])), float(tL(x[:, c]))
            assert np.isfinite(a) and np.isfinite(b), (name, L)
            assert abs(a - b) <= bound(d, a, width), "%s L=%d: lp %r vs %r (bound %g)" % (name, L, b, a, bound(d, a, width))
            worst = max(worst, abs(a - b) / (EPS[width] * max(abs(a), 1.0)))
    assert worst < 64.0             # observed: a few ulp of lp; the a-priori bound above is far looser


def _compare(ref, other, margin, ensemble=False):
    """`ref`: the traced oracle(1) run (margin per saved sample); `other`: the oracle(L) run.  Chains are compared up to their first
    within-margin decision: accept flags equal, states BIT-identical (the proposals do not depend on the target's summation order),
    log-densities within the bound.  Returns the fraction of (sample, chain) cells compared."""
    N, d1, C = ref["samples"].shape
    close = ~(ref["margin"] > margin)
    horizon = np.full(C, N)
    for c in range(C):
        hit = np.flatnonzero(close[:, c])
        if hit.size:
            horizon[c] = hit[0]
    if ensemble:
        horizon[:] = horizon.min()
    n = 0
    for c in range(C):
        h = horizon[c]
        if h == 0:
            continue
        assert np.array_equal(ref["accepted"][:h, c], other["accepted"][:h, c]), "chain %d: a decision with margin > %g differs" % (c, margin)
        assert np.array_equal(cases.bits(ref["samples"][:h, :d1 - 1, c]), cases.bits(other["samples"][:h, :d1 - 1, c])), "chain %d states" % c
        assert np.all(np.abs(ref["samples"][:h, d1 - 1, c].astype(np.float64) - other["samples"][:h, d1 - 1, c].astype(np.float64)) <= margin), "chain %d lp" % c
        n += int(h)
    return n / float(N * C)


@pytest.mark.parametrize("L", SHAPES)
def test_rwmh_chains_of_every_shape_decide_like_the_reference_order(oracle, width, L):
    O = oracle
    N, C = 40, 24
    for name, mk, d, draw in _targets(O):
        if L > d:
            continue
        s = float(np.float32(2.38 / d ** 0.5)) * {"funnel1000": 0.3, "ar1_50": 0.12, "rotated50": 0.12}.get(name, 1.0)
        init = np.ascontiguousarray(draw(C), dtype=O.real())
        prop = O.Proposal(O.PROP_ISO, s)
        ref = O.traced(O.rwmh, mk(1), prop, O.schedule(N), 7, 3, C, init=init)
        oth = O.rwmh(mk(L), prop, O.schedule(N), 7, 3, C, init=init)
        lp_scale = float(np.abs(ref["samples"][:, d, :]).max())
        frac = _compare(ref, oth, 2.0 * envelope(lp_scale, width))
        assert frac >= (0.9 if width == "f64" else 0.25), "%s L=%d: only %.2f of the cells lay before a doubtful decision" % (name, L, frac)
        assert 0.02 < ref["accepted"][1:].mean() < 0.98, name       # (decisions of both kinds were compared)


@pytest.mark.parametrize("L", SHAPES)
def test_stretch_move_and_mala_of_every_shape_decide_like_the_reference_order(oracle, width, L):
    O = oracle
    d, W, N = 50, 64, 12
    if L > d:
        pytest.skip("more lanes than dimensions")
    Sig = cases.sigma_ar1(d, 0.9)
    init = cases.emcee_init(d, W, 5)
    ref = O.traced(O.emcee, O.corr_gauss_from_cov(Sig, reduce_lanes=1), 2.0, 1, O.schedule(N), 21, 3, W, init)
    oth = O.emcee(O.corr_gauss_from_cov(Sig, reduce_lanes=L), 2.0, 1, O.schedule(N), 21, 3, W, init)
    lp_scale = float(np.abs(ref["samples"][:, d, :]).max())
    m = 2.0 * envelope(lp_scale, width)
    # walkers interact: everything after the first doubtful move of ANY walker is a different realisation
    frac = _compare(ref, oth, m, ensemble=True)
    assert frac >= (0.5 if width == "f64" else 0.05), frac
    # MALA: the gradient and the two proposal norms are summed in the same shape; its states depend on the gradient, so they are
    # compared to rounding, not bit for bit
    C, dm = 16, 100
    x0 = np.ascontiguousarray(np.random.default_rng(2).normal(size=(dm, C)), dtype=O.real())
    r1 = O.traced(O.mala, O.iso_gauss(dm, reduce_lanes=1), 0.02, O.schedule(20), 9, 0, C, x0)
    rL = O.mala(O.iso_gauss(dm, reduce_lanes=L), 0.02, O.schedule(20), 9, 0, C, x0)
    mm = 8.0 * envelope(float(np.abs(r1["samples"][:, dm, :]).max()), width)
    close = ~(r1["margin"] > mm)
    for c in range(C):
        hit = np.flatnonzero(close[:, c])
        h = hit[0] if hit.size else 20
        assert np.array_equal(r1["accepted"][:h, c], rL["accepted"][:h, c])
        np.testing.assert_allclose(r1["samples"][:h, :, c], rL["samples"][:h, :, c], rtol=1e3 * EPS[width], atol=1e3 * EPS[width] * dm)
