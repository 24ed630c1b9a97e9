"""The cases tests/julia/make_reference_traces.jl runs through the UNMODIFIED AdvancedMH.jl (same seeds, global
chain ids, schedules, models, samplers), here through the fp64 oracle.  tests/test_julia_reference_traces.py compares the
two when the Julia traces are present (tests/golden/julia/).  Every case runs with the oracle's trace sink on: `margin` says how
far the closest accept decision behind each saved sample was from flipping."""
import numpy as np

import cases


def _rwmh_iso(O):
    return O.traced(O.rwmh, O.iso_gauss(5), O.Proposal(O.PROP_ISO, 0.5), O.schedule(32), 11, 3, 8)


def _rwmh_dense_corr(O):
    d = 4
    L = np.linalg.cholesky(0.3 * cases.sigma_ar1(d, 0.5))
    return O.traced(O.rwmh, O.corr_gauss_from_cov(cases.sigma_ar1(d, 0.8)), O.Proposal(O.PROP_DENSE, vec=O.pack_lower(L)), O.schedule(20, 3, 2), 12, 0, 6)


def _rwmh_funnel(O):
    return O.traced(O.rwmh, O.Target(O.TARGET_FUNNEL, 6), O.Proposal(O.PROP_ISO, 0.4), O.schedule(24), 13, 100, 7)


def _rwmh_banana(O):
    return O.traced(O.rwmh, O.Target(O.TARGET_BANANA, 5, params=[0.03]), O.Proposal(O.PROP_DIAG, vec=[2.0, 0.5, 1.0, 1.0, 1.0]), O.schedule(24), 14, 0, 7)


def _rwmh_given_start(O):
    init = np.repeat(np.array([[0.5], [-1.0], [0.25]]), 4, axis=1)
    return O.traced(O.rwmh, O.iso_gauss(3), O.Proposal(O.PROP_ISO, 0.7), O.schedule(40), 9, 0, 4, init=init)


def _rwmh_iso_ziggurat(O):
    return O.traced(O.rwmh, O.iso_gauss(8), O.Proposal(O.PROP_ISO, 0.6, normal_gen=1), O.schedule(48), 15, 7, 64)


_MU = np.array([0.3, -0.2, 0.1, 0.25])


def _rwmh_drift(O):
    """a random walk with a non-zero mean: the Hastings ratio q(x | y) - q(y | x) is not zero (src/proposal.jl:58-64,190-192)"""
    return O.traced(O.rwmh, O.iso_gauss(4), O.Proposal(O.PROP_ISO, 0.6, mean=_MU), O.schedule(40), 16, 0, 6, init=np.zeros((4, 6)))


def _rwmh_static(O):
    """StaticMH: an independence sampler, ratio logpdf(p, x) - logpdf(p, y) (src/proposal.jl:9-11,66-83)"""
    return O.traced(O.rwmh, O.corr_gauss_from_cov(cases.sigma_ar1(4, 0.5)), O.Proposal(O.PROP_ISO, 1.2, mean=_MU, static=True), O.schedule(40), 17, 0, 6,
                    init=np.zeros((4, 6)))


def _ram(O):
    d = 4
    return O.traced(O.ram, O.corr_gauss_from_cov(cases.sigma_ar1(d, 0.7)), O.schedule(24, 0, 1, 16), 31, 2, 6, init=np.zeros((d, 6)))


def _ram_random_start(O):
    return O.traced(O.ram, O.corr_gauss_from_cov(cases.sigma_ar1(4, 0.7)), O.schedule(30, 0, 1, 30), 33, 0, 5)


def _ram_bounds(O):
    Sig = np.array([[10.0, 5.0], [5.0, 10.0]])
    return O.traced(O.ram, O.corr_gauss_from_cov(Sig), O.schedule(40, 0, 1, 40), 32, 0, 5, init=np.zeros((2, 5)), gamma=0.51, eig_lo=0.9, eig_hi=1.1)


def _mala_iso(O):
    return O.traced(O.mala, O.iso_gauss(5), 0.3, O.schedule(32), 41, 1, 6, np.full((5, 6), 0.25))


def _mala_corr(O):
    init = np.repeat(np.array([[1.0], [-0.5], [0.25], [0.0]]), 6, axis=1)
    return O.traced(O.mala, O.corr_gauss_from_cov(cases.sigma_ar1(4, 0.6)), 0.3, O.schedule(32, 2, 3), 42, 0, 6, init)


def _emcee_seq(O):
    d, W = 3, 10
    return O.traced(O.emcee, O.corr_gauss_from_cov(cases.sigma_ar1(d, 0.9)), 2.0, 0, O.schedule(16), 21, 0, W, None, prior=O.Proposal(O.PROP_ISO, 1.0))


JULIA_CASES = {
    "rwmh_iso": _rwmh_iso, "rwmh_dense_corr": _rwmh_dense_corr, "rwmh_funnel": _rwmh_funnel, "rwmh_banana": _rwmh_banana,
    "rwmh_given_start": _rwmh_given_start, "rwmh_iso_ziggurat": _rwmh_iso_ziggurat, "rwmh_drift": _rwmh_drift, "rwmh_static": _rwmh_static, "ram": _ram, "ram_random_start": _ram_random_start, "ram_bounds": _ram_bounds,
    "mala_iso": _mala_iso, "mala_corr": _mala_corr, "emcee_seq": _emcee_seq,
}
