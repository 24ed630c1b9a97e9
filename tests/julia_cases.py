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


def _rwmh_readme(O):
    """README.md:25-40 as written: DensityModel(density) over 30 data points (the first 30 of tests/golden/c1_normal_data.npy), RWMH(MvNormal(zeros(2), I))"""
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_normal_data.npy"))[:30].astype(np.float64)
    init = np.repeat(np.array([[0.0], [1.0]]), 4, axis=1)
    return O.traced(O.rwmh, O.Target(O.TARGET_IID_NORMAL, 2, params=data), O.Proposal(O.PROP_ISO, 1.0), O.schedule(64), 18, 0, 4, init=init)


def user_gauss_data(d=100):
    """per-dimension mean / standard deviation of the user log-density case: exact binary fractions (the Julia script writes the same)"""
    k = np.arange(d, dtype=np.float64)
    return np.concatenate([(k - 50.0) / 64.0, 0.5 + (k % 8.0) / 8.0])


def _rwmh_user_ziggurat(O):
    """a USER log-density (DensityModel(f), not a catalogue target) at d = 100 with the ziggurat normals -- on the device this is the
    register kernel's ziggurat form (round 4), one lane per chain: reduction shape 1, the plain ascending sum"""
    import user_targets
    d = 100
    ut = user_targets.host_target(O, user_targets.SHIFTED_GAUSS, d, data=user_gauss_data(d))
    return O.traced(O.rwmh, ut, O.Proposal(O.PROP_ISO, 0.25, normal_gen=1), O.schedule(24), 19, 5, 8)


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
    "rwmh_given_start": _rwmh_given_start, "rwmh_iso_ziggurat": _rwmh_iso_ziggurat, "rwmh_readme": _rwmh_readme,
    "rwmh_user_ziggurat": _rwmh_user_ziggurat, "rwmh_drift": _rwmh_drift, "rwmh_static": _rwmh_static, "ram": _ram, "ram_random_start": _ram_random_start, "ram_bounds": _ram_bounds,
    "mala_iso": _mala_iso, "mala_corr": _mala_corr, "emcee_seq": _emcee_seq,
}
