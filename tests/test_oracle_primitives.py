"""CPU: pin the oracle's arithmetic spec -- Philox KATs, transcendental accuracy,
target log-densities vs scipy, rank-1 Cholesky vs numpy, schedule arithmetic."""
import numpy as np
import pytest
from scipy import stats


# Random123 known-answer vectors for Philox4x32-10
KATS = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def py_philox(ctr, key):
    """independent pure-Python Philox4x32-10"""
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return c


@pytest.mark.parametrize("ctr,key,want", KATS)
def test_philox_known_answers(oracle, ctr, key, want):
    assert oracle.philox(ctr, key) == want
    assert py_philox(ctr, key) == want


def test_philox_random_inputs_match_pure_python(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        ctr = [int(v) for v in rng.integers(0, 2 ** 32, 4)]
        key = [int(v) for v in rng.integers(0, 2 ** 32, 2)]
        assert oracle.philox(ctr, key) == py_philox(ctr, key)


def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    return np.abs(got.astype(np.float64) - ref64) / np.spacing(np.abs(ref32))


def test_log_accuracy(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1, 20000), np.exp(rng.uniform(-87, 88, 20000)),
                        np.linspace(0.6, 1.4, 5000)]).astype(np.float32)
    x = x[x > 0]
    assert _ulp_err(oracle.logf(x), np.log(x.astype(np.float64))).max() < 1.0
    assert oracle.logf([0.0])[0] == -np.inf and np.isnan(oracle.logf([-1.0])[0])
    assert oracle.logf([np.inf])[0] == np.inf
    sub = np.float32(1e-41)
    assert abs(oracle.logf([sub])[0] - np.log(np.float64(sub))) < 1e-4


def test_exp_accuracy(oracle):
    x = np.random.default_rng(2).uniform(-87, 88, 30000).astype(np.float32)
    assert _ulp_err(oracle.expf(x), np.exp(x.astype(np.float64))).max() < 1.0
    assert oracle.expf([100.0])[0] == np.inf and oracle.expf([-110.0])[0] == 0.0
    assert oracle.expf([0.0])[0] == 1.0


def test_sincos_accuracy(oracle):
    ks = np.concatenate([np.random.default_rng(3).integers(0, 2 ** 32, 20000),
                         [0, 2 ** 30, 2 ** 31, 3 * 2 ** 30, 2 ** 32 - 1, 2 ** 29, 2 ** 29 - 1]])
    sc = np.array([oracle.sincos2pi_u32(int(k)) for k in ks])
    ang = 2 * np.pi * ks.astype(np.float64) / 2 ** 32
    assert np.abs(sc[:, 0] - np.sin(ang)).max() < 2e-7
    assert np.abs(sc[:, 1] - np.cos(ang)).max() < 2e-7


def test_normals_are_standard(oracle):
    n = np.concatenate([oracle.normals(99, c, 1, oracle.STREAM_PROPOSAL, 1000) for c in range(400)]).astype(np.float64)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01
    assert abs((n ** 3).mean()) < 0.03 and abs((n ** 4).mean() - 3) < 0.08
    assert stats.kstest(n[:50000], "norm").pvalue > 1e-3
    # a prefix of a longer draw is the shorter draw (block structure)
    assert np.array_equal(oracle.normals(5, 7, 3, 0, 10), oracle.normals(5, 7, 3, 0, 37)[:10])


def test_accept_uniform_is_exponential(oracle):
    e = -np.array([oracle.accept_logu(7, c, s) for c in range(200) for s in range(1, 101)])
    assert (e >= 0).all() and abs(e.mean() - 1) < 0.02 and abs(e.var() - 1) < 0.06


def test_targets_against_scipy(oracle):
    rng = np.random.default_rng(4)
    d = 7
    x = rng.normal(size=d).astype(np.float32)
    x64 = x.astype(np.float64)
    assert abs(oracle.iso_gauss(d)(x) - stats.multivariate_normal(np.zeros(d), np.eye(d)).logpdf(x64)) < 1e-4
    A = rng.normal(size=(d, d))
    Sig = A @ A.T + d * np.eye(d)
    assert abs(oracle.corr_gauss_from_cov(Sig)(x) - stats.multivariate_normal(np.zeros(d), Sig).logpdf(x64)) < 1e-4
    data = rng.normal(size=30).astype(np.float32)
    t = oracle.Target(oracle.TARGET_IID_NORMAL, 2, params=data)
    assert abs(t([0.3, 1.7]) - stats.norm(0.3, 1.7).logpdf(data.astype(np.float64)).sum()) < 1e-3
    assert t([0.3, -0.1]) == -np.inf and t([0.3, 0.0]) == -np.inf          # theta[2] >= 0 support
    b = 0.03
    tb = oracle.Target(oracle.TARGET_BANANA, d, params=[b])
    u = x64.copy()
    u[1] = x64[1] + b * (x64[0] ** 2 - 100)
    ref = stats.norm(0, 10).logpdf(u[0]) + stats.norm(0, 1).logpdf(u[1:]).sum()
    assert abs(tb(x) - ref) < 1e-4
    tf = oracle.Target(oracle.TARGET_FUNNEL, d)
    ref = stats.norm(0, 3).logpdf(x64[0]) + stats.norm(0, np.exp(x64[0] / 2)).logpdf(x64[1:]).sum()
    assert abs(tf(x) - ref) < 1e-4


@pytest.mark.parametrize("sign", [+1, -1])
def test_rank1_cholesky_against_numpy(oracle, sign):
    rng = np.random.default_rng(5)
    d = 9
    A = rng.normal(size=(d, d))
    L = np.linalg.cholesky(A @ A.T + d * np.eye(d))
    w = rng.normal(size=d) * (0.3 if sign < 0 else 1.0)
    rc, S = oracle.chol_rank1(oracle.pack_lower(L), w, sign)
    assert rc == 0
    want = np.linalg.cholesky(L @ L.T + sign * np.outer(w, w))
    assert np.abs(oracle.unpack_lower(S, d) - want).max() < 2e-5
    if sign < 0:                                    # a downdate that leaves the PD cone is reported
        rc, _ = oracle.chol_rank1(oracle.pack_lower(L), 10 * L[:, 0], -1)
        assert rc != 0


def test_schedule_counts(oracle):
    # N, discard, thinning, warmup -> transitions, adapting transitions  [upstream mcmcsample, restated]
    assert oracle.schedule_counts(oracle.schedule(10, 0, 1, 0)) == (9, 0)
    assert oracle.schedule_counts(oracle.schedule(10000, 25, 4, 0)) == (25 + 9999 * 4, 0)   # test/runtests.jl:129
    assert oracle.schedule_counts(oracle.schedule(1000, 0, 1, 1000)) == (999, 999)          # test/RobustAdaptiveMetropolis.jl:44-55
    assert oracle.schedule_counts(oracle.schedule(10000, 10000, 1, 10000)) == (19999, 10000)  # RAM doctest
    assert oracle.schedule_counts(oracle.schedule(10, 5, 3, 8)) == (5 + 27, 5 + 2 * 3)
