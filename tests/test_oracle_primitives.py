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


# ---- the fp64 build (the reference's Float64 arithmetic): its own polynomials, 52-bit uniforms, 2 Philox blocks per 4 normals
@pytest.fixture
def oracle64(oracle):
    old = oracle.get_dtype()
    oracle.set_dtype("f64")
    yield oracle
    oracle.set_dtype(old)


def _ulp64(got, want):
    import mpmath as mp
    ex = max(mp.floor(mp.log(abs(want), 2)), -1022) if want != 0 else -1074
    return float(abs(mp.mpf(float(got)) - want) / mp.mpf(2) ** (ex - 52))


def test_f64_log_exp_accuracy(oracle64):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1, 1500), np.exp(rng.uniform(-700, 700, 1000)), 1 + rng.uniform(-1e-3, 1e-3, 500),
                        rng.uniform(0.5, 2, 1500), [5e-324, 2.2250738585072014e-308]])
    assert max(_ulp64(g, mp.log(mp.mpf(float(v)))) for g, v in zip(oracle64.log(x), x)) < 1.0
    assert oracle64.log([0.0])[0] == -np.inf and np.isnan(oracle64.log([-1.0])[0]) and oracle64.log([np.inf])[0] == np.inf
    x = np.concatenate([rng.uniform(-745, 709.7, 2500), rng.uniform(-1, 1, 1000), rng.uniform(-30, 30, 1000)])
    assert max(_ulp64(g, mp.exp(mp.mpf(float(v)))) for g, v in zip(oracle64.exp(x), x)) < 1.0
    assert oracle64.exp([-800.0])[0] == 0.0 and oracle64.exp([800.0])[0] == np.inf and oracle64.exp([0.0])[0] == 1.0


def test_f64_sincos_accuracy(oracle64):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    worst = 0.0
    ks = [int(k) for k in np.random.default_rng(2).integers(0, 2 ** 64, 3000, dtype=np.uint64)] + [0, 1, 2 ** 61, 2 ** 62, 2 ** 63, 2 ** 64 - 1]
    for k in ks:
        s, c = oracle64.sincos2pi_u64(k)
        kk = (k + 2 ** 61) % 2 ** 64                     # the spec's angle: quadrant + 52-bit residual
        q, t = kk >> 62, ((kk & (2 ** 62 - 1)) - 2 ** 61) >> 10
        ang = 2 * mp.pi * (mp.mpf(q) / 4 + mp.mpf(t) * mp.mpf(2) ** -54)
        ws, wc = mp.sin(ang), mp.cos(ang)
        if abs(ws) > mp.mpf("1e-30"):
            worst = max(worst, _ulp64(s, ws))
        if abs(wc) > mp.mpf("1e-30"):
            worst = max(worst, _ulp64(c, wc))
        assert abs(s * s + c * c - 1.0) < 4e-16
    assert worst < 0.75
    assert oracle64.sincos2pi_u64(0) == (0.0, 1.0)


def test_f64_uniforms_and_normals(oracle64):
    # (k + 1/2) 2^-52 and k 2^-52 with k = hi:lo >> 12, exactly
    for hi, lo in ((0, 0), (0xffffffff, 0xffffffff), (0x12345678, 0x9abcdef0), (1, 0xfff), (0, 0x1000)):
        k = ((hi << 32) | lo) >> 12
        assert oracle64.u01_open(hi, lo) == (2 * k + 1) / 2.0 ** 53 and oracle64.u01_half(hi, lo) == k / 2.0 ** 52
    assert 0.0 < oracle64.u01_open(0, 0) and oracle64.u01_open(0xffffffff, 0xffffffff) < 1.0
    z = np.concatenate([oracle64.normals(7, c, 1, 0, 1000) for c in range(100)])
    assert z.dtype == np.float64 and abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # normals 4b .. 4b+3 come from Philox blocks 2b and 2b+1 of the stream: a prefix of a longer draw is the shorter draw
    assert np.array_equal(oracle64.normals(7, 3, 9, 0, 7), oracle64.normals(7, 3, 9, 0, 100)[:7])
    lu = np.array([oracle64.accept_logu(3, c, s) for c in range(60) for s in range(1, 41)])
    assert (lu < 0).all() and stats.kstest(-lu, "expon").pvalue > 1e-3
    # the accept block is shared by 2 consecutive steps (words 0:1 and 2:3 of one Philox block)
    w = oracle64.philox([5, 0, 7 >> 1, 1 << 28], [3, 0])
    assert oracle64.accept_logu(3, 5, 7) == oracle64.log([oracle64.u01_open(w[2], w[3])])[0]
    assert oracle64.accept_logu(3, 5, 6) == oracle64.log([oracle64.u01_open(w[0], w[1])])[0]


def test_c1_readme_model_runs_in_fp64(oracle64):
    """BASELINE configs[0] / SURVEY 8(d) C1: the README's 2-parameter Normal(mu, sigma) model, RWMH(MvNormal(zeros(2), I)),
    100 000 steps, one chain, fp64, seed 1234, on the CPU oracle (plumbing; the GPU twin is tests/test_gpu_misc.py)."""
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_normal_data.npy"))[:30]
    t = oracle64.Target(oracle64.TARGET_IID_NORMAL, 2, params=data)
    r = oracle64.rwmh(t, oracle64.Proposal(oracle64.PROP_ISO, 1.0), oracle64.schedule(100000), 1234, 0, 1,
                      init=np.array([[0.0], [1.0]]))
    assert r["samples"].dtype == np.float64
    mu, sig = r["samples"][:, 0, 0], r["samples"][:, 1, 0]
    assert abs(mu.mean() - data.mean()) < 0.1 and abs(sig.mean() - data.std()) < 0.15 and (sig >= 0).all()


# ---- the ziggurat normal generator of the fp64 spec (3.11) -------------------------------------------------------------
def _zig_table(path):
    import re
    txt = open(path).read()
    n = int(re.search(r"#define MHX_ZIG_N (\d+)", txt).group(1))
    body = txt[txt.index("#define MHX_ZIG_TABLE"):]
    body = body[:body.index("}")]                                            # (the fp32 table follows in the same file)
    vals = [float.fromhex(t) for t in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", body)]
    r = float.fromhex(re.search(r"#define MHX_ZIG_R (\S+)", txt).group(1))
    nri = float.fromhex(re.search(r"#define MHX_ZIG_NEG_RINV (\S+)", txt).group(1))
    return n, np.array(vals), r, nri


def test_ziggurat_table_is_equal_area_and_shared():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = os.path.join(root, "advancedmh.jl_amd", "csrc", "mhx_zig_table.h")
    orc = os.path.join(root, "oracle", "mhx_zig_table.h")
    assert open(dev).read() == open(orc).read(), "device and oracle must read the same generated table (tools/gen_zig_table.py)"
    n, x, r, nri = _zig_table(dev)
    assert n == 1024 and x.size == n + 1 and x[1] == r and x[n] == 0.0 and abs(nri + 1.0 / r) < 1e-16
    assert (np.diff(x) < 0).all()
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    f = lambda t: mp.exp(-mp.mpf(t) ** 2 / 2)
    v = mp.mpf(r) * f(r) + mp.sqrt(mp.pi / 2) * mp.erfc(mp.mpf(r) / mp.sqrt(2))
    assert abs(mp.mpf(x[0]) * f(r) / v - 1) < 1e-15                       # base strip: x[0] f(r) = v
    for i in (1, 2, 17, 511, 1000, n - 1):                                  # layer i: x[i] (f(x[i+1]) - f(x[i])) = v
        assert abs(mp.mpf(x[i]) * (f(x[i + 1]) - f(x[i])) / v - 1) < 1e-12, i


def test_ziggurat_normals_are_standard(oracle64):
    O = oracle64
    x = np.concatenate([O.zig_normals(20260928, c, 5, O.STREAM_PROPOSAL, 500000) for c in range(8)])
    n = x.size
    assert abs(x.mean()) < 4.5 / np.sqrt(n) and abs(x.var() - 1) < 4.5 * np.sqrt(2.0 / n)
    assert abs((x ** 3).mean()) < 4.5 * np.sqrt(15.0 / n) and abs((x ** 4).mean() - 3) < 4.5 * np.sqrt(96.0 / n)
    assert stats.kstest(x[::3], "norm").pvalue > 1e-3
    edges = stats.norm.ppf(np.linspace(0, 1, 257))
    cnt, _ = np.histogram(x, edges)
    chi2 = ((cnt - n / 256.0) ** 2 / (n / 256.0)).sum()
    assert chi2 < stats.chi2.ppf(1 - 1e-4, 255), chi2
    import os
    _, _, r, _ = _zig_table(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "mhx_zig_table.h"))
    for t in (1.0, 2.0, 3.0, r, 4.5):                                        # body, wedges and the tail beyond r
        e = n * 2 * stats.norm.sf(t)
        assert abs((np.abs(x) > t).sum() - e) < 4.5 * np.sqrt(e) + 1, (t, (np.abs(x) > t).sum(), e)
    assert (x == 0).sum() <= 1 and np.isfinite(x).all()
    # symmetric: flipping bit 11 of the low word flips the sign and nothing else -- the streams of +x and -x are the same
    assert abs((x > 0).mean() - 0.5) < 4.5 * 0.5 / np.sqrt(n)


def test_ziggurat_stream_layout(oracle64):
    """Normal n of a step depends on Philox block n >> 1 of its stream (words 0,1 / 2,3) and, on rejection, on blocks
    (n << 8 | t) of stream | 4 only: a prefix of a longer draw is the shorter draw, and rwmh with normal_gen = 1 uses exactly these."""
    O = oracle64
    a = O.zig_normals(3, 9, 2, O.STREAM_PROPOSAL, 40)
    b = O.zig_normals(3, 9, 2, O.STREAM_PROPOSAL, 1000)
    assert np.array_equal(a, b[:40])
    d, s = 6, 0.5
    r = O.rwmh(O.iso_gauss(d), O.Proposal(O.PROP_ISO, s, normal_gen=1), O.schedule(2), 3, 9, 1)
    x0 = s * O.zig_normals(3, 9, 0, O.STREAM_INIT, d)                       # the initial draw fma(s, n, 0) == s * n (s = 1/2: exact)
    assert np.array_equal(r["samples"][0, :d, 0], x0)
    z = O.zig_normals(3, 9, 1, O.STREAM_PROPOSAL, d)
    cand = x0 + s * z                                                       # s = 1/2: the product is exact, one rounding like the fma
    got = r["samples"][1, :d, 0]
    assert np.array_equal(got, cand if r["accepted"][1, 0] else x0)


def _zig32_table(path):
    import re
    txt = open(path).read()
    n = int(re.search(r"#define MHX_ZIG32_N (\d+)", txt).group(1))
    body = txt[txt.index("#define MHX_ZIG32_TABLE"):]
    vals = [float.fromhex(t) for t in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", body)]
    r = float.fromhex(re.search(r"#define MHX_ZIG32_R (\S+?)f\b", txt).group(1))
    nri = float.fromhex(re.search(r"#define MHX_ZIG32_NEG_RINV (\S+?)f\b", txt).group(1))
    return n, np.array(vals), r, nri


def test_fp32_ziggurat_table_normals_and_stream_layout(oracle):
    """round 6: the fp32 form of spec 3.11 -- 256 equal-area layers (floats correctly rounded from the 60-digit recursion), ONE 32-bit
    word per normal: normal n of a step comes from word n & 3 of Philox block n >> 2 of its stream and, on rejection, from blocks
    (n << 8 | t) of stream | 4 only.  The draws are standard normal; rwmh with normal_gen = 1 uses exactly these."""
    import os
    O = oracle
    old = O.get_dtype()
    O.set_dtype("f32")
    try:
        n, x, r, nri = _zig32_table(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "mhx_zig_table.h"))
        assert n == 256 and x.size == n + 1 and x[1] == r and x[n] == 0.0 and (np.diff(x) < 0).all()
        assert all(float(np.float32(v)) == v for v in x) and abs(nri + 1.0 / r) < 1e-7
        mp = pytest.importorskip("mpmath")
        mp.mp.dps = 30
        f = lambda t: mp.exp(-mp.mpf(t) ** 2 / 2)
        v = mp.mpf(r) * f(r) + mp.sqrt(mp.pi / 2) * mp.erfc(mp.mpf(r) / mp.sqrt(2))
        for i in (1, 2, 17, 128, 250, n - 1):                                # equal areas, to the floats' rounding
            assert abs(mp.mpf(x[i]) * (f(x[i + 1]) - f(x[i])) / v - 1) < 2e-5, i
        z = np.concatenate([O.zig_normals(20260929, c, 5, O.STREAM_PROPOSAL, 500000) for c in range(8)]).astype(np.float64)
        m = z.size
        assert z.dtype == np.float64 and abs(z.mean()) < 4.5 / np.sqrt(m) and abs(z.var() - 1) < 4.5 * np.sqrt(2.0 / m)
        assert abs((z ** 3).mean()) < 4.5 * np.sqrt(15.0 / m) and abs((z ** 4).mean() - 3) < 4.5 * np.sqrt(96.0 / m)
        assert stats.kstest(z[::3], "norm").pvalue > 1e-3
        for t in (1.0, 2.0, 3.0, r, 4.2):                                    # body, wedges and the tail beyond r
            e = m * 2 * stats.norm.sf(t)
            assert abs((np.abs(z) > t).sum() - e) < 4.5 * np.sqrt(e) + 1, (t, (np.abs(z) > t).sum(), e)
        assert np.isfinite(z).all() and abs((z > 0).mean() - 0.5) < 4.5 * 0.5 / np.sqrt(m)
        a = O.zig_normals(3, 9, 2, O.STREAM_PROPOSAL, 40)
        b = O.zig_normals(3, 9, 2, O.STREAM_PROPOSAL, 1000)
        assert a.dtype == np.float32 and np.array_equal(a, b[:40])
        # the fast path, restated: word j of block p -> layer, sign, 23-bit uniform
        d, sc = 6, 0.5
        rr = O.rwmh(O.iso_gauss(d), O.Proposal(O.PROP_ISO, sc, normal_gen=1), O.schedule(2), 3, 9, 1)
        x0 = np.float32(sc) * O.zig_normals(3, 9, 0, O.STREAM_INIT, d)
        assert np.array_equal(rr["samples"][0, :d, 0], x0)
        zz = O.zig_normals(3, 9, 1, O.STREAM_PROPOSAL, d)
        cand = x0 + np.float32(sc) * zz
        assert np.array_equal(rr["samples"][1, :d, 0], cand if rr["accepted"][1, 0] else x0)
    finally:
        O.set_dtype(old)
