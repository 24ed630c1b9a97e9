"""Seeded parity cases shared by the CPU tests (oracle vs committed golden traces) and the GPU
tests (HIP kernels vs oracle / golden traces).  Each case returns the oracle's result dict."""
import numpy as np

S238 = float(np.float32(0.238))


def f64():
    """is the engine under test running in fp64 (the `real` fixture switched the default)?"""
    import mhx
    return mhx.get_default_dtype() == "f64"


def R():
    """numpy dtype of the engine under test"""
    return np.float64 if f64() else np.float32


def bits(a):
    """the bit pattern of an array (floats as unsigned integers of their width): bit-exact comparisons, NaN-safe"""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])
    return a



def sigma_ar1(d, rho):
    i = np.arange(d)
    return rho ** np.abs(i[:, None] - i[None, :])


def case_rwmh_iso(O):
    return O.rwmh(O.iso_gauss(5), O.Proposal(O.PROP_ISO, 0.5), O.schedule(32), 11, 3, 8)


def case_rwmh_dense_corr(O):
    d = 4
    L = np.linalg.cholesky(0.3 * sigma_ar1(d, 0.5))
    prop = O.Proposal(O.PROP_DENSE, vec=O.pack_lower(L))
    return O.rwmh(O.corr_gauss_from_cov(sigma_ar1(d, 0.8)), prop, O.schedule(20, 3, 2), 12, 0, 6)


def case_rwmh_funnel(O):
    return O.rwmh(O.Target(O.TARGET_FUNNEL, 6), O.Proposal(O.PROP_ISO, 0.4), O.schedule(24), 13, 100, 7)


def case_rwmh_banana(O):
    t = O.Target(O.TARGET_BANANA, 5, params=[0.03])
    return O.rwmh(t, O.Proposal(O.PROP_DIAG, vec=[2.0, 0.5, 1.0, 1.0, 1.0]), O.schedule(24), 14, 0, 7)


def emcee_init(d, W, seed):
    return np.random.default_rng(seed).normal(size=(d, W)).astype(np.float32)


def case_emcee_split(O):
    d, W = 3, 10
    return O.emcee(O.corr_gauss_from_cov(sigma_ar1(d, 0.9)), 2.0, 1, O.schedule(16), 21, 0, W, emcee_init(d, W, 5))


def case_emcee_seq(O):
    d, W = 3, 10
    return O.emcee(O.corr_gauss_from_cov(sigma_ar1(d, 0.9)), 2.0, 0, O.schedule(16), 21, 0, W, emcee_init(d, W, 5))


def case_ram(O):
    d = 4
    r = O.ram(O.corr_gauss_from_cov(sigma_ar1(d, 0.7)), O.schedule(24, 0, 1, 16), 31, 2, 6,
              init=np.zeros((d, 6), dtype=np.float32))
    return r


def case_ram_bounds(O):
    d = 2
    Sig = np.array([[10.0, 5.0], [5.0, 10.0]])
    return O.ram(O.corr_gauss_from_cov(Sig), O.schedule(40, 0, 1, 40), 32, 0, 5, init=np.zeros((d, 5), dtype=np.float32),
                 gamma=0.51, eig_lo=0.9, eig_hi=1.1)


def ram_deferred_setup(d=6, C=4):
    """a start that moves from the first step: updates and downdates, short tail blocks, the warm-up's end inside the run"""
    rng = np.random.default_rng(61)
    return rng.normal(size=(d, C)), np.eye(d) * (2.38 / np.sqrt(d))


def case_ram_deferred(O):
    d, C = 6, 4
    init, S0 = ram_deferred_setup(d, C)
    return O.ram_deferred(O.corr_gauss_from_cov(sigma_ar1(d, 0.7)), O.schedule(40, 0, 1, 33), 33, 1, C, init=init,
                          S_in=np.tile(O.pack_lower(S0), (C, 1)))


TRACE_CASES = {
    "rwmh_iso": case_rwmh_iso,
    "rwmh_dense_corr": case_rwmh_dense_corr,
    "rwmh_funnel": case_rwmh_funnel,
    "rwmh_banana": case_rwmh_banana,
    "emcee_split": case_emcee_split,
    "emcee_seq": case_emcee_seq,
    "ram": case_ram,
    "ram_bounds": case_ram_bounds,
    "ram_deferred": case_ram_deferred,
}


def mfma_fits(d, lanes, nimages, real):
    """Host rule of the matrix-core RWMH kernel (csrc/mhx_api.hip mfma_fits / mfma_stream_fits): default or 4 lanes, 16 <= d, and
    either the operand images fit the 160 KB of LDS of a block with the lane's state (ceil(d/4) reals) in its register budget, or
    they are streamed through an LDS ring (state up to 100 / 64 reals per lane in fp32 / fp64)."""
    ns, nt = (d + 3) // 4, (d + 15) // 16
    rb = 8 if real == "f64" else 4
    if not (lanes in (0, 4) and d >= 16 and nimages >= 1):
        return False
    reals = (2 * (nt - 1) * nt + 4 * ((min(4 * nt, ns) + 3) // 4)) * 64
    if ns <= (44 if real == "f64" else 64) and nimages * reals * rb <= 163840:
        return True
    groups = lambda t: (min(4 * (t + 1), ns) + 3) // 4
    maxg = max(groups(2 * p) + (groups(2 * p + 1) if 2 * p + 1 < nt else 0) for p in range((nt + 1) // 2))
    pf = -(-(maxg * 64 * (4 * rb // 16)) // 256)
    if ns <= (64 if real == "f64" else 100) and 2 * pf * 256 * 16 <= 163840:
        return True
    # single-tile chunks, the chain state re-read from its slab: up to 128 (fp64) / 250 (fp32) reals of candidate per lane
    pf1 = -(-(max(groups(t) for t in range(nt)) * 64 * (4 * rb // 16)) // 256)
    return ns <= (128 if real == "f64" else 250) and 2 * pf1 * 256 * 16 <= 163840
