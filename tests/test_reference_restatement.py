"""A SECOND, independent restatement of the reference's step functions -- float64 numpy / scipy, written from
the formulas as the reference states them (library logpdfs, np.linalg.cholesky for the low-rank update), not
from the oracle's algebra -- driven by the oracle's own random draws and checked step by step against the
oracle's float32 traces ("teacher forcing": every step starts from the oracle's recorded state, so one
rounding-flipped decision cannot desynchronise the comparison; decisions whose margin is below 1e-3 are
skipped, there are a handful per thousand).

This does not pin the oracle to the Julia package (which cannot run here -- oracle/ header: "parity
unpinned"), but it does check that the oracle's short-cuts -- Hastings ratio 1/2|z|^2 - 1/2|z + 2L^-1 mu|^2,
static q(x) by forward substitution, MALA's ratio in the whitened draw, the sign-unified rank-1 sweep, the
step-size index -- equal the formulas of src/mh-core.jl:92-117, src/proposal.jl:31-35,58-83,190-192,
src/emcee.jl:39-102, src/MALA.jl:54-93 and src/RobustAdaptiveMetropolis.jl:123-173,239-278."""
import numpy as np
import pytest
from scipy.stats import multivariate_normal

import cases

STREAM_PROPOSAL, STREAM_EMCEE = 0, 3
MARGIN = 1e-3


def _gauss_lp(Sig):
    P = np.linalg.inv(Sig)
    c = -0.5 * (len(Sig) * np.log(2 * np.pi) + np.linalg.slogdet(Sig)[1])
    return lambda x: c - 0.5 * x @ P @ x, lambda x: -P @ x


def _close(a, b, what, tol=2e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.allclose(a, b, rtol=tol, atol=tol), "%s: max |diff| %.3g" % (what, np.abs(a - b).max())


@pytest.mark.parametrize("kind", ["rw_dense", "rw_dense_mean", "rw_diag_mean", "static_dense_mean", "static_iso"])
def test_metropolis_hastings_step(oracle, kind):
    """src/mh-core.jl:92-117: candidate from the proposal, log alpha = lp' - lp + q(x | x') - q(x' | x) with
    q(proposal, t, t_cond) = logpdf(proposal, t - t_cond) (random walk, src/proposal.jl:58-64) or
    logpdf(proposal, t) (static, :74-83), accept iff -randexp < log alpha."""
    d, C, N, seed = 4, 6, 60, 17
    rng = np.random.default_rng(3)
    Sig_t = cases.sigma_ar1(d, 0.6)
    lp_fn, _ = _gauss_lp(Sig_t)
    static = kind.startswith("static")
    mean = rng.normal(size=d) * 0.4 if kind.endswith("mean") else np.zeros(d)
    if "dense" in kind:
        A = rng.normal(size=(d, d)) * 0.4
        Sig_p = A @ A.T + 0.5 * np.eye(d)
        L = np.linalg.cholesky(Sig_p).astype(np.float32).astype(np.float64)
        op = dict(kind=oracle.PROP_DENSE, vec=oracle.pack_lower(L))
    elif "diag" in kind:
        s = (0.4 + rng.random(d)).astype(np.float32).astype(np.float64)
        L = np.diag(s)
        op = dict(kind=oracle.PROP_DIAG, vec=s)
    else:
        s = float(np.float32(1.4))
        L = s * np.eye(d)
        op = dict(kind=oracle.PROP_ISO, scale=s)
    mean32 = mean.astype(np.float32).astype(np.float64)
    prop = multivariate_normal(mean32, L @ L.T)
    init = (rng.normal(size=(d, C)) * 0.7).astype(np.float32)
    ref = oracle.rwmh(oracle.corr_gauss_from_cov(Sig_t),
                      oracle.Proposal(mean=mean32 if kind.endswith("mean") else None, static=static, **op),
                      oracle.schedule(N), seed, 5, C, init=init)
    checked = 0
    for c in range(C):
        for t in range(1, N):
            x = ref["samples"][t - 1, :d, c].astype(np.float64)
            z = oracle.normals(seed, 5 + c, t, STREAM_PROPOSAL, d).astype(np.float64)
            xi = mean32 + L @ z                                   # rand(rng, proposal)
            y = xi if static else x + xi
            if static:
                ratio = prop.logpdf(x) - prop.logpdf(y)
            else:
                ratio = prop.logpdf(x - y) - prop.logpdf(y - x)
            loga = lp_fn(y) - lp_fn(x) + ratio
            logu = oracle.accept_logu(seed, 5 + c, t)
            if abs(logu - loga) < MARGIN:
                continue
            acc = logu < loga
            assert bool(ref["accepted"][t, c]) == acc, (c, t, logu, loga)
            _close(ref["samples"][t, :d, c], y if acc else x, "state")
            _close(ref["samples"][t, d, c], lp_fn(y if acc else x), "lp", 5e-4)
            checked += 1
    assert checked > 0.97 * C * (N - 1)


def test_mala_step(oracle):
    """src/MALA.jl:54-93 with the proposal g -> MvNormal((sigma2/2) g, sigma2 I): candidate = x + rand(prop(grad x));
    ratio = q(prop(grad y), x, y) - q(prop(grad x), y, x), q(p, t, t_cond) = logpdf(p, t - t_cond)."""
    d, C, N, seed, s2 = 3, 5, 50, 9, 0.35
    Sig = cases.sigma_ar1(d, 0.5)
    lp_fn, grad = _gauss_lp(Sig)
    init = (np.random.default_rng(1).normal(size=(d, C))).astype(np.float32)
    ref = oracle.mala(oracle.corr_gauss_from_cov(Sig), s2, oracle.schedule(N), seed, 0, C, init)
    s2f = float(np.float32(s2))
    sig = float(np.sqrt(np.float32(s2)))
    checked = 0
    for c in range(C):
        for t in range(1, N):
            x = ref["samples"][t - 1, :d, c].astype(np.float64)
            z = oracle.normals(seed, c, t, STREAM_PROPOSAL, d).astype(np.float64)
            px = multivariate_normal(0.5 * s2f * grad(x), s2f * np.eye(d))
            y = x + (px.mean + sig * z)
            py = multivariate_normal(0.5 * s2f * grad(y), s2f * np.eye(d))
            loga = lp_fn(y) - lp_fn(x) + py.logpdf(x - y) - px.logpdf(y - x)
            logu = oracle.accept_logu(seed, c, t)
            if abs(logu - loga) < MARGIN:
                continue
            acc = logu < loga
            assert bool(ref["accepted"][t, c]) == acc
            _close(ref["samples"][t, :d, c], y if acc else x, "state")
            checked += 1
    assert checked > 0.97 * C * (N - 1)


@pytest.mark.parametrize("mode", [0, 1])
def test_emcee_sweep(oracle, mode):
    """src/emcee.jl:39-58 (mode 0: the sequential sweep, partner mod1(i + rand(1:W-1), W), already-moved walkers are
    seen in their new position) and the parallel half-split the device runs (mode 1); move: src/emcee.jl:70-102."""
    d, W, N, seed, a = 3, 12, 25, 4, 2.0
    Sig = cases.sigma_ar1(d, 0.7)
    lp_fn, _ = _gauss_lp(Sig)
    init = cases.emcee_init(d, W, 3)
    ref = oracle.emcee(oracle.corr_gauss_from_cov(Sig), a, mode, oracle.schedule(N), seed, 2, W, init)
    key = [seed & 0xFFFFFFFF, seed >> 32]
    checked = 0
    for t in range(1, N):
        old = ref["samples"][t - 1, :d, :].astype(np.float64)       # [d][W]
        new = old.copy()
        ok = True
        halves = [(0, W)] if mode == 0 else [(0, W // 2), (W // 2, W)]
        for lo, hi in halves:
            frozen = new.copy()                                      # mode 1: the other half as it is now
            for i in range(lo, hi):
                w = oracle.philox([i, 2, t, STREAM_EMCEE << 28], key)
                if mode == 0:
                    r = 1 + ((w[0] * (W - 1)) >> 32)                 # rand(1:W-1)
                    j = (i + r) % W                                  # mod1(i + r, W), 0-based
                    xj = new[:, j] if j < i else old[:, j]           # :53
                else:
                    ostart, osize = (W // 2, W - W // 2) if lo == 0 else (0, W // 2)
                    j = ostart + ((w[0] * osize) >> 32)
                    xj = frozen[:, j]
                u = (w[1] >> 8) * 2.0 ** -24                         # rand(rng) in [0, 1)
                z = ((a - 1.0) * u + 1.0) ** 2 / a                   # :81
                xi = new[:, i].copy()
                y = xj + z * (xi - xj)                               # :85
                alpha = (d - 1) * np.log(z) + lp_fn(y) - lp_fn(xi)   # :82,91
                logu = np.log(np.float32(w[2]) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33))   # -randexp
                if abs(logu - alpha) < MARGIN:
                    ok = False                                       # a coin flip: leave this sweep out
                if logu <= alpha:                                    # :93 (non-strict)
                    new[:, i] = y
        if ok:
            _close(ref["samples"][t, :d, :], new, "sweep %d" % t)
            checked += 1
    assert checked > 0.8 * (N - 1)


@pytest.mark.parametrize("bounds", [(0.0, np.inf), (0.7, 1.4)])
def test_ram_warmup_step(oracle, bounds):
    """src/RobustAdaptiveMetropolis.jl:123-173,239-278: x' = S U + x; log alpha = min(lp' - lp, 0); accept iff
    randexp > -log alpha; eta = iteration^-gamma with the iteration BEFORE its increment (1 at the first step);
    dS = sqrt(eta |d alpha|) S U / |U|; S_new = chol(S S' +- dS dS') (update iff sign(d alpha) == 1), kept only if its
    diagonal lies inside the bounds.  One oracle run per prefix length gives the state before and after step k
    (N = k + 1 samples with num_warmup = k + 1: every one of the k transitions is a step_warmup)."""
    d, C, seed, alpha0, gamma = 3, 4, 13, 0.234, 0.6
    Sig = cases.sigma_ar1(d, 0.5) * 2.0
    lp_fn, _ = _gauss_lp(Sig)
    tgt = oracle.corr_gauss_from_cov(Sig)
    init = np.zeros((d, C), dtype=np.float32)
    K = 25
    runs = [oracle.ram(tgt, oracle.schedule(k + 1, 0, 1, k + 1), seed, 0, C, init=init, alpha=alpha0, gamma=gamma,
                       eig_lo=bounds[0], eig_hi=bounds[1]) for k in range(K + 1)]
    checked = 0
    for k in range(1, K + 1):
        before, after = runs[k - 1], runs[k]
        for c in range(C):
            x = before["final_x"][:, c].astype(np.float64)
            S = oracle.unpack_lower(before["S"][c], d).astype(np.float64)
            U = oracle.normals(seed, c, k, STREAM_PROPOSAL, d).astype(np.float64)
            y = S @ U + x
            loga = min(lp_fn(y) - lp_fn(x), 0.0)
            logu = oracle.accept_logu(seed, c, k)
            if abs(logu - loga) < MARGIN:
                continue
            acc = logu < loga                                        # randexp > -log alpha
            da = np.exp(loga) - float(np.float32(alpha0))
            eta = float(k) ** -gamma                                 # state.iteration == k before this step
            dS = np.sqrt(eta * abs(da)) * (S @ U) / np.linalg.norm(U)
            M = S @ S.T + (1.0 if np.sign(da) == 1 else -1.0) * np.outer(dS, dS)
            S_new = np.linalg.cholesky(M)
            if not (bounds[0] == 0 and np.isinf(bounds[1])):
                if not np.all((bounds[0] <= np.diag(S_new)) & (np.diag(S_new) <= bounds[1])):
                    S_new = S
            _close(after["final_x"][:, c], y if acc else x, "x after step %d" % k)
            _close(oracle.unpack_lower(after["S"][c], d), S_new, "S after step %d" % k, 5e-4)
            checked += 1
    assert checked > 0.95 * C * K
