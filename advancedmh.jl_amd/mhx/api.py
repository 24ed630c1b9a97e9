"""Host mirror of the AdvancedMH.jl API surface for the GPU hot path.

Same names and keyword meaning as the reference so that its tests read the same:

    model   = DensityModel(IsoGaussian(100))                      # src/AdvancedMH.jl:52-54
    spl     = RWMH(MvNormal(zeros(100), 0.238**2 * I))            # src/mh-core.jl:50-51
    chain   = sample(model, spl, 1000, 65536; discard_initial=1000)  # AbstractMCMC.sample (re-export, src/AdvancedMH.jl:30)

Every call goes through the C ABI of libmhx.so (include/mhx.h); nothing here computes a sample on
the CPU.  A Python callable cannot be lowered to the device: DensityModel takes one of the
catalogue log-densities below or HipLogDensity(source) (hiprtc-compiled, inlined per lane).
"""
import ctypes as C
import math

import numpy as np

from . import _lib as L

# ------------------------------------------------------------------------------------------------
# distributions (the subset of Distributions.jl the GPU path accepts)


class _Identity:
    """LinearAlgebra.I; supports `s * I`."""

    def __init__(self, scale=1.0):
        self.scale = float(scale)

    def __rmul__(self, s):
        return _Identity(self.scale * float(s))

    __mul__ = __rmul__


I = _Identity()


def zeros(d):
    return np.zeros(int(d))


class Normal:
    def __init__(self, mu=0.0, sigma=1.0):
        self.mu, self.sigma = float(mu), float(sigma)

    def rand(self, rng):
        return self.mu + self.sigma * rng.standard_normal()


class InverseGamma:
    def __init__(self, shape, scale):
        self.shape, self.scale = float(shape), float(scale)

    def rand(self, rng):
        return self.scale / rng.gamma(self.shape)


class MvNormal:
    """MvNormal(mean, cov): cov may be `s*I`, a vector of variances (diagonal) or a dense PD matrix."""

    def __init__(self, mean, cov=I):
        self.mean = np.asarray(mean, dtype=np.float64)
        self.dim = int(self.mean.size)
        if isinstance(cov, _Identity):
            self.kind, self.scale, self.vec = L.PROP_ISO, math.sqrt(cov.scale), None
        else:
            cov = np.asarray(cov, dtype=np.float64)
            if cov.ndim == 1:
                if cov.size != self.dim:
                    raise L.ArgumentError(L.MHX_EINVAL, "MvNormal: mean and covariance dimensions differ")
                self.kind, self.scale, self.vec = L.PROP_DIAG, 1.0, np.sqrt(cov)
            else:
                if cov.shape != (self.dim, self.dim):
                    raise L.ArgumentError(L.MHX_EINVAL, "MvNormal: mean and covariance dimensions differ")
                try:
                    chol = np.linalg.cholesky(cov)
                except np.linalg.LinAlgError as e:
                    raise L.PosDefException(L.MHX_ENOTPD, "MvNormal: covariance is not positive definite") from e
                self.kind, self.scale, self.vec = L.PROP_DENSE, 1.0, pack_lower(chol)

    def rand(self, rng):
        z = rng.standard_normal(self.dim)
        if self.kind == L.PROP_ISO:
            return self.mean + self.scale * z
        if self.kind == L.PROP_DIAG:
            return self.mean + self.vec * z
        return self.mean + unpack_lower(self.vec, self.dim) @ z


def pack_lower(M):
    M = np.asarray(M)
    return np.concatenate([M[i, :i + 1] for i in range(M.shape[0])]).astype(np.float64)


def unpack_lower(p, d):
    M = np.zeros((d, d), dtype=np.asarray(p).dtype)
    o = 0
    for i in range(d):
        M[i, :i + 1] = p[o:o + i + 1]
        o += i + 1
    return M


def _as_mvnormal(dist):
    """RWMH accepts MvNormal, a vector of univariate Normals (src/proposal.jl:26-28) or an Int."""
    if isinstance(dist, (int, np.integer)):
        return MvNormal(zeros(dist), I)                           # src/mh-core.jl:51
    if isinstance(dist, MvNormal):
        return dist
    if isinstance(dist, Normal):                                      # scalar parameter: RandomWalkProposal(Normal(0, 1))
        return MvNormal([dist.mu], np.array([dist.sigma ** 2]))
    if isinstance(dist, (list, tuple)) and all(isinstance(p, Normal) for p in dist):
        mv = MvNormal([p.mu for p in dist], np.array([p.sigma ** 2 for p in dist]))
        return mv
    raise L.ArgumentError(L.MHX_EINVAL, "the GPU path supports MvNormal / vector-of-Normal random-walk proposals; "
                          "got %r (static, function and NamedTuple proposals stay on the CPU reference)" % (dist,))


# ------------------------------------------------------------------------------------------------
# log-densities that can live on the device


class _TargetSpec:
    kind = None
    dim = 0
    params = None

    def build(self, ctx):
        h = C.c_void_p()
        p = None if self.params is None else ctx.arr(self.params)      # rounded once to the context's dtype
        L.check(L.lib().mhx_target_builtin(ctx.h, self.kind, self.dim, L.rptr(p), 0 if p is None else p.size,
                                           C.byref(h)))
        return h


class IsoGaussian(_TargetSpec):
    """logpdf(MvNormal(zeros(d), I), x)"""
    kind = L.TARGET_ISO_GAUSS

    def __init__(self, d):
        self.dim = int(d)


MAX_BAND = 8        # MHX_EMCEE_MAX_BAND: the widest band the engine's band form is specialised for


def precision_factor(Sigma):
    """A = inv(chol(Sigma)) in float64 (log-density -1/2 |A x|^2 + log det A), with the STRUCTURAL zeros of a banded factor
    restored.  A Markov / autoregressive / banded-precision model (Sigma_ij = rho^|i-j|: A is bidiagonal) comes out of the
    inversion with round-off ~1e-15 where the exact factor is zero; the engine detects an EXACTLY banded factor and skips the
    zeros (same bits: fma(0, y, w) == w), which lets such a model run at its own cost instead of the dense d(d+1)/2.
    The rule is scale-aware and conservative: an OFF-diagonal entry is round-off iff |A_ij| <= 256 eps |A_jj| (relative
    to its COLUMN's scale: at stationarity x_j ~ 1/A_jj, so dropping the entry moves row i of A x -- a unit-variance number -- by
    less than 256 eps; a model whose standard deviations span many decades keeps every genuine entry);
    the diagonal is never touched; and the cleaned factor is used only if it really is banded (bandwidth <= 8, the engine's band
    form) -- otherwise the raw inverse is kept, the reference's MvNormal has no truncation at all."""
    A = np.tril(np.linalg.inv(np.linalg.cholesky(np.asarray(Sigma, dtype=np.float64))))
    d = A.shape[0]
    dg = np.abs(np.diag(A))
    noise = np.abs(A) <= 256.0 * np.finfo(np.float64).eps * dg[None, :]
    noise[np.arange(d), np.arange(d)] = False
    B = np.where(noise, 0.0, A)
    i, j = np.nonzero(B)
    return B if d > 1 and int((i - j).max()) <= min(MAX_BAND, d - 2) else A


class CorrGaussian(_TargetSpec):
    """logpdf(MvNormal(zeros(d), Sigma), x); stored as inv(chol(Sigma)) packed lower (precision_factor)."""
    kind = L.TARGET_CORR_GAUSS

    def __init__(self, Sigma):
        Sigma = np.asarray(Sigma, dtype=np.float64)
        self.dim = Sigma.shape[0]
        try:
            A = precision_factor(Sigma)
        except np.linalg.LinAlgError as e:
            raise L.PosDefException(L.MHX_ENOTPD, "CorrGaussian: Sigma is not positive definite") from e
        self.params = pack_lower(A)


class IIDNormal(_TargetSpec):
    """theta = (mu, sigma):  sigma >= 0 ? sum(logpdf.(Normal(mu, sigma), data)) : -Inf   (README.md:29-31)"""
    kind = L.TARGET_IID_NORMAL
    dim = 2

    def __init__(self, data):
        self.params = np.asarray(data, dtype=np.float64).ravel()


class Banana(_TargetSpec):
    kind = L.TARGET_BANANA

    def __init__(self, d, b=0.03):
        self.dim = int(d)
        self.params = np.array([b], dtype=np.float64)


class Funnel(_TargetSpec):
    kind = L.TARGET_FUNNEL

    def __init__(self, d):
        self.dim = int(d)


class HipLogDensity(_TargetSpec):
    """A user log-density as HIP source:  MHX_LOGDENSITY(x, d, data, ndata) { ...; return lp; }  written against
    `mhx_real` / `MHX_R(literal)` so that it compiles for either dtype."""
    kind = L.TARGET_USER

    def __init__(self, source, dim, data=None):
        self.source, self.dim = source, int(dim)
        self.params = None if data is None else np.asarray(data, dtype=np.float64).ravel()

    def build(self, ctx):
        h = C.c_void_p()
        p = None if self.params is None else ctx.arr(self.params)
        L.check(L.lib().mhx_target_from_hip_source(ctx.h, self.source.encode(), self.dim, L.rptr(p),
                                                   0 if p is None else p.size, C.byref(h)))
        return h


class DensityModel:
    """DensityModel(logdensity) -- src/AdvancedMH.jl:52-54.  `logdensity` is a catalogue log-density, HipLogDensity(source, dim),
    or a Python callable of the parameter vector together with `dim`: the callable is traced once (mhx/trace.py) and lowered
    to the HIP source form, with its reverse-mode gradient (MALA) unless gradient=False."""

    def __init__(self, logdensity, dim=None, gradient=True, names=None):
        self.traced = None
        if names is not None:                   # the closure reads its parameters by name (x.a, x["a"]): test/runtests.jl:184
            names = [str(n) for n in names]
            dim = len(names) if dim is None else dim
        if callable(logdensity) and not isinstance(logdensity, _TargetSpec):
            if dim is None:
                raise L.ArgumentError(L.MHX_EINVAL, "DensityModel(f): pass dim=<number of parameters> so that the callable "
                                      "can be traced (or a catalogue log-density / HipLogDensity(source, dim))")
            from . import trace as _trace
            try:
                self.traced = _trace.trace(logdensity, dim, gradient=gradient, names=names)
            except _trace.TraceError as e:
                raise L.ArgumentError(L.MHX_EINVAL, "DensityModel(f): the callable cannot be traced: %s" % e) from e
            logdensity = HipLogDensity(self.traced.source, dim, data=self.traced.data)      # (the rows of its sum_over loops)
        if not isinstance(logdensity, _TargetSpec):
            raise L.ArgumentError(L.MHX_EINVAL, "DensityModel: unsupported log-density %r" % (logdensity,))
        self.logdensity = logdensity
        self.param_names = names
        self._handles = {}

    @property
    def dim(self):
        return self.logdensity.dim

    def handle(self, ctx):
        if ctx not in self._handles:
            self._handles[ctx] = self.logdensity.build(ctx)
        return self._handles[ctx]


def _is_logdensity_problem(obj):
    """an object that implements the LogDensityProblems interface: dimension() and logdensity(theta)"""
    return callable(getattr(obj, "dimension", None)) and callable(getattr(obj, "logdensity", None))


class LogDensityModel(DensityModel):
    """AbstractMCMC.LogDensityModel(problem) -- the LogDensityProblems form of a model (src/AdvancedMH.jl:56,76-77; README.md:75-90),
    and the ONLY form the reference's RobustAdaptiveMetropolis takes (src/RobustAdaptiveMetropolis.jl:175-181,
    test/RobustAdaptiveMetropolis.jl:1-9,30-56).  `problem` has  dimension() -> d  and  logdensity(theta) -> lp  (the methods
    LogDensityProblems.dimension / .logdensity of the Julia object): the dimension comes from the problem, the log-density is traced
    through problem.logdensity exactly as a closure's is.  A catalogue log-density is its own problem.  `sample(problem, sampler, ...)`
    wraps by itself, as AbstractMCMC does."""

    def __init__(self, problem, gradient=True):
        if isinstance(problem, _TargetSpec):
            super().__init__(problem)
        elif _is_logdensity_problem(problem):
            super().__init__(lambda theta: problem.logdensity(theta), dim=int(problem.dimension()), gradient=gradient)
        else:
            raise L.ArgumentError(L.MHX_EINVAL, "LogDensityModel: %r does not implement dimension() / logdensity(theta)" % (problem,))
        self.problem = problem


def logdensity(model, x, ctx=None, dtype=None):
    """logdensity(model, params) for one point (d,) or a batch (d, n) -- src/AdvancedMH.jl:74."""
    if isinstance(x, Transition):
        return x.lp                                              # cached, src/AdvancedMH.jl:75
    ctx = ctx or L.Context.default(dtype=dtype)
    x = ctx.arr(x)
    single = x.ndim == 1
    xb = x.reshape(model.dim, -1)
    if xb.shape[0] != model.dim:
        raise L.ArgumentError(L.MHX_EINVAL, "logdensity: x has the wrong dimension")
    xb = np.ascontiguousarray(xb)
    lp = np.empty(xb.shape[1], dtype=ctx.real)
    L.check(L.lib().mhx_target_eval(ctx.h, model.handle(ctx), L.rptr(xb), xb.shape[1], L.rptr(lp)))
    return float(lp[0]) if single else lp


class Transition:
    """Transition(params, lp, accepted) -- src/AdvancedMH.jl:61-65."""

    def __init__(self, params, lp, accepted):
        self.params, self.lp, self.accepted = params, lp, accepted


# ------------------------------------------------------------------------------------------------
# samplers


class RandomWalkProposal:
    """RandomWalkProposal{issymmetric}(dist) -- src/proposal.jl:13-21."""

    def __init__(self, proposal, issymmetric=False):
        self.proposal = _as_mvnormal(proposal)
        self.issymmetric = issymmetric
        # a non-zero mean makes the walk drift; its Hastings ratio (src/proposal.jl:58-64,190-192) is then
        # evaluated on the device (generic kernel).  Declaring such a proposal symmetric would skip it.
        if issymmetric and np.any(self.proposal.mean != 0):
            raise L.ArgumentError(L.MHX_EINVAL, "a random-walk proposal with a non-zero mean is not symmetric")


def SymmetricRandomWalkProposal(proposal):
    return RandomWalkProposal(proposal, True)


class StaticProposal:
    """StaticProposal(dist) -- src/proposal.jl:9-11: every candidate is a fresh draw from `dist`, whatever the
    current state (independence sampler); the acceptance ratio carries logpdf(dist, x) - logpdf(dist, y)
    (src/proposal.jl:66-83).  `dist`: Normal / list of Normals / MvNormal."""

    issymmetric = False

    def __init__(self, proposal):
        self.proposal = _as_mvnormal(proposal)


class MetropolisHastings:
    """MetropolisHastings(proposal) -- src/mh-core.jl:44-46.  `proposal`: a RandomWalkProposal / StaticProposal, or -- the
    reference's NamedTuple of proposals, test/runtests.jl:136-160, :187 -- a dict {name: proposal} of scalar Normal proposals
    of ONE kind (all random-walk or all static); its keys name the parameters (src/AdvancedMH.jl:96-104)."""

    def __init__(self, proposal):
        self.param_names = None
        if isinstance(proposal, dict):
            if not proposal:
                raise L.ArgumentError(L.MHX_EINVAL, "MetropolisHastings: empty proposal")
            kinds = {type(p) for p in proposal.values()}
            if kinds not in ({RandomWalkProposal}, {StaticProposal}):
                raise L.ArgumentError(L.MHX_EINVAL, "a NamedTuple of proposals is lowered when every entry is a RandomWalkProposal or "
                                      "every entry is a StaticProposal (mixed / function proposals stay on the CPU reference)")
            parts = list(proposal.values())
            if any(q.proposal.dim != 1 for q in parts):
                raise L.ArgumentError(L.MHX_EINVAL, "a NamedTuple of proposals takes one scalar Normal per name")
            mv = MvNormal([float(q.proposal.mean[0]) for q in parts],
                          np.array([float(q.proposal.vec[0]) ** 2 if q.proposal.kind == L.PROP_DIAG else q.proposal.scale ** 2 for q in parts]))
            self.param_names = [str(k) for k in proposal]
            if kinds == {StaticProposal}:
                proposal = StaticProposal(mv)
            else:
                proposal = RandomWalkProposal(mv, all(q.issymmetric for q in parts))
        if not isinstance(proposal, (RandomWalkProposal, StaticProposal)):
            raise L.ArgumentError(L.MHX_EINVAL, "the GPU path implements RandomWalkProposal and StaticProposal "
                                  "over (Mv)Normal distributions only")
        self.proposal = proposal


def StaticMH(d):
    """StaticMH(d) -- src/mh-core.jl:48."""
    return MetropolisHastings(StaticProposal(d))


def RWMH(d):
    """RWMH(d) / RWMH(d::Int) -- src/mh-core.jl:50-51."""
    return MetropolisHastings(RandomWalkProposal(d))


class StretchProposal:
    """StretchProposal(prior, a = 2.0) -- src/emcee.jl:63-68."""

    def __init__(self, proposal, stretch_length=2.0):
        self.proposal, self.stretch_length = proposal, float(stretch_length)

    def rand_initial(self, rng):
        p = self.proposal
        if isinstance(p, (list, tuple)):
            return np.array([q.rand(rng) for q in p])
        return np.asarray(p.rand(rng))


class Ensemble:
    """Ensemble(n_walkers, proposal) -- src/emcee.jl:1-4."""

    def __init__(self, n_walkers, proposal):
        if not isinstance(proposal, StretchProposal):
            raise L.ArgumentError(L.MHX_EINVAL, "Ensemble: only StretchProposal is implemented (as in the reference)")
        self.n_walkers, self.proposal = int(n_walkers), proposal


class RobustAdaptiveMetropolis:
    """RobustAdaptiveMetropolis(; α=0.234, γ=0.6, S=nothing, eigenvalue_lower_bound=0, eigenvalue_upper_bound=Inf)
    -- src/RobustAdaptiveMetropolis.jl:75-87."""

    def __init__(self, α=0.234, γ=0.6, S=None, eigenvalue_lower_bound=0.0, eigenvalue_upper_bound=float("inf"),
                 alpha=None, gamma=None, deferred_factor=False):
        """deferred_factor=True: MHX_FLAG_RAM_DEFERRED (dim <= 256) -- up to 8 rank-one updates stay pending as O(dim) triples and
        are folded into S in one pass: the same chain in exact arithmetic, its own rounding (include/mhx.h)."""
        self.deferred_factor = bool(deferred_factor)
        self.α = float(alpha if alpha is not None else α)
        self.γ = float(gamma if gamma is not None else γ)
        self.S = None if S is None else np.asarray(S, dtype=np.float64)
        self.eigenvalue_lower_bound = float(eigenvalue_lower_bound)
        self.eigenvalue_upper_bound = float(eigenvalue_upper_bound)


class MALA:
    """MALA(sigma2), or the reference's own form `MALA(g -> MvNormal((sigma2 / 2) .* g, sigma2 * I))` (src/MALA.jl:1-11,
    test/runtests.jl:291,352) as a Python callable of the gradient: the callable is probed once the dimension is known and
    must be that Langevin proposal -- mean (sigma2 / 2) g, covariance sigma2 I; any other gradient -> Distribution closure
    stays on the CPU reference."""

    def __init__(self, sigma2):
        self.closure = sigma2 if callable(sigma2) else None
        self.sigma2 = None if callable(sigma2) else float(sigma2)

    def resolve(self, d):
        """sigma2 of the Langevin proposal the closure describes (probed at g = 0, g = 1 and a ramp)."""
        if self.closure is None:
            return self.sigma2
        bad = L.ArgumentError(L.MHX_EINVAL, "MALA on the GPU path takes sigma2 or the Langevin closure g -> MvNormal((sigma2/2) g, sigma2 I); "
                              "this closure is not of that form and cannot be lowered")
        try:
            p0, p1, p2 = (_as_mvnormal(self.closure(g)) for g in (np.zeros(d), np.ones(d), np.arange(1.0, d + 1.0)))
        except L.ArgumentError:
            raise bad from None
        for q in (p0, p1, p2):
            iso = q.kind == L.PROP_ISO or (q.kind == L.PROP_DIAG and np.all(q.vec == q.vec[0]))
            if q.dim != d or not iso:
                raise bad
        s2 = (p0.scale if p0.kind == L.PROP_ISO else float(p0.vec[0])) ** 2
        same = all(abs((q.scale if q.kind == L.PROP_ISO else float(q.vec[0])) ** 2 - s2) <= 1e-12 * s2 for q in (p1, p2))
        if not (same and np.all(p0.mean == 0) and np.allclose(p1.mean, 0.5 * s2, rtol=1e-12, atol=0)
                and np.allclose(p2.mean, 0.5 * s2 * np.arange(1.0, d + 1.0), rtol=1e-12, atol=0)):
            raise bad
        return float(s2)


# ------------------------------------------------------------------------------------------------
# chains container (the part of MCMCChains.Chains the reference tests touch)


class Chains:
    """value[iteration, parameter, chain]; the last parameter is the internal `lp`
    (ext/AdvancedMHMCMCChainsExt.jl:36, :96-118)."""

    def __init__(self, value, names, start, thin, accepted=None, stats=None, state=None):
        self.value = value
        self.names = list(names)
        self.start, self.thin = int(start), int(thin)
        self.accepted = accepted
        self.stats = stats or {}
        self.state = state
        self.internals = ["lp"]

    def range(self):
        n = self.value.shape[0]
        return range(self.start, self.start + self.thin * n, self.thin)

    def __getitem__(self, name):
        if isinstance(name, str):
            return self.value[:, self.names.index(name), :]
        return self.value[name]

    def __len__(self):
        return self.value.shape[0]

    @property
    def nchains(self):
        return self.value.shape[2]

    def mean(self, name):
        return float(np.mean(self[name], dtype=np.float64))

    def params(self):
        return [n for n in self.names if n not in self.internals]

    def summarystats(self, max_lag=0):
        """The table MCMCChains prints for the reference's chains (README.md:59-63): mean, std, ess_bulk, ess_tail
        and (split) rhat per parameter, computed on the device from the run's sample buffer (mhx_run_diagnostics,
        mhx_run_ess_bulk_tail); internals (lp) are left out as MCMCChains does."""
        if self.state is None:
            raise L.ArgumentError(L.MHX_EINVAL, "summarystats needs the live run (chain.state)")
        idx = [i for i, n in enumerate(self.names) if n not in self.internals]
        dg = self.state.diagnostics(max_lag=0, split=True)
        et = self.state.ess_bulk_tail(params=idx, max_lag=max_lag, split=True)
        full = self.state.diagnostics(max_lag=0, split=False)
        std = np.sqrt(np.maximum(full.get("var_plus", full["W"]), 0.0))
        return dict(parameters=[self.names[i] for i in idx], mean=full["mean"][idx], std=std[idx],
                    ess_bulk=et["ess_bulk"], ess_tail=et["ess_tail"],
                    rhat=dg["rhat"][idx] if "rhat" in dg else np.full(len(idx), np.nan))

    def __repr__(self):
        head = "Chains MCMC chain (%dx%dx%d Array{%s, 3}), iterations %d:%d:%d" % (
            self.value.shape[0], self.value.shape[1], self.value.shape[2],
            "Float64" if self.value.dtype == np.float64 else "Float32", self.start, self.thin, self.range()[-1])
        try:
            st = self.summarystats()
        except Exception:
            return head
        lines = [head, "  parameters        mean       std   ess_bulk   ess_tail      rhat"]
        for i, n in enumerate(st["parameters"][:20]):
            lines.append("  %-12s %9.4f %9.4f %10.1f %10.1f %9.4f" % (n, st["mean"][i], st["std"][i], st["ess_bulk"][i],
                                                                   st["ess_tail"][i], st["rhat"][i]))
        if len(st["parameters"]) > 20:
            lines.append("  ... %d more" % (len(st["parameters"]) - 20))
        return "\n".join(lines)


# ------------------------------------------------------------------------------------------------
# a live run (what AbstractMCMC's `state` is for the reference)


class Run:
    def __init__(self, model, sampler, nchains=1, seed=0, first_chain=0, ctx=None, flags=0, reduce_lanes=0, dtype=None,
                 normal_gen=None):
        """normal_gen: None / "box-muller" (default) or "ziggurat" (MHX_FLAG_ZIGGURAT: RWMH on the cooperative or the register
        kernel of an fp64 context, user log-densities included -- a different, cheaper stream of standard normals; reported in
        stats()["normal_gen"])."""
        if normal_gen not in (None, "box-muller", "ziggurat"):
            raise L.ArgumentError(L.MHX_EINVAL, "normal_gen must be None, 'box-muller' or 'ziggurat'")
        if normal_gen == "ziggurat":
            flags |= L.FLAG_ZIGGURAT
        self.ctx = ctx or L.Context.default(dtype=dtype)
        self.real = self.ctx.real
        f32 = self.ctx.arr                                  # (the name of round 1: now "an array of the context's reals")
        self.model, self.sampler = model, sampler
        self.h = C.c_void_p()
        lib = L.lib()
        d = model.dim
        self._keep = []
        if isinstance(sampler, MetropolisHastings):
            mv = sampler.proposal.proposal
            if mv.dim != d:
                raise L.ArgumentError(L.MHX_EINVAL, "proposal dimension %d != model dimension %d" % (mv.dim, d))
            vec = None if mv.vec is None else f32(mv.vec)
            mean = f32(mv.mean) if np.any(mv.mean != 0) else None
            self._keep += [vec, mean]
            if isinstance(sampler.proposal, StaticProposal):
                flags |= L.MHX_FLAG_STATIC_PROPOSAL
            cfg = L.RwmhCfg(d, nchains, seed, first_chain, mv.kind, mv.scale, L.fptr(vec), flags, L.fptr(mean), reduce_lanes)
            L.check(lib.mhx_rwmh_create(self.ctx.h, model.handle(self.ctx), C.byref(cfg), C.byref(self.h)))
            self.n = nchains
            self.kind = "rwmh"
        elif isinstance(sampler, Ensemble):
            # the distribution StretchProposal wraps is the prior of the initial walkers (src/emcee.jl:29-34): a (Mv)Normal
            # or a vector of Normals is drawn on the device, anything else by the host in init()
            self._prior = None
            try:
                self._prior = _as_mvnormal(sampler.proposal.proposal)
            except L.ArgumentError:
                pass
            ik, isc, ivec, imean = -1, 1.0, None, None
            if self._prior is not None and self._prior.dim == d:
                mvp = self._prior
                ik, isc = mvp.kind, mvp.scale
                ivec = None if mvp.vec is None else f32(mvp.vec)
                imean = f32(mvp.mean) if np.any(mvp.mean != 0) else None
                self._keep += [ivec, imean]
            else:
                self._prior = None
            # `sample(model, Ensemble(W, ..), MCMCThreads(), N, nchains)` runs nchains ENSEMBLES (README.md:135-148): here all of
            # them in one run, ids first_chain .. first_chain + nchains - 1, walkers side by side in the chain axis
            self.n_ensembles = max(1, int(nchains))
            cfg = L.EmceeCfg(d, sampler.n_walkers, seed, first_chain, sampler.proposal.stretch_length, flags, reduce_lanes,
                             ik, isc, L.rptr(ivec), L.rptr(imean), self.n_ensembles)
            L.check(lib.mhx_emcee_create(self.ctx.h, model.handle(self.ctx), C.byref(cfg), C.byref(self.h)))
            self.n = sampler.n_walkers * self.n_ensembles
            self.kind = "emcee"
        elif isinstance(sampler, MALA):
            cfg = L.MalaCfg(d, nchains, seed, first_chain, sampler.resolve(d), flags, reduce_lanes)
            L.check(lib.mhx_mala_create(self.ctx.h, model.handle(self.ctx), C.byref(cfg), C.byref(self.h)))
            self.n = nchains
            self.kind = "mala"
        elif isinstance(sampler, RobustAdaptiveMetropolis):
            if sampler.S is not None and sampler.S.shape != (d, d) and sampler.S.shape != (nchains, d, d):
                # src/RobustAdaptiveMetropolis.jl:202-204
                raise L.ArgumentError(L.MHX_EINVAL, "The provided `S` has the wrong dimensionality.")
            cfg = L.RamCfg(d, nchains, seed, first_chain, sampler.α, sampler.γ, sampler.eigenvalue_lower_bound,
                           sampler.eigenvalue_upper_bound, flags | (L.FLAG_RAM_DEFERRED if sampler.deferred_factor else 0))
            L.check(lib.mhx_ram_create(self.ctx.h, model.handle(self.ctx), C.byref(cfg), C.byref(self.h)))
            self.n = nchains
            self.kind = "ram"
            if sampler.S is not None:
                if sampler.S.ndim == 3:                      # one factor per chain
                    S = np.stack([pack_lower(np.tril(sampler.S[c])) for c in range(nchains)])
                    L.check(lib.mhx_ram_set_factor(self.h, L.rptr(f32(S))))
                else:                                        # the reference's form: one S for the sampler (RAM.jl:198-206)
                    L.check(lib.mhx_ram_set_factor_all(self.h, L.rptr(f32(pack_lower(np.tril(sampler.S))))))
        else:
            raise L.ArgumentError(L.MHX_EINVAL, "unsupported sampler %r" % (sampler,))
        self.dim = d
        self.seed = seed

    # -- initial AbstractMCMC.step
    def init(self, initial_params=None):
        ip = None
        if initial_params is not None:
            ip = np.asarray(initial_params, dtype=self.real)
            if self.kind == "emcee" and ip.ndim == 2 and ip.shape == (self.n, self.dim) and self.n != self.dim:
                ip = ip.T                                       # vector-of-walkers form
            if ip.ndim == 1:
                if ip.size != self.dim:
                    raise L.ArgumentError(L.MHX_EINVAL, "initial_params has the wrong dimension")
                ip = np.repeat(ip.reshape(self.dim, 1), self.n, axis=1)
            if ip.shape != (self.dim, self.n):
                raise L.ArgumentError(L.MHX_EINVAL, "initial_params must be (dim,) or (dim, nchains)")
            ip = self.ctx.arr(ip)
        elif self.kind == "emcee" and self._prior is None:
            # src/emcee.jl:29-34: W draws from the wrapped prior.  A (Mv)Normal prior is drawn on the device
            # (mhx_run_init(run, NULL)); any other Distribution: a one-off host draw (numpy Generator seeded from the run
            # seed), the device takes over from the first sweep.
            rng = np.random.default_rng([self.seed, 0xE3CEE])
            ip = self.ctx.arr(np.stack([self.sampler.proposal.rand_initial(rng) for _ in range(self.n)], axis=1))
        L.check(L.lib().mhx_run_init(self.h, L.fptr(ip)))

    def sample(self, n_samples, discard_initial=0, thinning=1, num_warmup=0, save=True):
        """save: True/1 sample tensor, False/0 nothing, "moments"/2 running moments only (mhx.h)."""
        s = L.Schedule(n_samples, discard_initial, thinning, num_warmup)
        mode = 2 if (isinstance(save, str) and save == "moments") or (save is not True and save == 2) else (1 if save else 0)
        L.check(L.lib().mhx_run_sample(self.h, C.byref(s), mode))
        self._last_n = n_samples if mode == 1 else 0

    def sample_to_host(self, n_samples, discard_initial=0, thinning=1, num_warmup=0, want_accepted=True, out=None,
                       out_accepted=None, slab_samples=0, pinned=None):
        """mhx_run_sample_to_host: the whole schedule with the samples streamed to the host while the chains keep running
        (two device slabs; what `sample` returns is a host tensor, ext/AdvancedMHMCMCChainsExt.jl:12-39).  Returns
        (samples [N][dim+1][n], accepted [N][n] or None).  `pinned`: allocate the results page-locked (mhx_host_alloc); None = only
        where a DMA writes them -- a save-all run of >= 1024 chains comes back accept-compacted (include/mhx.h: mhx_compact_hdr),
        host threads write the tensor, and page-locking it would cost more than the whole return (0.55 s for C2's 13 GB);
        `out` / `out_accepted`: caller-provided arrays (any host memory: registered for the call when a DMA targets them)."""
        s = L.Schedule(n_samples, discard_initial, thinning, num_warmup)
        shape = (n_samples, self.dim + 1, self.n)
        if pinned is None:
            oc = self.ctx.get_option("HOST_COMPACT")
            pinned = not (int(oc) != 0 if oc else (thinning == 1 and self.n >= 1024))

        def result(shp, dt):
            # page-locked when asked for and available; a host that cannot lock that much (ulimit -l, fragmented memory) gets
            # pageable memory -- the C side registers what it can for the call and otherwise copies pageable
            if pinned:
                try:
                    return L.host_array(shp, dt)
                except L.MhxError:
                    pass
            return np.empty(shp, dtype=dt)
        if out is None:
            out = result(shape, self.real)
        elif out.shape != shape or out.dtype != self.real or not out.flags.c_contiguous:
            raise L.ArgumentError(L.MHX_EINVAL, "sample_to_host: out must be a C-contiguous %s array of shape %s" % (np.dtype(self.real), shape))
        acc = out_accepted
        if acc is None and want_accepted:
            acc = result((n_samples, self.n), np.uint8)
        elif acc is not None and (acc.shape != (n_samples, self.n) or acc.dtype != np.uint8 or not acc.flags.c_contiguous):
            raise L.ArgumentError(L.MHX_EINVAL, "sample_to_host: out_accepted must be a C-contiguous uint8 array of shape (N, nchains)")
        L.check(L.lib().mhx_run_sample_to_host(self.h, C.byref(s), L.fptr(out), L.u8ptr(acc), slab_samples))
        self._last_n = n_samples
        return out, acc

    def host_stats(self):
        """mhx_run_host_stats: what the last sample_to_host moved -- tensor / wire bytes, link and expand time, whether the
        accept-compacted path ran and with how many host threads"""
        st = L.HostStats()
        L.check(L.lib().mhx_run_host_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in L.HostStats._fields_}

    def samples(self, want_accepted=True):
        n_saved = C.c_int64()
        L.check(L.lib().mhx_run_device_samples(self.h, None, None, C.byref(n_saved)))
        N = int(n_saved.value)
        out = np.empty((N, self.dim + 1, self.n), dtype=self.real)
        acc = np.empty((N, self.n), dtype=np.uint8) if want_accepted else None
        L.check(L.lib().mhx_run_get_samples(self.h, L.fptr(out), L.u8ptr(acc)))
        return out, acc

    def state(self):
        x = np.empty((self.dim, self.n), dtype=self.real)
        lp = np.empty(self.n, dtype=self.real)
        cnt = np.empty(self.n, dtype=np.uint32)
        L.check(L.lib().mhx_run_get_state(self.h, L.fptr(x), L.fptr(lp), L.u32ptr(cnt)))
        return x, lp, cnt

    def set_params(self, x):
        """AbstractMCMC.setparams!! (src/AdvancedMH.jl:150-157): replaces params, lp is re-evaluated."""
        x = self.ctx.arr(x)
        if x.shape != (self.dim, self.n):
            raise L.ArgumentError(L.MHX_EINVAL, "set_params: x must be (dim, nchains)")
        L.check(L.lib().mhx_run_set_state(self.h, L.fptr(x)))

    def factor(self):
        nS = self.dim * (self.dim + 1) // 2
        S = np.empty((self.n, nS), dtype=self.real)
        st = np.empty(self.n, dtype=np.uint8)
        L.check(L.lib().mhx_ram_get_factor(self.h, L.fptr(S), L.u8ptr(st)))
        return S, st

    def diag_range(self):
        lo = np.empty((self.dim, self.n), dtype=self.real)
        hi = np.empty((self.dim, self.n), dtype=self.real)
        L.check(L.lib().mhx_ram_get_diag_range(self.h, L.fptr(lo), L.fptr(hi)))
        return lo, hi

    def adapt_state(self):
        """The rest of RobustAdaptiveMetropolisState (src/RobustAdaptiveMetropolis.jl:99-114): logα of every chain's latest
        transition (min(lp' - lp, 0): average exp(logα) for the acceptance probability), the step size η of the latest
        warm-up transition, the iteration counter and isaccept."""
        la = np.empty(self.n, dtype=self.real)
        acc = np.empty(self.n, dtype=np.uint8)
        eta, it = C.c_double(), C.c_uint64()
        L.check(L.lib().mhx_ram_get_adapt_state(self.h, L.rptr(la), C.byref(eta), L.u8ptr(acc), C.byref(it)))
        return dict(logα=la, η=eta.value, iteration=int(it.value), isaccept=acc.astype(bool))

    def watch_factors(self, chains):
        """RobustAdaptiveMetropolis: keep `state.S` of these chains after EVERY recorded step of the following sampling calls -- what a
        reference callback that stores state.S records (test/RobustAdaptiveMetropolis.jl:11-28).  [] switches it off."""
        ch = np.ascontiguousarray(chains, dtype=np.int32)
        L.check(L.lib().mhx_ram_watch_factors(self.h, ch.ctypes.data_as(C.POINTER(C.c_int32)), ch.size))

    def watched_factors(self, full=True):
        """[n recorded][n watched] lower-triangular factors of the last sampling call (full=False: packed row-major)"""
        nrec, nw = C.c_int64(), C.c_int32()
        L.check(L.lib().mhx_ram_get_watched_factors(self.h, None, 0, C.byref(nrec), C.byref(nw)))
        tri = self.dim * (self.dim + 1) // 2
        S = np.empty((nrec.value, nw.value, tri), dtype=self.real)
        L.check(L.lib().mhx_ram_get_watched_factors(self.h, L.rptr(S), nrec.value, None, None))
        if not full:
            return S
        out = np.zeros((nrec.value, nw.value, self.dim, self.dim), dtype=self.real)
        il = np.tril_indices(self.dim)
        out[:, :, il[0], il[1]] = S
        return out

    def step_stats(self, n_samples=None):
        """RobustAdaptiveMetropolis: the sampler state after EVERY recorded step of the last sampling call -- what the
        reference's callback reads off `state` (test/RobustAdaptiveMetropolis.jl:11-28): dict(logα [N][nchains], η [N]);
        mean(exp(logα)) is the acceptance rate the adaptation steers to α (RAM.jl:141-147).  isaccept = the accepted tensor."""
        nrec = C.c_int64()
        L.check(L.lib().mhx_ram_get_step_stats(self.h, None, None, 0, C.byref(nrec)))      # the count the engine recorded
        if n_samples is not None and int(n_samples) != nrec.value:
            raise L.ArgumentError(L.MHX_EINVAL, "step_stats: n_samples = %d, the last sampling call recorded %d" % (n_samples, nrec.value))
        n_samples = int(nrec.value)
        la = np.empty((n_samples, self.n), dtype=self.real)
        eta = np.empty(n_samples, dtype=np.float64)
        L.check(L.lib().mhx_ram_get_step_stats(self.h, L.rptr(la), eta.ctypes.data_as(C.POINTER(C.c_double)), n_samples, None))
        return {"logα": la, "η": eta, "logalpha": la, "eta": eta}

    def stats(self):
        st = L.Stats()
        L.check(L.lib().mhx_run_stats(self.h, C.byref(st)))
        return dict(transitions=st.transitions, accepted=st.accepted, kernel_ms=st.kernel_ms, wall_ms=st.wall_ms,
                    kernel_variant=st.kernel_variant, launches=st.launches, reduce_lanes=st.reduce_lanes,
                    dtype="f64" if st.dtype == L.MHX_F64 else "f32", normal_gen=st.normal_gen, factor_band=st.factor_band,
                    tainted=st.tainted)

    def diagnostics(self, max_lag=0, ess_chains=256, split=False):
        """Sums for R-hat / between-chain ESS (all chains) and, if max_lag > 0, the Geyer ESS from the multi-chain
        autocorrelations.  split=True: every chain counts as two half-chains (split R-hat).  See include/mhx.h
        (mhx_run_diagnostics)."""
        d1 = self.dim + 1
        arrs = [np.zeros(d1, dtype=np.float64) for _ in range(4)]
        cfg = L.DiagCfg(max_lag, ess_chains, 1 if split else 0)
        ptrs = [a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs]
        if max_lag <= 0:
            ptrs[3] = None
        L.check(L.lib().mhx_run_diagnostics(self.h, C.byref(cfg), *ptrs))
        n_saved = C.c_int64()
        L.check(L.lib().mhx_run_device_samples(self.h, None, None, C.byref(n_saved)))
        parts = 2 if split else 1
        out = dict(sum_m=arrs[0], sum_m2=arrs[1], sum_v=arrs[2], n_chains=self.n * parts, n_samples=int(n_saved.value) // parts)
        if max_lag > 0:
            out["ess_geyer"] = np.abs(arrs[3])
            out["ess_geyer_truncated"] = arrs[3] < 0
        out.update(combine_diagnostics(out["sum_m"], out["sum_m2"], out["sum_v"], out["n_chains"], out["n_samples"]))
        return out

    def save_state(self):
        """The complete state of the run as bytes (checkpoint; upstream's `state` of step / `initial_state`)."""
        nb = C.c_size_t()
        L.check(L.lib().mhx_run_state_size(self.h, C.byref(nb)))
        buf = np.empty(nb.value, dtype=np.uint8)
        L.check(L.lib().mhx_run_save_state(self.h, buf.ctypes.data_as(C.c_void_p), nb.value))
        return buf.tobytes()

    def load_state(self, blob):
        """Continue a saved run: same sampler, model, dim and chain count; seed, global ids and the RNG step counter come
        from the blob, so the continuation is the uninterrupted run bit for bit."""
        buf = np.frombuffer(blob, dtype=np.uint8)
        L.check(L.lib().mhx_run_load_state(self.h, buf.ctypes.data_as(C.c_void_p), buf.size))

    def ess_bulk_tail(self, params=None, max_lag=0, ess_chains=256, split=True):
        """Rank-normalised bulk ESS and tail ESS (Vehtari et al. 2021; MCMCChains' ess_bulk / ess_tail) of the given
        parameter rows (default: all, lp included) of the last sample buffer, sorted and scored on the device.
        max_lag = 0: half the draws of a (half-)chain.  Negative values: upper bounds (see include/mhx.h)."""
        n_saved = C.c_int64()
        L.check(L.lib().mhx_run_device_samples(self.h, None, None, C.byref(n_saved)))
        N = int(n_saved.value) // (2 if split else 1)
        if max_lag <= 0:
            max_lag = max(2, N // 2)
        idx = np.arange(self.dim + 1, dtype=np.int32) if params is None else np.ascontiguousarray(params, dtype=np.int32)
        bulk, tail = np.zeros(len(idx)), np.zeros(len(idx))
        cfg = L.DiagCfg(max_lag, ess_chains, 1 if split else 0)
        dp = C.POINTER(C.c_double)
        L.check(L.lib().mhx_run_ess_bulk_tail(self.h, C.byref(cfg), idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx),
                                              bulk.ctypes.data_as(dp), tail.ctypes.data_as(dp)))
        return dict(params=idx, ess_bulk=np.abs(bulk), ess_tail=np.abs(tail), bulk_truncated=bulk < 0, tail_truncated=tail < 0)

    def close(self):
        if self.h:
            L.lib().mhx_run_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def combine_diagnostics(sum_m, sum_m2, sum_v, n_chains, n_samples):
    """R-hat and between-chain ESS from the (all-reduced) per-parameter sums; see include/mhx.h."""
    Cn, N = float(n_chains), float(n_samples)
    mean = sum_m / Cn
    W = sum_v / Cn
    out = dict(mean=mean, W=W)
    if n_chains > 1:
        Vm = np.maximum((sum_m2 - sum_m * sum_m / Cn) / (Cn - 1.0), 0.0)
        varp = (N - 1.0) / N * W + Vm
        with np.errstate(divide="ignore", invalid="ignore"):
            out["rhat"] = np.sqrt(varp / W)
            out["ess_between"] = Cn * varp / Vm
        out["var_plus"] = varp
    return out


class _ParallelTag:
    """MCMCSerial() / MCMCThreads() / MCMCDistributed() of AbstractMCMC (re-exported, src/AdvancedMH.jl:30) and the
    MCMCHIP() of the Julia glue: accepted as the third positional argument of `sample` for drop-in call sites
    (README.md:141-147, test/runtests.jl:99-108).  On this engine every form runs all chains together on the GPU."""


class MCMCSerial(_ParallelTag):
    pass


class MCMCThreads(_ParallelTag):
    pass


class MCMCDistributed(_ParallelTag):
    pass


class MCMCHIP(_ParallelTag):
    pass


def sample(model, sampler, N, nchains=1, *more, initial_params=None, discard_initial=None, thinning=1, num_warmup=0,
           param_names=None, chain_type=Chains, seed=0, first_chain=0, callback=None, ctx=None, flags=0,
           reduce_lanes=0, progress=False, dtype=None, normal_gen=None, allow_tainted=False):
    """sample(model, sampler, N[, nchains]; kwargs...) -- AbstractMCMC.sample as re-exported by the
    reference (src/AdvancedMH.jl:30).  All chains advance together on the GPU (what
    `sample(model, spl, MCMCThreads(), N, nchains)` does with one task per chain, README.md:141-147).

    dtype: "f64" (default: the reference's Float64) or "f32"; see mhx.set_default_dtype.
    discard_initial defaults to num_warmup [upstream]; `callback(run, i)` is called after each saved
    sample (the reference signature callback(rng, model, sampler, sample, state, i) carries objects
    that live on the device here -- the Run gives access to them).

    allow_tainted: a run of a context that carries a timing probe / fault-injection option of the TOOLS build (mhx_stats.tainted)
    may hold invalid chains; `sample` refuses to wrap it into a container unless this is set (tests of the hooks themselves)."""
    if isinstance(N, _ParallelTag):                 # sample(model, spl, MCMCThreads(), N, nchains)
        if not more:
            raise L.ArgumentError(L.MHX_EINVAL, "sample(model, sampler, parallel, N, nchains): nchains is missing")
        N, nchains = nchains, more[0]
    elif more:
        raise L.ArgumentError(L.MHX_EINVAL, "sample: too many positional arguments")
    if discard_initial is None:
        discard_initial = num_warmup
    if not isinstance(model, DensityModel) and (_is_logdensity_problem(model) or isinstance(model, _TargetSpec)):
        model = LogDensityModel(model)              # sample(problem, sampler, N): AbstractMCMC wraps what implements the interface
    run = Run(model, sampler, nchains=nchains, seed=seed, first_chain=first_chain, ctx=ctx, flags=flags,
              reduce_lanes=reduce_lanes, dtype=dtype, normal_gen=normal_gen)
    run.init(initial_params)
    if callback is None:
        # one call: the samples stream to the host while the chains advance (mhx_run_sample_to_host)
        value, acc = run.sample_to_host(N, discard_initial, thinning, num_warmup)
    else:
        # one saved sample per call, so that the callback can look at every state
        chunks, accs = [], []
        dfw = min(num_warmup, discard_initial)
        kfw = num_warmup - dfw
        run.sample(1, discard_initial, 1, dfw)
        v, a = run.samples()
        chunks.append(v), accs.append(a)
        callback(run, 1)
        for i in range(2, N + 1):
            warm = thinning if i <= kfw else 0
            # `thinning` transitions, the last one saved: discard = thinning - 1 ... expressed as
            # N=1, discard_initial=thinning (sample 1 = state after `thinning` transitions)
            run.sample(1, thinning, 1, warm)
            v, a = run.samples()
            chunks.append(v), accs.append(a)
            callback(run, i)
        value, acc = np.concatenate(chunks, axis=0), np.concatenate(accs, axis=0)
    if run.stats()["tainted"] and not allow_tainted:
        raise L.MhxError(L.MHX_ESTATE, "sample: the run's context is tainted (a probe / fault-injection option of the tools build was "
                         "set on it): its chains may be invalid; pass allow_tainted=True to look at them anyway")
    d = model.dim
    if param_names is None:                                         # a NamedTuple of proposals / named parameters carry their names
        param_names = getattr(sampler, "param_names", None) or getattr(model, "param_names", None)
    if param_names is None:
        names = ["param_%d" % (i + 1) for i in range(d)]           # src/AdvancedMH.jl:91
    else:
        names = [str(p) for p in param_names]
        if len(names) != d:
            raise L.ArgumentError(L.MHX_EINVAL, "param_names has the wrong length")
    names = names + ["lp"]
    if chain_type is Chains:
        return Chains(value, names, discard_initial + 1, thinning, accepted=acc, stats=run.stats(), state=run)
    if chain_type is np.ndarray:
        return value
    if chain_type is StructArray:
        return StructArray(value, names)
    if chain_type in (Transition, dict):
        # Vector{Transition} (the reference's default container, src/mh-core.jl:76-117) / Vector{NamedTuple}
        # (ext/AdvancedMHStructArraysExt.jl, src/AdvancedMH.jl:96-125: the parameters by name, then lp): one list per chain,
        # the bare list for a single chain
        out = bundle_samples(value, acc, names, chain_type)
        return out[0] if value.shape[2] == 1 else out
    raise L.ArgumentError(L.MHX_EINVAL, "chain_type must be Chains, numpy.ndarray, StructArray, Transition or dict")


class StructArray:
    """chain_type=StructArray (ext/AdvancedMHStructArraysExt.jl; test/runtests.jl:63-73 reads `chain.μ`): one array per
    parameter and `lp`, [iteration] for a single chain, [iteration, chain] otherwise -- views of the sample tensor."""

    def __init__(self, value, names):
        self._names = list(names)
        for k, n in enumerate(self._names):
            col = value[:, k, :]
            self.__dict__[n] = col[:, 0] if value.shape[2] == 1 else col

    def __getitem__(self, name):
        return self.__dict__[name]

    def keys(self):
        return tuple(self._names)

    def __len__(self):
        return len(self.__dict__[self._names[0]])


def bundle_samples(value, accepted, names, chain_type):
    """Host reshaping of the sample tensor value[iteration, parameter (+ lp), chain] into per-chain lists of Transition
    (params, lp, accepted) or of dicts {name: value, ..., "lp": lp} -- what the reference's bundle_samples methods build
    (src/AdvancedMH.jl:96-125)."""
    N, d1, C = value.shape
    out = []
    for c in range(C):
        if chain_type is Transition:
            out.append([Transition(value[i, :d1 - 1, c].copy(), value[i, d1 - 1, c], bool(accepted[i, c]) if accepted is not None else None)
                        for i in range(N)])
        else:
            out.append([dict(zip(names, value[i, :, c].tolist())) for i in range(N)])
    return out
