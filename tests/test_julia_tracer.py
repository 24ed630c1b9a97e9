"""DensityModel(f) for a Julia closure: the README density (README.md:25-40) and the NIG density of
test/emcee.jl:5-14 must reach MCMCHIP() unchanged.  The Julia tracer (advancedmh.jl_amd/julia/MHXTrace.jl) cannot be executed here;
what IS executed: its algorithm (tests/julia_tracer_model.py, a transliteration) on the closures WITH their branches, and the
Python tracer (mhx.trace) on the `where` twins -- both must give the committed fixtures character for character.  Same text => same
hiprtc module => the kernels the GPU suite already holds to the oracle (tests/test_gpu_trace.py)."""
import math
import os
import re

import numpy as np
import pytest

import julia_tracer_model as J

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
DATA = np.load(os.path.join(GOLD, "c1_normal_data.npy"))[:30].astype(np.float64)   # the 30 data points of C1 (README.md:26)


# ---- the closures as the reference writes them (branches on parameter values), for the model of the Julia tracer
def readme_julia(theta):
    def insupport(t):
        return t[1] >= 0

    def density(t):
        if not insupport(t):
            return -math.inf
        terms = [J.logpdf_normal(t[0], t[1], float(y)) for y in DATA]     # logpdf.(dist(θ), data): the broadcast first ...
        acc = terms[0]
        for term in terms[1:]:                                        # ... then sum(): left to right
            acc = acc + term
        return acc
    return density(theta)


def nig_julia(theta):
    s, m = theta
    if not (s > 0):
        return -math.inf
    sd = J.sqrt(s)                                                    # (mdist / obsdist: Normal(_, sqrt(s)), one node)
    # `a + b + c + d` is ONE call +(a, b, c, d) in Julia: the four terms are evaluated before the first addition
    a, b = J.logpdf_inverse_gamma(2.0, 3.0, s), J.logpdf_normal(0.0, sd, m)
    c, d = J.logpdf_normal(m, sd, 1.5), J.logpdf_normal(m, sd, 2.0)
    return a + b + c + d


# ---- the `where` twins for mhx.trace
def _twins():
    from mhx import trace as T

    def lpn(mu, sigma, x):
        z = (x - mu) / sigma
        return -(z * z + J.LOG2PI) / 2 - T.log(sigma)

    def readme(theta):
        c = theta[1] >= 0
        terms = [lpn(theta[0], theta[1], float(y)) for y in DATA]
        acc = terms[0]
        for term in terms[1:]:
            acc = acc + term
        return T.where(c, acc, -math.inf)

    def nig(theta):
        s, m = theta
        c = s > 0
        sd = T.sqrt(s)
        cst = 2.0 * math.log(3.0) - math.lgamma(2.0)
        ig = cst - (2.0 + 1) * T.log(s) - 3.0 / s
        b, c2, d = lpn(0.0, sd, m), lpn(m, sd, 1.5), lpn(m, sd, 2.0)
        lp = ig + b + c2 + d
        return T.where(c, lp, -math.inf)
    return readme, nig


def _fixture(name):
    return open(os.path.join(GOLD, name)).read()


@pytest.mark.parametrize("which", ["readme", "nig"])
def test_julia_tracer_algorithm_and_python_tracer_emit_the_committed_source(which):
    from mhx import trace as T
    readme, nig = _twins()
    twin, closure = (readme, readme_julia) if which == "readme" else (nig, nig_julia)
    want = _fixture("traced_%s.hip" % which)
    got_py = T.trace(twin, 2, gradient=False).source
    got_jl, npaths = J.trace_logdensity(closure, 2)
    assert got_py == want, "mhx.trace drifted from tests/golden/traced_%s.hip" % which
    assert got_jl == want, "the Julia tracer's algorithm (julia_tracer_model.py) does not reproduce tests/golden/traced_%s.hip" % which
    assert npaths == 2                                                # the branch taken and the branch not taken


def test_traced_readme_density_is_the_reference_formula():
    """the fixture is the README density: evaluated in float64 it equals sum(logpdf(Normal(mu, sigma), data)) / -Inf"""
    from mhx import trace as T
    import scipy.stats as st
    readme, nig = _twins()
    tr = T.trace(readme, 2, gradient=False)
    for mu, sg in ((0.1, 1.3), (-0.7, 0.4)):
        assert abs(tr.evaluate([mu, sg]) - st.norm(mu, sg).logpdf(DATA).sum()) < 1e-9
    assert tr.evaluate([0.0, -1.0]) == -math.inf
    tn = T.trace(nig, 2, gradient=False)
    s, m = 1.7, 0.9
    want = (st.invgamma(2, scale=3).logpdf(s) + st.norm(0, math.sqrt(s)).logpdf(m) + st.norm(m, math.sqrt(s)).logpdf(1.5)
            + st.norm(m, math.sqrt(s)).logpdf(2.0))
    assert abs(tn.evaluate([s, m]) - want) < 1e-12
    assert tn.evaluate([-1.0, 0.0]) == -math.inf


def test_path_enumeration_of_the_julia_tracer_model():
    """two independent branches: four paths, a decision tree of three selects; a condition met twice decides once; a path that
    throws contributes NaN (the reference would have thrown: the device rejects); a runaway number of paths is an error"""
    def f(t):
        a = t[0] * 2.0 if t[0] > 0 else t[0] * 3.0
        b = t[1] + 1.0 if t[1] > 0 else t[1] - 1.0
        return a + b
    src, n = J.trace_logdensity(f, 2)
    assert n == 4 and src.count("?") == 3

    def g(t):
        lo = 1.0 if t[0] > 0 else 2.0
        hi = 5.0 if t[0] > 0 else 7.0                               # the same condition: no new path
        return t[0] * lo + hi
    src, n = J.trace_logdensity(g, 1)
    assert n == 2 and src.count("?") == 1

    def h(t):
        if t[0] < 0:
            raise ValueError("DomainError")
        return J.sqrt(t[0])
    src, n = J.trace_logdensity(h, 1)
    assert n == 2 and "MHX_NAN" in src

    def many(t):
        acc = t[0]
        for k in range(8):
            acc = acc + (1.0 if t[0] > float(k) else 2.0)
        return acc
    with pytest.raises(J.TraceError):
        J.trace_logdensity(many, 1, max_paths=8)


def test_julia_tracer_file_mirrors_the_model():
    """what can be held WITHOUT Julia: MHXTrace.jl and the model agree on the operation names, the comparison texts, the literal
    formats and the emitted framing lines"""
    jl = open(os.path.join(HERE, "..", "advancedmh.jl_amd", "julia", "MHXTrace.jl")).read()
    for op, txt in J.CMP_TEXT.items():
        assert re.search(r":%s => \"%s\"" % (op, re.escape(txt)), jl), op
    for op, txt in J.ARITH_TEXT.items():
        assert re.search(r":%s => \"%s\"" % (op, re.escape(txt)), jl), op
    for text in ('"// traced by mhx.trace (advancedmh.jl_amd/mhx/trace.py): $(nops) operations in the source"',
                 '"MHX_LOGDENSITY(x, d, data, ndata)"', '"    const mhx_real t$(i) = "', '"    return "', '"MHX_NAN"', '"-MHX_INF"',
                 "const LOG2PI = 1.8378770664093453", "pad = 13"):
        assert text in jl, text
    # Python's float.hex() is the literal format: the Julia pyhex must produce these for the same doubles
    assert (1.5).hex() == "0x1.8000000000000p+0" and (0.0).hex() == "0x0.0p+0" and (-0.5).hex() == "-0x1.0000000000000p-1"
    assert (5e-324).hex() == "0x0.0000000000001p-1022"


# ---- the LogDensityProblems form: src/AdvancedMH.jl:56,76-77, README.md:75-90, the only model form the reference's
# RobustAdaptiveMetropolis has methods for (src/RobustAdaptiveMetropolis.jl:175-181, test/RobustAdaptiveMetropolis.jl:1-9,30-56)
class LogTargetDensity:
    """README.md:80-88: `LogDensityProblems.logdensity(p::LogTargetDensity, θ) = density(θ)`, `dimension(p) = 2`"""

    def __init__(self, density):
        self.density = density

    def dimension(self):
        return 2

    def logdensity(self, theta):
        return self.density(theta)


class Gaussian2:
    """test/RobustAdaptiveMetropolis.jl:1-9,33-40: a zero-mean Gaussian with Σ = [σ² ρ; ρ σ²], ρ = σ²/2, as a LogDensityProblems
    object.  The log-density is written out (z = L \\ x by forward substitution, L = chol(Σ) on the host) so that the traced text
    does not depend on how a library orders its operations; `sq` / `lg` are the square root / logarithm of whoever traces."""

    def __init__(self, s2, sq, lg):
        rho = s2 / 2
        self.l11 = math.sqrt(s2)
        self.l21 = rho / self.l11
        self.l22 = math.sqrt(s2 - self.l21 * self.l21)
        self.lg11, self.lg22 = math.log(self.l11), math.log(self.l22)     # (tests/julia/check_tracer.jl passes these very doubles)

    def dimension(self):
        return 2

    def logdensity(self, x):
        z1 = x[0] / self.l11
        z2 = (x[1] - self.l21 * z1) / self.l22
        return -(z1 * z1 + z2 * z2) / 2 - self.lg11 - self.lg22 - J.LOG2PI


def test_logdensityproblems_form_lowers_to_the_same_source_as_the_closure():
    """`sample(LogTargetDensity(), spl, MCMCHIP(), N, nchains)` must reach the kernels `sample(DensityModel(density), ...)` reaches:
    the glue's LogDensityModel method traces θ -> LogDensityProblems.logdensity(ℓ, θ) with the dimension the problem reports; the
    emitted text -- the key of the hiprtc module -- is the closure's, character for character (fixture tests/golden/traced_readme.hip)."""
    from mhx import trace as T
    import mhx
    readme, _ = _twins()
    want = _fixture("traced_readme.hip")
    got_jl, npaths = J.trace_logdensity_problem(LogTargetDensity(readme_julia))
    assert got_jl == want and npaths == 2
    m = mhx.LogDensityModel(LogTargetDensity(readme), gradient=False)     # the Python mirror's form of the same model
    assert m.traced.source == want and m.dim == 2
    with pytest.raises(mhx.ArgumentError):
        mhx.LogDensityModel(lambda theta: 0.0)                            # a bare closure is a DensityModel, not a problem


def test_gaussian_problem_of_the_ram_test_has_a_pinned_source():
    """test/RobustAdaptiveMetropolis.jl's model (σ² = 10 and 0.01): model of the Julia tracer == Python tracer == the committed
    fixture; evaluated, it is logpdf(MvNormal(zeros(2), Σ), x)"""
    from mhx import trace as T
    import mhx
    import scipy.stats as st
    for s2, fx in ((10.0, "traced_ldp_gauss2_s10.hip"), (0.01, "traced_ldp_gauss2_s001.hip")):
        want = _fixture(fx)
        got_jl, npaths = J.trace_logdensity_problem(Gaussian2(s2, J.sqrt, J.log))
        m = mhx.LogDensityModel(Gaussian2(s2, T.sqrt, T.log), gradient=False)
        assert npaths == 1 and got_jl == want, fx
        assert m.traced.source == want, fx
        Sig = np.array([[s2, s2 / 2], [s2 / 2, s2]])
        for x in ([0.3, -1.1], [2.0, 0.5]):
            assert abs(m.traced.evaluate(x) - st.multivariate_normal([0, 0], Sig).logpdf(x)) < 1e-12 * max(1.0, 1 / s2)


def test_julia_glue_has_the_logdensitymodel_method():
    """what can be held without Julia: the method exists with the dispatch signature AbstractMCMC's ensemble call uses, takes its
    dimension from LogDensityProblems.dimension, traces through LogDensityProblems.logdensity, and forwards every keyword"""
    jl = open(os.path.join(HERE, "..", "advancedmh.jl_amd", "julia", "AdvancedMHHIP.jl")).read()
    m = re.search(r"function AbstractMCMC\.sample\(rng::Random\.AbstractRNG, model::AbstractMCMC\.LogDensityModel, sampler::AdvancedMH\.MHSampler,\s*"
                  r"ens::MCMCHIP, N::Integer, nchains::Integer; kwargs\.\.\.\)(.*?)\nend\n", jl, flags=re.S)
    assert m, "no sample(rng, ::LogDensityModel, ::MHSampler, ::MCMCHIP, N, nchains; kwargs...) method"
    body = m.group(1)
    assert "LogDensityProblems.dimension(ℓ)" in body and "LogDensityProblems.logdensity(ℓ, θ)" in body
    assert "ℓ isa DeviceLogDensity ? ℓ" in body                           # a catalogue target keeps its built-in kernel
    assert re.search(r"AbstractMCMC\.sample\(rng, AdvancedMH\.DensityModel\(dev\), sampler, ens, N, nchains; kwargs\.\.\.\)", body)
    assert "import LogDensityProblems" in jl
    assert "LogDensityProblems.capabilities(::Type{<:DeviceLogDensity})" in jl   # LogDensityModel(ℓ) refuses an object without it
