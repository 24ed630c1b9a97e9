"""The algorithm of advancedmh.jl_amd/julia/MHXTrace.jl EXECUTED: a line-for-line transliteration into Python (no Julia exists in
the build container or on the GPU box).  A Julia closure may branch on parameter values (`cond ? a : b`, `s > 0 || return -Inf`:
README.md:29-31, test/emcee.jl:8), which needs a real Bool -- so a comparison of traced numbers returns the DECISION of the path
being traced and the function is run once per reachable combination of decisions, all runs recording into one hash-consed DAG; the
value is the decision tree over the paths' results.  Emission is the grammar of mhx/trace.py: the same operations in the same order
give the same text.  tests/test_julia_tracer.py holds this model, mhx.trace on the `where` twins and the committed fixtures
(tests/golden/traced_*.hip) to one another character for character; tests/julia/check_tracer.jl does the same for the Julia file."""
import math

CMP_TEXT = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "ne": "!="}
ARITH_TEXT = {"add": "+", "sub": "-", "mul": "*", "div": "/"}
LOG2PI = 1.8378770664093453


class TraceError(TypeError):
    pass


class Graph:
    def __init__(self):
        self.nodes, self.index = [], {}
        self.decisions, self.met, self.taken = [], [], {}

    def node(self, key):
        i = self.index.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self.index[key] = i
        return i


CURRENT = [None]


def graph():
    if CURRENT[0] is None:
        raise TraceError("a traced number was used outside trace_logdensity")
    return CURRENT[0]


def pyhex(v):
    v = float(v)
    if v != v:
        return "nan"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    return v.hex()


def lift(x):
    if isinstance(x, Traced):
        return x
    if isinstance(x, bool):
        raise TraceError("a condition is not a number")
    return Traced(graph().node(("c", pyhex(x))))


def bin_(op, a, b):
    x = lift(a)
    y = lift(b)
    return Traced(graph().node((op, x.i, y.i)))


def decide(key):
    g = graph()
    if key in g.taken:
        return g.taken[key]
    k = len(g.met) + 1
    d = g.decisions[k - 1] if k <= len(g.decisions) else True
    g.met.append(key)
    g.taken[key] = d
    return d


def cmp_(op, a, b):
    x = lift(a)
    y = lift(b)
    return decide(("cmp", op, x.i, y.i))


class Traced:
    """a traced real number (MHXTrace.Traced <: Real)"""
    __slots__ = ("i",)

    def __init__(self, i):
        self.i = i

    def __add__(self, o): return bin_("add", self, o)
    def __radd__(self, o): return bin_("add", o, self)
    def __sub__(self, o): return bin_("sub", self, o)
    def __rsub__(self, o): return bin_("sub", o, self)
    def __mul__(self, o): return bin_("mul", self, o)
    def __rmul__(self, o): return bin_("mul", o, self)
    def __truediv__(self, o): return bin_("div", self, o)
    def __rtruediv__(self, o): return bin_("div", o, self)
    def __neg__(self): return Traced(graph().node(("neg", self.i)))
    def __lt__(self, o): return cmp_("lt", self, o)
    def __le__(self, o): return cmp_("le", self, o)
    def __gt__(self, o): return cmp_("gt", self, o)
    def __ge__(self, o): return cmp_("ge", self, o)
    def __eq__(self, o): return cmp_("eq", self, o)
    def __ne__(self, o): return cmp_("ne", self, o)
    __hash__ = None

    def __pow__(self, n):
        if n == 0:
            return lift(1.0)
        if n == 0.5:
            return sqrt(self)
        r = self
        for _ in range(abs(int(n)) - 1):
            r = r * self
        return r if n > 0 else 1.0 / r


def _unary(op):
    def f(a):
        return Traced(graph().node((op, a.i)))
    return f


log, exp, sqrt, abs_ = _unary("log"), _unary("exp"), _unary("sqrt"), _unary("abs")


# Distributions with traced parameters (MHXTrace.jl: Normal built unchecked; the two log-densities written out)
def logpdf_normal(mu, sigma, x):
    z = (x - mu) / sigma
    return -(z * z + LOG2PI) / 2 - log(sigma)


def logpdf_inverse_gamma(alpha, theta, x):
    c = alpha * math.log(theta) - math.lgamma(alpha)
    return c - (alpha + 1) * log(x) - theta / x


def lit(key):
    return {"nan": "MHX_NAN", "inf": "MHX_INF", "-inf": "-MHX_INF"}.get(key) or "MHX_R(%s)" % key


def name(g, i):
    return lit(g.nodes[i][1]) if g.nodes[i][0] == "c" else "t%d" % i


def deps(g, i):
    key = g.nodes[i]
    op = key[0]
    if op in ("c", "x"):
        return []
    if op == "sel":
        return [key[2], key[3], key[1][2], key[1][3]]
    return list(key[1:])


def reachable(g, root):
    seen, stack = set(), [root]
    while stack:
        i = stack.pop()
        if i in seen:
            continue
        seen.add(i)
        stack.extend(deps(g, i))
    return seen


def cond_src(g, key):
    return "(%s %s %s)" % (name(g, key[2]), CMP_TEXT[key[1]], name(g, key[3]))


def rhs(g, i):
    key = g.nodes[i]
    op = key[0]
    if op == "x":
        return "x[%d]" % key[1]
    if op in ARITH_TEXT:
        return "%s %s %s" % (name(g, key[1]), ARITH_TEXT[op], name(g, key[2]))
    if op == "neg":
        return "-" + name(g, key[1])
    if op in ("log", "exp", "sqrt", "abs"):
        return "mhx_%s(%s)" % (op, name(g, key[1]))
    if op == "fma":
        return "mhx_fma(%s, %s, %s)" % (name(g, key[1]), name(g, key[2]), name(g, key[3]))
    if op == "sel":
        return "%s ? %s : %s" % (cond_src(g, key[1]), name(g, key[2]), name(g, key[3]))
    raise TraceError("unknown operation %s" % op)


def emit(g, out):
    lines, nops = [], 0
    for i in sorted(reachable(g, out)):
        op = g.nodes[i][0]
        if op == "c":
            continue
        lines.append("    const mhx_real t%d = %s;" % (i, rhs(g, i)))
        nops += op != "x"
    src = ["// traced by mhx.trace (advancedmh.jl_amd/mhx/trace.py): %d operations in the source" % nops,
           "MHX_LOGDENSITY(x, d, data, ndata)", "{"] + lines + ["    return %s;" % name(g, out), "}"]
    return "\n".join(src) + "\n"


def build(g, paths, depth):
    first = paths[0]
    if len(first["met"]) <= depth:
        return first["out"]
    key = first["met"][depth]
    yes = [p for p in paths if p["dec"][depth]]
    no = [p for p in paths if not p["dec"][depth]]
    a = build(g, yes, depth + 1)
    b = a if not no else build(g, no, depth + 1)
    if a == b:
        return a
    return g.node(("sel", key, a, b))


def trace_logdensity(f, dim, max_paths=64):
    if dim < 1:
        raise TraceError("dim must be >= 1")
    g = Graph()
    old = CURRENT[0]
    CURRENT[0] = g
    try:
        g.node(("c", pyhex(0.0)))
        theta = [Traced(g.node(("x", k))) for k in range(dim)]
        paths, stack = [], [[]]
        while stack:
            dec = stack.pop()
            g.decisions = dec
            g.met, g.taken = [], {}
            try:
                r = f(theta)
                if not isinstance(r, (Traced, int, float)):
                    raise TraceError("the log-density must return one number, got %s" % type(r).__name__)
                out = lift(r).i
            except TraceError:
                raise
            except Exception:
                out = g.node(("c", "nan"))
            full = [g.taken[c] for c in g.met]
            paths.append({"met": list(g.met), "dec": full, "out": out})
            if len(paths) > max_paths:
                raise TraceError("more than %d control-flow paths depend on parameter values" % max_paths)
            for k in range(len(dec) + 1, len(full) + 1):
                stack.append(full[:k - 1] + [False])
        return emit(g, build(g, paths, 0)), len(paths)
    finally:
        CURRENT[0] = old


def trace_logdensity_problem(problem, max_paths=64):
    """AdvancedMHHIP.jl's method for AbstractMCMC.LogDensityModel: the dimension from LogDensityProblems.dimension(l), the log-density
    traced through  theta -> LogDensityProblems.logdensity(l, theta)  -- `problem` has dimension() and logdensity(theta)."""
    return trace_logdensity(lambda theta: problem.logdensity(theta), int(problem.dimension()), max_paths)
