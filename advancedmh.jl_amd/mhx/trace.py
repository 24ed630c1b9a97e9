"""DensityModel(f) for a Python callable: trace `f` once over symbolic parameters and lower the recorded arithmetic to
the HIP source form of a user log-density (MHX_LOGDENSITY / MHX_LOGDENSITY_AND_GRADIENT, include/mhx.h
`mhx_target_from_hip_source`), which hiprtc then inlines into the sampling kernels.

Reference: src/AdvancedMH.jl:52-54 (`DensityModel(logdensity)` takes any function of the parameter vector),
README.md:26-31 (a closure over data), src/MALA.jl:54-93 (`logdensity_and_gradient`; the reference differentiates the
closure with ForwardDiff through LogDensityProblemsAD -- here the gradient is the reverse-mode sweep of the same trace).

What is traced is what runs: every `+ - * /`, `log`, `exp`, `sqrt`, `abs`, `fma`, `where` becomes ONE operation of the
engine's arithmetic (no re-association, no contraction: `a * b + c` is a product and a sum; ask for `fma(a, b, c)` to get
one), in the order the Python function performed them; sums run left to right.  Closed-over numbers and numpy arrays
become literals (exact: hexadecimal floating point; fp32 engines round each literal once).  Control flow on parameter
VALUES cannot be traced -- `if x[0] > 0:` raises; write `where(x[0] > 0, lp, -inf)` (both sides are evaluated).  Loops
over Python ranges unroll, so the program grows with the number of operations; a sum over the rows of a data set is
recorded as ONE loop by `sum_over(data, fn)` -- the rows travel to the device as the log-density's data block, the body is
traced once, what does not depend on the row is computed outside, and the gradient is one fused adjoint loop over the same
rows -- so a likelihood over 10^5 points costs a dozen lines of kernel source.

    def nig(theta):                       # test/emcee.jl:5-14
        s, m = theta
        lp = 2 * math.log(3) - 3 * T.log(s) - 3 / s
        for y in (0.0, 1.5, 2.0):
            lp = lp - 0.5 * (math.log(2 * math.pi) + T.log(s)) - 0.5 * (y - m) ** 2 / s
        return T.where(s > 0, lp, -math.inf)
    model = mhx.DensityModel(nig, dim=2)

The same callable still works on plain floats (`nig([1.0, 0.5])`): every function here dispatches on its argument.
"""
import math

import numpy as np

__all__ = ["log", "exp", "sqrt", "abs", "fma", "where", "minimum", "maximum", "square", "sum_over", "trace", "Traced", "Sym", "Vec", "NamedVec", "TraceError"]


class TraceError(TypeError):
    pass


class _Graph:
    """Hash-consed expression DAG of one trace (common subexpressions are recorded once)."""

    def __init__(self):
        self.nodes = []                 # (op, args) ; args are node ids, floats (op 'c') or ints (op 'x')
        self.index = {}
        self.loops, self.data, self.data_len, self.in_loop = [], [], 0, None      # sum_over: rows, offsets into the data block
        self.loop_group = {}            # loop node -> group key (the loops of one group are emitted as one)

    def node(self, op, *args):
        key = (op,) + args
        i = self.index.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self.index[key] = i
        return i


_ARITH = ("add", "sub", "mul", "div")
_CMP = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "ne": "!="}


def _const_key(v):
    v = float(v)
    return "nan" if v != v else v.hex()


class Sym:
    """A traced real number."""
    __slots__ = ("g", "i")
    __array_ufunc__ = None              # numpy defers to our reflected operators

    def __init__(self, g, i):
        self.g, self.i = g, i

    def _lift(self, o):
        if isinstance(o, Sym):
            if o.g is not self.g:
                raise TraceError("values of two different traces were mixed")
            return o
        if isinstance(o, (bool, np.bool_)) or isinstance(o, Cond):
            raise TraceError("a condition is not a number: use where(cond, a, b)")
        if isinstance(o, (int, float, np.integer, np.floating)):
            return Sym(self.g, self.g.node("c", _const_key(o)))
        return None

    def _bin(self, op, o, swap=False):
        b = self._lift(o)
        if b is None:
            return NotImplemented
        a = self
        if swap:
            a, b = b, a
        return Sym(self.g, self.g.node(op, a.i, b.i))

    def __add__(self, o):
        if type(o) is int and o == 0:   # sum()'s start value: no operation recorded
            return self
        return self._bin("add", o)

    def __radd__(self, o):
        if type(o) is int and o == 0:
            return self
        return self._bin("add", o, True)

    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return Sym(self.g, self.g.node("neg", self.i))
    def __pos__(self): return self
    def __abs__(self): return Sym(self.g, self.g.node("abs", self.i))

    def __pow__(self, n):
        """Integer powers by repeated multiplication (x**2 = x*x, x**3 = (x*x)*x, negative: 1 / x**|n|); x**0.5 = sqrt."""
        if isinstance(n, float) and n == 0.5:
            return sqrt(self)
        if isinstance(n, (float, np.floating)) and float(n).is_integer():
            n = int(n)
        if not isinstance(n, (int, np.integer)) or isinstance(n, bool):
            raise TraceError("only integer powers (and 0.5) can be traced; write exp(p * log(x)) for a real power")
        n = int(n)
        if n == 0:
            return self._lift(1.0)
        r = self
        for _ in range(builtins_abs(n) - 1):
            r = r * self
        return r if n > 0 else 1.0 / r

    def _cmp(self, op, o):
        b = self._lift(o)
        if b is None:
            return NotImplemented
        return Cond(self.g, ("cmp", op, self.i, b.i))

    def __lt__(self, o): return self._cmp("lt", o)
    def __le__(self, o): return self._cmp("le", o)
    def __gt__(self, o): return self._cmp("gt", o)
    def __ge__(self, o): return self._cmp("ge", o)
    def __eq__(self, o): return self._cmp("eq", o)
    def __ne__(self, o): return self._cmp("ne", o)
    __hash__ = None

    def __bool__(self):
        raise TraceError("the truth value of a traced number is not known while tracing")

    def __float__(self):
        raise TraceError("a traced number has no value while tracing (math.log(x) -> mhx.trace.log(x), float(x) -> x)")

    def __repr__(self):
        return "Sym(t%d)" % self.i


builtins_abs = abs


class Cond:
    """A traced condition: comparisons of traced numbers, combined with & | ~."""
    __slots__ = ("g", "key")
    __array_ufunc__ = None

    def __init__(self, g, key):
        self.g, self.key = g, key

    def __and__(self, o): return Cond(self.g, ("and", self.key, _cond_key(self, o)))
    def __or__(self, o): return Cond(self.g, ("or", self.key, _cond_key(self, o)))
    __rand__, __ror__ = __and__, __or__
    def __invert__(self): return Cond(self.g, ("not", self.key))

    def __bool__(self):
        raise TraceError("a branch on parameter values cannot be traced: `if cond:` -> where(cond, a, b); "
                         "`a and b` -> a & b")


def _cond_key(c, o):
    if isinstance(o, Cond):
        if o.g is not c.g:
            raise TraceError("values of two different traces were mixed")
        return o.key
    if isinstance(o, (bool, np.bool_)):
        return ("const", bool(o))
    raise TraceError("cannot combine a condition with %r" % (o,))


def _is_traced(*xs):
    return any(isinstance(x, (Sym, Vec)) for x in xs)


def _unary(op, pyfn):
    def f(x):
        if isinstance(x, Sym):
            return Sym(x.g, x.g.node(op, x.i))
        if isinstance(x, Vec):
            return Vec([f(e) for e in x])
        if isinstance(x, np.ndarray):
            return np.vectorize(pyfn, otypes=[float])(x)
        return pyfn(x)
    f.__name__ = op
    return f


def _pylog(x):
    x = float(x)
    return math.log(x) if x > 0 else (-math.inf if x == 0 else math.nan)


def _pyexp(x):
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def _pysqrt(x):
    return math.sqrt(x) if x >= 0 else math.nan


log = _unary("log", _pylog)
exp = _unary("exp", _pyexp)
sqrt = _unary("sqrt", _pysqrt)
abs = _unary("abs", lambda x: builtins_abs(x))    # noqa: A001 (mirrors the engine's name)


def square(x):
    return x * x


def fma(a, b, c):
    """a * b + c with ONE rounding (the engine's mhx_fma)."""
    s = next((v for v in (a, b, c) if isinstance(v, Sym)), None)
    if s is None:
        if _is_traced(a, b, c):
            return Vec._broadcast3(fma, a, b, c)
        return _fma_float(float(a), float(b), float(c))
    a, b, c = (s._lift(v) for v in (a, b, c))
    if a is None or b is None or c is None:
        return Vec._broadcast3(fma, a, b, c)
    return Sym(s.g, s.g.node("fma", a.i, b.i, c.i))


def _fma_float(a, b, c):
    if hasattr(math, "fma"):
        return math.fma(a, b, c)
    from fractions import Fraction
    if not all(map(math.isfinite, (a, b, c))):
        return a * b + c
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def where(cond, a, b):
    """cond ? a : b.  Both sides are evaluated (they have no side effects); a NaN comparison is false."""
    if isinstance(cond, Cond):
        g = cond.g
        lift = Sym(g, 0)._lift
        sa, sb = lift(a), lift(b)
        if sa is None or sb is None:
            raise TraceError("where(cond, a, b): a and b must be numbers")
        return Sym(g, g.node("sel", cond.key, sa.i, sb.i))
    if isinstance(cond, (bool, np.bool_)):
        return a if cond else b
    raise TraceError("where(): the first argument must be a comparison")


def minimum(a, b):
    return where(a < b, a, b)


def maximum(a, b):
    return where(a > b, a, b)


class Vec:
    """A vector of traced numbers (what `f` receives): indexing, slicing, iteration, element-wise arithmetic with numbers,
    sequences and numpy arrays, `.sum()`, `.dot(w)`, `A @ x` with a 2-D numpy array.  Sums run left to right."""
    __array_ufunc__ = None

    def __init__(self, items):
        self.items = list(items)

    def __len__(self): return len(self.items)
    def __iter__(self): return iter(self.items)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return Vec(self.items[k])
        if isinstance(k, (list, np.ndarray)):
            return Vec([self.items[int(i)] for i in k])
        return self.items[k]

    @staticmethod
    def _seq(o, n):
        if isinstance(o, Vec):
            o = o.items
        elif isinstance(o, np.ndarray):
            if o.ndim == 0:
                return [o.item()] * n
            if o.ndim != 1:
                raise TraceError("element-wise arithmetic between a parameter vector and a %d-D array" % o.ndim)
            o = list(o)
        elif isinstance(o, (list, tuple)):
            o = list(o)
        else:
            return [o] * n
        if len(o) != n:
            raise TraceError("length mismatch: %d vs %d" % (len(o), n))
        return o

    @staticmethod
    def _broadcast3(f, a, b, c):
        n = max(len(v) for v in (a, b, c) if isinstance(v, (Vec, list, tuple, np.ndarray)))
        return Vec([f(x, y, z) for x, y, z in zip(Vec._seq(a, n), Vec._seq(b, n), Vec._seq(c, n))])

    def _ew(self, o, f):
        return Vec([f(a, b) for a, b in zip(self.items, Vec._seq(o, len(self.items)))])

    def __add__(self, o): return self._ew(o, lambda a, b: a + b)
    def __radd__(self, o): return self._ew(o, lambda a, b: b + a)
    def __sub__(self, o): return self._ew(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._ew(o, lambda a, b: b - a)
    def __mul__(self, o): return self._ew(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._ew(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._ew(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._ew(o, lambda a, b: b / a)
    def __neg__(self): return Vec([-a for a in self.items])
    def __pow__(self, n): return Vec([a ** n for a in self.items])

    def sum(self):
        if not self.items:
            return 0.0
        s = self.items[0]
        for a in self.items[1:]:
            s = s + a
        return s

    def dot(self, o):
        return (self * o).sum()

    def __matmul__(self, o):
        if isinstance(o, np.ndarray) and o.ndim == 2:              # x @ A = (A^T x)
            return Vec([self.dot(o[:, j]) for j in range(o.shape[1])])
        return self.dot(o)

    def __rmatmul__(self, o):
        o = np.asarray(o)
        if o.ndim == 2:                                            # A @ x
            return Vec([Vec(self.items).__rdot(o[i]) for i in range(o.shape[0])])
        return self.__rdot(o)

    def __rdot(self, w):
        return Vec([b * a for a, b in zip(self.items, Vec._seq(w, len(self.items)))]).sum()

    def __repr__(self):
        return "Vec(%r)" % (self.items,)


# ---------------------------------------------------------------------------------------------------------------------
# lowering


def _lit(key):
    if key == "nan":
        return "MHX_NAN"
    v = float.fromhex(key)
    if math.isinf(v):
        return "MHX_INF" if v > 0 else "-MHX_INF"
    return "MHX_R(%s)" % key


_ACTIVE = []          # graphs being traced (innermost last): sum_over() records into the innermost one


def sum_over(data, fn):
    """sum(fn(row) for row in data), recorded as ONE loop over the rows instead of len(data) copies of fn's operations: `data`
    (a 1-D array: fn gets a number; 2-D: fn gets the row as a vector) travels to the device as the log-density's data block and
    the kernel source stays as small as fn.  Everything inside fn that does not depend on the row is computed once, outside the
    loop.  The sum runs over the rows in order, `acc = acc + term`, from 0.  Outside a trace (plain floats) it is that sum."""
    arr = np.asarray(data, dtype=np.float64)
    if arr.ndim not in (1, 2) or arr.size == 0:
        raise TraceError("sum_over: data must be a non-empty 1-D or 2-D array")
    rows = arr.reshape(arr.shape[0], -1)
    if not _ACTIVE:
        acc = 0.0
        for r in rows:
            acc = acc + fn(float(r[0]) if arr.ndim == 1 else Vec([float(v) for v in r]))
        return acc
    g = _ACTIVE[-1]
    if g.in_loop is not None:
        raise TraceError("sum_over inside sum_over is not supported")
    lid = len(g.loops)
    g.loops.append({"off": g.data_len, "n": rows.shape[0], "m": rows.shape[1]})
    g.data.append(rows.ravel())
    g.data_len += rows.size
    lv = [Sym(g, g.node("lv", lid, j)) for j in range(rows.shape[1])]
    g.in_loop = lid
    try:
        r = fn(lv[0] if arr.ndim == 1 else Vec(lv))
    finally:
        g.in_loop = None
    if isinstance(r, (int, float, np.integer, np.floating)):
        r = Sym(g, g.node("c", _const_key(r)))
    if not isinstance(r, Sym) or r.g is not g:
        raise TraceError("sum_over: fn must return one number")
    node = g.node("loop", lid, r.i)
    g.loop_group.setdefault(node, ("f", node))
    return Sym(g, node)


class Traced:
    """The recorded program of one log-density: `.source` (HIP source form, with the reverse-mode gradient unless
    gradient=False), `.dim`, `.data` (the rows of its sum_over loops, what the source's `data` argument must point to, or
    None), `.evaluate(x)` / `.gradient(x)` (the same program in numpy float64, for checks)."""

    def __init__(self, g, out, dim, gradient=True):
        self.g, self.out, self.dim = g, out, int(dim)
        self._ld = {}
        self.grad = self._reverse() if gradient else None
        self.data = np.concatenate(g.data) if g.data else None
        self.source = self._emit()

    # -- which loop (if any) a node is computed in: the loop whose row it depends on
    def _loop_of(self, i):
        memo, nodes = self._ld, self.g.nodes
        stack = [i]
        while stack:
            j = stack[-1]
            if j in memo:
                stack.pop()
                continue
            key = nodes[j]
            if key[0] == "lv":
                memo[j] = key[1]
                stack.pop()
                continue
            if key[0] in ("c", "x", "loop"):                  # (a loop node is the finished sum: an outer value)
                memo[j] = None
                stack.pop()
                continue
            deps = self._deps(j)
            todo = [d for d in deps if d not in memo]
            if todo:
                stack.extend(todo)
                continue
            ls = {memo[d] for d in deps} - {None}
            if len(ls) > 1:
                raise TraceError("a value depends on the rows of two different sum_over loops")
            memo[j] = ls.pop() if ls else None
            stack.pop()
        return memo[i]

    # -- one operation's vector-Jacobian product: (input node, contribution) pairs for the adjoint z of node i
    def _vjp(self, i, z):
        g = self.g
        S = lambda k: Sym(g, k)
        key = g.nodes[i]
        op, a = key[0], key[1:]
        if op == "add":
            return [(a[0], z), (a[1], z)]
        if op == "sub":
            return [(a[0], z), (a[1], -z)]
        if op == "mul":
            return [(a[0], z * S(a[1])), (a[1], z * S(a[0]))]
        if op == "div":
            q = z / S(a[1])
            return [(a[0], q), (a[1], -(q * S(i)))]
        if op == "neg":
            return [(a[0], -z)]
        if op == "log":
            return [(a[0], z / S(a[0]))]
        if op == "exp":
            return [(a[0], z * S(i))]
        if op == "sqrt":
            return [(a[0], z * (0.5 / S(i)))]
        if op == "abs":
            return [(a[0], where(S(a[0]) >= 0.0, z, -z))]
        if op == "fma":
            return [(a[0], z * S(a[1])), (a[1], z * S(a[0])), (a[2], z)]
        if op == "sel":
            c = Cond(g, a[0])
            return [(a[1], where(c, z, 0.0)), (a[2], where(c, 0.0, z))]
        return []

    # -- reverse mode: adjoints are recorded into the same graph, so that they share subexpressions with the value
    def _reverse(self):
        g = self.g
        S = lambda i: Sym(g, i)
        n_fwd = len(g.nodes)
        bar = {self.out: S(g.node("c", _const_key(1.0)))}
        reach = self._reachable([self.out])

        def acc(d, i, v):
            d[i] = v if i not in d else d[i] + v

        for i in sorted(reach, reverse=True):
            if i not in bar or i >= n_fwd or self._loop_of(i) is not None:
                continue
            z, key = bar[i], g.nodes[i]
            if key[0] != "loop":
                for k, v in self._vjp(i, z):
                    acc(bar, k, v)
                continue
            # d/du sum_rows f(row; u) = sum_rows df/du: the adjoint sweep of the loop body with the (row-independent) seed z
            # gives, per outer input u of the body, a row-dependent term; each is summed by a loop of its own over the same
            # rows (the emitter fuses the loops of one group into one)
            lid, root = key[1], key[2]
            if self._loop_of(root) is None:                         # the term does not depend on the row: n * term
                acc(bar, root, z * float(g.loops[lid]["n"]))
                continue
            inner = sorted((k for k in self._reachable([root]) if self._loop_of(k) == lid), reverse=True)
            ibar, obar = {root: z}, {}
            for k in inner:
                if k not in ibar:
                    continue
                for u, v in self._vjp(k, ibar[k]):
                    if g.nodes[u][0] in ("c", "lv"):
                        continue
                    acc(ibar if self._loop_of(u) == lid else obar, u, v)
            members = []
            for u in sorted(obar):
                node = g.node("loop", lid, obar[u].i)
                g.loop_group[node] = ("r", i)
                members.append((u, node))
            for u, node in members:                                 # (only now: nothing between the members may depend on one)
                acc(bar, u, S(node))
        zero = g.node("c", _const_key(0.0))
        out = []
        for k in range(self.dim):
            xi = g.index.get(("x", k))
            out.append(bar[xi].i if xi is not None and xi in bar else zero)
        return out

    def _cond_nodes(self, key, acc):
        if key[0] == "cmp":
            acc.extend(key[2:])
        elif key[0] in ("and", "or"):
            self._cond_nodes(key[1], acc); self._cond_nodes(key[2], acc)
        elif key[0] == "not":
            self._cond_nodes(key[1], acc)

    def _deps(self, i):
        key = self.g.nodes[i]
        op = key[0]
        if op in ("c", "x", "lv"):
            return []
        if op == "loop":
            return [key[2]]
        if op == "sel":
            d = list(key[2:])
            self._cond_nodes(key[1], d)
            return d
        return list(key[1:])

    def _reachable(self, roots):
        seen, stack = set(), list(roots)
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i)
            stack.extend(self._deps(i))
        return seen

    def _cond_src(self, key, name):
        if key[0] == "cmp":
            return "(%s %s %s)" % (name(key[2]), _CMP[key[1]], name(key[3]))
        if key[0] == "and":
            return "(%s && %s)" % (self._cond_src(key[1], name), self._cond_src(key[2], name))
        if key[0] == "or":
            return "(%s || %s)" % (self._cond_src(key[1], name), self._cond_src(key[2], name))
        if key[0] == "not":
            return "(!%s)" % self._cond_src(key[1], name)
        return "true" if key[1] else "false"

    def _rhs(self, i, name):
        key = self.g.nodes[i]
        op, a = key[0], key[1:]
        if op == "x":
            return "x[%d]" % a[0]
        if op == "lv":
            lp = self.g.loops[a[0]]
            return "data[%d + k * %d + %d]" % (lp["off"], lp["m"], a[1]) if lp["m"] > 1 else "data[%d + k]" % lp["off"]
        if op in _ARITH:
            return "%s %s %s" % (name(a[0]), {"add": "+", "sub": "-", "mul": "*", "div": "/"}[op], name(a[1]))
        if op == "neg":
            return "-%s" % name(a[0])
        if op in ("log", "exp", "sqrt", "abs"):
            return "mhx_%s(%s)" % (op, name(a[0]))
        if op == "fma":
            return "mhx_fma(%s, %s, %s)" % tuple(name(v) for v in a)
        if op == "sel":
            return "%s ? %s : %s" % (self._cond_src(a[0], name), name(a[1]), name(a[2]))
        raise AssertionError(op)

    def _body(self, roots):
        nodes, g = self.g.nodes, self.g
        live = self._reachable(roots)
        name = lambda i: _lit(nodes[i][1]) if nodes[i][0] == "c" else "t%d" % i
        # loop nodes of one group are emitted as ONE loop, at the position of the group's last live member
        groups = {}
        for i in live:
            if nodes[i][0] == "loop":
                groups.setdefault(g.loop_group[i], []).append(i)
        at = {max(m): sorted(m) for m in groups.values()}
        lines, nops = [], 0
        for i in sorted(live):
            op = nodes[i][0]
            if op == "c" or (op != "loop" and self._loop_of(i) is not None):
                continue
            if op != "loop":
                lines.append("    const mhx_real t%d = %s;" % (i, self._rhs(i, name)))
                nops += op != "x"
                continue
            if i not in at:
                continue
            members = at[i]
            lid = nodes[i][1]
            body = sorted(k for k in self._reachable([nodes[m][2] for m in members]) if self._loop_of(k) == lid)
            lines.append("    mhx_real %s;" % ", ".join("t%d = MHX_R(0x0.0p+0)" % m for m in members))
            lines.append("    for (int k = 0; k < %d; ++k) {" % g.loops[lid]["n"])
            for k in body:
                lines.append("        const mhx_real t%d = %s;" % (k, self._rhs(k, name)))
                nops += nodes[k][0] != "lv"
            for m in members:
                lines.append("        t%d = t%d + %s;" % (m, m, name(nodes[m][2])))
                nops += 1
            lines.append("    }")
        self._nops = nops
        return lines, name

    def _emit(self):
        lines, name = self._body([self.out])
        src = ["// traced by mhx.trace (advancedmh.jl_amd/mhx/trace.py): %d operations in the source%s" % (
                   self._nops, "" if self.data is None else "; data block of %d reals" % self.data.size),
               "MHX_LOGDENSITY(x, d, data, ndata)", "{"] + lines + ["    return %s;" % name(self.out), "}"]
        if self.grad is not None:
            lines, name = self._body([self.out] + self.grad)
            src += ["MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata)", "{"] + lines
            src += ["    g.set(%d, %s);" % (k, name(gi)) for k, gi in enumerate(self.grad)]
            src += ["    return %s;" % name(self.out), "}"]
        return "\n".join(src) + "\n"

    # -- the same program on numpy float64 (checks; the engine's log / exp differ from libm's in the last bit)
    def _eval_node(self, i, val, x, row):
        key = self.g.nodes[i]
        op, a = key[0], key[1:]

        def cond(c):
            if c[0] == "cmp":
                p, q = val[c[2]], val[c[3]]
                return {"lt": p < q, "le": p <= q, "gt": p > q, "ge": p >= q, "eq": p == q, "ne": p != q}[c[1]]
            if c[0] == "and":
                return cond(c[1]) and cond(c[2])
            if c[0] == "or":
                return cond(c[1]) or cond(c[2])
            if c[0] == "not":
                return not cond(c[1])
            return c[1]

        if op == "c":
            return math.nan if a[0] == "nan" else float.fromhex(a[0])
        if op == "x":
            return x[a[0]]
        if op == "lv":
            return row[a[1]]
        if op == "add":
            return val[a[0]] + val[a[1]]
        if op == "sub":
            return val[a[0]] - val[a[1]]
        if op == "mul":
            return val[a[0]] * val[a[1]]
        if op == "div":
            return np.float64(val[a[0]]) / np.float64(val[a[1]])
        if op == "neg":
            return -val[a[0]]
        if op == "log":
            return _pylog(val[a[0]])
        if op == "exp":
            return _pyexp(val[a[0]])
        if op == "sqrt":
            return _pysqrt(val[a[0]])
        if op == "abs":
            return builtins_abs(val[a[0]])
        if op == "fma":
            return _fma_float(float(val[a[0]]), float(val[a[1]]), float(val[a[2]]))
        if op == "sel":
            return val[a[1]] if cond(a[0]) else val[a[2]]
        raise AssertionError(op)

    def _run(self, x, roots):
        nodes, g, val = self.g.nodes, self.g, {}
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(all="ignore"):
            for i in sorted(self._reachable(roots)):
                if self._loop_of(i) is not None:
                    continue
                if nodes[i][0] != "loop":
                    val[i] = np.float64(self._eval_node(i, val, x, None))
                    continue
                lid, root = nodes[i][1], nodes[i][2]
                lp = g.loops[lid]
                rows = g.data[lid].reshape(lp["n"], lp["m"])
                body = sorted(k for k in self._reachable([root]) if self._loop_of(k) == lid)
                acc = np.float64(0.0)
                for r in rows:
                    for k in body:
                        val[k] = np.float64(self._eval_node(k, val, x, r))
                    acc = acc + val[root]
                val[i] = acc
        return [float(val[r]) for r in roots]

    def evaluate(self, x):
        return self._run(x, [self.out])[0]

    def gradient(self, x):
        if self.grad is None:
            raise TraceError("traced with gradient=False")
        return np.array(self._run(x, self.grad))

    @property
    def n_operations(self):
        """operations of the value function's source (a loop body counts once)"""
        self._body([self.out])
        return self._nops


class NamedVec(Vec):
    """Parameters that also answer to their names: x.a, x["a"] (the reference's NamedTuple parameters, test/runtests.jl:184)."""

    def __init__(self, items, names):
        Vec.__init__(self, items)
        self.__dict__["_names"] = {n: k for k, n in enumerate(names)}

    def __getattr__(self, name):
        names = self.__dict__.get("_names", {})
        if name in names:
            return self.items[names[name]]
        raise AttributeError(name)

    def __getitem__(self, k):
        if isinstance(k, str):
            return self.items[self._names[k]]
        return Vec.__getitem__(self, k)


def trace(f, dim, gradient=True, names=None):
    """Run `f` once on a vector of `dim` traced parameters and return the recorded program (a `Traced`).  With `names` the
    parameters are also reachable as x.<name> / x["<name>"]."""
    dim = int(dim)
    if dim < 1:
        raise TraceError("dim must be >= 1")
    if names is not None and len(names) != dim:
        raise TraceError("names: %d given for %d parameters" % (len(names), dim))
    g = _Graph()
    g.node("c", _const_key(0.0))        # node 0: lets where() lift plain numbers without a Sym at hand
    x = [Sym(g, g.node("x", k)) for k in range(dim)]
    x = NamedVec(x, names) if names is not None else Vec(x)
    _ACTIVE.append(g)
    try:
        out = f(x)
    finally:
        _ACTIVE.pop()
    if isinstance(out, Vec) and len(out) == 1:
        out = out[0]
    if isinstance(out, (int, float, np.integer, np.floating)):      # a constant density: still a program
        out = Sym(g, g.node("c", _const_key(out)))
    if not isinstance(out, Sym):
        raise TraceError("the log-density must return one number, got %r" % (type(out).__name__,))
    if out.g is not g:
        raise TraceError("the returned value belongs to another trace")
    return Traced(g, out.i, dim, gradient)
