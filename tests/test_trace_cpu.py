"""DensityModel(f) for a Python callable (advancedmh.jl_amd/mhx/trace.py): the recorded program equals the callable on floats,
its reverse-mode gradient equals finite differences, the emitted source compiles for the host and agrees with both, and
what cannot be traced says so.  Reference: src/AdvancedMH.jl:52-54, README.md:26-38, src/MALA.jl:54-93."""
import ctypes as C
import math

import numpy as np
import pytest

import mhx.trace as T
import traced_models as M
import user_targets


@pytest.mark.parametrize("name", sorted(M.MODELS))
def test_traced_program_equals_the_callable(name):
    f, d, x0 = M.MODELS[name]
    tr = T.trace(f, d)
    rng = np.random.default_rng(5)
    for _ in range(20):
        x = np.array(x0) + 0.3 * rng.normal(size=d)
        want = f(T.Vec(list(x)))                                   # the same operations in the same order, on floats
        got = tr.evaluate(x)
        assert got == want, (name, x, got, want)
        assert got == f(np.array(x)) or abs(got - f(np.array(x))) < 1e-12 * abs(got)    # numpy's own summation order
        if math.isfinite(want):
            g = tr.gradient(x)
            h = 1e-6
            fd = np.array([(f(x + h * e) - f(x - h * e)) / (2 * h) for e in np.eye(d)])
            assert np.allclose(g, fd, rtol=2e-5, atol=2e-6), (name, x, g, fd)
    assert "MHX_LOGDENSITY(x, d, data, ndata)" in tr.source and "MHX_LOGDENSITY_AND_GRADIENT" in tr.source
    assert "MHX_LOGDENSITY_AND_GRADIENT" not in T.trace(f, d, gradient=False).source


def test_support_check_is_a_select_and_its_gradient_vanishes_outside():
    tr = T.trace(M.nig, 2)
    assert tr.evaluate([-1.0, 0.3]) == -math.inf
    assert (tr.gradient([-1.0, 0.3]) == 0).all()
    assert tr.evaluate([float("nan"), 0.3]) == -math.inf          # a NaN comparison is false, like `s > 0 || return -Inf`


def test_common_subexpressions_are_recorded_once():
    tr = T.trace(M.nig, 2)
    assert tr.source.count("mhx_log(") == 2                        # one in the value, one in the value + gradient function
    n = T.trace(lambda x: sum((x[0] - y) ** 2 for y in range(100)), 1).n_operations
    assert n == 100 + 100 + 99                                     # 100 differences, 100 squares, 99 additions: sum()'s 0 is not an operation


def test_literals_are_exact():
    v = 0.1 + 0.2
    tr = T.trace(lambda x: x[0] * v, 1)
    assert "MHX_R(%s)" % v.hex() in tr.source
    assert "MHX_INF" in T.trace(lambda x: T.where(x[0] > 0, x[0], -math.inf), 1).source


def test_what_cannot_be_traced_raises():
    with pytest.raises(T.TraceError, match="where"):
        T.trace(lambda x: x[0] if x[0] > 0 else -math.inf, 1)
    with pytest.raises(T.TraceError, match="mhx.trace.log"):
        T.trace(lambda x: math.log(x[0]), 1)
    with pytest.raises(T.TraceError, match="integer powers"):
        T.trace(lambda x: x[0] ** 1.5, 1)
    with pytest.raises(T.TraceError, match="one number"):
        T.trace(lambda x: x, 2)
    with pytest.raises(T.TraceError, match="length mismatch"):
        T.trace(lambda x: (x * np.ones(3)).sum(), 2)
    other = []
    T.trace(lambda x: other.append(x[0]) or x[0], 1)
    with pytest.raises(T.TraceError, match="two different traces"):
        T.trace(lambda x: x[0] + other[0], 1)


def test_vector_arithmetic_matches_numpy():
    rng = np.random.default_rng(2)
    A, w, x = rng.normal(size=(4, 4)), rng.normal(size=4), rng.normal(size=4)

    def f(v):
        r = (A @ v - w) / (1.0 + w * w)
        return -0.5 * r.dot(r) + (v[1:3] * 2.0).sum() - (T.exp(-(v * v))).sum() + (v @ A)[2]

    tr = T.trace(f, 4)
    r = (A @ x - w) / (1.0 + w * w)
    want = -0.5 * r @ r + (x[1:3] * 2).sum() - np.exp(-x * x).sum() + (x @ A)[2]
    assert abs(tr.evaluate(x) - want) < 1e-12
    assert tr.evaluate(x) == f(T.Vec(list(x)))                     # the same operations in the same order on floats


def test_density_model_takes_a_callable(monkeypatch):
    import mhx
    m = mhx.DensityModel(M.nig, dim=2)
    assert m.dim == 2 and isinstance(m.logdensity, mhx.HipLogDensity) and m.logdensity.source == m.traced.source
    with pytest.raises(mhx.ArgumentError, match="dim="):
        mhx.DensityModel(M.nig)
    with pytest.raises(mhx.ArgumentError, match="cannot be traced"):
        mhx.DensityModel(lambda x: math.log(x[0]), dim=1)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("name", sorted(M.MODELS))
def test_emitted_source_compiles_for_the_host_and_agrees(name, dt):
    """The emitted text is a valid user log-density (the form tests/user_targets.py compiles for the oracle): value and
    gradient agree with the recorded program (libm's log / exp vs the spec's: a few ulp)."""
    from oracle import oracle as O
    old = O.get_dtype()
    O.set_dtype(dt)
    try:
        f, d, x0 = M.MODELS[name]
        tr = T.trace(f, d)
        ut = user_targets.host_target(O, tr.source, d, data=tr.data)
        R = O.real()
        tol = 1e-12 if dt == "f64" else 2e-5
        rng = np.random.default_rng(8)
        gfn = C.CFUNCTYPE(C.c_double if dt == "f64" else C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)(ut.grad_addr)
        for _ in range(10):
            x = (np.array(x0) + 0.2 * rng.normal(size=d)).astype(R)
            want = tr.evaluate(x.astype(np.float64))
            got = float(ut(x))
            assert abs(got - want) <= tol * max(1.0, abs(want)), (name, got, want)
            g = np.empty(d, dtype=R)
            lp = gfn(x.ctypes.data, g.ctypes.data, d, C.addressof(ut._keep[2]) if ut._keep[2] is not None else None)
            assert np.float64(lp) == np.float64(got)
            gw = tr.gradient(x.astype(np.float64))
            assert np.allclose(g, gw, rtol=50 * tol, atol=50 * tol), (name, g, gw)
    finally:
        O.set_dtype(old)


def test_sum_over_is_one_loop_with_the_row_independent_part_outside():
    tr = T.trace(M.big_data_density, 2)
    value_fn = tr.source.split("MHX_LOGDENSITY_AND_GRADIENT")[0]
    assert value_fn.count("for (int k = 0; k < 2000; ++k)") == 1
    assert value_fn.index("mhx_log(") < value_fn.index("for (int k")          # log(sigma) is not recomputed per row
    assert tr.n_operations < 20 and tr.data.shape == (2000,)
    grad_fn = tr.source.split("MHX_LOGDENSITY_AND_GRADIENT")[1]
    assert grad_fn.count("for (int k = 0; k < 2000; ++k)") == 2               # the value's loop and ONE fused loop for both partials
    # outside a trace sum_over is the plain sum, in the same order
    x = [0.3, 1.9]
    assert tr.evaluate(x) == M.big_data_density(x)
    assert T.trace(M.regression, 3).data.shape == (300,)                      # two columns per row, row-major


def test_sum_over_misuse_is_reported():
    with pytest.raises(T.TraceError, match="inside sum_over"):
        T.trace(lambda x: T.sum_over([1.0, 2.0], lambda y: T.sum_over([3.0], lambda z: x[0] * y * z)), 1)
    with pytest.raises(T.TraceError, match="two different sum_over"):
        keep = []
        T.trace(lambda x: T.sum_over([1.0, 2.0], lambda y: keep.append(y) or x[0] * y) + T.sum_over([3.0, 4.0], lambda z: keep[0] * z), 1)
    with pytest.raises(T.TraceError, match="non-empty"):
        T.trace(lambda x: T.sum_over([], lambda y: x[0] * y), 1)
    # a term that does not depend on the row, and one that does not depend on the parameters either
    tr = T.trace(lambda x: T.sum_over([1.0, 2.0, 4.0], lambda y: x[0] * x[0]) + T.sum_over([1.0, 2.0, 4.0], lambda y: y), 1)
    assert tr.evaluate([3.0]) == 27.0 + 7.0 and tr.gradient([3.0])[0] == 18.0
