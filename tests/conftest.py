import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "advancedmh.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "soak_f32: the fp32 instance of this test belongs to the soak tier")
    config.addinivalue_line("markers", "soak: the long tail of the parameter products (random configurations, knob x width x shape); "
                                       "skipped unless the mark expression names it: -m 'gpu and soak' (or MHX_SOAK=1)")


    # Which compiler builds the run-time kernels (include/mhx.h: mhx_ctx_jit_compiler): the product's default is the installation's
    # clang++ as a child process -- 0.5 s more per kernel than hiprtc in-process, 150 s over the few hundred small kernels of the
    # default tier (profiles/r06_gpu_suite_time.txt).  The default tier therefore compiles with hiprtc (same source, same options, same
    # bits: test_run_time_kernels_by_either_compiler_give_the_same_chains) EXCEPT the full-size tests of the bench's own configurations
    # (tests/test_gpu_fullsize.py) and the compiler test; the soak tier runs everything under the product's default.
    expr = config.getoption("-m") or ""
    if not ("soak" in expr or os.environ.get("MHX_SOAK")):
        os.environ.setdefault("MHX_JIT_COMPILER", "hiprtc")


@pytest.fixture
def product_jit(monkeypatch):
    """the product's own choice of compiler for the run-time kernels (see pytest_configure)"""
    monkeypatch.delenv("MHX_JIT_COMPILER", raising=False)


def soak_tail(values, keep):
    """the first `keep` values in the default tier, the rest in the soak tier (the default `-m gpu` run must leave
    the driver's time limit a wide margin on a slow lease; every SURVEY section-8 row keeps its full-size test in the default tier)"""
    values = list(values)
    return values[:keep] + [pytest.param(*(v if isinstance(v, tuple) else (v,)), marks=pytest.mark.soak) for v in values[keep:]]


def pytest_collection_modifyitems(config, items):
    expr = config.getoption("-m") or ""
    if "soak" in expr or os.environ.get("MHX_SOAK"):
        return
    skip = pytest.mark.skip(reason="soak tier: run with -m 'gpu and soak' (or MHX_SOAK=1)")
    for it in items:
        # soak_f32: the fp32 instance of a slow test goes to the soak tier (fp64 is the reference's arithmetic and stays)
        f32_only = "soak_f32" in it.keywords and getattr(getattr(it, "callspec", None), "params", {}).get("real") == "f32"
        if "soak" in it.keywords or f32_only:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def mhx():
    import mhx as m
    m.lib()          # fails loudly if libmhx.so is missing
    return m


def pytest_generate_tests(metafunc):
    """Every `-m gpu` parity test runs in both widths: the reference computes in Float64 (f64), f32 is the same engine
    at half the bytes.  The `real` fixture switches the engine's default dtype and the oracle's build together."""
    if "real" in metafunc.fixturenames:
        metafunc.parametrize("real", ["f32", "f64"], indirect=True)


@pytest.fixture
def real(request):
    import mhx as m
    from oracle import oracle as O
    dt = request.param
    old_m, old_o = m.get_default_dtype(), O.get_dtype()
    m.set_default_dtype(dt)
    O.set_dtype(dt)
    yield dt
    m.set_default_dtype(old_m)
    O.set_dtype(old_o)


@pytest.fixture(autouse=True)
def _default_width(request):
    """Tests that do not ask for `real` run in fp32 (the golden fixtures of round 1 and the CPU-side oracle tests)."""
    if "real" in request.fixturenames:
        yield
        return
    import mhx as m
    from oracle import oracle as O
    old_m, old_o = m.get_default_dtype(), O.get_dtype()
    m.set_default_dtype("f32")
    O.set_dtype("f32")
    yield
    m.set_default_dtype(old_m)
    O.set_dtype(old_o)


class _EngineOptions:
    """mhx_ctx_set_option on the default contexts for the duration of one test.  The library reads no tuning variable from the
    environment; `setenv("MHX_EMCEE_MFMA", "0")` keeps the call shape of the rounds that did (name with or without the MHX_ prefix)."""

    def __init__(self, m):
        self.m = m

    def setenv(self, name, value):
        self.m.set_option(name[4:] if name.startswith("MHX_") else name, value)

    set = setenv

    def delenv(self, name):
        self.m.set_option(name[4:] if name.startswith("MHX_") else name, None)


@pytest.fixture
def engine():
    """explicit engine options (kernel form / tuning) on the release library"""
    import mhx as m
    yield _EngineOptions(m)
    m.clear_options()


@pytest.fixture
def tools_engine():
    """the tools build (libmhx_tools.so: timing probes, fault injection) bound for one test; its probe options taint the
    context -- chains of such runs are only built with allow_tainted=True"""
    import mhx as m
    m.use_library(m.TOOLS_LIB_PATH)
    try:
        yield _EngineOptions(m)
    finally:
        m.clear_options()
        m.use_library()
