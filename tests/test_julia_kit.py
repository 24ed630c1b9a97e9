"""What can be checked about the Julia parity kit WITHOUT Julia (none exists in the build container or on the GPU box):

* every hex-float literal and Philox constant of tests/julia/PhiloxStreams.jl equals, in order, the one in the fp64 block of
  oracle/mhx_oracle.c (a one-digit drift would otherwise surface only on a maintainer's machine);
* the integer masks / offsets of the argument reductions agree;
* tests/julia/make_reference_traces.jl and tests/julia_cases.py list the same cases with the same (N, seed, first chain,
  chains) and schedule keywords;
* no RobustAdaptiveMetropolis / MALA case hands a `DensityModel` to `sample`: their `step` methods dispatch on
  `AbstractMCMC.LogDensityModel` only (/root/reference/src/RobustAdaptiveMetropolis.jl:175-181,216-222,247-253,
  src/MALA.jl:101-104), so the model must be a LogDensityProblems object as in test/RobustAdaptiveMetropolis.jl:1-9.
"""
import os
import re

import pytest

import julia_cases

HERE = os.path.dirname(os.path.abspath(__file__))
JL_STREAMS = os.path.join(HERE, "julia", "PhiloxStreams.jl")
JL_TRACES = os.path.join(HERE, "julia", "make_reference_traces.jl")
ORACLE_C = os.path.join(HERE, "..", "oracle", "mhx_oracle.c")

HEXF = re.compile(r"-?0x[0-9a-fA-F]+(?:\.[0-9a-fA-F]*)?p[+-]?\d+")
HEXI = re.compile(r"0x[0-9a-fA-F]{8,16}(?![0-9a-fA-F.p])")


def _jl_function(src, name):
    m = re.search(r"^function %s\(.*?^end$" % re.escape(name), src, re.S | re.M)
    assert m, "PhiloxStreams.jl: function %s not found" % name
    return m.group(0)


def _c_fp64_block(src):
    i = src.index("#else /* ORC_F64 */")
    j = src.index("#endif", src.index("static void emcee_draws", i))
    return src[i:j]


def _c_function(block, signature):
    i = block.index(signature)
    j = block.index("\n}\n", i)
    return block[i:j]


def _floats(text):
    """hex-float literals in order of appearance, comments of either language removed"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#.*?$", "", text, flags=re.M)
    return [float.fromhex(t) for t in HEXF.findall(text)]


def _ints(text, comment):
    text = re.sub(comment, "", text, flags=re.S | re.M)
    return [int(t, 16) for t in HEXI.findall(text)]


@pytest.fixture(scope="module")
def sources():
    return open(JL_STREAMS).read(), _c_fp64_block(open(ORACLE_C).read()), open(ORACLE_C).read()


@pytest.mark.parametrize("jl_name,c_sig", [("spec_log", "double orc_log(double x)"),
                                           ("spec_sincos2pi", "void orc_sincos2pi_u64(")])
def test_polynomial_literals_match_the_oracle(sources, jl_name, c_sig):
    jl, c64, _ = sources
    jf = _floats(_jl_function(jl, jl_name))
    cf = _floats(_c_function(c64, c_sig))
    # the C side carries range-handling literals the Julia side does not need (subnormal rescue 0x1p54 in orc_log)
    cf = [v for v in cf if v != float.fromhex("0x1p54")]
    assert len(jf) >= 7 and jf == cf, "hex-float literals drifted between PhiloxStreams.jl:%s and mhx_oracle.c:%s\n jl %s\n c  %s" % (
        jl_name, c_sig, [v.hex() for v in jf], [v.hex() for v in cf])


def test_integer_reduction_constants_match_the_oracle(sources):
    jl, c64, _ = sources
    for jl_name, c_sig in (("spec_log", "double orc_log(double x)"), ("spec_sincos2pi", "void orc_sincos2pi_u64(")):
        ji = _ints(_jl_function(jl, jl_name), r"#.*?$")
        ci = _ints(_c_function(c64, c_sig), r"/\*.*?\*/")
        # orc_log also tests for inf / subnormal inputs (never reached by uniforms in (0,1)): drop those two masks
        ci = [v for v in ci if v not in (0x7ff0000000000000, 0x0010000000000000)]
        assert ji and ji == ci, (jl_name, [hex(v) for v in ji], [hex(v) for v in ci])


def test_ln2_split_and_uniform_scales_match_the_oracle(sources):
    jl, c64, _ = sources
    for name in ("LN2_HI", "LN2_LO"):
        j = float.fromhex(re.search(r"const LN2_HI = (\S+)" if name == "LN2_HI" else r"const LN2_LO = (\S+)", jl).group(1))
        c = float.fromhex(re.search(r"#define %s\s+(\S+)" % name, c64).group(1))
        assert j == c, name
    # u01_open / u01_half: fma(k, 2^-52, 2^-53) and k 2^-52 with k = hi:lo >> 12 (hi << 20 | lo >> 12)
    ju = re.search(r"^u01_open\(.*$", jl, re.M).group(0) + re.search(r"^u01_half\(.*$", jl, re.M).group(0)
    cu = _c_function(c64, "double orc_u01_open(") + _c_function(c64, "double orc_u01_half(")
    assert sorted(set(_floats(ju))) == sorted(set(_floats(cu))) == [2.0 ** -53, 2.0 ** -52]
    for t in (ju, cu):
        assert "<< 20" in t and ">> 12" in t


def test_philox_constants_and_stream_tags_match_the_oracle(sources):
    jl, _, c = sources
    cphil = c[c.index("void orc_philox4x32_10"):c.index("static void philox_at")]
    want = {int(t, 16) for t in re.findall(r"0x[0-9A-Fa-f]{8}(?=u)", cphil)}
    got = {int(t, 16) for t in re.findall(r"0x[0-9A-Fa-f]{8}\b", jl[jl.index("const PHILOX_M0"):jl.index("function philox4x32_10")])}
    assert want == got == {0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85}
    m = re.search(r"const STREAM_PROPOSAL, STREAM_ACCEPT, STREAM_INIT, STREAM_EMCEE = UInt32\((\d)\), UInt32\((\d)\), UInt32\((\d)\), UInt32\((\d)\)", jl)
    from oracle import oracle as O
    assert tuple(int(g) for g in m.groups()) == (O.STREAM_PROPOSAL, O.STREAM_ACCEPT, O.STREAM_INIT, O.STREAM_EMCEE)
    assert "(stream << 28) | blk" in jl and "(stream << 28) | block" in c


def test_philox_kat_through_a_python_reading_of_the_julia_round_function(sources):
    """The Julia round function, transliterated token by token (same operand order), must give the Random123 vectors: catches a
    swapped word or key in PhiloxStreams.jl's philox4x32_10 as far as a reading can."""
    jl, _, _ = sources
    body = _jl_function(jl, "philox4x32_10")
    assert re.search(r"p0 = UInt64\(PHILOX_M0\) \* UInt64\(c0\)", body) and re.search(r"p1 = UInt64\(PHILOX_M1\) \* UInt64\(c2\)", body)
    assert "n0 = (UInt32(p1 >> 32) ⊻ c1) ⊻ k0" in body and "n2 = (UInt32(p0 >> 32) ⊻ c3) ⊻ k1" in body
    assert "n1 = UInt32(p1 & 0xffffffff)" in body and "n3 = UInt32(p0 & 0xffffffff)" in body
    assert "k0 += PHILOX_W0" in body and "k1 += PHILOX_W1" in body and "for _ in 1:10" in body

    def philox(c, k):
        c0, c1, c2, c3 = c
        k0, k1 = k
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
            c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0), p1 & 0xffffffff, ((p0 >> 32) ^ c3 ^ k1), p0 & 0xffffffff
            k0, k1 = (k0 + 0x9E3779B9) & 0xffffffff, (k1 + 0xBB67AE85) & 0xffffffff
        return (c0, c1, c2, c3)
    assert philox((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


# ---- make_reference_traces.jl <-> julia_cases.py -------------------------------------------------------------------------------
def _jl_cases():
    src = re.sub(r"#.*?$", "", open(JL_TRACES).read(), flags=re.M)
    out = {}
    for m in re.finditer(r'run_chains\("(\w+)",\s*(.*?)\)\s*$', src, re.S | re.M):
        out[m.group(1)] = " ".join(m.group(2).split())
    return src, out


def test_trace_script_and_python_case_list_agree():
    src, jl = _jl_cases()
    names = set(jl) | set(re.findall(r'"(emcee_\w+?)_samples\.npy"', src))
    assert names == set(julia_cases.JULIA_CASES), names ^ set(julia_cases.JULIA_CASES)


def test_no_density_model_reaches_ram_or_mala():
    _, jl = _jl_cases()
    checked = 0
    for name, args in jl.items():
        if name.startswith(("ram", "mala")):
            assert not args.lstrip().startswith("DensityModel"), "%s: RobustAdaptiveMetropolis / MALA dispatch on LogDensityModel only" % name
            assert args.lstrip().startswith("GaussianLDP("), name
            checked += 1
        else:
            assert args.lstrip().startswith("DensityModel("), name
    assert checked >= 5
    src = open(JL_TRACES).read()
    # the LogDensityProblems interface the reference's own test model implements (test/RobustAdaptiveMetropolis.jl:1-9)
    for needle in ("LogDensityProblems.dimension(m::GaussianLDP)", "LogDensityProblems.capabilities(::Type{<:GaussianLDP})",
                   "LogDensityProblems.logdensity(m::GaussianLDP, x)", "LogDensityProblems.logdensity_and_gradient(m::GaussianLDP, x)",
                   "using AdvancedMH, AbstractMCMC, Distributions, LinearAlgebra, LogDensityProblems, Random"):
        assert needle in src, needle


def test_case_numbers_agree_between_julia_and_python(oracle):
    """(N, seed, first_chain, chains, dim) and the schedule keywords of every run_chains call against what the Python case
    hands the oracle (read back from the result shapes and from the call's source)."""
    import inspect
    old = oracle.get_dtype()
    oracle.set_dtype("f64")
    try:
        _, jl = _jl_cases()
        for name, args in jl.items():
            m = re.search(r",\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\w+)\s*(?:;|$)", args)
            assert m, (name, args)
            N, seed, first, C = (int(g) for g in m.groups()[:4])
            r = julia_cases.JULIA_CASES[name](oracle)
            d = r["samples"].shape[1] - 1
            assert r["samples"].shape == (N, d + 1, C), name
            pysrc = inspect.getsource(julia_cases.JULIA_CASES[name])
            assert re.search(r"\b%d, %d, %d\b" % (seed, first, C), pysrc), (name, seed, first, C)
            kw = dict(re.findall(r"(discard_initial|thinning|num_warmup)\s*=\s*(\d+)", args))
            sched = re.search(r"O\.schedule\(([^)]*)\)", pysrc).group(1).split(",")
            sched = [int(v) for v in sched] + [0, 1, 0][len(sched) - 1:]
            want_di = int(kw.get("discard_initial", kw.get("num_warmup", 0)))      # upstream: discard_initial defaults to num_warmup
            assert sched[0] == N and sched[1] == want_di and sched[2] == int(kw.get("thinning", 1)) and sched[3] == int(kw.get("num_warmup", 0)), (name, sched, kw)
    finally:
        oracle.set_dtype(old)


def test_exp_literals_and_ziggurat_table_match_the_oracle(sources):
    """spec_exp (used by the ziggurat's wedge test) against orc_exp, and tests/julia/zig_table.jl against the generated header
    the device and the oracle compile (one generator writes all three: tools/gen_zig_table.py)."""
    jl, c64, _ = sources
    jf = _floats(_jl_function(jl, "spec_exp"))
    cf = _floats(_c_function(c64, "double orc_exp(double x)"))
    log2e = float.fromhex(re.search(r"#define LOG2E\s+(\S+)", c64).group(1))     # a macro on the C side, a literal in the Julia function
    assert jf.count(log2e) == 1
    jf.remove(log2e)
    assert len(jf) >= 13 and jf == cf, ([v.hex() for v in jf], [v.hex() for v in cf])
    for needle in ("fma(n, -LN2_HI, x)", "fma(n, -LN2_LO, r)", "fma(r * r, p, r) + 1.0", "div(ni, 2)"):
        assert needle in _jl_function(jl, "spec_exp"), needle
    hdr = open(os.path.join(HERE, "..", "oracle", "mhx_zig_table.h")).read()
    jt = open(os.path.join(HERE, "julia", "zig_table.jl")).read()
    # (the fp32 table of round 6 follows the fp64 one in both files: compare table by table)
    h64 = hdr[hdr.index("#define MHX_ZIG_TABLE"):]
    h64 = h64[:h64.index("}")]
    j64 = jt[jt.index("const ZIG_X"):]
    j64 = j64[:j64.index("]")]
    hx = [float.fromhex(t) for t in HEXF.findall(h64)]
    jx = [float.fromhex(t) for t in HEXF.findall(j64)]
    assert len(hx) == 1025 and hx == jx
    h32 = hdr[hdr.index("#define MHX_ZIG32_TABLE"):]
    j32 = jt[jt.index("const ZIG32_X"):]
    hx32 = [float.fromhex(t) for t in HEXF.findall(h32[:h32.index("}")])]
    jx32 = [float.fromhex(t) for t in HEXF.findall(j32[:j32.index("]")])]
    assert len(hx32) == 257 and hx32 == jx32
    for name_h, name_j in (("MHX_ZIG_R", "ZIG_R"), ("MHX_ZIG_NEG_RINV", "ZIG_NEG_RINV")):
        a = float.fromhex(re.search(r"#define %s (\S+)" % name_h, hdr).group(1))
        b = float.fromhex(re.search(r"const %s = (\S+)" % name_j, jt).group(1))
        assert a == b, name_h
    assert int(re.search(r"const ZIG_N = (\d+)", jt).group(1)) == int(re.search(r"#define MHX_ZIG_N (\d+)", hdr).group(1)) == 1024


def test_julia_ziggurat_follows_the_oracle_bit_layout(sources):
    jl, _, c = sources
    body = _jl_function(jl, "zig_try")
    # layer = bits 0..9, mantissa = (bits 11..30 of lo) : hi, sign = bit 31 of lo -- the same expressions as oracle zig_try
    assert "lo & UInt32(ZIG_N - 1)" in body and "(lo >> 11) & 0x000fffff" in body and "(lo >> 31) == 1" in body
    czt = c[c.index("static int zig_try("):c.index("double orc_zig_normal(")]
    assert "(lo >> 11) & 0xfffffu" in czt and "(lo >> 31)" in czt and "lo & (uint32_t)(MHX_ZIG_N - 1)" in czt
    jz = _jl_function(jl, "zig_normal_at")
    cz = c[c.index("double orc_zig_normal("):c.index("static void normals_gen(")]
    # retry stream and block numbering, tail and wedge tests
    assert "stream | UInt32(4)" in jz and "stream | 4u" in cz
    assert "(UInt32(n) << 8) | (t & 0x000000ff)" in jz and "(n << 8) | (t & 255u)" in cz
    assert "yy + yy >= xx * xx" in jz and "yy + yy >= xx * xx" in cz
    assert "fma(u01_half(v[3], v[4]), f0 - f1, f1) < 1.0" in jz and "fma(orc_u01_half(w[2], w[3]), f0 - f1, f1) < 1.0" in cz
