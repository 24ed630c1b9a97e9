# PhiloxStreams.jl -- the means to PIN the oracle (and through it the HIP kernels) against AdvancedMH.jl itself.
#
# `PhiloxStream <: Random.AbstractRNG` serves `randn`, `randexp`, `rand` and `rand(1:n)` from exactly the counter-based
# streams of DESIGN.md section 3 (Philox4x32-10, Float64 build: 52-bit uniforms, Box-Muller with the spec's own log /
# sincos polynomials), in the order the reference's samplers consume them.  Handing it to the UNMODIFIED package,
#
#     chain = sample(PhiloxStream(seed, chain_id; dim = d), model, RWMH(...), N; chain_type = Vector{...})
#
# makes `src/mh-core.jl:92-117`, `src/emcee.jl:39-102` and `src/RobustAdaptiveMetropolis.jl:123-278` consume the draws the
# oracle and the device consume, so their traces are comparable number by number (tests/julia/make_reference_traces.jl writes
# them, tests/test_julia_reference_traces.py checks them).  What stays different is rounding only: the engine's spec fuses
# `x + sigma z` and the sums of the log-densities with fma, Julia rounds products and sums separately -- relative
# differences of a few ulp per step, which is why the comparison is toleranced (1e-9) and accept decisions are compared
# wherever their margin exceeds it.
#
# STATUS: written against DESIGN.md section 3 and oracle/mhx_oracle.c; no `julia` binary exists in the build container
# or on the GPU box, so this file has never been executed.
module PhiloxStreams

using Random

export PhiloxStream

# ---- Philox4x32-10 ---------------------------------------------------------------------------------------------------
const PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
const PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85

function philox4x32_10(c0::UInt32, c1::UInt32, c2::UInt32, c3::UInt32, k0::UInt32, k1::UInt32)
    for _ in 1:10
        p0 = UInt64(PHILOX_M0) * UInt64(c0)
        p1 = UInt64(PHILOX_M1) * UInt64(c2)
        n0 = (UInt32(p1 >> 32) ⊻ c1) ⊻ k0
        n1 = UInt32(p1 & 0xffffffff)
        n2 = (UInt32(p0 >> 32) ⊻ c3) ⊻ k1
        n3 = UInt32(p0 & 0xffffffff)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 += PHILOX_W0                    # UInt32 arithmetic wraps
        k1 += PHILOX_W1
    end
    return (c0, c1, c2, c3)
end

const STREAM_PROPOSAL, STREAM_ACCEPT, STREAM_INIT, STREAM_EMCEE = UInt32(0), UInt32(1), UInt32(2), UInt32(3)

# counter = (id_lo, id_hi, step, stream << 28 | block), key = (seed_lo, seed_hi)
function block(seed::UInt64, id::UInt64, step::UInt32, stream::UInt32, blk::UInt32)
    return philox4x32_10(UInt32(id & 0xffffffff), UInt32(id >> 32), step, (stream << 28) | blk,
                         UInt32(seed & 0xffffffff), UInt32(seed >> 32))
end

# ---- the Float64 arithmetic spec (coefficients of tools/fit_coeffs64.py; same literals as oracle/mhx_oracle.c) ---------
const LN2_HI = 0x1.62e42feep-1
const LN2_LO = 0x1.a39ef35793c76p-33

"log of a positive normal finite x: m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), z = s^2"
function spec_log(x::Float64)
    ix = reinterpret(UInt64, x)
    t = ix - 0x3fe6a09e667f3bcd
    e = reinterpret(Int64, t) >> 52
    m = reinterpret(Float64, ix - (reinterpret(UInt64, e) << 52))
    f = m - 1.0
    s = f / (2.0 + f)
    z = s * s
    p = 0x1.2b59b70eb76c6p-3
    p = fma(p, z, 0x1.39fe42e9d4a8ap-3)
    p = fma(p, z, 0x1.7462b58e4403ap-3)
    p = fma(p, z, 0x1.c71c62e26212ep-3)
    p = fma(p, z, 0x1.2492492df3ba9p-2)
    p = fma(p, z, 0x1.99999999952ccp-2)
    p = fma(p, z, 0x1.5555555555558p-1)
    hfsq = (0.5 * f) * f
    ef = Float64(e)
    t1 = s * fma(z, p, hfsq)
    t2 = fma(ef, LN2_LO, t1)
    t3 = hfsq - t2
    t4 = f - t3
    return fma(ef, LN2_HI, t4)
end

"sin and cos of 2 pi a / 2^64, a = hi:lo"
function spec_sincos2pi(hi::UInt32, lo::UInt32)
    a = (UInt64(hi) << 32) | UInt64(lo)
    kk = a + 0x2000000000000000                                   # wraps mod 2^64
    q = Int(kk >> 62)
    ri = reinterpret(Int64, kk & 0x3fffffffffffffff) - Int64(0x2000000000000000)
    ti = ri >> 10                                                  # arithmetic shift: [-2^51, 2^51)
    r = Float64(ti) * 0x1p-54
    u = r * r
    s1 = -0x1.6cc577dadd922p-1
    s1 = fma(s1, u, 0x1.e8f036bcd3237p+1)
    s1 = fma(s1, u, -0x1.e3074d2614b2dp+3)
    s1 = fma(s1, u, 0x1.50783486facaap+5)
    s1 = fma(s1, u, -0x1.32d2cce62b872p+6)
    s1 = fma(s1, u, 0x1.466bc6775aae1p+6)
    s1 = fma(s1, u, -0x1.4abbce625be53p+5)
    ts = (r * u) * s1
    sp = fma(r, 0x1.921fb54442d18p+2, fma(r, 0x1.1a62633145c07p-52, ts))
    c2 = 0x1.1ebe62242e9d8p-2
    c2 = fma(c2, u, -0x1.b6df855cc99ffp+0)
    c2 = fma(c2, u, 0x1.f9d38850e5eedp+2)
    c2 = fma(c2, u, -0x1.a6d1f2a15a701p+4)
    c2 = fma(c2, u, 0x1.e1f506891b72fp+5)
    c2 = fma(c2, u, -0x1.55d3c7e3cbffap+6)
    c2 = fma(c2, u, 0x1.03c1f081b5ac4p+6)
    wc = (u * u) * c2
    vc = fma(u, -0x1.692b71366cc04p-50, wc)
    ac = fma(u, -0x1.3bd3cc9be45dep+4, 1.0)
    ec = fma(u, -0x1.3bd3cc9be45dep+4, 1.0 - ac)
    cp = ac + (vc + ec)
    ss = isodd(q) ? cp : sp
    cc = isodd(q) ? sp : cp
    (q == 2 || q == 3) && (ss = -ss)
    (q == 1 || q == 2) && (cc = -cc)
    return ss, cc
end

# 52-bit uniforms from two Philox words, k = hi:lo >> 12
u01_open(hi::UInt32, lo::UInt32) = fma(Float64((UInt64(hi) << 20) | UInt64(lo >> 12)), 0x1p-52, 0x1p-53)   # (0, 1)
u01_half(hi::UInt32, lo::UInt32) = Float64((UInt64(hi) << 20) | UInt64(lo >> 12)) * 0x1p-52                # [0, 1)

"Box-Muller from one Philox block: radius from (w0, w1), angle from (w2, w3)"
function normal_pair(w::NTuple{4,UInt32})
    l = spec_log(u01_open(w[1], w[2]))
    rad = sqrt(-2.0 * l)
    s, c = spec_sincos2pi(w[3], w[4])
    return rad * c, rad * s
end

"standard normal number k (0-based) of (seed, id, step, stream): normals 2b, 2b+1 come from Philox block b"
function normal_at(seed::UInt64, id::UInt64, step::UInt32, stream::UInt32, k::Int)
    n0, n1 = normal_pair(block(seed, id, step, stream, UInt32(k >> 1)))
    return iseven(k) ? n0 : n1
end

# ---- the ZIGGURAT generator of the fp64 spec (DESIGN.md section 3.11; oracle/mhx_oracle.c: zig_try, orc_zig_normal) -------
include(joinpath(@__DIR__, "zig_table.jl"))          # ZIG_N, ZIG_R, ZIG_NEG_RINV, ZIG_X (generated; same literals as mhx_zig_table.h)

"exp of the spec: n = rint(x log2e), Cody-Waite with the 32-bit LN2_HI, degree-11 polynomial, two exact scalings"
function spec_exp(x::Float64)
    isnan(x) && return x
    x > 0x1.62e42fefa39efp+9 && return Inf
    x < -0x1.74910d52d3052p+9 && return 0.0
    n = round(x * 0x1.71547652b82fep+0, RoundNearest)
    r = fma(n, -LN2_HI, x)
    r = fma(n, -LN2_LO, r)
    p = 0x1.61bfaa228dde5p-33
    p = fma(p, r, 0x1.1f7f2776cfaf2p-29)
    p = fma(p, r, 0x1.ae642c82e33d5p-26)
    p = fma(p, r, 0x1.27e4d41966f2fp-22)
    p = fma(p, r, 0x1.71de3a5aa7bb7p-19)
    p = fma(p, r, 0x1.a01a01a9e991bp-16)
    p = fma(p, r, 0x1.a01a01a0196acp-13)
    p = fma(p, r, 0x1.6c16c16c15a68p-10)
    p = fma(p, r, 0x1.1111111111111p-7)
    p = fma(p, r, 0x1.5555555555557p-5)
    p = fma(p, r, 0x1.5555555555555p-3)
    p = fma(p, r, 0.5)
    y = fma(r * r, p, r) + 1.0
    ni = Int(n); n1 = div(ni, 2); n2 = ni - n1            # div truncates toward zero like C's `/`
    y = y * reinterpret(Float64, UInt64(n1 + 1023) << 52)
    return y * reinterpret(Float64, UInt64(n2 + 1023) << 52)
end

"one candidate from the words (hi, lo): (accepted at once?, x, layer)"
function zig_try(hi::UInt32, lo::UInt32)
    layer = Int(lo & UInt32(ZIG_N - 1))                                          # bits 0..9
    k = (UInt64((lo >> 11) & 0x000fffff) << 32) | UInt64(hi)                     # 52 bits: (bits 11..30 of lo) : hi
    ax = (Float64(k) * 0x1p-52) * ZIG_X[layer + 1]
    x = (lo >> 31) == 1 ? -ax : ax                                               # bit 31 of lo is the sign
    return ax < ZIG_X[layer + 2], x, layer
end

"standard normal number n (0-based) of (seed, id, step, stream) by the ziggurat: attempt 0 from block n >> 1, rejections from stream | 4"
function zig_normal_at(seed::UInt64, id::UInt64, step::UInt32, stream::UInt32, n::Int)
    w = block(seed, id, step, stream, UInt32(n >> 1))
    ok, x, layer = isodd(n) ? zig_try(w[3], w[4]) : zig_try(w[1], w[2])
    ok && return x
    t = UInt32(1)
    while true
        v = block(seed, id, step, stream | UInt32(4), (UInt32(n) << 8) | (t & 0x000000ff))
        if layer == 0                                                            # Marsaglia's tail beyond r
            xx = spec_log(u01_open(v[1], v[2])) * ZIG_NEG_RINV
            yy = -spec_log(u01_open(v[3], v[4]))
            yy + yy >= xx * xx && return signbit(x) ? -(ZIG_R + xx) : (ZIG_R + xx)
        else                                                                     # the wedge of layer `layer`
            xl, xl1, xsq = ZIG_X[layer + 1], ZIG_X[layer + 2], x * x
            f0 = spec_exp(-0.5 * (xl * xl - xsq)); f1 = spec_exp(-0.5 * (xl1 * xl1 - xsq))
            fma(u01_half(v[3], v[4]), f0 - f1, f1) < 1.0 && return x
            ok, x, layer = zig_try(v[1], v[2])
            ok && return x
        end
        t += UInt32(1)
    end
end

"log of the accept uniform of `step` (one Philox block serves 2 consecutive steps)"
function accept_logu(seed::UInt64, id::UInt64, step::UInt32)
    w = block(seed, id, step >> 1, STREAM_ACCEPT, UInt32(0))
    return isodd(step) ? spec_log(u01_open(w[3], w[4])) : spec_log(u01_open(w[1], w[2]))
end

# ---- the scripted RNG ----------------------------------------------------------------------------------------------
"""
    PhiloxStream(seed, id; dim, nwalkers = 0, initial_draw = true, ziggurat = false)

One chain (`nwalkers == 0`: RWMH / MALA / RobustAdaptiveMetropolis -- per transition `dim` calls of `randn`, then one
`randexp`) or one ensemble (`nwalkers > 0`: per move `rand(sampler of 1:W-1)`, `rand()`, `randexp()`, walkers in order,
src/emcee.jl:39-58).  `id` is the global chain id (the ensemble id).  With `initial_draw` the first `dim` (ensemble:
`nwalkers * dim`) normals come from stream INIT (the initial `propose`, src/mh-core.jl:83, src/emcee.jl:29-34,
…RAM.jl:193); pass `initial_draw = false` when `initial_params` is given.  `ziggurat = true`: a chain's normals by the
engine's ziggurat generator (MHX_FLAG_ZIGGURAT; RWMH runs) instead of Box-Muller.
"""
mutable struct PhiloxStream <: Random.AbstractRNG
    seed::UInt64
    id::UInt64
    dim::Int
    nwalkers::Int
    step::UInt32        # transition (sweep) being served; 0 = the initial draws
    k::Int              # chain: normals served in this step; ensemble, step 0: normals served so far
    walker::Int         # ensemble: 0-based walker whose move is being served
    ziggurat::Bool      # chain: normals by the ziggurat generator
end
function PhiloxStream(seed::Integer, id::Integer; dim::Integer, nwalkers::Integer = 0, initial_draw::Bool = true, ziggurat::Bool = false)
    return PhiloxStream(UInt64(seed), UInt64(id), Int(dim), Int(nwalkers), initial_draw ? UInt32(0) : UInt32(1), 0, 0, ziggurat)
end

Base.copy(r::PhiloxStream) = PhiloxStream(r.seed, r.id, r.dim, r.nwalkers, r.step, r.k, r.walker, r.ziggurat)
Random.seed!(r::PhiloxStream, args...) = r                                   # the stream is fixed by (seed, id)

function Random.randn(r::PhiloxStream, ::Type{Float64} = Float64)
    if r.nwalkers > 0
        r.step == 0 || error("PhiloxStream: an ensemble draws normals only for its initial walkers")
        w, k = divrem(r.k, r.dim)                                            # walker w, its normal k
        z = normal_at(r.seed, (r.id << 32) | UInt64(w), UInt32(0), STREAM_INIT, k)
        r.k += 1
        r.k == r.nwalkers * r.dim && (r.step = UInt32(1); r.k = 0)
        return z
    end
    z = (r.ziggurat ? zig_normal_at : normal_at)(r.seed, r.id, r.step, r.step == 0 ? STREAM_INIT : STREAM_PROPOSAL, r.k)
    r.k += 1
    if r.step == 0 && r.k == r.dim                                           # the initial draw is complete
        r.step = UInt32(1); r.k = 0
    end
    return z
end

# -randexp(rng) is compared with the log acceptance ratio (src/mh-core.jl:108, src/emcee.jl:93, …RAM.jl:148): randexp = -log u
function Random.randexp(r::PhiloxStream, ::Type{Float64} = Float64)
    if r.nwalkers > 0
        w = block(r.seed, (r.id << 32) | UInt64(r.walker), r.step, STREAM_EMCEE, UInt32(1))
        e = -spec_log(u01_open(w[1], w[2]))
        r.walker += 1                                                        # the move is over
        r.walker == r.nwalkers && (r.walker = 0; r.step += UInt32(1))
        return e
    end
    e = -accept_logu(r.seed, r.id, r.step)
    r.step += UInt32(1)                                                      # the transition is over
    r.k = 0
    return e
end

# rand(rng) :: Float64 in [0, 1): the stretch uniform of src/emcee.jl:81 -- words 1, 2 of block 0 of the move
function Random.rand(r::PhiloxStream, ::Random.SamplerTrivial{Random.CloseOpen01{Float64}})
    r.nwalkers > 0 || error("PhiloxStream: rand() is scripted for ensemble moves only")
    w = block(r.seed, (r.id << 32) | UInt64(r.walker), r.step, STREAM_EMCEE, UInt32(0))
    return u01_half(w[2], w[3])
end

# rand(rng, Random.Sampler(rng, 1:n)): the partner offset of src/emcee.jl:48,52 -- first(r) + mulhi(word 0 of block 0, n)
struct PartnerSampler <: Random.Sampler{Int}
    lo::Int
    n::Int
end
Random.Sampler(::Type{PhiloxStream}, r::AbstractUnitRange{Int}, ::Random.Repetition) = PartnerSampler(first(r), length(r))
function Random.rand(r::PhiloxStream, sp::PartnerSampler)
    w = block(r.seed, (r.id << 32) | UInt64(r.walker), r.step, STREAM_EMCEE, UInt32(0))
    return sp.lo + Int((UInt64(w[1]) * UInt64(sp.n)) >> 32)
end

# raw words, for anything unscripted (e.g. AbstractMCMC drawing a seed): a separate stream that the samplers never touch
Random.rand(r::PhiloxStream, ::Random.SamplerType{UInt64}) =
    (w = block(r.seed, r.id, UInt32(0xffffffff), UInt32(15), UInt32(0)); (UInt64(w[1]) << 32) | UInt64(w[2]))

end # module
