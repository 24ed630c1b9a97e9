# MHXTrace.jl -- DensityModel(f) for a Julia closure: run `f` on traced numbers and lower the recorded arithmetic to the HIP source
# form of a user log-density (MHX_LOGDENSITY, include/mhx.h: mhx_target_from_hip_source), which hiprtc inlines into the
# sampling kernels -- "the user log-density is JIT-lowered to a device function evaluated per lane".
#
# Reference: src/AdvancedMH.jl:52-54,74 (`DensityModel(logdensity)` takes any function of the parameter vector),
# README.md:25-40 (a closure over data with a `cond ? a : b` on a parameter), test/emcee.jl:5-14 (`s > 0 || return -Inf`).
#
# The grammar is the one of the Python tracer (advancedmh.jl_amd/mhx/trace.py): a hash-consed DAG whose nodes are numbered in the
# order the function performed them (node 0 is the constant 0, nodes 1..dim the parameters x[0..dim-1], a literal takes a number
# when it is first used), emitted as one `const mhx_real t<i> = <one operation>;` per node -- no re-association, no contraction:
# `a * b + c` is a product and a sum -- with literals as exact hexadecimal floats `MHX_R(0x1.8000000000000p+1)`.  The same
# operations in the same order give the SAME TEXT, hence the same hiprtc module as the Python mirror, whose kernels the GPU suite
# holds to the oracle bit for bit (tests/golden/traced_*.hip pin the text; tests/julia_tracer_model.py is this algorithm executed).
#
# What Julia adds: `cond ? a : b`, `if`, `||` need a `Bool`, so a comparison of traced numbers cannot return a symbol.  It returns
# the DECISION of the path being traced, and `trace` runs the function once per reachable combination of decisions (first all
# `true`, then depth first with the last open decision flipped), all runs recording into one DAG (a prefix shared by two paths is
# recorded once).  The value is the decision tree over the paths' results, `cond ? a : b` per node -- both sides are evaluated on
# the device, as in the Python tracer's `where`.  A path that throws (a constructor's argument check, a DomainError) contributes NaN:
# the device rejects such a candidate where the reference would have thrown.  More than `max_paths` (64) paths is a TraceError.
#
# STATUS: no `julia` binary exists in the build container or on the GPU box; this file has never been parsed.  Its algorithm has
# been EXECUTED as tests/julia_tracer_model.py (a line-for-line transliteration), which reproduces the committed fixtures;
# tests/julia/check_tracer.jl compares this file's own output with the same fixtures on a maintainer's machine.
module MHXTrace

using Distributions

export trace_logdensity, TraceError, Traced

struct TraceError <: Exception
    msg::String
end
Base.showerror(io::IO, e::TraceError) = print(io, "TraceError: ", e.msg)

# a node: (op, args...) -- args are node ids (0-based, as they appear in the source), a literal's key (op :c) or a parameter
# index (op :x); a condition: (:cmp, op, a, b)
const NodeKey = Tuple
mutable struct Graph
    nodes::Vector{NodeKey}
    index::Dict{NodeKey,Int}
    decisions::Vector{Bool}        # the path being followed: the decision for the k-th NEW condition met on this run
    met::Vector{NodeKey}           # the conditions met on this run, in order
    taken::Dict{NodeKey,Bool}      # ... and what was decided (a condition met twice decides the same way)
end
Graph() = Graph(NodeKey[], Dict{NodeKey,Int}(), Bool[], NodeKey[], Dict{NodeKey,Bool}())

function node!(g::Graph, key::NodeKey)
    i = get(g.index, key, -1)
    if i < 0
        i = length(g.nodes)                       # 0-based id == position in the source
        push!(g.nodes, key)
        g.index[key] = i
    end
    return i
end

const CURRENT = Ref{Union{Graph,Nothing}}(nothing)    # the graph being traced (promotion and zero(T) need it to make literals)
function graph()
    g = CURRENT[]
    g === nothing && throw(TraceError("a traced number was used outside trace_logdensity"))
    return g
end

"a traced real number: node `i` of the graph being recorded"
struct Traced <: Real
    i::Int
    Traced(::Val{:node}, i::Int) = new(i)          # (no Traced(::Int): generic code that writes T(0) must get the LITERAL 0, see below)
end
tnode(i::Int) = Traced(Val(:node), i)

# Python's float.hex(): sign, 0x1.<13 hex digits>p<signed exponent>; 0x0.0p+0; subnormals 0x0.<13 digits>p-1022
function pyhex(v::Float64)
    isnan(v) && return "nan"
    isinf(v) && return v > 0 ? "inf" : "-inf"
    s = signbit(v) ? "-" : ""
    v == 0 && return s * "0x0.0p+0"
    b = reinterpret(UInt64, abs(v))
    e = Int((b >> 52) & 0x7ff)
    m = b & 0x000fffffffffffff
    lead, ex = e == 0 ? (0, -1022) : (1, e - 1023)
    return string(s, "0x", lead, ".", string(m, base = 16, pad = 13), "p", ex >= 0 ? "+" : "-", abs(ex))
end

lift(x::Traced) = x
lift(x::Bool) = throw(TraceError("a condition is not a number"))
lift(x::Real) = tnode(node!(graph(), (:c, pyhex(Float64(x)))))

Base.convert(::Type{Traced}, x::Traced) = x
Base.convert(::Type{Traced}, x::Real) = lift(x)
Traced(x::Real) = lift(x)
Base.promote_rule(::Type{Traced}, ::Type{<:Real}) = Traced
Base.zero(::Type{Traced}) = lift(0.0)
Base.one(::Type{Traced}) = lift(1.0)
Base.zero(::Traced) = lift(0.0)
Base.one(::Traced) = lift(1.0)
Base.float(x::Traced) = x
Base.real(x::Traced) = x
Base.AbstractFloat(x::Traced) = x
Base.Float64(::Traced) = throw(TraceError("a traced number has no value while tracing"))
Base.show(io::IO, x::Traced) = print(io, "Traced(t", x.i, ")")

# ---- arithmetic: ONE node per operation, operands lifted left to right (a literal is numbered when it is first used)
bin(op::Symbol, a, b) = (x = lift(a); y = lift(b); tnode(node!(graph(), (op, x.i, y.i))))
for (f, op) in ((:+, :add), (:-, :sub), (:*, :mul), (:/, :div))
    @eval Base.$f(a::Traced, b::Traced) = bin($(QuoteNode(op)), a, b)
    @eval Base.$f(a::Traced, b::Real) = bin($(QuoteNode(op)), a, b)
    @eval Base.$f(a::Real, b::Traced) = bin($(QuoteNode(op)), a, b)
end
Base.:-(a::Traced) = tnode(node!(graph(), (:neg, a.i)))
Base.:+(a::Traced) = a
for f in (:log, :exp, :sqrt, :abs)
    @eval Base.$f(a::Traced) = tnode(node!(graph(), ($(QuoteNode(f)), a.i)))
end
Base.abs2(a::Traced) = a * a
Base.inv(a::Traced) = 1.0 / a
Base.fma(a::Traced, b::Traced, c::Traced) = tnode(node!(graph(), (:fma, a.i, b.i, c.i)))
Base.fma(a::Real, b::Real, c::Traced) = fma(lift(a), lift(b), c)
Base.fma(a::Traced, b::Real, c::Real) = (x = a; y = lift(b); z = lift(c); fma(x, y, z))
Base.fma(a::Real, b::Traced, c::Real) = (x = lift(a); y = b; z = lift(c); fma(x, y, z))
Base.muladd(a::Traced, b::Real, c::Real) = a * b + c            # (what is traced is what runs: a product and a sum)
Base.muladd(a::Real, b::Traced, c::Real) = a * b + c
Base.muladd(a::Real, b::Real, c::Traced) = a * b + c
# integer powers by repeated multiplication (x^2 = x*x, x^3 = (x*x)*x, negative: 1 / x^|n|), x^0.5 = sqrt
function Base.:^(a::Traced, n::Integer)
    n == 0 && return lift(1.0)
    r = a
    for _ in 1:(abs(n) - 1)
        r = r * a
    end
    return n > 0 ? r : 1.0 / r
end
Base.literal_pow(::typeof(^), a::Traced, ::Val{n}) where {n} = a^n
function Base.:^(a::Traced, p::AbstractFloat)
    p == 0.5 && return sqrt(a)
    isinteger(p) && return a^Int(p)
    throw(TraceError("only integer powers (and 0.5) can be traced; write exp(p * log(x)) for a real power"))
end

# ---- conditions: a comparison returns the DECISION of the path being traced
function decide(key::NodeKey)
    g = graph()
    haskey(g.taken, key) && return g.taken[key]
    k = length(g.met) + 1
    d = k <= length(g.decisions) ? g.decisions[k] : true
    push!(g.met, key)
    g.taken[key] = d
    return d
end
compare(op::Symbol, a, b) = (x = lift(a); y = lift(b); decide((:cmp, op, x.i, y.i)))
for (f, op) in ((:<, :lt), (:<=, :le), (:>, :gt), (:>=, :ge), (:(==), :eq), (:!=, :ne))
    @eval Base.$f(a::Traced, b::Traced) = compare($(QuoteNode(op)), a, b)
    @eval Base.$f(a::Traced, b::Real) = compare($(QuoteNode(op)), a, b)
    @eval Base.$f(a::Real, b::Traced) = compare($(QuoteNode(op)), a, b)
end
Base.isless(a::Traced, b::Traced) = a < b
Base.isless(a::Traced, b::Real) = a < b
Base.isless(a::Real, b::Traced) = a < b
Base.isless(a::Traced, b::AbstractFloat) = a < b         # (Base has isless(::Real, ::AbstractFloat) and its mirror: without these two
Base.isless(a::AbstractFloat, b::Traced) = a < b         #  `isless(θ[1], 0.0)` is an ambiguity error instead of a decision)
Base.isnan(a::Traced) = a != a
Base.isinf(a::Traced) = abs(a) == Inf
Base.isfinite(a::Traced) = abs(a) < Inf
Base.iszero(a::Traced) = a == 0.0
Base.signbit(a::Traced) = a < 0.0
# min / max / ifelse as ONE select (no new path): cond ? a : b with both sides evaluated
sel(key::NodeKey, a, b) = (x = lift(a); y = lift(b); tnode(node!(graph(), (:sel, key, x.i, y.i))))
Base.min(a::Traced, b::Traced) = sel((:cmp, :lt, a.i, b.i), a, b)
Base.max(a::Traced, b::Traced) = sel((:cmp, :gt, a.i, b.i), a, b)
Base.min(a::Traced, b::Real) = min(a, lift(b))
Base.min(a::Real, b::Traced) = min(lift(a), b)
Base.max(a::Traced, b::Real) = max(a, lift(b))
Base.max(a::Real, b::Traced) = max(lift(a), b)

# ---- Distributions with traced parameters.  The constructors' argument checks (`σ >= 0`) would each open a path whose other side
# throws; the device form has no exception to throw, so a traced Normal is built unchecked (σ < 0: log σ = NaN, the candidate is
# rejected) and the two log-densities the reference's examples use are written out as the operations they are.
const LOG2PI = 1.8378770664093453           # log(2π), the literal of the Python twin
Distributions.Normal(μ::Traced, σ::Traced; check_args::Bool = true) = Distributions.Normal{Traced}(μ, σ)
function Distributions.logpdf(d::Distributions.Normal{Traced}, x::Real)
    z = (x - d.μ) / d.σ
    return -(z * z + LOG2PI) / 2 - log(d.σ)
end
# logpdf(InverseGamma(α, θ), x) = (α log θ − lgamma α) − (α + 1) log x − θ / x, the constant folded on the host
function Distributions.logpdf(d::Distributions.InverseGamma{<:AbstractFloat}, x::Traced)
    α, θ = Distributions.params(d)
    c = α * log(θ) - Distributions.SpecialFunctions.loggamma(α)
    return c - (α + 1) * log(x) - θ / x
end

# ---- lowering
const CMP_TEXT = Dict(:lt => "<", :le => "<=", :gt => ">", :ge => ">=", :eq => "==", :ne => "!=")
const ARITH_TEXT = Dict(:add => "+", :sub => "-", :mul => "*", :div => "/")

function lit(key::String)
    key == "nan" && return "MHX_NAN"
    key == "inf" && return "MHX_INF"
    key == "-inf" && return "-MHX_INF"
    return "MHX_R(" * key * ")"
end
name(g::Graph, i::Int) = g.nodes[i + 1][1] === :c ? lit(g.nodes[i + 1][2]) : "t$(i)"

function deps(g::Graph, i::Int)
    key = g.nodes[i + 1]
    op = key[1]
    (op === :c || op === :x) && return Int[]
    op === :sel && return Int[key[3], key[4], key[2][3], key[2][4]]
    return Int[k for k in key[2:end]]
end
function reachable(g::Graph, root::Int)
    seen = Set{Int}()
    stack = [root]
    while !isempty(stack)
        i = pop!(stack)
        i in seen && continue
        push!(seen, i)
        append!(stack, deps(g, i))
    end
    return seen
end
cond_src(g::Graph, key::NodeKey) = "(" * name(g, key[3]) * " " * CMP_TEXT[key[2]] * " " * name(g, key[4]) * ")"
function rhs(g::Graph, i::Int)
    key = g.nodes[i + 1]
    op = key[1]
    op === :x && return "x[$(key[2])]"
    haskey(ARITH_TEXT, op) && return name(g, key[2]) * " " * ARITH_TEXT[op] * " " * name(g, key[3])
    op === :neg && return "-" * name(g, key[2])
    op in (:log, :exp, :sqrt, :abs) && return "mhx_$(op)(" * name(g, key[2]) * ")"
    op === :fma && return "mhx_fma(" * name(g, key[2]) * ", " * name(g, key[3]) * ", " * name(g, key[4]) * ")"
    op === :sel && return cond_src(g, key[2]) * " ? " * name(g, key[3]) * " : " * name(g, key[4])
    throw(TraceError("unknown operation $(op)"))
end

function emit(g::Graph, out::Int)
    live = sort!(collect(reachable(g, out)))
    lines = String[]
    nops = 0
    for i in live
        op = g.nodes[i + 1][1]
        op === :c && continue
        push!(lines, "    const mhx_real t$(i) = " * rhs(g, i) * ";")
        nops += op !== :x
    end
    # (the first line is the Python tracer's, verbatim: the text is the key of the compiled module, shared by both mirrors)
    src = ["// traced by mhx.trace (advancedmh.jl_amd/mhx/trace.py): $(nops) operations in the source",
           "MHX_LOGDENSITY(x, d, data, ndata)", "{"]
    append!(src, lines)
    push!(src, "    return " * name(g, out) * ";")
    push!(src, "}")
    return join(src, "\n") * "\n"
end

# the decision tree over the traced paths: paths that agree on their first `depth` decisions are split by the next condition
function build(g::Graph, paths, depth::Int)
    first_path = paths[1]
    length(first_path.met) <= depth && return first_path.out                 # no further condition on this prefix: one result
    key = first_path.met[depth + 1]
    yes = [p for p in paths if p.dec[depth + 1]]
    no = [p for p in paths if !p.dec[depth + 1]]
    a = build(g, yes, depth + 1)
    b = isempty(no) ? a : build(g, no, depth + 1)
    a == b && return a
    return node!(g, (:sel, key, a, b))
end

"""
    trace_logdensity(f, dim; max_paths = 64) -> String

The HIP source (`MHX_LOGDENSITY(x, d, data, ndata) { ... }`) of `f(θ::Vector)`; see the header of this file.
"""
function trace_logdensity(f, dim::Integer; max_paths::Integer = 64)
    dim >= 1 || throw(TraceError("dim must be >= 1"))
    g = Graph()
    old = CURRENT[]
    CURRENT[] = g
    try
        node!(g, (:c, pyhex(0.0)))                                           # node 0, as in the Python tracer
        θ = Traced[tnode(node!(g, (:x, k))) for k in 0:(dim - 1)]
        paths = NamedTuple{(:met, :dec, :out),Tuple{Vector{NodeKey},Vector{Bool},Int}}[]
        stack = Vector{Bool}[Bool[]]
        while !isempty(stack)
            dec = pop!(stack)
            g.decisions = dec
            empty!(g.met)
            empty!(g.taken)
            out = try
                r = f(θ)
                r isa AbstractVector && length(r) == 1 && (r = r[1])
                r isa Real || throw(TraceError("the log-density must return one number, got $(typeof(r))"))
                lift(r).i
            catch e
                e isa TraceError && rethrow()
                node!(g, (:c, "nan"))                                        # the reference would have thrown here: reject
            end
            full = Bool[g.taken[c] for c in g.met]
            push!(paths, (met = copy(g.met), dec = full, out = out))
            length(paths) > max_paths && throw(TraceError("more than $(max_paths) control-flow paths depend on parameter values"))
            for k in (length(dec) + 1):length(full)                         # (the last one pushed -- the deepest open decision -- is traced next)
                push!(stack, vcat(full[1:(k - 1)], false))
            end
        end
        return emit(g, build(g, paths, 0))
    finally
        CURRENT[] = old
    end
end

end # module
