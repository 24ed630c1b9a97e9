# AdvancedMHHIP.jl -- thin `ccall` layer over libmhx.so (include/mhx.h, ABI 0.6) that plugs the MI355X engine
# into AdvancedMH.jl through AbstractMCMC's ensemble dispatch:
#
#     model = DensityModel(density)                       # README.md:25-40, unchanged: the closure is traced and JIT-lowered
#     chain = sample(model, RWMH(MvNormal(zeros(2), I)), MCMCHIP(), 100_000, 65_536;
#                    param_names = ["μ", "σ"], chain_type = Chains)
#     chain, stats = sample(model, spl, MCMCHIP(devices = 0:7), 1_000, 262_144; return_stats = true)   # 8 GPUs, ONE call
#     sample(LogTargetDensity(), spl, MCMCHIP(), 100_000, 65_536)   # README.md:75-90: the LogDensityProblems form -- the ONLY form the
#     sample(Gaussian(Σ), RobustAdaptiveMetropolis(), MCMCHIP(), N, nchains; num_warmup = N)   # reference's RAM takes (…RAM.jl:175-181)
#
# One `ccall` sequence runs all chains x all steps on the GPU(s) and returns the (iterations, params..., lp, chains) tensor that
# ext/AdvancedMHMCMCChainsExt.jl:96-118 wraps.  The engine computes in Float64 by default -- the reference's arithmetic -- or
# in Float32 (`MCMCHIP(T = Float32)`).
#
# STATUS: written against include/mhx.h and reviewed by hand; neither this container nor the GPU box has a `julia` binary, so this
# file has never been executed (DESIGN.md section 2).  tests/test_abi_mirrors.py holds its structs, `ccall` names and arities to
# the header; the executable mirror of the same calls is advancedmh.jl_amd/mhx (Python/ctypes).
module AdvancedMHHIP

using AdvancedMH, AbstractMCMC, Distributions, LinearAlgebra, Random
import LogDensityProblems
import MCMCChains

include("MHXTrace.jl")
using .MHXTrace: trace_logdensity, TraceError

const libmhx = get(ENV, "MHX_LIB", joinpath(@__DIR__, "..", "libmhx.so"))

# --- status handling (nothing throws across the ABI; we re-raise on the Julia side) ------------
const MHX_EINVAL, MHX_ENOTPD = Cint(-1), Cint(-5)
function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:mhx_last_error, libmhx), Cstring, ()))
    rc == MHX_EINVAL && throw(ArgumentError(msg))
    rc == MHX_ENOTPD && throw(LinearAlgebra.PosDefException(0))
    error("libmhx error $rc: $msg")
end

# --- plain C structs of include/mhx.h (scalars travel as double; real buffers are void* of the context's dtype) ---
struct Schedule
    n_samples::Int32; discard_initial::Int32; thinning::Int32; num_warmup::Int32
end
struct RwmhCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64
    proposal_kind::Int32; proposal_scale::Cdouble; proposal_vec::Ptr{Cvoid}; flags::Int32
    proposal_mean::Ptr{Cvoid}; reduce_lanes::Int32
end
struct EmceeCfg
    dim::Int32; nwalkers::Int32; seed::UInt64; ensemble_id::UInt64; stretch::Cdouble; flags::Int32; reduce_lanes::Int32
    init_kind::Int32; init_scale::Cdouble; init_vec::Ptr{Cvoid}; init_mean::Ptr{Cvoid}
    n_ensembles::Int32
end
struct MalaCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64; sigma2::Cdouble; flags::Int32; reduce_lanes::Int32
end
struct RamCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64
    alpha::Cdouble; gamma::Cdouble; eig_lo::Cdouble; eig_hi::Cdouble; flags::Int32
end
struct Stats
    transitions::UInt64; accepted::UInt64; kernel_ms::Cdouble; wall_ms::Cdouble
    kernel_variant::Int32; launches::Int32; reduce_lanes::Int32; dtype::Int32; normal_gen::Int32; factor_band::Int32
    tainted::Int32; reserved_::Int32
end
struct DiagCfg
    max_lag::Int32; ess_chains::Int32; split::Int32
end
const MHX_FLAG_STATIC_PROPOSAL = Int32(4)
const MHX_FLAG_EMCEE_SEQUENTIAL = Int32(8)
const MHX_FLAG_ZIGGURAT = Int32(16)
const MHX_FLAG_RAM_DEFERRED = Int32(64)

"""
    LangevinProposal(σ²)

Stands for the reference's `g -> MvNormal((σ² / 2) .* g, σ² * I)` (test/runtests.jl:291): `MALA(LangevinProposal(σ²))`
dispatches to the device; a general closure keeps running on the CPU path.
"""
struct LangevinProposal; sigma2::Float64; end
(p::LangevinProposal)(g) = MvNormal((p.sigma2 / 2) .* g, p.sigma2 * I)

# --- the ensemble tag AbstractMCMC dispatches on -------------------------------------------------
"""
    MCMCHIP(; device = 0, devices = nothing, first_chain = 0, T = Float64, sequential_ensemble = false, ziggurat = false,
            ram_deferred_factor = false)

Run all chains of `sample(model, sampler, MCMCHIP(), N, nchains)` on one MI355X -- or, with `devices = 0:7`, on several from THIS
process: what `sample(model, spl, MCMCThreads(), N, nchains)` is on the CPU (README.md:135-148, one task per chain) with one host
thread per GPU behind the C ABI (`mhx_group_*`).  Chains carry global ids in their RNG counters, so the chains of a group (or of
several processes that each take a shard via `first_chain`) are the unsharded run bit for bit; a group's `stats` (see
`return_stats`) are over ALL its chains.  `T` is the arithmetic of the engine: `Float64` (what AdvancedMH.jl computes in) or
`Float32`.  `sequential_ensemble = true` runs `Ensemble` with the reference's own Gauss-Seidel sweep (src/emcee.jl:39-58) instead
of the parallel half-split.  `ziggurat = true` (Float64, RWMH with an isotropic / diagonal proposal, any log-density): standard
normals by the engine's table ziggurat instead of Box-Muller (MHX_FLAG_ZIGGURAT -- what Julia's own `randn` is; a third faster).
`ram_deferred_factor = true` (RobustAdaptiveMetropolis, dim <= 256): MHX_FLAG_RAM_DEFERRED -- up to 8 rank-one updates of `S` stay
pending as O(dim) triples and are folded in one pass; the same chain in exact arithmetic, its own rounding, a third faster at d = 200.
"""
Base.@kwdef struct MCMCHIP <: AbstractMCMC.AbstractMCMCEnsemble
    device::Int = 0
    devices::Union{Nothing,AbstractVector{<:Integer}} = nothing
    first_chain::Int = 0
    T::DataType = Float64
    sequential_ensemble::Bool = false
    ziggurat::Bool = false
    ram_deferred_factor::Bool = false
end
dtype_code(::Type{Float32}) = Cint(0)
dtype_code(::Type{Float64}) = Cint(1)

# --- ONE context per (process, device, dtype): creating one costs a stream, two events and -- with the first specialised kernel --
# the JIT cache's disk reads; `sample` used to create and destroy one per call.  A handle is not thread-safe: a lock per context
# serialises `sample` calls that share it (distinct devices run concurrently).
const CONTEXTS = Dict{Tuple{Int,DataType},Ptr{Cvoid}}()
const CONTEXT_LOCKS = Dict{Tuple{Int,DataType},ReentrantLock}()
const GROUPS = Dict{Tuple{Vector{Int32},DataType},Ptr{Cvoid}}()
const REGISTRY_LOCK = ReentrantLock()
function context(device::Integer, ::Type{T}) where {T}
    lock(REGISTRY_LOCK) do
        key = (Int(device), T)
        if !haskey(CONTEXTS, key)
            h = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:mhx_ctx_create, libmhx), Cint, (Cint, Cint, Ref{Ptr{Cvoid}}), device, dtype_code(T), h))
            CONTEXTS[key] = h[]
            CONTEXT_LOCKS[key] = ReentrantLock()
        end
        return CONTEXTS[key], CONTEXT_LOCKS[key]
    end
end
function group(devices::AbstractVector{<:Integer}, ::Type{T}) where {T}
    lock(REGISTRY_LOCK) do
        devs = Int32.(collect(devices))
        key = (devs, T)
        if !haskey(GROUPS, key)
            h = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:mhx_group_create, libmhx), Cint, (Ptr{Int32}, Int32, Cint, Ref{Ptr{Cvoid}}), devs, length(devs), dtype_code(T), h))
            GROUPS[key] = h[]
        end
        return GROUPS[key]
    end
end
const GROUP_LOCK = ReentrantLock()              # one group call at a time (a group is a handle)
function release_all()
    lock(REGISTRY_LOCK) do
        for h in values(GROUPS); ccall((:mhx_group_destroy, libmhx), Cint, (Ptr{Cvoid},), h); end
        for h in values(CONTEXTS); ccall((:mhx_ctx_destroy, libmhx), Cint, (Ptr{Cvoid},), h); end
        empty!(GROUPS); empty!(CONTEXTS); empty!(CONTEXT_LOCKS)
    end
end
__init__() = atexit(release_all)

"explicit engine option (kernel form / tuning: include/mhx.h, mhx_ctx_set_option) on the context of (device, T); `nothing` unsets"
function set_option(name::AbstractString, value; device::Integer = 0, T::DataType = Float64)
    ctx, _ = context(device, T)
    if value === nothing
        check(ccall((:mhx_ctx_set_option, libmhx), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cvoid}), ctx, name, C_NULL))
    else
        check(ccall((:mhx_ctx_set_option, libmhx), Cint, (Ptr{Cvoid}, Cstring, Cstring), ctx, name, string(value)))
    end
end

# --- device log-densities: the catalogue, a hand-written HIP source, or ANY Julia closure (traced: MHXTrace.jl) ------------------
abstract type DeviceLogDensity end
struct IsoGaussian <: DeviceLogDensity; dim::Int; end
struct CorrGaussian <: DeviceLogDensity; Σ::Matrix{Float64}; end
struct IIDNormal <: DeviceLogDensity; data::Vector{Float64}; end          # README.md:29-31
struct Banana <: DeviceLogDensity; dim::Int; b::Float64; end
struct Funnel <: DeviceLogDensity; dim::Int; end
struct HipSource <: DeviceLogDensity; src::String; dim::Int; data::Vector{Float64}; end   # written against mhx_real / MHX_R()

# The catalogue speaks the LogDensityProblems interface too (src/AdvancedMH.jl:76-77; the only model form the reference's
# RobustAdaptiveMetropolis has methods for, src/RobustAdaptiveMetropolis.jl:175-181): `sample(CorrGaussian(Σ), spl, N)` is then the
# reference's own CPU run of the very object `sample(CorrGaussian(Σ), spl, MCMCHIP(), N, nchains)` puts on the GPU.
LogDensityProblems.capabilities(::Type{<:DeviceLogDensity}) = LogDensityProblems.LogDensityOrder{0}()
LogDensityProblems.dimension(t::Union{IsoGaussian,Banana,Funnel,HipSource}) = t.dim
LogDensityProblems.dimension(t::CorrGaussian) = size(t.Σ, 1)
LogDensityProblems.dimension(::IIDNormal) = 2
LogDensityProblems.logdensity(t::IsoGaussian, x) = logpdf(MvNormal(zeros(t.dim), I), x)
LogDensityProblems.logdensity(t::CorrGaussian, x) = logpdf(MvNormal(zeros(size(t.Σ, 1)), t.Σ), x)
LogDensityProblems.logdensity(t::IIDNormal, θ) = θ[2] >= 0 ? sum(logpdf.(Normal(θ[1], θ[2]), t.data)) : -Inf   # README.md:29-31
LogDensityProblems.logdensity(t::Banana, x) =                             # N(0, diag(100, 1, ...)) twisted: x2 + b (x1^2 - 100)
    logpdf(Normal(0, 10), x[1]) + logpdf(Normal(0, 1), x[2] + t.b * (x[1]^2 - 100)) + sum(logpdf.(Normal(0, 1), x[3:end]))
LogDensityProblems.logdensity(t::Funnel, x) = logpdf(Normal(0, 3), x[1]) + sum(logpdf.(Normal(0, exp(x[1] / 2)), x[2:end]))
LogDensityProblems.logdensity(::HipSource, x) = throw(ArgumentError("a HipSource log-density is evaluated on the device only"))

"""
    lower(f, dim) -> HipSource

`DensityModel(f)` for a Julia closure (src/AdvancedMH.jl:52-54): `f` is run on traced numbers once per control-flow path that
depends on a parameter (README.md:31 `insupport(θ) ? ... : -Inf`, test/emcee.jl:8 `s > 0 || return -Inf`) and the recorded
arithmetic becomes the HIP source hiprtc inlines into the sampling kernels.  `sample(DensityModel(f), spl, MCMCHIP(), ...)` calls
this by itself; `dim` comes from the proposal / prior / `initial_params`.
"""
lower(f, dim::Integer) = HipSource(trace_logdensity(f, dim), Int(dim), Float64[])

packlower(::Type{T}, M) where {T} = T[M[i, j] for i in axes(M, 1) for j in 1:i]            # row-major packed lower

"""
    precision_factor(Σ) -> A = inv(chol(Σ)) with the structural zeros of a banded factor restored

Σ_ij = ρ^|i-j| has a bidiagonal A; the inversion leaves round-off ~1e-15 where the exact factor is zero.  An OFF-diagonal entry
counts as round-off iff |A_ij| <= 256 eps |A_jj| (relative to its column's scale: x_j ~ 1/A_jj at stationarity; the diagonal is never
touched), and the cleaned factor is used only if it is then banded (bandwidth <= 8: the engine detects an exactly banded factor
and skips the zeros, same bits) -- otherwise the raw inverse is kept.  The Python mirror (`mhx.precision_factor`) does the same.
"""
function precision_factor(Σ)
    A = Matrix(inv(cholesky(Symmetric(Matrix{Float64}(Σ))).L))
    d = size(A, 1)
    B = copy(A)
    bw = 0
    for j in 1:d, i in j:d
        if i != j && abs(A[i, j]) <= 256 * eps(Float64) * abs(A[j, j])
            B[i, j] = 0.0
        elseif B[i, j] != 0.0
            bw = max(bw, i - j)
        end
    end
    return LowerTriangular(d > 1 && bw <= min(8, d - 2) ? B : A)
end

function target(::Type{T}, ctx::Ptr{Cvoid}, t::DeviceLogDensity) where {T}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if t isa HipSource
        data = T.(t.data)
        GC.@preserve data check(ccall((:mhx_target_from_hip_source, libmhx), Cint,
            (Ptr{Cvoid}, Cstring, Cint, Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}),
            ctx, t.src, t.dim, data, length(data), h))
        return h[], t.dim
    end
    kind, dim, p = t isa IsoGaussian ? (0, t.dim, T[]) :
                   t isa CorrGaussian ? (1, size(t.Σ, 1), packlower(T, precision_factor(t.Σ))) :
                   t isa IIDNormal ? (2, 2, T.(t.data)) :
                   t isa Banana ? (3, t.dim, T[t.b]) : (4, t.dim, T[])
    GC.@preserve p check(ccall((:mhx_target_builtin, libmhx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx, kind, dim, p, length(p), h))
    return h[], dim
end

# (Mv)Normal -> (kind, scale, vec); a non-zero mean is passed separately (drifting walk, Hastings ratio on the device)
function proposal_spec(::Type{T}, d::MvNormal) where {T}
    Σ = cov(d)
    if Σ ≈ Σ[1, 1] * I
        return Int32(0), sqrt(Σ[1, 1]), T[]
    elseif isdiag(Σ)
        return Int32(1), 1.0, T.(sqrt.(diag(Σ)))
    else
        return Int32(2), 1.0, packlower(T, cholesky(Symmetric(Matrix(Σ))).L)
    end
end
proposal_spec(::Type{T}, d::Normal) where {T} = (Int32(1), 1.0, T[std(d)])
proposal_spec(::Type{T}, ds::AbstractVector{<:Normal}) where {T} = (Int32(1), 1.0, T[std(d) for d in ds])
proposal_mean(d::MvNormal) = mean(d)
proposal_mean(d::Normal) = [mean(d)]
proposal_mean(ds::AbstractVector{<:Normal}) = [mean(d) for d in ds]
ptr_or_null(v::Vector) = isempty(v) ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(v))

# the number of parameters of a closure's model, from what the call carries (the reference never needs it: a closure takes any vector)
function model_dim(sampler, initial_params)
    initial_params isa AbstractVector{<:Real} && return length(initial_params)
    initial_params isa AbstractMatrix && return size(initial_params, 1)
    if sampler isa AdvancedMH.MetropolisHastings && sampler.proposal isa Union{AdvancedMH.RandomWalkProposal,AdvancedMH.StaticProposal}
        return length(sampler.proposal.proposal)             # MvNormal: its dimension; a vector of Normals: their number; Normal: 1
    elseif sampler isa AdvancedMH.Ensemble
        return length(sampler.proposal.proposal)             # the prior the initial walkers are drawn from
    end
    throw(ArgumentError("DensityModel(f) on MCMCHIP(): pass `initial_params` so that the closure's dimension is known"))
end

# --- the state of a run: what `(sample, state) = step(...)` hands on, for ALL chains at once --------------------------------------
"""
    HIPState

The complete state of a finished `sample(...; return_state = true)` call -- every chain's position, cached log-density,
accept bookkeeping, RAM factors and the RNG step counter, as ONE opaque blob (`mhx_run_save_state`) -- plus the positions in the
clear.  Hand it back as `initial_state = st` (upstream's keyword) to continue the SAME chains bit for bit: the engine's streams are
counter-based, so a resumed run is the uninterrupted one (src/mh-core.jl:92-117; src/RobustAdaptiveMetropolis.jl:99-114).
`AbstractMCMC.getparams(st)` returns the positions (dim x nchains, the `initial_params` shape); `setparams!!(st, params)` returns a
state whose chains restart from `params` with their log-density re-evaluated (`mhx_run_set_state`), everything else -- factors,
counters, streams -- kept: src/AdvancedMH.jl:146-157, src/RobustAdaptiveMetropolis.jl:116-121.
"""
struct HIPState{T}
    blob::Vector{UInt8}
    params::Matrix{T}                       # (dim, nchains)
    lp::Vector{T}
    new_params::Union{Nothing,Matrix{T}}    # set by setparams!!: applied after the blob is loaded
end
AbstractMCMC.getparams(st::HIPState) = st.new_params === nothing ? st.params : st.new_params
AbstractMCMC.getparams(::AbstractMCMC.AbstractModel, st::HIPState) = AbstractMCMC.getparams(st)
AbstractMCMC.setparams!!(st::HIPState{T}, params) where {T} = HIPState{T}(st.blob, st.params, st.lp, Matrix{T}(params))
AbstractMCMC.setparams!!(::AbstractMCMC.AbstractModel, st::HIPState, params) = AbstractMCMC.setparams!!(st, params)

function save_state(::Type{T}, run::Ptr{Cvoid}, n::Integer, d::Integer) where {T}
    nb = Ref{Csize_t}(0)
    check(ccall((:mhx_run_state_size, libmhx), Cint, (Ptr{Cvoid}, Ref{Csize_t}), run, nb))
    blob = Vector{UInt8}(undef, nb[])
    GC.@preserve blob check(ccall((:mhx_run_save_state, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), run, blob, nb[]))
    x = Matrix{T}(undef, n, d); lp = Vector{T}(undef, n)          # host layout x[dim][nchains], chain fastest == Matrix{T}(n, d)
    GC.@preserve x lp check(ccall((:mhx_run_get_state, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt32}), run, x, lp, C_NULL))
    return HIPState{T}(blob, permutedims(x), lp, nothing)
end
function load_state(run::Ptr{Cvoid}, st::HIPState{T}) where {T}
    blob = st.blob
    GC.@preserve blob check(ccall((:mhx_run_load_state, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), run, blob, length(blob)))
    if st.new_params !== nothing
        x = Matrix{T}(permutedims(st.new_params))                 # (dim, nchains) -> (nchains, dim)
        GC.@preserve x check(ccall((:mhx_run_set_state, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), run, x))
    end
    return nothing
end

# --- the result tensor --------------------------------------------------------------------------------------------------------
# The engine fills a C-order [N][dim+1][nchains] buffer == a Julia Array{T,3}(nchains, dim+1, N).  Page-locked memory
# (mhx_host_alloc) makes the device-to-host copies run at the link rate; the block must outlive every array that views it, so the
# tensor `sample` returns OWNS it: `SampleTensor` holds the HostBlock, whose finalizer frees the block once the tensor -- and every
# view, Chains or copy-free wrapper that references the tensor -- is unreachable.  (Round 4 kept only the unsafe_wrap'ped Array,
# which does not reference the block: a use-after-free as soon as a GC ran -- ADVICE r4, high.)
mutable struct HostBlock
    ptr::Ptr{Cvoid}
    function HostBlock(p::Ptr{Cvoid})
        b = new(p)
        finalizer(release!, b)
        return b
    end
end
function release!(b::HostBlock)
    if b.ptr != C_NULL
        ccall((:mhx_host_free, libmhx), Cint, (Ptr{Cvoid},), b.ptr)
        b.ptr = C_NULL
    end
    return nothing
end
"""
    SampleTensor{T} <: AbstractArray{T,3}

`(iterations, params..lp, chains)` -- the layout of ext/AdvancedMHMCMCChainsExt.jl:96-105 -- over the buffer the engine filled: no
permuted copy, no conversion (C2's 13 GB tensor stays ONE allocation).  The tensor owns the page-locked block; keep the tensor
(not `tensor.raw`) to keep the samples.
"""
struct SampleTensor{T} <: AbstractArray{T,3}
    raw::Array{T,3}                          # (nchains, dim+1, N) over the block (or a plain Julia Array when page-locking failed)
    blk::Union{HostBlock,Nothing}
end
Base.size(a::SampleTensor) = (size(a.raw, 3), size(a.raw, 2), size(a.raw, 1))
Base.IndexStyle(::Type{<:SampleTensor}) = IndexCartesian()
Base.@propagate_inbounds Base.getindex(a::SampleTensor, i::Int, j::Int, k::Int) = a.raw[k, j, i]
Base.@propagate_inbounds Base.setindex!(a::SampleTensor, v, i::Int, j::Int, k::Int) = (a.raw[k, j, i] = v)
# `pinned = false`: a plain Julia array.  A save-all run of >= 1024 chains comes back accept-compacted (include/mhx.h:
# mhx_compact_hdr): host threads inside the library write the tensor, nothing is copied into it by DMA, and page-locking 13 GB
# (0.55 s on the box) would cost four times what the whole return takes.
function SampleTensor{T}(n::Integer, d1::Integer, N::Integer; pinned::Bool = true) where {T}
    pinned || return SampleTensor{T}(Array{T,3}(undef, n, d1, N), nothing)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mhx_host_alloc, libmhx), Cint, (Csize_t, Ref{Ptr{Cvoid}}), n * d1 * N * sizeof(T), p)
    if rc != 0 || p[] == C_NULL                                   # a host that cannot page-lock that much: a Julia Array,
        return SampleTensor{T}(Array{T,3}(undef, n, d1, N), nothing)   # which the engine registers for the duration of the call
    end
    blk = HostBlock(p[])
    return SampleTensor{T}(unsafe_wrap(Array, Ptr{T}(p[]), (Int(n), Int(d1), Int(N)); own = false), blk)
end

# --- statistics: what the reference's users read off the Chains summary (README.md:59-63) and what shards all-reduce -------------
"""
    HIPStats (a NamedTuple)

`transitions, accepted, acceptance_rate` -- totals of the call;  `kernel_ms, wall_ms, kernel_variant, reduce_lanes, normal_gen,
tainted` -- how it ran (`mhx_stats`);  `n_chains, n_samples, sum_m, sum_m2, sum_v` -- the per-parameter sums of
`mhx_run_diagnostics` over split chains (each `dim + 1` long, `lp` last), UN-normalised so that shards combine by addition;
`rhat` -- split R-hat from those sums;  `ess_bulk, ess_tail` -- rank-normalised ESS (`mhx_run_ess_bulk_tail`; of this process's
chains).  For several processes: `v = pack_stats(st); allreduce_sum!(comm, v); unpack_stats(v, st)` gives the statistics of ALL
chains (ONE all-reduce of 3(dim+1)+3 doubles over RCCL: the "global acceptance statistic and R-hat" of the design).
"""
const HIPStats = NamedTuple

function rhat_from_sums(sum_m, sum_m2, sum_v, C::Real, N::Real)
    W = sum_v ./ C
    Vm = max.((sum_m2 .- sum_m .* sum_m ./ C) ./ (C - 1), 0.0)
    varp = (N - 1) / N .* W .+ Vm
    return sqrt.(varp ./ W)
end
function stats_tuple(s::Stats, sum_m, sum_m2, sum_v, n_chains::Integer, n_samples::Integer, ess_bulk, ess_tail)
    return (transitions = Int(s.transitions), accepted = Int(s.accepted),
            acceptance_rate = s.transitions == 0 ? NaN : s.accepted / s.transitions,
            kernel_ms = s.kernel_ms, wall_ms = s.wall_ms, kernel_variant = Int(s.kernel_variant), reduce_lanes = Int(s.reduce_lanes),
            normal_gen = Int(s.normal_gen), tainted = s.tainted != 0,
            n_chains = Int(n_chains), n_samples = Int(n_samples), sum_m = sum_m, sum_m2 = sum_m2, sum_v = sum_v,
            rhat = n_chains > 1 && n_samples > 1 ? rhat_from_sums(sum_m, sum_m2, sum_v, n_chains, n_samples) : fill(NaN, length(sum_m)),
            ess_bulk = ess_bulk, ess_tail = ess_tail)
end
function run_stats(run::Ptr{Cvoid}, d::Integer, n::Integer, N::Integer)
    st = Ref{Stats}()
    check(ccall((:mhx_run_stats, libmhx), Cint, (Ptr{Cvoid}, Ref{Stats}), run, st))
    d1 = d + 1
    sum_m = zeros(d1); sum_m2 = zeros(d1); sum_v = zeros(d1)
    bulk = fill(NaN, d1); tail = fill(NaN, d1)
    split = N >= 4
    # a tensor larger than the device's free memory was streamed through two slabs and the device kept nothing: the sums are then
    # "not available" (NaN), not an error after a finished run (MHX_ESTATE = -6)
    kept = Ref{Int64}(0)
    check(ccall((:mhx_run_device_samples, libmhx), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ref{Int64}), run, C_NULL, C_NULL, kept))
    if kept[] < N
        nan = fill(NaN, d1)
        return stats_tuple(st[], nan, copy(nan), copy(nan), split ? 2n : n, split ? N ÷ 2 : N, bulk, tail)
    end
    if N >= 2
        cfg = DiagCfg(0, 0, split ? 1 : 0)
        check(ccall((:mhx_run_diagnostics, libmhx), Cint, (Ptr{Cvoid}, Ref{DiagCfg}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                    run, cfg, sum_m, sum_m2, sum_v, C_NULL))
    end
    if N >= 8
        half = N ÷ 2
        cfg = DiagCfg(max(2, half ÷ 2), 256, 1)
        params = Int32.(0:d)
        check(ccall((:mhx_run_ess_bulk_tail, libmhx), Cint, (Ptr{Cvoid}, Ref{DiagCfg}, Ptr{Int32}, Int32, Ptr{Cdouble}, Ptr{Cdouble}),
                    run, cfg, params, length(params), bulk, tail))
        bulk .= abs.(bulk); tail .= abs.(tail)                     # (negated values mark upper bounds: include/mhx.h)
    end
    return stats_tuple(st[], sum_m, sum_m2, sum_v, split ? 2n : n, split ? N ÷ 2 : N, bulk, tail)
end
pack_stats(s) = vcat(s.sum_m, s.sum_m2, s.sum_v, Float64[s.accepted, s.transitions, s.n_chains])
function unpack_stats(v::Vector{Float64}, s)
    d1 = length(s.sum_m)
    sm, sm2, sv = v[1:d1], v[(d1 + 1):(2d1)], v[(2d1 + 1):(3d1)]
    acc, tr, nch = v[3d1 + 1], v[3d1 + 2], round(Int, v[3d1 + 3])
    return merge(s, (transitions = round(Int, tr), accepted = round(Int, acc), acceptance_rate = acc / tr, n_chains = nch,
                     sum_m = sm, sum_m2 = sm2, sum_v = sv, rhat = rhat_from_sums(sm, sm2, sv, nch, s.n_samples)))
end

# --- one run on one context -----------------------------------------------------------------------------------------------------
# returns (run handle, chains of this run, initial_params in effect)
function make_run(::Type{T}, ctx::Ptr{Cvoid}, tgt::Ptr{Cvoid}, d::Integer, sampler, ens::MCMCHIP, rng, seed::UInt64,
                  first::Integer, n::Integer, initial_params) where {T}
    run = Ref{Ptr{Cvoid}}(C_NULL)
    if sampler isa AdvancedMH.MetropolisHastings
        prop = sampler.proposal
        prop isa Union{AdvancedMH.RandomWalkProposal, AdvancedMH.StaticProposal} ||
            throw(ArgumentError("the GPU path implements RandomWalkProposal and StaticProposal over (Mv)Normal only"))
        kind, scale, vec = proposal_spec(T, prop.proposal)
        μ = T.(proposal_mean(prop.proposal))
        flags = prop isa AdvancedMH.StaticProposal ? MHX_FLAG_STATIC_PROPOSAL : Int32(0)
        ens.ziggurat && (flags |= MHX_FLAG_ZIGGURAT)
        GC.@preserve vec μ begin
            cfg = RwmhCfg(d, n, seed, first, kind, scale, ptr_or_null(vec), flags,
                          all(iszero, μ) ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(μ)), 0)
            check(ccall((:mhx_rwmh_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RwmhCfg}, Ref{Ptr{Cvoid}}), ctx, tgt, cfg, run))
        end
    elseif sampler isa AdvancedMH.Ensemble
        # `sample(model, Ensemble(W, ..), MCMCThreads(), N, nchains)` runs nchains ENSEMBLES (README.md:135-148): all of them in this
        # run (mhx_emcee_cfg.n_ensembles), ids first .. first + nchains - 1, their walkers side by side in the chain axis
        nens = max(Int(n), 1)
        n = sampler.n_walkers * nens
        prior = sampler.proposal.proposal                               # what StretchProposal wraps: the prior of the initial walkers
        flags = ens.sequential_ensemble ? MHX_FLAG_EMCEE_SEQUENTIAL : Int32(0)
        if prior isa Union{MvNormal, Normal, AbstractVector{<:Normal}}  # drawn on the device (src/emcee.jl:29-34)
            kind, scale, vec = proposal_spec(T, prior)
            μ = T.(proposal_mean(prior))
            GC.@preserve vec μ begin
                cfg = EmceeCfg(d, sampler.n_walkers, seed, first, sampler.proposal.stretch_length, flags, 0,
                               kind, scale, ptr_or_null(vec), all(iszero, μ) ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(μ)), nens)
                check(ccall((:mhx_emcee_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{EmceeCfg}, Ref{Ptr{Cvoid}}), ctx, tgt, cfg, run))
            end
        else                                                            # any other Distribution: W host draws, handed over
            cfg = EmceeCfg(d, sampler.n_walkers, seed, first, sampler.proposal.stretch_length, flags, 0,
                           Int32(-1), 1.0, Ptr{Cvoid}(C_NULL), Ptr{Cvoid}(C_NULL), nens)
            check(ccall((:mhx_emcee_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{EmceeCfg}, Ref{Ptr{Cvoid}}), ctx, tgt, cfg, run))
            if initial_params === nothing
                initial_params = reduce(hcat, [prior isa AbstractVector ? map(p -> rand(rng, p), prior) : rand(rng, prior) for _ in 1:n])
            end
        end
    elseif sampler isa AdvancedMH.MALA
        prop = sampler.proposal.proposal
        prop isa LangevinProposal || throw(ArgumentError("the GPU path implements MALA(LangevinProposal(σ²)) only"))
        initial_params === nothing && error("please specify initial parameters")   # src/MALA.jl:37
        cfg = MalaCfg(d, n, seed, first, prop.sigma2, ens.ziggurat ? MHX_FLAG_ZIGGURAT : Int32(0), 0)
        check(ccall((:mhx_mala_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{MalaCfg}, Ref{Ptr{Cvoid}}), ctx, tgt, cfg, run))
    elseif sampler isa AdvancedMH.RobustAdaptiveMetropolis
        cfg = RamCfg(d, n, seed, first, sampler.α, sampler.γ, sampler.eigenvalue_lower_bound, sampler.eigenvalue_upper_bound,
                     ens.ram_deferred_factor ? MHX_FLAG_RAM_DEFERRED : Int32(0))
        check(ccall((:mhx_ram_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RamCfg}, Ref{Ptr{Cvoid}}), ctx, tgt, cfg, run))
        if sampler.S !== nothing
            if size(sampler.S) != (d, d)
                ccall((:mhx_run_destroy, libmhx), Cint, (Ptr{Cvoid},), run[])
                throw(ArgumentError("The provided `S` has the wrong dimensionality."))
            end
            S = packlower(T, LowerTriangular(sampler.S))                 # one factor for every chain (…RAM.jl:198-206)
            GC.@preserve S check(ccall((:mhx_ram_set_factor_all, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), run[], S))
        end
    else
        throw(ArgumentError("unsupported sampler $(typeof(sampler))"))
    end
    return run[], Int(n), initial_params
end

# initial AbstractMCMC.step: host layout x[dim][nchains], chain fastest == Julia Matrix{T}(n, d); columns [lo, lo + n) of a
# (dim, nchains_total) matrix for a member of a group
function initial_matrix(::Type{T}, initial_params, n::Integer, lo::Integer = 1) where {T}
    initial_params === nothing && return nothing
    initial_params isa AbstractVector{<:Real} && return repeat(T.(initial_params)', n, 1)          # one point for every chain
    return Matrix{T}(permutedims(initial_params[:, lo:(lo + n - 1)]))                             # (dim, nchains) -> (nchains, dim)
end

function finish(vals, names, chain_type, discard_initial, thinning)
    if chain_type === MCMCChains.Chains
        # same call as ext/AdvancedMHMCMCChainsExt.jl:116-120; Chains keeps (or copies) `vals`: the tensor stays alive with it
        return MCMCChains.Chains(vals, vcat(names, [:lp]), (parameters = names, internals = [:lp]);
                                 start = discard_initial + 1, thin = thinning)
    end
    return vals
end
param_symbols(param_names, d) = ismissing(param_names) ? [Symbol(:param_, i) for i in 1:d] : Symbol.(param_names)

# --- the entry point: one device ----------------------------------------------------------------------------------------------
function AbstractMCMC.sample(
    rng::Random.AbstractRNG, model::AdvancedMH.DensityModel{<:DeviceLogDensity}, sampler::AdvancedMH.MHSampler,
    ens::MCMCHIP, N::Integer, nchains::Integer;
    initial_params = nothing, initial_state = nothing, return_state = false, return_stats = false, discard_initial = nothing, thinning = 1,
    num_warmup = 0, param_names = missing, chain_type = MCMCChains.Chains, kwargs...,
)
    ens.devices === nothing || return sample_group(rng, model, sampler, ens, N, nchains; initial_params, initial_state, return_state,
                                                   return_stats, discard_initial, thinning, num_warmup, param_names, chain_type, kwargs...)
    T = ens.T
    discard_initial === nothing && (discard_initial = num_warmup)           # upstream default
    seed = rand(rng, UInt64)                                                # per-run seed from the parent rng
    ctx, ctxlock = context(ens.device, T)                                   # one context per process, not one per call
    tgt = Ptr{Cvoid}(C_NULL)
    run = Ptr{Cvoid}(C_NULL)
    lock(ctxlock)
    try                                     # every handle is released on every path (a thrown status must not leak device memory)
        tgt, d = target(T, ctx, model.logdensity)
        run, n, initial_params = make_run(T, ctx, tgt, d, sampler, ens, rng, seed, ens.first_chain, nchains, initial_params)
        if initial_state !== nothing                                       # upstream's `initial_state`: continue the same chains
            initial_state isa HIPState{T} || throw(ArgumentError("initial_state must be the HIPState{$T} a `return_state = true` call returned"))
            load_state(run, initial_state)
        else
            x0 = initial_matrix(T, initial_params, n)
            GC.@preserve x0 check(ccall((:mhx_run_init, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), run, x0 === nothing ? C_NULL : pointer(x0)))
        end
        ram = sampler isa AdvancedMH.RobustAdaptiveMetropolis
        watch = ram && haskey(kwargs, :watch_chains) ? Int32.(collect(kwargs[:watch_chains]) .- 1) : Int32[]
        if !isempty(watch)                                                   # state.S of these chains after every saved step
            haskey(kwargs, :sampler_stats) || throw(ArgumentError("watch_chains needs `sampler_stats = Ref{Any}()` to hand the factors back"))
            check(ccall((:mhx_ram_watch_factors, libmhx), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int32), run, watch, length(watch)))
        end
        sched = Schedule(N, discard_initial, thinning, num_warmup)
        # ONE call: the schedule runs while finished slabs of samples stream into the tensor on a second HIP stream
        tensor = SampleTensor{T}(n, d + 1, N; pinned = !(thinning == 1 && n >= 1024))
        GC.@preserve tensor check(ccall((:mhx_run_sample_to_host, libmhx), Cint, (Ptr{Cvoid}, Ref{Schedule}, Ptr{Cvoid}, Ptr{UInt8}, Int32),
                                        run, sched, pointer(tensor.raw), C_NULL, 0))
        rst = Ref{Stats}()
        check(ccall((:mhx_run_stats, libmhx), Cint, (Ptr{Cvoid}, Ref{Stats}), run, rst))
        rst[].tainted != 0 &&                      # whether or not statistics were asked for (include/mhx.h: mhx_stats.tainted)
            error("the run's context carries a probe option of the tools build (libmhx_tools.so): its chains may be invalid")
        state = return_state ? save_state(T, run, n, d) : nothing
        stats = return_stats ? run_stats(run, d, n, N) : nothing
        if ram && haskey(kwargs, :sampler_stats)
            # what a callback reads off `state` after every saved step (test/RobustAdaptiveMetropolis.jl:11-28): logα (N x n), η (N)
            logα = Matrix{T}(undef, n, N); η = Vector{Float64}(undef, N)
            check(ccall((:mhx_ram_get_step_stats, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Ptr{Int64}), run, logα, η, size(logα, 2), C_NULL))
            Ss = nothing
            if !isempty(watch)                                               # (tri, watched, N): packed lower factors, row-major
                tri = d * (d + 1) ÷ 2
                Sp = Array{T,3}(undef, tri, length(watch), N)
                check(ccall((:mhx_ram_get_watched_factors, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int32}),
                            run, Sp, N, C_NULL, C_NULL))
                unpack(p) = LowerTriangular([i >= j ? p[i * (i - 1) ÷ 2 + j] : zero(T) for i in 1:d, j in 1:d])
                Ss = [unpack(view(Sp, :, w, i)) for i in 1:N, w in 1:length(watch)]   # Ss[i, w] == state.S of chain watch[w] after saved step i
            end
            kwargs[:sampler_stats][] = (logα = permutedims(logα), η = η, S = Ss)
        end
        # Float64: the tensor itself (it owns its block).  Float32 results are widened ONCE into a Julia array; the block goes now.
        vals = T === Float64 ? tensor : (w = Array{Float64,3}(tensor); tensor.blk === nothing || release!(tensor.blk); w)
        out = finish(vals, param_symbols(param_names, d), chain_type, discard_initial, thinning)
        return return_state && return_stats ? (out, state, stats) : return_state ? (out, state) : return_stats ? (out, stats) : out
    finally
        run != C_NULL && ccall((:mhx_run_destroy, libmhx), Cint, (Ptr{Cvoid},), run)
        tgt != C_NULL && ccall((:mhx_target_destroy, libmhx), Cint, (Ptr{Cvoid},), tgt)
        unlock(ctxlock)
    end
end

# --- the entry point: several devices from this process (mhx_group_*) -------------------------------------------------------------
function sample_group(rng, model, sampler, ens::MCMCHIP, N::Integer, nchains::Integer;
                      initial_params = nothing, initial_state = nothing, return_state = false, return_stats = false, discard_initial = nothing,
                      thinning = 1, num_warmup = 0, param_names = missing, chain_type = MCMCChains.Chains, kwargs...)
    (initial_state === nothing && !return_state) ||
        throw(ArgumentError("MCMCHIP(devices = ...): resume / return_state per device -- run one `sample` per device with `first_chain`"))
    T = ens.T
    discard_initial === nothing && (discard_initial = num_warmup)
    seed = rand(rng, UInt64)
    g = group(ens.devices, T)
    m = length(ens.devices)
    tgts = Ptr{Cvoid}[]; runs = Ptr{Cvoid}[]; counts = Int[]
    lock(GROUP_LOCK)
    try
        d = 0
        lo = 1
        inits = Any[]
        for i in 0:(m - 1)
            ctx = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:mhx_group_ctx, libmhx), Cint, (Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), g, i, ctx))
            tgt, d = target(T, ctx[], model.logdensity)
            push!(tgts, tgt)
            first = Ref{UInt64}(0); cnt = Ref{Int32}(0)
            units = sampler isa AdvancedMH.Ensemble ? max(nchains, m) : nchains      # ensembles (at least one per member) or chains
            check(ccall((:mhx_group_shard, libmhx), Cint, (Ptr{Cvoid}, Int64, Int32, Ref{UInt64}, Ref{Int32}), g, units, i, first, cnt))
            # member i takes global ids first_chain + [first, first + cnt): of chains, or of whole ensembles
            id0 = ens.first_chain + Int(first[])
            run, n, ip = make_run(T, ctx[], tgt, d, sampler, ens, rng, seed, id0, Int(cnt[]), initial_params)
            push!(runs, run); push!(counts, n)
            push!(inits, sampler isa AdvancedMH.Ensemble ? initial_matrix(T, ip, n) : initial_matrix(T, ip, n, lo))
            lo += n
        end
        check(ccall((:mhx_group_attach, libmhx), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), g, runs))
        if all(x -> x === nothing, inits)
            check(ccall((:mhx_group_init, libmhx), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), g, C_NULL))
        else
            ptrs = Ptr{Cvoid}[x === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(x)) for x in inits]
            GC.@preserve inits check(ccall((:mhx_group_init, libmhx), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), g, ptrs))
        end
        sched = Schedule(N, discard_initial, thinning, num_warmup)
        tensors = [SampleTensor{T}(n, d + 1, N; pinned = !(thinning == 1 && n >= 1024)) for n in counts]   # one block per member
        outs = Ptr{Cvoid}[Ptr{Cvoid}(pointer(t.raw)) for t in tensors]
        GC.@preserve tensors check(ccall((:mhx_group_sample_to_host, libmhx), Cint, (Ptr{Cvoid}, Ref{Schedule}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{UInt8}}, Int32),
                                         g, sched, outs, C_NULL, 0))
        stats = nothing
        st = Ref{Stats}()
        check(ccall((:mhx_group_stats, libmhx), Cint, (Ptr{Cvoid}, Ref{Stats}), g, st))
        st[].tainted != 0 && error("a member context carries a probe option of the tools build: the chains may be invalid")
        if return_stats                                                    # over ALL chains: the members' sums added on the host
            d1 = d + 1
            sum_m = zeros(d1); sum_m2 = zeros(d1); sum_v = zeros(d1); nch = Ref{Int64}(0)
            bulk = fill(NaN, d1); tail = fill(NaN, d1)
            split = N >= 4
            if N >= 2
                cfg = DiagCfg(0, 0, split ? 1 : 0)
                check(ccall((:mhx_group_diagnostics, libmhx), Cint,
                            (Ptr{Cvoid}, Ref{DiagCfg}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Int64}),
                            g, cfg, sum_m, sum_m2, sum_v, C_NULL, nch))
            end
            if N >= 8
                cfg = DiagCfg(max(2, N ÷ 4), 256, 1)
                params = Int32.(0:d)
                check(ccall((:mhx_group_ess_bulk_tail, libmhx), Cint, (Ptr{Cvoid}, Ref{DiagCfg}, Ptr{Int32}, Int32, Ptr{Cdouble}, Ptr{Cdouble}),
                            g, cfg, params, length(params), bulk, tail))
                bulk .= abs.(bulk); tail .= abs.(tail)
            end
            stats = stats_tuple(st[], sum_m, sum_m2, sum_v, Int(nch[]), split ? N ÷ 2 : N, bulk, tail)
        end
        names = param_symbols(param_names, d)
        if chain_type === :shards                                           # the members' tensors as they are: no copy at all
            out = tensors
        else                                                                # one (N, dim+1, nchains) array: chains in global-id order
            vals = Array{Float64,3}(undef, N, d + 1, sum(counts))
            lo = 1
            for t in tensors
                vals[:, :, lo:(lo + size(t, 3) - 1)] .= t
                lo += size(t, 3)
                t.blk === nothing || release!(t.blk)                        # copied: the block goes now
            end
            out = finish(vals, names, chain_type, discard_initial, thinning)
        end
        return return_stats ? (out, stats) : out
    finally
        for r in runs; ccall((:mhx_run_destroy, libmhx), Cint, (Ptr{Cvoid},), r); end
        for t in tgts; ccall((:mhx_target_destroy, libmhx), Cint, (Ptr{Cvoid},), t); end
        unlock(GROUP_LOCK)
    end
end

# --- DensityModel(f) with a Julia closure: traced, lowered, then the method above ------------------------------------------------------
function AbstractMCMC.sample(rng::Random.AbstractRNG, model::AdvancedMH.DensityModel, sampler::AdvancedMH.MHSampler,
                             ens::MCMCHIP, N::Integer, nchains::Integer; initial_params = nothing, kwargs...)
    dim = model_dim(sampler, initial_params)
    lowered = AdvancedMH.DensityModel(lower(model.logdensity, dim))
    return AbstractMCMC.sample(rng, lowered, sampler, ens, N, nchains; initial_params, kwargs...)
end

# --- the LogDensityProblems form (src/AdvancedMH.jl:56,76-77; README.md:75-90; the only form RobustAdaptiveMetropolis takes:
# src/RobustAdaptiveMetropolis.jl:175-181,216-222,247-253, test/RobustAdaptiveMetropolis.jl:30-56).  AbstractMCMC wraps an object
# that implements the interface into `LogDensityModel` before dispatching on the ensemble, so `sample(ℓ, spl, MCMCHIP(), N, nchains)`
# lands here too.  The dimension comes from the problem itself; a catalogue target goes to its built-in kernel, anything else is
# traced through `LogDensityProblems.logdensity(ℓ, θ::Vector{Traced})` -- the same lowering as a closure's.  (MALA needs a gradient on
# the device: the traced source of this round's Julia tracer has none, so the engine refuses as src/MALA.jl:42-52 does.)
function AbstractMCMC.sample(rng::Random.AbstractRNG, model::AbstractMCMC.LogDensityModel, sampler::AdvancedMH.MHSampler,
                             ens::MCMCHIP, N::Integer, nchains::Integer; kwargs...)
    ℓ = model.logdensity
    dev = ℓ isa DeviceLogDensity ? ℓ : lower(θ -> LogDensityProblems.logdensity(ℓ, θ), LogDensityProblems.dimension(ℓ))
    return AbstractMCMC.sample(rng, AdvancedMH.DensityModel(dev), sampler, ens, N, nchains; kwargs...)
end

# convenience: default rng, and the ensemble samplers' nchains-free form
AbstractMCMC.sample(model::Union{AdvancedMH.DensityModel,AbstractMCMC.LogDensityModel}, sampler::AdvancedMH.MHSampler, ens::MCMCHIP, N::Integer,
                    nchains::Integer = 1; kwargs...) =
    AbstractMCMC.sample(Random.default_rng(), model, sampler, ens, N, nchains; kwargs...)

# --- collectives of a sharded run (RCCL over xGMI behind the C ABI; one Julia process per GPU, e.g. Distributed.jl) -----
"""
    unique_id() -> Vector{UInt8}     # 128 bytes, made on ONE process and sent to the others (Distributed.jl, MPI, a file)
    comm = comm_init(rank, world, id; device = rank, T = Float64, timeout = 300.0)
    chain, st = sample(model, spl, MCMCHIP(device = rank, first_chain = rank * n), N, n; return_stats = true)
    v = pack_stats(st); allreduce_sum!(comm, v); all = unpack_stats(v, st)     # acceptance rate and R-hat of ALL chains
"""
function unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:mhx_comm_unique_id, libmhx), Cint, (Ptr{Cvoid},), id))
    return id
end
function comm_init(rank::Integer, world::Integer, id::Vector{UInt8}; device::Integer = rank, T::DataType = Float64, timeout::Real = 300.0)
    ctx, _ = context(device, T)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    # ncclCommInitRank returns only when every rank has called it: a deadline turns a missing rank into an error that names it
    check(ccall((:mhx_comm_init_timed, libmhx), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), ctx, rank, world, id, timeout, h))
    check(ccall((:mhx_comm_set_timeout, libmhx), Cint, (Ptr{Cvoid}, Cdouble), h[], timeout))
    return h[]
end
allreduce_sum!(comm::Ptr{Cvoid}, v::Vector{Float64}) =
    (check(ccall((:mhx_comm_allreduce_sum, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t), comm, v, length(v))); v)
comm_destroy(comm::Ptr{Cvoid}) = ccall((:mhx_comm_destroy, libmhx), Cint, (Ptr{Cvoid},), comm)

export MCMCHIP, HIPState, HIPStats, SampleTensor, LangevinProposal, IsoGaussian, CorrGaussian, IIDNormal, Banana, Funnel, HipSource, lower,
       pack_stats, unpack_stats, allreduce_sum!, unique_id, comm_init, comm_destroy
end # module
