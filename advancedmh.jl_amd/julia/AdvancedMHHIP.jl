# AdvancedMHHIP.jl -- thin `ccall` layer over libmhx.so (include/mhx.h, ABI 0.4) that plugs the MI355X engine
# into AdvancedMH.jl through AbstractMCMC's ensemble dispatch:
#
#     chain = sample(model, RWMH(MvNormal(zeros(100), 0.0566I)), MCMCHIP(), 1_000, 65_536;
#                    chain_type = Chains, discard_initial = 1_000)
#
# One `ccall` sequence runs all chains x all steps on the GPU and returns the
# (iterations, params..., lp, chains) tensor that ext/AdvancedMHMCMCChainsExt.jl:96-118 wraps.
# The engine computes in Float64 by default -- the reference's arithmetic -- or in Float32 (`MCMCHIP(T = Float32)`).
#
# STATUS: written against include/mhx.h and reviewed by hand; neither this container nor the GPU
# box has a `julia` binary, so this file has never been executed (DESIGN.md section 2).  The
# executable mirror of the same calls is advancedmh.jl_amd/mhx (Python/ctypes).
module AdvancedMHHIP

using AdvancedMH, AbstractMCMC, Distributions, LinearAlgebra, Random
import MCMCChains

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
end
struct MalaCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64; sigma2::Cdouble; flags::Int32; reduce_lanes::Int32
end
struct RamCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64
    alpha::Cdouble; gamma::Cdouble; eig_lo::Cdouble; eig_hi::Cdouble; flags::Int32
end
const MHX_FLAG_STATIC_PROPOSAL = Int32(4)
const MHX_FLAG_EMCEE_SEQUENTIAL = Int32(8)
const MHX_FLAG_ZIGGURAT = Int32(16)

"""
    LangevinProposal(σ²)

Stands for the reference's `g -> MvNormal((σ² / 2) .* g, σ² * I)` (test/runtests.jl:291): `MALA(LangevinProposal(σ²))`
dispatches to the device; a general closure keeps running on the CPU path.
"""
struct LangevinProposal; sigma2::Float64; end
(p::LangevinProposal)(g) = MvNormal((p.sigma2 / 2) .* g, p.sigma2 * I)

# --- the ensemble tag AbstractMCMC dispatches on -------------------------------------------------
"""
    MCMCHIP(; device = 0, first_chain = 0, T = Float64, sequential_ensemble = false, ziggurat = false)

Run all chains of `sample(model, sampler, MCMCHIP(), N, nchains)` on one MI355X.  `T` is the arithmetic of the engine:
`Float64` (what AdvancedMH.jl computes in) or `Float32`.  With several processes (one per GPU) give each its shard via
`first_chain`: chains carry global ids in their RNG counters, so the union of the shards is the unsharded run.
`sequential_ensemble = true` runs `Ensemble` with the reference's own Gauss-Seidel sweep (src/emcee.jl:39-58) instead
of the parallel half-split.  `ziggurat = true` (Float64, RWMH with an isotropic / diagonal proposal: catalogue targets and `HipSource` log-densities): standard normals by the
engine's table ziggurat instead of Box-Muller (MHX_FLAG_ZIGGURAT -- what Julia's own `randn` is; a third faster).
"""
Base.@kwdef struct MCMCHIP <: AbstractMCMC.AbstractMCMCEnsemble
    device::Int = 0
    first_chain::Int = 0
    T::DataType = Float64
    sequential_ensemble::Bool = false
    ziggurat::Bool = false
end
dtype_code(::Type{Float32}) = Cint(0)
dtype_code(::Type{Float64}) = Cint(1)

# --- device log-densities (DensityModel(f) cannot be lowered from a Julia closure) --------------
abstract type DeviceLogDensity end
struct IsoGaussian <: DeviceLogDensity; dim::Int; end
struct CorrGaussian <: DeviceLogDensity; Σ::Matrix{Float64}; end
struct IIDNormal <: DeviceLogDensity; data::Vector{Float64}; end          # README.md:29-31
struct Banana <: DeviceLogDensity; dim::Int; b::Float64; end
struct Funnel <: DeviceLogDensity; dim::Int; end
struct HipSource <: DeviceLogDensity; src::String; dim::Int; data::Vector{Float64}; end   # written against mhx_real / MHX_R()

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

"""
    SampleTensor{T} <: AbstractArray{T,3}

The (iterations, params..lp, chains) view of the C-order `[N][dim+1][nchains]` buffer the engine filled -- `PermutedDimsArray` over
the page-locked block of `mhx_host_alloc`, no copy; released by a finalizer (`mhx_host_free`).
"""
mutable struct HostBlock{T}
    ptr::Ptr{T}
    raw::Array{T,3}                          # (nchains, dim+1, N) over `ptr`
end
function host_tensor(::Type{T}, n::Integer, d1::Integer, N::Integer) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:mhx_host_alloc, libmhx), Cint, (Csize_t, Ref{Ptr{Cvoid}}), n * d1 * N * sizeof(T), p)
    (rc != 0 || p[] == C_NULL) && return nothing                  # a host that cannot page-lock that much: the caller falls back to an Array
    blk = HostBlock{T}(Ptr{T}(p[]), unsafe_wrap(Array, Ptr{T}(p[]), (n, d1, N); own = false))
    finalizer(b -> ccall((:mhx_host_free, libmhx), Cint, (Ptr{Cvoid},), b.ptr), blk)
    return blk
end

# --- the one entry point -------------------------------------------------------------------------
function AbstractMCMC.sample(
    rng::Random.AbstractRNG, model::AdvancedMH.DensityModel{<:DeviceLogDensity}, sampler::AdvancedMH.MHSampler,
    ens::MCMCHIP, N::Integer, nchains::Integer;
    initial_params = nothing, initial_state = nothing, return_state = false, discard_initial = nothing, thinning = 1, num_warmup = 0,
    param_names = missing, chain_type = MCMCChains.Chains, kwargs...,
)
    T = ens.T
    discard_initial === nothing && (discard_initial = num_warmup)           # upstream default
    seed = rand(rng, UInt64)                                                # per-run seed from the parent rng
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    tgt = Ptr{Cvoid}(C_NULL)
    run = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mhx_ctx_create, libmhx), Cint, (Cint, Cint, Ref{Ptr{Cvoid}}), ens.device, dtype_code(T), ctx))
    try                                     # every handle is released on every path (a thrown status must not leak device memory)
        tgt, d = target(T, ctx[], model.logdensity)
        n = nchains
        if sampler isa AdvancedMH.MetropolisHastings
            prop = sampler.proposal
            prop isa Union{AdvancedMH.RandomWalkProposal, AdvancedMH.StaticProposal} ||
                throw(ArgumentError("the GPU path implements RandomWalkProposal and StaticProposal over (Mv)Normal only"))
            kind, scale, vec = proposal_spec(T, prop.proposal)
            μ = T.(proposal_mean(prop.proposal))
            flags = prop isa AdvancedMH.StaticProposal ? MHX_FLAG_STATIC_PROPOSAL : Int32(0)
            ens.ziggurat && (flags |= MHX_FLAG_ZIGGURAT)
            GC.@preserve vec μ begin
                cfg = RwmhCfg(d, n, seed, ens.first_chain, kind, scale, ptr_or_null(vec), flags,
                              all(iszero, μ) ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(μ)), 0)
                check(ccall((:mhx_rwmh_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RwmhCfg}, Ref{Ptr{Cvoid}}),
                            ctx[], tgt, cfg, run))
            end
        elseif sampler isa AdvancedMH.Ensemble
            n = sampler.n_walkers
            prior = sampler.proposal.proposal                               # what StretchProposal wraps: the prior of the initial walkers
            flags = ens.sequential_ensemble ? MHX_FLAG_EMCEE_SEQUENTIAL : Int32(0)
            if prior isa Union{MvNormal, Normal, AbstractVector{<:Normal}}  # drawn on the device (src/emcee.jl:29-34)
                kind, scale, vec = proposal_spec(T, prior)
                μ = T.(proposal_mean(prior))
                GC.@preserve vec μ begin
                    cfg = EmceeCfg(d, n, seed, ens.first_chain, sampler.proposal.stretch_length, flags, 0,
                                   kind, scale, ptr_or_null(vec), all(iszero, μ) ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(μ)))
                    check(ccall((:mhx_emcee_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{EmceeCfg}, Ref{Ptr{Cvoid}}),
                                ctx[], tgt, cfg, run))
                end
            else                                                            # any other Distribution: W host draws, handed over
                cfg = EmceeCfg(d, n, seed, ens.first_chain, sampler.proposal.stretch_length, flags, 0,
                               Int32(-1), 1.0, Ptr{Cvoid}(C_NULL), Ptr{Cvoid}(C_NULL))
                check(ccall((:mhx_emcee_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{EmceeCfg}, Ref{Ptr{Cvoid}}),
                            ctx[], tgt, cfg, run))
                if initial_params === nothing
                    initial_params = reduce(hcat, [prior isa AbstractVector ? map(p -> rand(rng, p), prior) : rand(rng, prior) for _ in 1:n])
                end
            end
        elseif sampler isa AdvancedMH.MALA
            prop = sampler.proposal.proposal
            prop isa LangevinProposal || throw(ArgumentError("the GPU path implements MALA(LangevinProposal(σ²)) only"))
            initial_params === nothing && error("please specify initial parameters")   # src/MALA.jl:37
            cfg = MalaCfg(d, n, seed, ens.first_chain, prop.sigma2, 0, 0)
            check(ccall((:mhx_mala_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{MalaCfg}, Ref{Ptr{Cvoid}}),
                        ctx[], tgt, cfg, run))
        elseif sampler isa AdvancedMH.RobustAdaptiveMetropolis
            cfg = RamCfg(d, n, seed, ens.first_chain, sampler.α, sampler.γ,
                         sampler.eigenvalue_lower_bound, sampler.eigenvalue_upper_bound, 0)
            check(ccall((:mhx_ram_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RamCfg}, Ref{Ptr{Cvoid}}),
                        ctx[], tgt, cfg, run))
            if sampler.S !== nothing
                size(sampler.S) == (d, d) || throw(ArgumentError("The provided `S` has the wrong dimensionality."))
                S = packlower(T, LowerTriangular(sampler.S))                 # one factor for every chain (…RAM.jl:198-206)
                GC.@preserve S check(ccall((:mhx_ram_set_factor_all, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), run[], S))
            end
        else
            throw(ArgumentError("unsupported sampler $(typeof(sampler))"))
        end

        # initial AbstractMCMC.step: host layout x[dim][nchains], chain fastest == Julia Matrix{T}(n, d)
        if initial_state !== nothing                                       # upstream's `initial_state`: continue the same chains
            initial_state isa HIPState{T} || throw(ArgumentError("initial_state must be the HIPState{$T} a `return_state = true` call returned"))
            load_state(run[], initial_state)
        elseif initial_params === nothing
            check(ccall((:mhx_run_init, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), run[], C_NULL))
        else
            x0 = initial_params isa AbstractVector{<:Real} ? repeat(T.(initial_params)', n, 1) :     # one point for every chain
                                                             Matrix{T}(permutedims(initial_params))   # (dim, nchains) -> (nchains, dim)
            GC.@preserve x0 check(ccall((:mhx_run_init, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), run[], x0))
        end
        watch = sampler isa AdvancedMH.RobustAdaptiveMetropolis && haskey(kwargs, :watch_chains) ? Int32.(collect(kwargs[:watch_chains]) .- 1) : Int32[]
        if !isempty(watch)                                                   # state.S of these chains after every saved step
            GC.@preserve watch check(ccall((:mhx_ram_watch_factors, libmhx), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int32), run[], watch, length(watch)))
        end
        sched = Schedule(N, discard_initial, thinning, num_warmup)
        # ONE call: the schedule runs while finished slabs of samples stream into `raw` on a second HIP stream
        # (mhx_run_sample_to_host registers the Julia array for the duration of the call; 0 = default slab size).
        # C order [N][d+1][n] with the chain fastest == Julia Array{T,3}(n, d+1, N)
        blk = host_tensor(T, n, d + 1, N)                                  # page-locked: the copies run at the link rate
        raw = blk === nothing ? Array{T,3}(undef, n, d + 1, N) : blk.raw
        GC.@preserve raw blk check(ccall((:mhx_run_sample_to_host, libmhx), Cint, (Ptr{Cvoid}, Ref{Schedule}, Ptr{Cvoid}, Ptr{UInt8}, Int32),
                                         run[], sched, raw, C_NULL, 0))
        state = return_state ? save_state(T, run[], n, d) : nothing
        if sampler isa AdvancedMH.RobustAdaptiveMetropolis && haskey(kwargs, :sampler_stats)
            # what a callback reads off `state` after every saved step (test/RobustAdaptiveMetropolis.jl:11-28): logα (N x n), η (N)
            logα = Matrix{T}(undef, n, N); η = Vector{Float64}(undef, N)
            GC.@preserve logα η check(ccall((:mhx_ram_get_step_stats, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Ptr{Int64}), run[], logα, η, size(logα, 2), C_NULL))
            Ss = nothing
            if !isempty(watch)                                               # (tri, watched, N): packed lower factors, row-major
                tri = d * (d + 1) ÷ 2
                Sp = Array{T,3}(undef, tri, length(watch), N)
                GC.@preserve Sp check(ccall((:mhx_ram_get_watched_factors, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int32}),
                                            run[], Sp, N, C_NULL, C_NULL))
                unpack(p) = LowerTriangular([i >= j ? p[i * (i - 1) ÷ 2 + j] : zero(T) for i in 1:d, j in 1:d])
                Ss = [unpack(view(Sp, :, w, i)) for i in 1:N, w in 1:length(watch)]   # Ss[i, w] == state.S of chain watch[w] after saved step i
            end
            kwargs[:sampler_stats][] = (logα = permutedims(logα), η = η, S = Ss)
        end
        # (iterations, params..lp, chains): for T == Float64 a VIEW of the buffer the engine filled (no permuted copy, no conversion:
        # C2's 13 GB tensor stays one allocation); the view keeps the block alive.  Float32 results are widened once.
        view3 = PermutedDimsArray(raw, (3, 2, 1))
        vals = T === Float64 ? view3 : Float64.(view3)
        names = ismissing(param_names) ? [Symbol(:param_, i) for i in 1:d] : Symbol.(param_names)
        if chain_type === MCMCChains.Chains
            # same call as ext/AdvancedMHMCMCChainsExt.jl:116-120 (Chains copies into its own AxisArray)
            vals = MCMCChains.Chains(vals, vcat(names, [:lp]), (parameters = names, internals = [:lp]);
                                     start = discard_initial + 1, thin = thinning)
        end
        return return_state ? (vals, state) : vals
    finally
        run[] != C_NULL && ccall((:mhx_run_destroy, libmhx), Cint, (Ptr{Cvoid},), run[])
        tgt != C_NULL && ccall((:mhx_target_destroy, libmhx), Cint, (Ptr{Cvoid},), tgt)
        ccall((:mhx_ctx_destroy, libmhx), Cint, (Ptr{Cvoid},), ctx[])
    end
end

# convenience: default rng, and the ensemble samplers' nchains-free form
AbstractMCMC.sample(model::AdvancedMH.DensityModel{<:DeviceLogDensity}, sampler::AdvancedMH.MHSampler,
                    ens::MCMCHIP, N::Integer, nchains::Integer = 1; kwargs...) =
    AbstractMCMC.sample(Random.default_rng(), model, sampler, ens, N, nchains; kwargs...)

# --- collectives of a sharded run (RCCL over xGMI behind the C ABI; one Julia process per GPU, e.g. Distributed.jl) -----
"""
    unique_id() -> Vector{UInt8}     # 128 bytes, made on ONE process and sent to the others (Distributed.jl, MPI, a file)
    comm = comm_init(ctx, rank, world, id)
    allreduce_sum!(comm, v::Vector{Float64})     # acceptance totals + the R-hat / ESS sums of mhx_run_diagnostics
"""
function unique_id()
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id check(ccall((:mhx_comm_unique_id, libmhx), Cint, (Ptr{Cvoid},), id))
    return id
end
function comm_init(ctx::Ptr{Cvoid}, rank::Integer, world::Integer, id::Vector{UInt8})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve id check(ccall((:mhx_comm_init, libmhx), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx, rank, world, id, h))
    return h[]
end
allreduce_sum!(comm::Ptr{Cvoid}, v::Vector{Float64}) =
    (GC.@preserve v check(ccall((:mhx_comm_allreduce_sum, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t), comm, v, length(v))); v)
comm_destroy(comm::Ptr{Cvoid}) = ccall((:mhx_comm_destroy, libmhx), Cint, (Ptr{Cvoid},), comm)

export MCMCHIP, HIPState, LangevinProposal, IsoGaussian, CorrGaussian, IIDNormal, Banana, Funnel, HipSource
end # module
