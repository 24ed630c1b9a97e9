# AdvancedMHHIP.jl -- thin `ccall` layer over libmhx.so (include/mhx.h) that plugs the MI355X engine
# into AdvancedMH.jl through AbstractMCMC's ensemble dispatch:
#
#     chain = sample(model, RWMH(MvNormal(zeros(100), 0.0566I)), MCMCHIP(), 1_000, 65_536;
#                    chain_type = Chains, discard_initial = 1_000)
#
# One `ccall` sequence runs all chains x all steps on the GPU and returns the
# (iterations, params..., lp, chains) tensor that ext/AdvancedMHMCMCChainsExt.jl:96-118 wraps.
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

# --- plain C structs of include/mhx.h ------------------------------------------------------------
struct Schedule
    n_samples::Int32; discard_initial::Int32; thinning::Int32; num_warmup::Int32
end
struct RwmhCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64
    proposal_kind::Int32; proposal_scale::Cfloat; proposal_vec::Ptr{Cfloat}; flags::Int32
    proposal_mean::Ptr{Cfloat}; reduce_lanes::Int32
end
struct EmceeCfg
    dim::Int32; nwalkers::Int32; seed::UInt64; ensemble_id::UInt64; stretch::Cfloat; flags::Int32; reduce_lanes::Int32
end
struct MalaCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64; sigma2::Cfloat; flags::Int32
end
"""
    LangevinProposal(σ²)

Stands for the reference's `g -> MvNormal((σ² / 2) .* g, σ² * I)` (test/runtests.jl:291): `MALA(LangevinProposal(σ²))`
dispatches to the device; a general closure keeps running on the CPU path.
"""
struct LangevinProposal; sigma2::Float64; end
(p::LangevinProposal)(g) = MvNormal((p.sigma2 / 2) .* g, p.sigma2 * I)
struct RamCfg
    dim::Int32; nchains::Int32; seed::UInt64; first_chain::UInt64
    alpha::Cfloat; gamma::Cfloat; eig_lo::Cfloat; eig_hi::Cfloat; flags::Int32
end

# --- the ensemble tag AbstractMCMC dispatches on -------------------------------------------------
"""
    MCMCHIP(; device = 0, first_chain = 0)

Run all chains of `sample(model, sampler, MCMCHIP(), N, nchains)` on one MI355X.  With several
processes (one per GPU) give each its shard via `first_chain`: chains carry global ids in their
RNG counters, so the union of the shards is the unsharded run.
"""
Base.@kwdef struct MCMCHIP <: AbstractMCMC.AbstractMCMCEnsemble
    device::Int = 0
    first_chain::Int = 0
end

# --- device log-densities (DensityModel(f) cannot be lowered from a Julia closure) --------------
abstract type DeviceLogDensity end
struct IsoGaussian <: DeviceLogDensity; dim::Int; end
struct CorrGaussian <: DeviceLogDensity; Σ::Matrix{Float64}; end
struct IIDNormal <: DeviceLogDensity; data::Vector{Float32}; end          # README.md:29-31
struct Banana <: DeviceLogDensity; dim::Int; b::Float32; end
struct Funnel <: DeviceLogDensity; dim::Int; end
struct HipSource <: DeviceLogDensity; src::String; dim::Int; data::Vector{Float32}; end

packlower(M) = Float32[M[i, j] for i in axes(M, 1) for j in 1:i]            # row-major packed lower

function target(ctx::Ptr{Cvoid}, t::DeviceLogDensity)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if t isa HipSource
        GC.@preserve t check(ccall((:mhx_target_from_hip_source, libmhx), Cint,
            (Ptr{Cvoid}, Cstring, Cint, Ptr{Cfloat}, Csize_t, Ref{Ptr{Cvoid}}),
            ctx, t.src, t.dim, t.data, length(t.data), h))
        return h[], t.dim
    end
    kind, dim, p = t isa IsoGaussian ? (0, t.dim, Float32[]) :
                   t isa CorrGaussian ? (1, size(t.Σ, 1), packlower(inv(cholesky(Symmetric(t.Σ)).L))) :
                   t isa IIDNormal ? (2, 2, t.data) :
                   t isa Banana ? (3, t.dim, Float32[t.b]) : (4, t.dim, Float32[])
    GC.@preserve p check(ccall((:mhx_target_builtin, libmhx), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cfloat}, Csize_t, Ref{Ptr{Cvoid}}), ctx, kind, dim, p, length(p), h))
    return h[], dim
end

# MvNormal -> (kind, scale, vec); a non-zero mean is passed separately (drifting walk, Hastings ratio on the device)
function proposal_spec(d::MvNormal)
    Σ = cov(d)
    if Σ ≈ Σ[1, 1] * I
        return Int32(0), Float32(sqrt(Σ[1, 1])), Float32[]
    elseif isdiag(Σ)
        return Int32(1), 1.0f0, Float32.(sqrt.(diag(Σ)))
    else
        return Int32(2), 1.0f0, packlower(cholesky(Symmetric(Matrix(Σ))).L)
    end
end

# --- the one entry point -------------------------------------------------------------------------
function AbstractMCMC.sample(
    rng::Random.AbstractRNG, model::AdvancedMH.DensityModel{<:DeviceLogDensity}, sampler::AdvancedMH.MHSampler,
    ens::MCMCHIP, N::Integer, nchains::Integer;
    initial_params = nothing, discard_initial = nothing, thinning = 1, num_warmup = 0,
    param_names = missing, chain_type = MCMCChains.Chains, kwargs...,
)
    discard_initial === nothing && (discard_initial = num_warmup)           # upstream default
    seed = rand(rng, UInt64)                                                # per-run seed from the parent rng
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mhx_ctx_create, libmhx), Cint, (Cint, Ref{Ptr{Cvoid}}), ens.device, ctx))
    tgt, d = target(ctx[], model.logdensity)
    run = Ref{Ptr{Cvoid}}(C_NULL)
    n = nchains
    if sampler isa AdvancedMH.MetropolisHastings
        prop = sampler.proposal
        prop isa Union{AdvancedMH.RandomWalkProposal, AdvancedMH.StaticProposal} ||
            throw(ArgumentError("the GPU path implements RandomWalkProposal and StaticProposal over (Mv)Normal only"))
        kind, scale, vec = proposal_spec(prop.proposal)
        μ = Float32.(mean(prop.proposal))
        flags = prop isa AdvancedMH.StaticProposal ? Int32(4) : Int32(0)    # MHX_FLAG_STATIC_PROPOSAL
        GC.@preserve vec μ begin
            cfg = RwmhCfg(d, n, seed, ens.first_chain, kind, scale, pointer(vec), flags,
                          all(iszero, μ) ? Ptr{Cfloat}(C_NULL) : pointer(μ), 0)
            check(ccall((:mhx_rwmh_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RwmhCfg}, Ref{Ptr{Cvoid}}),
                        ctx[], tgt, cfg, run))
        end
    elseif sampler isa AdvancedMH.Ensemble
        n = sampler.n_walkers
        cfg = EmceeCfg(d, n, seed, ens.first_chain, Float32(sampler.proposal.stretch_length), 0, 0)
        check(ccall((:mhx_emcee_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{EmceeCfg}, Ref{Ptr{Cvoid}}),
                    ctx[], tgt, cfg, run))
        if initial_params === nothing                                       # src/emcee.jl:29-34
            initial_params = reduce(hcat, [rand(rng, sampler.proposal.proposal) for _ in 1:n])
        end
    elseif sampler isa AdvancedMH.MALA
        prop = sampler.proposal.proposal
        prop isa LangevinProposal || throw(ArgumentError("the GPU path implements MALA(LangevinProposal(σ²)) only"))
        initial_params === nothing && error("please specify initial parameters")   # src/MALA.jl:37
        cfg = MalaCfg(d, n, seed, ens.first_chain, prop.sigma2, 0)
        check(ccall((:mhx_mala_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{MalaCfg}, Ref{Ptr{Cvoid}}),
                    ctx[], tgt, cfg, run))
    elseif sampler isa AdvancedMH.RobustAdaptiveMetropolis
        cfg = RamCfg(d, n, seed, ens.first_chain, sampler.α, sampler.γ,
                     sampler.eigenvalue_lower_bound, sampler.eigenvalue_upper_bound, 0)
        check(ccall((:mhx_ram_create, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{RamCfg}, Ref{Ptr{Cvoid}}),
                    ctx[], tgt, cfg, run))
        if sampler.S !== nothing
            size(sampler.S) == (d, d) || throw(ArgumentError("The provided `S` has the wrong dimensionality."))
            S = repeat(packlower(LowerTriangular(sampler.S)), 1, n)          # [tri, n] column-major == [n][tri] in C
            GC.@preserve S check(ccall((:mhx_ram_set_factor, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cfloat}), run[], S))
        end
    else
        throw(ArgumentError("unsupported sampler $(typeof(sampler))"))
    end

    # initial AbstractMCMC.step: host layout x[dim][nchains], chain fastest == Julia Matrix{Float32}(n, d)
    init = initial_params === nothing ? Ptr{Cfloat}(C_NULL) :
           (x0 = initial_params isa AbstractVector ? repeat(Float32.(initial_params)', n, 1) :
                                                      Matrix{Float32}(permutedims(initial_params));
            x0)
    GC.@preserve init check(ccall((:mhx_run_init, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cfloat}), run[],
                                  init isa Ptr ? init : pointer(init)))
    sched = Schedule(N, discard_initial, thinning, num_warmup)
    check(ccall((:mhx_run_sample, libmhx), Cint, (Ptr{Cvoid}, Ref{Schedule}, Cint), run[], sched, 1))

    # C order [N][d+1][n] with the chain fastest == Julia Array{Float32,3}(n, d+1, N)
    raw = Array{Float32,3}(undef, n, d + 1, N)
    GC.@preserve raw check(ccall((:mhx_run_get_samples, libmhx), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{UInt8}),
                                 run[], raw, C_NULL))
    vals = Float64.(permutedims(raw, (3, 2, 1)))                             # (iterations, params..lp, chains)
    ccall((:mhx_run_destroy, libmhx), Cint, (Ptr{Cvoid},), run[])
    ccall((:mhx_target_destroy, libmhx), Cint, (Ptr{Cvoid},), tgt)
    ccall((:mhx_ctx_destroy, libmhx), Cint, (Ptr{Cvoid},), ctx[])

    names = ismissing(param_names) ? [Symbol(:param_, i) for i in 1:d] : Symbol.(param_names)
    chain_type === MCMCChains.Chains || return vals
    # same call as ext/AdvancedMHMCMCChainsExt.jl:116-120
    return MCMCChains.Chains(vals, vcat(names, [:lp]), (parameters = names, internals = [:lp]);
                             start = discard_initial + 1, thin = thinning)
end

# convenience: default rng, and the ensemble samplers' nchains-free form
AbstractMCMC.sample(model::AdvancedMH.DensityModel{<:DeviceLogDensity}, sampler::AdvancedMH.MHSampler,
                    ens::MCMCHIP, N::Integer, nchains::Integer = 1; kwargs...) =
    AbstractMCMC.sample(Random.default_rng(), model, sampler, ens, N, nchains; kwargs...)

export MCMCHIP, LangevinProposal, IsoGaussian, CorrGaussian, IIDNormal, Banana, Funnel, HipSource
end # module
