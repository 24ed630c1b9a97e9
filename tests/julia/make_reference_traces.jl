# make_reference_traces.jl -- run the UNMODIFIED AdvancedMH.jl on the engine's random streams and write the traces the
# parity test reads.  One command turns "parity unpinned" into a red / green test:
#
#     julia --project=/path/to/AdvancedMH.jl tests/julia/make_reference_traces.jl tests/golden/julia
#     python -m pytest tests/test_julia_reference_traces.py
#
# Every case below mirrors a case of tests/julia_cases.py (same seed, global chain ids, schedule, model, sampler).  The
# package's own `sample` runs each chain with a `PhiloxStream` (tests/julia/PhiloxStreams.jl) as its rng, so
# src/mh-core.jl:92-117, src/emcee.jl:39-102 and src/RobustAdaptiveMetropolis.jl:123-278 consume the very draws the oracle
# consumes; the test compares number by number (states to 1e-9, accept decisions wherever their margin exceeds it --
# Julia rounds `x + sigma z` and the log-density sums in separate operations where the engine's spec fuses them).
#
# Dispatch notes (the reference's own tests are the model):
#   * RWMH / Ensemble accept a `DensityModel(f)` (src/mh-core.jl:76-117, src/emcee.jl:14-24 dispatch on
#     `DensityModelOrLogDensityModel`);
#   * RobustAdaptiveMetropolis' `step` methods accept ONLY `AbstractMCMC.LogDensityModel`
#     (src/RobustAdaptiveMetropolis.jl:175-181, 216-222, 247-253), so its model is a LogDensityProblems object handed
#     straight to `sample`, which wraps it -- exactly test/RobustAdaptiveMetropolis.jl:1-9,45-55;
#   * MALA needs `logdensity_and_gradient` (src/MALA.jl:36-52,101-104): a LogDensityProblems object of order 1 with an
#     analytic gradient, the form of test/runtests.jl:334-365 (no ForwardDiff needed).
#
# STATUS: never executed (no `julia` binary in the build container); written against AdvancedMH.jl v0.8.8's public API.
# tests/test_julia_kit.py checks what can be checked without Julia: literals against the oracle, case lists in step
# with tests/julia_cases.py, and that no RobustAdaptiveMetropolis / MALA case is given a DensityModel.
using AdvancedMH, AbstractMCMC, Distributions, LinearAlgebra, LogDensityProblems, Random
include(joinpath(@__DIR__, "PhiloxStreams.jl"))
using .PhiloxStreams

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "golden", "julia")
mkpath(outdir)

# ---- a minimal .npy writer (format 1.0; Julia arrays are column-major: fortran_order = True) -----------------------
function write_npy(path::AbstractString, A::Array{T}) where {T<:Union{Float64,UInt8}}
    descr = T === Float64 ? "<f8" : "|u1"
    shape = join(size(A), ", ") * (ndims(A) == 1 ? "," : "")
    hdr = "{'descr': '$descr', 'fortran_order': True, 'shape': ($shape), }"
    pad = (64 - (10 + length(hdr) + 1) % 64) % 64
    hdr = hdr * " "^pad * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93]); write(io, "NUMPY"); write(io, UInt8[0x01, 0x00])
        write(io, UInt16(length(hdr))); write(io, hdr); write(io, A)
    end
end

ar1(d, rho) = [rho^abs(i - j) for i in 1:d, j in 1:d]

# ---- the catalogue log-densities as plain Julia functions (DensityModel(f), src/AdvancedMH.jl:52-54) ------------------
iso_gauss(d) = x -> logpdf(MvNormal(zeros(d), I), x)
corr_gauss(Σ) = x -> logpdf(MvNormal(zeros(size(Σ, 1)), Symmetric(Σ)), x)
function funnel(d)                          # x1 ~ N(0, 9), x_k ~ N(0, exp(x1))
    return x -> logpdf(Normal(0, 3), x[1]) + sum(logpdf.(Normal(0, exp(x[1] / 2)), x[2:end]))
end
function banana(d, b)                       # N(0, diag(100, 1, ...)) with x2 <- x2 + b (x1^2 - 100)
    return x -> logpdf(Normal(0, 10), x[1]) + logpdf(Normal(0, 1), x[2] + b * (x[1]^2 - 100)) + sum(logpdf.(Normal(0, 1), x[3:end]))
end

# ---- the same Gaussians as LogDensityProblems objects (what RobustAdaptiveMetropolis and MALA dispatch on) -----------------
struct GaussianLDP{A,B}
    Σ::A            # covariance: the log-density is logpdf(MvNormal(0, Σ), x), as test/RobustAdaptiveMetropolis.jl:6-9
    P::B            # inv(Σ), for the gradient -P x
end
GaussianLDP(Σ::AbstractMatrix) = GaussianLDP(Matrix(Σ), inv(Symmetric(Matrix(Σ))))
LogDensityProblems.dimension(m::GaussianLDP) = size(m.Σ, 1)
LogDensityProblems.capabilities(::Type{<:GaussianLDP}) = LogDensityProblems.LogDensityOrder{1}()
LogDensityProblems.logdensity(m::GaussianLDP, x) = logpdf(MvNormal(zeros(size(m.Σ, 1)), Symmetric(m.Σ)), x)
LogDensityProblems.logdensity_and_gradient(m::GaussianLDP, x) = (LogDensityProblems.logdensity(m, x), -(m.P * x))

# one chain of sampler `spl`, global id `id`: (N, d+1) samples and N accept flags
function run_chain(model, spl, N, seed, id, d; initial_params = nothing, ziggurat = false, kw...)
    rng = PhiloxStream(seed, id; dim = d, initial_draw = initial_params === nothing, ziggurat = ziggurat)
    ts = sample(rng, model, spl, N; chain_type = Any, progress = false, initial_params = initial_params, kw...)
    S = Matrix{Float64}(undef, N, d + 1)
    acc = Vector{UInt8}(undef, N)
    for (i, t) in enumerate(ts)
        S[i, 1:d] .= t.params
        S[i, d + 1] = t.lp
        acc[i] = t.accepted ? 0x01 : 0x00
    end
    return S, acc
end

function run_chains(name, model, spl, N, seed, first_chain, C, d; kw...)
    S = Array{Float64,3}(undef, N, d + 1, C)
    A = Matrix{UInt8}(undef, N, C)
    for c in 1:C
        S[:, :, c], A[:, c] = run_chain(model, spl, N, seed, first_chain + c - 1, d; kw...)
    end
    write_npy(joinpath(outdir, name * "_samples.npy"), S)
    write_npy(joinpath(outdir, name * "_accepted.npy"), A)
    println("wrote ", name, ": ", size(S))
end

# ---- random-walk Metropolis-Hastings (src/mh-core.jl:76-117) -----------------------------------------------------------
run_chains("rwmh_iso", DensityModel(iso_gauss(5)), RWMH(MvNormal(zeros(5), 0.5^2 * I)), 32, 11, 3, 8, 5)
let d = 4, Σp = 0.3 * ar1(4, 0.5)
    run_chains("rwmh_dense_corr", DensityModel(corr_gauss(ar1(d, 0.8))), RWMH(MvNormal(zeros(d), Symmetric(Σp))), 20, 12, 0, 6, d;
               discard_initial = 3, thinning = 2)
end
run_chains("rwmh_funnel", DensityModel(funnel(6)), RWMH(MvNormal(zeros(6), 0.4^2 * I)), 24, 13, 100, 7, 6)
run_chains("rwmh_banana", DensityModel(banana(5, 0.03)), RWMH([Normal(0, 2.0), Normal(0, 0.5), Normal(0, 1.0), Normal(0, 1.0), Normal(0, 1.0)]),
           24, 14, 0, 7, 5)
run_chains("rwmh_given_start", DensityModel(iso_gauss(3)), RWMH(MvNormal(zeros(3), 0.7^2 * I)), 40, 9, 0, 4, 3;
           initial_params = [0.5, -1.0, 0.25])
# the engine's ziggurat normals (MHX_FLAG_ZIGGURAT): 64 chains x 48 transitions x 8 normals = 2.5e4 draws, ~100 through the slow paths
run_chains("rwmh_iso_ziggurat", DensityModel(iso_gauss(8)), RWMH(MvNormal(zeros(8), 0.6^2 * I)), 48, 15, 7, 64, 8; ziggurat = true)

# README.md:25-40 AS WRITTEN -- a closure over data with a branch on a parameter (what MCMCHIP() traces and JIT-lowers; here the
# reference runs it itself): the first 30 points of tests/golden/c1_normal_data.npy
let data = parse.(Float64, split(strip(read(joinpath(@__DIR__, "..", "golden", "c1_normal_data_30.txt"), String))))
    insupport(θ) = θ[2] >= 0
    dist(θ) = Normal(θ[1], θ[2])
    density(θ) = insupport(θ) ? sum(logpdf.(dist(θ), data)) : -Inf
    run_chains("rwmh_readme", DensityModel(density), RWMH(MvNormal(zeros(2), I)), 64, 18, 0, 4, 2; initial_params = [0.0, 1.0])
end
# a USER log-density at d = 100 on the ziggurat normals (on the device: the register kernel's ziggurat form, one lane per chain)
let d = 100, μ = [(k - 50.0) / 64.0 for k in 0:99], σ = [0.5 + (k % 8) / 8.0 for k in 0:99]
    shifted(θ) = -0.5 * sum(abs2, (θ .- μ) ./ σ)
    run_chains("rwmh_user_ziggurat", DensityModel(shifted), RWMH(MvNormal(zeros(d), 0.25^2 * I)), 24, 19, 5, 8, d; ziggurat = true)
end

# a drifting random walk (non-zero proposal mean: the Hastings ratio of src/proposal.jl:58-64,190-192 is not zero) and an independence
# sampler (StaticMH: src/proposal.jl:9-11,66-83); Distributions' logpdf rounds differently from the engine's ratio, the accept-margin
# logic of tests/test_julia_reference_traces.py absorbs that
let μ = [0.3, -0.2, 0.1, 0.25]
    run_chains("rwmh_drift", DensityModel(iso_gauss(4)), RWMH(MvNormal(μ, 0.6^2 * I)), 40, 16, 0, 6, 4; initial_params = zeros(4))
    run_chains("rwmh_static", DensityModel(corr_gauss(ar1(4, 0.5))), StaticMH(MvNormal(μ, 1.2^2 * I)), 40, 17, 0, 6, 4; initial_params = zeros(4))
end

# ---- RobustAdaptiveMetropolis (src/RobustAdaptiveMetropolis.jl:123-278) ------------------------------------------------
# the model is a LogDensityProblems object, NOT a DensityModel (see the dispatch notes above)
let d = 4
    run_chains("ram", GaussianLDP(ar1(d, 0.7)), RobustAdaptiveMetropolis(), 24, 31, 2, 6, d;
               initial_params = zeros(d), num_warmup = 16, discard_initial = 0)
    run_chains("ram_random_start", GaussianLDP(ar1(d, 0.7)), RobustAdaptiveMetropolis(), 30, 33, 0, 5, d;
               num_warmup = 30, discard_initial = 0)
end
let Σ = [10.0 5.0; 5.0 10.0]
    spl = RobustAdaptiveMetropolis(; γ = 0.51, eigenvalue_lower_bound = 0.9, eigenvalue_upper_bound = 1.1)
    run_chains("ram_bounds", GaussianLDP(Σ), spl, 40, 32, 0, 5, 2; initial_params = zeros(2), num_warmup = 40, discard_initial = 0)
end

# ---- MALA with the standard Langevin proposal (src/MALA.jl:54-93; the form of test/runtests.jl:352) ----------------------
let σ² = 0.3
    mala = MALA(g -> MvNormal((σ² / 2) .* g, σ² * I))
    run_chains("mala_iso", GaussianLDP(Matrix(1.0I, 5, 5)), mala, 32, 41, 1, 6, 5; initial_params = fill(0.25, 5))
    run_chains("mala_corr", GaussianLDP(ar1(4, 0.6)), mala, 32, 42, 0, 6, 4; initial_params = [1.0, -0.5, 0.25, 0.0],
               discard_initial = 2, thinning = 3)
end

# ---- Ensemble / StretchProposal, the reference's own sequential sweep (src/emcee.jl:14-102) ----------------------------
let d = 3, W = 10, N = 16
    model = DensityModel(corr_gauss(ar1(d, 0.9)))
    spl = Ensemble(W, StretchProposal(MvNormal(zeros(d), I)))                 # initial walkers: W draws from the prior (:29-34)
    rng = PhiloxStream(21, 0; dim = d, nwalkers = W)
    ts = sample(rng, model, spl, N; chain_type = Any, progress = false)       # Vector (sweeps) of Vector{Transition} (walkers)
    S = Array{Float64,3}(undef, N, d + 1, W)
    A = Matrix{UInt8}(undef, N, W)
    for (i, sweep) in enumerate(ts), (w, t) in enumerate(sweep)
        S[i, 1:d, w] .= t.params
        S[i, d + 1, w] = t.lp
        A[i, w] = t.accepted ? 0x01 : 0x00
    end
    write_npy(joinpath(outdir, "emcee_seq_samples.npy"), S)
    write_npy(joinpath(outdir, "emcee_seq_accepted.npy"), A)
    println("wrote emcee_seq: ", size(S))
end
