# check_tracer.jl -- run on a machine WITH Julia (none exists in the build container or on the GPU box):
#
#     julia --project=<env with AdvancedMH, Distributions> tests/julia/check_tracer.jl
#
# Traces the README density (README.md:25-40) and the NIG density of test/emcee.jl:5-14 -- written exactly as the reference writes
# them, branches on parameter values included -- and the same README model plus the RAM test's Gaussian in the LogDensityProblems
# form (README.md:75-90, test/RobustAdaptiveMetropolis.jl:1-9) -- with advancedmh.jl_amd/julia/MHXTrace.jl and compares the emitted HIP source with
# the committed fixtures tests/golden/traced_readme.hip / traced_nig.hip CHARACTER FOR CHARACTER.  The fixtures are what the Python
# tracer (mhx.trace) emits for the same arithmetic and what tests/julia_tracer_model.py (this tracer's algorithm, executed) emits;
# same text => same hiprtc module => the kernels the GPU suite holds to the oracle bit for bit.  Exit code 0 = identical.
using Distributions

include(joinpath(@__DIR__, "..", "..", "advancedmh.jl_amd", "julia", "MHXTrace.jl"))
using .MHXTrace

# the first 30 points of tests/golden/c1_normal_data.npy (float32 values, exact in Float64), as hexadecimal floats
const data = parse.(Float64, split(strip(read(joinpath(@__DIR__, "..", "golden", "c1_normal_data_30.txt"), String))))

# README.md:29-31, verbatim
insupport(θ) = θ[2] >= 0
dist(θ) = Normal(θ[1], θ[2])
density(θ) = insupport(θ) ? sum(logpdf.(dist(θ), data)) : -Inf

# test/emcee.jl:5-14, verbatim
function logprob(θ)
    s, m = θ
    s > 0 || return -Inf

    mdist = Normal(0, sqrt(s))
    obsdist = Normal(m, sqrt(s))

    return logpdf(InverseGamma(2, 3), s) + logpdf(mdist, m) +
        logpdf(obsdist, 1.5) + logpdf(obsdist, 2.0)
end

# ---- the LogDensityProblems form (README.md:75-90; the only model form RobustAdaptiveMetropolis takes, …RAM.jl:175-181): the glue's
# method for AbstractMCMC.LogDensityModel lowers  θ -> LogDensityProblems.logdensity(ℓ, θ)  with dim = LogDensityProblems.dimension(ℓ)
import LogDensityProblems
struct LogTargetDensity end                                                  # README.md:80-88, verbatim
LogDensityProblems.logdensity(p::LogTargetDensity, θ) = density(θ)
LogDensityProblems.dimension(p::LogTargetDensity) = 2
LogDensityProblems.capabilities(::LogTargetDensity) = LogDensityProblems.LogDensityOrder{0}()
lower_problem(ℓ) = trace_logdensity(θ -> LogDensityProblems.logdensity(ℓ, θ), LogDensityProblems.dimension(ℓ))   # AdvancedMHHIP.jl, same line

# test/RobustAdaptiveMetropolis.jl:1-9,33-40: Σ = [σ² ρ; ρ σ²], ρ = σ²/2, with the log-density written out (forward substitution
# with L = chol(Σ)); the two logarithms are passed as the doubles the fixtures were made with (a libm may round `log` differently)
struct Gaussian2
    l11::Float64; l21::Float64; l22::Float64; lg11::Float64; lg22::Float64
end
function Gaussian2(σ²::Float64, lg11::Float64, lg22::Float64)
    l11 = sqrt(σ²); l21 = (σ² / 2) / l11
    return Gaussian2(l11, l21, sqrt(σ² - l21 * l21), lg11, lg22)
end
LogDensityProblems.dimension(::Gaussian2) = 2
LogDensityProblems.capabilities(::Gaussian2) = LogDensityProblems.LogDensityOrder{0}()
function LogDensityProblems.logdensity(g::Gaussian2, x)
    z1 = x[1] / g.l11
    z2 = (x[2] - g.l21 * z1) / g.l22
    return -(z1 * z1 + z2 * z2) / 2 - g.lg11 - g.lg22 - 1.8378770664093453
end

# ... and the reference test's own model, through Distributions' generic MvNormal code: no pinned text (the order of a library's
# operations is not ours to fix), it only has to trace
struct Gaussian{A}
    Σ::A
end
LogDensityProblems.dimension(model::Gaussian) = size(model.Σ, 1)
LogDensityProblems.capabilities(::Gaussian) = LogDensityProblems.LogDensityOrder{0}()
LogDensityProblems.logdensity(model::Gaussian, x) = logpdf(MvNormal(zeros(LogDensityProblems.dimension(model)), model.Σ), x)

bad = 0
for (name, f) in (("readme", density), ("nig", logprob), ("readme", LogTargetDensity()),
                  ("ldp_gauss2_s10", Gaussian2(10.0, 0x1.26bb1bbb55516p+0, 0x1.01e85798eb9a3p+0)),
                  ("ldp_gauss2_s001", Gaussian2(0.01, -0x1.26bb1bbb55515p+1, -0x1.39247dcc8a2cfp+1)))
    got = f isa Function ? trace_logdensity(f, 2) : lower_problem(f)
    want = read(joinpath(@__DIR__, "..", "golden", "traced_$(name).hip"), String)
    if got == want
        println("traced_$(name).hip: identical (", count(==('\n'), got), " lines)")
    else
        global bad += 1
        gl, wl = split(got, '\n'), split(want, '\n')
        k = findfirst(i -> i > length(gl) || i > length(wl) || gl[i] != wl[i], 1:max(length(gl), length(wl)))
        println("traced_$(name).hip: DIFFERS at line $(k)\n  julia : ", k <= length(gl) ? gl[k] : "<end>", "\n  golden: ", k <= length(wl) ? wl[k] : "<end>")
    end
end
try
    src = lower_problem(Gaussian([10.0 5.0; 5.0 10.0]))
    println("Gaussian(Σ) through Distributions.MvNormal: traces (", count(==('\n'), src), " lines)")
catch e
    println("Gaussian(Σ) through Distributions.MvNormal: does NOT trace (", e, ") -- write the density out, or use CorrGaussian(Σ)")
end
exit(bad)
