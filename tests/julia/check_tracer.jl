# check_tracer.jl -- run on a machine WITH Julia (none exists in the build container or on the GPU box):
#
#     julia --project=<env with AdvancedMH, Distributions> tests/julia/check_tracer.jl
#
# Traces the README density (README.md:25-40) and the NIG density of test/emcee.jl:5-14 -- written exactly as the reference writes
# them, branches on parameter values included -- with advancedmh.jl_amd/julia/MHXTrace.jl and compares the emitted HIP source with
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

bad = 0
for (name, f) in (("readme", density), ("nig", logprob))
    got = trace_logdensity(f, 2)
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
exit(bad)
