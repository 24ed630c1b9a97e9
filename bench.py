#!/usr/bin/env python3
"""bench.py -- headline benchmark: MH steps/sec (all chains) + ESS/sec.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--dtype f64|f32]

Default: BASELINE.json configs[1] (C2) -- RWMH on the isotropic 100-dim Gaussian, 65 536 chains per GPU -- in fp64, the
arithmetic the reference computes in (Distributions' Float64 rand / logpdf).  The same run also reports the fp32 engine
as a second figure (`f32`), never as `value`.

One "step" = one pass of the hot path over one batch = ONE mhx_run_sample call that advances all chains of this rank by
`--inner` transitions (C2: every state recorded into the HBM-resident sample tensor [inner][d+1][chains], the save-all
semantics of the reference's `sample`).  Inputs (chain state) and outputs (samples) stay in HBM; nothing crosses PCIe
inside the timed region.  Chains are sharded over ranks by global chain id (first_chain = rank * chains), no data-path
collective; scaling is weak.  The only collective -- acceptance totals + R-hat sums, and the max-over-ranks time -- is
ONE small all-reduce through the C ABI (mhx_comm_*: RCCL over xGMI).

Launching.  `python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) makes THIS process the launcher: it starts
N ranks of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), each bound to ITS device before
HIP initialises (HIP_VISIBLE_DEVICES = the rank's entry of the visible list), and rank 0 prints the line.  Under torchrun
(WORLD_SIZE set) the process is one rank of the launcher's and takes device LOCAL_RANK.  `--gpus N --single-process` is the
third way: ONE process, N member contexts and N host threads behind the C ABI (mhx_group_*), the statistics summed on the host.
A line never claims what did not run: --gpus must equal WORLD_SIZE, the box must have N devices (unless --allow-gloo, the one-GPU
rehearsal), `n_gpus` is the number of DISTINCT PCI bus ids the ranks / members opened (ordinals depend on the visibility
variables, bus ids do not), `n_ranks` the number of processes, and `config.collective` names the transport that really carried
the small all-reduces.  A multi-GPU run must not end with NO line: the transports form a ladder -- RCCL behind the C ABI
(mhx_comm_*, the product path: ncclCommInitRank and every collective under a deadline that names the rank), then
torch.distributed's own nccl backend, then gloo -- every rung agreed on by all ranks, a failed rung reported in the line
(`config.rccl_error`, `config.transport_errors`); `--require-rccl` turns the first fallback into an error instead.  `--dry-run`
runs launcher, rendezvous, ladder and the host-side combining with no device and no engine (`value` is null).

Prints ONE JSON line on rank 0, < 8 KB (the driver keeps an 8 KB tail): numbers only -- what each field means, the byte
models and the workload texts are DESIGN.md section 7.  `roofline` prices the dominant kernel (bound "hbm": algorithmic bytes
of DESIGN.md section 7 over the HIP-event launch time measured here; bound "valu": PMC wave-instructions per launch over the
same time against the issue peak); `cpu_baseline` is the CPU oracle (a port of the reference algorithm, oracle/, rebuilt -O3
-march=native on the host that times it) on a bounded sample -- rank 0, N=1 only.  At N=1 the default (c2) run also carries,
all measured after the timed region:
  `e2e_host`  the rate THROUGH the boundary -- samples back on the host (mhx_run_sample_to_host), save-all and thinned;
  `configs`   the other BASELINE.json configs and their SURVEY 8(d) variants (c1, c2_literal, c3, c3_rotated, c4,
              c4_moving, c4_fixed, c4_deferred, c5, c5_banana): {value, ms_per_step, acc, bound, frac, traffic_ratio, gen, cpu, ...} each;
  `ess`       the ESS/sec window;  `f32` the fp32 engine on the same workload.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 units, a wave64 VALU instruction issues over 2 cycles at 2.4 GHz (measured:
# profiles/r02a_valu_rates.log -- v_fma_f32 1.10 ns per instruction per SIMD at 8 waves; v_fma_f64 is 4 cycles, v_mad_u64_u32
# ~5, v_rcp_f64 / v_sqrt_f64 ~16).  The issue peak below is the 2-cycle one for every instruction class.
VALU_PEAK = 256 * 4 * 2.4e9 / 2.0
RB = {"f32": 4, "f64": 8}


GEN_TEXT = {None: "Box-Muller", "box-muller": "Box-Muller", "ziggurat": "ziggurat"}


def pick_gen(args, dtype, coop=True):
    """RWMH: the ziggurat generator (what Julia's own randn is) -- the cooperative and the register kernel have the form in both
    widths (fp32 since round 6); on the fp32 REGISTER kernel Box-Muller is still the faster stream by 5 % (profiles/r06_zig32_ab.txt)
    and stays the default there; --normal-gen overrides."""
    g = getattr(args, "normal_gen", "auto")
    if g == "auto":
        return "ziggurat" if dtype == "f64" or coop else None
    return None if g == "box-muller" else g


def sigma_ar1(d, rho):
    import numpy as np
    i = np.arange(d)
    return rho ** np.abs(i[:, None] - i[None, :])


def sigma_illcond(d, kappa=1e3, seed=7):
    """Sigma = Q diag(lambda) Q^T, lambda_i = kappa^((i-1)/(d-1)), Q = QR of a seeded Gaussian matrix"""
    import numpy as np
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(d, d)))
    lam = kappa ** (np.arange(d) / (d - 1.0))
    return (Q * lam) @ Q.T


def sig(x, n=5):
    """a float rounded to n significant digits (the line is numbers, not prose: < 8 KB)"""
    if x is None or isinstance(x, (bool, str)):
        return x
    x = float(x)
    if x == 0.0 or not math.isfinite(x):
        return x
    return float("%.*g" % (n, x))


def cgroup_cpu_quota():
    """cores the container's CPU controller grants (cgroup v2 cpu.max, v1 cfs_quota / cfs_period), or None when unlimited"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return q / p
    except Exception:
        pass
    return None


def host_cores():
    """(threads to run, logical CPUs of the affinity mask, SMT siblings per core, cgroup quota or None): the affinity mask can be
    wider than what the CPU controller grants -- a quota caps the thread count"""
    try:
        logical = max(1, len(os.sched_getaffinity(0)))
    except Exception:
        logical = os.cpu_count() or 1
    smt = 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len([x for part in sib.split(",") for x in ([part] if "-" not in part else
                                                                  range(int(part.split("-")[0]), int(part.split("-")[1]) + 1))]))
    except Exception:
        pass
    quota = cgroup_cpu_quota()
    threads = logical if quota is None else max(1, min(logical, int(math.ceil(quota))))
    return threads, logical, smt, quota


def mt_baseline(run_mt, units_per_chain, target_seconds):
    """The CPU baseline of a many-chain config: oracle/mhx_oracle_mt.c -- POSIX threads, a chain per task, each chain's record
    contiguous (the MCMCThreads shape).  run_mt(nchains, nthreads) -> (wall seconds, per-thread CPU seconds).  Single-thread rate
    first, then every hardware thread the host grants and (with SMT) one thread per physical core; the better one is `value`.
    parallel_efficiency = value / (threads x single-thread rate); below 0.5 `why` says where it went: "cpu-not-granted" (the
    threads got less CPU time than wall x threads: a quota or a busy host), else "smt-or-memory" (the time was granted and each
    thread ran slower: hyperthread siblings share a core's pipes, or the records exceed the caches)."""
    w, _ = run_mt(1, 1)                                       # warm: page faults, thread start
    w, _ = run_mt(2, 1)
    n1 = max(2, int(2 * min(1.0, 0.15 * target_seconds) / max(w, 1e-6)))
    w1, _ = run_mt(n1, 1)
    single = n1 * units_per_chain / w1
    threads, logical, smt, quota = host_cores()
    tries = [threads] + ([threads // smt] if smt > 1 and threads // smt >= 1 and quota is None else [])
    best = None
    for nt in tries:
        per = 0.8 * target_seconds / len(tries)
        n = max(2 * nt, int(nt * single * per / units_per_chain))
        w, busy = run_mt(n, nt)
        rate = n * units_per_chain / w
        granted = float(busy.sum()) / w
        if best is None or rate > best["value"]:
            best = {"value": rate, "cores": nt, "chains": n, "wall_s": w, "granted": granted}
    eff = best["value"] / (best["cores"] * single)
    out = {"value": best["value"], "cores": best["cores"], "threads_used": best["cores"], "single_thread": single,
           "parallel_efficiency": eff, "cpu_granted": best["granted"], "logical_cpus": logical, "smt": smt, "cgroup_quota": quota,
           "chains": best["chains"], "wall_s": best["wall_s"]}
    if eff < 0.5:
        out["why"] = "cpu-not-granted" if best["granted"] < 0.5 * best["cores"] else "smt-or-memory"
    return out


# ------------------------------------------------------------------------------------------------------------------
# workloads: each builds a Run for this rank and says what one step is, what it moves and how the CPU oracle does it


class C2:
    """BASELINE configs[1]: isotropic 100-dim standard MvNormal, RWMH, 65 536 chains per GPU, every state recorded"""
    name = "c2"

    def __init__(self, args, dtype):
        self.d, self.C, self.inner, self.dtype = args.dim or 100, args.chains or 65536, args.inner or 250, dtype
        self.lanes = args.lanes
        self.literal = getattr(args, "c2_literal", False)
        self.user = getattr(args, "c2_user", False)       # the same target as a user log-density in HIP source: DensityModel(f), JIT-lowered
        self.gen = pick_gen(args, dtype, coop=not self.user)

    USER_SOURCE = """
MHX_LOGDENSITY(x, d, data, ndata)
{
    mhx_real q = MHX_R(0.0);
    for (int k = 0; k < d; ++k) q = mhx_fma(x[k], x[k], q);
    return -MHX_R(0.5) * q;
}
"""

    def build(self, mhx, ctx, rank):
        import numpy as np
        d = self.d
        self.s = 1.0 if self.literal else float(np.float32(2.38 / d ** 0.5))
        model = mhx.DensityModel(mhx.HipLogDensity(self.USER_SOURCE, d) if self.user else mhx.IsoGaussian(d))
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), self.s * self.s * mhx.I))
        self.run = mhx.Run(model, spl, nchains=self.C, seed=0xC0FFEE, first_chain=rank * self.C, ctx=ctx, reduce_lanes=self.lanes,
                           normal_gen=self.gen)
        self.run.init(None)                                # x0 ~ proposal draw (src/mh-core.jl:83), on the device
        return self.run

    def sched(self):
        # N = inner saved samples, the first one being the state after 1 transition: inner transitions
        return (self.inner, 1, 1, 0, True)

    def step(self):
        self.run.sample(*self.sched()[:4], save=self.sched()[4])
        return self.run.stats()

    def units_per_step(self):
        return self.C * self.inner

    def bytes_per_launch(self):
        """sample record B(d+1)+1 per chain-step (save-all) + one state round trip per launch"""
        B = RB[self.dtype]
        return self.C * (self.inner * (B * (self.d + 1) + 1) + 2 * (B * self.d + B + 4 + 1))

    def describe(self):
        return "RWMH d=%d iso-Gaussian%s, %d chains/GPU, proposal %s, %d transitions/launch, save-all, %s normals" % (
            self.d, " as a user log-density (HIP source, hiprtc)" if self.user else "", self.C,
            "N(0,I) (literal)" if self.literal else "N(0,(2.38/sqrt d)^2 I)", self.inner, GEN_TEXT[self.gen])

    def cpu_baseline(self, O, target_seconds):
        tgt = O.iso_gauss(self.d)
        prop = O.Proposal(O.PROP_ISO, self.s, normal_gen=1 if self.gen == "ziggurat" else 0)
        sched = O.schedule(self.inner + 1)
        r = mt_baseline(lambda n, nt: O.mt_rwmh(tgt, prop, sched, 0xC0FFEE, 0, n, nt, save=True), self.inner, target_seconds)
        r["sample"] = "%d chains x %d transitions, d=%d, save-all" % (r["chains"], self.inner, self.d)
        return r


class C5(C2):
    """BASELINE configs[4], one GPU's shard: 1000-dim Neal's funnel, RWMH, 32 768 chains per GPU (262 144 over 8), no sample
    tensor -- running moments of every 10th state for the R-hat reduction"""
    name = "c5"

    def __init__(self, args, dtype):
        self.d, self.C, self.inner, self.dtype = args.dim or 1000, args.chains or 32768, args.inner or 200, dtype
        self.lanes = args.lanes
        self.gen = pick_gen(args, dtype)
        self.banana = getattr(args, "c5_banana", False)    # x2 <- x2 + b (x1^2 - 100), b = 0.03

    def build(self, mhx, ctx, rank):
        import numpy as np
        d = self.d
        self.s = float(np.float32(2.38 / d ** 0.5))
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), self.s * self.s * mhx.I))
        model = mhx.DensityModel(mhx.Banana(d, 0.03) if self.banana else mhx.Funnel(d))
        self.run = mhx.Run(model, spl, nchains=self.C, seed=5, first_chain=rank * self.C, ctx=ctx,
                           reduce_lanes=self.lanes, normal_gen=self.gen)
        if self.banana:                 # a stationary start: x1 ~ N(0, 100), x2 = z - b (x1^2 - 100), the rest N(0, 1)
            rng = np.random.default_rng(2000 + rank)
            x0 = rng.normal(size=(d, self.C))
            x0[0] *= 10.0
            x0[1] -= 0.03 * (x0[0] ** 2 - 100.0)
            self.run.init(x0)
            return self.run
        # a stationary start (a draw from the funnel itself: v ~ N(0, 9), x_k ~ N(0, e^v)) instead of a burn-in: from the
        # proposal-scale start of init(None) the chains spend thousands of transitions inflating |x|^2
        rng = np.random.default_rng(1000 + rank)
        x0 = rng.normal(size=(d, self.C))
        x0[0] *= 3.0
        x0[1:] *= np.exp(0.5 * x0[0])
        self.run.init(x0)
        return self.run

    def sched(self):
        return (self.inner // 10, 10, 10, 0, "moments")                  # every 10th state folded into the moments

    def bytes_per_launch(self):
        """the state lives in registers for a launch: HBM sees one state round trip and one moments round trip per launch"""
        B = RB[self.dtype]
        return self.C * (2 * (B * self.d + B + 4 + 1) + 4 * B * (self.d + 1))

    def describe(self):
        return "RWMH d=%d %s, %d chains/GPU (1 of 8 shards, global ids), proposal N(0,(2.38/sqrt d)^2 I), %d transitions/launch, moments of every 10th state, %s normals" % (
            self.d, "banana b=0.03" if self.banana else "funnel", self.C, self.inner, GEN_TEXT[self.gen])

    def cpu_baseline(self, O, target_seconds):
        tgt = O.Target(O.TARGET_BANANA, self.d, params=[0.03]) if self.banana else O.Target(O.TARGET_FUNNEL, self.d)
        prop = O.Proposal(O.PROP_ISO, self.s, normal_gen=1 if self.gen == "ziggurat" else 0)
        sched = O.schedule(1, self.inner)
        r = mt_baseline(lambda n, nt: O.mt_rwmh(tgt, prop, sched, 5, 0, n, nt, save=False), self.inner, target_seconds)
        r["sample"] = "%d chains x %d transitions, d=%d %s" % (r["chains"], self.inner, self.d, "banana" if self.banana else "funnel")
        return r


class C3:
    """BASELINE configs[2]: emcee Ensemble(StretchProposal), 50-dim correlated Gaussian (rho = 0.9), 16 384 walkers"""
    name = "c3"

    def __init__(self, args, dtype):
        self.d, self.W, self.inner, self.dtype = args.dim or 50, args.chains or 16384, args.inner or 500, dtype
        self.E = 1
        if getattr(args, "c3_small", False):
            # the sizes emcee is run at (test/emcee.jl:24: 1 000 walkers), as MANY ensembles in one run -- what
            # `sample(model, Ensemble(1000, ..), MCMCThreads(), N, nchains)` is (README.md:135-148); --ensembles 1 = one ensemble alone
            self.d, self.W = args.dim or 5, args.chains or 1000
            self.E = max(1, getattr(args, "ensembles", 0) or 256)
        self.lanes = args.lanes
        self.rotated = getattr(args, "c3_rotated", False)
        self.user = getattr(args, "c3_user", False)       # the AR(1) target as a user log-density in HIP source (DensityModel(f)): lane per walker

    # log N(0, Sigma), Sigma_ij = rho^|i-j|, through the bidiagonal precision factor: data = [diag[d], sub[d]]
    USER_SOURCE = """
MHX_LOGDENSITY(x, d, data, ndata)
{
    mhx_real q = MHX_R(0.0);
    mhx_real prev = MHX_R(0.0);
    for (int k = 0; k < d; ++k) {
        const mhx_real w = mhx_fma(data[k], x[k], data[d + k] * prev);
        q = mhx_fma(w, w, q);
        prev = x[k];
    }
    return -MHX_R(0.5) * q;
}
"""

    def build(self, mhx, ctx, rank):
        d = self.d
        import numpy as np
        self.Sig = sigma_ar1(d, 0.9)
        if self.rotated:                                   # "also a dense-rotated variant" -- no structural zeros in the factor
            Q, _ = np.linalg.qr(np.random.default_rng(50).normal(size=(d, d)))
            self.Sig = Q @ self.Sig @ Q.T
        if self.user:
            A = np.linalg.inv(np.linalg.cholesky(self.Sig))                      # lower bidiagonal for the AR(1) model
            data = np.concatenate([np.diag(A), np.concatenate([[0.0], np.diag(A, -1)])])
            model = mhx.DensityModel(mhx.HipLogDensity(self.USER_SOURCE, d, data=data))
        else:
            model = mhx.DensityModel(mhx.CorrGaussian(self.Sig))
        spl = mhx.Ensemble(self.W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
        # one ensemble per GPU (replicas), or E of them in one run (ids rank E .. rank E + E - 1)
        self.run = mhx.Run(model, spl, nchains=self.E, seed=3, first_chain=rank * self.E, ctx=ctx, reduce_lanes=self.lanes)
        self.run.init(None)
        return self.run

    def sched(self):
        return (self.inner, 1, 1, 0, True)

    def step(self):
        self.run.sample(*self.sched()[:4], save=self.sched()[4])
        return self.run.stats()

    def units_per_step(self):
        return self.W * self.E * self.inner

    def bytes_per_launch(self):
        """per move read x_i, x_j, write x_i, lp r/w (3Bd + 2B) + record B(d+1)+1"""
        B = RB[self.dtype]
        if hasattr(self, "run") and self.run.stats().get("kernel_variant") == 6:
            # a persistent block per ensemble: walkers stay in registers / LDS for the whole launch -- the record is all that moves
            return self.W * self.E * (self.inner * (B * (self.d + 1) + 1) + 2 * (B * self.d + 2 * B))
        return self.W * self.E * self.inner * (3 * B * self.d + 2 * B + B * (self.d + 1) + 1)

    def describe(self):
        st = self.run.stats() if hasattr(self, "run") else {}
        band = st.get("factor_band", -1)
        per_sweep = "one launch per sweep" if 0 < st.get("launches", 0) <= self.inner else "two launches per sweep"
        if st.get("kernel_variant") == 6:
            per_sweep = "a persistent block per ensemble: one launch per step"
        return "emcee stretch a=2, d=%d Gaussian %s, %d walkers (%s), %d sweeps/step, save-all, half-split sweep (%s), %s" % (
            self.d, "Q(0.9^|i-j|)Q^T" if self.rotated else "0.9^|i-j|", self.W, "one ensemble/GPU" if self.E == 1 else "%d ensembles in one run" % self.E,
            self.inner, per_sweep,
            "user log-density (HIP source, hiprtc)" if self.user else "factor " + ("band %d" % band if band >= 0 else "dense"))

    def cpu_baseline(self, O, target_seconds):
        import numpy as np
        tgt = O.corr_gauss_from_cov(self.Sig)
        W = self.W
        init = np.random.default_rng(1).normal(size=(self.d, W))
        t0 = time.perf_counter()
        O.emcee(tgt, 2.0, 0, O.schedule(3), 3, 0, W, init, save=True)          # mode 0: the reference's sequential sweep
        rate = 2 * W / (time.perf_counter() - t0)
        sweeps = max(2, int(rate * target_seconds / W))
        t0 = time.perf_counter()
        O.emcee(tgt, 2.0, 0, O.schedule(sweeps + 1), 3, 0, W, init, save=True)
        dt = time.perf_counter() - t0
        return {"value": W * sweeps / dt, "cores": 1, "threads_used": 1, "single_thread": W * sweeps / dt, "parallel_efficiency": 1.0,
                "wall_s": dt, "sample": "%d sequential sweeps (src/emcee.jl:39-58 is serial in W), %d walkers" % (sweeps, W)}


class C4:
    """BASELINE configs[3]: RobustAdaptiveMetropolis, 200-dim ill-conditioned Gaussian (kappa = 1e3), 32 768 chains, adapting"""
    name = "c4"

    def __init__(self, args, dtype):
        self.d, self.C, self.inner, self.dtype = args.dim or 200, args.chains or 32768, args.inner or 100, dtype
        self.moving = args.c4_moving
        self.fixed = getattr(args, "c4_fixed", False)      # "... + 500 fixed steps": S frozen after a warm-up
        self.deferred = getattr(args, "c4_deferred", False)  # MHX_FLAG_RAM_DEFERRED: up to 8 updates pending, one fold per 8 steps

    def build(self, mhx, ctx, rank):
        import numpy as np
        d = self.d
        self.Sig = sigma_illcond(d)
        model = mhx.DensityModel(mhx.CorrGaussian(self.Sig))
        S0 = None
        if self.moving:           # a variant that moves: random start on the target's scale, S0 = 2.38/sqrt(d) I (about 0.17 I)
            S0 = (2.38 / d ** 0.5) * np.eye(d)
        self.run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(S=S0, deferred_factor=self.deferred), nchains=self.C, seed=4,
                           first_chain=rank * self.C, ctx=ctx)
        if self.moving:
            L = np.linalg.cholesky(self.Sig)
            self.run.init(L @ np.random.default_rng(11).normal(size=(d, self.C)))
        else:
            self.run.init(np.zeros(d))                      # x0 = 0, S0 = I
        if self.fixed:
            self.run.sample(1, self.inner, 1, self.inner, save=False)   # a warm-up first: the timed steps run on adapted factors
        return self.run

    def sched(self):
        # `inner` adapting transitions (step_warmup), or `inner` transitions with the factor frozen (RAM.jl:216-237)
        return (1, self.inner, 1, 0 if self.fixed else self.inner, False)

    def step(self):
        self.run.sample(*self.sched()[:4], save=self.sched()[4])
        return self.run.stats()

    def units_per_step(self):
        return self.C * self.inner

    def bytes_per_launch(self):
        """one read + one write of the packed factor per adapting step, B d(d+1) + 2Bd + 2B"""
        B = RB[self.dtype]
        if self.fixed:
            return self.C * self.inner * (B * self.d * (self.d + 1) // 2 + 2 * B * self.d + 2 * B)
        if self.deferred:
            # its own byte model: per 8 adapting steps 7 reads of S (the proposals; the first one of a block rides on the previous
            # fold's output) + 1 read + 1 write (the fold) = 9/8 passes per step -- NOT priced against the sweep form's 1 + 1
            tri = B * self.d * (self.d + 1) // 2
            return self.C * self.inner * (tri * 9 // 8 + 2 * B * self.d + 2 * B)
        return self.C * self.inner * (B * self.d * (self.d + 1) + 2 * B * self.d + 2 * B)

    def describe(self):
        return "RAM alpha=0.234 gamma=0.6, d=%d Gaussian kappa=1e3, %d chains/GPU (own factor each), %d %s transitions/launch, %s" % (
            self.d, self.C, self.inner, "fixed-factor" if self.fixed else ("adapting (deferred factor, K=8)" if self.deferred else "adapting"),
            "random start, S0=2.38/sqrt(d) I" if self.moving else "x0=0, S0=I")

    def cpu_baseline(self, O, target_seconds):
        import numpy as np
        tgt = O.corr_gauss_from_cov(self.Sig)
        inner, d = self.inner, self.d
        init1 = np.zeros(d)
        warm = 0 if self.fixed else inner                  # fixed-factor steps: the oracle's steps on S0 cost what they cost on any S
        sched = O.schedule(1, inner, 1, warm)
        r = mt_baseline(lambda n, nt: O.mt_ram(tgt, sched, 4, 0, n, nt, save=False, init1=init1), inner, target_seconds)
        r["sample"] = "%d chains x %d %s transitions, d=%d" % (r["chains"], inner, "fixed-factor" if self.fixed else "adapting", d)
        return r


class C1:
    """BASELINE configs[0], the reference's own CPU-runnable case (README.md:18-63): d = 2, Normal(mu, sigma) likelihood of 30
    data points, RWMH with N(0, I), ONE chain, 100 000 draws.  On a GPU this is plumbing -- a single chain occupies one lane
    of one wave and runs at the latency of one transition -- reported so that the line says what a user who ports the README
    example unchanged gets; `chains` > 1 is what the engine is for."""
    name = "c1"

    def __init__(self, args, dtype):
        self.d, self.C, self.inner, self.dtype = 2, args.chains or 1, args.inner or 100000, dtype

    def build(self, mhx, ctx, rank):
        import numpy as np
        self.data = np.random.default_rng(1234).normal(size=30)
        model = mhx.DensityModel(mhx.IIDNormal(self.data))
        self.run = mhx.Run(model, mhx.RWMH(mhx.MvNormal(mhx.zeros(2), mhx.I)), nchains=self.C, seed=1234, first_chain=rank * self.C, ctx=ctx)
        self.run.init(np.array([0.0, 1.0]))
        return self.run

    def sched(self):
        return (self.inner, 1, 1, 0, True)

    def step(self):
        self.run.sample(*self.sched()[:4], save=self.sched()[4])
        return self.run.stats()

    def units_per_step(self):
        return self.C * self.inner

    def bytes_per_launch(self):
        B = RB[self.dtype]
        return self.C * (self.inner * (B * 3 + 1) + 2 * (B * 2 + B + 5))

    def describe(self):
        return "RWMH d=2 Normal(mu,sigma) likelihood of 30 points (README.md:18-63), proposal N(0,I), %d chain x %d draws, save-all" % (self.C, self.inner)

    def cpu_baseline(self, O, target_seconds):
        tgt = O.Target(O.TARGET_IID_NORMAL, 2, params=self.data)
        prop = O.Proposal(O.PROP_ISO, 1.0)
        import numpy as np
        t0 = time.perf_counter()
        O.rwmh(tgt, prop, O.schedule(self.inner), 1234, 0, 1, init=np.array([[0.0], [1.0]]), save=True)
        dt = time.perf_counter() - t0
        return {"value": self.inner / dt, "cores": 1, "threads_used": 1, "single_thread": self.inner / dt, "parallel_efficiency": 1.0,
                "wall_s": dt, "sample": "1 chain x %d draws, one thread (`sample(model, spl, N)`)" % self.inner}


WORKLOADS = {"c1": C1, "c2": C2, "c3": C3, "c4": C4, "c5": C5}


_ORACLE_FLAGS = None


def cpu_baseline(wl, dtype, target_seconds=10.0, compact=False):
    """The oracle (same algorithm, same Philox streams, scalar loop per chain) on the host cores: a chain per task over POSIX
    threads (oracle/mhx_oracle_mt.c -- the MCMCThreads analogue; each chain's record contiguous).  Bounded sample.  Built on
    this host with -O3 -march=native as BASELINE.md's CPU-baseline plan says (oracle/Makefile `native`; bit-identical to the
    portable build the tests use)."""
    global _ORACLE_FLAGS
    from oracle import oracle as O
    O.build()
    if _ORACLE_FLAGS is None:
        _ORACLE_FLAGS = O.use_native()
    O.set_dtype(dtype)
    r = wl.cpu_baseline(O, target_seconds)
    if compact:
        out = {"value": sig(r["value"]), "cores": r["cores"], "single_thread": sig(r["single_thread"], 4),
               "parallel_efficiency": sig(r["parallel_efficiency"], 3)}
        if "why" in r:
            out["why"] = r["why"]
        return out
    out = {"value": r["value"], "unit": "MH steps/s", "cores": r["cores"], "kind": "port",
           "sample": "%s; oracle %s, %s, %.1f s" % (r["sample"], dtype, "native" if "native" in _ORACLE_FLAGS else "portable", r["wall_s"]),
           "threads_used": r["threads_used"], "single_thread": sig(r["single_thread"]), "parallel_efficiency": sig(r["parallel_efficiency"], 3)}
    for k in ("cpu_granted", "logical_cpus", "smt", "cgroup_quota", "why"):
        if k in r:
            out[k] = sig(r[k], 4)
    return out


class GlooSum:
    """Last rung of the transport ladder, and the transport of --dry-run: the bench's tiny host-side all-reduces over
    torch.distributed gloo.  Same interface as mhx.dist.Comm.allreduce_sum."""
    name = "gloo"

    def __init__(self, dist):
        self.dist = dist

    def allreduce_sum(self, v):
        import numpy as np
        import torch
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.numpy()

    def ranks(self):
        return self.dist.get_world_size()

    def close(self):
        pass


class TorchNcclSum:
    """Second rung: torch.distributed's own nccl backend (the RCCL torch ships and initialises its own way) on device tensors --
    tried when the communicator behind the C ABI could not be created on every rank."""
    name = "torch.distributed nccl"

    def __init__(self, dist, device, seconds):
        import datetime
        import torch
        self.dist, self.torch = dist, torch
        self.device = torch.device("cuda", device)
        self.group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(10.0, seconds)))

    def allreduce_sum(self, v):
        import numpy as np
        t = self.torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy()).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def ranks(self):
        return self.dist.get_world_size(self.group)

    def close(self):
        pass


class MhxCommSum:
    """First rung, the product path: mhx_comm_* of the C ABI -- RCCL over xGMI, resolved by libmhx.so itself."""
    name = "rccl(mhx_comm_*)"

    def __init__(self, ctx, rank, world, uid, seconds):
        import ctypes as C
        import mhx._lib as L
        from mhx.dist import Comm
        self.comm = Comm.__new__(Comm)
        self.comm.ctx, self.comm.rank, self.comm.world, self.comm.h = ctx, rank, world, C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(uid))
        L.check(L.lib().mhx_comm_init_timed(ctx.h, rank, world, C.cast(buf, C.c_void_p), float(seconds), C.byref(self.comm.h)))
        L.check(L.lib().mhx_comm_set_timeout(self.comm.h, float(seconds)))
        self.rank, self.world = rank, world

    def allreduce_sum(self, v):
        return self.comm.allreduce_sum(v)

    def ranks(self):
        r, w = self.comm.rank_world()                  # what RCCL itself reports (ncclCommUserRank / ncclCommCount)
        if w != self.world or r != self.rank:
            raise RuntimeError("RCCL reports rank %d of %d, the launcher said %d of %d" % (r, w, self.rank, self.world))
        return w

    def close(self):
        self.comm.close()


def with_deadline(fn, seconds, what):
    """fn() on a helper thread, abandoned after `seconds`: a transport that hangs must not take the line with it.  ctypes and torch
    release the GIL inside their calls, so the wait below really times out.  Returns (value, None) or (None, error text)."""
    import threading
    box = {}

    def work():
        try:
            box["v"] = fn()
        except BaseException as e:                                # noqa: BLE001 -- the error text goes into the line
            box["e"] = ("%s: %s" % (type(e).__name__, " ".join(str(e).split())))[:220]
    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return None, "%s: no answer within %.0f s (abandoned)" % (what, seconds)
    if "e" in box:
        return None, box["e"]
    return box.get("v"), None


ABANDONED = []          # transports that hung: their threads still sit in a library call, so the process leaves through os._exit


def rendezvous(rank, world, seconds=300.0):
    """torch.distributed's own rendezvous (gloo on CPU; MASTER_ADDR / MASTER_PORT from the launcher): it carries the 128-byte RCCL
    id from rank 0 to the others, decides the ladder's rungs for all ranks together, and is the last rung itself.  `seconds`
    bounds the wait for ranks that never arrive."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=seconds))
    return dist


def make_collective(ctx, rank, world, args, device=0):
    """One process per GPU.  The bench's collectives -- barrier, max-over-ranks time, acceptance totals and R-hat sums -- as a
    LADDER of transports, so that the first contact with an N-GPU node cannot end with no line at all:
        1. rccl(mhx_comm_*)        the product path: RCCL behind the C ABI (ncclCommInitRank and every collective under a deadline)
        2. torch.distributed nccl  torch's own RCCL, initialised torch's way, on device tensors
        3. gloo                    host TCP; always there (it already carried the rendezvous)
    A rung is taken only if EVERY rank built it and passed a test all-reduce within --transport-timeout (agreed over gloo with a MIN);
    otherwise all ranks step down together.  Returns (transport, info): info = {collective, ranks_reported_by_transport,
    rccl_error, transport_errors}.  --require-rccl: stepping down from rung 1 is an error.  ctx None = --dry-run (no device:
    rungs 1 and 2 report that and the flow ends on gloo).  --fault-rccl-init fail|hang injects a failure into rung 1."""
    import numpy as np
    import torch
    dist = rendezvous(rank, world, args.rendezvous_timeout)
    T = float(args.transport_timeout)
    fault = getattr(args, "fault_rccl_init", None)

    def agree(ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.item() >= 1.0

    def gather_errors(err):
        texts = [None] * world
        dist.all_gather_object(texts, err)
        return {"rank%d" % r: e for r, e in enumerate(texts) if e}

    uid = [None]
    if ctx is not None and rank == 0:
        from mhx.dist import Comm
        uid[0], e = with_deadline(Comm.unique_id, T, "mhx_comm_unique_id")
    dist.broadcast_object_list(uid, src=0)

    def rung_rccl():
        if fault == "fail":
            raise RuntimeError("mhx_comm_init: rank %d of %d: injected failure (--fault-rccl-init fail)" % (rank, world))
        if fault == "hang":
            time.sleep(1e6)
        if ctx is None:
            raise RuntimeError("dry run: no device")
        if uid[0] is None:
            raise RuntimeError("rank 0 could not create the RCCL unique id (librccl not loadable?)")
        t = MhxCommSum(ctx, rank, world, uid[0], T)
        t.allreduce_sum(np.zeros(1))
        t.ranks()
        return t

    def rung_torch_nccl():
        if ctx is None or not torch.cuda.is_available():
            raise RuntimeError("dry run: no device")
        t = TorchNcclSum(dist, device, T)
        t.allreduce_sum(np.zeros(1))
        return t

    errors = {}
    for name, build in (("rccl(mhx_comm_*)", rung_rccl), ("torch.distributed nccl", rung_torch_nccl)):
        tr, err = with_deadline(build, T + 5.0, name)
        if tr is None and err and "abandoned" in err:
            ABANDONED.append(name)
        ok = agree(tr is not None)
        if ok:
            info = {"collective": "%s, %d ranks" % (name, tr.ranks()), "ranks_reported_by_transport": tr.ranks(),
                    "rccl_error": errors.get("rccl(mhx_comm_*)"),
                    "transport_errors": {k: v for k, v in errors.items() if not k.startswith("rccl")} or None}
            return tr, info
        errs = gather_errors(err or ("another rank failed" if tr is not None else "failed"))
        errors[name] = errs
        if tr is not None:
            try:
                tr.close()
            except Exception:
                pass
        if name.startswith("rccl") and args.require_rccl:
            raise RuntimeError("bench.py --gpus %d --require-rccl: the RCCL communicator behind the C ABI (mhx_comm_init) could not be "
                               "created on every rank: %s" % (world, json.dumps(errs)))
    tr = GlooSum(dist)
    info = {"collective": "gloo, %d ranks (fallback: see rccl_error)" % tr.ranks(), "ranks_reported_by_transport": tr.ranks(),
            "rccl_error": errors.get("rccl(mhx_comm_*)"),
            "transport_errors": {k: v for k, v in errors.items() if not k.startswith("rccl")} or None}
    return tr, info


def pci_number(bus_id):
    """"0000:05:00.0" -> an integer that identifies the device on this node (domain : bus : device . function)"""
    try:
        dom, bus, rest = bus_id.split(":")
        dev, fn = rest.split(".")
        return (int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn, 16)
    except Exception:
        import zlib
        return zlib.crc32(bus_id.encode()) | (1 << 40)


def timed(wl, steps, warmup, barrier, spin=30):
    for _ in range(max(0, spin - warmup)):    # device spin-up (setup): the first ~20 launches after idle run below the steady clock
        wl.step()
    for _ in range(warmup):
        wl.step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, accepted, transitions, st = 0.0, 0, 0, None
    for _ in range(steps):
        st = wl.step()                        # blocking: the stream is synchronised on return
        kernel_ms += st["kernel_ms"]
        accepted += st["accepted"]
        transitions += st["transitions"]
    barrier()
    return time.perf_counter() - t0, kernel_ms, accepted, transitions, st


def ess_window(mhx, wl, world):
    """ESS/sec of SURVEY 8(d): rank-normalised bulk ESS (split chains, Geyer truncation; mhx_run_ess_bulk_tail) of a THINNED
    window long enough for the autocorrelations to die out -- 256 draws `thin` transitions apart, thin chosen so that the window
    spans ~40 autocorrelation times (RWMH at the optimal scale: tau ~ d / 0.3; each split half ~20) -- over the wall time of
    producing that window."""
    import numpy as np
    d, run = wl.d, wl.run
    thin = max(1, int(round(40 * (d / 0.3) / 256)))        # each split half spans ~20 autocorrelation times
    n_draws = 256
    # the timed launches used a buffer of `inner` draws: size it for this window first (an allocation of GBs is not sampling)
    run.sample(n_draws, 1, 1, 0, save=True)
    t0 = time.perf_counter()
    run.sample(n_draws, thin, thin, 0, save=True)
    wall = time.perf_counter() - t0
    st = run.stats()
    params = sorted(set([0, d // 2, d - 1]))
    b = run.ess_bulk_tail(params=params, max_lag=n_draws // 2 - 2, ess_chains=512, split=True)
    dg = run.diagnostics(max_lag=0, split=True)
    med = float(np.median(b["ess_bulk"]))
    return {"draws": n_draws, "thin": thin, "chains": run.n, "params": [int(p) for p in params],
            "ess_bulk": [sig(v) for v in b["ess_bulk"]], "ess_tail": [sig(v) for v in b["ess_tail"]],
            "tail_ess_per_sec": sig(float(np.median(b["ess_tail"])) / wall), "median": sig(med),
            "per_transition_per_chain": sig(med / (run.n * n_draws * thin)), "wall_s": sig(wall), "kernel_ms": sig(st["kernel_ms"]),
            "rhat_max_split": sig(float(np.nanmax(dg["rhat"][:d]))), "ess_per_sec": med / wall}


KERNELS = {0: "generic", 1: "prebuilt-register", 2: "hiprtc-register", 3: "prebuilt-cooperative", 4: "hiprtc-cooperative",
           5: "hiprtc-dense-cooperative", 6: "persistent-block-ensemble", 7: "sequential-ensemble-sweep", 8: "matrix-core",
           9: "scalar-factor-ensemble", 10: "matrix-core-ensemble", 11: "wave-per-chain"}   # 6: one persistent block for a small ensemble


def kernel_name(wl, st):
    if wl.name == "c4" and getattr(wl, "deferred", False):
        return "ram-deferred-factor"                               # k_ram_defer<R,K>: pending triples, one fold per K steps
    if wl.name == "c4":
        return "ram-streamed-factor"                               # k_ram<G,R,W>: lane groups, the factor streamed through an LDS ring
    return KERNELS.get(st["kernel_variant"], str(st["kernel_variant"]))


def config_block(wl, name, st, collective):
    return {"workload": wl.describe(), "name": name, "units_per_step_per_gpu": wl.units_per_step(),
            "kernel_variant": kernel_name(wl, st), "lanes_per_unit": st["reduce_lanes"],
            "launches_per_step": max(1, st["launches"]),
            "sharding": "chains by global id, no data-path collective" if name != "c3" else "one ensemble per GPU (replicas)",
            "collective": collective}


def pmc_of(wl, name, dtype):
    """PMC figures of tools/profile_round.sh (profiles/traffic.json) when they were taken on exactly this workload"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath)).get("%s_%s" % (name, dtype), {})
    except Exception:
        return {}
    return tj if tj.get("units_per_launch") == wl.units_per_step() else {}


def isa_mix_of(key, dtype):
    """mean issue cycles per VALU instruction of the config's dominant kernel (tools/isa_mix.py: PMC class counts x the issue costs
    tools/ubench/valu_rates.hip measured on this chip), or None when no count was taken"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json"))).get("%s_%s" % (key, dtype), {}).get("mean_cycles_per_valu_instruction")
    except Exception:
        return None


def roofline_block(wl, key, dtype, kernel_ms, steps, st):
    """The dominant kernel against its bound.  HBM-bound configs: algorithmic bytes per step over the
    HIP-event time of the step's launches measured here.  C5 keeps its state in registers for a whole launch and is bound by VALU
    issue: PMC wave-instructions per launch (profiles/traffic.json) over the same time against 1024 SIMDs x 1 instruction per 2
    cycles at 2.4 GHz; its HBM figure rides along as `hbm_frac`.  `traffic` = PMC HBM bytes per step (FETCH_SIZE x 2 + WRITE_SIZE)."""
    launches = max(1, st["launches"])
    launch_s = kernel_ms * 1e-3 / steps
    bytes_launch = wl.bytes_per_launch()
    achieved = bytes_launch / launch_s / 1e9
    tj = pmc_of(wl, key, dtype)
    traffic = tj.get("hbm_bytes_per_launch")
    valu_frac = tj["valu_insts_per_launch"] / launch_s / VALU_PEAK if tj.get("valu_insts_per_launch") else None
    # the same instructions priced by their class (2 cycles is the cheapest class: fp64, 32-bit multiplies, v_bitop3, 64-bit mads
    # cost more): issue cycles the launch NEEDED over the cycles it HAD -- counted, not assumed (tools/isa_mix.py)
    cpi = isa_mix_of(key, dtype)
    weighted = valu_frac * cpi / 2.0 if valu_frac is not None and cpi else None
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "avg_launch_ms": launch_s * 1e3 / launches, "algorithmic_bytes_per_step": bytes_launch,
           "valu_frac": sig(valu_frac, 4), "valu_weighted_frac": sig(weighted, 4)}
    if wl.name == "c5" and valu_frac is not None:
        out = {"bound": "valu", "achieved": tj["valu_insts_per_launch"] / launch_s, "peak": VALU_PEAK, "unit": "wave-inst/s",
               "frac": valu_frac, "traffic": traffic, "avg_launch_ms": launch_s * 1e3 / launches,
               "valu_insts_per_step": tj["valu_insts_per_launch"], "algorithmic_bytes_per_step": bytes_launch,
               "hbm_frac": sig(achieved / HBM_PEAK_GBS, 4), "valu_weighted_frac": sig(weighted, 4),
               "issue_cycles_per_valu_inst": sig(cpi, 4)}
    return out


def c3_latency_floor(mhx, ctx, w, st):
    """What bounds C3 is not bandwidth (26.6 MB per sweep live in the 256 MiB Infinity Cache) but the launch's DEPENDENT CHAIN:
    kernel boundary -> partner draw -> row gather -> A y -> accept -> state and record stores visible to the next launch.  The floor
    is measured, not modelled: the SAME kernel form on the same target with 256 walkers (one or two blocks per half, every CU but a
    few idle) -- a sweep of it is the chain with no throughput term.  frac = that time / the time of a full-size sweep."""
    import numpy as np
    fused = st["launches"] <= w.inner
    opts = {"EMCEE_MFMA": "1" if st["kernel_variant"] == 10 else "0", "EMCEE_PERSIST": "0", "EMCEE_FUSED": "1" if fused else "0"}
    old = {k: ctx.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            ctx.set_option(k, v)
        a = type("A", (), {})()
        a.dim, a.chains, a.inner, a.lanes = w.d, 256, w.inner, w.lanes
        a.c3_rotated, a.c3_user, a.c3_small = w.rotated, w.user, False
        f = C3(a, w.dtype)
        f.build(mhx, ctx, 0)
        f.run.sample(*f.sched()[:4], save=True)
        ms = []
        for _ in range(5):
            f.run.sample(*f.sched()[:4], save=True)
            ms.append(f.run.stats()["kernel_ms"])
        fs = f.run.stats()
        f.run.close()
        if fs["kernel_variant"] != st["kernel_variant"] or fs["reduce_lanes"] != st["reduce_lanes"]:
            return None
        return float(np.median(ms)) * 1e3 / w.inner               # microseconds per sweep
    finally:
        for k, v in old.items():
            ctx.set_option(k, v or None)


def other_configs(mhx, ctx, args, barrier):
    """The other BASELINE.json configs in the same driver run (N = 1, rank 0, after the headline's timed region), one compact
    block each: value, ms_per_step, acceptance, bound / frac of the dominant kernel, PMC traffic over algorithmic bytes, launch
    time, and the CPU baseline (2 s samples).  What each key is: DESIGN.md section 7."""
    import copy
    plan = [("c1", "c1", {}, 3, 1), ("c2_literal", "c2", {"c2_literal": True}, 20, 10), ("c2_user", "c2", {"c2_user": True}, 10, 5),
            ("c3", "c3", {}, 10, 10), ("c3_rotated", "c3", {"c3_rotated": True}, 10, 10), ("c3_user", "c3", {"c3_user": True}, 10, 10),
            # emcee at the sizes it is used at: 256 ensembles x 1000 walkers in one run (a persistent block per ensemble), and one alone
            ("c3_small", "c3", {"c3_small": True}, 5, 3), ("c3_small_one", "c3", {"c3_small": True, "ensembles": 1}, 5, 3),
            ("c4", "c4", {}, 3, 2),
            ("c4_moving", "c4", {"c4_moving": True}, 3, 2), ("c4_fixed", "c4", {"c4_fixed": True}, 3, 2),
            ("c4_deferred", "c4", {"c4_moving": True, "c4_deferred": True}, 3, 2),
            ("c5", "c5", {}, 10, 10), ("c5_banana", "c5", {"c5_banana": True}, 10, 10)]
    res = {}
    for key, name, over, steps, spin in plan:
        try:
            a = copy.copy(args)
            a.inner = a.chains = a.dim = a.lanes = 0
            a.c4_moving = a.c4_fixed = a.c4_deferred = a.c3_rotated = a.c5_banana = a.c2_literal = a.c2_user = a.c3_user = a.c3_small = False
            a.ensembles = 0
            for k, v in over.items():
                setattr(a, k, v)
            w = WORKLOADS[name](a, args.dtype)
            w.build(mhx, ctx, 0)
            dt, kms, acc, tr, st = timed(w, steps, 2, barrier, spin=spin)
            rf = roofline_block(w, key, args.dtype, kms, steps, st)
            blk = {"value": sig(w.units_per_step() * steps / dt), "ms_per_step": sig(dt * 1e3 / steps), "acc": sig(acc / float(tr), 3),
                   "bound": rf["bound"], "frac": sig(rf["frac"], 4),
                   "traffic_ratio": sig(rf["traffic"] / rf["algorithmic_bytes_per_step"], 4) if rf.get("traffic") else None,
                   "launch_us": sig(rf["avg_launch_ms"] * 1e3, 4), "kernel": kernel_name(w, st), "lanes": st["reduce_lanes"],
                   # which generator turned the stream bits into normals (mhx_stats.normal_gen; the stretch move draws none)
                   "gen": "-" if name == "c3" else ("zig" if st.get("normal_gen") else "bm")}
            if rf["bound"] == "valu":
                blk["hbm_frac"] = rf["hbm_frac"]
                blk["valu_weighted_frac"] = rf.get("valu_weighted_frac")     # class-weighted issue bound (tools/isa_mix.py), None until counted
            if st.get("factor_band", -1) >= 0:
                blk["band"] = st["factor_band"]
            if name == "c3" and w.E == 1:
                # the honest ruler for one ensemble per launch: the dependent chain (c3_latency_floor), not HBM
                fl = c3_latency_floor(mhx, ctx, w, st)
                if fl:
                    sweep_us = dt * 1e6 / (steps * w.inner)
                    blk.update({"bound": "latency", "hbm_frac": blk["frac"], "floor_us": sig(fl, 4), "sweep_us": sig(sweep_us, 4),
                                "frac": sig(fl / sweep_us, 4)})
            elif name == "c3":
                blk["ensembles"] = w.E
            if key == "c3":
                # the same sweeps with every walker's state back on the host (what `sample` returns): the accept-compacted return
                # path at the stretch move's acceptance -- moves/s through the boundary, tensor and wire bytes per step
                try:
                    import numpy as np
                    hb = mhx.host_array((w.inner, w.d + 1, w.W * w.E), w.run.real)
                    ha = mhx.host_array((w.inner, w.W * w.E), np.uint8)
                    w.run.sample_to_host(w.inner, 1, 1, 0, out=hb, out_accepted=ha)
                    t0 = time.perf_counter()
                    for _ in range(3):
                        w.run.sample_to_host(w.inner, 1, 1, 0, out=hb, out_accepted=ha)
                    wt = (time.perf_counter() - t0) / 3
                    hs = w.run.host_stats()
                    blk["e2e_save_all"] = {"value": sig(w.units_per_step() / wt), "host_GB": sig((hb.nbytes + ha.nbytes) / 1e9, 4),
                                           "wire_GB": sig(hs["wire_bytes"] / 1e9, 4), "compact": hs["compact"]}
                    del hb, ha
                except Exception as e:
                    blk["e2e_save_all"] = {"error": str(e)[:80]}
            w.run.close()
            if not args.no_cpu_baseline:
                blk["cpu"] = cpu_baseline(w, args.dtype, 2.0, compact=True)
            res[key] = blk
        except Exception as e:                                   # one config must not take the line down
            res[key] = {"error": str(e)[:160]}
    return res


def e2e_host(mhx, wl):
    """"kernel time AND end-to-end including D2H of whatever is kept, both reported".  The reference's `sample`
    returns a host container; mhx_run_sample_to_host streams the samples into page-locked host memory while the chains run.
    Three figures on the headline workload: save_all -- every state of `inner` transitions to the host (the PCIe link is the
    bound: B(d+1)+1 bytes per chain-step); thinned -- the ESS window's schedule, which returns at the kernel rate; round2_path --
    mhx_run_sample, then one synchronous mhx_run_get_samples into pageable memory."""
    import numpy as np
    run, d, C, inner = wl.run, wl.d, wl.C, wl.inner
    out = {}
    t0 = time.perf_counter()
    buf = mhx.host_array((inner, d + 1, C), run.real)
    accb = mhx.host_array((inner, C), np.uint8)
    out["pinned_alloc_s"] = sig(time.perf_counter() - t0, 3)
    def timed(label, out_buf, out_acc, reps=3):
        run.sample_to_host(inner, 1, 1, 0, out=out_buf, out_accepted=out_acc)   # untimed: first touch of the slabs, staging and streams
        t0 = time.perf_counter()
        hs = None
        for _ in range(reps):
            run.sample_to_host(inner, 1, 1, 0, out=out_buf, out_accepted=out_acc)
            h = run.host_stats()
            hs = h if hs is None else {k: (hs[k] + h[k] if k in ("wire_bytes", "link_ms", "expand_ms") else h[k]) for k in h}
        w = (time.perf_counter() - t0) / reps
        nbytes = out_buf.nbytes + out_acc.nbytes
        blk = {"value": sig(C * inner / w), "wall_s": sig(w, 4), "host_GB": sig(nbytes / 1e9, 4), "kernel_ms": sig(run.stats()["kernel_ms"], 4),
               "compact": hs["compact"], "wire_GB": sig(hs["wire_bytes"] / reps / 1e9, 4)}
        if hs["compact"]:
            # link_GBps: the blocks over the time their copies held the link; host_expand_GBps: tensor bytes the host threads wrote
            # over the time they were at it (first worker in to last worker out, per block); tensor_GBps: what the caller sees
            blk.update({"link_GBps": sig(hs["wire_bytes"] / max(hs["link_ms"], 1e-9) / 1e6, 4),
                        "host_expand_GBps": sig(nbytes * reps / max(hs["expand_ms"], 1e-9) / 1e6, 4),
                        "tensor_GBps": sig(nbytes / w / 1e9, 4), "host_threads": hs["threads"], "slabs": hs["slabs"]})
        else:
            blk["link_GBps"] = sig(nbytes / w / 1e9, 4)
        out[label] = blk
    # save_all: the boundary's default for a save-all run -- accept-compacted blocks over the link, host threads rebuild the tensor
    timed("save_all", buf, accb)
    # ... into plain pageable memory (a numpy / Julia array, touched once before): nothing is page-locked on this path
    try:
        pb = np.empty((inner, d + 1, C), run.real)
        pa = np.empty((inner, C), np.uint8)
        pb[:] = 0
        pa[:] = 0
        timed("save_all_pageable", pb, pa, reps=2)
        del pb, pa
    except MemoryError as e:
        out["save_all_pageable"] = {"error": str(e)[:80]}
    # ... into a tensor allocated INSIDE the timed region, as a one-shot `sample` call does: the kernel's zeroing of 13 GB of fresh
    # pages is part of it (the library asks for huge pages on the range; the expanding threads fault them in)
    try:
        t0 = time.perf_counter()
        fb = np.empty((inner, d + 1, C), run.real)
        fa = np.empty((inner, C), np.uint8)
        run.sample_to_host(inner, 1, 1, 0, out=fb, out_accepted=fa)
        w = time.perf_counter() - t0
        hs = run.host_stats()
        out["save_all_fresh"] = {"value": sig(C * inner / w), "wall_s": sig(w, 4), "compact": hs["compact"],
                                 "host_expand_GBps": sig((fb.nbytes + fa.nbytes) / max(hs["expand_ms"], 1e-9) / 1e6, 4)}
        del fb, fa
    except MemoryError as e:
        out["save_all_fresh"] = {"error": str(e)[:80]}
    # ... and the plain path (every row over the link, rounds 3-5): the PCIe link is the bound, B(d+1)+1 bytes per chain-step
    run.ctx.set_option("HOST_COMPACT", "0")
    try:
        timed("save_all_plain", buf, accb, reps=2)
    finally:
        run.ctx.set_option("HOST_COMPACT", None)
    thin = max(1, int(round(40 * (d / 0.3) / 256)))
    n_draws = 256
    tb = mhx.host_array((n_draws, d + 1, C), run.real)
    ta = mhx.host_array((n_draws, C), np.uint8)
    run.sample_to_host(n_draws, thin, thin, 0, out=tb, out_accepted=ta)
    t0 = time.perf_counter()
    run.sample_to_host(n_draws, thin, thin, 0, out=tb, out_accepted=ta)
    w = time.perf_counter() - t0
    out["thinned"] = {"value": sig(C * n_draws * thin / w), "wall_s": sig(w, 4), "draws": n_draws, "thin": thin,
                      "host_GB": sig((tb.nbytes + ta.nbytes) / 1e9, 4), "kernel_ms": sig(run.stats()["kernel_ms"], 4)}
    del tb, ta
    t0 = time.perf_counter()
    run.sample(inner, 1, 1, 0, save=True)
    old, _ = run.samples()
    w = time.perf_counter() - t0
    out["round2_path"] = {"value": sig(C * inner / w), "wall_s": sig(w, 4)}
    del old
    return out


# ------------------------------------------------------------------------------------------------------------------
# launching: --gpus N without a launcher starts N ranks of this script; a rank refuses a command line that disagrees with it


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_devices():
    import torch
    return torch.cuda.device_count()


def launch_ranks(args):
    """This process is the launcher (WORLD_SIZE unset, --gpus N > 1): N children of the same command line, one per GPU, rank 0's
    stdout is ours.  Each child is bound to ITS device before HIP initialises: HIP_VISIBLE_DEVICES = the r-th entry of the list
    this process sees (--bind visible, the default; --bind ordinal leaves every device visible and the rank takes LOCAL_RANK like a
    torchrun rank does).  Exit code: the first non-zero child's.  A failing rank takes the others down (they would wait in a
    collective until its deadline)."""
    n = args.gpus
    have = 0
    if not args.dry_run:
        have = visible_devices()
        if have < n and not args.allow_gloo:
            sys.stderr.write("bench.py --gpus %d: this box has %d GPU(s); refusing to print a %d-GPU line (the one-GPU rehearsal of "
                             "the %d-rank flow is --allow-gloo, the device-free one --dry-run)\n" % (n, have, n, n))
            return 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               MHX_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what RCCL needs between processes on this driver
    outer = [v for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v != ""]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        if args.bind == "visible" and have > 0:
            slot = r % have                                        # (several ranks per device only in the --allow-gloo rehearsal)
            e["HIP_VISIBLE_DEVICES"] = outer[slot] if slot < len(outer) else str(slot)
            e["MHX_BENCH_BOUND"] = "1"                             # the rank sees ONE device: ordinal 0
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            c = p.poll()
            if c is None:
                continue
            alive.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in alive:                                   # our own children, by pid
                    q.terminate()
    return rc


def leave(code=0):
    """end of a rank: through os._exit when a transport was abandoned in a hung library call (its thread cannot be joined)"""
    sys.stdout.flush()
    sys.stderr.flush()
    if ABANDONED:
        os._exit(code)
    sys.exit(code)


def dry_run(args, rank, world):
    """--dry-run: the N-rank flow with no device and no engine -- rendezvous, the transport ladder (it ends on gloo: there is no
    device), barrier, max-over-ranks time and the summed totals; each rank's "step" is a counter.  Proves on a CPU box that
    `--gpus N` really is N processes and that a failing / hanging RCCL init (--fault-rccl-init) still ends in a line."""
    import numpy as np
    comm, info = (None, {}) if world == 1 else make_collective(None, rank, world, args)

    def barrier():
        if comm is not None:
            comm.allreduce_sum(np.zeros(1))
    barrier()
    t0 = time.perf_counter()
    units = 0
    for _ in range(args.steps):
        units += 1000
    barrier()
    dt = time.perf_counter() - t0
    ranks, total, pids = 1, float(units), [os.getpid()]
    if comm is not None:
        t = np.zeros(world)
        t[rank] = dt
        dt = float(comm.allreduce_sum(t).max())
        v = comm.allreduce_sum(np.array([1.0, float(units)]))
        ranks, total = int(round(v[0])), float(v[1])
        p = np.zeros(world)
        p[rank] = os.getpid()
        pids = [int(x) for x in comm.allreduce_sum(p)]
    if rank == 0:
        cfg = {"workload": "dry run of the %d-rank flow" % world, "name": args.config,
               "collective": info.get("collective", "none (1 rank)") + " (dry run)"}
        for k in ("ranks_reported_by_transport", "rccl_error", "transport_errors"):
            if info.get(k) is not None:
                cfg[k] = info[k]
        print(json.dumps({"metric": "MH steps/sec (all chains) + ESS/sec", "value": None, "unit": "MH steps/s", "dry_run": True,
                          "n_gpus": 0, "n_ranks": ranks, "distinct_processes": len(set(pids)), "steps": args.steps, "warmup": args.warmup,
                          "units_all_ranks": total, "ms_per_step": dt * 1e3 / max(1, args.steps), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "none (dry run: no device, no engine)",
                          "config": cfg}), flush=True)
    if comm is not None:
        barrier()
        comm.dist.destroy_process_group()
    return 0


class GroupWorkload:
    """--single-process: N member contexts of ONE process behind mhx_group_* (N host threads inside libmhx.so); member i builds the
    workload of rank i, the group drives them side by side.  Same interface as a workload for timed()."""

    def __init__(self, mhx, args, dtype, devices):
        self.g = mhx.Group(devices, dtype)
        for k, v in args.opt:
            self.g.set_option(k, v)
        self.members = []
        for i, ctx in enumerate(self.g.ctxs):
            w = WORKLOADS[args.config](args, dtype)
            w.build(mhx, ctx, i)
            self.members.append(w)
        self.g.attach([w.run for w in self.members])
        self.first = self.members[0]
        self.name, self.d = self.first.name, self.first.d

    def step(self):
        sc = self.first.sched()
        self.g.sample(*sc[:4], save=sc[4])
        return self.g.stats()

    def units_per_step(self):
        return sum(w.units_per_step() for w in self.members)

    def bytes_per_launch(self):
        return self.first.bytes_per_launch()          # per member: the roofline prices ONE device's kernel

    def describe(self):
        return self.first.describe()


def rhat_block(diag, d, chains, window, transport):
    """configs[4]: the R-hat of the whole sharded run from the all-reduced (or host-summed) between / within sums"""
    import numpy as np
    rh = np.asarray(diag["rhat"][:d], dtype=np.float64)
    return {"max": sig(float(np.nanmax(rh))), "median": sig(float(np.nanmedian(rh))), "chains": int(chains),
            "draws_per_chain": int(diag["n_samples"]), "window": window, "reduced_over": transport}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5, help="untimed launches right before the timed ones")
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="c2", help="BASELINE.json workload (c2 = the headline)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64", help="arithmetic of the engine (the reference is Float64)")
    ap.add_argument("--inner", type=int, default=0, help="transitions (sweeps) per chain per step (launch); 0 = the config's default")
    ap.add_argument("--chains", type=int, default=0, help="chains (walkers) per GPU; 0 = the config's default")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per chain (0 = engine's choice)")
    ap.add_argument("--c2-literal", action="store_true", help="c2: proposal N(0, I) as the config text reads (acceptance ~ 0 at d = 100) "
                    "instead of the tuned 2.38/sqrt(d)")
    ap.add_argument("--c2-user", action="store_true", help="c2: the target as a user log-density in HIP source (DensityModel(f), JIT-lowered, one lane per chain)")
    ap.add_argument("--c3-user", action="store_true", help="c3: the AR(1) target as a user log-density in HIP source (lane per walker, any-target kernel)")
    ap.add_argument("--c3-small", action="store_true", help="c3: d = 5, 1 000 walkers (test/emcee.jl:24) x --ensembles ensembles in ONE run (mhx_emcee_cfg.n_ensembles)")
    ap.add_argument("--ensembles", type=int, default=0, help="c3 --c3-small: ensembles in the run (default 256)")
    ap.add_argument("--c3-rotated", action="store_true", help="c3: the dense-rotated variant Sigma = Q (0.9^|i-j|) Q^T (no banded factor)")
    ap.add_argument("--c4-fixed", action="store_true", help="c4: the fixed-factor steps that follow the warm-up (1 read of S per step)")
    ap.add_argument("--c5-banana", action="store_true", help="c5: the banana target of SURVEY 8(d) (ii) instead of Neal's funnel")
    ap.add_argument("--c4-deferred", action="store_true", help="c4: MHX_FLAG_RAM_DEFERRED (its own rounding and byte model: 10/8 passes over S per step)")
    ap.add_argument("--c4-moving", action="store_true", help="c4: random start and S0 = 2.38/sqrt(d) I instead of x0 = 0, S0 = I")
    ap.add_argument("--normal-gen", choices=["auto", "ziggurat", "box-muller"], default="auto",
                    help="c2 / c5: how the RWMH kernel turns stream bits into standard normals (auto: ziggurat in fp64, Box-Muller in fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-gloo", action="store_true", help="--gpus > 1: the one-GPU rehearsal -- fewer devices than ranks / members are "
                    "accepted (they share devices; n_gpus says how many were really opened, the line says `oversubscribed`)")
    ap.add_argument("--require-rccl", action="store_true", help="--gpus > 1: no transport ladder -- if the RCCL communicator behind the C ABI "
                    "cannot be created on every rank the run fails instead of stepping down to torch's nccl backend / gloo")
    ap.add_argument("--single-process", action="store_true", help="--gpus N as ONE process: N member contexts and N host threads behind the "
                    "C ABI (mhx_group_*), statistics summed on the host -- no launcher, no collective library")
    ap.add_argument("--bind", choices=["visible", "ordinal"], default="visible", help="our own launcher: visible = each rank sees only its "
                    "device (HIP_VISIBLE_DEVICES, set before HIP initialises); ordinal = all devices visible, device LOCAL_RANK")
    ap.add_argument("--transport-timeout", type=float, default=120.0, help="seconds a rung of the transport ladder may take (init + test all-reduce)")
    ap.add_argument("--rendezvous-timeout", type=float, default=300.0, help="seconds to wait for every rank at the rendezvous")
    ap.add_argument("--fault-rccl-init", choices=["fail", "hang"], default=None, help="test hook of the BENCH (not of the library): the first "
                    "rung of the ladder fails / never answers")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="explicit engine option (mhx_ctx_set_option), repeatable")
    ap.add_argument("--lib", default="", help="bind this build of the library (tools/build_variant.sh: A/B experiments on compile-time tuning macros)")
    ap.add_argument("--tools-lib", action="store_true", help="bind libmhx_tools.so (timing probes, fault injection: --opt EMCEE_PROBE=3 ...); "
                    "a run with a probe option is tainted and the line says so")
    ap.add_argument("--dry-run", action="store_true", help="launcher + rendezvous + ladder + host-side combining only: no device, no engine")
    ap.add_argument("--no-other-configs", action="store_true", help="c2 at N=1: skip the C3 / C4 / C4-moving / C5 lines under `configs`")
    ap.add_argument("--no-e2e", action="store_true", help="c2 at N=1: skip the rate through the boundary (samples back on the host)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="size of the CPU-baseline sample of the headline config")
    ap.add_argument("--no-second-dtype", action="store_true", help="skip the fp32 figure")
    ap.add_argument("--no-ess", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    args.opt = [tuple(o.split("=", 1)) if "=" in o else (o, "1") for o in args.opt]

    single = args.single_process and args.gpus > 1
    if "WORLD_SIZE" not in os.environ or single:
        if single and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            sys.stderr.write("bench.py --single-process under a launcher with WORLD_SIZE=%s: refusing (one process is the point)\n" % os.environ["WORLD_SIZE"])
            sys.exit(2)
        if args.gpus > 1 and not single:
            sys.exit(launch_ranks(args))                          # this process is the launcher
        rank, local_rank, world = 0, 0, 1
    else:
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:                                   # never print an N-GPU line from a different number of ranks
            sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to run\n" % (args.gpus, world))
            sys.exit(2)
    if args.dry_run:
        rc = dry_run(args, rank, world)
        leave(rc)

    import numpy as np
    import torch
    import mhx
    from mhx.dist import allreduce_stats
    if args.tools_lib:
        mhx.use_library(mhx.TOOLS_LIB_PATH)
    if args.lib:
        mhx.use_library(args.lib)

    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.stderr.write("bench.py: no GPU visible (the engine has no CPU path); --dry-run rehearses the launch without one\n")
        sys.exit(2)
    bound = os.environ.get("MHX_BENCH_BOUND") == "1"             # our launcher made this rank's device the only visible one
    need = args.gpus if single else (1 if bound else world)
    if ndev < need and not args.allow_gloo:
        sys.stderr.write("bench.py --gpus %d: %d device(s) visible to rank %d; refusing (ranks would share a GPU). --allow-gloo accepts it "
                         "as a rehearsal\n" % (args.gpus, ndev, rank))
        sys.exit(2)
    device = 0 if bound else local_rank % ndev                   # (several ranks on one device only in the --allow-gloo rehearsal)
    torch.cuda.set_device(device)
    ctx = mhx.Context(device, args.dtype)
    for k, v in args.opt:
        ctx.set_option(k, v)
    comm, info = None, {}
    collective = None
    if world > 1 or os.environ.get("MHX_BENCH_FORCE_DIST"):      # the env knob exercises the collective path on 1 GPU
        try:
            comm, info = make_collective(ctx, rank, world, args, device)
            collective = info["collective"]
        except Exception as e:
            if args.require_rccl:
                raise
            # not even the rendezvous: this rank reports alone rather than not at all
            info = {"collective": "none: %s" % str(e)[:200], "ranks_reported_by_transport": 1, "rendezvous_error": str(e)[:300]}
            collective = info["collective"]

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.allreduce_sum(np.zeros(1))                      # every rank has arrived
            torch.cuda.synchronize()

    if single:
        devs = [i % ndev for i in range(args.gpus)]
        wl = GroupWorkload(mhx, args, args.dtype, devs)
        bus = wl.g.pci_bus_ids()
        n_gpus = len(set(bus))
        collective = "single process: %d host threads behind mhx_group_*, host sum of the statistics" % args.gpus
    else:
        # the devices the ranks REALLY opened, by PCI bus id (ordinals depend on HIP_VISIBLE_DEVICES): n_gpus is their number
        n_gpus = 1
        if comm is not None:
            ids = np.zeros(world)
            ids[rank] = float(pci_number(ctx.pci_bus_id()))
            n_gpus = len(set(int(round(v)) for v in comm.allreduce_sum(ids)))
            if n_gpus != world and not args.allow_gloo:
                sys.stderr.write("bench.py --gpus %d: the %d ranks opened %d distinct device(s)\n" % (args.gpus, world, n_gpus))
                leave(2)
        wl = WORKLOADS[args.config](args, args.dtype)
        if world > 1 and getattr(wl, "C", 0):
            # a rank's chains are a shard of world x C: the engine picks the kernel form (the summation order) for the whole run's
            # count, so the ranks' union is the unsharded run bit for bit (include/mhx.h: option TOTAL_CHAINS)
            ctx.set_option("TOTAL_CHAINS", wl.C * world)
        wl.build(mhx, ctx, rank)
    dt, kernel_ms, accepted, transitions, st = timed(wl, args.steps, args.warmup, barrier)

    # outside the timed region: diagnostics, the ESS window, the fp32 figure, the CPU baseline
    ess, diag = None, None
    chains_all = None
    if args.config in ("c2", "c5"):
        if single:
            diag = wl.g.diagnostics(max_lag=0)                   # the members' sums added on the host
            chains_all = diag["n_chains"]
        else:
            diag = wl.run.diagnostics(max_lag=0)
    if comm is not None:
        t = np.zeros(world)
        t[rank] = dt
        dt = float(comm.allreduce_sum(t).max())                  # MAX over ranks of the timed region
        if diag is not None:
            diag = allreduce_stats(diag, accepted, transitions, comm=comm)       # ONE all-reduce of 3(d+1)+3 doubles
            acc_rate = diag["acceptance_rate"]
            chains_all = diag["n_chains"]
        else:
            v = comm.allreduce_sum(np.array([float(accepted), float(transitions)]))
            acc_rate = v[0] / v[1]
    else:
        acc_rate = accepted / float(transitions)
    lone = world == 1 and not single
    if args.config == "c2" and not args.no_ess and not args.c2_literal and not args.c2_user and rank == 0 and lone:
        try:
            ess = ess_window(mhx, wl, world)
        except Exception as e:                                   # never let a diagnostic break the bench line
            ess = {"error": str(e)[:160]}

    if rank == 0:
        reporting = info.get("ranks_reported_by_transport", 1) if comm is None and world > 1 else world
        units = float(wl.units_per_step()) * args.steps * (1 if single else reporting)
        value = units / dt
        out = {
            "metric": "MH steps/sec (all chains) + ESS/sec", "value": value, "unit": "MH steps/s",
            "n_gpus": n_gpus, "n_ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": config_block(wl.first if single else wl, args.config, st, collective),
            "acceptance_rate": sig(acc_rate, 4),
            "roofline": roofline_block(wl, args.config, args.dtype, kernel_ms, args.steps, st),
        }
        if single:
            out["n_members"] = args.gpus
            out["config"]["members"] = "%d member contexts on devices %s (PCI %s)" % (args.gpus, devs, sorted(set(bus)))
            if n_gpus != args.gpus:
                out["oversubscribed"] = "%d members share %d device(s): a rehearsal of the flow, not a scaling point" % (args.gpus, n_gpus)
        if world > 1:
            for k in ("ranks_reported_by_transport", "rccl_error", "transport_errors", "rendezvous_error"):
                if info.get(k) is not None:
                    out["config"][k] = info[k]
            out["config"]["binding"] = "HIP_VISIBLE_DEVICES per rank (set by the launcher before HIP initialised)" if bound else "all devices visible, device = LOCAL_RANK"
            if comm is None:
                out["ranks_in_value"] = 1
            if n_gpus != world and comm is not None:
                out["oversubscribed"] = "%d ranks share %d device(s): a rehearsal of the flow, not a scaling point" % (world, n_gpus)
        if args.opt:
            out["config"]["options"] = dict(args.opt)
        try:                                               # who builds the run-time kernels: the installation's clang++, or hiprtc
            cid, next_ = ctx.jit_compiler()                # (inside Python: the PyTorch wheel's older copy of it; include/mhx.h)
            out["config"]["jit_compiler"] = (os.path.basename(cid.split(":")[1]) + " of " + cid.split(":")[1].split("/lib/llvm")[0]) if cid else "hiprtc"
        except Exception:
            pass
        if st.get("tainted"):
            out["tainted"] = "a probe option of the tools build was set: NOT a valid measurement of the chains"
        if ess is not None:
            out["ess_per_sec"] = ess.get("ess_per_sec")
            out["ess"] = ess
        if diag is not None:
            if args.config == "c5":
                # configs[4]: the R-hat of the whole sharded run (262 144 chains at 8 x 32 768), reduced over the transport above
                sc = (wl.first if single else wl).sched()
                out["rhat"] = rhat_block(diag, wl.d, chains_all or diag["n_chains"],
                                         "last timed launch: %d states per chain, %d transitions apart" % (sc[0], sc[2]), collective or "one rank")
            # the all-reduced between / within statistic of the LAST timed launch only (consecutive transitions, shorter than one
            # autocorrelation time at d = 100): it exercises the collective, it is not a convergence claim -- see ess.rhat_max_split
            out["rhat_last_launch"] = sig(float(np.nanmax(diag["rhat"][:wl.d])))
        if lone and args.config == "c2" and not args.no_e2e and not args.c2_literal and not args.c2_user:
            try:
                out["e2e_host"] = e2e_host(mhx, wl)
            except Exception as e:
                out["e2e_host"] = {"error": str(e)[:160]}
        if lone and not args.no_second_dtype and args.dtype == "f64":
            try:                                                  # the same workload on the fp32 engine: a second figure, never `value`
                ctx32 = mhx.Context(device, "f32")
                for k, v in args.opt:
                    ctx32.set_option(k, v)
                wl32 = WORKLOADS[args.config](args, "f32")
                wl32.build(mhx, ctx32, rank)
                n32 = max(5, args.steps // 2)
                dt32, k32, a32, t32, st32 = timed(wl32, n32, 2, barrier)
                out["f32"] = {"value": sig(wl32.units_per_step() * n32 / dt32), "ms_per_step": sig(dt32 * 1e3 / n32),
                              "acc": sig(a32 / float(t32), 3), "lanes": st32["reduce_lanes"],
                              "frac": sig(wl32.bytes_per_launch() / (k32 * 1e-3 / n32) / 1e9 / HBM_PEAK_GBS, 4)}
                rf32 = roofline_block(wl32, args.config, "f32", k32, n32, st32)      # (PMC counts of the fp32 kernel, where taken)
                for k in ("valu_frac", "valu_weighted_frac"):
                    if rf32.get(k) is not None:
                        out["f32"][k] = rf32[k]
                wl32.run.close()
            except Exception as e:
                out["f32"] = {"error": str(e)[:160]}
        if lone and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.dtype, args.cpu_seconds)
        if lone and args.config == "c2" and not args.no_other_configs and not args.c2_literal and not args.c2_user:
            wl.run.close()                                        # 13.4 GB of samples: C4 needs the room
            out["configs"] = other_configs(mhx, ctx, args, barrier)
        try:                                                      # run-time kernels of the whole run: built here / by clang++ / from the disk cache
            comp, hits = ctx.jit_counts()
            out["config"]["jit"] = {"compiled": comp, "by_clang": ctx.jit_compiler()[1], "from_cache": hits}
        except Exception:
            pass
        line = json.dumps(out, separators=(",", ":"))
        if len(line) >= 8000:                                     # the driver keeps an 8 KB tail: drop detail before the contract keys go
            for k in ("ess", "e2e_host"):
                out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
        # RCCL writes its version banner to the C stdout buffer: push it out first so the JSON line is the last one
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)
    if comm is not None:
        try:
            barrier()
            comm.close()
        except Exception:
            pass
        import torch.distributed as dist
        if dist.is_initialized() and not ABANDONED:
            dist.destroy_process_group()
    leave(0)


if __name__ == "__main__":
    main()
