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

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against HBM with the algorithmic bytes of
DESIGN.md section 7 and the HIP-event launch time measured here; `cpu_baseline` is the CPU oracle (a port of the
reference algorithm, oracle/, rebuilt -O3 -march=native on the host that times it) on a bounded sample -- rank 0, N=1 only.
At N=1 the default (c2) run also carries, all measured after the timed region:
  `e2e_host`  the rate THROUGH the boundary -- samples back on the host (mhx_run_sample_to_host), save-all and thinned;
  `configs`   the other BASELINE.json GPU configs and their SURVEY 8(d) variants (c1, c2_literal, c3, c3_rotated, c4,
              c4_moving, c4_fixed, c5, c5_banana): value, roofline and cpu_baseline each;
  `ess`       the ESS/sec window;  `f32` the fp32 engine on the same workload.
"""
import argparse
import json
import os
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
VARIANTS = {0: "generic", 1: "prebuilt-register", 2: "hiprtc-register", 3: "prebuilt-cooperative",
            4: "hiprtc-cooperative", 5: "hiprtc-dense-cooperative", 6: "persistent-ensemble",
            7: "sequential-ensemble-sweep (the reference's Gauss-Seidel order)", 8: "matrix-core (v_mfma_*_16x16x4, shared dense factor)"}
RB = {"f32": 4, "f64": 8}


GEN_TEXT = {None: "Box-Muller", "box-muller": "Box-Muller",
            "ziggurat": "the table ziggurat of the fp64 spec (MHX_FLAG_ZIGGURAT: 1024 layers, 64 bits per normal, exact rejection sampling)"}


def pick_gen(args, dtype):
    """RWMH on the cooperative kernel: the ziggurat generator in fp64 (what Julia's own randn is), Box-Muller in fp32 (the fp64
    engine alone has the ziggurat); --normal-gen overrides."""
    g = getattr(args, "normal_gen", "auto")
    if g == "auto":
        return "ziggurat" if dtype == "f64" else None
    return None if g == "box-muller" or dtype != "f64" else g


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


def host_cores():
    """the cores this process may run on (a container's cpuset can be smaller than os.cpu_count())"""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except Exception:
        return os.cpu_count() or 1


def threads_rate(work, cores, target_seconds):
    """work(thread_index, nchains) on `cores` threads, chains per thread sized to ~target_seconds of wall time;
    returns (chains in total, seconds)"""
    import concurrent.futures as cf

    def pool(n):
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda i: work(i, n), range(cores)))
        return time.perf_counter() - t0

    pool(1)                                                  # thread start-up, page faults
    t1 = pool(2)                                             # calibration: 2 chains per thread
    n = max(2, int(2 * target_seconds / t1))
    return cores * n, pool(n)


# ------------------------------------------------------------------------------------------------------------------
# workloads: each builds a Run for this rank and says what one step is, what it moves and how the CPU oracle does it


class C2:
    """BASELINE configs[1]: isotropic 100-dim standard MvNormal, RWMH, 65 536 chains per GPU, every state recorded"""
    name = "c2"

    def __init__(self, args, dtype):
        self.d, self.C, self.inner, self.dtype = args.dim or 100, args.chains or 65536, args.inner or 250, dtype
        self.lanes = args.lanes
        self.literal = getattr(args, "c2_literal", False)
        self.gen = pick_gen(args, dtype)

    def build(self, mhx, ctx, rank):
        import numpy as np
        d = self.d
        self.s = 1.0 if self.literal else float(np.float32(2.38 / d ** 0.5))
        model = mhx.DensityModel(mhx.IsoGaussian(d))
        spl = mhx.RWMH(mhx.MvNormal(mhx.zeros(d), self.s * self.s * mhx.I))
        self.run = mhx.Run(model, spl, nchains=self.C, seed=0xC0FFEE, first_chain=rank * self.C, ctx=ctx, reduce_lanes=self.lanes,
                           normal_gen=self.gen)
        self.run.init(None)                                # x0 ~ proposal draw (src/mh-core.jl:83), on the device
        return self.run

    def step(self):
        # N = inner saved samples, the first one being the state after 1 transition: inner transitions
        self.run.sample(self.inner, 1, 1, 0, save=True)
        return self.run.stats()

    def units_per_step(self):
        return self.C * self.inner

    def bytes_per_launch(self):
        """sample record B(d+1)+1 per chain-step (save-all) + one state round trip per launch"""
        B = RB[self.dtype]
        return self.C * (self.inner * (B * (self.d + 1) + 1) + 2 * (B * self.d + B + 4 + 1))

    def bytes_model(self):
        return "record %d(d+1)+1 B per chain-step + state round trip per launch" % RB[self.dtype]

    def describe(self):
        return ("RWMH, isotropic %d-dim standard MvNormal target, %d chains per GPU, proposal %s, "
                "%d transitions per launch, every state recorded; standard normals by %s" % (
                    self.d, self.C, "N(0, I) (the literal RWMH(MvNormal(zeros(d), I)) of the config text)" if self.literal
                    else "N(0,(2.38/sqrt(d))^2 I)", self.inner, GEN_TEXT[self.gen]))

    def cpu_baseline(self, O, target_seconds):
        tgt = O.iso_gauss(self.d)
        prop = O.Proposal(O.PROP_ISO, self.s, normal_gen=1 if self.gen == "ziggurat" else 0)
        inner = self.inner

        def work(i, nchains):
            O.rwmh(tgt, prop, O.schedule(inner + 1), 0xC0FFEE, i * nchains, nchains, save=True)
        t0 = time.perf_counter()
        O.rwmh(tgt, prop, O.schedule(101), 0xC0FFEE, 0, 8, save=True)
        rate1 = 800 / (time.perf_counter() - t0)
        cores = host_cores()
        n, dt = threads_rate(work, cores, target_seconds)
        return n * inner, dt, cores, "%d chains x %d transitions of the same d=%d workload; single thread %.3g steps/s" % (n, inner, self.d, rate1)


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

    def step(self):
        self.run.sample(self.inner // 10, 10, 10, 0, save="moments")     # every 10th state folded into the moments
        return self.run.stats()

    def bytes_per_launch(self):
        """the state lives in registers for a launch: HBM sees one state round trip and one moments round trip per launch"""
        B = RB[self.dtype]
        return self.C * (2 * (B * self.d + B + 4 + 1) + 4 * B * (self.d + 1))

    def bytes_model(self):
        return "state + running-moments round trip per launch (the chain state never leaves the registers inside a launch)"

    def describe(self):
        return ("RWMH, 1000-dim %s, %d chains per GPU (global ids: shard of 8 x 32 768), proposal N(0,(2.38/sqrt(d))^2 I), "
                "%d transitions per launch, running moments of every 10th state, R-hat by one all-reduce; standard normals by %s" % (
                    "banana (b = 0.03 on N(0, diag(100, 1, ...)))" if self.banana else "Neal's funnel", self.C, self.inner, GEN_TEXT[self.gen]))

    def cpu_baseline(self, O, target_seconds):
        tgt = O.Target(O.TARGET_BANANA, self.d, params=[0.03]) if self.banana else O.Target(O.TARGET_FUNNEL, self.d)
        prop = O.Proposal(O.PROP_ISO, self.s, normal_gen=1 if self.gen == "ziggurat" else 0)
        inner = self.inner

        def work(i, nchains):
            O.rwmh(tgt, prop, O.schedule(1, inner), 5, i * nchains, nchains, save=False)
        cores = host_cores()
        n, dt = threads_rate(work, cores, target_seconds)
        return n * inner, dt, cores, "%d chains x %d transitions of the same d=%d %s" % (n, inner, self.d, "banana" if self.banana else "funnel")


class C3:
    """BASELINE configs[2]: emcee Ensemble(StretchProposal), 50-dim correlated Gaussian (rho = 0.9), 16 384 walkers"""
    name = "c3"

    def __init__(self, args, dtype):
        self.d, self.W, self.inner, self.dtype = args.dim or 50, args.chains or 16384, args.inner or 500, dtype
        self.lanes = args.lanes
        self.rotated = getattr(args, "c3_rotated", False)

    def build(self, mhx, ctx, rank):
        d = self.d
        import numpy as np
        self.Sig = sigma_ar1(d, 0.9)
        if self.rotated:                                   # "also a dense-rotated variant" -- no structural zeros in the factor
            Q, _ = np.linalg.qr(np.random.default_rng(50).normal(size=(d, d)))
            self.Sig = Q @ self.Sig @ Q.T
        model = mhx.DensityModel(mhx.CorrGaussian(self.Sig))
        spl = mhx.Ensemble(self.W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I)))
        self.run = mhx.Run(model, spl, seed=3, first_chain=rank, ctx=ctx, reduce_lanes=self.lanes)   # one ensemble per GPU (replicas)
        self.run.init(None)
        return self.run

    def step(self):
        self.run.sample(self.inner, 1, 1, 0, save=True)
        return self.run.stats()

    def units_per_step(self):
        return self.W * self.inner

    def bytes_per_launch(self):
        """per move read x_i, x_j, write x_i, lp r/w (3Bd + 2B) + record B(d+1)+1"""
        B = RB[self.dtype]
        return self.W * self.inner * (3 * B * self.d + 2 * B + B * (self.d + 1) + 1)

    def bytes_model(self):
        B = RB[self.dtype]
        return "per move %d d + %d (x_i, x_j, x_i', lp) + record %d(d+1)+1" % (3 * B, 2 * B, B)

    def describe(self):
        band = self.run.stats().get("factor_band", -1) if hasattr(self, "run") else -1
        return ("emcee Ensemble(StretchProposal a=2), %d-dim Gaussian %s, %d walkers (one ensemble per GPU), "
                "%d sweeps per launch, every sweep recorded; parallel half-split sweep; %s" % (
                    self.d, "Sigma = Q (0.9^|i-j|) Q^T, Q a seeded rotation (dense precision factor)" if self.rotated else "Sigma_ij = 0.9^|i-j|",
                    self.W, self.inner,
                    "the precision factor inv(chol Sigma) is banded (bandwidth %d, detected): the row products skip its structural zeros" % band
                    if band >= 0 else "dense precision factor: d(d+1)/2 products per move"))

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
        return W * sweeps, dt, 1, "%d sequential sweeps (src/emcee.jl:39-58: inherently serial in W) of the same %d-walker ensemble, one thread" % (sweeps, W)


class C4:
    """BASELINE configs[3]: RobustAdaptiveMetropolis, 200-dim ill-conditioned Gaussian (kappa = 1e3), 32 768 chains, adapting"""
    name = "c4"

    def __init__(self, args, dtype):
        self.d, self.C, self.inner, self.dtype = args.dim or 200, args.chains or 32768, args.inner or 100, dtype
        self.moving = args.c4_moving
        self.fixed = getattr(args, "c4_fixed", False)      # "... + 500 fixed steps": S frozen after a warm-up

    def build(self, mhx, ctx, rank):
        import numpy as np
        d = self.d
        self.Sig = sigma_illcond(d)
        model = mhx.DensityModel(mhx.CorrGaussian(self.Sig))
        S0 = None
        if self.moving:           # a variant that moves: random start on the target's scale, S0 = 2.38/sqrt(d) I (about 0.17 I)
            S0 = (2.38 / d ** 0.5) * np.eye(d)
        self.run = mhx.Run(model, mhx.RobustAdaptiveMetropolis(S=S0), nchains=self.C, seed=4, first_chain=rank * self.C, ctx=ctx)
        if self.moving:
            L = np.linalg.cholesky(self.Sig)
            self.run.init(L @ np.random.default_rng(11).normal(size=(d, self.C)))
        else:
            self.run.init(np.zeros(d))                      # x0 = 0, S0 = I
        if self.fixed:
            self.run.sample(1, self.inner, 1, self.inner, save=False)   # a warm-up first: the timed steps run on adapted factors
        return self.run

    def step(self):
        # `inner` adapting transitions (step_warmup), or `inner` transitions with the factor frozen (RAM.jl:216-237)
        self.run.sample(1, self.inner, 1, 0 if self.fixed else self.inner, save=False)
        return self.run.stats()

    def units_per_step(self):
        return self.C * self.inner

    def bytes_per_launch(self):
        """one read + one write of the packed factor per adapting step, B d(d+1) + 2Bd + 2B"""
        B = RB[self.dtype]
        if self.fixed:
            return self.C * self.inner * (B * self.d * (self.d + 1) // 2 + 2 * B * self.d + 2 * B)
        return self.C * self.inner * (B * self.d * (self.d + 1) + 2 * B * self.d + 2 * B)

    def bytes_model(self):
        B = RB[self.dtype]
        if self.fixed:
            return "per fixed-factor step 1 read of the packed factor: %d d(d+1)/2 + %d d + %d" % (B, 2 * B, 2 * B)
        return "per adapting step 1 read + 1 write of the packed factor: %d d(d+1)/2 x 2 + %d d + %d" % (B, 2 * B, 2 * B)

    def describe(self):
        return ("RobustAdaptiveMetropolis (alpha 0.234, gamma 0.6), %d-dim Gaussian, kappa = 1e3 (Q diag Q^T), %d chains per GPU each "
                "with its own factor, %d %s transitions per launch, %s" % (
                    self.d, self.C, self.inner, "fixed-factor (after %d adapting ones)" % self.inner if self.fixed else "adapting",
                    "random start, S0 = 2.38/sqrt(d) I (the variant that moves)" if self.moving else "x0 = 0, S0 = I"))

    def cpu_baseline(self, O, target_seconds):
        import numpy as np
        tgt = O.corr_gauss_from_cov(self.Sig)
        inner, d = self.inner, self.d
        init1 = np.zeros((d, 1))

        warm = 0 if self.fixed else inner                  # fixed-factor steps: the oracle's steps on S0 cost what they cost on any S

        def work(i, nchains):
            O.ram(tgt, O.schedule(1, inner, 1, warm), 4, i * nchains, nchains, init=np.zeros((d, nchains)), save=False)
        t0 = time.perf_counter()
        O.ram(tgt, O.schedule(1, 20, 1, 20 if warm else 0), 4, 0, 1, init=init1, save=False)
        rate1 = 20 / (time.perf_counter() - t0)
        cores = host_cores()
        n, dt = threads_rate(work, cores, target_seconds)
        return n * inner, dt, cores, "%d chains x %d %s transitions of the same d=%d workload; single thread %.3g steps/s" % (
            n, inner, "fixed-factor" if self.fixed else "adapting", d, rate1)


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

    def step(self):
        self.run.sample(self.inner, 1, 1, 0, save=True)
        return self.run.stats()

    def units_per_step(self):
        return self.C * self.inner

    def bytes_per_launch(self):
        B = RB[self.dtype]
        return self.C * (self.inner * (B * 3 + 1) + 2 * (B * 2 + B + 5))

    def bytes_model(self):
        return "record %d(d+1)+1 B per chain-step (a latency-bound single chain: the HBM fraction is not the point)" % RB[self.dtype]

    def describe(self):
        return ("RWMH, d = 2 Normal(mu, sigma) likelihood of 30 data points (README.md:18-63), proposal N(0, I), %d chain(s) x %d draws, "
                "every state recorded" % (self.C, self.inner))

    def cpu_baseline(self, O, target_seconds):
        tgt = O.Target(O.TARGET_IID_NORMAL, 2, params=self.data)
        prop = O.Proposal(O.PROP_ISO, 1.0)
        import numpy as np
        t0 = time.perf_counter()
        O.rwmh(tgt, prop, O.schedule(self.inner), 1234, 0, 1, init=np.array([[0.0], [1.0]]), save=True)
        dt = time.perf_counter() - t0
        return self.inner, dt, 1, "1 chain x %d draws of the same model, one thread (what `sample(model, spl, N)` is)" % self.inner


WORKLOADS = {"c1": C1, "c2": C2, "c3": C3, "c4": C4, "c5": C5}


_ORACLE_FLAGS = None


def cpu_baseline(wl, dtype, target_seconds=10.0):
    """The oracle (same algorithm, same Philox streams, scalar loop per chain) on the host cores: chains statically
    partitioned over threads (the MCMCThreads analogue).  Bounded sample.  Built on this host with -O3 -march=native as
    BASELINE.md's CPU-baseline plan says (oracle/Makefile `native`; bit-identical to the portable build the tests use)."""
    global _ORACLE_FLAGS
    from oracle import oracle as O
    O.build()
    if _ORACLE_FLAGS is None:
        _ORACLE_FLAGS = O.use_native()
    O.set_dtype(dtype)
    units, dt, cores, sample = wl.cpu_baseline(O, target_seconds)
    return {"value": units / dt, "unit": "MH steps/s", "cores": cores, "kind": "port",
            "sample": "%s (oracle/mhx_oracle.c in %s, %s, %d thread(s), %.1f s)" % (sample, dtype, _ORACLE_FLAGS, cores, dt)}


class GlooSum:
    """Fallback transport of the bench's three tiny host-side all-reduces (torch.distributed, gloo): same interface as
    mhx.dist.Comm.allreduce_sum.  Only used when the RCCL communicator behind the C ABI cannot be created."""

    def __init__(self, dist):
        self.dist = dist

    def allreduce_sum(self, v):
        import numpy as np
        import torch
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.numpy()

    def close(self):
        pass


def make_collective(ctx, rank, world, allow_gloo=False):
    """One process per GPU under torchrun.  The rendezvous is torch.distributed's own (gloo on CPU: it only carries the
    128-byte RCCL id from rank 0 to the others); the bench's collectives -- barrier, max-over-ranks time, acceptance totals
    and R-hat sums -- then go through the C ABI (mhx_comm_*: RCCL over xGMI).  If the RCCL communicator cannot be created on
    EVERY rank, all ranks fall back to gloo for those three small host-side all-reduces (reported in config.collective)."""
    import numpy as np
    import torch.distributed as dist
    from mhx.dist import Comm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world)
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm, err = None, ""
    try:
        comm = Comm(ctx, rank, world, box[0])
        comm.allreduce_sum(np.zeros(1))
    except Exception as e:                                       # e.g. no librccl, or two ranks on one device
        comm, err = None, str(e)[:200]
    import torch
    ok = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                    # all ranks or none
    if ok.item() >= 1.0:
        return comm, "mhx_comm_* (RCCL over xGMI, through the C ABI); rendezvous: torch.distributed gloo"
    if comm is not None:
        comm.close()
    if not allow_gloo:
        # a multi-GPU line must not silently skip RCCL: without --allow-gloo the missing communicator is an error
        raise RuntimeError("bench.py --gpus %d: the RCCL communicator behind the C ABI (mhx_comm_init) could not be created on every "
                           "rank (%s); pass --allow-gloo to run the three small host-side all-reduces over torch.distributed gloo instead"
                           % (world, err or "another rank failed"))
    return GlooSum(dist), "torch.distributed gloo (--allow-gloo fallback: the RCCL communicator could not be created: %s)" % err


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
    """ESS/sec of SURVEY 8(d): rank-normalised bulk ESS (split chains, Geyer truncation) of a THINNED window long enough
    for the autocorrelations to die out -- 256 draws `thin` transitions apart, thin chosen so that the window spans ~40
    autocorrelation times (RWMH at the optimal scale: tau ~ d / 0.3; each split half ~20) -- over the wall time of
    producing that window.  `reached_max_lag`: the multi-chain autocorrelation rho_t = 1 - (W - A_t) / var+ keeps the
    floor (var+ - W) / var+ of finite chains, so with thousands of chains averaged its pair sums stay positive up to the
    last lag; the estimate then carries every lag of the window."""
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
    return {"estimator": "rank-normalised bulk ESS, split chains, Geyer initial monotone sequence (mhx_run_ess_bulk_tail)",
            "window": "%d draws x %d chains, %d transitions apart (%d transitions per chain)" % (n_draws, run.n, thin, n_draws * thin),
            "params": [int(p) for p in params], "ess_bulk": [float(v) for v in b["ess_bulk"]],
            "ess_tail": [float(v) for v in b["ess_tail"]], "tail_ess_per_sec": float(np.median(b["ess_tail"])) / wall,
            "reached_max_lag": [bool(v) for v in b["bulk_truncated"]],
            "median": med, "per_transition_per_chain": med / (run.n * n_draws * thin),
            "wall_s": wall, "kernel_ms": st["kernel_ms"], "rhat_max_split": float(np.nanmax(dg["rhat"][:d])),
            "ess_per_sec": med / wall}


def config_block(wl, name, st, collective):
    return {"workload": wl.describe(), "name": name, "units_per_step_per_gpu": wl.units_per_step(),
            "kernel_variant": VARIANTS.get(st["kernel_variant"], str(st["kernel_variant"])), "lanes_per_unit": st["reduce_lanes"],
            "launches_per_step": max(1, st["launches"]),
            "sharding": "chains by global id, no data-path collective" if name != "c3" else "one ensemble per GPU (replicas)",
            "collective": collective}


def roofline_block(wl, name, dtype, kernel_ms, steps, st):
    """The dominant kernel against HBM: algorithmic bytes per step over the HIP-event time of the step's
    launches measured here; `traffic` / `valu` from the PMC passes of tools/profile_round.sh (profiles/traffic.json) when they
    were taken on this workload."""
    launches = max(1, st["launches"])
    launch_s = kernel_ms * 1e-3 / steps
    bytes_launch = wl.bytes_per_launch()
    achieved = bytes_launch / launch_s / 1e9
    traffic, valu = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    key = "%s_%s" % (name, dtype)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath)).get(key, {})
            # (a variant whose kernel or byte count differs from the profiled one carries no PMC figures)
            own_pmc = not (getattr(wl, "fixed", False) or getattr(wl, "banana", False) or getattr(wl, "rotated", False))
            if own_pmc and tj.get("units_per_launch") == wl.units_per_step():
                traffic = tj.get("hbm_bytes_per_launch")
                if tj.get("valu_insts_per_launch"):
                    rate = tj["valu_insts_per_launch"] / launch_s
                    valu = {"wave_insts_per_launch": tj["valu_insts_per_launch"], "achieved_per_s": rate, "peak_per_s": VALU_PEAK,
                            "frac": rate / VALU_PEAK, "source": tj.get("source"),
                            "note": "wave64 VALU instructions (PMC SQ_INSTS_VALU) per second against 1024 SIMDs x one instruction "
                                    "per 2 cycles at 2.4 GHz; the fp64 / 64-bit-multiply / transcendental instructions of the mix "
                                    "take 4-16 cycles each, so 1.0 is not reachable by this instruction mix"}
        except Exception:
            traffic, valu = None, None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "avg_launch_ms": launch_s * 1e3 / launches, "kernel_ms_per_step": launch_s * 1e3,
            "algorithmic_bytes_per_step": bytes_launch, "bytes_model": wl.bytes_model(), "valu": valu}


def other_configs(mhx, ctx, args, barrier):
    """The other BASELINE.json GPU configs in the same driver run (N = 1, rank 0, after the headline's timed region): value,
    roofline of the dominant kernel and a CPU baseline (2 s samples) for C2 with the literal N(0, I) proposal, C3 (and its
    dense-rotated variant), C4 as specified, from a start that moves, and with the factor frozen, and C5's per-GPU shard on the
    funnel and on the banana.  Fewer steps than the headline (C4 runs 0.26 s per step): every figure says how many."""
    import copy
    plan = [("c1", "c1", {}, 3, 1), ("c2_literal", "c2", {"c2_literal": True}, 20, 10),
            ("c3", "c3", {}, 10, 10), ("c3_rotated", "c3", {"c3_rotated": True}, 10, 10), ("c4", "c4", {}, 3, 2),
            ("c4_moving", "c4", {"c4_moving": True}, 3, 2), ("c4_fixed", "c4", {"c4_fixed": True}, 3, 2),
            ("c5", "c5", {}, 10, 10), ("c5_banana", "c5", {"c5_banana": True}, 10, 10)]
    res = {}
    for key, name, over, steps, spin in plan:
        try:
            a = copy.copy(args)
            a.inner = a.chains = a.dim = a.lanes = 0
            a.c4_moving = a.c4_fixed = a.c3_rotated = a.c5_banana = a.c2_literal = False
            for k, v in over.items():
                setattr(a, k, v)
            w = WORKLOADS[name](a, args.dtype)
            w.build(mhx, ctx, 0)
            dt, kms, acc, tr, st = timed(w, steps, 2, barrier, spin=spin)
            blk = {"value": w.units_per_step() * steps / dt, "unit": "MH steps/s", "steps": steps, "ms_per_step": dt * 1e3 / steps,
                   "dtype": args.dtype, "acceptance_rate": acc / float(tr), "config": config_block(w, name, st, None),
                   "roofline": roofline_block(w, name, args.dtype, kms, steps, st)}
            w.run.close()
            if not args.no_cpu_baseline:
                blk["cpu_baseline"] = cpu_baseline(w, args.dtype, 2.0)
            res[key] = blk
        except Exception as e:                                   # one config must not take the line down
            res[key] = {"error": str(e)[:300]}
    return res


def e2e_host(mhx, wl):
    """"kernel time AND end-to-end including D2H of whatever is kept, both reported".  The reference's `sample`
    returns a host container; mhx_run_sample_to_host streams the samples into page-locked host memory while the chains run.
    Three figures on the headline workload: (a) save-all, every state of `inner` transitions to the host (the PCIe link is the
    bound: B(d+1)+1 bytes per chain-step); (b) a thinned run (the ESS window's schedule), which returns at the kernel rate;
    (c) the round-2 path for comparison -- mhx_run_sample, then one synchronous mhx_run_get_samples into pageable memory."""
    import numpy as np
    run, d, C, inner = wl.run, wl.d, wl.C, wl.inner
    out = {}
    t0 = time.perf_counter()
    buf = mhx.host_array((inner, d + 1, C), run.real)
    accb = mhx.host_array((inner, C), np.uint8)
    out["pinned_alloc_s"] = time.perf_counter() - t0
    run.sample_to_host(inner, 1, 1, 0, out=buf, out_accepted=accb)          # untimed: first touch of the slabs and streams
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run.sample_to_host(inner, 1, 1, 0, out=buf, out_accepted=accb)
    w = (time.perf_counter() - t0) / reps
    nbytes = buf.nbytes + accb.nbytes
    out["save_all"] = {"value": C * inner / w, "unit": "MH steps/s with every state on the host", "wall_s": w, "host_GB": nbytes / 1e9,
                       "link_GBps": nbytes / w / 1e9, "kernel_ms": run.stats()["kernel_ms"],
                       "note": "mhx_run_sample_to_host into page-locked memory; bound by the PCIe Gen5 x16 link (63 GB/s spec), not by the kernel"}
    thin = max(1, int(round(40 * (d / 0.3) / 256)))
    n_draws = 256
    tb = mhx.host_array((n_draws, d + 1, C), run.real)
    ta = mhx.host_array((n_draws, C), np.uint8)
    run.sample_to_host(n_draws, thin, thin, 0, out=tb, out_accepted=ta)
    t0 = time.perf_counter()
    run.sample_to_host(n_draws, thin, thin, 0, out=tb, out_accepted=ta)
    w = time.perf_counter() - t0
    out["thinned"] = {"value": C * n_draws * thin / w, "unit": "MH steps/s with the kept draws on the host", "wall_s": w,
                      "schedule": "%d draws %d transitions apart" % (n_draws, thin), "host_GB": (tb.nbytes + ta.nbytes) / 1e9,
                      "kernel_ms": run.stats()["kernel_ms"]}
    del tb, ta
    t0 = time.perf_counter()
    run.sample(inner, 1, 1, 0, save=True)
    old, _ = run.samples()
    w = time.perf_counter() - t0
    out["round2_path"] = {"value": C * inner / w, "wall_s": w, "note": "mhx_run_sample, then one synchronous mhx_run_get_samples into a fresh pageable array"}
    del old
    return out


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
    ap.add_argument("--c3-rotated", action="store_true", help="c3: the dense-rotated variant Sigma = Q (0.9^|i-j|) Q^T (no banded factor)")
    ap.add_argument("--c4-fixed", action="store_true", help="c4: the fixed-factor steps that follow the warm-up (1 read of S per step)")
    ap.add_argument("--c5-banana", action="store_true", help="c5: the banana target of SURVEY 8(d) (ii) instead of Neal's funnel")
    ap.add_argument("--c4-moving", action="store_true", help="c4: random start and S0 = 2.38/sqrt(d) I instead of x0 = 0, S0 = I")
    ap.add_argument("--normal-gen", choices=["auto", "ziggurat", "box-muller"], default="auto",
                    help="c2 / c5: how the RWMH kernel turns stream bits into standard normals (auto: ziggurat in fp64, Box-Muller in fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-gloo", action="store_true", help="--gpus > 1: fall back to torch.distributed gloo for the host-side "
                    "all-reduces when the RCCL communicator cannot be created (default: that is an error)")
    ap.add_argument("--no-other-configs", action="store_true", help="c2 at N=1: skip the C3 / C4 / C4-moving / C5 lines under `configs`")
    ap.add_argument("--no-e2e", action="store_true", help="c2 at N=1: skip the rate through the boundary (samples back on the host)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="size of the CPU-baseline sample of the headline config")
    ap.add_argument("--no-second-dtype", action="store_true", help="skip the fp32 figure")
    ap.add_argument("--no-ess", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import mhx
    from mhx.dist import Comm, allreduce_stats

    local_rank %= max(1, torch.cuda.device_count())             # (several ranks on one device only when a box has fewer GPUs than ranks)
    torch.cuda.set_device(local_rank)
    ctx = mhx.Context(local_rank, args.dtype)
    comm, collective = None, None
    if world > 1 or os.environ.get("MHX_BENCH_FORCE_DIST"):      # the env knob exercises the collective path on 1 GPU
        comm, collective = make_collective(ctx, rank, world, args.allow_gloo)

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.allreduce_sum(np.zeros(1))                      # every rank has arrived
            torch.cuda.synchronize()

    wl = WORKLOADS[args.config](args, args.dtype)
    wl.build(mhx, ctx, rank)
    dt, kernel_ms, accepted, transitions, st = timed(wl, args.steps, args.warmup, barrier)
    variant = st["kernel_variant"]

    # outside the timed region: diagnostics, the ESS window, the fp32 figure, the CPU baseline
    ess, diag = None, None
    if args.config in ("c2", "c5"):
        diag = wl.run.diagnostics(max_lag=0)
    if comm is not None:
        t = np.zeros(world)
        t[rank] = dt
        dt = float(comm.allreduce_sum(t).max())                  # MAX over ranks of the timed region
        if diag is not None:
            diag = allreduce_stats(diag, accepted, transitions, comm=comm)       # ONE all-reduce: RCCL over xGMI through the C ABI
            acc_rate = diag["acceptance_rate"]
        else:
            v = comm.allreduce_sum(np.array([float(accepted), float(transitions)]))
            acc_rate = v[0] / v[1]
    else:
        acc_rate = accepted / float(transitions)
    if args.config == "c2" and not args.no_ess and not args.c2_literal and rank == 0:
        try:
            ess = ess_window(mhx, wl, world)
        except Exception as e:                                   # never let a diagnostic break the bench line
            ess = {"error": str(e)}

    if rank == 0:
        units = float(wl.units_per_step()) * args.steps * world
        value = units / dt
        out = {
            "metric": "MH steps/sec (all chains) + ESS/sec", "value": value, "unit": "MH steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": config_block(wl, args.config, st, collective),
            "acceptance_rate": acc_rate,
            "roofline": roofline_block(wl, args.config, args.dtype, kernel_ms, args.steps, st),
        }
        if ess is not None:
            out["ess_per_sec"] = ess.get("ess_per_sec")
            out["ess"] = ess
        if diag is not None:
            # the all-reduced between / within statistic of the LAST timed launch only (250 consecutive transitions, shorter
            # than one autocorrelation time at d = 100): it exercises the collective, it is not a convergence claim -- the
            # thinned window's split R-hat is ess.rhat_max_split
            out["rhat_last_launch"] = {"value": float(np.nanmax(diag["rhat"][:wl.d])),
                                       "note": "one launch of consecutive transitions across all ranks (the R-hat all-reduce); see ess.rhat_max_split"}
        if world == 1 and args.config == "c2" and not args.no_e2e and not args.c2_literal:
            try:
                out["e2e_host"] = e2e_host(mhx, wl)
            except Exception as e:
                out["e2e_host"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_second_dtype and args.dtype == "f64":
            try:                                                  # the same workload on the fp32 engine: a second figure, never `value`
                ctx32 = mhx.Context(local_rank, "f32")
                wl32 = WORKLOADS[args.config](args, "f32")
                wl32.build(mhx, ctx32, rank)
                n32 = max(5, args.steps // 2)
                dt32, k32, a32, t32, st32 = timed(wl32, n32, 2, barrier)
                out["f32"] = {"value": wl32.units_per_step() * n32 / dt32, "unit": "MH steps/s", "ms_per_step": dt32 * 1e3 / n32,
                              "acceptance_rate": a32 / float(t32), "lanes_per_unit": st32["reduce_lanes"],
                              "roofline_frac_hbm": wl32.bytes_per_launch() / (k32 * 1e-3 / n32) / 1e9 / HBM_PEAK_GBS,
                              "note": "same engine compiled with mhx_real = float: half the bytes; not the reference's arithmetic"}
                wl32.run.close()
            except Exception as e:
                out["f32"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.dtype, args.cpu_seconds)
        if world == 1 and args.config == "c2" and not args.no_other_configs and not args.c2_literal:
            wl.run.close()                                        # 13.4 GB of samples: C4 needs the room
            out["configs"] = other_configs(mhx, ctx, args, barrier)
        # RCCL writes its version banner to the C stdout buffer: push it out first so the JSON line is the last one
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if comm is not None:
        barrier()
        comm.close()
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
