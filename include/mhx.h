/*
 * mhx.h -- C ABI of libmhx.so, the MI355X (gfx950) many-chain Metropolis-Hastings engine.
 *
 * The reference (TuringLang/AdvancedMH.jl v0.8.8) is pure Julia and has no FFI today; its operator
 * API for this path is the AbstractMCMC interface.  Each entry point below names the reference
 * interface it replaces (file:line under the reference tree).  A Julia maintainer binds these with
 * `ccall` (advancedmh.jl_amd/julia/AdvancedMHHIP.jl); the executable host mirror in
 * this repo is the Python package advancedmh.jl_amd/mhx (ctypes).
 *
 * Conventions
 *   - every function returns MHX_OK (0) or a negative mhx_status; mhx_last_error() gives the
 *     message for the calling thread.  Nothing throws across the boundary.
 *   - the caller owns every host buffer; the library owns device memory behind opaque handles.
 *   - calls are blocking: the device work they enqueue has completed when they return.
 *   - host tensor layouts are C order with the CHAIN index fastest:
 *       x        [dim][nchains]
 *       samples  [n_samples][dim + 1][nchains]   (row `dim` is lp; cf. ext/AdvancedMHMCMCChainsExt.jl:96-105)
 *       accepted [n_samples][nchains]            (Transition.accepted, src/AdvancedMH.jl:61-65)
 *       S        [nchains][dim*(dim+1)/2]        packed lower triangle, row-major (RAM factor)
 *   - `real` below is the context's arithmetic type: double for MHX_F64 -- what the reference computes in (Distributions'
 *     Float64 rand / logpdf; src/RobustAdaptiveMetropolis.jl:187-196: T = eltype(sampler.gamma) = Float64) -- or float
 *     for MHX_F32 (the same engine at half the bytes and ~3x the rate).  Every `void *` / `const void *` real buffer of
 *     a call is an array of the dtype of the context the handle belongs to; scalar parameters travel as double and are
 *     rounded once to the context's type.
 *   - a handle is not thread-safe; distinct contexts (one per GPU) may be driven concurrently.
 *   - chains carry GLOBAL ids first_chain .. first_chain+nchains-1 in their RNG counters, so a run
 *     sharded over several GPUs/processes is bit-identical to the unsharded run.
 */
#ifndef MHX_H
#define MHX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHX_VERSION 600 /* 0.6.0: the accept-compacted return path (mhx_compact_hdr, mhx_compact_expand, mhx_run_host_stats, options HOST_COMPACT / HOST_THREADS / HOST_CHUNK); 0.5.0: mhx_group_* (many GPUs from one process), mhx_ctx_set_option (explicit engine options: the library no longer reads tuning variables from the environment), mhx_stats.tainted, mhx_comm_init_timed / deadlines on the collectives, mhx_ctx_pci_bus_id, mhx_run_shape; 0.4.0: mhx_ram_get_step_stats takes a capacity, watched factors, host pin accounting, kernel variant 9 (0.3.0: overlapped host return path, page-locked host buffers, persistent JIT cache) */

typedef enum {
    MHX_OK = 0,
    MHX_EINVAL = -1,   /* bad argument: dim mismatch, non-zero-mean RW proposal, missing initial_params ... */
    MHX_ENOMEM = -2,
    MHX_EHIP = -3,     /* HIP runtime error */
    MHX_EJIT = -4,     /* hiprtc failed to compile a user log-density / specialised kernel */
    MHX_ENOTPD = -5,   /* replaces LinearAlgebra.PosDefException of lowrankdowndate (reported, never thrown) */
    MHX_ESTATE = -6    /* call out of order (e.g. sampling before init) */
} mhx_status;

typedef struct mhx_ctx mhx_ctx;       /* one per GPU: device, stream, JIT cache */
typedef struct mhx_target mhx_target; /* a log-density on the device  == DensityModel (src/AdvancedMH.jl:52-54) */
typedef struct mhx_run mhx_run;       /* device-resident chains of one sampler + their sample buffer */

int mhx_version(void);
const char *mhx_last_error(void);

typedef enum { MHX_F32 = 0, MHX_F64 = 1 } mhx_dtype;

int mhx_ctx_create(int device, int dtype /* mhx_dtype */, mhx_ctx **out);
int mhx_ctx_dtype(const mhx_ctx *ctx);
int mhx_ctx_device(const mhx_ctx *ctx, int *device);
int mhx_ctx_destroy(mhx_ctx *ctx);
/* "dddd:bb:dd.f" of the context's device (hipDeviceGetPCIBusId): ordinals depend on HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, the
 * bus id does not -- N ranks (or the members of a group) run on N GPUs iff their bus ids are distinct.  len >= 16. */
int mhx_ctx_pci_bus_id(const mhx_ctx *ctx, char *buf, size_t len);

/* Engine options: explicit, per context, set BEFORE the runs they steer are created.  The library reads nothing but
 * MHX_CACHE_DIR / MHX_NO_JIT_CACHE / XDG_CACHE_HOME / HOME (where compiled kernels are kept), MHX_JIT_COMPILER / MHX_JIT_CLANG / ROCM_PATH / MHX_JIT_VERBOSE (who compiles them) and MHX_RCCL_LIB from the environment;
 * which kernel form runs is chosen from (dim, chains, dtype) unless an option says otherwise, and mhx_stats reports the form.
 * Every option below yields the SAME chain law and, for a given reduction shape, the same bits (each form is held to the oracle
 * in tests/): they exist so that every form can be reached at every size, and for A/B measurements.
 *   kernel form     NO_PREBUILT NO_MFMA EMCEE_MFMA EMCEE_SCALAR EMCEE_FUSED EMCEE_PERSIST EMCEE_DEFER EMCEE_SWEEP_DEFER
 *                   EMCEE_PRELOAD RAM_G
 *   tuning          COOP_WAVES MFMA_WAVES REG_MAX_DIM REG_XR REG_UNROLL REG_WAVES REG_ZSLAB MALA_XR EMCEE_WAVES EMCEE_MFMA_WAVES
 *                   EMCEE_SCAL_WPB EMCEE_SCAL_MODE EMCEE_SCAL_REC EMCEE_REC_STORE EMCEE_ROW_STORE EMCEE_COOP_REC RAM_LDS_PAD
 *                   WAVE_K (4 | 8 speculative candidates per round of the wave-per-chain kernel; default: by the last call's acceptance)
 *   return path     HOST_COMPACT (1 / 0: the accept-compacted form of mhx_run_sample_to_host on / off; default: on for thinning == 1
 *                   and >= 1024 chains) HOST_THREADS (host threads that expand; default: what the process may use) HOST_CHUNK (chains
 *                   per unit of their work, a multiple of 64; default: so that a unit's state fits a core's L2) HOST_NUMA (0: do not
 *                   place the staging memory and the expanding threads on the GPU's memory node)
 *   run-time kernels JIT_COMPILER ("hiprtc" | "clang" | unset = clang++ of the installation when found, else hiprtc; mhx_ctx_jit_compiler)
 *   sharding        TOTAL_CHAINS (the chains of the WHOLE run this context's runs are shards of: where the engine picks a kernel
 *                   form -- hence a summation order -- from the chain count (reduce_lanes = 0), it picks for that count, so a shard
 *                   runs what the unsharded run runs; mhx_group_shard sets it on the member's context, a process of a
 *                   multi-process run sets it itself)
 * value == NULL unsets.  An unknown name is MHX_EINVAL.  The tools build (libmhx_tools.so, `make tools`) additionally knows
 * timing probes and fault injection (ZIG_PROBE, EMCEE_PROBE, EMCEE_STAMPS, ZIG_FORCE_FAIL, FAULT_SLAB, JIT_DEFS, JIT_FLAGS, RAM_PROF): setting one
 * marks the context TAINTED -- mhx_stats.tainted = 1 for every run of it, and the host mirrors refuse to build a Chains from
 * such a run.  In libmhx.so those names do not exist. */
int mhx_ctx_set_option(mhx_ctx *ctx, const char *name, const char *value);
/* the value in effect ("" when unset) copied to buf; MHX_EINVAL for an unknown name */
int mhx_ctx_get_option(const mhx_ctx *ctx, const char *name, char *buf, size_t len);
/* run-time specialisations of this context so far: compiled / loaded from the on-disk code-object cache
 * ($MHX_CACHE_DIR, default ~/.cache/mhx; key = hash(source, device headers, options, compiler + runtime version);
 * MHX_CACHE_DIR="" or MHX_NO_JIT_CACHE=1 switches the cache off).  Either pointer may be NULL. */
int mhx_ctx_jit_counts(const mhx_ctx *ctx, int64_t *compiles, int64_t *cache_hits);
/* WHICH compiler builds them (0.6.0).  The installation's own clang++ where one is found -- $MHX_JIT_CLANG (a path; "0" = none),
 * $ROCM_PATH/lib/llvm/bin/clang++, /opt/rocm/lib/llvm/bin/clang++ -- run as a child process on the same source, headers and options;
 * the hiprtc library otherwise, when that fails, or when option JIT_COMPILER = "hiprtc" ("clang" = no fall-back; $MHX_JIT_COMPILER is
 * the process-wide default of the option: a child process costs ~0.5 s more per kernel than hiprtc in-process, which a test suite that
 * compiles hundreds of small kernels may not want to pay).  Why: hiprtc is
 * whichever libhiprtc.so.7 / libamd_comgr.so.3 the PROCESS loaded first -- inside Python after `import torch` the wheel's bundled,
 * older compiler, whose cooperative RWMH kernel runs 7 % behind the one hipcc builds from the same text.  `compiler` receives
 * "clang++:<path>:<size>:<mtime>" or "" (none found: hiprtc only), ext_compiles how many of mhx_ctx_jit_counts' compilations it did. */
int mhx_ctx_jit_compiler(const mhx_ctx *ctx, char *compiler, size_t len, int64_t *ext_compiles);

/* Page-locked host memory for result tensors (the `Chains` array of ext/AdvancedMHMCMCChainsExt.jl:12-39 lives on the
 * host): device-to-host copies into it run at the link rate and asynchronously.  Any other host buffer works too --
 * mhx_run_sample_to_host registers it for the duration of the call. */
int mhx_host_alloc(size_t bytes, void **out);
/* Caller buffers this context page-locked for the duration of a mhx_run_sample_to_host call (hipHostRegister) and released
 * again: equal between calls, whatever path a call left by (every exit first drains both streams).  Either may be NULL. */
int mhx_ctx_host_pin_counts(const mhx_ctx *ctx, int64_t *registered, int64_t *released);
int mhx_host_free(void *p);

/* ---------------------------------------------------------------------------------------------
 * Targets.  Replaces DensityModel(f) / logdensity(model, x) (src/AdvancedMH.jl:52-54, :74-77). */
typedef enum {
    MHX_TARGET_ISO_GAUSS = 0,  /* logpdf(MvNormal(zeros(d), I), x);            params: none               */
    MHX_TARGET_CORR_GAUSS = 1, /* logpdf(MvNormal(zeros(d), Sigma), x);        params: inv(chol(Sigma)) packed lower */
    MHX_TARGET_IID_NORMAL = 2, /* README.md:29-31: theta=(mu,sigma), sum logpdf(Normal(mu,sigma), data); params: data */
    MHX_TARGET_BANANA = 3,     /* twisted Gaussian N(0, diag(100,1,...)) with x2 += b(x1^2-100); params: {b}   */
    MHX_TARGET_FUNNEL = 4,     /* Neal's funnel: x1~N(0,9), xk~N(0,exp(x1));  params: none               */
    MHX_TARGET_USER = 100      /* hiprtc-compiled user source                                              */
} mhx_target_kind;

int mhx_target_builtin(mhx_ctx *ctx, int kind, int dim, const void *params, size_t nparams,
                       mhx_target **out);
/* `src` must define   MHX_LOGDENSITY(x, d, data, ndata) { ... return lp; }   using x[k] and the
 * mhx_fma / mhx_log / mhx_exp / mhx_sqrt device functions, written against `mhx_real` and `MHX_R(literal)` so that one
 * text serves both dtypes (a source that says `float` is fp32-only); it is compiled by hiprtc and inlined
 * into the sampler kernels (the "JIT-lowered user log-density" of the design). */
int mhx_target_from_hip_source(mhx_ctx *ctx, const char *src, int dim, const void *data, size_t ndata,
                               mhx_target **out);
int mhx_target_destroy(mhx_target *t);
/* logdensity(model, x) for a batch: x [dim][n] (host) -> lp [n] (host).  src/AdvancedMH.jl:74 */
int mhx_target_eval(mhx_ctx *ctx, const mhx_target *t, const void *x, int n, void *lp);

/* ---------------------------------------------------------------------------------------------
 * Schedule.  Replaces the kwargs of AbstractMCMC.sample/mcmcsample [upstream]: N, discard_initial,
 * thinning, num_warmup.  Sample 1 is the state after `discard_initial` transitions from the
 * initial state; sample i is `thinning` transitions after sample i-1. */
typedef struct {
    int32_t n_samples;
    int32_t discard_initial;
    int32_t thinning;   /* >= 1 */
    int32_t num_warmup; /* RAM: transitions that adapt (step_warmup) */
} mhx_schedule;

/* ---------------------------------------------------------------------------------------------
 * Random-walk Metropolis-Hastings.  Replaces MetropolisHastings{RandomWalkProposal{_, <:MvNormal}}
 * (src/mh-core.jl:44-51, src/proposal.jl:13-25) and its step methods (src/mh-core.jl:76-117). */
typedef enum { MHX_PROP_ISO = 0, MHX_PROP_DIAG = 1, MHX_PROP_DENSE = 2 } mhx_proposal_kind;

typedef struct {
    int32_t dim;
    int32_t nchains;
    uint64_t seed;
    uint64_t first_chain;   /* global id of chain 0 of this shard */
    int32_t proposal_kind;  /* one of mhx_proposal_kind; src/proposal.jl:49-64 */
    double proposal_scale;   /* ISO: sigma of N(0, sigma^2 I) */
    const void *proposal_vec; /* DIAG: sigma_k [dim]; DENSE: chol(Sigma) packed lower [dim(dim+1)/2] */
    int32_t flags;          /* MHX_FLAG_* */
    const void *proposal_mean; /* NULL = zero mean.  mu[dim]: a drifting random walk x + mu + L z; its Hastings ratio
                               q(x | y) - q(y | x) (src/proposal.jl:58-64,190-192) is then non-zero and is computed.
                               Runs on the generic kernel. */
    int32_t reduce_lanes;   /* lanes that share one chain (power of two <= 64) for the separable catalogue
                               targets (MHX_TARGET_IID_NORMAL: 64 = a wave per chain, the few-chain kernel, which is also the
                               engine's choice up to 2048 chains); 0 = let the engine choose from nchains and dim, 1 = one lane per chain.
                               The value in effect is reported in mhx_stats.reduce_lanes: it fixes the
                               summation order of the log-density and therefore the exact chain. */
} mhx_rwmh_cfg;

#define MHX_FLAG_NO_JIT 1 /* never specialise with hiprtc; use the pre-built kernels only */
#define MHX_FLAG_GENERIC 2 /* force the generic (state-in-HBM) kernel even when a register kernel exists */
#define MHX_FLAG_EMCEE_SEQUENTIAL 8 /* Ensemble runs only: the reference's sweep (src/emcee.jl:39-58) -- walkers move one after
                                      another and pair with already-updated walkers (Gauss-Seidel), one wavefront, for
                                      fidelity checks at the reference's test sizes; the default is the parallel half-split */
#define MHX_FLAG_ZIGGURAT 16 /* RWMH and (round 5) MALA runs: standard normals by the table ZIGGURAT of the arithmetic spec (DESIGN.md
                                section 3.11: fp64 contexts 1024 equal-area layers and 64 bits per normal; fp32 contexts -- since 0.6.0 --
                                256 layers and 32 bits per normal; exact rejection sampling) instead of
                                Box-Muller -- a third fewer instructions per transition on the cooperative kernel (separable
                                catalogue targets) and on the register kernel (any target within its dimension limit, a user's
                                HIP source included), ISO / DIAG proposals.  It selects the STREAM of normals, so the chain differs
                                from the Box-Muller chain of the same seed (both target the same law); the value in effect is
                                reported in mhx_stats.normal_gen and fixes the chain bit for bit.  MHX_EINVAL where the run's
                                kernel has no ziggurat form (dense factors, the matrix-core and state-in-HBM kernels).  MALA: the
                                noise z of the Langevin proposal, on the lane-per-chain register kernel (any target with a gradient
                                within that kernel's dimension limit; reduce_lanes <= 1). */
#define MHX_FLAG_DENSE_FACTOR 32 /* Ensemble runs: treat the precision factor of a dense-Gaussian target as dense even when it is
                                   banded (exact zeros below a band of width <= 8 are otherwise detected and skipped -- same bits) */
#define MHX_FLAG_RAM_DEFERRED 64 /* RAM runs, dim <= 256: the DEFERRED-FACTOR form.  ram_adapt's rank-one step is S <- S chol(I +- c^2 U U')
                                   (src/RobustAdaptiveMetropolis.jl:153-173 with w = c S U), a triangular factor known from U and two
                                   scalars: up to 8 accepted updates stay pending as O(dim) triples -- proposals read the stored factor
                                   once and apply the pending factors to the noise first -- and are folded into S in ONE read + write pass
                                   (kernel variant 12): 9/8 reads + 1/8 writes of S per adapting step instead of 1 + 1.  The same Markov
                                   chain in exact arithmetic, ANOTHER ROUNDING than the reference's sequential lowrankupdate! sweeps
                                   (arithmetic spec 3.12; its own oracle twin, orc_ram_deferred): S S' agrees with the default form's to
                                   ~1e-15 relative per step, accept decisions may differ where |log u - log alpha| is at rounding level.
                                   Folds happen when 8 updates are pending, at the end of the warm-up and at the end of every launch (a
                                   sampling call is cut into launches of 4096 transitions), so the factor is whole whenever the host can
                                   see it; mhx_ram_watch_factors is refused (MHX_EINVAL). */
#define MHX_FLAG_STATIC_PROPOSAL 4 /* RWMH runs only: the proposal is a StaticProposal (src/proposal.jl:9-11,66-83) --
                                      the candidate is a draw mean + L z that ignores the current state (independence
                                      sampler) and the ratio is logpdf(p, x) - logpdf(p, y) */

int mhx_rwmh_create(mhx_ctx *ctx, const mhx_target *t, const mhx_rwmh_cfg *cfg, mhx_run **out);

/* ---------------------------------------------------------------------------------------------
 * Affine-invariant ensemble.  Replaces Ensemble{StretchProposal} (src/emcee.jl:1-4, :63-68), its
 * step (:14-24), sweep (:39-58) and stretch move (:70-102).  The device sweep is the parallel
 * half-split of the walker set, not the reference's sequential loop. */
typedef struct {
    int32_t dim;
    int32_t nwalkers;
    uint64_t seed;
    uint64_t ensemble_id;    /* < 2^32: one word of the RNG counter (walker index is the other) */
    double stretch;          /* a = 2.0 */
    int32_t flags;          /* MHX_FLAG_*; MHX_FLAG_EMCEE_SEQUENTIAL selects the reference's own sweep */
    int32_t reduce_lanes;   /* lanes per walker (dense-Gaussian target): 0 = engine's choice, 1 = one lane per walker */
    /* the distribution StretchProposal wraps (src/emcee.jl:63-68), used ONLY for the initial walkers (:29-34: W draws
     * from it): a (Mv)Normal  mu + L z  drawn on the device by mhx_run_init(run, NULL).  init_kind < 0: none given --
     * mhx_run_init then requires the walkers (any other prior is drawn by the host). */
    int32_t init_kind;      /* mhx_proposal_kind, or -1 */
    double init_scale;      /* ISO: sigma */
    const void *init_vec;   /* DIAG: sigma_k [dim]; DENSE: chol(Sigma) packed lower [dim(dim+1)/2] */
    const void *init_mean;  /* mu [dim] or NULL */
    int32_t n_ensembles;    /* 0.6.0: E independent ensembles in ONE run -- `sample(model, Ensemble(W, ..), MCMCThreads(), N, nchains)`
                               (README.md:135-148) runs nchains ENSEMBLES, and emcee is run at sizes (test/emcee.jl:24: 1000 walkers) that
                               leave the chip idle one at a time.  0 / 1 = one.  Ensemble e carries id ensemble_id + e in its RNG counters
                               and is bit for bit the run of that id alone; its walkers are columns e W .. e W + W - 1 of every
                               [..][E W] array (x, samples, accepted; mhx_run_shape reports E W chains).  Every launch takes E as a second
                               grid dimension; an ensemble of <= 1024 walkers (kernel variant 6) is one persistent block, so E of them
                               occupy E CUs.  The sharded-ensemble building blocks below (mhx_emcee_half_step ...) need E = 1. */
} mhx_emcee_cfg;

int mhx_emcee_create(mhx_ctx *ctx, const mhx_target *t, const mhx_emcee_cfg *cfg, mhx_run **out);

/* ---------------------------------------------------------------------------------------------
 * Robust adaptive Metropolis.  Replaces RobustAdaptiveMetropolis (src/RobustAdaptiveMetropolis.jl:75-87),
 * its state (:99-114) and step / step_warmup (:175-278). */
typedef struct {
    int32_t dim;
    int32_t nchains;
    uint64_t seed;
    uint64_t first_chain;
    double alpha;            /* 0.234 */
    double gamma;            /* 0.6 */
    double eig_lo, eig_hi;   /* eigenvalue (diagonal) bounds, 0 and +inf */
    int32_t flags;           /* MHX_FLAG_RAM_DEFERRED */
} mhx_ram_cfg;

int mhx_ram_create(mhx_ctx *ctx, const mhx_target *t, const mhx_ram_cfg *cfg, mhx_run **out);
/* in/out Cholesky factors, [nchains][dim(dim+1)/2]; S == NULL on set means identity */
int mhx_ram_set_factor(mhx_run *run, const void *S);
/* the same factor for every chain (RobustAdaptiveMetropolis(S = ...), :198-206): S [dim(dim+1)/2] */
int mhx_ram_set_factor_all(mhx_run *run, const void *S);
int mhx_ram_get_factor(mhx_run *run, void *S, uint8_t *status /* [nchains] or NULL */);
/* running min / max of diag(S) over every adapted state so far, [dim][nchains] each */
int mhx_ram_get_diag_range(mhx_run *run, void *diag_min, void *diag_max);
/* the rest of RobustAdaptiveMetropolisState (src/RobustAdaptiveMetropolis.jl:99-114): log_alpha [nchains] reals = the log
 * acceptance ratio min(lp' - lp, 0) of each chain's latest transition (kept bounded at 0 so that users can average
 * exp(log_alpha), :141-147); *eta = the adaptation step size iteration^-gamma of the latest warm-up transition (0 before
 * any, :211); isaccept [nchains]; *iteration = 1 + transitions so far.  Any pointer may be NULL. */
int mhx_ram_get_adapt_state(mhx_run *run, void *log_alpha, double *eta, uint8_t *isaccept, uint64_t *iteration);
/* The same statistics for EVERY recorded step of the last mhx_run_sample / mhx_run_sample_to_host that kept samples -- what a
 * `callback(rng, model, sampler, sample, state, i)` reads off `state` after each saved step in the reference
 * (test/RobustAdaptiveMetropolis.jl:11-28,55): log_alpha [n_samples][nchains] reals = state.logα after the recorded transition
 * (min(lp' - lp, 0): average exp(log_alpha) for the acceptance rate, RAM.jl:141-147; sample 1 of an un-discarded call carries
 * the state's own value, 0 right after init, :211), eta [n_samples] doubles = state.η (iteration^-gamma of the latest adapting
 * transition at or before that step; the same for every chain).  state.isaccept is the `accepted` tensor.  Either may be NULL.
 * `capacity` = the number of samples the two buffers hold: fewer than the last call recorded is MHX_EINVAL (nothing is written);
 * n_recorded (may be NULL) receives that count -- with both buffers NULL the call is just this query. */
int mhx_ram_get_step_stats(mhx_run *run, void *log_alpha, double *eta, int64_t capacity, int64_t *n_recorded);

/* state.S of a few WATCHED chains after EVERY recorded step -- what a reference callback that stores `state.S` keeps
 * (test/RobustAdaptiveMetropolis.jl:11-28; :57-69 checks the eigenvalue bounds on that record).  mhx_ram_watch_factors(run, chains, n)
 * names the chains (indices into this run, n = 0 switches it off); every later mhx_run_sample / mhx_run_sample_to_host that keeps
 * samples also keeps S [n_samples][n][dim (dim + 1) / 2] (packed lower, row-major, like mhx_ram_get_factor); a watched run ends a
 * launch at every recorded transition, others are unaffected.  mhx_ram_get_watched_factors copies the record out: `capacity` = the
 * number of samples S holds (fewer than recorded: MHX_EINVAL); n_recorded / n_watched may be NULL; S == NULL just queries them. */
int mhx_ram_watch_factors(mhx_run *run, const int32_t *chains, int32_t n);
int mhx_ram_get_watched_factors(mhx_run *run, void *S, int64_t capacity, int64_t *n_recorded, int32_t *n_watched);

/* ---------------------------------------------------------------------------------------------
 * Metropolis-adjusted Langevin.  Replaces MALA (src/MALA.jl:1-11), GradientTransition (:14-19) and its step
 * (:54-93) for the standard proposal  g -> MvNormal((sigma2/2) g, sigma2 I)  of the reference's tests
 * (test/runtests.jl:291,352).  Gradients are analytic for the catalogue targets; a user source must also
 * define  MHX_LOGDENSITY_AND_GRADIENT(x, g, d, data, ndata) { ... g.set(k, dlp_dxk); ... return lp; }
 * (the reference throws in check_capabilities, src/MALA.jl:42-52, when no gradient is available).
 * mhx_run_init requires initial_params (src/MALA.jl:37). */
typedef struct {
    int32_t dim;
    int32_t nchains;
    uint64_t seed;
    uint64_t first_chain;
    double sigma2;
    int32_t flags;
    int32_t reduce_lanes;   /* lanes per chain (separable catalogue targets; 4 = the matrix-core kernel of the dense Gaussian target): 0 = engine's choice, 1 = one lane per chain; the
                               value in effect (mhx_stats.reduce_lanes) fixes the summation order of the three sums of a step */
} mhx_mala_cfg;

int mhx_mala_create(mhx_ctx *ctx, const mhx_target *t, const mhx_mala_cfg *cfg, mhx_run **out);

/* ---------------------------------------------------------------------------------------------
 * Running chains.  mhx_run_init == the initial AbstractMCMC.step (src/mh-core.jl:76-86,
 * src/emcee.jl:29-34, src/RobustAdaptiveMetropolis.jl:175-214): x0 = initial_params if given, else
 * a draw (RWMH: from the proposal; RAM: randn(d); Ensemble: W draws from the (Mv)Normal of cfg.init_*, else required).
 * mhx_run_sample == the mcmcsample loop + bundle_samples into the device sample buffer.  It may be
 * called repeatedly; each call continues the chains (counter-based RNG => resumable).  RWMH: a -0.0 coordinate of a caller's state
 * (here and in mhx_run_set_state) enters the chain as +0.0 -- equal under ==, so `chain[1].params == initial_params` holds. */
int mhx_run_init(mhx_run *run, const void *initial_params /* host [dim][nchains] or NULL */);
/* save_samples: 0 = keep nothing, 1 = sample tensor, 2 = running moments only (per chain and parameter the
 * mean / M2 of the states the schedule selects; for runs whose sample tensor would not fit, e.g.
 * 262 144 chains x 1000 dims -- mhx_run_diagnostics then works from the moments; RWMH on the cooperative or
 * generic kernel) */
#define MHX_SAVE_NONE 0
#define MHX_SAVE_SAMPLES 1
#define MHX_SAVE_MOMENTS 2
int mhx_run_sample(mhx_run *run, const mhx_schedule *sched, int save_samples);

/* copy the sample buffer of the last mhx_run_sample to the host (either pointer may be NULL) */
int mhx_run_get_samples(mhx_run *run, void *samples, uint8_t *accepted);
/* mhx_run_sample + mhx_run_get_samples in ONE call with the copy overlapped: what `sample(model, sampler, N)` returns is
 * a host container (bundle_samples, src/AdvancedMH.jl:80-104, ext/AdvancedMHMCMCChainsExt.jl:12-39).  The schedule is cut
 * into slabs of |slab_samples| saved samples (0 = about 256 MiB); the kernels fill one device slab while the previous one
 * drains to `samples` [n_samples][dim+1][nchains] / `accepted` [n_samples][nchains] (NULL = not wanted) on a second
 * stream.  A tensor that fits into HBM stays there as well (mhx_run_device_samples and the diagnostics work as after
 * mhx_run_sample); one that does not -- or slab_samples < 0 -- goes through TWO alternating slabs and the device keeps
 * nothing (mhx_run_device_samples then reports 0 samples): n_samples is bounded by host memory, not by HBM.  Blocking; the
 * buffers are complete on return.  Bit-identical to mhx_run_sample + mhx_run_get_samples.
 * mhx_stats.kernel_ms then spans the kernels including their waits for a free slab, wall_ms the whole call. */
int mhx_run_sample_to_host(mhx_run *run, const mhx_schedule *sched, void *samples, uint8_t *accepted, int32_t slab_samples);

/* The ACCEPT-COMPACTED form of that return path (0.6.0).  A rejected transition re-emits the previous Transition
 * (src/mh-core.jl:109-114; src/emcee.jl:93-101; src/RobustAdaptiveMetropolis.jl:148-150), so at the acceptance rates these samplers
 * are tuned to three quarters of a save-all tensor repeat the row above, byte for byte.  With thinning == 1 and >= 1024 chains
 * (or context option HOST_COMPACT = 1; 0 switches it off) mhx_run_sample_to_host therefore moves, per slab, one BLOCK over the link:
 * a bit per (sample, chain) -- "this chain's column [dim+1] differs from the sample before", found by comparing the bits of the two
 * columns on the device, whatever the sampler -- and the columns of the chains whose bit is set; sample 0 of a call travels whole.
 * Host threads inside the library (context option HOST_THREADS; default = what the process may use: affinity mask and cgroup CPU
 * quota, at most 64) rebuild the caller's tensor from the blocks with non-temporal stores while the next slab is in flight.  The
 * caller's buffers need not be page-locked on this path (the CPU writes them) and none is registered.  Bit-identical to the plain
 * path by construction.  C2 in fp64: 13.4 GB of tensor per 250 transitions of 65 536 chains cross the link as 3.2 GB.
 *
 * A block:  mhx_compact_hdr | mask u64[count][words] | rank u32[count][words] (pad to 8) | accepted u8[count][nchains] (pad to 8)
 *           | payload: for sample i  elem[dim1][m_i]   (m_i = set bits of mask[i][.]; row k holds parameter k of the changed
 *           chains in chain order; parameter dim1-1 is lp)
 * rank[i][w] = set bits of the block's masks before word (i, w) in (sample, word) order, so sample i's payload starts at element
 * dim1 * rank[i][0] and chain c of it sits at  rank[i][c/64] - rank[i][0] + popcount(mask[i][c/64] & ((1 << c%64) - 1)). */
#define MHX_COMPACT_MAGIC 0x4358484du /* "MHXC" */
typedef struct {
    uint32_t magic;
    uint32_t elem_bytes;      /* 4 | 8: the run's dtype */
    uint32_t dim1;            /* dim + 1 */
    uint32_t nchains;
    uint64_t first_sample;    /* row of the call's tensor the block starts at */
    uint32_t count;           /* samples in the block */
    uint32_t words;           /* (nchains + 63) / 64 */
    uint64_t total_changed;   /* sum of m_i */
    uint64_t payload_offset;  /* bytes from the start of the block */
    uint64_t block_bytes;     /* payload_offset + total_changed * dim1 * elem_bytes */
    uint64_t reserved_;
} mhx_compact_hdr;            /* 64 bytes */
/* Expand ONE block into the caller's whole tensor samples [n_samples][dim1][nchains] (and accepted [n_samples][nchains], or NULL);
 * blocks of a call are expanded in order (a block reads the row above its first sample).  The block is validated (sizes, ranks
 * against masks) before anything is written: MHX_EINVAL otherwise.  For hosts that move samples with their own transport -- the
 * workers of Distributed.jl or MPI ranks ship blocks, not tensors.  threads <= 0: as above.  Needs no GPU. */
int mhx_compact_expand(const void *block, size_t block_bytes, void *samples, uint8_t *accepted, int64_t n_samples, int32_t threads);

/* What the last mhx_run_sample_to_host of the run moved, and how */
typedef struct {
    uint64_t tensor_bytes;    /* samples + accepted delivered to the caller */
    uint64_t wire_bytes;      /* bytes of device-to-host copies behind them */
    double link_ms;           /* time the copies occupied the link (stream events, summed over the slabs) */
    double expand_ms;         /* compacted path: time the host threads spent on blocks (first worker in to last worker out, summed) */
    int32_t compact;          /* 1: the accept-compacted path ran */
    int32_t threads;          /* host threads that expanded */
    int32_t slabs;
    int32_t ring;             /* 1: two alternating device slabs (the device kept nothing) */
} mhx_host_stats;
int mhx_run_host_stats(mhx_run *run, mhx_host_stats *out);
/* getparams / setparams!! (src/AdvancedMH.jl:146-157, src/RobustAdaptiveMetropolis.jl:116-121) */
int mhx_run_get_state(mhx_run *run, void *x, void *lp, uint32_t *accept_counts);
int mhx_run_set_state(mhx_run *run, const void *x /* lp is recomputed */);

/* Checkpoint / resume -- the `state` half of AbstractMCMC's (sample, state) = step(...) and of upstream's
 * `initial_state` keyword (src/mh-core.jl:92-117, src/emcee.jl:14-24, src/RobustAdaptiveMetropolis.jl:99-114): the
 * complete state of a run -- positions, cached log-densities, accept bookkeeping, the RNG step counter with its
 * seed and global ids, and per sampler the RAM factors / selectors / diagonal ranges, the MALA gradients, the
 * static proposal's log-densities -- as one host blob.  A run created with the same sampler, model, dim and chain
 * count continues the saved one bit for bit after mhx_run_load_state (its own seed / first_chain are replaced). */
int mhx_run_state_size(mhx_run *run, size_t *bytes);
int mhx_run_save_state(mhx_run *run, void *blob, size_t bytes);
int mhx_run_load_state(mhx_run *run, const void *blob, size_t bytes);

typedef struct {
    uint64_t transitions;      /* chain-steps executed by the last mhx_run_sample (all chains)         */
    uint64_t accepted;         /* accepted proposals among them (wavefront ballot + popcount reduction) */
    double kernel_ms;          /* device time of the sampler kernels (hipEvent)                         */
    double wall_ms;            /* host wall time of the call                                            */
    int32_t kernel_variant;    /* 0 generic (HBM state), 1 pre-built register kernel, 2 hiprtc-specialised register
                                  kernel, 3 pre-built cooperative kernel, 4 hiprtc-specialised cooperative kernel,
                                  5 hiprtc-specialised cooperative kernel for the dense Gaussian target (RWMH),
                                  6 a small ensemble (<= 1024 walkers, lane per walker) as ONE persistent block: a whole sampling
                                  call per launch, block barriers between the half-steps,
                                  7 the reference's sequential ensemble sweep (MHX_FLAG_EMCEE_SEQUENTIAL),
                                  8 matrix-core kernel: RWMH / MALA with one dense factor for all chains (dense Gaussian
                                  target and / or dense proposal) on v_mfma_*_16x16x4, reduction shape 4,
                                  9 scalar-factor form of the cooperative stretch move (dense precision factor: a lane owns a
                                  walker during A y, the wave-uniform factor entry is a DPP-broadcast / SGPR operand; reduction
                                  shape = reduce_lanes = waves per block),
                                  10 matrix-core form of the stretch move (dense precision factor, dim <= 64 in fp64 / 128 in fp32:
                                  4 lanes per walker, A y of the 16 walkers of a wave on v_mfma_*_16x16x4, the factor's operands
                                  fetched into registers from an image built once per run; reduction shape 4),
                                  11 a WAVE per chain (RWMH on the data-sum target MHX_TARGET_IID_NORMAL with few chains -- the
                                  reference's own README example, one chain: the 64 lanes split the likelihood's terms, the draws of
                                  64 steps are made side by side off the chain's critical path; reduction shape 64),
                                  12 RAM with a deferred factor (MHX_FLAG_RAM_DEFERRED) */
    int32_t launches;
    int32_t reduce_lanes;      /* lanes per chain in effect (1 unless a cooperative kernel runs) */
    int32_t dtype;             /* mhx_dtype of the run's context */
    int32_t normal_gen;        /* 0 Box-Muller, 1 ziggurat (MHX_FLAG_ZIGGURAT): how the run turns stream bits into normals */
    int32_t factor_band;       /* Ensemble runs on a dense-Gaussian target: the bandwidth of the precision factor the kernel exploits
                                  (0 = diagonal, 1 = bidiagonal: an AR(1) / Markov model, ...), -1 = none (dense form, other samplers) */
    int32_t tainted;           /* 1: a probe / fault-injection option of the tools build was set on the run's context -- the chains of
                                  such a run may be INVALID (timing probes skip work).  Always 0 from libmhx.so. */
    int32_t reserved_;
} mhx_stats;
int mhx_run_stats(mhx_run *run, mhx_stats *out);
/* dimension and number of chains (walkers) of a run; either pointer may be NULL */
int mhx_run_shape(const mhx_run *run, int32_t *dim, int32_t *nchains);

/* device pointers of the sample buffer of the last mhx_run_sample (for zero-copy consumers on the
 * same HIP runtime, e.g. diagnostics or a torch tensor view); valid until the next sample/destroy */
int mhx_run_device_samples(mhx_run *run, void **samples, void **accepted, int64_t *n_samples);

int mhx_run_destroy(mhx_run *run);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics on the device sample buffer of the last mhx_run_sample (what MCMCChains prints for the
 * reference, README.md:59-63).  Per parameter p (dim parameters, then lp), each array [dim+1]:
 *   sum_m  = sum_c m_c        sum_m2 = sum_c m_c^2        sum_v = sum_c s2_c
 * with m_c / s2_c the mean / unbiased variance of chain c over the N draws.  They are returned
 * un-normalised so that shards on several GPUs combine with ONE all-reduce of 3(dim+1)+1 doubles
 * (DESIGN.md section 8); R-hat and the between-chain ESS follow on the host:
 *   W = sum_v/C, Vm = (sum_m2 - sum_m^2/C)/(C-1), var+ = (N-1)/N W + Vm,
 *   R-hat = sqrt(var+/W), ESS_between = C var+/Vm.
 * ess[p] (optional) = C N / tau_p with tau_p from Geyer's initial monotone sequence on the multi-chain
 * autocorrelations rho_t = 1 - (W' - A_t)/var+ (Vehtari et al. 2021, eq. 10: A_t the lag-t autocovariance
 * averaged over the first `ess_chains` chains, W' = A_0, var+ from all chains; one chain: A_t/A_0), lags
 * 0..max_lag; it is returned NEGATED when the sequence was still positive at max_lag (|ess| is then an upper
 * bound).  With cfg.split every chain is two half-chains: C -> 2C, N -> floor(N/2) in all of the above. */
typedef struct {
    int32_t max_lag;    /* 0: skip the autocovariance ESS (ess[] = NaN) */
    int32_t ess_chains; /* chains used for the autocovariances, 0 = all */
    int32_t split;      /* 1: every chain counts as two half-chains of floor(N/2) draws (split R-hat, Vehtari et al.
                           2021): the sums then run over 2 nchains chains, and so do the autocovariances */
} mhx_diag_cfg;
int mhx_run_diagnostics(mhx_run *run, const mhx_diag_cfg *cfg, double *sum_m, double *sum_m2,
                        double *sum_v, double *ess /* each [dim+1], any may be NULL */);

/* ---- building blocks of ONE ensemble sharded over several GPUs (src/emcee.jl:14-24, parallel half-split form).
 * (mhx_emcee_half_step is stream-ordered, not blocking: the exchange and the next half-step queue behind it.)
 * Every rank holds the whole ensemble; for each half h of a sweep every rank moves its own slice
 * [begin, begin + count) of the moving half with mhx_emcee_half_step and the ranks then exchange the slices (an
 * all-gather of the walker-major rows and the lp / accept arrays that mhx_emcee_device_state exposes -- device
 * pointers, valid for the life of the run; pitch = floats per walker row); mhx_emcee_end_sweep advances the RNG
 * sweep counter after both halves.  Walkers carry their global index in the RNG counter, so any partition moves
 * exactly the walkers a single GPU would (advancedmh.jl_amd/mhx/dist.py: ShardedEnsemble). */
int mhx_emcee_half_step(mhx_run *run, int half, int begin, int count);
int mhx_emcee_end_sweep(mhx_run *run);
int mhx_emcee_device_state(mhx_run *run, void **xw, int32_t *pitch /* reals per walker row */, void **lp, uint32_t **acc_count,
                           uint8_t **last_acc);

/* The exchange step of the sharded ensemble as stream-ordered pieces (what mhx_comm_allgather_walkers does with RCCL; a
 * host that brings its own transport -- MPI, Distributed.jl -- uses them directly): every rank's slice of half `half` is
 * walkers [cnt q / world, cnt (q+1) / world) of that half; *stride = bytes of one rank's part of the staging buffer
 * ([world][stride], device memory), *stream = the hipStream_t the run's kernels are queued on.  pack writes this rank's
 * slice (rows, lp, accept counts, last accept flags) to `part`, unpack scatters the parts of all OTHER ranks into the run. */
int mhx_emcee_exchange_plan(mhx_run *run, int half, int world, size_t *stride, void **stream);
int mhx_emcee_exchange_pack(mhx_run *run, int half, int rank, int world, void *part);
int mhx_emcee_exchange_unpack(mhx_run *run, int half, int rank, int world, const void *stage, size_t stride);

/* ---------------------------------------------------------------------------------------------
 * Collectives (RCCL over xGMI; one process per GPU).  Chains shard by global id with no data-path collective; these
 * carry the acceptance totals and the R-hat / ESS sums of a sharded run (ONE all-reduce of 3(dim+1)+3 doubles per
 * reporting interval) and the half-step exchange of ONE ensemble sharded over the GPUs (ONE all-gather).  librccl is
 * resolved at run time.  mhx_comm_unique_id is called on one rank; the 128 bytes travel to the others by whatever the
 * host has (Distributed.jl, MPI, a file) and every rank calls mhx_comm_init with them. */
typedef struct mhx_comm mhx_comm;
#define MHX_COMM_ID_BYTES 128
int mhx_comm_unique_id(void *id128);
int mhx_comm_init(mhx_ctx *ctx, int rank, int world, const void *id128, mhx_comm **out);
/* ncclCommInitRank returns only when EVERY rank has called it: a missing rank, a stale id or a fabric that cannot connect the
 * ranks is a hang, not an error.  mhx_comm_init waits MHX_COMM_INIT_TIMEOUT_S for it, mhx_comm_init_timed `timeout_s` (<= 0: the
 * default); on expiry MHX_EHIP with a message that names the rank -- the caller can report it or fall back to another transport
 * (bench.py: RCCL behind this ABI, then torch.distributed's own nccl backend, then gloo).  mhx_comm_set_timeout bounds every
 * later blocking collective of the communicator the same way (default: the same 300 s). */
#define MHX_COMM_INIT_TIMEOUT_S 300.0
int mhx_comm_init_timed(mhx_ctx *ctx, int rank, int world, const void *id128, double timeout_s, mhx_comm **out);
int mhx_comm_set_timeout(mhx_comm *comm, double seconds);
int mhx_comm_destroy(mhx_comm *comm);
int mhx_comm_rank(const mhx_comm *comm, int *rank, int *world);
/* in-place sum over the ranks of n doubles in HOST memory (blocking) */
int mhx_comm_allreduce_sum(mhx_comm *comm, double *inout, size_t n);
/* this rank's slice [*begin, *begin + *count) of a half with cnt walkers */
int mhx_comm_slice(const mhx_comm *comm, int cnt, int *begin, int *count);
/* after mhx_emcee_half_step(run, half, begin, count) with the slice above: exchange the moved slices (stream-ordered) */
int mhx_comm_allgather_walkers(mhx_comm *comm, mhx_run *run, int half);

/* Rank-normalised bulk ESS and tail ESS (Vehtari et al. 2021, sections 4.1-4.3; what MCMCChains / ArviZ print as
 * ess_bulk, ess_tail) of the parameters params[0..nparams) (indices into the dim+1 rows, lp = dim) of the sample
 * buffer: the draws of one parameter are sorted on the device, replaced by the normal scores of their ranks (bulk)
 * and by the indicators of the 5 % / 95 % quantiles (tail: the smaller of the two), and the multi-chain ESS above
 * (cfg.max_lag, cfg.ess_chains, cfg.split) is taken of those series.  Negated values: as for ess[] above. */
int mhx_run_ess_bulk_tail(mhx_run *run, const mhx_diag_cfg *cfg, const int32_t *params, int32_t nparams,
                          double *ess_bulk, double *ess_tail /* each [nparams], either may be NULL */);

/* ---------------------------------------------------------------------------------------------
 * Many chains over many GPUs as ONE call from ONE process.  Replaces `sample(model, sampler, MCMCThreads(), N, nchains)`
 * (README.md:135-148: one task per chain) -- here one host thread per GPU behind the ABI.  A group is N member contexts (one
 * per entry of `devices`; entries may repeat -- several members on one device are legal, which is how the form is verified on a
 * one-GPU box), N persistent worker threads and the host-side sum of the statistics that ranks of a multi-process run all-reduce
 * over RCCL (mhx_comm_allreduce_sum).  Chains shard by global id with no data-path collective, so member i's run is created by
 * the caller on mhx_group_ctx(g, i) like any run, with cfg.first_chain / cfg.nchains from mhx_group_shard (ensembles: one per
 * member, distinct ensemble_id), and attached; the group then drives all members CONCURRENTLY:
 *     mhx_group_init / _sample / _sample_to_host   == mhx_run_init / _sample / _sample_to_host of every member, side by side
 *     mhx_group_stats        transitions / accepted summed, kernel_ms = the slowest member, wall_ms = the whole call
 *     mhx_group_diagnostics  the sums of mhx_run_diagnostics added in member order (+ the chain count): R-hat over ALL chains
 * The union of the members' chains is bit for bit the unsharded run (tests/test_gpu_group.py).  A failing member's message is
 * returned with its index and device.  Ownership: the group owns contexts and threads; the caller destroys runs and targets
 * BEFORE mhx_group_destroy.  One group call at a time (a group is a handle: not thread-safe). */
typedef struct mhx_group mhx_group;
int mhx_group_create(const int32_t *devices, int32_t n, int dtype /* mhx_dtype */, mhx_group **out);
int mhx_group_destroy(mhx_group *g);
int mhx_group_size(const mhx_group *g, int32_t *n);
int mhx_group_ctx(mhx_group *g, int32_t i, mhx_ctx **ctx);      /* borrowed: valid until mhx_group_destroy */
/* member i's contiguous block of `nchains_total` global chain ids (sizes differ by at most one); also tells member i's context that
 * its runs are shards of a run of nchains_total chains (option TOTAL_CHAINS: the kernel form is picked for the whole run) */
int mhx_group_shard(const mhx_group *g, int64_t nchains_total, int32_t i, uint64_t *first_chain, int32_t *nchains);
int mhx_group_attach(mhx_group *g, mhx_run *const *runs /* [n]: runs[i] was created on mhx_group_ctx(g, i); equal dim */);
int mhx_group_run(mhx_group *g, int32_t i, mhx_run **run);
int mhx_group_init(mhx_group *g, const void *const *initial_params /* NULL, or [n] host pointers, each NULL or [dim][nchains_i] */);
int mhx_group_sample(mhx_group *g, const mhx_schedule *sched, int save_samples);
int mhx_group_sample_to_host(mhx_group *g, const mhx_schedule *sched, void *const *samples /* [n] host tensors [n_samples][dim+1][nchains_i] */,
                             uint8_t *const *accepted /* NULL, or [n] (entries may be NULL) */, int32_t slab_samples);
int mhx_group_stats(mhx_group *g, mhx_stats *out);
/* ess (optional): the members' ESS added (each from the autocovariances of its own shard); negated when any member's was.
 * *n_chains = the chains behind the sums (2x with cfg.split). */
int mhx_group_diagnostics(mhx_group *g, const mhx_diag_cfg *cfg, double *sum_m, double *sum_m2, double *sum_v, double *ess /* each [dim+1] or NULL */,
                          int64_t *n_chains);
/* bulk / tail ESS of every member (ranks pooled within a member's shard) added over the members */
int mhx_group_ess_bulk_tail(mhx_group *g, const mhx_diag_cfg *cfg, const int32_t *params, int32_t nparams, double *ess_bulk, double *ess_tail);

#ifdef __cplusplus
}
#endif
#endif /* MHX_H */
