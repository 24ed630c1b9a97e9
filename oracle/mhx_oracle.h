/*
 * mhx_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the AdvancedMH.jl hot path (reference @ /root/reference, v0.8.8):
 *   src/mh-core.jl:76-117                      RWMH initial step + propose/logdensity/accept step
 *   src/proposal.jl:13-25,41-64,190-196        RandomWalkProposal draw / Hastings ratio (== 0)
 *   src/emcee.jl:1-102                         Ensemble sweep + stretch move
 *   src/RobustAdaptiveMetropolis.jl:123-278    RAM inner step, rank-1 Cholesky adapt, step/step_warmup
 *   ext/AdvancedMHMCMCChainsExt.jl:80-121      sample tensor layout (iterations x (params.., lp) x chains)
 * plus the upstream pieces the package calls (AbstractMCMC.mcmcsample schedule, Distributions
 * MvNormal rand/logpdf, LinearAlgebra.lowrankupdate/lowrankdowndate) restated from their
 * published algorithms -- marked [upstream, restated] where they occur.
 *
 * PARITY STATUS: "parity unpinned" at bit level versus the Julia reference: the reference is
 * pure Julia, no julia binary exists in this image, and its tests hold no golden vectors (all
 * are statistical).  Julia's Xoshiro256++/ziggurat stream is replaced here by Philox4x32-10 with
 * the arithmetic spec of DESIGN.md section 3.  What IS pinned: Philox against the Random123
 * known-answer vectors, the transcendental polynomials against libm, the target log-densities
 * against scipy, the rank-1 Cholesky up/downdate against numpy.linalg.cholesky, and the samplers
 * against every analytic known answer the reference's own tests use (tests/test_oracle_*.py) and, step
 * by step, against a second float64 numpy / scipy restatement of the reference's formulas
 * (tests/test_reference_restatement.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef MHX_ORACLE_H
#define MHX_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* `real` is float in libmhx_oracle.so (ORC_F64=0) and double in libmhx_oracle64.so (ORC_F64=1, the reference's Float64) */
#ifndef ORC_F64
#define ORC_F64 0
#endif
#if ORC_F64
typedef double real;
#else
typedef float real;
#endif

/* ---- arithmetic spec primitives (exported so tests can pin them) ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
real orc_log(real x);
real orc_exp(real x);
#if ORC_F64
void   orc_sincos2pi_u64(uint32_t hi, uint32_t lo, double *s, double *c);   /* angle 2 pi (hi:lo) / 2^64 */
double orc_u01_open(uint32_t hi, uint32_t lo);   /* (0,1): (k + 1/2) 2^-52, k = hi:lo >> 12 */
double orc_u01_half(uint32_t hi, uint32_t lo);   /* [0,1): k 2^-52                          */
void   orc_normal_pair(const uint32_t w[4], double *n0, double *n1);
#else
void  orc_sincos2pi_u32(uint32_t k, float *s, float *c);
float orc_u01_open(uint32_t k);      /* (0,1]  : fmaf((float)k, 2^-32, 2^-33) */
float orc_u01_half(uint32_t k);      /* [0,1)  : (k>>8) * 2^-24               */
void  orc_normal_pair(uint32_t k0, uint32_t k1, float *n0, float *n1);
#endif
/* d standard normals of (seed, chain, step, stream) -- the proposal noise of one chain-step */
void orc_normals(uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, int d, real *out);
real orc_accept_logu(uint64_t seed, uint64_t chain, uint32_t step);

/* RNG stream tags (c3 = tag<<28 | block) */
enum { ORC_STREAM_PROPOSAL = 0, ORC_STREAM_ACCEPT = 1, ORC_STREAM_INIT = 2, ORC_STREAM_EMCEE = 3 };

/* ---- targets ---- */
enum {
    ORC_TARGET_ISO_GAUSS   = 0,  /* params: none                                                  */
    ORC_TARGET_CORR_GAUSS  = 1,  /* params: A = inv(chol(Sigma)) packed lower row-major, d(d+1)/2 */
    ORC_TARGET_IID_NORMAL  = 2,  /* theta=(mu,sigma); params: n data points                       */
    ORC_TARGET_BANANA      = 3,  /* params: {b}                                                   */
    ORC_TARGET_FUNNEL      = 4,  /* params: none                                                  */
    ORC_TARGET_CALLBACK    = 100 /* user function pointer (DensityModel(f))                       */
};
typedef real (*orc_logdensity_fn)(const real *x, int d, const void *data);

typedef struct {
    int kind;
    int dim;
    const real *params;     /* kind-specific, see above */
    int nparams;
    orc_logdensity_fn fn;    /* ORC_TARGET_CALLBACK */
    const void *fn_data;
    /* reduction shape of the separable targets (ISO_GAUSS, BANANA, FUNNEL): L lanes per chain.
     * lane l accumulates the Philox blocks b = l, l+L, ... (4 dimensions each) sequentially, the L
     * partial sums are combined by an xor-butterfly (offsets 1, 2, 4, ...).  0 or 1 = sequential.
     * CORR_GAUSS (cooperative ensemble kernel): lane l owns the rows i = l, l+L, ... of A x. */
    int reduce_lanes;
} orc_target;

real orc_target_eval(const orc_target *t, const real *x);

/* ---- proposals (RandomWalkProposal{_, <:MvNormal}, zero mean) ---- */
enum { ORC_PROP_ISO = 0, ORC_PROP_DIAG = 1, ORC_PROP_DENSE = 2 };
typedef struct {
    int kind;
    real scale;          /* ISO: sigma                                  */
    const real *vec;     /* DIAG: sigma_k[d]; DENSE: packed lower L     */
    const real *mean;    /* NULL = zero mean; else mu[d]: the random walk drifts and the Hastings ratio
                             q(x | y) - q(y | x) of src/proposal.jl:58-64,190-192 is no longer zero */
    int is_static;        /* StaticProposal (src/proposal.jl:9-11,66-83): the candidate is a draw mu + L z that
                             ignores the current state; ratio = logpdf(p, x) - logpdf(p, y)            */
    int normal_gen;       /* 0 Box-Muller; 1 (fp64 build only) the table ziggurat of spec 3.11 (MHX_FLAG_ZIGGURAT of the ABI):
                             how the INIT and PROPOSAL stream bits of an RWMH run become standard normals */
} orc_proposal;

/* ---- schedule [upstream AbstractMCMC.mcmcsample, restated] ---- */
typedef struct {
    int n_samples;        /* N                      */
    int discard_initial;
    int thinning;
    int num_warmup;
} orc_schedule;
/* total transitions after the initial state, and how many leading ones are warm-up (adapting) */
void orc_schedule_counts(const orc_schedule *s, int64_t *n_transitions, int64_t *n_adapt);

/* ---- samplers.  Output tensors: samples[N][d+1][C] (chain fastest, lp last row),
 *      accepted[N][C].  chains are global ids first_chain .. first_chain+C-1. ---- */
int orc_rwmh(const orc_target *t, const orc_proposal *p, const orc_schedule *s,
             uint64_t seed, uint64_t first_chain, int nchains,
             const real *init /* [d][C] or NULL */, real *samples, uint8_t *accepted,
             real *final_x /* [d][C] */, real *final_lp, uint32_t *accept_counts);

/* Ensemble(W, StretchProposal(prior, a)). mode 0 = reference-faithful sequential sweep
 * (src/emcee.jl:39-58), mode 1 = parallel half-split (what the HIP kernel runs). */
int orc_emcee(const orc_target *t, real a, int mode, const orc_schedule *s,
              uint64_t seed, uint64_t ensemble_id, int nwalkers,
              const real *init /* [d][W], or NULL: W draws from `prior` (src/emcee.jl:29-34) */,
              const orc_proposal *prior /* the (Mv)Normal StretchProposal wraps, or NULL */, real *samples, uint8_t *accepted,
              real *final_x, real *final_lp, uint32_t *accept_counts);

typedef struct {
    real alpha;          /* target acceptance, 0.234 */
    real gamma;          /* 0.6 */
    real eig_lo, eig_hi; /* 0, +inf */
} orc_ram_cfg;
/* S: packed lower row-major [C][d(d+1)/2], in = initial factor (NULL -> identity), out = final.
 * logalpha_trace (optional) [n_transitions][C]. status[C]: bit0 = a downdate hit s^2>1. */
int orc_ram(const orc_target *t, const orc_ram_cfg *cfg, const orc_schedule *s,
            uint64_t seed, uint64_t first_chain, int nchains,
            const real *init /* [d][C] or NULL -> randn */, const real *S_in, real *S_out,
            real *samples, uint8_t *accepted, real *final_x, real *final_lp,
            uint32_t *accept_counts, uint8_t *status, real *diag_min, real *diag_max);

/* RAM with a deferred factor (the twin of MHX_FLAG_RAM_DEFERRED): the chain of orc_ram in exact arithmetic,
 * other rounding -- up to K accepted rank-1 updates stay pending as O(d) triples and are folded into the stored factor in one
 * pass; flush_at[nflush]: transitions (ascending) after which a flush is forced (the engine's launch ends). dim <= 256. */
int orc_ram_deferred(const orc_target *t, const orc_ram_cfg *cfg, const orc_schedule *s,
                     uint64_t seed, uint64_t first_chain, int nchains,
                     const real *init, const real *S_in, real *S_out,
                     real *samples, uint8_t *accepted, real *final_x, real *final_lp,
                     uint32_t *accept_counts, uint8_t *status, real *diag_min, real *diag_max,
                     int K, const int64_t *flush_at, int nflush);

/* value and gradient of a catalogue target (MALA); g[d] out.  CALLBACK targets: fn_grad in `t->fn_data`
 * is not supported -- user gradients are exercised through gcc-built sources in tests/user_targets.py. */
typedef real (*orc_logdensity_grad_fn)(const real *x, real *g, int d, const void *data);
real orc_target_grad(const orc_target *t, const real *x, real *g, orc_logdensity_grad_fn user);

/* MALA(g -> MvNormal((sigma2/2) g, sigma2 I)): src/MALA.jl:54-93.  init [d][C] is required (:37). */
int orc_mala(const orc_target *t, orc_logdensity_grad_fn user, real sigma2, const orc_schedule *s,
             uint64_t seed, uint64_t first_chain, int nchains, const real *init,
             real *samples, uint8_t *accepted, real *final_x, real *final_lp, uint32_t *accept_counts, int normal_gen /* 0 Box-Muller, 1 ziggurat (fp64) */);

/* the d standard normals of (seed, chain, step, stream) by generator `gen` (0 Box-Muller, 1 ziggurat) */
void orc_normals_gen(int gen, uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, int d, real *out);
/* standard normal number n (0-based) of (seed, chain, step, stream) by the ziggurat generator (spec 3.11: 1024 layers and 64 bits per
 * normal in the fp64 build, 256 layers and 32 bits in the fp32 build) */
real orc_zig_normal(uint64_t seed, uint64_t chain, uint32_t step, uint32_t stream, uint32_t n);

/* Trace sink for tests (thread-local; all three optional, [n_samples][nchains], NULL switches off):
 * margin = smallest |logu - logalpha| over the transitions that led to a saved slot; logalpha / eta = RAM's state.logalpha
 * and state.eta after the saved transition (src/RobustAdaptiveMetropolis.jl:99-114, 141-147). */
void orc_set_trace(real *margin, real *logalpha, real *eta);

/* rank-1 Cholesky update (sign=+1) / downdate (sign=-1) of a packed lower factor, in place.
 * returns 0, or i+1 if the downdate failed at column i (S is then partially modified). */
int orc_chol_rank1(real *S, real *w, int d, int sign);

#ifdef __cplusplus
}
#endif
#endif
