/*
 * mhx_oracle_mt.c -- the CPU BASELINE driver of bench.py (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * What the reference does with many chains on a host is `sample(model, spl, MCMCThreads(), N, nchains)`
 * (/root/reference/README.md:135-148): every chain is an independent task, each keeps ITS OWN vector of transitions.
 * This file times exactly that shape with the oracle's samplers (mhx_oracle.c, unchanged): `nthreads` POSIX threads take
 * chains one at a time off a shared counter and run the whole schedule of a chain with ONE call of orc_rwmh / orc_ram with
 * nchains = 1 -- so the record of a chain is a contiguous [N][d+1] array (the chain's own Vector{Transition}), reused by
 * the thread for its next chain, instead of the device's chain-fastest [N][d+1][C] tensor that a per-chain loop can only
 * fill one cache line per element.  No Python in the timed region (the round-3 baseline ran its threads through ctypes).
 *
 * Returns the wall time of the parallel region; per-thread busy seconds let the caller see whether the host really gave
 * it `nthreads` cores (a cgroup quota below the affinity mask shows up as busy time >> wall time x cores granted).
 */
#define _GNU_SOURCE
#include "mhx_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int kind;                       /* 0 rwmh, 1 ram */
    const orc_target *t;
    const orc_proposal *p;
    const orc_ram_cfg *cfg;
    const orc_schedule *s;
    uint64_t seed, first_chain;
    int nchains, save;
    const real *init1;              /* [d] one initial state shared by all chains, or NULL */
    long next;                      /* shared chain counter */
} mt_job;

typedef struct {
    mt_job *job;
    double busy;                    /* CPU seconds of this thread (CLOCK_THREAD_CPUTIME_ID) */
    long done;
} mt_thread;

static double now_mono(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double now_thread(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *mt_worker(void *arg)
{
    mt_thread *th = (mt_thread *)arg;
    mt_job *j = th->job;
    const int d = j->t->dim, N = j->s->n_samples;
    /* the chain's own record: [N][d+1] contiguous (nchains = 1 makes the oracle's [N][d+1][C] exactly that) */
    real *rec = j->save ? malloc(sizeof(real) * (size_t)N * (size_t)(d + 1)) : NULL;
    uint8_t *acc = j->save ? malloc((size_t)N) : NULL;
    real *fx = malloc(sizeof(real) * (size_t)d), flp;
    real *S = j->kind == 1 ? malloc(sizeof(real) * (size_t)d * (size_t)(d + 1) / 2) : NULL;
    real *dmin = j->kind == 1 ? malloc(sizeof(real) * (size_t)d * 2) : NULL;
    uint32_t cnt;
    uint8_t status;
    const double t0 = now_thread();
    for (;;) {
        long c = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (c >= j->nchains) break;
        if (j->kind == 0)
            orc_rwmh(j->t, j->p, j->s, j->seed, j->first_chain + (uint64_t)c, 1, j->init1, rec, acc, fx, &flp, &cnt);
        else
            orc_ram(j->t, j->cfg, j->s, j->seed, j->first_chain + (uint64_t)c, 1, j->init1, NULL, S, rec, acc, fx, &flp, &cnt,
                    &status, dmin, dmin + d);
        ++th->done;
    }
    th->busy = now_thread() - t0;
    free(rec); free(acc); free(fx); free(S); free(dmin);
    return NULL;
}

static double mt_run(mt_job *job, int nthreads, double *busy /* [nthreads] or NULL */)
{
    if (nthreads < 1) nthreads = 1;
    pthread_t *tid = malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_thread *th = calloc((size_t)nthreads, sizeof(mt_thread));
    job->next = 0;
    const double t0 = now_mono();
    int started = 0;
    for (int i = 0; i < nthreads; ++i) {
        th[i].job = job;
        if (pthread_create(&tid[i], NULL, mt_worker, &th[i]) != 0) break;
        ++started;
    }
    if (started == 0) {                       /* no thread could be created: run on the caller */
        th[0].job = job;
        mt_worker(&th[0]);
        started = 1;
    } else {
        for (int i = 0; i < started; ++i) pthread_join(tid[i], NULL);
    }
    const double wall = now_mono() - t0;
    if (busy) for (int i = 0; i < nthreads; ++i) busy[i] = i < started ? th[i].busy : 0.0;
    free(tid); free(th);
    return wall;
}

/* `nchains` independent RWMH chains (global ids first_chain ...) on `nthreads` threads; save != 0: every chain records its
 * N x (d+1) states into its thread's contiguous buffer (what `sample` keeps per chain).  Returns wall seconds. */
double orc_mt_rwmh(const orc_target *t, const orc_proposal *p, const orc_schedule *s, uint64_t seed, uint64_t first_chain,
                   int nchains, int nthreads, int save, const real *init1, double *busy)
{
    mt_job job;
    memset(&job, 0, sizeof job);
    job.kind = 0; job.t = t; job.p = p; job.s = s; job.seed = seed; job.first_chain = first_chain;
    job.nchains = nchains; job.save = save; job.init1 = init1;
    return mt_run(&job, nthreads, busy);
}

/* the same for RobustAdaptiveMetropolis chains (each with its own factor, identity start) */
double orc_mt_ram(const orc_target *t, const orc_ram_cfg *cfg, const orc_schedule *s, uint64_t seed, uint64_t first_chain,
                  int nchains, int nthreads, int save, const real *init1, double *busy)
{
    mt_job job;
    memset(&job, 0, sizeof job);
    job.kind = 1; job.t = t; job.cfg = cfg; job.s = s; job.seed = seed; job.first_chain = first_chain;
    job.nchains = nchains; job.save = save; job.init1 = init1;
    return mt_run(&job, nthreads, busy);
}
