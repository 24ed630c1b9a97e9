// mhx_group.cpp -- many chains over many GPUs as ONE call from ONE process (include/mhx.h: mhx_group_*).
//
// Replaces `sample(model, sampler, MCMCThreads(), N, nchains)` of the reference (README.md:135-148): there one Julia task per
// chain, here one host thread per GPU behind the C ABI.  Chains shard by global id with no data-path collective (DESIGN.md
// section 8), so a group is N member contexts (mhx_ctx, one per entry of `devices` -- entries may repeat: several members on one
// device are legal, which is how the form is verified on a one-GPU box), N persistent worker threads, and the host-side sum of
// the 3(dim+1)+3 doubles a multi-process run would all-reduce over RCCL (mhx_comm_*).  Everything here goes through the public
// entry points of mhx.h: a group drives member runs exactly like N processes would.
#include "mhx_impl.h"

#include <algorithm>
#include "mhx_host_expand.h"    // mhx_host_usable_cpus

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct member {
    int device = 0;
    mhx_ctx* ctx = nullptr;
    mhx_run* run = nullptr;
    int dim = 0, nchains = 0;
    // worker
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
    std::string err;
};

void worker(member* m)
{
    for (;;) {
        std::function<int()> job;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv.wait(lk, [&] { return m->has_job || m->quit; });
            if (m->quit) return;
            job = m->job;
        }
        const int rc = job();
        const std::string err = rc ? std::string(mhx_last_error()) : std::string();      // mhx_last_error is per thread: carry it over
        {
            std::lock_guard<std::mutex> lk(m->mu);
            m->rc = rc; m->err = err;
            m->has_job = false; m->done = true;
        }
        m->cv.notify_all();
    }
}

}  // namespace

struct mhx_group {
    int32_t dtype = MHX_F64;
    std::vector<std::unique_ptr<member>> mem;
    bool attached = false;
    double wall_ms = 0.0;                   // of the last mhx_group_sample / _sample_to_host (all members, start to last finish)
};

namespace {

// run fn(i) on every member's own thread, concurrently; the first failure (lowest member) becomes the caller's error
int for_all(mhx_group* g, const char* who, const std::function<int(int)>& fn)
{
    const int n = (int)g->mem.size();
    for (int i = 0; i < n; ++i) {
        member* m = g->mem[i].get();
        {
            std::lock_guard<std::mutex> lk(m->mu);
            m->job = [fn, i] { return fn(i); };
            m->done = false; m->has_job = true;
        }
        m->cv.notify_all();
    }
    int rc = MHX_OK, bad = -1;
    std::string err;
    for (int i = 0; i < n; ++i) {
        member* m = g->mem[i].get();
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv.wait(lk, [&] { return m->done; });
        if (m->rc && !rc) { rc = m->rc; bad = i; err = m->err; }
    }
    if (rc) return mhx_fail(rc, "%s: member %d (device %d): %s", who, bad, g->mem[bad]->device, err.c_str());
    return MHX_OK;
}

int need_runs(const mhx_group* g, const char* who)
{
    if (!g) return mhx_fail(MHX_EINVAL, "%s: group is NULL", who);
    if (!g->attached) return mhx_fail(MHX_ESTATE, "%s: no runs attached (mhx_group_attach)", who);
    return MHX_OK;
}

}  // namespace

extern "C" int mhx_group_create(const int32_t* devices, int32_t n, int dtype, mhx_group** out)
{
    if (!out) return mhx_fail(MHX_EINVAL, "mhx_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 1024) return mhx_fail(MHX_EINVAL, "mhx_group_create: %d members", (int)n);
    std::unique_ptr<mhx_group> g(new mhx_group);
    g->dtype = dtype;
    int rc = MHX_OK;
    for (int i = 0; i < n && !rc; ++i) {
        std::unique_ptr<member> m(new member);
        m->device = devices[i];
        rc = mhx_ctx_create(devices[i], dtype, &m->ctx);
        if (!rc) g->mem.push_back(std::move(m));
    }
    if (rc) {                                                       // the message of the failing mhx_ctx_create stays in place
        const std::string err = mhx_last_error();
        for (auto& m : g->mem) (void)mhx_ctx_destroy(m->ctx);
        return mhx_fail(rc, "mhx_group_create: member %d: %s", (int)g->mem.size(), err.c_str());
    }
    // the members return their samples at the same time: the host threads that expand accept-compacted blocks are shared out
    // (each member's own pool would otherwise be sized for the whole machine: n x 16 threads on a quota of 16 CPUs)
    if (n > 1) {
        char buf[16];
        snprintf(buf, sizeof buf, "%d", std::max(1, mhx_host_usable_cpus() / (int)n));
        for (auto& m : g->mem) (void)mhx_ctx_set_option(m->ctx, "HOST_THREADS", buf);
    }
    for (auto& m : g->mem) m->th = std::thread(worker, m.get());
    *out = g.release();
    return MHX_OK;
}

extern "C" int mhx_group_destroy(mhx_group* g)
{
    if (!g) return MHX_OK;
    for (auto& m : g->mem) {
        {
            std::lock_guard<std::mutex> lk(m->mu);
            m->quit = true;
        }
        m->cv.notify_all();
        if (m->th.joinable()) m->th.join();
    }
    int rc = MHX_OK;
    for (auto& m : g->mem) {
        const int r = mhx_ctx_destroy(m->ctx);
        if (r && !rc) rc = r;
    }
    delete g;
    return rc;
}

extern "C" int mhx_group_size(const mhx_group* g, int32_t* n)
{
    if (!g || !n) return mhx_fail(MHX_EINVAL, "mhx_group_size: NULL argument");
    *n = (int32_t)g->mem.size();
    return MHX_OK;
}

extern "C" int mhx_group_ctx(mhx_group* g, int32_t i, mhx_ctx** ctx)
{
    if (!g || !ctx) return mhx_fail(MHX_EINVAL, "mhx_group_ctx: NULL argument");
    if (i < 0 || i >= (int32_t)g->mem.size()) return mhx_fail(MHX_EINVAL, "mhx_group_ctx: member %d of %zu", (int)i, g->mem.size());
    *ctx = g->mem[i]->ctx;
    return MHX_OK;
}

// contiguous blocks of global chain ids, sizes differing by at most one (the rule of mhx/dist.py: shard_chains)
extern "C" int mhx_group_shard(const mhx_group* g, int64_t nchains_total, int32_t i, uint64_t* first_chain, int32_t* nchains)
{
    if (!g) return mhx_fail(MHX_EINVAL, "mhx_group_shard: group is NULL");
    const int64_t w = (int64_t)g->mem.size();
    if (i < 0 || i >= w || nchains_total < 0) return mhx_fail(MHX_EINVAL, "mhx_group_shard: member %d of %lld, %lld chains", (int)i, (long long)w, (long long)nchains_total);
    const int64_t base = nchains_total / w, rem = nchains_total % w;
    if (first_chain) *first_chain = (uint64_t)(i * base + (i < rem ? i : rem));
    if (nchains) *nchains = (int32_t)(base + (i < rem ? 1 : 0));
    // the member's runs are parts of a run of nchains_total chains: the engine picks the kernel form (the summation order) for THAT
    // count, so the union of the members is the unsharded run bit for bit whatever the shard sizes (option TOTAL_CHAINS)
    char buf[32];
    snprintf(buf, sizeof buf, "%lld", (long long)nchains_total);
    return mhx_ctx_set_option(g->mem[(size_t)i]->ctx, "TOTAL_CHAINS", buf);
}

extern "C" int mhx_group_attach(mhx_group* g, mhx_run* const* runs)
{
    if (!g || !runs) return mhx_fail(MHX_EINVAL, "mhx_group_attach: NULL argument");
    const int n = (int)g->mem.size();
    std::vector<int> dims(n), cnt(n);
    for (int i = 0; i < n; ++i) {
        if (!runs[i]) return mhx_fail(MHX_EINVAL, "mhx_group_attach: run %d is NULL", i);
        int32_t d = 0, c = 0;
        const int rc = mhx_run_shape(runs[i], &d, &c);
        if (rc) return rc;
        if (d != (i ? dims[0] : d)) return mhx_fail(MHX_EINVAL, "mhx_group_attach: run %d has dimension %d, run 0 has %d", i, (int)d, dims[0]);
        dims[i] = d; cnt[i] = c;
    }
    for (int i = 0; i < n; ++i) { g->mem[i]->run = runs[i]; g->mem[i]->dim = dims[i]; g->mem[i]->nchains = cnt[i]; }
    g->attached = true;
    return MHX_OK;
}

extern "C" int mhx_group_run(mhx_group* g, int32_t i, mhx_run** run)
{
    int rc = need_runs(g, "mhx_group_run");
    if (rc) return rc;
    if (!run || i < 0 || i >= (int32_t)g->mem.size()) return mhx_fail(MHX_EINVAL, "mhx_group_run: member %d of %zu", (int)i, g->mem.size());
    *run = g->mem[i]->run;
    return MHX_OK;
}

extern "C" int mhx_group_init(mhx_group* g, const void* const* initial_params)
{
    int rc = need_runs(g, "mhx_group_init");
    if (rc) return rc;
    return for_all(g, "mhx_group_init", [g, initial_params](int i) {
        return mhx_run_init(g->mem[i]->run, initial_params ? initial_params[i] : nullptr);
    });
}

extern "C" int mhx_group_sample(mhx_group* g, const mhx_schedule* s, int save_samples)
{
    int rc = need_runs(g, "mhx_group_sample");
    if (rc) return rc;
    if (!s) return mhx_fail(MHX_EINVAL, "mhx_group_sample: schedule is NULL");
    const mhx_schedule sched = *s;
    const auto t0 = std::chrono::steady_clock::now();
    rc = for_all(g, "mhx_group_sample", [g, sched, save_samples](int i) { return mhx_run_sample(g->mem[i]->run, &sched, save_samples); });
    g->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" int mhx_group_sample_to_host(mhx_group* g, const mhx_schedule* s, void* const* samples, uint8_t* const* accepted, int32_t slab_samples)
{
    int rc = need_runs(g, "mhx_group_sample_to_host");
    if (rc) return rc;
    if (!s || !samples) return mhx_fail(MHX_EINVAL, "mhx_group_sample_to_host: NULL argument");
    const mhx_schedule sched = *s;
    const auto t0 = std::chrono::steady_clock::now();
    rc = for_all(g, "mhx_group_sample_to_host", [g, sched, samples, accepted, slab_samples](int i) {
        return mhx_run_sample_to_host(g->mem[i]->run, &sched, samples[i], accepted ? accepted[i] : nullptr, slab_samples);
    });
    g->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" int mhx_group_stats(mhx_group* g, mhx_stats* out)
{
    int rc = need_runs(g, "mhx_group_stats");
    if (rc) return rc;
    if (!out) return mhx_fail(MHX_EINVAL, "mhx_group_stats: out is NULL");
    mhx_stats tot{};
    for (size_t i = 0; i < g->mem.size(); ++i) {
        mhx_stats st{};
        if ((rc = mhx_run_stats(g->mem[i]->run, &st))) return rc;
        if (i == 0) tot = st;
        else {
            tot.transitions += st.transitions;
            tot.accepted += st.accepted;
            tot.kernel_ms = std::max(tot.kernel_ms, st.kernel_ms);         // the members ran side by side: the slowest one
            tot.launches = std::max(tot.launches, st.launches);
            tot.tainted |= st.tainted;
        }
    }
    tot.wall_ms = g->wall_ms;
    *out = tot;
    return MHX_OK;
}

// the sums of mhx_run_diagnostics added over the members in member order (what ranks all-reduce over RCCL: the same
// 3(dim+1) doubles + the chain count); ess: the members' Geyer ESS added (each from its own shard's autocovariances),
// negated when any member's sequence was still positive at max_lag
extern "C" int mhx_group_diagnostics(mhx_group* g, const mhx_diag_cfg* cfg, double* sum_m, double* sum_m2, double* sum_v, double* ess,
                                     int64_t* n_chains)
{
    int rc = need_runs(g, "mhx_group_diagnostics");
    if (rc) return rc;
    if (!cfg) return mhx_fail(MHX_EINVAL, "mhx_group_diagnostics: cfg is NULL");
    const int n = (int)g->mem.size(), d1 = g->mem[0]->dim + 1;
    const mhx_diag_cfg c = *cfg;
    std::vector<double> buf((size_t)n * 4 * d1, 0.0);
    const bool want_ess = ess != nullptr;
    rc = for_all(g, "mhx_group_diagnostics", [g, c, &buf, d1, want_ess](int i) {
        double* b = buf.data() + (size_t)i * 4 * d1;
        return mhx_run_diagnostics(g->mem[i]->run, &c, b, b + d1, b + 2 * d1, want_ess ? b + 3 * d1 : nullptr);
    });
    if (rc) return rc;
    int64_t chains = 0;
    for (int i = 0; i < n; ++i) chains += (int64_t)g->mem[i]->nchains * (c.split ? 2 : 1);
    for (int p = 0; p < d1; ++p) {
        double a = 0, b = 0, v = 0, e = 0;
        bool trunc = false, nan = false;
        for (int i = 0; i < n; ++i) {
            const double* m = buf.data() + (size_t)i * 4 * d1;
            a += m[p]; b += m[d1 + p]; v += m[2 * d1 + p];
            if (want_ess) { const double x = m[3 * d1 + p]; if (x != x) nan = true; else { e += std::fabs(x); trunc |= x < 0; } }
        }
        if (sum_m) sum_m[p] = a;
        if (sum_m2) sum_m2[p] = b;
        if (sum_v) sum_v[p] = v;
        if (ess) ess[p] = nan ? std::nan("") : (trunc ? -e : e);
    }
    if (n_chains) *n_chains = chains;
    return MHX_OK;
}

extern "C" int mhx_group_ess_bulk_tail(mhx_group* g, const mhx_diag_cfg* cfg, const int32_t* params, int32_t nparams, double* ess_bulk,
                                       double* ess_tail)
{
    int rc = need_runs(g, "mhx_group_ess_bulk_tail");
    if (rc) return rc;
    if (!cfg || (!params && nparams) || nparams < 0) return mhx_fail(MHX_EINVAL, "mhx_group_ess_bulk_tail: bad argument");
    const int n = (int)g->mem.size();
    const mhx_diag_cfg c = *cfg;
    std::vector<double> buf((size_t)n * 2 * nparams, 0.0);
    rc = for_all(g, "mhx_group_ess_bulk_tail", [g, c, params, nparams, &buf](int i) {
        double* b = buf.data() + (size_t)i * 2 * nparams;
        return mhx_run_ess_bulk_tail(g->mem[i]->run, &c, params, nparams, b, b + nparams);
    });
    if (rc) return rc;
    for (int k = 0; k < 2; ++k) {
        double* dst = k ? ess_tail : ess_bulk;
        if (!dst) continue;
        for (int p = 0; p < nparams; ++p) {
            double e = 0;
            bool trunc = false;
            for (int i = 0; i < n; ++i) { const double x = buf[(size_t)i * 2 * nparams + (size_t)k * nparams + p]; e += std::fabs(x); trunc |= x < 0; }
            dst[p] = trunc ? -e : e;
        }
    }
    return MHX_OK;
}
