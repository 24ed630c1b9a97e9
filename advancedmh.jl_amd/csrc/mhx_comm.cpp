// mhx_comm.cpp -- the collectives of a run sharded over the GPUs of a node, behind the C ABI (include/mhx.h).
//
// One process per GPU; chains shard by global id with no data-path collective.  What needs
// exchanging: the acceptance totals and the R-hat / ESS sums (ONE all-reduce of 3(dim+1)+3 doubles per reporting
// interval) and, for ONE ensemble sharded over the GPUs, the moved slice of the walkers after every half-step (ONE
// all-gather of a packed staging buffer).  RCCL over xGMI; librccl is resolved at run time (dlopen) so that libmhx.so
// loads on a box without it and so that a process that already carries an RCCL (torch ships its own) uses that one.
#include "mhx_impl.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace {

struct rccl_api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;        // optional: what the communicator itself reports
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                    // optional: release a communicator whose collective hangs
    bool ok = false;
};

rccl_api g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    const char* override_path = getenv("MHX_RCCL_LIB");
    const char* names[] = {override_path, "librccl.so.1", "librccl.so"};
    // an RCCL that is already part of the process first (RTLD_NOLOAD), then a fresh load
    for (int pass = 0; pass < 2 && !g_rccl.lib; ++pass)
        for (const char* n : names)
            if (n && !g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (!g_rccl.lib) return;
    auto sym = [&](const char* s) { return dlsym(g_rccl.lib, s); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))sym("ncclCommUserRank");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))sym("ncclCommAbort");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.AllGather &&
                g_rccl.GetErrorString;
}

int need_rccl(const char* who)
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) return mhx_fail(MHX_EHIP, "%s: librccl could not be loaded (set MHX_RCCL_LIB to its path)", who);
    return MHX_OK;
}

}  // namespace

struct mhx_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* d_buf = nullptr;       // staging of the stats all-reduce
    double* h_buf = nullptr;       // ... and its host side (page-locked, owned here: a copy still queued after a missed
    size_t cap = 0;                //     deadline lands in memory that outlives the caller's array)
    bool poisoned = false;         // a collective missed its deadline: its stream still holds it, nothing more is queued behind it
    void* d_stage = nullptr;       // staging of the walker all-gather: [world][stride] bytes
    size_t stage_cap = 0;
    double op_timeout_s = MHX_COMM_INIT_TIMEOUT_S;   // deadline of one blocking collective (mhx_comm_set_timeout)
};

#define RCCL_TRY(expr)                                                                                            \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return mhx_fail(MHX_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));      \
    } while (0)
#define HIP_TRY(expr)                                                                                             \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return mhx_fail(MHX_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));           \
    } while (0)

extern "C" int mhx_comm_unique_id(void* id128)
{
    if (!id128) return mhx_fail(MHX_EINVAL, "mhx_comm_unique_id: id is NULL");
    int rc = need_rccl("mhx_comm_unique_id");
    if (rc) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof id == MHX_COMM_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof id);
    return MHX_OK;
}

// ncclCommInitRank blocks until EVERY rank of `world` has called it: a rank that died, a wrong id or a broken fabric is a hang,
// not an error.  The call therefore runs on a helper thread and the caller waits for it with a deadline; on expiry the helper
// (and whatever RCCL holds) is abandoned -- the process can still report, fall back to another transport, and exit.
namespace {
struct init_job {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    bool abandoned = false;             // the caller's deadline passed: a communicator that still arrives is the helper's to release
    ncclResult_t res = ncclSuccess;
    hipError_t dev = hipSuccess;
    ncclComm_t comm = nullptr;
};
}  // namespace

extern "C" int mhx_comm_init_timed(mhx_ctx* ctx, int rank, int world, const void* id128, double timeout_s, mhx_comm** out)
{
    if (!ctx || !out || !id128) return mhx_fail(MHX_EINVAL, "mhx_comm_init: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return mhx_fail(MHX_EINVAL, "mhx_comm_init: rank %d of %d", rank, world);
    if (!(timeout_s > 0)) timeout_s = MHX_COMM_INIT_TIMEOUT_S;
    int rc = need_rccl("mhx_comm_init");
    if (rc) return rc;
    int device = 0;
    if ((rc = mhx_ctx_device(ctx, &device))) return rc;
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mhx_comm> c(new mhx_comm);
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
#ifdef MHX_TOOLS_BUILD
    // fault injection of the tools build (libmhx_tools.so; tests of the transport ladder): "fail" = an error, "hang" = never returns
    const char* fault = getenv("MHX_FAULT_RCCL_INIT");
    if (fault && !strcmp(fault, "fail")) return mhx_fail(MHX_EHIP, "mhx_comm_init: rank %d of %d: injected failure (MHX_FAULT_RCCL_INIT)", rank, world);
#endif
    auto job = std::make_shared<init_job>();
    std::thread([job, id, world, rank, device
#ifdef MHX_TOOLS_BUILD
                 , fault
#endif
    ] {
        ncclComm_t comm = nullptr;
        ncclResult_t r = ncclSuccess;
        const hipError_t e = hipSetDevice(device);
#ifdef MHX_TOOLS_BUILD
        if (fault && !strcmp(fault, "hang")) for (;;) std::this_thread::sleep_for(std::chrono::seconds(3600));
#endif
        if (e == hipSuccess) r = g_rccl.CommInitRank(&comm, world, id, rank);
        std::unique_lock<std::mutex> lk(job->mu);
        if (job->abandoned) {                            // nobody is waiting any more: do not leak what arrived late
            lk.unlock();
            if (comm && r == ncclSuccess) (void)(g_rccl.CommAbort ? g_rccl.CommAbort(comm) : g_rccl.CommDestroy(comm));
            return;
        }
        job->dev = e; job->res = r; job->comm = comm; job->done = true;
        job->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(job->mu);
        if (!job->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return job->done; })) {
            job->abandoned = true;
            return mhx_fail(MHX_EHIP, "mhx_comm_init: rank %d of %d (device %d): ncclCommInitRank did not return within %.0f s "
                            "(a rank that never arrived, a stale id, or a fabric that cannot connect the ranks)", rank, world, device, timeout_s);
        }
    }
    if (job->dev != hipSuccess) return mhx_fail(MHX_EHIP, "mhx_comm_init: rank %d: hipSetDevice(%d): %s", rank, device, hipGetErrorString(job->dev));
    if (job->res != ncclSuccess)
        return mhx_fail(MHX_EHIP, "mhx_comm_init: rank %d of %d (device %d): ncclCommInitRank failed: %s", rank, world, device, g_rccl.GetErrorString(job->res));
    c->comm = job->comm;
    const hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (es != hipSuccess) {                             // the unique_ptr frees the struct only: release the communicator too
        (void)g_rccl.CommDestroy(c->comm);
        return mhx_fail(MHX_EHIP, "mhx_comm_init: hipStreamCreateWithFlags: %s", hipGetErrorString(es));
    }
    *out = c.release();
    return MHX_OK;
}

extern "C" int mhx_comm_init(mhx_ctx* ctx, int rank, int world, const void* id128, mhx_comm** out)
{
    return mhx_comm_init_timed(ctx, rank, world, id128, MHX_COMM_INIT_TIMEOUT_S, out);
}

extern "C" int mhx_comm_destroy(mhx_comm* c)
{
    if (!c) return MHX_OK;
    (void)hipSetDevice(c->device);
    if (c->poisoned) {
        // a collective no peer joined is still on the stream: destroying the communicator or freeing what the queued copies
        // touch would block or fault.  Abort what RCCL holds if it can, leave the small staging buffers to the process.
        if (c->comm && g_rccl.ok && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
        delete c;
        return MHX_OK;
    }
    if (c->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_buf) (void)hipHostFree(c->h_buf);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MHX_OK;
}

extern "C" int mhx_comm_set_timeout(mhx_comm* c, double seconds)
{
    if (!c || !(seconds > 0)) return mhx_fail(MHX_EINVAL, "mhx_comm_set_timeout: comm is NULL or seconds <= 0");
    c->op_timeout_s = seconds;
    return MHX_OK;
}

extern "C" int mhx_comm_rank(const mhx_comm* c, int* rank, int* world)
{
    if (!c) return mhx_fail(MHX_EINVAL, "mhx_comm_rank: comm is NULL");
    int r = c->rank, w = c->world;
    // what RCCL itself says about this communicator (a launcher's claim is not evidence): ncclCommUserRank / ncclCommCount
    if (c->comm && g_rccl.CommCount && g_rccl.CommUserRank) {
        RCCL_TRY(g_rccl.CommCount(c->comm, &w));
        RCCL_TRY(g_rccl.CommUserRank(c->comm, &r));
    }
    if (rank) *rank = r;
    if (world) *world = w;
    return MHX_OK;
}

// sum over the ranks of n doubles given and returned in host memory: the acceptance totals and the R-hat / ESS sums of
// mhx_run_diagnostics (latency-bound: 24 KB at dim = 1000)
extern "C" int mhx_comm_allreduce_sum(mhx_comm* c, double* inout, size_t n)
{
    if (!c || (!inout && n)) return mhx_fail(MHX_EINVAL, "mhx_comm_allreduce_sum: NULL argument");
    if (!n) return MHX_OK;
    if (c->poisoned)
        return mhx_fail(MHX_ESTATE, "mhx_comm_allreduce_sum: rank %d of %d: an earlier collective on this communicator missed its deadline; "
                        "destroy it and build a new one", c->rank, c->world);
    HIP_TRY(hipSetDevice(c->device));
    if (c->cap < n) {
        if (c->d_buf) (void)hipFree(c->d_buf);
        if (c->h_buf) (void)hipHostFree(c->h_buf);
        c->d_buf = nullptr; c->h_buf = nullptr; c->cap = 0;
        HIP_TRY(hipMalloc(&c->d_buf, n * sizeof(double)));
        HIP_TRY(hipHostMalloc((void**)&c->h_buf, n * sizeof(double), hipHostMallocDefault));
        c->cap = n;
    }
    memcpy(c->h_buf, inout, n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(c->d_buf, c->h_buf, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(g_rccl.AllReduce(c->d_buf, c->d_buf, n, ncclDouble, ncclSum, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_buf, c->d_buf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    // a collective a peer never joins does not fail, it waits: poll with a deadline instead of hipStreamSynchronize
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return mhx_fail(MHX_EHIP, "mhx_comm_allreduce_sum: %s", hipGetErrorString(q));
        if (spins > 2000) {
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > c->op_timeout_s) {
                c->poisoned = true;                       // the caller's array is untouched; the queued copy writes into h_buf
                return mhx_fail(MHX_EHIP, "mhx_comm_allreduce_sum: rank %d of %d: the all-reduce of %zu doubles did not complete within %.0f s "
                                "(a peer never joined it)", c->rank, c->world, n, c->op_timeout_s);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    memcpy(inout, c->h_buf, n * sizeof(double));
    return MHX_OK;
}

// ONE ensemble sharded over the ranks: after mhx_emcee_half_step moved this rank's slice of half `half`, exchange the
// slices.  The slice of rank r is walkers [cnt r / world, cnt (r+1) / world) of the moving half (cnt = its size):
// mhx_comm_slice below.  The rows, log-densities and accept bookkeeping of the slice are packed into one staging buffer
// (fixed stride per rank), all-gathered with ONE collective on the run's stream and unpacked -- no host synchronisation.
extern "C" int mhx_comm_slice(const mhx_comm* c, int cnt, int* begin, int* count)
{
    if (!c) return mhx_fail(MHX_EINVAL, "mhx_comm_slice: comm is NULL");
    const long b = (long)cnt * c->rank / c->world, e = (long)cnt * (c->rank + 1) / c->world;
    if (begin) *begin = (int)b;
    if (count) *count = (int)(e - b);
    return MHX_OK;
}

extern "C" int mhx_comm_allgather_walkers(mhx_comm* c, mhx_run* run, int half)
{
    if (!c || !run) return mhx_fail(MHX_EINVAL, "mhx_comm_allgather_walkers: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    size_t stride = 0;
    void* stream = nullptr;
    int rc = mhx_emcee_exchange_plan(run, half, c->world, &stride, &stream);
    if (rc) return rc;
    const size_t need = stride * (size_t)c->world;
    if (c->stage_cap < need) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr; c->stage_cap = 0;
        HIP_TRY(hipMalloc(&c->d_stage, need));
        c->stage_cap = need;
    }
    if ((rc = mhx_emcee_exchange_pack(run, half, c->rank, c->world, (char*)c->d_stage + stride * (size_t)c->rank))) return rc;
    if (c->world > 1)
        RCCL_TRY(g_rccl.AllGather((char*)c->d_stage + stride * (size_t)c->rank, c->d_stage, stride, ncclChar, c->comm, (hipStream_t)stream));
    return mhx_emcee_exchange_unpack(run, half, c->rank, c->world, c->d_stage, stride);
}
