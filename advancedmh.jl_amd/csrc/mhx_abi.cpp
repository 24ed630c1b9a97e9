// mhx_abi.cpp -- the C ABI of include/mhx.h: every entry point forwards to the instantiation of the engine the handle
// belongs to (namespace mhx_f32: mhx_real = float, namespace mhx_f64: mhx_real = double; mhx_impl.h).  The first word
// of every handle is its mhx_dtype.  No device code here.
#include "mhx_impl.h"
#include "mhx_host_expand.h"    // mhx_numa_*: placement of page-locked result memory

#include <hip/hip_runtime_api.h>   // hipHostMalloc / hipHostFree only (mhx_host_alloc): no device code here

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>

static thread_local std::string g_err;
static std::mutex g_jit_mu;
void mhx_jit_lock() { g_jit_mu.lock(); }
void mhx_jit_unlock() { g_jit_mu.unlock(); }

int mhx_fail(int code, const char* fmt, ...)
{
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" int mhx_version(void) { return MHX_VERSION; }
extern "C" const char* mhx_last_error(void) { return g_err.c_str(); }

static inline bool is64(const void* h) { return reinterpret_cast<const mhx_handle_hdr*>(h)->dtype == MHX_F64; }
// same dtype for every handle of a call (a target built on an fp32 context cannot serve an fp64 run)
static inline bool same(const void* a, const void* b) { return !a || !b || is64(a) == is64(b); }

#define C32(p) reinterpret_cast<mhx_f32::mhx_ctx*>(p)
#define C64(p) reinterpret_cast<mhx_f64::mhx_ctx*>(p)
#define T32(p) reinterpret_cast<mhx_f32::mhx_target*>(p)
#define T64(p) reinterpret_cast<mhx_f64::mhx_target*>(p)
#define CT32(p) reinterpret_cast<const mhx_f32::mhx_target*>(p)
#define CT64(p) reinterpret_cast<const mhx_f64::mhx_target*>(p)
#define R32(p) reinterpret_cast<mhx_f32::mhx_run*>(p)
#define R64(p) reinterpret_cast<mhx_f64::mhx_run*>(p)
#define F(p) static_cast<float*>(p)
#define D(p) static_cast<double*>(p)
#define CF(p) static_cast<const float*>(p)
#define CD(p) static_cast<const double*>(p)

#define NEED(h, name)                                                                  \
    do {                                                                               \
        if (!(h)) return mhx_fail(MHX_EINVAL, name ": handle is NULL");                \
    } while (0)
#define MIXED(name) mhx_fail(MHX_EINVAL, name ": the handles belong to contexts of different dtype")

extern "C" int mhx_ctx_create(int device, int dtype, mhx_ctx** out)
{
    if (!out) return mhx_fail(MHX_EINVAL, "mhx_ctx_create: out is NULL");
    if (dtype == MHX_F64) return mhx_f64::api_ctx_create(device, reinterpret_cast<mhx_f64::mhx_ctx**>(out));
    if (dtype == MHX_F32) return mhx_f32::api_ctx_create(device, reinterpret_cast<mhx_f32::mhx_ctx**>(out));
    return mhx_fail(MHX_EINVAL, "mhx_ctx_create: dtype must be MHX_F32 (0) or MHX_F64 (1), got %d", dtype);
}
extern "C" int mhx_ctx_dtype(const mhx_ctx* ctx) { return ctx ? (is64(ctx) ? MHX_F64 : MHX_F32) : MHX_EINVAL; }
extern "C" int mhx_ctx_device(const mhx_ctx* ctx, int* device)
{
    if (!ctx || !device) return mhx_fail(MHX_EINVAL, "mhx_ctx_device: NULL argument");
    *device = is64(ctx) ? mhx_f64::api_ctx_device(reinterpret_cast<const mhx_f64::mhx_ctx*>(ctx)) : mhx_f32::api_ctx_device(reinterpret_cast<const mhx_f32::mhx_ctx*>(ctx));
    return MHX_OK;
}
extern "C" int mhx_ctx_set_option(mhx_ctx* ctx, const char* name, const char* value)
{
    NEED(ctx, "mhx_ctx_set_option");
    return is64(ctx) ? mhx_f64::api_ctx_set_option(C64(ctx), name, value) : mhx_f32::api_ctx_set_option(C32(ctx), name, value);
}
extern "C" int mhx_ctx_get_option(const mhx_ctx* ctx, const char* name, char* buf, size_t len)
{
    NEED(ctx, "mhx_ctx_get_option");
    return is64(ctx) ? mhx_f64::api_ctx_get_option(reinterpret_cast<const mhx_f64::mhx_ctx*>(ctx), name, buf, len)
                     : mhx_f32::api_ctx_get_option(reinterpret_cast<const mhx_f32::mhx_ctx*>(ctx), name, buf, len);
}
extern "C" int mhx_ctx_pci_bus_id(const mhx_ctx* ctx, char* buf, size_t len)
{
    NEED(ctx, "mhx_ctx_pci_bus_id");
    return is64(ctx) ? mhx_f64::api_ctx_pci_bus_id(reinterpret_cast<const mhx_f64::mhx_ctx*>(ctx), buf, len)
                     : mhx_f32::api_ctx_pci_bus_id(reinterpret_cast<const mhx_f32::mhx_ctx*>(ctx), buf, len);
}
extern "C" int mhx_ctx_jit_counts(const mhx_ctx* ctx, int64_t* compiles, int64_t* cache_hits)
{
    if (!ctx) return mhx_fail(MHX_EINVAL, "mhx_ctx_jit_counts: ctx is NULL");
    long a = 0, b = 0;
    if (is64(ctx)) mhx_f64::api_ctx_jit_counts(reinterpret_cast<const mhx_f64::mhx_ctx*>(ctx), &a, &b);
    else mhx_f32::api_ctx_jit_counts(reinterpret_cast<const mhx_f32::mhx_ctx*>(ctx), &a, &b);
    if (compiles) *compiles = a;
    if (cache_hits) *cache_hits = b;
    return MHX_OK;
}
extern "C" int mhx_ctx_jit_compiler(const mhx_ctx* ctx, char* compiler, size_t len, int64_t* ext_compiles)
{
    if (!ctx) return mhx_fail(MHX_EINVAL, "mhx_ctx_jit_compiler: ctx is NULL");
    long n = 0;
    return is64(ctx) ? mhx_f64::api_ctx_jit_compiler(reinterpret_cast<const mhx_f64::mhx_ctx*>(ctx), compiler, len, ext_compiles ? &n : nullptr) ||
                           (ext_compiles ? (*ext_compiles = n, 0) : 0)
                     : mhx_f32::api_ctx_jit_compiler(reinterpret_cast<const mhx_f32::mhx_ctx*>(ctx), compiler, len, ext_compiles ? &n : nullptr) ||
                           (ext_compiles ? (*ext_compiles = n, 0) : 0);
}
extern "C" int mhx_ctx_host_pin_counts(const mhx_ctx* ctx, int64_t* registered, int64_t* released)
{
    if (!ctx) return mhx_fail(MHX_EINVAL, "mhx_ctx_host_pin_counts: ctx is NULL");
    long a = 0, b = 0;
    if (is64(ctx)) mhx_f64::api_ctx_host_pin_counts(reinterpret_cast<const mhx_f64::mhx_ctx*>(ctx), &a, &b);
    else mhx_f32::api_ctx_host_pin_counts(reinterpret_cast<const mhx_f32::mhx_ctx*>(ctx), &a, &b);
    if (registered) *registered = a;
    if (released) *released = b;
    return MHX_OK;
}
extern "C" int mhx_host_alloc(size_t bytes, void** out)
{
    if (!out) return mhx_fail(MHX_EINVAL, "mhx_host_alloc: out is NULL");
    *out = nullptr;
    if (!bytes) return MHX_OK;
    // on the memory node of the calling thread's current device: what fills a result tensor -- the GPU's DMA engine, or the host
    // threads that expand accept-compacted blocks, which keep to that node -- then works on local memory
    int dev = 0, node = -1;
    char bus[32] = {0};
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) == hipSuccess) node = mhx_numa_node_of_pci(bus);
    else (void)hipGetLastError();
    if (node >= 0) mhx_numa_prefer(node);
    hipError_t e = hipHostMalloc(out, bytes, node >= 0 ? hipHostMallocNumaUser : hipHostMallocDefault);
    if (node >= 0) mhx_numa_default();
    if (e != hipSuccess) {
        *out = nullptr;
        return mhx_fail(MHX_ENOMEM, "mhx_host_alloc: %zu page-locked bytes: %s", bytes, hipGetErrorString(e));
    }
    return MHX_OK;
}
extern "C" int mhx_host_free(void* p)
{
    if (!p) return MHX_OK;
    hipError_t e = hipHostFree(p);
    return e == hipSuccess ? MHX_OK : mhx_fail(MHX_EHIP, "mhx_host_free: %s", hipGetErrorString(e));
}
extern "C" int mhx_ctx_destroy(mhx_ctx* ctx)
{
    if (!ctx) return MHX_OK;
    return is64(ctx) ? mhx_f64::api_ctx_destroy(C64(ctx)) : mhx_f32::api_ctx_destroy(C32(ctx));
}

extern "C" int mhx_target_builtin(mhx_ctx* ctx, int kind, int dim, const void* params, size_t nparams, mhx_target** out)
{
    NEED(ctx, "mhx_target_builtin");
    return is64(ctx) ? mhx_f64::api_target_builtin(C64(ctx), kind, dim, CD(params), nparams, reinterpret_cast<mhx_f64::mhx_target**>(out))
                     : mhx_f32::api_target_builtin(C32(ctx), kind, dim, CF(params), nparams, reinterpret_cast<mhx_f32::mhx_target**>(out));
}
extern "C" int mhx_target_from_hip_source(mhx_ctx* ctx, const char* src, int dim, const void* data, size_t ndata, mhx_target** out)
{
    NEED(ctx, "mhx_target_from_hip_source");
    return is64(ctx) ? mhx_f64::api_target_from_hip_source(C64(ctx), src, dim, CD(data), ndata, reinterpret_cast<mhx_f64::mhx_target**>(out))
                     : mhx_f32::api_target_from_hip_source(C32(ctx), src, dim, CF(data), ndata, reinterpret_cast<mhx_f32::mhx_target**>(out));
}
extern "C" int mhx_target_destroy(mhx_target* t)
{
    if (!t) return MHX_OK;
    return is64(t) ? mhx_f64::api_target_destroy(T64(t)) : mhx_f32::api_target_destroy(T32(t));
}
extern "C" int mhx_target_eval(mhx_ctx* ctx, const mhx_target* t, const void* x, int n, void* lp)
{
    NEED(ctx, "mhx_target_eval");
    if (!same(ctx, t)) return MIXED("mhx_target_eval");
    return is64(ctx) ? mhx_f64::api_target_eval(C64(ctx), CT64(t), CD(x), n, D(lp)) : mhx_f32::api_target_eval(C32(ctx), CT32(t), CF(x), n, F(lp));
}

#define CREATE(NAME, CFG)                                                                                              \
    extern "C" int mhx_##NAME##_create(mhx_ctx* ctx, const mhx_target* t, const CFG* cfg, mhx_run** out)               \
    {                                                                                                                  \
        NEED(ctx, "mhx_" #NAME "_create");                                                                             \
        if (!same(ctx, t)) return MIXED("mhx_" #NAME "_create");                                                       \
        return is64(ctx) ? mhx_f64::api_##NAME##_create(C64(ctx), CT64(t), cfg, reinterpret_cast<mhx_f64::mhx_run**>(out)) \
                         : mhx_f32::api_##NAME##_create(C32(ctx), CT32(t), cfg, reinterpret_cast<mhx_f32::mhx_run**>(out)); \
    }
CREATE(rwmh, mhx_rwmh_cfg)
CREATE(emcee, mhx_emcee_cfg)
CREATE(ram, mhx_ram_cfg)
CREATE(mala, mhx_mala_cfg)

extern "C" int mhx_ram_set_factor(mhx_run* r, const void* S)
{
    NEED(r, "mhx_ram_set_factor");
    return is64(r) ? mhx_f64::api_ram_set_factor(R64(r), CD(S)) : mhx_f32::api_ram_set_factor(R32(r), CF(S));
}
extern "C" int mhx_ram_set_factor_all(mhx_run* r, const void* S)
{
    NEED(r, "mhx_ram_set_factor_all");
    return is64(r) ? mhx_f64::api_ram_set_factor_all(R64(r), CD(S)) : mhx_f32::api_ram_set_factor_all(R32(r), CF(S));
}
extern "C" int mhx_ram_get_factor(mhx_run* r, void* S, uint8_t* status)
{
    NEED(r, "mhx_ram_get_factor");
    return is64(r) ? mhx_f64::api_ram_get_factor(R64(r), D(S), status) : mhx_f32::api_ram_get_factor(R32(r), F(S), status);
}
extern "C" int mhx_ram_get_diag_range(mhx_run* r, void* diag_min, void* diag_max)
{
    NEED(r, "mhx_ram_get_diag_range");
    return is64(r) ? mhx_f64::api_ram_get_diag_range(R64(r), D(diag_min), D(diag_max)) : mhx_f32::api_ram_get_diag_range(R32(r), F(diag_min), F(diag_max));
}
extern "C" int mhx_ram_watch_factors(mhx_run* r, const int32_t* chains, int32_t n)
{
    NEED(r, "mhx_ram_watch_factors");
    return is64(r) ? mhx_f64::api_ram_watch_factors(R64(r), chains, n) : mhx_f32::api_ram_watch_factors(R32(r), chains, n);
}
extern "C" int mhx_ram_get_watched_factors(mhx_run* r, void* S, int64_t capacity, int64_t* n_recorded, int32_t* n_watched)
{
    NEED(r, "mhx_ram_get_watched_factors");
    long nrec = 0; int nw = 0;
    const int rc = is64(r) ? mhx_f64::api_ram_get_watched_factors(R64(r), D(S), (long)capacity, &nrec, &nw)
                           : mhx_f32::api_ram_get_watched_factors(R32(r), F(S), (long)capacity, &nrec, &nw);
    if (n_recorded) *n_recorded = nrec;
    if (n_watched) *n_watched = nw;
    return rc;
}
extern "C" int mhx_ram_get_step_stats(mhx_run* r, void* log_alpha, double* eta, int64_t capacity, int64_t* n_recorded)
{
    NEED(r, "mhx_ram_get_step_stats");
    long nrec = 0;
    const int rc = is64(r) ? mhx_f64::api_ram_get_step_stats(R64(r), D(log_alpha), eta, (long)capacity, &nrec)
                           : mhx_f32::api_ram_get_step_stats(R32(r), F(log_alpha), eta, (long)capacity, &nrec);
    if (n_recorded) *n_recorded = nrec;
    return rc;
}
extern "C" int mhx_ram_get_adapt_state(mhx_run* r, void* log_alpha, double* eta, uint8_t* isaccept, uint64_t* iteration)
{
    NEED(r, "mhx_ram_get_adapt_state");
    return is64(r) ? mhx_f64::api_ram_get_adapt_state(R64(r), D(log_alpha), eta, isaccept, iteration)
                   : mhx_f32::api_ram_get_adapt_state(R32(r), F(log_alpha), eta, isaccept, iteration);
}

extern "C" int mhx_run_init(mhx_run* r, const void* initial_params)
{
    NEED(r, "mhx_run_init");
    return is64(r) ? mhx_f64::api_run_init(R64(r), CD(initial_params)) : mhx_f32::api_run_init(R32(r), CF(initial_params));
}
extern "C" int mhx_run_sample(mhx_run* r, const mhx_schedule* s, int save_samples)
{
    NEED(r, "mhx_run_sample");
    return is64(r) ? mhx_f64::api_run_sample(R64(r), s, save_samples) : mhx_f32::api_run_sample(R32(r), s, save_samples);
}
extern "C" int mhx_run_get_samples(mhx_run* r, void* samples, uint8_t* accepted)
{
    NEED(r, "mhx_run_get_samples");
    return is64(r) ? mhx_f64::api_run_get_samples(R64(r), D(samples), accepted) : mhx_f32::api_run_get_samples(R32(r), F(samples), accepted);
}
extern "C" int mhx_run_host_stats(mhx_run* r, mhx_host_stats* out)
{
    NEED(r, "mhx_run_host_stats");
    return is64(r) ? mhx_f64::api_run_host_stats(R64(r), out) : mhx_f32::api_run_host_stats(R32(r), out);
}
extern "C" int mhx_run_sample_to_host(mhx_run* r, const mhx_schedule* s, void* samples, uint8_t* accepted, int32_t slab_samples)
{
    NEED(r, "mhx_run_sample_to_host");
    return is64(r) ? mhx_f64::api_run_sample_to_host(R64(r), s, D(samples), accepted, slab_samples)
                   : mhx_f32::api_run_sample_to_host(R32(r), s, F(samples), accepted, slab_samples);
}
extern "C" int mhx_run_device_samples(mhx_run* r, void** samples, void** accepted, int64_t* n_samples)
{
    NEED(r, "mhx_run_device_samples");
    return is64(r) ? mhx_f64::api_run_device_samples(R64(r), samples, accepted, n_samples) : mhx_f32::api_run_device_samples(R32(r), samples, accepted, n_samples);
}
extern "C" int mhx_run_get_state(mhx_run* r, void* x, void* lp, uint32_t* accept_counts)
{
    NEED(r, "mhx_run_get_state");
    return is64(r) ? mhx_f64::api_run_get_state(R64(r), D(x), D(lp), accept_counts) : mhx_f32::api_run_get_state(R32(r), F(x), F(lp), accept_counts);
}
extern "C" int mhx_run_set_state(mhx_run* r, const void* x)
{
    NEED(r, "mhx_run_set_state");
    return is64(r) ? mhx_f64::api_run_set_state(R64(r), CD(x)) : mhx_f32::api_run_set_state(R32(r), CF(x));
}
extern "C" int mhx_run_state_size(mhx_run* r, size_t* bytes)
{
    NEED(r, "mhx_run_state_size");
    return is64(r) ? mhx_f64::api_run_state_size(R64(r), bytes) : mhx_f32::api_run_state_size(R32(r), bytes);
}
extern "C" int mhx_run_save_state(mhx_run* r, void* blob, size_t bytes)
{
    NEED(r, "mhx_run_save_state");
    return is64(r) ? mhx_f64::api_run_save_state(R64(r), blob, bytes) : mhx_f32::api_run_save_state(R32(r), blob, bytes);
}
extern "C" int mhx_run_load_state(mhx_run* r, const void* blob, size_t bytes)
{
    NEED(r, "mhx_run_load_state");
    return is64(r) ? mhx_f64::api_run_load_state(R64(r), blob, bytes) : mhx_f32::api_run_load_state(R32(r), blob, bytes);
}
extern "C" int mhx_run_stats(mhx_run* r, mhx_stats* out)
{
    NEED(r, "mhx_run_stats");
    return is64(r) ? mhx_f64::api_run_stats(R64(r), out) : mhx_f32::api_run_stats(R32(r), out);
}
extern "C" int mhx_run_shape(const mhx_run* r, int32_t* dim, int32_t* nchains)
{
    NEED(r, "mhx_run_shape");
    return is64(r) ? mhx_f64::api_run_shape(reinterpret_cast<const mhx_f64::mhx_run*>(r), dim, nchains)
                   : mhx_f32::api_run_shape(reinterpret_cast<const mhx_f32::mhx_run*>(r), dim, nchains);
}
extern "C" int mhx_run_destroy(mhx_run* r)
{
    if (!r) return MHX_OK;
    return is64(r) ? mhx_f64::api_run_destroy(R64(r)) : mhx_f32::api_run_destroy(R32(r));
}
extern "C" int mhx_run_diagnostics(mhx_run* r, const mhx_diag_cfg* cfg, double* sum_m, double* sum_m2, double* sum_v, double* ess)
{
    NEED(r, "mhx_run_diagnostics");
    return is64(r) ? mhx_f64::api_run_diagnostics(R64(r), cfg, sum_m, sum_m2, sum_v, ess) : mhx_f32::api_run_diagnostics(R32(r), cfg, sum_m, sum_m2, sum_v, ess);
}
extern "C" int mhx_run_ess_bulk_tail(mhx_run* r, const mhx_diag_cfg* cfg, const int32_t* params, int32_t nparams, double* ess_bulk, double* ess_tail)
{
    NEED(r, "mhx_run_ess_bulk_tail");
    return is64(r) ? mhx_f64::api_run_ess_bulk_tail(R64(r), cfg, params, nparams, ess_bulk, ess_tail)
                   : mhx_f32::api_run_ess_bulk_tail(R32(r), cfg, params, nparams, ess_bulk, ess_tail);
}
extern "C" int mhx_emcee_half_step(mhx_run* r, int half, int begin, int count)
{
    NEED(r, "mhx_emcee_half_step");
    return is64(r) ? mhx_f64::api_emcee_half_step(R64(r), half, begin, count) : mhx_f32::api_emcee_half_step(R32(r), half, begin, count);
}
extern "C" int mhx_emcee_end_sweep(mhx_run* r)
{
    NEED(r, "mhx_emcee_end_sweep");
    return is64(r) ? mhx_f64::api_emcee_end_sweep(R64(r)) : mhx_f32::api_emcee_end_sweep(R32(r));
}
extern "C" int mhx_emcee_device_state(mhx_run* r, void** xw, int32_t* pitch, void** lp, uint32_t** acc_count, uint8_t** last_acc)
{
    NEED(r, "mhx_emcee_device_state");
    return is64(r) ? mhx_f64::api_emcee_device_state(R64(r), reinterpret_cast<double**>(xw), pitch, reinterpret_cast<double**>(lp), acc_count, last_acc)
                   : mhx_f32::api_emcee_device_state(R32(r), reinterpret_cast<float**>(xw), pitch, reinterpret_cast<float**>(lp), acc_count, last_acc);
}
extern "C" int mhx_emcee_exchange_plan(mhx_run* r, int half, int world, size_t* stride, void** stream)
{
    NEED(r, "mhx_emcee_exchange_plan");
    return is64(r) ? mhx_f64::api_emcee_exchange_plan(R64(r), half, world, stride, stream) : mhx_f32::api_emcee_exchange_plan(R32(r), half, world, stride, stream);
}
extern "C" int mhx_emcee_exchange_pack(mhx_run* r, int half, int rank, int world, void* part)
{
    NEED(r, "mhx_emcee_exchange_pack");
    return is64(r) ? mhx_f64::api_emcee_exchange_pack(R64(r), half, rank, world, part) : mhx_f32::api_emcee_exchange_pack(R32(r), half, rank, world, part);
}
extern "C" int mhx_emcee_exchange_unpack(mhx_run* r, int half, int rank, int world, const void* stage, size_t stride)
{
    NEED(r, "mhx_emcee_exchange_unpack");
    return is64(r) ? mhx_f64::api_emcee_exchange_unpack(R64(r), half, rank, world, stage, stride)
                   : mhx_f32::api_emcee_exchange_unpack(R32(r), half, rank, world, stage, stride);
}
