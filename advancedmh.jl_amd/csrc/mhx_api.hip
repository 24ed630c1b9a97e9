// mhx_api.hip -- host side of libmhx.so: the C ABI of include/mhx.h, device memory ownership,
// kernel selection (pre-built register kernel / hiprtc-specialised kernel / generic kernel) and
// the mcmcsample-equivalent launch schedule.  No torch types, no C++ types across the boundary.
#include "../../include/mhx.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <errno.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <rocprim/rocprim.hpp>   // device radix sort for the rank-normalised ESS (after <cstring>: it calls memset)

#include "mhx_rwmh_kernels.h"
#include "mhx_emcee_kernels.h"
#include "mhx_ram_kernels.h"
#include "mhx_mala_kernels.h"
#include "mhx_rwmh_dense_kernels.h"
#include "mhx_rwmh_mfma_kernels.h"
#include "mhx_mala_mfma_kernels.h"
#include "mhx_diag_kernels.h"
#ifdef MHX_TOOLS_BUILD
#include "mhx_jit_embed_tools.inc"   // generated: the device headers as string literals for hiprtc, probes included
#else
#include "mhx_jit_embed.inc"         // generated (embed_headers.py --release): the same text without its MHX_TOOLS_BUILD blocks
#endif
#include "mhx_jit_ext.h"        // run-time kernels through the installation's clang++ (pure host code)
#include "mhx_host_expand.h"    // the host threads of the accept-compacted return path (pure host code, shared by both instantiations)
#include "mhx_impl.h"           // the prototypes of this instantiation (api_*), shared with the dispatcher mhx_abi.cpp

#define HIP_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return mhx_fail(MHX_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                        __LINE__);                                                                  \
    } while (0)

// Every copy that touches run / target state goes through the context's stream (created non-blocking: the null stream
// does not order against it) and is waited for, so that all accesses to device state are ordered on ONE stream.
#define COPY_SYNC(stream, dst, src, bytes, kind)                                                    \
    do {                                                                                            \
        HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), (kind), (stream)));                           \
        HIP_TRY(hipStreamSynchronize(stream));                                                      \
    } while (0)

namespace MHX_NS {

// ---------------------------------------------------------------------------------------------
// pre-built kernels (hipcc, gfx950)
template <int D, int TK, int PK>
__global__ void __launch_bounds__(64)
k_rwmh_reg(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    mhx_rwmh_reg_body<D, TK, PK>(a, tparams, pvec);
}
template <int PK, int K>
__global__ void __launch_bounds__(64)
k_rwmh_wave(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    mhx_rwmh_wave_body<PK, K>(a, tparams, pvec);
}
__global__ void __launch_bounds__(256)
k_rwmh_generic(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    mhx_rwmh_generic_body<MHX_TARGET_DYNAMIC>(a, tparams, pvec);
}
__global__ void __launch_bounds__(256)
k_rwmh_init(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec, const int draw)
{
    mhx_rwmh_init_body<MHX_TARGET_DYNAMIC>(a, tparams, pvec, draw);
}
__global__ void __launch_bounds__(256)
k_target_eval(const mhx_real* __restrict__ x, mhx_real* __restrict__ lp, const int n, const int d, const int kind,
              const mhx_real* __restrict__ tparams, const int ntparams, const mhx_real tconst, const int lanes)
{
    mhx_target_eval_body<MHX_TARGET_DYNAMIC>(x, lp, n, d, kind, tparams, ntparams, tconst, lanes);
}
__global__ void __launch_bounds__(256)
k_record_state(const mhx_real* __restrict__ x, const mhx_real* __restrict__ lp, const unsigned char* __restrict__ last_acc,
               mhx_real* samples, unsigned char* accepted, const int n, const long ld, const int d, const long slot)
{
    mhx_record_state_body(x, lp, last_acc, samples, accepted, n, ld, d, slot);
}

// (MHX_COOP_WAVES, the occupancy target of the cooperative kernel, comes from mhx_rwmh_kernels.h)
template <int L, int NBL, int TK, int PK, bool MOM, int GEN = MHX_GEN_BOX_MULLER>
__global__ void __launch_bounds__(256, MHX_COOP_WAVES(NBL))
k_rwmh_coop(const mhx_rwmh_args a, const mhx_real* __restrict__ tparams, const mhx_real* __restrict__ pvec)
{
    mhx_rwmh_coop_body<L, NBL, TK, PK, MOM, MHX_WALK_PLAIN, GEN>(a, tparams, pvec);
}
__global__ void __launch_bounds__(256)
k_moments_first(const mhx_real* __restrict__ x, const mhx_real* __restrict__ lp, mhx_real* mean, mhx_real* m2, const int n,
                const long ld, const int d)
{
    mhx_moments_first_body(x, lp, mean, m2, n, ld, d);
}

// `fn` records samples, `fn_mom` keeps running moments instead (null: specialised by hiprtc on demand)
struct prebuilt_coop {
    int L, NBL, TK, PK;
    void (*fn)(const mhx_rwmh_args, const mhx_real*, const mhx_real*);
    void (*fn_mom)(const mhx_rwmh_args, const mhx_real*, const mhx_real*);
    int gen;                             // MHX_GEN_*: the normal generator the kernel was built with
};
// the BASELINE shapes: C2 (65 536 chains x d = 100) and C5 (32 768 chains per GPU x d = 1000); a double takes two VGPRs,
// so the fp64 engine spreads a chain over more lanes
#if MHX_REAL64
#define MHX_C2_L 2
#define MHX_C2_NBL 13
#define MHX_C5_L 64
#define MHX_C5_NBL 4
#else
#define MHX_C2_L 2
#define MHX_C2_NBL 13
#define MHX_C5_L 32
#define MHX_C5_NBL 8
#endif
static const prebuilt_coop k_prebuilt_coop[] = {
    {MHX_C2_L, MHX_C2_NBL, MHX_TARGET_ISO_GAUSS, MHX_PROP_ISO, k_rwmh_coop<MHX_C2_L, MHX_C2_NBL, MHX_TARGET_ISO_GAUSS, MHX_PROP_ISO, false>, nullptr},
    {MHX_C5_L, MHX_C5_NBL, MHX_TARGET_FUNNEL, MHX_PROP_ISO, k_rwmh_coop<MHX_C5_L, MHX_C5_NBL, MHX_TARGET_FUNNEL, MHX_PROP_ISO, false>,
     k_rwmh_coop<MHX_C5_L, MHX_C5_NBL, MHX_TARGET_FUNNEL, MHX_PROP_ISO, true>},
    {MHX_C5_L, MHX_C5_NBL, MHX_TARGET_BANANA, MHX_PROP_ISO, k_rwmh_coop<MHX_C5_L, MHX_C5_NBL, MHX_TARGET_BANANA, MHX_PROP_ISO, false>,
     k_rwmh_coop<MHX_C5_L, MHX_C5_NBL, MHX_TARGET_BANANA, MHX_PROP_ISO, true>},
    // the same shapes with the ziggurat generator (MHX_FLAG_ZIGGURAT; both widths since round 6)
    {MHX_C2_L, MHX_C2_NBL, MHX_TARGET_ISO_GAUSS, MHX_PROP_ISO,
     k_rwmh_coop<MHX_C2_L, MHX_C2_NBL, MHX_TARGET_ISO_GAUSS, MHX_PROP_ISO, false, MHX_GEN_ZIGGURAT>, nullptr, MHX_GEN_ZIGGURAT},
    {MHX_C5_L, MHX_C5_NBL, MHX_TARGET_FUNNEL, MHX_PROP_ISO,
     k_rwmh_coop<MHX_C5_L, MHX_C5_NBL, MHX_TARGET_FUNNEL, MHX_PROP_ISO, false, MHX_GEN_ZIGGURAT>,
     k_rwmh_coop<MHX_C5_L, MHX_C5_NBL, MHX_TARGET_FUNNEL, MHX_PROP_ISO, true, MHX_GEN_ZIGGURAT>, MHX_GEN_ZIGGURAT},
};

// sum of a u32 array into a u64 (one atomic per block)
__global__ void __launch_bounds__(256)
k_sum_u32(const unsigned* __restrict__ v, const long n, unsigned long long* out)
{
    __shared__ unsigned long long red[4];
    unsigned long long s = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

struct prebuilt_reg { int D, TK, PK; void (*fn)(const mhx_rwmh_args, const mhx_real*, const mhx_real*); };
static const prebuilt_reg k_prebuilt_reg[] = {
#if !MHX_REAL64
    {100, MHX_TARGET_ISO_GAUSS, MHX_PROP_ISO, k_rwmh_reg<100, MHX_TARGET_ISO_GAUSS, MHX_PROP_ISO>},
#endif
    {2, MHX_TARGET_IID_NORMAL, MHX_PROP_ISO, k_rwmh_reg<2, MHX_TARGET_IID_NORMAL, MHX_PROP_ISO>},     // C1, README.md:29-31
};

// ---------------------------------------------------------------------------------------------
// context + JIT cache
struct jit_module {
    hipModule_t mod = nullptr;
    std::map<std::string, hipFunction_t> fns;
};

// one of three buffer sets a compacted slab travels through (mhx_api_host.inc): device block -> page-locked staging -> host threads
struct compact_pair {
    unsigned char* dev_fixed = nullptr;  // header + masks + ranks + accept flags
    size_t dev_fixed_cap = 0;
    unsigned char* dev_pay = nullptr;    // the changed columns
    size_t dev_pay_cap = 0;
    unsigned char* host = nullptr;       // the block as the host threads read it (page-locked)
    size_t host_cap = 0;
    hipEvent_t c0 = nullptr, c1 = nullptr, scattered = nullptr;   // copy begin / end (timed), payload written
    uint64_t seq = 0;                    // the expander job that reads `host`
    bool used = false;
    int device = 0;
    double link_ms = 0.0;                // copy time of this pair's blocks in the current call (written by one expander thread)
};

struct mhx_ctx : mhx_handle_hdr {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, std::unique_ptr<jit_module>> jit;
    long jit_compiles = 0, jit_cache_hits = 0;      // compilations / code objects taken from the on-disk cache
    long jit_ext_compiles = 0;                      // ... of the compilations, by the installation's clang++ (mhx_jit_ext.h)
    // the return path of mhx_run_sample_to_host: a second stream for the D2H copies + the slab hand-over events
    hipStream_t copy_stream = nullptr;
    hipEvent_t slab_done[2] = {nullptr, nullptr}, slab_free[2] = {nullptr, nullptr};
    long pins_registered = 0, pins_released = 0;    // caller buffers page-locked for the duration of a call / released again
    std::map<std::string, std::string> options;     // mhx_ctx_set_option: explicit engine options (never the environment)
    bool tainted = false;                           // a probe / fault-injection option of the tools build was set: results may be invalid
    // the accept-compacted return path: three buffer sets, the pinned landing place of a block's header, the host threads
    compact_pair cpair[3];
    mhx_compact_hdr* hdr_pinned = nullptr;
    mhx_expander* expander = nullptr;
    int expander_threads = 0, expander_chunk = 0, expander_numa = -1;   // the options / memory node the expander was created with
    ~mhx_ctx()
    {
        if (expander) mhx_expander_destroy(expander);
        for (compact_pair& P : cpair) {
            if (P.dev_fixed) (void)hipFree(P.dev_fixed);
            if (P.dev_pay) (void)hipFree(P.dev_pay);
            if (P.host) (void)hipHostFree(P.host);
            if (P.c0) (void)hipEventDestroy(P.c0);
            if (P.c1) (void)hipEventDestroy(P.c1);
            if (P.scattered) (void)hipEventDestroy(P.scattered);
        }
        if (hdr_pinned) (void)hipHostFree(hdr_pinned);
        for (int i = 0; i < 2; ++i) {
            if (slab_done[i]) (void)hipEventDestroy(slab_done[i]);
            if (slab_free[i]) (void)hipEventDestroy(slab_free[i]);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto& kv : jit)
            if (kv.second && kv.second->mod) (void)hipModuleUnload(kv.second->mod);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

int api_ctx_create(int device, mhx_ctx** out)
{
    if (!out) return mhx_fail(MHX_EINVAL, "mhx_ctx_create: out is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return mhx_fail(MHX_EINVAL, "mhx_ctx_create: device %d of %d", device, n);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mhx_ctx> c(new mhx_ctx);
    c->device = device;
    c->dtype = MHX_REAL64 ? MHX_F64 : MHX_F32;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    *out = c.release();
    return MHX_OK;
}

int api_ctx_device(const mhx_ctx* ctx) { return ctx->device; }

// ---- engine options (include/mhx.h: mhx_ctx_set_option) -------------------------------------------------------------------
// probe = 1: timing probes and fault injection -- they exist in the tools build only (libmhx_tools.so, -DMHX_TOOLS_BUILD) and
// taint the context; the release library does not even carry their names.
struct opt_name { const char* name; int probe; };
static const opt_name k_opt_names[] = {
    {"NO_PREBUILT", 0}, {"NO_MFMA", 0}, {"MFMA_WAVES", 0}, {"REG_MAX_DIM", 0}, {"REG_XR", 0}, {"REG_UNROLL", 0}, {"REG_WAVES", 0}, {"REG_ZSLAB", 0}, {"COOP_WAVES", 0},
    {"MALA_XR", 0}, {"RAM_G", 0}, {"RAM_LDS_PAD", 0}, {"WAVE_K", 0},
    {"EMCEE_MFMA", 0}, {"EMCEE_MFMA_WAVES", 0}, {"EMCEE_SCALAR", 0}, {"EMCEE_SCAL_MODE", 0}, {"EMCEE_SCAL_WPB", 0}, {"EMCEE_SCAL_REC", 0},
    {"EMCEE_FUSED", 0}, {"EMCEE_PERSIST", 0}, {"EMCEE_PRELOAD", 0}, {"EMCEE_DEFER", 0}, {"EMCEE_SWEEP_DEFER", 0}, {"EMCEE_WAVES", 0},
    {"EMCEE_REC_STORE", 0}, {"EMCEE_ROW_STORE", 0}, {"EMCEE_COOP_REC", 0},
    {"HOST_COMPACT", 0}, {"HOST_THREADS", 0}, {"HOST_CHUNK", 0}, {"HOST_NUMA", 0}, {"TOTAL_CHAINS", 0}, {"JIT_COMPILER", 0},
#ifdef MHX_TOOLS_BUILD
    {"ZIG_PROBE", 1}, {"ZIG_FORCE_FAIL", 1}, {"JIT_DEFS", 1}, {"JIT_FLAGS", 1}, {"EMCEE_PROBE", 1}, {"EMCEE_STAMPS", 1}, {"EMCEE_STAMPS_FILE", 1},
    {"FAULT_SLAB", 1}, {"RAM_PROF", 1},
#endif
};
static const opt_name* opt_find(const char* name)
{
    if (name)
        for (const opt_name& o : k_opt_names)
            if (!strcmp(o.name, name)) return &o;
    return nullptr;
}
// probe options are compiled out of the release library, names included
#ifdef MHX_TOOLS_BUILD
#define MHX_PROBE_OPT(ctx, name) opt(ctx, name)
#else
#define MHX_PROBE_OPT(ctx, name) ((const char*)nullptr)
#endif
// the value of an option, or nullptr when unset (the shape of the getenv() calls this replaced)
static const char* opt(const mhx_ctx* ctx, const char* name)
{
    if (!ctx || ctx->options.empty()) return nullptr;
    auto it = ctx->options.find(name);
    return it == ctx->options.end() ? nullptr : it->second.c_str();
}
// The chains of the WHOLE run a shard belongs to (option TOTAL_CHAINS; mhx_group_shard sets it on the member's context, a process
// of a multi-process run sets it itself): the kernel form -- and with it the summation order of the log-density -- is chosen for
// that count, so that a shard takes the form the unsharded run takes and stays bit for bit its part of it (the
// wave-per-chain kernel was picked from the LOCAL count: 8 x 1024 chains ran shape 64 where 8192 run shape 1).
static long chains_of_whole_run(const mhx_ctx* ctx, long local)
{
    const char* v = opt(ctx, "TOTAL_CHAINS");
    const long all = v ? atol(v) : 0;
    return all > local ? all : local;
}

int api_ctx_set_option(mhx_ctx* ctx, const char* name, const char* value)
{
    const opt_name* o = opt_find(name);
    if (!o) return mhx_fail(MHX_EINVAL, "mhx_ctx_set_option: unknown option '%s'%s", name ? name : "(null)",
#ifdef MHX_TOOLS_BUILD
                            "");
#else
                            " (timing probes and fault injection exist in the tools build only: libmhx_tools.so)");
#endif
    if (!value) { ctx->options.erase(name); return MHX_OK; }
    ctx->options[name] = value;
    if (o->probe) ctx->tainted = true;              // sticky: a context that ever carried a probe stays marked
    return MHX_OK;
}
int api_ctx_get_option(const mhx_ctx* ctx, const char* name, char* buf, size_t len)
{
    if (!opt_find(name)) return mhx_fail(MHX_EINVAL, "mhx_ctx_get_option: unknown option '%s'", name ? name : "(null)");
    if (!buf || !len) return mhx_fail(MHX_EINVAL, "mhx_ctx_get_option: no buffer");
    const char* v = opt(ctx, name);
    snprintf(buf, len, "%s", v ? v : "");
    return MHX_OK;
}
int api_ctx_pci_bus_id(const mhx_ctx* ctx, char* buf, size_t len)
{
    if (!buf || len < 16) return mhx_fail(MHX_EINVAL, "mhx_ctx_pci_bus_id: buffer of >= 16 bytes needed");
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)len, ctx->device));
    return MHX_OK;
}
int api_ctx_host_pin_counts(const mhx_ctx* ctx, long* registered, long* released)
{
    if (registered) *registered = ctx->pins_registered;
    if (released) *released = ctx->pins_released;
    return MHX_OK;
}

int api_ctx_jit_counts(const mhx_ctx* ctx, long* compiles, long* cache_hits)
{
    if (compiles) *compiles = ctx->jit_compiles;
    if (cache_hits) *cache_hits = ctx->jit_cache_hits;
    return MHX_OK;
}

int api_ctx_jit_compiler(const mhx_ctx* ctx, char* compiler, size_t len, long* ext_compiles)
{
    if (compiler && len) {
        const char* jc = nullptr;
        auto it = ctx->options.find("JIT_COMPILER");
        if (it != ctx->options.end()) jc = it->second.c_str();
        if (!jc) { jc = getenv("MHX_JIT_COMPILER"); if (jc && !*jc) jc = nullptr; }
        const std::string id = (jc && !strcmp(jc, "hiprtc")) ? std::string() : mhx_jit_ext_identity();
        snprintf(compiler, len, "%s", id.c_str());
    }
    if (ext_compiles) *ext_compiles = ctx->jit_ext_compiles;
    return MHX_OK;
}

int api_ctx_destroy(mhx_ctx* ctx)
{
    if (!ctx) return MHX_OK;
    (void)hipSetDevice(ctx->device);
    delete ctx;
    return MHX_OK;
}

// ---- persistent code-object cache -------------------------------------------------------------------------------------
// hiprtc takes 0.3 s for a small specialisation and up to a minute for the unrolled matrix-core kernels (d = 1000); the code
// object depends only on (source, embedded headers, options, compiler), so it is kept on disk: $MHX_CACHE_DIR, else
// $XDG_CACHE_HOME/mhx, else $HOME/.cache/mhx; MHX_CACHE_DIR="" (or MHX_NO_JIT_CACHE=1) switches it off.  Files are written to
// a temporary name and renamed, so concurrent processes (one per GPU) never see a partial object.
static unsigned long long fnv64(const void* data, size_t n, unsigned long long h)
{
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
static std::string jit_cache_dir()
{
    if (const char* off = getenv("MHX_NO_JIT_CACHE")) if (*off && *off != '0') return "";
    std::string dir;
    if (const char* e = getenv("MHX_CACHE_DIR")) { if (!*e) return ""; dir = e; }
    else if (const char* x = getenv("XDG_CACHE_HOME")) { if (*x) dir = std::string(x) + "/mhx"; }
    if (dir.empty()) { const char* h = getenv("HOME"); if (!h || !*h) return ""; dir = std::string(h) + "/.cache/mhx"; }
    // mkdir -p (two levels are enough for the defaults; an explicit MHX_CACHE_DIR must have an existing parent)
    const size_t cut = dir.find_last_of('/');
    if (cut != std::string::npos && cut > 0) (void)mkdir(dir.substr(0, cut).c_str(), 0755);
    if (mkdir(dir.c_str(), 0755) != 0 && errno != EEXIST) return "";
    return dir;
}
static std::string jit_cache_name(const std::string& source, const std::vector<std::string>& opts, const char* const* hdr_src, int nhdr,
                                  const std::string& compiler_id)
{
    unsigned long long h1 = 1469598103934665603ull, h2 = 0x9e3779b97f4a7c15ull;
    auto mix = [&](const void* d, size_t n) { h1 = fnv64(d, n, h1); h2 = fnv64(d, n, h2 ^ (unsigned long long)n); };
    mix(source.data(), source.size());
    mix(compiler_id.data(), compiler_id.size());
    for (auto& o : opts) { mix(o.data(), o.size()); mix("\0", 1); }
    for (int i = 0; i < nhdr; ++i) mix(hdr_src[i], strlen(hdr_src[i]));
    int vmaj = 0, vmin = 0, rt = 0;
    (void)hiprtcVersion(&vmaj, &vmin);
    (void)hipRuntimeGetVersion(&rt);
    const int ver[4] = {vmaj, vmin, rt, MHX_VERSION};
    mix(ver, sizeof ver);
    char b[64];
    snprintf(b, sizeof b, "%016llx%016llx.hsaco", h1, h2);
    return b;
}
static bool jit_cache_read(const std::string& path, std::vector<char>* code)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long n = ftell(f);
        if (n > 16 && fseek(f, 0, SEEK_SET) == 0) {
            code->resize((size_t)n);
            ok = fread(code->data(), 1, (size_t)n, f) == (size_t)n && memcmp(code->data(), "\x7f" "ELF", 4) == 0;
        }
    }
    fclose(f);
    return ok;
}
static void jit_cache_write(const std::string& dir, const std::string& name, const std::vector<char>& code)
{
    char tmp[64];
    snprintf(tmp, sizeof tmp, "/.tmp.%ld.%llx", (long)getpid(), (unsigned long long)(size_t)code.data());
    const std::string t = dir + tmp;
    FILE* f = fopen(t.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    if (fclose(f) != 0 || !ok || rename(t.c_str(), (dir + "/" + name).c_str()) != 0) (void)remove(t.c_str());
}

// compile `source` (which #includes the embedded device headers) with hiprtc for gfx950
static int jit_compile(mhx_ctx* ctx, const std::string& key_in, const std::string& source,
                       const std::vector<std::string>& defines, jit_module** out, const std::vector<std::string>& extra_opts = {})
{
    // tools build, option JIT_DEFS = "NAME=VALUE NAME=VALUE": extra defines for every hiprtc compile of the context -- how the A/B
    // scripts reach the kernels' compile-time knobs; part of both cache keys.  It taints the context (a define can change a
    // kernel's LDS layout behind the host's back), and the release library has no such door.
    const char* xdefs = MHX_PROBE_OPT(ctx, "JIT_DEFS");
    struct jit_guard { jit_guard() { mhx_jit_lock(); } ~jit_guard() { mhx_jit_unlock(); } } one_at_a_time;
    const char* xflags = MHX_PROBE_OPT(ctx, "JIT_FLAGS");      // (tools build: raw compiler options, blank-separated, for A/B of code generation)
    const std::string key = ((xdefs && *xdefs) ? key_in + "/xd=" + xdefs : key_in) + ((xflags && *xflags) ? std::string("/xf=") + xflags : std::string());
    auto it = ctx->jit.find(key);
    if (it != ctx->jit.end()) { *out = it->second.get(); return MHX_OK; }
    const char* hdr_src[] = {k_src_mhx_zig_table_h, k_src_mhx_device_math_h, k_src_mhx_targets_h, k_src_mhx_rwmh_kernels_h,
                             k_src_mhx_emcee_kernels_h, k_src_mhx_ram_kernels_h, k_src_mhx_mala_kernels_h,
                             k_src_mhx_rwmh_dense_kernels_h, k_src_mhx_rwmh_mfma_kernels_h,
                             k_src_mhx_mala_mfma_kernels_h, k_src_mhx_emcee_mfma_kernels_h};
    const char* hdr_name[] = {"mhx_zig_table.h", "mhx_device_math.h", "mhx_targets.h", "mhx_rwmh_kernels.h",
                              "mhx_emcee_kernels.h", "mhx_ram_kernels.h", "mhx_mala_kernels.h",
                              "mhx_rwmh_dense_kernels.h", "mhx_rwmh_mfma_kernels.h", "mhx_mala_mfma_kernels.h", "mhx_emcee_mfma_kernels.h"};
    std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                                     MHX_REAL64 ? "-DMHX_REAL64=1" : "-DMHX_REAL64=0", "-DMHX_XW_LINE=" + std::to_string(MHX_XW_LINE)};
#ifdef MHX_TOOLS_BUILD
    opts.push_back("-DMHX_TOOLS_BUILD=1");              // the embedded headers of this build carry the probes
#endif
    for (auto& d : defines) opts.push_back("-D" + d);
    if (xdefs)
        for (const char* p = xdefs; *p;) {
            while (*p == ' ') ++p;
            const char* e = p;
            while (*e && *e != ' ') ++e;
            if (e > p) opts.push_back("-D" + std::string(p, e));
            p = e;
        }
    for (auto& o : extra_opts) opts.push_back(o);
    if (xflags)
        for (const char* p = xflags; *p;) {
            while (*p == ' ') ++p;
            const char* e = p;
            while (*e && *e != ' ') ++e;
            if (e > p) opts.push_back(std::string(p, e));
            p = e;
        }
    // Which compiler: the installation's clang++ where there is one (mhx_jit_ext.h: hiprtc is whichever copy the process loaded first,
    // a PyTorch wheel's older one inside Python), hiprtc otherwise or when option JIT_COMPILER says "hiprtc"; "clang" = no fall-back
    const char* jc = opt(ctx, "JIT_COMPILER");
    if (!jc) { jc = getenv("MHX_JIT_COMPILER"); if (jc && !*jc) jc = nullptr; }     // (the process-wide default of the option)
    const bool want_ext = !(jc && !strcmp(jc, "hiprtc")) && !mhx_jit_ext_identity().empty();
    if (jc && !strcmp(jc, "clang") && !want_ext) return mhx_fail(MHX_EJIT, "JIT_COMPILER=clang: no clang++ found (MHX_JIT_CLANG, ROCM_PATH, /opt/rocm)");
    const std::string cdir = jit_cache_dir();
    const std::string cname = cdir.empty() ? std::string() : jit_cache_name(source, opts, hdr_src, 11, want_ext ? mhx_jit_ext_identity() : std::string());
    // (where the offline compiler is preferred but failed on this source before, hiprtc's object is there under its own name)
    const std::string cname_rtc = (cdir.empty() || !want_ext) ? cname : jit_cache_name(source, opts, hdr_src, 11, std::string());
    std::vector<char> code;
    bool from_cache = !cdir.empty() && (jit_cache_read(cdir + "/" + cname, &code) || (want_ext && jit_cache_read(cdir + "/" + cname_rtc, &code)));
    if (from_cache) {
        std::unique_ptr<jit_module> m(new jit_module);
        if (hipModuleLoadData(&m->mod, code.data()) == hipSuccess) {
            ctx->jit_cache_hits++;
            *out = m.get();
            ctx->jit[key] = std::move(m);
            return MHX_OK;
        }
        (void)hipGetLastError();            // a stale / foreign object: fall through and compile
        from_cache = false;
    }
    if (want_ext) {
        std::string xlog;
        if (mhx_jit_ext_compile(source, hdr_src, hdr_name, 11, opts, &code, &xlog)) {
            std::unique_ptr<jit_module> m(new jit_module);
            const hipError_t e = hipModuleLoadData(&m->mod, code.data());
            if (e == hipSuccess) {
                ctx->jit_compiles++;
                ctx->jit_ext_compiles++;
                if (!cdir.empty()) jit_cache_write(cdir, cname, code);
                *out = m.get();
                ctx->jit[key] = std::move(m);
                return MHX_OK;
            }
            (void)hipGetLastError();
            xlog = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        }
        if (jc && !strcmp(jc, "clang")) {
            if (xlog.size() > 1800) xlog.resize(1800);
            return mhx_fail(MHX_EJIT, "JIT_COMPILER=clang: %s", xlog.c_str());
        }
        // (a source that does not compile fails here too: hiprtc's log is the one the caller gets; MHX_JIT_VERBOSE=1 shows this one)
        if (const char* v = getenv("MHX_JIT_VERBOSE")) if (*v && *v != '0')
            fprintf(stderr, "mhx: the offline compiler failed on %s, hiprtc takes over: %.1500s\n", key.c_str(), xlog.c_str());
    }
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, source.c_str(), "mhx_jit.hip", 11, hdr_src, hdr_name);
    if (r != HIPRTC_SUCCESS) return mhx_fail(MHX_EJIT, "hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
    std::vector<const char*> copts;
    for (auto& o : opts) copts.push_back(o.c_str());
    r = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        if (log.size() > 1800) log.resize(1800);
        return mhx_fail(MHX_EJIT, "hiprtc compile failed (%s): %s", hiprtcGetErrorString(r), log.c_str());
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    ctx->jit_compiles++;
    std::unique_ptr<jit_module> m(new jit_module);
    hipError_t e = hipModuleLoadData(&m->mod, code.data());
    if (e != hipSuccess) return mhx_fail(MHX_EJIT, "hipModuleLoadData: %s", hipGetErrorString(e));
    if (!cdir.empty()) jit_cache_write(cdir, cname_rtc, code);
    *out = m.get();
    ctx->jit[key] = std::move(m);
    return MHX_OK;
}

static int jit_function(jit_module* m, const char* name, hipFunction_t* fn)
{
    auto it = m->fns.find(name);
    if (it != m->fns.end()) { *fn = it->second; return MHX_OK; }
    hipError_t e = hipModuleGetFunction(fn, m->mod, name);
    if (e != hipSuccess) return mhx_fail(MHX_EJIT, "hipModuleGetFunction(%s): %s", name, hipGetErrorString(e));
    m->fns[name] = *fn;
    return MHX_OK;
}

// ---------------------------------------------------------------------------------------------
// targets
struct mhx_target : mhx_handle_hdr {
    mhx_ctx* ctx = nullptr;
    int kind = 0, dim = 0, nparams = 0;
    mhx_real cst = MHX_R(0.0);
    mhx_real* dparams = nullptr;
    std::string user_src;        // MHX_TARGET_USER
    std::string user_key;        // hash of the source, for the JIT cache
    int bandwidth = -1;          // CORR_GAUSS: largest r - c with A_rc != 0 (exact zeros only), -1 otherwise: a banded precision
                                 // factor (Markov / autoregressive model) lets the row products skip the zeros
    ~mhx_target() { if (dparams) (void)hipFree(dparams); }
};

static mhx_real target_const(int kind, int dim, const mhx_real* p)
{
    const double LOG_2PI = 1.8378770664093454835606594728112;
    double c = -0.5 * (double)dim * LOG_2PI;
    switch (kind) {
    case MHX_TARGET_CORR_GAUSS: {
        size_t off = 0;
        for (int i = 0; i < dim; ++i) { c += std::log((double)p[off + i]); off += (size_t)i + 1; }
        break;
    }
    case MHX_TARGET_BANANA: c -= 0.5 * std::log(100.0); break;
    case MHX_TARGET_FUNNEL: c -= std::log(3.0); break;
    case MHX_TARGET_IID_NORMAL:
    case MHX_TARGET_USER: c = 0.0; break;
    default: break;
    }
    return (mhx_real)c;
}

static int target_upload(mhx_target* t, const mhx_real* params, size_t nparams)
{
    t->nparams = (int)nparams;
    // one dummy element keeps the pointer valid for kinds without parameters
    const size_t n = nparams ? nparams : 1;
    HIP_TRY(hipMalloc(&t->dparams, n * sizeof(mhx_real)));
    if (nparams) COPY_SYNC(t->ctx->stream, t->dparams, params, nparams * sizeof(mhx_real), hipMemcpyHostToDevice);
    else HIP_TRY(hipMemsetAsync(t->dparams, 0, sizeof(mhx_real), t->ctx->stream));
    return MHX_OK;
}

int api_target_builtin(mhx_ctx* ctx, int kind, int dim, const mhx_real* params, size_t nparams,
                                  mhx_target** out)
{
    if (!ctx || !out) return mhx_fail(MHX_EINVAL, "mhx_target_builtin: NULL argument");
    if (dim <= 0) return mhx_fail(MHX_EINVAL, "mhx_target_builtin: dim must be positive, got %d", dim);
    if (nparams && !params) return mhx_fail(MHX_EINVAL, "mhx_target_builtin: params is NULL");
    const size_t tri = (size_t)dim * ((size_t)dim + 1) / 2;
    switch (kind) {
    case MHX_TARGET_ISO_GAUSS:
        if (nparams) return mhx_fail(MHX_EINVAL, "ISO_GAUSS takes no parameters");
        break;
    case MHX_TARGET_CORR_GAUSS:
        if (nparams != tri) return mhx_fail(MHX_EINVAL, "CORR_GAUSS needs %zu packed-lower parameters, got %zu", tri, nparams);
        for (int i = 0; i < dim; ++i)
            if (!(params[(size_t)i * (i + 1) / 2 + i] > MHX_R(0.0)))
                return mhx_fail(MHX_ENOTPD, "CORR_GAUSS: diagonal %d of inv(chol(Sigma)) is not positive", i);
        break;
    case MHX_TARGET_IID_NORMAL:
        if (dim != 2) return mhx_fail(MHX_EINVAL, "IID_NORMAL is a 2-parameter (mu, sigma) model, dim=%d", dim);
        if (nparams < 1) return mhx_fail(MHX_EINVAL, "IID_NORMAL needs at least one data point");
        break;
    case MHX_TARGET_BANANA:
        if (dim < 2 || nparams != 1) return mhx_fail(MHX_EINVAL, "BANANA needs dim >= 2 and params = {b}");
        break;
    case MHX_TARGET_FUNNEL:
        if (dim < 2 || nparams != 0) return mhx_fail(MHX_EINVAL, "FUNNEL needs dim >= 2 and no parameters");
        break;
    default:
        return mhx_fail(MHX_EINVAL, "mhx_target_builtin: unknown kind %d", kind);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<mhx_target> t(new mhx_target);
    t->dtype = ctx->dtype;
    t->ctx = ctx;
    t->kind = kind;
    t->dim = dim;
    t->cst = target_const(kind, dim, params);
    if (kind == MHX_TARGET_CORR_GAUSS) {
        int bw = 0;
        size_t off = 0;
        for (int r = 0; r < dim; ++r) {
            for (int c = 0; c < r - bw; ++c) if (params[off + c] != MHX_R(0.0)) { bw = r - c; break; }
            off += (size_t)r + 1;
        }
        t->bandwidth = bw;
    }
    int rc = target_upload(t.get(), params, nparams);
    if (rc) return rc;
    *out = t.release();
    return MHX_OK;
}

static std::string fnv_hex(const std::string& s)
{
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    char b[32];
    snprintf(b, sizeof b, "%016llx", h);
    return b;
}

static std::string jit_source(const mhx_target* t, const char* family_header)
{
    std::string s = "#include \"mhx_device_math.h\"\n";
    if (t->kind == MHX_TARGET_USER) {
        s += "#line 1 \"user_logdensity.hip\"\n";
        s += t->user_src;
        s += "\n#define MHX_HAVE_USER_TARGET 1\n";
    }
    s += "#include \"";
    s += family_header;
    s += "\"\n";
    return s;
}

static int jit_generic_rwmh(const mhx_target* t, jit_module** m)
{
    const std::string key = "rwmh_generic/tk=" + std::to_string(t->kind) + "/" + t->user_key;
    return jit_compile(t->ctx, key, jit_source(t, "mhx_rwmh_kernels.h"),
                       {"MHX_JIT_RWMH_GENERIC=1", "MHX_JIT_TK=" + std::to_string(t->kind)}, m);
}

int api_target_from_hip_source(mhx_ctx* ctx, const char* src, int dim, const mhx_real* data,
                                          size_t ndata, mhx_target** out)
{
    if (!ctx || !src || !out) return mhx_fail(MHX_EINVAL, "mhx_target_from_hip_source: NULL argument");
    if (dim <= 0) return mhx_fail(MHX_EINVAL, "mhx_target_from_hip_source: dim must be positive");
    if (ndata && !data) return mhx_fail(MHX_EINVAL, "mhx_target_from_hip_source: data is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<mhx_target> t(new mhx_target);
    t->dtype = ctx->dtype;
    t->ctx = ctx;
    t->kind = MHX_TARGET_USER;
    t->dim = dim;
    t->cst = MHX_R(0.0);
    t->user_src = src;
    t->user_key = fnv_hex(t->user_src);
    int rc = target_upload(t.get(), data, ndata);
    if (rc) return rc;
    // compile now so that a bad source is reported at model construction, like a MethodError would be
    jit_module* m = nullptr;
    rc = jit_generic_rwmh(t.get(), &m);
    if (rc) return rc;
    *out = t.release();
    return MHX_OK;
}

int api_target_destroy(mhx_target* t)
{
    if (!t) return MHX_OK;
    (void)hipSetDevice(t->ctx->device);
    delete t;
    return MHX_OK;
}

static int launch_module(hipFunction_t fn, unsigned grid, unsigned block, hipStream_t s, void** params)
{
    HIP_TRY(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, s, params, nullptr));
    return MHX_OK;
}

int api_target_eval(mhx_ctx* ctx, const mhx_target* t, const mhx_real* x, int n, mhx_real* lp)
{
    if (!ctx || !t || !x || !lp || n <= 0) return mhx_fail(MHX_EINVAL, "mhx_target_eval: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    mhx_real *dx = nullptr, *dlp = nullptr;
    const size_t nx = (size_t)t->dim * (size_t)n;
    HIP_TRY(hipMalloc(&dx, nx * sizeof(mhx_real)));
    HIP_TRY(hipMalloc(&dlp, (size_t)n * sizeof(mhx_real)));
    int rc = MHX_OK;
    do {
        if (hipMemcpyAsync(dx, x, nx * sizeof(mhx_real), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = mhx_fail(MHX_EHIP, "H2D copy failed"); break; }
        const unsigned grid = (unsigned)((n + 255) / 256);
        int d = t->dim, kind = t->kind, np = t->nparams, lanes = 1;
        mhx_real cst = t->cst;
        const mhx_real* tp = t->dparams;
        if (t->kind == MHX_TARGET_USER) {
            jit_module* m = nullptr;
            hipFunction_t fn;
            if ((rc = jit_generic_rwmh(t, &m))) break;
            if ((rc = jit_function(m, "mhx_jit_target_eval", &fn))) break;
            void* params[] = {&dx, &dlp, &n, &d, &kind, &tp, &np, &cst, &lanes};
            if ((rc = launch_module(fn, grid, 256, ctx->stream, params))) break;
        } else {
            hipLaunchKernelGGL(k_target_eval, dim3(grid), dim3(256), 0, ctx->stream, dx, dlp, n, d, kind, tp, np, cst, lanes);
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) { rc = mhx_fail(MHX_EHIP, "target_eval kernel failed"); break; }
        if (hipMemcpyAsync(lp, dlp, (size_t)n * sizeof(mhx_real), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = mhx_fail(MHX_EHIP, "D2H copy failed"); break; }
    } while (0);
    (void)hipFree(dx);
    (void)hipFree(dlp);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// runs
enum run_kind { RUN_RWMH = 0, RUN_EMCEE = 1, RUN_RAM = 2, RUN_MALA = 3 };

struct mhx_run : mhx_handle_hdr {
    mhx_ctx* ctx = nullptr;
    const mhx_target* target = nullptr;
    int kind = RUN_RWMH;
    int dim = 0, n = 0;                  // n = chains / walkers (an Ensemble run: all walkers of all its ensembles)
    int ens_w = 0, n_ens = 1;            // Ensemble runs: walkers per ensemble, ensembles (ids first_id .. first_id + n_ens - 1)
    uint64_t seed = 0, first_id = 0;
    int flags = 0;
    bool initialised = false;
    uint64_t tau = 0;                    // transitions done so far (RNG step counter)
    // rwmh
    int prop_kind = 0;
    mhx_real prop_scale = MHX_R(1.0);
    mhx_real* d_pvec = nullptr;
    // emcee
    mhx_real stretch = MHX_R(2.0);
    size_t dense_lds = 0;                // dynamic LDS bytes of the dense cooperative RWMH kernel
    mhx_real* d_xw = nullptr;               // walker-major copy [W][round4(dim)]: the state while the cooperative kernel runs
    mhx_real* d_pmean = nullptr;            // drifting walk: mu[dim] then 2 L^-1 mu [dim]
    mhx_real* d_mfma_img = nullptr;         // streamed matrix-core kernel: the operand images [target][proposal] in global memory
    bool mfma_stream = false;
    mhx_real* d_qx = nullptr;               // static proposal: logpdf of the proposal at each chain's state (up to its constant)
    // mala
    mhx_real mala_sigma = MHX_R(1.0);
    mhx_real *d_gx = nullptr, *d_gy = nullptr, *d_z = nullptr;
    // ram
    mhx_ram_cfg ramcfg{};
    mhx_real* d_S = nullptr;                        // packed factors [n][2][tri_pad]: both buffers of a chain side by side
    unsigned char* d_Ssel = nullptr;             // which buffer holds chain c's current factor
    unsigned char* d_status = nullptr;
    mhx_real *d_dmin = nullptr, *d_dmax = nullptr;  // [dim][n]
    mhx_real* d_eta = nullptr;                      // adaptation step sizes of the current launch
    mhx_real* d_loga = nullptr;                     // [n] log acceptance ratio of each chain's latest transition
    mhx_real* d_defer = nullptr;                    // RAM, deferred-factor form: [n][MHX_RAM_DEFER_REALS] pending updates
    int defer_R = 0;                                // rows per lane of that form's kernel (0: the sweep form)
    double last_eta = 0.0;                          // step size of the latest adapting transition (state.η; 0 before any)
    mhx_real* d_rec_loga = nullptr;                 // [n_saved][n] logα of every recorded transition of the last sampling call
    mhx_real* rec_loga_view = nullptr;              // where slot 0 of the current launches lands in it (slab-wise calls)
    size_t rec_loga_cap = 0;
    // RAM: the factors of a few watched chains after every recorded step (mhx_ram_watch_factors)
    int32_t* d_watch_chains = nullptr;
    mhx_real* d_watch = nullptr;                    // [n recorded][watch_n][dim (dim + 1) / 2], packed row-major
    size_t watch_cap = 0;
    int watch_n = 0;
    long watch_count = 0;
    std::vector<double> rec_eta;                    // [n_saved] state.η after every recorded transition
    int64_t rec_n = 0;                              // recorded transitions held by the two above
    size_t eta_cap = 0;
    // state
    mhx_real *d_x = nullptr, *d_lp = nullptr, *d_ybuf = nullptr;
    uint32_t* d_acc = nullptr;
    unsigned char* d_last = nullptr;
    unsigned long long* d_acc_total = nullptr;
    // sample buffer of the last mhx_run_sample
    mhx_real* d_samples = nullptr;
    unsigned char* d_accepted = nullptr;
    size_t samples_cap = 0, accepted_cap = 0;
    int64_t n_saved = 0;
    // running moments of the last mhx_run_sample(save = 2)
    mhx_real *d_mom_mean = nullptr, *d_mom_m2 = nullptr;
    size_t mom_cap = 0, mom_cap2 = 0;
    uint64_t mom_n = 0;                  // states folded in so far
    bool moments_mode = false;
    void (*reg_fn_mom)(const mhx_rwmh_args, const mhx_real*, const mhx_real*) = nullptr;
    hipFunction_t jit_step_mom = nullptr;
    std::string coop_key;                // JIT key / defines of the cooperative kernel (for its moments twin)
    std::vector<std::string> coop_defs;
    // kernel choice
    int normal_gen = MHX_GEN_BOX_MULLER; // how stream bits become standard normals (MHX_FLAG_ZIGGURAT: the table ziggurat, fp64)
    size_t reg_lds = 0;                  // dynamic LDS of the register kernel: the tail of a state that does not fit beside the candidate
    size_t coop_lds = 0;                 // dynamic LDS of the cooperative kernel (ziggurat: layer table + the step's normals)
    int coop_tr = 0;                     // ... and whether it holds the row-transposition buffer of the one / two-chains-per-wave shapes
    int coop_L = 1;                      // lanes per chain (reduction shape of the separable targets)
    int emcee_band = -1;                 // bandwidth of the precision factor the cooperative stretch move exploits (-1: dense form)
    int coop_waves = MHX_EMCEE_COOP_WAVES;  // waves per block of the cooperative stretch move (tuning knob MHX_EMCEE_WAVES)
    size_t emcee_stamp_words = 0;        // MHX_EMCEE_STAMPS (tools): 64-bit words of the stamp buffer in d_ybuf
    int emcee_wpb = 64;                  // walkers per block of the scalar-factor form
    hipFunction_t jit_persist = nullptr; // a small ensemble as one persistent block (mhx_emcee_persist_body): a whole call per launch
    size_t persist_lds = 0;
    hipFunction_t jit_sweep = nullptr;   // the stretch move as ONE launch per sweep (mhx_emcee_coop_sweep_body); state double-buffered:
    mhx_real *d_xw2 = nullptr, *d_lp2 = nullptr;   // ... the buffers the next sweep writes (swapped with d_xw / d_lp after every launch)
    size_t sweep_lds = 0;
    bool emcee_preload = false;          // the half-step kernel takes its hot arguments as preloaded scalars (MHX_JIT_PRELOAD)
    bool emcee_mfma = false;             // the matrix-core form of the stretch move (variant 10): 4 lanes per walker, operand image in d_mfma_img
    bool emcee_scal = false;             // the scalar-factor form of the cooperative stretch move (variant 9): coop_L waves per block, 64 walkers
    int variant = 0;
    void (*reg_fn)(const mhx_rwmh_args, const mhx_real*, const mhx_real*) = nullptr;
    void (*reg_fn8)(const mhx_rwmh_args, const mhx_real*, const mhx_real*) = nullptr;   // variant 11: the K = 8 build of the wave-per-chain kernel
    int wave_k = 4;                      // variant 11: candidates per round of the next call (8 after a call that accepted < 1 step in 8)
    hipFunction_t jit_step = nullptr, jit_init = nullptr;
    mhx_stats stats{};
    mhx_host_stats host_stats{};         // what the last mhx_run_sample_to_host moved

    ~mhx_run()
    {
        void* ptrs[] = {d_pvec, d_S, d_Ssel, d_status, d_dmin, d_dmax, d_eta, d_x, d_lp, d_ybuf,
                        d_acc, d_last, d_acc_total, d_samples, d_accepted, d_mom_mean, d_mom_m2, d_gx, d_gy, d_z, d_pmean, d_qx, d_xw, d_loga, d_mfma_img, d_rec_loga,
                        d_watch_chains, d_watch, d_xw2, d_lp2, d_defer};
        for (void* p : ptrs) if (p) (void)hipFree(p);
    }
};

static int run_alloc_state(mhx_run* r)
{
    const size_t n = (size_t)r->n, d = (size_t)r->dim;
    HIP_TRY(hipMalloc(&r->d_x, d * n * sizeof(mhx_real)));
    HIP_TRY(hipMalloc(&r->d_lp, n * sizeof(mhx_real)));
    HIP_TRY(hipMalloc(&r->d_acc, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&r->d_last, n));
    HIP_TRY(hipMalloc(&r->d_acc_total, sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(r->d_acc, 0, n * sizeof(uint32_t), r->ctx->stream));
    HIP_TRY(hipMemsetAsync(r->d_last, 0, n, r->ctx->stream));
    HIP_TRY(hipMemsetAsync(r->d_acc_total, 0, sizeof(unsigned long long), r->ctx->stream));
    return MHX_OK;
}

static mhx_rwmh_args rwmh_args(const mhx_run* r)
{
    mhx_rwmh_args a;
    memset(&a, 0, sizeof a);
    a.x = r->d_x; a.lp = r->d_lp; a.acc_count = r->d_acc; a.acc_total = r->d_acc_total;
    a.samples = r->d_samples; a.accepted = r->d_accepted; a.last_acc = r->d_last; a.ybuf = r->d_ybuf;
    a.seed = r->seed; a.first_chain = r->first_id;
    a.nchains = r->n; a.ld = r->n; a.dim = r->dim;
    a.target_kind = r->target->kind; a.ntparams = r->target->nparams; a.tconst = r->target->cst;
    a.prop_kind = r->prop_kind; a.pscale = r->prop_scale;
    a.save_next = MHX_NO_SAVE; a.thinning = 1;
    a.reduce_lanes = r->coop_L;
    if (r->moments_mode) { a.mom_mean = r->d_mom_mean; a.mom_m2 = r->d_mom_m2; a.mom_n0 = (mhx_u32)r->mom_n; }
    a.pmean = r->d_pmean;
    a.qx = r->d_qx;
    a.normal_gen = r->normal_gen;
    a.tr_lds = r->coop_tr;
    return a;
}

__global__ void __launch_bounds__(256)
k_rwmh_whiten(const mhx_rwmh_args a, const mhx_real* __restrict__ pvec)
{
    mhx_rwmh_whiten_body(a, pvec);
}
// logpdf of a static proposal at the current states (after init / set_state)
static int rwmh_whiten(mhx_run* r)
{
    if (!r->d_qx) return MHX_OK;
    mhx_rwmh_args a = rwmh_args(r);
    hipLaunchKernelGGL(k_rwmh_whiten, dim3((unsigned)((r->n + 255) / 256)), dim3(256), 0, r->ctx->stream, a, r->d_pvec);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    return MHX_OK;
}

// register budgets: a double takes two VGPRs
#define MHX_REG_MAX_DIM (MHX_REAL64 ? 80 : 160)
#define MHX_REG_MAX_DIM_DENSE (MHX_REAL64 ? 48 : 96)
#define MHX_REG_XR_ALL (MHX_REAL64 ? 64 : 128)          // up to here the whole state stays in registers beside the candidate
#define MHX_REG_XR_CAP (MHX_REAL64 ? 60 : 120)          // above: this many coordinates of it, the rest in LDS
#define MHX_DENSE_COOP_MAX_DIM 256
#define MHX_EMCEE_MAX_BAND 8                 // widest band the band form of the cooperative stretch move is specialised for
#define MHX_LDS_PER_BLOCK 163840             // gfx950: 160 KB of LDS, all of it available to one block
// lanes per chain of the dense cooperative kernel: at most 12.5 rows of a factor per lane
static int dense_coop_lanes(int d) { int L = 2; while ((MHX_REAL64 ? 4 : 2) * d > 25 * L && L < 64) L *= 2; return L; }
// dynamic LDS of the dense cooperative kernel: the candidate rows of a 4-wave block + `nimages` factor images
static size_t dense_coop_lds_bytes(int d, int L, int nimages, int waves = MHX_EMCEE_COOP_WAVES)
{
    long total4 = 0;
    for (int m = 0; m * L < d; ++m) total4 += (long)((std::min(L * (m + 1), d) + 3) / 4) * L;
    const long rows4 = (long)waves * (64 / L) * ((((d + 3) & ~3) + 4) / 4);
    return (size_t)(rows4 + nimages * total4) * 4 * sizeof(mhx_real);
}
static bool dense_coop_fits(int d, int L, int nimages)
{
    if (L < 2 || L > 64 || (L & (L - 1)) || L > d) return true;      // let the caller report the bad shape
    return dense_coop_lds_bytes(d, L, nimages) <= MHX_LDS_PER_BLOCK;
}

// the matrix-core kernel (mhx_rwmh_mfma_kernels.h): 4 lanes per chain, NS = ceil(d/4) reals of state and of candidate per
// lane, `nimages` operand images in LDS
#define MHX_MFMA_MAX_NS (MHX_REAL64 ? 44 : 64)
static size_t mfma_image_reals(int d)
{
    const int NS = (d + 3) / 4, NT = (d + 15) / 16;
    const int last = std::min(4 * NT, NS);
    return (size_t)(2 * (NT - 1) * NT + 4 * ((last + 3) / 4)) * 64;
}
static size_t mfma_image_reals_T(int d) { const int NT = (d + 15) / 16; return (size_t)(NT * (NT + 1) / 2) * 256; }   // the image of A^T
static bool mfma_fits(const mhx_ctx* ctx, int d, int reduce_lanes, int nimages)
{
    const char* no_mfma = opt(ctx, "NO_MFMA");                      // tuning knob: the vector kernel instead
    return (reduce_lanes == 0 || reduce_lanes == 4) && d >= 16 && (d + 3) / 4 <= MHX_MFMA_MAX_NS && !(no_mfma && atoi(no_mfma)) &&
           nimages * mfma_image_reals(d) * sizeof(mhx_real) <= MHX_LDS_PER_BLOCK;
}
// streamed images (larger than a block's LDS): two ring buffers of whole 256-thread rounds of 16-byte pieces of the largest chunk
// -- a tile pair, or (the largest dimensions) a single tile with the chain state re-read from its slab instead of held in registers
#define MHX_MFMA_STREAM_MAX_NS (MHX_REAL64 ? 64 : 100)
#define MHX_MFMA_XMEM_MAX_NS (MHX_REAL64 ? 128 : 250)   // fp64 d <= 512, fp32 d <= 1000: the candidate alone fills the registers of a wave (1 per SIMD)
static size_t mfma_ring_bytes(int d, bool pair = true)
{
    const int NS = (d + 3) / 4, NT = (d + 15) / 16;
    auto groups = [&](int t) { return (std::min(4 * (t + 1), NS) + 3) / 4; };
    int maxg = 0;
    if (pair) for (int p = 0; 2 * p < NT; ++p) maxg = std::max(maxg, groups(2 * p) + (2 * p + 1 < NT ? groups(2 * p + 1) : 0));
    else for (int t = 0; t < NT; ++t) maxg = std::max(maxg, groups(t));
    const int p16 = (int)(4 * sizeof(mhx_real)) / 16;
    const int pf = (maxg * 64 * p16 + 64 * MHX_MFMA_WAVES - 1) / (64 * MHX_MFMA_WAVES);
    return (size_t)2 * pf * 64 * MHX_MFMA_WAVES * 16;
}
// 0 = no, 1 = tile pairs + state in registers, 2 = single tiles + state in HBM
static int mfma_stream_mode(const mhx_ctx* ctx, int d, int reduce_lanes)
{
    const char* no_mfma = opt(ctx, "NO_MFMA");
    if (!((reduce_lanes == 0 || reduce_lanes == 4) && d >= 16) || (no_mfma && atoi(no_mfma))) return 0;
    const int NS = (d + 3) / 4;
    if (NS <= MHX_MFMA_STREAM_MAX_NS && mfma_ring_bytes(d, true) <= MHX_LDS_PER_BLOCK) return 1;
    if (NS <= MHX_MFMA_XMEM_MAX_NS && mfma_ring_bytes(d, false) <= MHX_LDS_PER_BLOCK) return 2;
    return 0;
}
static bool mfma_stream_fits(const mhx_ctx* ctx, int d, int reduce_lanes) { return mfma_stream_mode(ctx, d, reduce_lanes) != 0; }
static int mfma_waves(const mhx_ctx* ctx, int d)
{
    const int NS = (d + 3) / 4;
    if (const char* w = opt(ctx, "MFMA_WAVES")) return std::max(1, atoi(w));        // tuning knob
    return MHX_REAL64 ? (NS <= 25 ? 2 : 1) : (NS <= 25 ? 3 : (NS <= 40 ? 2 : 1));
}

int api_rwmh_create(mhx_ctx* ctx, const mhx_target* t, const mhx_rwmh_cfg* cfg, mhx_run** out)
{
    if (!ctx || !t || !cfg || !out) return mhx_fail(MHX_EINVAL, "mhx_rwmh_create: NULL argument");
    if (cfg->dim != t->dim)
        return mhx_fail(MHX_EINVAL, "mhx_rwmh_create: proposal dim %d != model dim %d", cfg->dim, t->dim);
    if (cfg->nchains <= 0) return mhx_fail(MHX_EINVAL, "mhx_rwmh_create: nchains must be positive");
    const int d = cfg->dim;
    const mhx_real* cfg_vec = (const mhx_real*)cfg->proposal_vec;       // reals of this instantiation (include/mhx.h)
    const mhx_real* cfg_mean = (const mhx_real*)cfg->proposal_mean;
    size_t nvec = 0;
    switch (cfg->proposal_kind) {
    case MHX_PROP_ISO:
        if (!(cfg->proposal_scale > 0.0)) return mhx_fail(MHX_EINVAL, "ISO proposal needs a positive scale");
        break;
    case MHX_PROP_DIAG: nvec = (size_t)d; break;
    case MHX_PROP_DENSE: nvec = (size_t)d * ((size_t)d + 1) / 2; break;
    default: return mhx_fail(MHX_EINVAL, "mhx_rwmh_create: unknown proposal kind %d", cfg->proposal_kind);
    }
    if (nvec && !cfg_vec) return mhx_fail(MHX_EINVAL, "mhx_rwmh_create: proposal_vec is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<mhx_run> r(new mhx_run);
    r->dtype = ctx->dtype;
    r->ctx = ctx; r->target = t; r->kind = RUN_RWMH;
    r->dim = d; r->n = cfg->nchains; r->seed = cfg->seed; r->first_id = cfg->first_chain;
    r->flags = cfg->flags;
    r->prop_kind = cfg->proposal_kind; r->prop_scale = (mhx_real)cfg->proposal_scale;
    HIP_TRY(hipMalloc(&r->d_pvec, (nvec ? nvec : 1) * sizeof(mhx_real)));
    if (nvec) COPY_SYNC(ctx->stream, r->d_pvec, cfg_vec, nvec * sizeof(mhx_real), hipMemcpyHostToDevice);
    int rc = run_alloc_state(r.get());
    if (rc) return rc;
    // drifting random walk: keep mu and 2 L^-1 mu (double arithmetic on the host, rounded once) and use the
    // generic kernel, the only one that evaluates the Hastings ratio
    bool drift = false;
    if (cfg_mean)
        for (int k = 0; k < d; ++k) drift = drift || cfg_mean[k] != MHX_R(0.0);
    if (drift) {
        std::vector<mhx_real> pm(2 * (size_t)d);
        std::vector<double> m((size_t)d);
        size_t off = 0;
        for (int i = 0; i < d; ++i) {
            double acc = (double)cfg_mean[i];
            if (cfg->proposal_kind == MHX_PROP_ISO) m[i] = acc / (double)cfg->proposal_scale;
            else if (cfg->proposal_kind == MHX_PROP_DIAG) m[i] = acc / (double)cfg_vec[i];
            else {
                for (int j = 0; j < i; ++j) acc -= (double)cfg_vec[off + j] * m[j];
                m[i] = acc / (double)cfg_vec[off + i];
                off += (size_t)i + 1;
            }
            pm[i] = cfg_mean[i];
            pm[(size_t)d + i] = (mhx_real)(2.0 * m[i]);
        }
        HIP_TRY(hipMalloc(&r->d_pmean, pm.size() * sizeof(mhx_real)));
        COPY_SYNC(ctx->stream, r->d_pmean, pm.data(), pm.size() * sizeof(mhx_real), hipMemcpyHostToDevice);
    }
    // static (independence) proposal, src/proposal.jl:9-11,66-83: one more mhx_real of state per chain
    if (cfg->flags & MHX_FLAG_STATIC_PROPOSAL) HIP_TRY(hipMalloc(&r->d_qx, (size_t)r->n * sizeof(mhx_real)));
    // walks with a Hastings ratio run on the cooperative kernel (separable target, ISO / DIAG proposal) or on the
    // state-in-HBM kernel; the register kernels know the plain random walk only
    const int walk = (cfg->flags & MHX_FLAG_STATIC_PROPOSAL) ? MHX_WALK_STATIC : (drift ? MHX_WALK_DRIFT : MHX_WALK_PLAIN);

    // the register / cooperative kernels address a [dim+1][nchains] slab with 32-bit byte offsets
    if (((uint64_t)d + 1) * (uint64_t)r->n * (uint64_t)sizeof(mhx_real) >= (1ull << 32)) r->flags |= MHX_FLAG_GENERIC;
    // ---- kernel choice
    r->variant = 0;
    const int tk = t->kind, pk = r->prop_kind;
    // (fp32, round 6: 256 layers, one Philox word per normal -- on the cooperative kernel, the register kernel and MALA's register kernel
    // like fp64; the check at the end of this function refuses a run whose kernel has no ziggurat form)
    if ((cfg->flags & MHX_FLAG_ZIGGURAT) && d >= (1 << 20))      // the retry blocks are numbered (normal index << 8 | attempt) in 28 bits
        return mhx_fail(MHX_EINVAL, "MHX_FLAG_ZIGGURAT: dim must be below 2^20");
    int regmax = pk == MHX_PROP_DENSE ? MHX_REG_MAX_DIM_DENSE : MHX_REG_MAX_DIM;
    if (const char* rm = opt(ctx, "REG_MAX_DIM")) regmax = atoi(rm);      // tuning knob
    const int nblk = (d + 3) / 4;
    const bool separable = tk == MHX_TARGET_ISO_GAUSS || tk == MHX_TARGET_BANANA || tk == MHX_TARGET_FUNNEL;
    // lanes per chain: cfg->reduce_lanes, or (auto) the smallest power of two that (a) keeps a lane's
    // blocks in registers (<= 13 blocks = 104 VGPRs of state) and (b) gives the chip >= 2 waves per SIMD
    // A wave per chain (variant 11): the data-sum target of the README example with FEW chains -- the lanes split the likelihood's
    // terms (reduction shape 64).  Asked for (reduce_lanes = 64), or by default while every chain can have a wave to itself (<= 2 per
    // SIMD) and there are terms to split; the plain random walk, ISO / DIAG proposal, Box-Muller normals.
    if (tk == MHX_TARGET_IID_NORMAL && d == 2 && pk != MHX_PROP_DENSE && walk == MHX_WALK_PLAIN &&
        !(r->flags & (MHX_FLAG_GENERIC | MHX_FLAG_ZIGGURAT)) &&
        (cfg->reduce_lanes == 64 || (cfg->reduce_lanes == 0 && t->nparams >= 8 && chains_of_whole_run(ctx, r->n) <= 2048))) {
        r->reg_fn = pk == MHX_PROP_ISO ? k_rwmh_wave<MHX_PROP_ISO, 4> : k_rwmh_wave<MHX_PROP_DIAG, 4>;
        r->reg_fn8 = pk == MHX_PROP_ISO ? k_rwmh_wave<MHX_PROP_ISO, 8> : k_rwmh_wave<MHX_PROP_DIAG, 8>;
        r->variant = 11;
        r->coop_L = 64;
    } else if (tk == MHX_TARGET_IID_NORMAL && cfg->reduce_lanes > 1) {
        return mhx_fail(MHX_EINVAL, "mhx_rwmh_create: the data-sum target knows reduce_lanes 0, 1 and 64 (a wave per chain: plain walk, "
                        "ISO / DIAG proposal, Box-Muller), got %d", cfg->reduce_lanes);
    }
    int L = 1;
    bool coop_one_lane = false;                       // a walk with a Hastings ratio on ONE lane per chain: still the cooperative body
    if (separable && pk != MHX_PROP_DENSE && !(r->flags & MHX_FLAG_GENERIC) && !(walk && (r->flags & MHX_FLAG_NO_JIT))) {
        if (cfg->reduce_lanes > 1 || (cfg->reduce_lanes == 1 && walk)) {
            L = cfg->reduce_lanes;
            if (L > 64 || (L & (L - 1))) return mhx_fail(MHX_EINVAL, "reduce_lanes must be a power of two <= 64, got %d", L);
        } else if (cfg->reduce_lanes == 0) {
            while (L < 64 && (nblk + L - 1) / L > MHX_COOP_NBL_AUTO) L *= 2;
            // the kernels are bound by VALU throughput, and every block slot of a lane costs a full Philox + Box-Muller
            // round whether it holds dimensions or padding: halve the lanes while that removes > 5 % of the slots and the
            // blocks still fit a lane (C2 in fp64: 4 lanes x 7 = 28 slots for 25 blocks -> 2 lanes x 13 = 26: +4 %)
            while (L > 1 && (nblk + L / 2 - 1) / (L / 2) <= MHX_COOP_NBL_MAX &&
                   (long)(L / 2) * ((nblk + L / 2 - 1) / (L / 2)) * 105 < (long)L * ((nblk + L - 1) / L) * 100) L /= 2;
            while (L < 64 && 2 * L <= nblk && chains_of_whole_run(ctx, r->n) * L / 64 < 2048) L *= 2;
            if ((nblk + L - 1) / L > MHX_COOP_NBL_MAX) L = 1;      // not even a whole wave holds the chain: state in HBM
        }
        // (the ziggurat generator lives in the cooperative body: a chain on ONE lane still runs it there)
        coop_one_lane = (walk || (cfg->flags & MHX_FLAG_ZIGGURAT)) && L == 1 && nblk <= MHX_COOP_NBL_MAX;
    } else if (((tk == MHX_TARGET_CORR_GAUSS) || (tk == MHX_TARGET_ISO_GAUSS && pk == MHX_PROP_DENSE)) &&
               !walk && !(r->flags & (MHX_FLAG_GENERIC | MHX_FLAG_NO_JIT)) && d >= 2 && d <= 4 * std::max(MHX_MFMA_STREAM_MAX_NS, MHX_MFMA_XMEM_MAX_NS) &&
               (cfg->reduce_lanes > 1 || (cfg->reduce_lanes == 0 && d >= 16)) &&
               (mfma_fits(ctx, d, cfg->reduce_lanes, (tk == MHX_TARGET_CORR_GAUSS ? 1 : 0) + (pk == MHX_PROP_DENSE ? 1 : 0)) ||
                mfma_stream_fits(ctx, d, cfg->reduce_lanes) ||
                dense_coop_fits(d, cfg->reduce_lanes > 1 ? cfg->reduce_lanes : dense_coop_lanes(d),
                                (tk == MHX_TARGET_CORR_GAUSS ? 1 : 0) + (pk == MHX_PROP_DENSE ? 1 : 0)))) {
        // a dense factor in play (dense Gaussian target, dense proposal, or both): the cooperative kernel of
        // mhx_rwmh_dense_kernels.h (L lanes per chain, factor images in LDS).  Lanes per chain by default: at most
        // 12.5 rows per lane (measured at 65 536 chains, dense target: d = 32 / 50 / 64 / 100 / 128 run 1.8e10 /
        // 8.5e9 / 6.8e9 / 2.9e9 / 2.0e9 steps/s; the lane-per-chain register kernel 1.4e10 / 7.2e9 / 4.9e9 / - / -;
        // more rows per lane than ~16 spill)
        // ONE factor for all chains makes the row products a GEMM over the chains of a wave: the matrix-core kernel
        // (reduction shape L = 4, bit-identical to the vector kernel in that shape) wherever its images and state fit
        const int nimg = (tk == MHX_TARGET_CORR_GAUSS ? 1 : 0) + (pk == MHX_PROP_DENSE ? 1 : 0);
        if (mfma_fits(ctx, d, cfg->reduce_lanes, nimg)) {
            jit_module* m = nullptr;
            const int waves = mfma_waves(ctx, d);
            const std::string key = "rwmh_mfma/d=" + std::to_string(d) + "/pk=" + std::to_string(pk) + "/tk=" + std::to_string(tk) +
                                    "/w=" + std::to_string(waves);
            rc = jit_compile(ctx, key, jit_source(t, "mhx_rwmh_mfma_kernels.h"),
                             {"MHX_JIT_RWMH_MFMA=1", "MHX_JIT_DIM=" + std::to_string(d), "MHX_JIT_PK=" + std::to_string(pk),
                              "MHX_JIT_TK=" + std::to_string(tk), "MHX_JIT_WAVES=" + std::to_string(waves)}, &m);
            if (rc == MHX_OK) rc = jit_function(m, "mhx_jit_rwmh_mfma", &r->jit_step);
            if (rc == MHX_OK) {
                r->dense_lds = nimg * mfma_image_reals(d) * sizeof(mhx_real);
                if (hipFuncSetAttribute((const void*)r->jit_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->dense_lds) != hipSuccess)
                    rc = mhx_fail(MHX_EHIP, "matrix-core kernel: %zu bytes of LDS refused", r->dense_lds);
            }
            if (rc == MHX_OK) { r->variant = 8; r->coop_L = 4; }
        }
        if (!r->variant && mfma_stream_fits(ctx, d, cfg->reduce_lanes)) {
            // the images do not fit a block's LDS: they stay in global memory (built once, here) and every block walks them
            // through an LDS ring, one tile pair at a time
            jit_module* m = nullptr;
            const int smode = mfma_stream_mode(ctx, d, cfg->reduce_lanes);
            const std::string key = "rwmh_mfma_stream/d=" + std::to_string(d) + "/pk=" + std::to_string(pk) + "/tk=" + std::to_string(tk) +
                                    "/mode=" + std::to_string(smode);
            // the kernel is fully unrolled over its tiles: lift hipcc's size limit for `#pragma unroll` (past it the candidate
            // would be indexed dynamically, i.e. live in scratch)
            rc = jit_compile(ctx, key, jit_source(t, "mhx_rwmh_mfma_kernels.h"),
                             {"MHX_JIT_RWMH_MFMA_STREAM=1", "MHX_JIT_DIM=" + std::to_string(d), "MHX_JIT_PK=" + std::to_string(pk),
                              "MHX_JIT_TK=" + std::to_string(tk), "MHX_JIT_WAVES=1", std::string("MHX_JIT_PAIR=") + (smode == 1 ? "1" : "0"),
                              std::string("MHX_JIT_XMEM=") + (smode == 2 ? "1" : "0")}, &m,
                             {"-mllvm", "-pragma-unroll-threshold=4000000"});
            hipFunction_t fimg = nullptr;
            if (rc == MHX_OK) rc = jit_function(m, "mhx_jit_rwmh_mfma_stream", &r->jit_step);
            if (rc == MHX_OK) rc = jit_function(m, "mhx_jit_mfma_image", &fimg);
            if (rc == MHX_OK) {
                r->dense_lds = mfma_ring_bytes(d, smode == 1);
                if (hipFuncSetAttribute((const void*)r->jit_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->dense_lds) != hipSuccess)
                    rc = mhx_fail(MHX_EHIP, "streamed matrix-core kernel: %zu bytes of LDS refused", r->dense_lds);
            }
            if (rc == MHX_OK) {
                const size_t reals = mfma_image_reals(d);
                HIP_TRY(hipMalloc(&r->d_mfma_img, 2 * reals * sizeof(mhx_real)));
                HIP_TRY(hipMemsetAsync(r->d_mfma_img, 0, 2 * reals * sizeof(mhx_real), ctx->stream));
                const mhx_real* srcs[2] = {tk == MHX_TARGET_CORR_GAUSS ? t->dparams : nullptr, pk == MHX_PROP_DENSE ? r->d_pvec : nullptr};
                for (int im = 0; im < 2 && rc == MHX_OK; ++im) {
                    if (!srcs[im]) continue;
                    mhx_real* dst = r->d_mfma_img + im * reals;
                    void* params[] = {(void*)&srcs[im], &dst};
                    rc = launch_module(fimg, 1, 256, ctx->stream, params);
                }
                if (rc == MHX_OK) HIP_TRY(hipStreamSynchronize(ctx->stream));
            }
            if (rc == MHX_OK) { r->variant = 8; r->coop_L = 4; r->mfma_stream = true; }
            else { r->jit_step = nullptr; }
        }
        // reduce_lanes = 4 asked for explicitly and a matrix-core kernel was tried but did not come up: the error is the
        // caller's, like every other explicit reduce_lanes (no silent change of the summation shape)
        if (!r->variant && rc != MHX_OK && cfg->reduce_lanes == 4) return rc;
        L = cfg->reduce_lanes > 1 ? cfg->reduce_lanes : dense_coop_lanes(d);
        if (r->variant == 8 || !dense_coop_fits(d, L, nimg)) {
            L = 1;
        } else {
        if (L > 64 || (L & (L - 1)) || L > d) return mhx_fail(MHX_EINVAL, "reduce_lanes must be a power of two <= min(64, dim), got %d", L);
        jit_module* m = nullptr;
        const std::string key = "rwmh_dense/d=" + std::to_string(d) + "/l=" + std::to_string(L) + "/pk=" + std::to_string(pk) +
                                "/tk=" + std::to_string(tk);
        rc = jit_compile(ctx, key, jit_source(t, "mhx_rwmh_dense_kernels.h"),
                         {"MHX_JIT_RWMH_DENSE=1", "MHX_JIT_DIM=" + std::to_string(d), "MHX_JIT_L=" + std::to_string(L),
                          "MHX_JIT_PK=" + std::to_string(pk), "MHX_JIT_TK=" + std::to_string(tk)}, &m);
        if (rc == MHX_OK) rc = jit_function(m, "mhx_jit_rwmh_dense", &r->jit_step);
        if (rc == MHX_OK) {
            r->dense_lds = dense_coop_lds_bytes(d, L, (tk == MHX_TARGET_CORR_GAUSS ? 1 : 0) + (pk == MHX_PROP_DENSE ? 1 : 0));
            if (hipFuncSetAttribute((const void*)r->jit_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->dense_lds) != hipSuccess)
                rc = mhx_fail(MHX_EHIP, "dense cooperative kernel: %zu bytes of LDS refused", r->dense_lds);
        }
        if (rc == MHX_OK) { r->variant = 5; r->coop_L = L; }
        else if (cfg->reduce_lanes > 1) return rc;
        L = 1;                                                        // not the separable cooperative path below
        }
    } else if (cfg->reduce_lanes > 1 && r->variant != 11) {
        return mhx_fail(MHX_EINVAL, "reduce_lanes > 1 needs a separable catalogue target or the dense Gaussian target "
                                "(dim <= 128, JIT), and an ISO/DIAG proposal");
    }
    if (L > 1 || coop_one_lane) {
        const int NBL = (nblk + L - 1) / L;
        if (NBL > MHX_COOP_NBL_MAX) return mhx_fail(MHX_EINVAL, "reduce_lanes=%d leaves %d blocks per lane (max %d)", L, NBL, MHX_COOP_NBL_MAX);
        // tuning knobs: MHX_NO_PREBUILT=1 specialises with hiprtc even where a pre-built kernel exists; MHX_COOP_WAVES=w
        // overrides the waves-per-SIMD launch bound of the hiprtc kernel (register budget 512 / w)
        const char* no_prebuilt = opt(ctx, "NO_PREBUILT");
        const char* waves_env = opt(ctx, "COOP_WAVES");
        const int waves_override = waves_env ? atoi(waves_env) : 0;
        const bool zig = (cfg->flags & MHX_FLAG_ZIGGURAT) != 0;
        if (zig && MHX_ZIG_LDS_BYTES(NBL) > MHX_LDS_PER_BLOCK)
            return mhx_fail(MHX_EINVAL, "MHX_FLAG_ZIGGURAT: %d blocks per lane need %d bytes of LDS per block (limit %d); use more lanes per chain",
                            NBL, (int)MHX_ZIG_LDS_BYTES(NBL), (int)MHX_LDS_PER_BLOCK);
        if (!(no_prebuilt && atoi(no_prebuilt)) && !waves_override && !walk)
            for (const auto& pb : k_prebuilt_coop)
                if (pb.L == L && pb.NBL == NBL && pb.TK == tk && pb.PK == pk && pb.gen == (zig ? MHX_GEN_ZIGGURAT : MHX_GEN_BOX_MULLER)) {
                    r->reg_fn = pb.fn; r->reg_fn_mom = pb.fn_mom; r->variant = 3; r->normal_gen = pb.gen;
                    if (zig) {
                        r->coop_lds = MHX_ZIG_LDS_BYTES(NBL);
                        for (auto f : {pb.fn, pb.fn_mom})
                            if (f && hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->coop_lds) != hipSuccess)
                                return mhx_fail(MHX_EHIP, "cooperative kernel (ziggurat): %zu bytes of LDS refused", r->coop_lds);
                    }
                }
        if (!r->variant && !(r->flags & MHX_FLAG_NO_JIT)) {
            jit_module* m = nullptr;
            const std::string key = "rwmh_coop/l=" + std::to_string(L) + "/nbl=" + std::to_string(NBL) + "/tk=" +
                                    std::to_string(tk) + "/pk=" + std::to_string(pk) + "/w=" + std::to_string(waves_override) +
                                    "/walk=" + std::to_string(walk) + "/gen=" + std::to_string(zig ? 1 : 0)
#ifdef MHX_TOOLS_BUILD
                                    + (opt(ctx, "ZIG_PROBE") ? std::string("/zp=") + opt(ctx, "ZIG_PROBE") : std::string()) +
                                    (opt(ctx, "ZIG_FORCE_FAIL") ? std::string("/zff=") + opt(ctx, "ZIG_FORCE_FAIL") : std::string())
#endif
                                    ;
            std::vector<std::string> defs = {"MHX_JIT_RWMH_COOP=1", "MHX_JIT_L=" + std::to_string(L), "MHX_JIT_NBL=" + std::to_string(NBL),
                                             "MHX_JIT_TK=" + std::to_string(tk), "MHX_JIT_PK=" + std::to_string(pk), "MHX_JIT_MOM=0",
                                             "MHX_JIT_WALK=" + std::to_string(walk), std::string("MHX_JIT_GEN=") + (zig ? "1" : "0")};
            if (waves_override > 0) defs.push_back("MHX_JIT_WAVES=" + std::to_string(waves_override));
#ifdef MHX_TOOLS_BUILD
            if (const char* zf = opt(ctx, "ZIG_FORCE_FAIL"))                       // test hook: see mhx_rwmh_kernels.h (the chains stay valid)
                if (atoi(zf) > 0) defs.push_back(std::string("MHX_ZIG_FORCE_FAIL=") + std::to_string(atoi(zf)));
            if (const char* zp = opt(ctx, "ZIG_PROBE")) {                          // timing probe: the chains of such a run are NOT valid
                defs.push_back(std::string("MHX_ZIG_PROBE=") + zp);
                fprintf(stderr, "mhx (tools build): ZIG_PROBE=%s -- timing probe, this run's chains are NOT valid (stats.tainted = 1)\n", zp);
            }
#endif
            rc = jit_compile(ctx, key + "/mom=0", jit_source(t, "mhx_rwmh_kernels.h"), defs, &m);
            if (rc == MHX_OK) rc = jit_function(m, "mhx_jit_rwmh_coop", &r->jit_step);
            if (rc == MHX_OK && zig) {
                r->coop_lds = MHX_ZIG_LDS_BYTES(NBL);
                if (hipFuncSetAttribute((const void*)r->jit_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->coop_lds) != hipSuccess)
                    rc = mhx_fail(MHX_EHIP, "cooperative kernel (ziggurat): %zu bytes of LDS refused", r->coop_lds);
            }
            if (rc == MHX_OK) { r->variant = 4; r->normal_gen = zig ? MHX_GEN_ZIGGURAT : MHX_GEN_BOX_MULLER; }
            else if (cfg->reduce_lanes > 1 || zig) return rc;      // the caller asked for this shape explicitly
        }
        if (r->variant) {
            // one or two chains per wave: rows of the [dim][chains] arrays move through LDS, CB = 4 * (64 / L) chains at a time
            // (mhx_rwmh_coop_body); the ziggurat's slab holds that buffer, the Box-Muller kernels get one of their own
            if (L >= 32) {
                const size_t need = (size_t)d * (size_t)(4 * (64 / L)) * sizeof(mhx_real);
                if (zig) r->coop_tr = 1;
                else if (need <= 65536) { r->coop_lds = (need + 15) & ~(size_t)15; r->coop_tr = 1; }
            }
            r->coop_L = L;
            r->coop_key = "rwmh_coop/l=" + std::to_string(L) + "/nbl=" + std::to_string(NBL) + "/tk=" + std::to_string(tk) +
                          "/pk=" + std::to_string(pk) + "/walk=" + std::to_string(walk) + "/gen=" + std::to_string(zig ? 1 : 0);
            r->coop_defs = {"MHX_JIT_RWMH_COOP=1", "MHX_JIT_L=" + std::to_string(L), "MHX_JIT_NBL=" + std::to_string(NBL),
                            "MHX_JIT_TK=" + std::to_string(tk), "MHX_JIT_PK=" + std::to_string(pk), "MHX_JIT_MOM=1",
                            "MHX_JIT_WALK=" + std::to_string(walk), std::string("MHX_JIT_GEN=") + (zig ? "1" : "0")};
        } else if (cfg->reduce_lanes > 1) return mhx_fail(MHX_EINVAL, "reduce_lanes=%d: no pre-built kernel and JIT disabled", L);
    }
    if (!r->variant && walk) {
        // no cooperative kernel for this target / proposal family: the state-in-HBM kernel evaluates the ratio
        if (cfg->reduce_lanes > 1) return mhx_fail(MHX_EINVAL, "reduce_lanes > 1 with a drifting / static proposal needs a separable catalogue target and an ISO / DIAG proposal");
        r->flags |= MHX_FLAG_GENERIC;
    }
    if (!r->variant && !(r->flags & MHX_FLAG_GENERIC)) {
        if (tk != MHX_TARGET_USER && !(cfg->flags & MHX_FLAG_ZIGGURAT))
            for (const auto& pb : k_prebuilt_reg)
                if (pb.D == d && pb.TK == tk && pb.PK == pk) { r->reg_fn = pb.fn; r->variant = 1; }
        // The candidate must be whole in a lane's registers (an arbitrary log-density reads all of it); the state need not be: above
        // MHX_REG_XR_ALL dimensions only its first MHX_REG_XR_CAP coordinates stay in registers, the tail lives in LDS
        // (mhx_rwmh_reg_body<..., XR>).  That carries the kernel to twice its old dimension limit -- an fp64 user log-density at
        // d = 100 ran the state-in-HBM kernel at 4.9e8 steps/s, now 2.4e9 -- and is faster below it too (fp64 d = 80: 2.8 -> 3.3e9,
        // fp32 d = 160: 1.6 -> 2.8e9): tools/reg_xr_sweep.sh, profiles/r04v_reg_xr_sweep.log.  ISO / DIAG proposals.
        // -amdgpu-unroll-threshold-private: hipcc unrolls a loop over a private array only up to a cost of 2 700; a user's
        // `for (k < d)` loop above that (d = 100 with one division per term) stays a loop, its array lands in scratch memory and
        // the kernel runs 3 x slower (the cliff between d = 96 and d = 100 of the first measurement).
        const bool split = d > MHX_REG_XR_ALL && d <= 2 * MHX_REG_MAX_DIM && pk != MHX_PROP_DENSE;
        if (!opt(ctx, "REG_MAX_DIM") && split) regmax = 2 * MHX_REG_MAX_DIM;
        if (!r->variant && !(r->flags & MHX_FLAG_NO_JIT) && d <= regmax &&
            !(tk == MHX_TARGET_CORR_GAUSS && d > (MHX_REAL64 ? 32 : 64)) && !(tk == MHX_TARGET_IID_NORMAL && t->nparams > 4096)) {
            jit_module* m = nullptr;
            int xr = split ? MHX_REG_XR_CAP : d;
            if (const char* xe = opt(ctx, "REG_XR")) xr = std::max(0, std::min(d, atoi(xe)));      // tuning knobs
            const char* ut = opt(ctx, "REG_UNROLL");
            std::vector<std::string> xo;
            if (!ut || atoi(ut) > 0) xo = {"-mllvm", std::string("-amdgpu-unroll-threshold-private=") + (ut ? ut : "100000")};
            // MHX_FLAG_ZIGGURAT (ISO / DIAG proposal; both widths since round 6): the register kernel's ziggurat form (mhx_rwmh_reg_zig_body) -- any target,
            // a user's HIP source included
            const bool zig_reg = (cfg->flags & MHX_FLAG_ZIGGURAT) && pk != MHX_PROP_DENSE;
            // waves per SIMD the kernel's register budget is cut for (REG_WAVES; measured on the fp32 ziggurat form: no gain from 2)
            int rwaves = 1;
            if (const char* we = opt(ctx, "REG_WAVES")) rwaves = std::max(1, std::min(4, atoi(we)));
            // fp32: the step's normals through an LDS slab instead of the hand-back walk, where four one-wave blocks of it fit a CU
            // (option REG_ZSLAB = 0 / 1 overrides)
            bool zslab = !MHX_REAL64 && zig_reg && 4 * (MHX_REG_ZIG_LDS_BYTES(d, xr) + MHX_REG_ZIG_SLAB_BYTES(d)) <= 163840;
            if (const char* ze = opt(ctx, "REG_ZSLAB")) zslab = !MHX_REAL64 && zig_reg && atoi(ze) != 0;
            const std::string key = "rwmh_reg/d=" + std::to_string(d) + "/tk=" + std::to_string(tk) + "/pk=" +
                                    std::to_string(pk) + "/xr=" + std::to_string(xr) + (ut ? std::string("/ut=") + ut : std::string()) +
                                    (zig_reg ? "/gen=1" : "") + (zslab ? "/slab" : "") + (rwaves > 1 ? "/w=" + std::to_string(rwaves) : std::string()) +
                                    "/" + t->user_key;
            std::vector<std::string> rdefs = {"MHX_JIT_RWMH_REG=1", "MHX_JIT_DIM=" + std::to_string(d),
                                              "MHX_JIT_TK=" + std::to_string(tk), "MHX_JIT_PK=" + std::to_string(pk), "MHX_JIT_XR=" + std::to_string(xr)};
            if (zig_reg) rdefs.push_back("MHX_JIT_GEN=1");
            if (zslab) rdefs.push_back("MHX_JIT_ZSLAB=1");
            if (rwaves > 1) rdefs.push_back("MHX_JIT_REG_WAVES=" + std::to_string(rwaves));
            rc = jit_compile(ctx, key, jit_source(t, "mhx_rwmh_kernels.h"), rdefs, &m, xo);
            if (rc == MHX_OK) rc = jit_function(m, "mhx_jit_rwmh_reg", &r->jit_step);
            r->reg_lds = (size_t)(d - xr) * 64 * sizeof(mhx_real);
            if (zig_reg) r->reg_lds = MHX_REG_ZIG_LDS_BYTES(d, xr) + (zslab ? MHX_REG_ZIG_SLAB_BYTES(d) : 0);
            if (rc == MHX_OK && r->reg_lds > 65536 &&
                hipFuncSetAttribute((const void*)r->jit_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->reg_lds) != hipSuccess)
                rc = mhx_fail(MHX_EHIP, "register kernel: %zu bytes of LDS refused", r->reg_lds);
            if (rc == MHX_OK) { r->variant = 2; if (zig_reg) r->normal_gen = MHX_GEN_ZIGGURAT; }
            else if (tk == MHX_TARGET_USER) return rc;      // no pre-built kernel can run a user source
            // built-in target: the generic kernel below computes the same chain; keep the message
        }
    }
    if (tk == MHX_TARGET_USER) {
        jit_module* m = nullptr;
        if ((rc = jit_generic_rwmh(t, &m))) return rc;
        if ((rc = jit_function(m, "mhx_jit_rwmh_init", &r->jit_init))) return rc;
        if (r->variant == 0 && (rc = jit_function(m, "mhx_jit_rwmh_generic", &r->jit_step))) return rc;
    }
    if ((cfg->flags & MHX_FLAG_ZIGGURAT) && r->normal_gen != MHX_GEN_ZIGGURAT)
        return mhx_fail(MHX_EINVAL, "MHX_FLAG_ZIGGURAT: this run's kernel (variant %d) has no ziggurat form -- it exists on the cooperative "
                                    "kernel (separable catalogue target) and on the register kernel (any target within its "
                                    "dimension limit), ISO / DIAG proposal, JIT allowed", r->variant);
    // candidate scratch of the state-in-HBM kernel; a static proposal whitens the state into it whatever kernel steps the chain
    if (r->variant == 0 || r->d_qx) HIP_TRY(hipMalloc(&r->d_ybuf, (size_t)d * (size_t)r->n * sizeof(mhx_real)));
    *out = r.release();
    return MHX_OK;
}

// ---- emcee / ram creation, init and stepping live in their own sections below
static int emcee_init(mhx_run* r, const mhx_real* init);
static int emcee_sync_state(mhx_run* r, int to_abi);
static int emcee_advance(mhx_run* r, uint64_t nsteps, uint32_t save_next, int save_slot, int thinning);
static int ram_init(mhx_run* r, const mhx_real* init);
static int ram_advance(mhx_run* r, uint64_t nsteps, uint64_t n_adapt, uint32_t save_next, int save_slot, int thinning);
static int ram_prepare_step_stats(mhx_run* r, const mhx_schedule* s, uint64_t n_adapt);
static int mala_init(mhx_run* r, const mhx_real* init);
static int mala_eval_state(mhx_run* r, int reset_counts);
static int mala_advance(mhx_run* r, uint64_t nsteps, uint32_t save_next, int save_slot, int thinning);

// A caller-supplied state enters an RWMH chain with -0.0 coordinates as +0.0 (x + 0.0: every other value, NaN and the
// infinities included, is unchanged; -0.0 == +0.0, so `chain[1].params == initial_params` of test/runtests.jl:203-213 holds).
// The chain itself never produces a -0.0 (a rounded sum is -0 only if both terms are), and the fp64 plain-walk kernel relies on
// that: it re-forms a REJECTED state as fma(0, n, x), which is x exactly for every x but -0.0.  The oracle does the same
// (orc_rwmh, `init`).
__global__ void k_plus_zero(mhx_real* x, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] + MHX_R(0.0);
}

static int rwmh_canonical_zero(mhx_run* r)
{
    const size_t nx = (size_t)r->dim * (size_t)r->n;
    hipLaunchKernelGGL(k_plus_zero, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, r->ctx->stream, r->d_x, nx);
    HIP_TRY(hipGetLastError());
    return MHX_OK;
}

static int rwmh_init(mhx_run* r, const mhx_real* init)
{
    mhx_ctx* ctx = r->ctx;
    const size_t nx = (size_t)r->dim * (size_t)r->n;
    if (init) {
        HIP_TRY(hipMemcpyAsync(r->d_x, init, nx * sizeof(mhx_real), hipMemcpyHostToDevice, ctx->stream));
        int rcz = rwmh_canonical_zero(r);
        if (rcz) return rcz;
    }
    mhx_rwmh_args a = rwmh_args(r);
    const mhx_real* tp = r->target->dparams;
    const mhx_real* pv = r->d_pvec;
    int draw = init ? 0 : 1;
    const unsigned grid = (unsigned)((r->n + 255) / 256);
    if (r->target->kind == MHX_TARGET_USER) {
        void* params[] = {&a, &tp, &pv, &draw};
        int rc = launch_module(r->jit_init, grid, 256, ctx->stream, params);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_rwmh_init, dim3(grid), dim3(256), 0, ctx->stream, a, tp, pv, draw);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return rwmh_whiten(r);
}

#define MHX_MAX_STEPS_PER_LAUNCH 65536ull

static int rwmh_advance(mhx_run* r, uint64_t nsteps, uint32_t save_next, int save_slot, int thinning)
{
    mhx_ctx* ctx = r->ctx;
    const mhx_real* tp = r->target->dparams;
    const mhx_real* pv = r->d_pvec;
    uint64_t done = 0;
    while (done < nsteps) {
        const uint64_t chunk = std::min<uint64_t>(nsteps - done, MHX_MAX_STEPS_PER_LAUNCH);
        mhx_rwmh_args a = rwmh_args(r);
        a.step0 = (uint32_t)(r->tau + 1);
        a.nsteps = (int)chunk;
        a.save_next = save_next;
        a.save_slot = save_slot;
        a.thinning = thinning;
        if (r->variant == 3 || r->variant == 4) {
            const long threads = (((long)r->n + (64 / r->coop_L) - 1) / (64 / r->coop_L)) * 64;   // whole waves
            unsigned grid = (unsigned)((threads + 255) / 256);
            if (r->coop_L >= 32) grid = (grid + 7u) & ~7u;          // the kernel's XCD-aware block -> chain map wants whole rounds of 8
            if (r->moments_mode && r->reg_fn_mom) {
                hipLaunchKernelGGL(r->reg_fn_mom, dim3(grid), dim3(256), r->coop_lds, ctx->stream, a, tp, pv);
            } else if (r->moments_mode) {
                void* params[] = {&a, &tp, &pv};
                HIP_TRY(hipModuleLaunchKernel(r->jit_step_mom, grid, 1, 1, 256, 1, 1, (unsigned)r->coop_lds, ctx->stream, params, nullptr));
            } else if (r->variant == 3) {
                hipLaunchKernelGGL(r->reg_fn, dim3(grid), dim3(256), r->coop_lds, ctx->stream, a, tp, pv);
            } else {
                void* params[] = {&a, &tp, &pv};
                HIP_TRY(hipModuleLaunchKernel(r->jit_step, grid, 1, 1, 256, 1, 1, (unsigned)r->coop_lds, ctx->stream, params, nullptr));
            }
        } else if (r->variant == 8 && r->mfma_stream) {
            const unsigned grid = (unsigned)(((long)r->n + 16 * MHX_MFMA_WAVES - 1) / (16 * MHX_MFMA_WAVES));
            mhx_real* gA = r->d_mfma_img;
            mhx_real* gL = r->d_mfma_img + mfma_image_reals(r->dim);
            void* params[] = {&a, &tp, &pv, &gA, &gL};
            HIP_TRY(hipModuleLaunchKernel(r->jit_step, grid, 1, 1, 64 * MHX_MFMA_WAVES, 1, 1, (unsigned)r->dense_lds,
                                          ctx->stream, params, nullptr));
        } else if (r->variant == 8) {
            const unsigned grid = (unsigned)(((long)r->n + 16 * MHX_MFMA_WAVES - 1) / (16 * MHX_MFMA_WAVES));
            void* params[] = {&a, &tp, &pv};
            HIP_TRY(hipModuleLaunchKernel(r->jit_step, grid, 1, 1, 64 * MHX_MFMA_WAVES, 1, 1, (unsigned)r->dense_lds,
                                          ctx->stream, params, nullptr));
        } else if (r->variant == 5) {
            const long per_block = (64 / r->coop_L) * MHX_EMCEE_COOP_WAVES;          // chains per block
            const unsigned grid = (unsigned)(((long)r->n + per_block - 1) / per_block);
            void* params[] = {&a, &tp, &pv};
            HIP_TRY(hipModuleLaunchKernel(r->jit_step, grid, 1, 1, 64 * MHX_EMCEE_COOP_WAVES, 1, 1, (unsigned)r->dense_lds,
                                          ctx->stream, params, nullptr));
        } else if (r->variant == 11) {                           // a wave per chain; candidates per round: by the last call's acceptance
            const char* wk = opt(r->ctx, "WAVE_K");
            const int k = wk ? atoi(wk) : r->wave_k;
            hipLaunchKernelGGL(k == 8 ? r->reg_fn8 : r->reg_fn, dim3((unsigned)r->n), dim3(64), 0, ctx->stream, a, tp, pv);
        } else if (r->variant == 1) {
            const unsigned grid = (unsigned)((r->n + 63) / 64);
            hipLaunchKernelGGL(r->reg_fn, dim3(grid), dim3(64), 0, ctx->stream, a, tp, pv);
        } else if (r->variant == 2) {
            const unsigned grid = (unsigned)((r->n + 63) / 64);
            void* params[] = {&a, &tp, &pv};
            HIP_TRY(hipModuleLaunchKernel(r->jit_step, grid, 1, 1, 64, 1, 1, (unsigned)r->reg_lds, ctx->stream, params, nullptr));
        } else if (r->target->kind == MHX_TARGET_USER) {
            const unsigned grid = (unsigned)((r->n + 255) / 256);
            void* params[] = {&a, &tp, &pv};
            int rc = launch_module(r->jit_step, grid, 256, ctx->stream, params);
            if (rc) return rc;
        } else {
            const unsigned grid = (unsigned)((r->n + 255) / 256);
            hipLaunchKernelGGL(k_rwmh_generic, dim3(grid), dim3(256), 0, ctx->stream, a, tp, pv);
        }
        HIP_TRY(hipGetLastError());
        r->stats.launches++;
        // advance the save cursor past this chunk
        if (save_next != MHX_NO_SAVE) {
            const uint64_t last = r->tau + chunk;
            if ((uint64_t)save_next <= last) {
                const uint64_t k = (last - save_next) / (uint64_t)thinning + 1;
                save_next += (uint32_t)(k * (uint64_t)thinning);
                save_slot += (int)k;
                if (r->moments_mode) r->mom_n += k;
            }
        }
        r->tau += chunk;
        done += chunk;
    }
    return MHX_OK;
}

int api_run_init(mhx_run* r, const mhx_real* initial_params)
{
    if (!r) return mhx_fail(MHX_EINVAL, "mhx_run_init: run is NULL");
    HIP_TRY(hipSetDevice(r->ctx->device));
    int rc;
    switch (r->kind) {
    case RUN_RWMH: rc = rwmh_init(r, initial_params); break;
    case RUN_EMCEE: rc = emcee_init(r, initial_params); break;
    case RUN_MALA: rc = mala_init(r, initial_params); break;
    default: rc = ram_init(r, initial_params); break;
    }
    if (rc) return rc;
    r->initialised = true;
    r->tau = 0;
    HIP_TRY(hipMemsetAsync(r->d_acc_total, 0, sizeof(unsigned long long), r->ctx->stream));
    return MHX_OK;
}

// [upstream AbstractMCMC.mcmcsample, restated from memory]: see DESIGN.md section 5
static void schedule_counts(const mhx_schedule* s, uint64_t* n_transitions, uint64_t* n_adapt)
{
    const int64_t N = s->n_samples, di = s->discard_initial, th = s->thinning, nw = s->num_warmup;
    const int64_t dfw = nw < di ? nw : di;
    const int64_t kfw = nw - dfw;
    const int64_t k = kfw < N ? kfw : N;
    *n_transitions = (uint64_t)(di + (N - 1) * th);
    *n_adapt = (uint64_t)(dfw + (k >= 2 ? (k - 1) * th : 0));
}

static int ensure_buffer(void** p, size_t* cap, size_t need)
{
    if (*cap >= need && *p) return MHX_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) return mhx_fail(MHX_ENOMEM, "sample buffer of %zu bytes: %s", need, hipGetErrorString(e));
    *cap = need;
    return MHX_OK;
}

// accepted proposals of all chains so far
static int run_total_accepts(mhx_run* r, unsigned long long* out)
{
    mhx_ctx* ctx = r->ctx;
    if (r->kind == RUN_EMCEE) {            // per-walker counts summed on demand (no atomics in the half-step kernels)
        HIP_TRY(hipMemsetAsync(r->d_acc_total, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k_sum_u32, dim3(64), dim3(256), 0, ctx->stream, (const unsigned*)r->d_acc, (long)r->n, r->d_acc_total);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    COPY_SYNC(r->ctx->stream, out, r->d_acc_total, sizeof *out, hipMemcpyDeviceToHost);
    return MHX_OK;
}

int api_run_sample(mhx_run* r, const mhx_schedule* s, int save_samples)
{
    if (!r || !s) return mhx_fail(MHX_EINVAL, "mhx_run_sample: NULL argument");
    if (!r->initialised) return mhx_fail(MHX_ESTATE, "mhx_run_sample: call mhx_run_init first");
    if (s->n_samples < 1 || s->thinning < 1 || s->discard_initial < 0 || s->num_warmup < 0)
        return mhx_fail(MHX_EINVAL, "mhx_run_sample: bad schedule (N=%d discard=%d thinning=%d warmup=%d)",
                    s->n_samples, s->discard_initial, s->thinning, s->num_warmup);
    mhx_ctx* ctx = r->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    uint64_t nT = 0, nA = 0;
    schedule_counts(s, &nT, &nA);
    if (r->tau + nT >= 0xffffffffull) return mhx_fail(MHX_EINVAL, "step counter would exceed 2^32-1; start a new seed");
    const auto t0 = std::chrono::steady_clock::now();
    r->stats = mhx_stats{};
    r->stats.kernel_variant = r->variant;
    r->stats.reduce_lanes = r->coop_L;
    r->stats.dtype = r->dtype;
    r->stats.normal_gen = r->normal_gen;
    r->stats.factor_band = r->kind == RUN_EMCEE ? r->emcee_band : -1;
    auto total_accepts = [&](unsigned long long* out) -> int { return run_total_accepts(r, out); };
    unsigned long long acc_before = 0;
    { int rc0 = total_accepts(&acc_before); if (rc0) return rc0; }

    uint32_t save_next = MHX_NO_SAVE;
    int save_slot = 0;
    r->n_saved = 0;
    r->moments_mode = false;
    r->rec_n = 0;
    r->watch_count = 0;              // (with rec_n: a call that keeps nothing must not leave the previous call's watched factors behind)
    r->rec_loga_view = nullptr;
    if (save_samples == MHX_SAVE_MOMENTS) {
        // running moments instead of a sample tensor
        if (r->kind != RUN_RWMH || (r->variant != 0 && r->variant != 3 && r->variant != 4))
            return mhx_fail(MHX_EINVAL, "running moments need an RWMH run on the cooperative or the generic kernel "
                                    "(separable target, or MHX_FLAG_GENERIC); this run uses kernel variant %d", r->variant);
        if (((r->variant == 3 && !r->reg_fn_mom) || r->variant == 4) && !r->jit_step_mom) {
            jit_module* m = nullptr;
            int rcj = jit_compile(ctx, r->coop_key + "/mom=1", jit_source(r->target, "mhx_rwmh_kernels.h"), r->coop_defs, &m);
            if (rcj == MHX_OK) rcj = jit_function(m, "mhx_jit_rwmh_coop", &r->jit_step_mom);
            if (rcj == MHX_OK && r->coop_lds &&
                hipFuncSetAttribute((const void*)r->jit_step_mom, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->coop_lds) != hipSuccess)
                rcj = mhx_fail(MHX_EHIP, "cooperative kernel (ziggurat, moments): %zu bytes of LDS refused", r->coop_lds);
            if (rcj) return rcj;
        }
        const size_t bytes = ((size_t)r->dim + 1) * (size_t)r->n * sizeof(mhx_real);
        int rc = ensure_buffer((void**)&r->d_mom_mean, &r->mom_cap, bytes);
        if (rc) return rc;
        rc = ensure_buffer((void**)&r->d_mom_m2, &r->mom_cap2, bytes);
        if (rc) return rc;
        r->moments_mode = true;
        r->mom_n = 0;
        r->n_saved = s->n_samples;
        if (s->discard_initial == 0) {
            const unsigned grid = (unsigned)((r->n + 255) / 256);
            hipLaunchKernelGGL(k_moments_first, dim3(grid), dim3(256), 0, ctx->stream, r->d_x, r->d_lp, r->d_mom_mean,
                               r->d_mom_m2, r->n, (long)r->n, r->dim);
            HIP_TRY(hipGetLastError());
            r->mom_n = 1;
            save_next = s->n_samples > 1 ? (uint32_t)(r->tau + (uint64_t)s->thinning) : MHX_NO_SAVE;
        } else {
            save_next = (uint32_t)(r->tau + (uint64_t)s->discard_initial);
        }
    } else if (save_samples) {
        const size_t N = (size_t)s->n_samples, n = (size_t)r->n, d1 = (size_t)r->dim + 1;
        int rc = ensure_buffer((void**)&r->d_samples, &r->samples_cap, N * d1 * n * sizeof(mhx_real));
        if (rc) return rc;
        rc = ensure_buffer((void**)&r->d_accepted, &r->accepted_cap, N * n);
        if (rc) return rc;
        r->n_saved = s->n_samples;
        if (r->kind == RUN_RAM) { rc = ram_prepare_step_stats(r, s, nA); if (rc) return rc; }
        if (s->discard_initial == 0) {
            // sample 1 is the current state itself (test/runtests.jl:203-213: chain[1].params == initial_params)
            // an ensemble on the register / cooperative kernels lives in its walker-major copy: bring d_x up to date
            if (r->kind == RUN_EMCEE) { int rcs = emcee_sync_state(r, 1); if (rcs) return rcs; }
            const unsigned grid = (unsigned)((r->n + 255) / 256);
            hipLaunchKernelGGL(k_record_state, dim3(grid), dim3(256), 0, ctx->stream, r->d_x, r->d_lp, r->d_last,
                               r->d_samples, r->d_accepted, r->n, (long)r->n, r->dim, 0L);
            HIP_TRY(hipGetLastError());
            save_slot = 1;
            save_next = s->n_samples > 1 ? (uint32_t)(r->tau + (uint64_t)s->thinning) : MHX_NO_SAVE;
        } else {
            save_slot = 0;
            save_next = (uint32_t)(r->tau + (uint64_t)s->discard_initial);
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    int rc = MHX_OK;
    if (nT) {
        switch (r->kind) {
        case RUN_RWMH: rc = rwmh_advance(r, nT, save_next, save_slot, s->thinning); break;
        case RUN_EMCEE: rc = emcee_advance(r, nT, save_next, save_slot, s->thinning); break;
        case RUN_MALA: rc = mala_advance(r, nT, save_next, save_slot, s->thinning); break;
        default: rc = ram_advance(r, nT, nA, save_next, save_slot, s->thinning); break;
        }
    }
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    unsigned long long acc_after = 0;
    { int rc1 = total_accepts(&acc_after); if (rc1) return rc1; }
    r->stats.kernel_ms = ms;
    r->stats.transitions = nT * (uint64_t)r->n;
    r->stats.accepted = acc_after - acc_before;
    r->stats.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (r->variant == 11 && r->stats.transitions >= 64)          // (the same chain for either K: only the speed depends on it)
        r->wave_k = 8.0 * (double)r->stats.accepted < (double)r->stats.transitions ? 8 : 4;
    return MHX_OK;
}

int api_run_get_samples(mhx_run* r, mhx_real* samples, uint8_t* accepted)
{
    if (!r) return mhx_fail(MHX_EINVAL, "mhx_run_get_samples: run is NULL");
    if (r->n_saved <= 0 || r->moments_mode)
        return mhx_fail(MHX_ESTATE, "mhx_run_get_samples: the last mhx_run_sample kept no sample tensor");
    HIP_TRY(hipSetDevice(r->ctx->device));
    const size_t N = (size_t)r->n_saved, n = (size_t)r->n, d1 = (size_t)r->dim + 1;
    if (samples) COPY_SYNC(r->ctx->stream, samples, r->d_samples, N * d1 * n * sizeof(mhx_real), hipMemcpyDeviceToHost);
    if (accepted) COPY_SYNC(r->ctx->stream, accepted, r->d_accepted, N * n, hipMemcpyDeviceToHost);
    return MHX_OK;
}

#include "mhx_api_host.inc"

int api_run_device_samples(mhx_run* r, void** samples, void** accepted, int64_t* n_samples)
{
    if (!r) return mhx_fail(MHX_EINVAL, "mhx_run_device_samples: run is NULL");
    if (samples) *samples = r->d_samples;
    if (accepted) *accepted = r->d_accepted;
    if (n_samples) *n_samples = r->n_saved;
    return MHX_OK;
}

int api_run_get_state(mhx_run* r, mhx_real* x, mhx_real* lp, uint32_t* accept_counts)
{
    if (!r) return mhx_fail(MHX_EINVAL, "mhx_run_get_state: run is NULL");
    if (!r->initialised) return mhx_fail(MHX_ESTATE, "mhx_run_get_state: run is not initialised");
    HIP_TRY(hipSetDevice(r->ctx->device));
    const size_t n = (size_t)r->n, d = (size_t)r->dim;
    if (x && r->kind == RUN_EMCEE) { int rc = emcee_sync_state(r, 1); if (rc) return rc; }
    if (x) COPY_SYNC(r->ctx->stream, x, r->d_x, d * n * sizeof(mhx_real), hipMemcpyDeviceToHost);
    if (lp) COPY_SYNC(r->ctx->stream, lp, r->d_lp, n * sizeof(mhx_real), hipMemcpyDeviceToHost);
    if (accept_counts) COPY_SYNC(r->ctx->stream, accept_counts, r->d_acc, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    return MHX_OK;
}

int api_run_set_state(mhx_run* r, const mhx_real* x)
{
    if (!r || !x) return mhx_fail(MHX_EINVAL, "mhx_run_set_state: NULL argument");
    if (!r->initialised) return mhx_fail(MHX_ESTATE, "mhx_run_set_state: run is not initialised");
    mhx_ctx* ctx = r->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)r->n, d = (size_t)r->dim;
    COPY_SYNC(r->ctx->stream, r->d_x, x, d * n * sizeof(mhx_real), hipMemcpyHostToDevice);
    if (r->kind == RUN_RWMH) { int rc = rwmh_canonical_zero(r); if (rc) return rc; }
    if (r->kind == RUN_EMCEE) { int rc = emcee_sync_state(r, 0); if (rc) return rc; }
    if (r->kind == RUN_MALA) return mala_eval_state(r, 0);   // src/MALA.jl:27-35: lp and gradient are recomputed
    // lp is a cache of logdensity(model, x) (src/AdvancedMH.jl:75): recompute it
    const unsigned grid = (unsigned)((r->n + 255) / 256);
    int nn = r->n, dd = r->dim, kind = r->target->kind, np = r->target->nparams, lanes = r->coop_L;
    mhx_real cst = r->target->cst;
    const mhx_real* tp = r->target->dparams;
    const mhx_real* dx = r->d_x;
    mhx_real* dlp = r->d_lp;
    if (kind == MHX_TARGET_USER) {
        jit_module* m = nullptr;
        hipFunction_t fn;
        int rc = jit_generic_rwmh(r->target, &m);
        if (rc) return rc;
        if ((rc = jit_function(m, "mhx_jit_target_eval", &fn))) return rc;
        void* params[] = {&dx, &dlp, &nn, &dd, &kind, &tp, &np, &cst, &lanes};
        if ((rc = launch_module(fn, grid, 256, ctx->stream, params))) return rc;
    } else {
        hipLaunchKernelGGL(k_target_eval, dim3(grid), dim3(256), 0, ctx->stream, dx, dlp, nn, dd, kind, tp, np, cst, lanes);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return r->kind == RUN_RWMH ? rwmh_whiten(r) : MHX_OK;
}

// ---- checkpoint / resume: the complete state of a run as one host blob ------------------------------------------
struct ckpt_header {
    uint32_t magic, version;
    int32_t kind, dim, n, flags;
    uint64_t tau, seed, first_id;
    // what fixes the arithmetic of the continuation: the log-density, the proposal family, the kernel variant and
    // the lanes per chain (summation order of lp); and the exact payload size
    int32_t target_kind, prop_kind, variant, coop_L;
    uint64_t payload_bytes;
    double last_eta;           // RAM: step size of the latest adapting transition
};
static const uint32_t k_ckpt_magic = 0x5848484du;          // "MHXX"
// the static-proposal bit and, for Ensemble runs, how many ensembles the walkers form (2 x 8 walkers are not 1 x 16)
static int32_t ckpt_flags(const mhx_run* r) { return (r->flags & MHX_FLAG_STATIC_PROPOSAL) | (r->n_ens > 1 ? r->n_ens << 8 : 0); }
struct ckpt_part { void* dev; size_t bytes; };
// the device arrays that make up the state, in blob order
static std::vector<ckpt_part> ckpt_parts(mhx_run* r)
{
    const size_t n = (size_t)r->n, d = (size_t)r->dim;
    std::vector<ckpt_part> p = {{r->d_x, d * n * sizeof(mhx_real)}, {r->d_lp, n * sizeof(mhx_real)},
                                {r->d_acc, n * sizeof(uint32_t)}, {r->d_last, n}};
    if (r->kind == RUN_RAM) {
        const size_t trip = (size_t)mhx_ram_tri_pad(r->dim);
        p.push_back({r->d_S, 2 * trip * n * sizeof(mhx_real)});
        p.push_back({r->d_Ssel, n});
        p.push_back({r->d_status, n});
        p.push_back({r->d_dmin, d * n * sizeof(mhx_real)});
        p.push_back({r->d_dmax, d * n * sizeof(mhx_real)});
        p.push_back({r->d_loga, n * sizeof(mhx_real)});
    }
    if (r->kind == RUN_MALA) p.push_back({r->d_gx, d * n * sizeof(mhx_real)});
    if (r->d_qx) p.push_back({r->d_qx, n * sizeof(mhx_real)});
    return p;
}
int api_run_state_size(mhx_run* r, size_t* bytes)
{
    if (!r || !bytes) return mhx_fail(MHX_EINVAL, "mhx_run_state_size: NULL argument");
    size_t total = sizeof(ckpt_header);
    for (const auto& p : ckpt_parts(r)) total += p.bytes;
    *bytes = total;
    return MHX_OK;
}
int api_run_save_state(mhx_run* r, void* blob, size_t bytes)
{
    if (!r || !blob) return mhx_fail(MHX_EINVAL, "mhx_run_save_state: NULL argument");
    if (!r->initialised) return mhx_fail(MHX_ESTATE, "mhx_run_save_state: run is not initialised");
    size_t need = 0;
    api_run_state_size(r, &need);
    if (bytes < need) return mhx_fail(MHX_EINVAL, "mhx_run_save_state: the blob holds %zu bytes, the state needs %zu", bytes, need);
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->kind == RUN_EMCEE) { int rc = emcee_sync_state(r, 1); if (rc) return rc; }      // walker-major -> ABI layout
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    ckpt_header h = {k_ckpt_magic, 2u, (int32_t)r->kind, r->dim, r->n, ckpt_flags(r), r->tau, r->seed, r->first_id,
                     r->target->kind, r->prop_kind, r->variant, r->coop_L, (uint64_t)(need - sizeof(ckpt_header)), r->last_eta};
    char* out = (char*)blob;
    memcpy(out, &h, sizeof h);
    out += sizeof h;
    for (const auto& p : ckpt_parts(r)) {
        COPY_SYNC(r->ctx->stream, out, p.dev, p.bytes, hipMemcpyDeviceToHost);
        out += p.bytes;
    }
    return MHX_OK;
}
int api_run_load_state(mhx_run* r, const void* blob, size_t bytes)
{
    if (!r || !blob) return mhx_fail(MHX_EINVAL, "mhx_run_load_state: NULL argument");
    size_t need = 0;
    api_run_state_size(r, &need);
    ckpt_header h;
    if (bytes < sizeof h) return mhx_fail(MHX_EINVAL, "mhx_run_load_state: the blob is too short");
    memcpy(&h, blob, sizeof h);
    if (h.magic != k_ckpt_magic || h.version != 2u) return mhx_fail(MHX_EINVAL, "mhx_run_load_state: not a state blob of this library version");
    if (h.kind != (int32_t)r->kind || h.dim != r->dim || h.n != r->n || bytes != need || h.payload_bytes != need - sizeof h)
        return mhx_fail(MHX_EINVAL, "mhx_run_load_state: the blob is a state of sampler kind %d, dim %d, %d chains (%zu bytes); "
                                "this run is kind %d, dim %d, %d chains (%zu bytes)", h.kind, h.dim, h.n, bytes, (int)r->kind, r->dim, r->n, need);
    if (h.flags != ckpt_flags(r) || h.target_kind != r->target->kind || h.prop_kind != r->prop_kind ||
        h.variant != r->variant || h.coop_L != r->coop_L)
        return mhx_fail(MHX_EINVAL, "mhx_run_load_state: the blob was saved by a different configuration (target kind %d, proposal kind %d, "
                                "static %d, kernel variant %d, %d lane(s) per chain; this run: %d, %d, %d, %d, %d) -- the continuation "
                                "would not be the saved chain", h.target_kind, h.prop_kind, h.flags, h.variant, h.coop_L,
                    r->target->kind, r->prop_kind, ckpt_flags(r), r->variant, r->coop_L);
    HIP_TRY(hipSetDevice(r->ctx->device));
    const char* in = (const char*)blob + sizeof h;
    for (const auto& p : ckpt_parts(r)) {
        COPY_SYNC(r->ctx->stream, p.dev, in, p.bytes, hipMemcpyHostToDevice);
        in += p.bytes;
    }
    // the counter-based streams continue where the saved run stopped: same seed, same global ids, same step counter
    r->seed = h.seed; r->first_id = h.first_id; r->tau = h.tau; r->last_eta = h.last_eta;
    r->initialised = true;
    r->n_saved = 0;
    HIP_TRY(hipMemsetAsync(r->d_acc_total, 0, sizeof(unsigned long long), r->ctx->stream));
    if (r->kind == RUN_EMCEE) { int rc = emcee_sync_state(r, 0); if (rc) return rc; }      // ABI layout -> walker-major
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    return MHX_OK;
}

int api_run_stats(mhx_run* r, mhx_stats* out)
{
    if (!r || !out) return mhx_fail(MHX_EINVAL, "mhx_run_stats: NULL argument");
    *out = r->stats;
    out->tainted = r->ctx->tainted ? 1 : 0;
    out->reserved_ = 0;
    return MHX_OK;
}

int api_run_host_stats(mhx_run* r, mhx_host_stats* out)
{
    if (!r || !out) return mhx_fail(MHX_EINVAL, "mhx_run_host_stats: NULL argument");
    *out = r->host_stats;
    return MHX_OK;
}

int api_run_shape(const mhx_run* r, int32_t* dim, int32_t* nchains)
{
    if (dim) *dim = r->dim;
    if (nchains) *nchains = r->n;
    return MHX_OK;
}

int api_run_destroy(mhx_run* r)
{
    if (!r) return MHX_OK;
    (void)hipSetDevice(r->ctx->device);
    delete r;
    return MHX_OK;
}

#include "mhx_api_emcee.inc"
#include "mhx_api_ram.inc"
#include "mhx_api_mala.inc"
#include "mhx_api_diag.inc"

}  // namespace MHX_NS
