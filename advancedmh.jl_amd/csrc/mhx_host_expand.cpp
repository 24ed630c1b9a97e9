// mhx_host_expand.cpp -- host threads that rebuild the caller's sample tensor from accept-compacted slabs (mhx_host_expand.h,
// include/mhx.h: mhx_compact_hdr).  No HIP here.
//
// Work split: a block (one slab: `count` consecutive samples of all chains) is cut into CHUNKS of chains (a multiple of 64, so a
// chunk is whole mask words); a worker takes a chunk through all samples of the block, keeping the chunk's current state
// cur[dim+1][chunk] in its L2 -- the row above never has to be read back from DRAM -- and for every (sample, parameter) row
//   1. merges the changed chains' values into cur's row  (AVX-512: one vexpandpd / vexpandps per 64 bytes, mask = 8 / 16 bits of
//      the block's mask word; otherwise a scalar walk over the set bits),
//   2. streams the row to the caller's tensor with non-temporal stores (it will not be read again by this loop).
// DRAM traffic per block = the tensor rows written once + the payload read once + one row per chunk and block (the state at the
// block's first sample, taken from the row the previous block wrote).
#include "mhx_host_expand.h"

#include "mhx_impl.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <emmintrin.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace {

static_assert(sizeof(mhx_compact_hdr) == 64, "the wire header is 64 bytes");

inline size_t pad8(size_t x) { return (x + 7) & ~(size_t)7; }

int cgroup_quota_cpus()
{
    // cgroup v2: "max 100000" or "<quota> <period>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us
    double q = -1.0, p = 0.0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = {0};
        if (fscanf(f, "%63s %lf", a, &p) == 2 && strcmp(a, "max") != 0) q = atof(a);
        fclose(f);
    } else {
        FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%lf", &q) == 1 && fscanf(fp, "%lf", &p) == 1) {}
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (q > 0.0 && p > 0.0) return (int)(q / p + 0.5) > 0 ? (int)(q / p + 0.5) : 1;
    return 0;                            // no quota
}

struct block_view {
    mhx_compact_hdr h;
    const uint64_t* mask;                // [count][words]
    const uint32_t* rank;                // [count][words]: set bits of the block before word (i, w)
    const uint8_t* acc;                  // [count][nchains]
    const unsigned char* payload;
};

// the block's fixed part against the job: 0 or an error code (message set)
int view_block(const void* block, int64_t n_samples, block_view* v)
{
    memcpy(&v->h, block, sizeof v->h);
    const mhx_compact_hdr& h = v->h;
    if (h.magic != MHX_COMPACT_MAGIC) return mhx_fail(MHX_EINVAL, "compact block: bad magic %08x", h.magic);
    if ((h.elem_bytes != 4 && h.elem_bytes != 8) || !h.dim1 || !h.nchains || !h.count)
        return mhx_fail(MHX_EINVAL, "compact block: elem_bytes %u dim1 %u nchains %u count %u", h.elem_bytes, h.dim1, h.nchains, h.count);
    if (h.words != (h.nchains + 63) / 64) return mhx_fail(MHX_EINVAL, "compact block: %u mask words for %u chains", h.words, h.nchains);
    if ((int64_t)(h.first_sample + h.count) > n_samples)
        return mhx_fail(MHX_EINVAL, "compact block: samples [%llu, %llu) of a tensor with %lld", (unsigned long long)h.first_sample,
                        (unsigned long long)(h.first_sample + h.count), (long long)n_samples);
    const size_t po = mhx_compact_payload_offset(h.count, h.words, h.nchains);
    if (h.payload_offset != po || h.block_bytes != po + h.total_changed * h.dim1 * (uint64_t)h.elem_bytes)
        return mhx_fail(MHX_EINVAL, "compact block: inconsistent sizes (payload at %llu, expected %zu; %llu bytes for %llu columns)",
                        (unsigned long long)h.payload_offset, po, (unsigned long long)h.block_bytes, (unsigned long long)h.total_changed);
    const unsigned char* b = (const unsigned char*)block;
    const size_t nw = (size_t)h.count * h.words;
    v->mask = (const uint64_t*)(b + sizeof(mhx_compact_hdr));
    v->rank = (const uint32_t*)(b + sizeof(mhx_compact_hdr) + 8 * nw);
    v->acc = b + sizeof(mhx_compact_hdr) + 8 * nw + pad8(4 * nw);
    v->payload = b + po;
    return MHX_OK;
}

// rank[] must be the running popcount of mask[] and bits beyond nchains must be clear: every later read of the payload is
// indexed through them
int check_ranks(const block_view& v)
{
    const mhx_compact_hdr& h = v.h;
    uint64_t run = 0;
    const uint64_t tail = (h.nchains & 63) ? ~0ull << (h.nchains & 63) : 0ull;
    for (uint32_t i = 0; i < h.count; ++i)
        for (uint32_t w = 0; w < h.words; ++w) {
            const size_t at = (size_t)i * h.words + w;
            if (v.rank[at] != run) return mhx_fail(MHX_EINVAL, "compact block: rank of word (%u, %u) is %u, the masks before it hold %llu", i, w, v.rank[at], (unsigned long long)run);
            if (w + 1 == h.words && (v.mask[at] & tail)) return mhx_fail(MHX_EINVAL, "compact block: mask bits beyond chain %u in sample %u", h.nchains, i);
            run += (uint64_t)__builtin_popcountll(v.mask[at]);
        }
    if (run != h.total_changed) return mhx_fail(MHX_EINVAL, "compact block: the masks hold %llu columns, the header says %llu", (unsigned long long)run, (unsigned long long)h.total_changed);
    if (h.first_sample == 0) {           // nothing precedes sample 0: all of it must be there
        for (uint32_t w = 0; w < h.words; ++w) {
            const uint64_t want = (w + 1 == h.words && tail) ? ~tail : ~0ull;
            if (v.mask[w] != want) return mhx_fail(MHX_EINVAL, "compact block: sample 0 must carry every chain");
        }
    }
    return MHX_OK;
}

// ---- one row of one chunk: merge `src` (the changed chains' values, in chain order) into cur[0, nc) under the mask words, then
// write cur[0, nc) to out.  T = uint32_t / uint64_t (the element's bits).
template <typename T>
void row_scalar(T* cur, T* out, const uint64_t* mw, int nwords, int nc, const T* src, bool stream)
{
    for (int w = 0; w < nwords; ++w) {
        uint64_t bits = mw[w];
        T* c = cur + 64 * w;
        while (bits) {
            c[__builtin_ctzll(bits)] = *src++;
            bits &= bits - 1;
        }
    }
    const size_t bytes = (size_t)nc * sizeof(T);
    if (stream && bytes >= 128) {
        unsigned char* o = (unsigned char*)out;
        const unsigned char* c = (const unsigned char*)cur;
        const size_t head = (size_t)(-(intptr_t)(uintptr_t)o) & 15;
        memcpy(o, c, head);
        const size_t body = (bytes - head) & ~(size_t)15;
        for (size_t i = 0; i < body; i += 16) _mm_stream_si128((__m128i*)(o + head + i), _mm_loadu_si128((const __m128i*)(c + head + i)));
        memcpy(o + head + body, c + head + body, bytes - head - body);
    } else {
        memcpy(out, cur, bytes);
    }
}

// cur[0, bytes) -> out with non-temporal stores on out's own 64-byte lines (out at any alignment: numpy hands out 16-byte aligned
// blocks); the partial lines at both ends go through the cache
__attribute__((target("avx512f"))) void stream_bytes_avx512(unsigned char* out, const unsigned char* cur, size_t bytes)
{
    size_t head = (size_t)(-(intptr_t)(uintptr_t)out) & 63;
    if (head > bytes) head = bytes;
    memcpy(out, cur, head);
    const size_t body = (bytes - head) & ~(size_t)63;
    for (size_t i = 0; i < body; i += 64) _mm512_stream_si512((__m512i*)(out + head + i), _mm512_loadu_si512((const void*)(cur + head + i)));
    memcpy(out + head + body, cur + head + body, bytes - head - body);
}

__attribute__((target("avx512f"))) void row_avx512_64(uint64_t* cur, uint64_t* out, const uint64_t* mw, int nwords, int nc, const uint64_t* src,
                                                     bool stream)
{
    const int full = nc / 64;            // words whose 64 chains all exist
    const bool nt = stream && ((uintptr_t)out & 63) == 0;
    if (stream && !nt) {                 // merge in place, then stream the row on the destination's own lines
        for (int w = 0; w < full; ++w) {
            const uint64_t bits = mw[w];
            if (!bits) continue;
            double* c = (double*)(cur + 64 * w);
            for (int g = 0; g < 8; ++g) {
                const __mmask8 m = (__mmask8)(bits >> (8 * g));
                if (!m) continue;
                _mm512_store_pd(c + 8 * g, _mm512_mask_expandloadu_pd(_mm512_load_pd(c + 8 * g), m, src));
                src += __builtin_popcount((unsigned)m);
            }
        }
        stream_bytes_avx512((unsigned char*)out, (const unsigned char*)cur, (size_t)full * 64 * 8);
        if (full < nwords) row_scalar<uint64_t>(cur + 64 * full, out + 64 * full, mw + full, nwords - full, nc - 64 * full, src, false);
        return;
    }
    for (int w = 0; w < full; ++w) {
        const uint64_t bits = mw[w];
        double* c = (double*)(cur + 64 * w);
        double* o = (double*)(out + 64 * w);
        if (bits == 0) {
            for (int g = 0; g < 8; ++g) {
                const __m512d v = _mm512_load_pd(c + 8 * g);
                if (nt) _mm512_stream_pd(o + 8 * g, v); else _mm512_storeu_pd(o + 8 * g, v);
            }
            continue;
        }
        for (int g = 0; g < 8; ++g) {
            const __mmask8 m = (__mmask8)(bits >> (8 * g));
            __m512d v = _mm512_load_pd(c + 8 * g);
            if (m) {
                v = _mm512_mask_expandloadu_pd(v, m, src);
                src += __builtin_popcount((unsigned)m);
                _mm512_store_pd(c + 8 * g, v);
            }
            if (nt) _mm512_stream_pd(o + 8 * g, v); else _mm512_storeu_pd(o + 8 * g, v);
        }
    }
    if (full < nwords) row_scalar<uint64_t>(cur + 64 * full, out + 64 * full, mw + full, nwords - full, nc - 64 * full, src, false);
}

__attribute__((target("avx512f"))) void row_avx512_32(uint32_t* cur, uint32_t* out, const uint64_t* mw, int nwords, int nc, const uint32_t* src,
                                                     bool stream)
{
    const int full = nc / 64;
    const bool nt = stream && ((uintptr_t)out & 63) == 0;
    if (stream && !nt) {
        for (int w = 0; w < full; ++w) {
            const uint64_t bits = mw[w];
            if (!bits) continue;
            float* c = (float*)(cur + 64 * w);
            for (int g = 0; g < 4; ++g) {
                const __mmask16 m = (__mmask16)(bits >> (16 * g));
                if (!m) continue;
                _mm512_store_ps(c + 16 * g, _mm512_mask_expandloadu_ps(_mm512_load_ps(c + 16 * g), m, src));
                src += __builtin_popcount((unsigned)m);
            }
        }
        stream_bytes_avx512((unsigned char*)out, (const unsigned char*)cur, (size_t)full * 64 * 4);
        if (full < nwords) row_scalar<uint32_t>(cur + 64 * full, out + 64 * full, mw + full, nwords - full, nc - 64 * full, src, false);
        return;
    }
    for (int w = 0; w < full; ++w) {
        const uint64_t bits = mw[w];
        float* c = (float*)(cur + 64 * w);
        float* o = (float*)(out + 64 * w);
        for (int g = 0; g < 4; ++g) {
            const __mmask16 m = (__mmask16)(bits >> (16 * g));
            __m512 v = _mm512_load_ps(c + 16 * g);
            if (m) {
                v = _mm512_mask_expandloadu_ps(v, m, src);
                src += __builtin_popcount((unsigned)m);
                _mm512_store_ps(c + 16 * g, v);
            }
            if (nt) _mm512_stream_ps(o + 16 * g, v); else _mm512_storeu_ps(o + 16 * g, v);
        }
    }
    if (full < nwords) row_scalar<uint32_t>(cur + 64 * full, out + 64 * full, mw + full, nwords - full, nc - 64 * full, src, false);
}

bool have_avx512()
{
    static const bool v = __builtin_cpu_supports("avx512f") && !getenv("MHX_EXPAND_NO_AVX512");
    return v;
}

// one chunk of chains [c0, c1) through every sample of the block
template <typename T>
void expand_chunk(const block_view& v, T* samples, uint8_t* accepted, uint32_t c0, uint32_t c1, T* cur, size_t cur_ld, bool stream)
{
    const mhx_compact_hdr& h = v.h;
    const size_t n = h.nchains, d1 = h.dim1;
    const uint32_t w0 = c0 / 64, nwords = (c1 - c0 + 63) / 64;
    const int nc = (int)(c1 - c0);
    const bool wide = have_avx512();
    if (h.first_sample) {                // the chunk's state at the row above the block: the previous block wrote it
        const T* above = samples + (size_t)(h.first_sample - 1) * d1 * n + c0;
        for (size_t k = 0; k < d1; ++k) memcpy(cur + k * cur_ld, above + k * n, (size_t)nc * sizeof(T));
    }
    const T* pay = (const T*)v.payload;
    for (uint32_t i = 0; i < h.count; ++i) {
        const size_t at = (size_t)i * h.words;
        const uint64_t first = v.rank[at];                                               // columns of the block before sample i
        const uint64_t m_i = (i + 1 < h.count ? (uint64_t)v.rank[at + h.words] : h.total_changed) - first;
        const uint64_t r0 = v.rank[at + w0] - first;                                     // changed chains of sample i before c0
        const T* src = pay + first * d1 + r0;
        T* out = samples + (size_t)(h.first_sample + i) * d1 * n + c0;
        const uint64_t* mw = v.mask + at + w0;
        for (size_t k = 0; k < d1; ++k) {
            if (sizeof(T) == 8) {
                if (wide) row_avx512_64((uint64_t*)cur + k * cur_ld, (uint64_t*)out + k * n, mw, (int)nwords, nc, (const uint64_t*)src + k * m_i, stream);
                else row_scalar<uint64_t>((uint64_t*)cur + k * cur_ld, (uint64_t*)out + k * n, mw, (int)nwords, nc, (const uint64_t*)src + k * m_i, stream);
            } else {
                if (wide) row_avx512_32((uint32_t*)cur + k * cur_ld, (uint32_t*)out + k * n, mw, (int)nwords, nc, (const uint32_t*)src + k * m_i, stream);
                else row_scalar<uint32_t>((uint32_t*)cur + k * cur_ld, (uint32_t*)out + k * n, mw, (int)nwords, nc, (const uint32_t*)src + k * m_i, stream);
            }
        }
        if (accepted) memcpy(accepted + (size_t)(h.first_sample + i) * n + c0, v.acc + (size_t)i * n + c0, (size_t)nc);
    }
    _mm_sfence();                        // the non-temporal rows are visible before the chunk is reported done
}

}  // namespace

// ---- NUMA placement (mhx_host_expand.h) -----------------------------------------------------------------------------------
int mhx_numa_node_of_pci(const char* bus_id)
{
    if (!bus_id || !*bus_id) return -1;
    char path[256], id[64];
    size_t n = 0;
    for (; bus_id[n] && n + 1 < sizeof id; ++n) id[n] = (char)((bus_id[n] >= 'A' && bus_id[n] <= 'F') ? bus_id[n] + 32 : bus_id[n]);   // sysfs spells hex in lower case
    id[n] = 0;
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", id);
    int node = -1;
    if (FILE* f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return -1;
    if (FILE* f = fopen("/sys/devices/system/node/node1/cpulist", "r")) fclose(f); else return -1;      // a single node: nothing to place
    return node;
}
void mhx_numa_prefer(int node)
{
#ifdef SYS_set_mempolicy
    if (node < 0 || node >= 1024) return;
    unsigned long mask[16] = {0};
    mask[node / 64] = 1ul << (node % 64);
    (void)syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, 1025ul);
#else
    (void)node;
#endif
}
void mhx_numa_default(void)
{
#ifdef SYS_set_mempolicy
    (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
#endif
}
// the CPUs of `node` this process may run on ("0-63,128-191"); false when there are none (a cpuset elsewhere: stay unpinned)
static bool node_cpus(int node, cpu_set_t* out)
{
    char path[128];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) { fclose(f); return false; }
    CPU_ZERO(out);
    int a = 0, b = 0, got = 0;
    char sep = 0;
    while (fscanf(f, "%d", &a) == 1) {
        b = a;
        if (fscanf(f, "%c", &sep) == 1 && sep == '-') { if (fscanf(f, "%d", &b) != 1) break; if (fscanf(f, "%c", &sep) != 1) sep = 0; }
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) { CPU_SET(c, out); ++got; }
        if (sep != ',') break;
    }
    fclose(f);
    return got > 0;
}

int mhx_host_usable_cpus(void)
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && (n <= 0 || a < n)) n = a;
    }
    const int q = cgroup_quota_cpus();
    if (q > 0 && (n <= 0 || q < n)) n = q;
    return n > 0 ? n : 1;
}

size_t mhx_compact_payload_offset(uint32_t count, uint32_t words, uint32_t nchains)
{
    const size_t nw = (size_t)count * words;
    return sizeof(mhx_compact_hdr) + 8 * nw + pad8(4 * nw) + pad8((size_t)count * nchains);
}

// ---------------------------------------------------------------------------------------------------------------------------
class mhx_expander {
public:
    mhx_expander(int threads, int chunk, int numa_node = -1) : chunk_(chunk)
    {
        if (threads <= 0) threads = mhx_host_usable_cpus();
        if (threads > 64) threads = 64;
        nthreads_ = threads;
        scratch_.resize((size_t)threads);
        pin_ = numa_node >= 0 && node_cpus(numa_node, &cpus_);
        for (int t = 0; t < threads; ++t)
            workers_.emplace_back([this, t] {
                if (pin_) (void)pthread_setaffinity_np(pthread_self(), sizeof cpus_, &cpus_);   // the node's CPUs, not one each: the kernel balances
                work(t);
            });
    }
    ~mhx_expander()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
        for (auto& s : scratch_) free(s.p);
    }
    int threads() const { return nthreads_; }

    uint64_t submit(const mhx_expand_job& j)
    {
        std::lock_guard<std::mutex> g(mu_);
        jobs_.push_back(j);
        const uint64_t seq = submitted_++;
        cv_.notify_all();
        return seq;
    }
    void wait(uint64_t seq)
    {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return finished_ > seq; });
    }
    int drain()
    {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return finished_ == submitted_; });
        const int rc = err_;
        if (rc) mhx_fail(rc, "%s", errmsg_.c_str());
        err_ = 0;
        errmsg_.clear();
        return rc;
    }
    double busy(int reset)
    {
        std::lock_guard<std::mutex> g(mu_);
        const double b = busy_s_;
        if (reset) busy_s_ = 0.0;
        return b;
    }

private:
    struct scratch { void* p = nullptr; size_t cap = 0; };
    // what the workers share about the job in flight (guarded by mu_ except the chunk counter)
    struct current {
        uint64_t seq = ~0ull;
        bool ready = false, failed = false;
        block_view v{};
        mhx_expand_job job{};
        uint32_t chunk = 64, nchunks = 0;
        size_t cur_ld = 0;
        bool stream = false;
        int left = 0;                    // workers still inside this job
        std::chrono::steady_clock::time_point t0;
    } cur_;
    std::atomic<uint32_t> next_chunk_{0};

    uint32_t choose_chunk(const mhx_compact_hdr& h) const
    {
        if (chunk_ > 0) return (uint32_t)((chunk_ + 63) / 64 * 64);
        // the chunk's state cur[dim1][chunk] should sit in the worker's L2 beside the rows being streamed: ~320 KB
        size_t c = ((size_t)320 << 10) / ((size_t)h.dim1 * h.elem_bytes) / 64 * 64;
        if (c < 64) c = 64;
        if (c > 4096) c = 4096;
        // enough chunks for the threads to share (two each), as long as a chunk stays a whole mask word
        const size_t per = (size_t)h.nchains / (2 * (size_t)nthreads_) / 64 * 64;
        if (per >= 64 && per < c) c = per;
        if (per < 64) c = 64;
        return (uint32_t)c;
    }

    void fail_locked(int rc)
    {
        if (!err_) { err_ = rc; errmsg_ = mhx_last_error(); }
        cur_.failed = true;
    }

    void work(int t)
    {
        uint64_t seq = 0;
        for (;; ++seq) {
            // ---- worker 0 opens job `seq`: waits for its copy, reads the header, publishes the chunking
            if (t == 0) {
                mhx_expand_job job;
                {
                    std::unique_lock<std::mutex> g(mu_);
                    cv_.wait(g, [&] { return stop_ || (submitted_ > seq && finished_ == seq); });
                    if (stop_) return;
                    job = jobs_.front();
                    jobs_.pop_front();
                }
                int rc = err_ ? err_ : MHX_OK;         // after a failure the rest of the queue is only drained
                if (job.wait) { const int rw = job.wait(job.wait_arg); if (!rc) rc = rw; }
                block_view v{};
                if (!rc) rc = view_block(job.block, job.n_samples, &v);
                if (!rc) rc = check_ranks(v);
                std::lock_guard<std::mutex> g(mu_);
                cur_.seq = seq;
                cur_.job = job;
                cur_.v = v;
                cur_.failed = false;
                if (rc) fail_locked(rc);
                else {
                    cur_.chunk = choose_chunk(v.h);
                    cur_.nchunks = (v.h.nchains + cur_.chunk - 1) / cur_.chunk;
                    cur_.cur_ld = (size_t)cur_.chunk + 64 / v.h.elem_bytes;      // one line of padding: rows do not alias in L1
                    // streaming stores pay off once the tensor is far larger than the caches
                    cur_.stream = (size_t)v.h.count * v.h.dim1 * v.h.nchains * v.h.elem_bytes >= ((size_t)8 << 20);
                }
                next_chunk_.store(0, std::memory_order_relaxed);
                cur_.left = nthreads_;
                cur_.t0 = std::chrono::steady_clock::now();
                cur_.ready = true;
                cv_.notify_all();
            }
            // ---- everyone: take chunks
            current c;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || (cur_.ready && cur_.seq == seq); });
                if (stop_ && !(cur_.ready && cur_.seq == seq)) return;
                c = cur_;
            }
            if (!c.failed) {
                const mhx_compact_hdr& h = c.v.h;
                const size_t need = (size_t)h.dim1 * c.cur_ld * h.elem_bytes + 64;
                scratch& s = scratch_[(size_t)t];
                if (s.cap < need) {
                    free(s.p);
                    s.p = aligned_alloc(64, (need + 63) / 64 * 64);
                    s.cap = s.p ? need : 0;
                }
                if (!s.p) {
                    std::lock_guard<std::mutex> g(mu_);
                    mhx_fail(MHX_ENOMEM, "compact expand: %zu bytes of scratch", need);
                    fail_locked(MHX_ENOMEM);
                } else {
                    for (;;) {
                        const uint32_t g = next_chunk_.fetch_add(1, std::memory_order_relaxed);
                        if (g >= c.nchunks) break;
                        const uint32_t c0 = g * c.chunk, c1 = c0 + c.chunk < h.nchains ? c0 + c.chunk : h.nchains;
                        if (h.elem_bytes == 8) expand_chunk<uint64_t>(c.v, (uint64_t*)c.job.samples, c.job.accepted, c0, c1, (uint64_t*)s.p, c.cur_ld, c.stream);
                        else expand_chunk<uint32_t>(c.v, (uint32_t*)c.job.samples, c.job.accepted, c0, c1, (uint32_t*)s.p, c.cur_ld, c.stream);
                    }
                }
            }
            // ---- the last one out closes the job
            {
                std::unique_lock<std::mutex> g(mu_);
                if (--cur_.left == 0) {
                    busy_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - cur_.t0).count();
                    cur_.ready = false;
                    finished_ = seq + 1;
                    cv_.notify_all();
                } else {
                    cv_.wait(g, [&] { return stop_ || finished_ > seq; });
                    if (stop_ && finished_ <= seq) return;
                }
            }
        }
    }

    int nthreads_ = 1, chunk_ = 0;
    bool pin_ = false;
    cpu_set_t cpus_;
    std::vector<std::thread> workers_;
    std::vector<scratch> scratch_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<mhx_expand_job> jobs_;
    uint64_t submitted_ = 0, finished_ = 0;
    bool stop_ = false;
    int err_ = 0;
    std::string errmsg_;
    double busy_s_ = 0.0;
};

mhx_expander* mhx_expander_create(int threads, int chunk_chains, int numa_node) { return new mhx_expander(threads, chunk_chains, numa_node); }
void mhx_expander_destroy(mhx_expander* e) { delete e; }
int mhx_expander_threads(const mhx_expander* e) { return e->threads(); }
uint64_t mhx_expander_submit(mhx_expander* e, const mhx_expand_job& job) { return e->submit(job); }
void mhx_expander_wait(mhx_expander* e, uint64_t seq) { e->wait(seq); }
int mhx_expander_drain(mhx_expander* e) { return e->drain(); }
double mhx_expander_busy_seconds(mhx_expander* e, int reset) { return e->busy(reset); }

// include/mhx.h: expand ONE block into the caller's tensor (blocking).  A host with its own transport -- the workers of
// Distributed.jl, MPI ranks shipping their shards' samples to the master -- moves blocks instead of tensors and expands them here.
extern "C" int mhx_compact_expand(const void* block, size_t block_bytes, void* samples, uint8_t* accepted, int64_t n_samples, int32_t threads)
{
    if (!block || !samples) return mhx_fail(MHX_EINVAL, "mhx_compact_expand: NULL argument");
    if (block_bytes < sizeof(mhx_compact_hdr)) return mhx_fail(MHX_EINVAL, "mhx_compact_expand: %zu bytes hold no header", block_bytes);
    mhx_compact_hdr h;
    memcpy(&h, block, sizeof h);
    if (h.magic != MHX_COMPACT_MAGIC) return mhx_fail(MHX_EINVAL, "mhx_compact_expand: bad magic %08x", h.magic);
    if (h.block_bytes > block_bytes) return mhx_fail(MHX_EINVAL, "mhx_compact_expand: the block says %llu bytes, %zu were given", (unsigned long long)h.block_bytes, block_bytes);
    // sizes first (view_block), so that the arrays the checks walk lie inside the buffer
    if (h.count && h.words && mhx_compact_payload_offset(h.count, h.words, h.nchains) > block_bytes)
        return mhx_fail(MHX_EINVAL, "mhx_compact_expand: the block's arrays do not fit into %zu bytes", block_bytes);
    mhx_expander e(threads, 0, -1);
    mhx_expand_job j;
    j.block = block;
    j.samples = samples;
    j.accepted = accepted;
    j.n_samples = n_samples;
    e.submit(j);
    return e.drain();
}
