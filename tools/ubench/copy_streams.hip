// copy_streams.hip -- what HBM gives a kernel shaped like the RAM sweep: one wave per 80 KB segment,
// streaming it in 1-2 KB pieces; read-only, and read + write to a second segment.
//   hipcc --offload-arch=gfx950 -O3 -o copy_streams copy_streams.hip && ./copy_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, bool WRITE>
__global__ void __launch_bounds__(64) k_stream(const f4* __restrict__ in, f4* __restrict__ out, float* sink, int nvec, long stride4)
{
    const long seg = blockIdx.x;
    const f4* src = in + seg * stride4;
    f4* dst = out + seg * stride4;
    const int t = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    f4 regs[NV];
    const int nch = (nvec + NV * 64 - 1) / (NV * 64);
#pragma unroll
    for (int v = 0; v < NV; ++v) { const int i = v * 64 + t; regs[v] = i < nvec ? src[i] : acc; }
    for (int k = 0; k < nch; ++k) {
        f4 cur[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[v] = regs[v];
        if (k + 1 < nch) {
#pragma unroll
            for (int v = 0; v < NV; ++v) { const int i = ((k + 1) * NV + v) * 64 + t; regs[v] = i < nvec ? src[i] : acc; }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = (k * NV + v) * 64 + t;
            if (WRITE) { if (i < nvec) dst[i] = cur[v] * 1.0001f; }
            else acc += cur[v];
        }
    }
    if (!WRITE && acc.x == 12345.678f) sink[0] = acc.y;
}

// the same copy with LDS-DMA loads and hand-counted waits: the wave never waits for a store to be acknowledged
// (hipcc's own bookkeeping puts s_waitcnt vmcnt(0) in the loop above: loads and stores share the counter)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int NV>
__global__ void __launch_bounds__(64) k_stream_dma(const f4* __restrict__ in, f4* __restrict__ out, int nvec, long stride4)
{
    __shared__ f4 ring[2 * NV * 64];
    const long seg = blockIdx.x;
    const f4* src = in + seg * stride4;
    f4* dst = out + seg * stride4;
    const int t = threadIdx.x;
    const int nch = nvec / (NV * 64);                    // whole chunks only (the tail is left out of the byte count)
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
#pragma unroll
    for (int v = 0; v < NV; ++v) glds16(src + v * 64 + t, __builtin_amdgcn_readfirstlane(base + (unsigned)(v * 1024)));
    for (int k = 0; k < nch; ++k) {
        const int sl = k & 1;
        if (k + 1 < nch) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                glds16(src + ((k + 1) * NV + v) * 64 + t, __builtin_amdgcn_readfirstlane(base + (unsigned)(((sl ^ 1) * NV + v) * 1024)));
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NV) : "memory");      // L(k) landed; S(k-1), L(k+1) may fly
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NV) : "memory");
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f4 x = ring[(sl * NV + v) * 64 + t];
            dst[(k * NV + v) * 64 + t] = x * 1.0001f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
// write-only: what the sample recorder of the RWMH kernel asks of HBM (one-shot flat float4 stores)
__global__ void __launch_bounds__(256) k_fill(f4* __restrict__ out, long n4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const f4 v = {1.0f, 2.0f, 3.0f, (float)i};
    if (i < n4) out[i] = v;
}
// the recorder's own shape: a wave writes 128-byte row segments, rows `ld` floats apart
__global__ void __launch_bounds__(256) k_fill_rows(float* __restrict__ out, int rows, long ld)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;               // chain
    for (int r = 0; r < rows; ++r) out[(long)r * ld + c] = (float)r;
}
static void run_fill(f4* out, long n4)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)((n4 + 255) / 256);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, out, n4);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, out, n4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("write-only flat float4 fill          %.1f GB/s\n", (double)n4 * 16.0 * reps / (ms * 1e-3) / 1e9);
    const long ld = 65536; const int rows = (int)(n4 * 4 / ld);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_fill_rows, dim3(ld / 256), dim3(256), 0, 0, (float*)out, rows, ld);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_fill_rows, dim3(ld / 256), dim3(256), 0, 0, (float*)out, rows, ld);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("write-only rows of 65536 floats, a lane per column   %.1f GB/s\n", (double)rows * ld * 4.0 * reps / (ms * 1e-3) / 1e9);
}

template <int NV>
static void run_dma(f4* in, f4* out, int nseg, int nvec, long stride4)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream_dma<NV>), dim3(nseg), dim3(64), 0, 0, in, out, nvec, stride4);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream_dma<NV>), dim3(nseg), dim3(64), 0, 0, in, out, nvec, stride4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nseg * (nvec / (NV * 64)) * (NV * 64) * 32.0 * reps;
    printf("copy, LDS-DMA + counted waits NV=%d  %.1f GB/s\n", NV, bytes / (ms * 1e-3) / 1e9);
}

// the textbook copy: consecutive blocks touch consecutive 4 KB, grid-stride
__global__ void __launch_bounds__(256) k_flat(const f4* __restrict__ in, f4* __restrict__ out, long n4)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) out[i] = in[i] * 1.0001f;
}
static void run_flat(f4* in, f4* out, long n4, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, in, out, n4);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, in, out, n4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("flat float4 copy, grid %-7d     %.1f GB/s\n", grid, (double)n4 * 32.0 * reps / (ms * 1e-3) / 1e9);
}


// the RAM sweep's residency (8 - 12 waves per CU, set by dynamic LDS) and two layouts of the segments: private contiguous 80 KB
// (PRIVATE) or chunk-interleaved (chunk k of segment s at (k nseg + s) chunks: waves in step read one contiguous region)
template <int NV, bool ILV>
__global__ void __launch_bounds__(64) k_stream_res(const f4* __restrict__ in, f4* __restrict__ out, int nvec, long stride4, int nseg)
{
    extern __shared__ float pad[];
    const long seg = blockIdx.x;
    const int t = threadIdx.x;
    const int nch = nvec / (NV * 64);
    auto at = [&](int k, int v) -> long { return ILV ? ((long)k * nseg + seg) * (NV * 64) + v * 64 + t : seg * stride4 + (long)(k * NV + v) * 64 + t; };
    f4 regs[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) regs[v] = in[at(0, v)];
    for (int k = 0; k < nch; ++k) {
        f4 cur[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[v] = regs[v];
        if (k + 1 < nch) {
#pragma unroll
            for (int v = 0; v < NV; ++v) regs[v] = in[at(k + 1, v)];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) out[at(k, v)] = cur[v] * 1.0001f;
    }
    if (nvec < 0) pad[t] = 0.0f;
}
template <int NV, bool ILV>
static void run_res(f4* in, f4* out, int nseg, int nvec, long stride4, int lds_bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream_res<NV, ILV>), dim3(nseg), dim3(64), lds_bytes, 0, in, out, nvec, stride4, nseg);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream_res<NV, ILV>), dim3(nseg), dim3(64), lds_bytes, 0, in, out, nvec, stride4, nseg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nseg * (nvec / (NV * 64)) * (NV * 64) * 32.0 * reps;
    printf("copy, %-11s NV=%d, %2d waves per CU  %.1f GB/s\n", ILV ? "interleaved" : "private", NV, 163840 / lds_bytes, bytes / (ms * 1e-3) / 1e9);
}

// the copy split over TWO waves of a block: wave 0 only loads (global -> registers -> LDS), wave 1 only stores (LDS -> global), one
// block barrier per chunk, two LDS slots -- neither wave's in-order memory counter ever holds both kinds of access
template <int NV>
__global__ void __launch_bounds__(128) k_stream_split(const f4* __restrict__ in, f4* __restrict__ out, int nvec, long stride4)
{
    __shared__ f4 ring[2 * NV * 64];
    const long seg = blockIdx.x;
    const f4* src = in + seg * stride4;
    f4* dst = out + seg * stride4;
    const int t = threadIdx.x & 63;
    const bool loader = threadIdx.x < 64;
    const int nch = nvec / (NV * 64);
    f4 regs[NV];
    if (loader) {
#pragma unroll
        for (int v = 0; v < NV; ++v) regs[v] = src[v * 64 + t];
    }
    for (int k = 0; k < nch; ++k) {
        const int sl = k & 1;
        if (loader) {
#pragma unroll
            for (int v = 0; v < NV; ++v) ring[(sl * NV + v) * 64 + t] = regs[v];
            if (k + 1 < nch) {
#pragma unroll
                for (int v = 0; v < NV; ++v) regs[v] = src[((k + 1) * NV + v) * 64 + t];
            }
        }
        __syncthreads();
        if (!loader) {
#pragma unroll
            for (int v = 0; v < NV; ++v) dst[(k * NV + v) * 64 + t] = ring[(sl * NV + v) * 64 + t] * 1.0001f;
        }
    }
}
template <int NV>
static void run_split(f4* in, f4* out, int nseg, int nvec, long stride4)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream_split<NV>), dim3(nseg), dim3(128), 0, 0, in, out, nvec, stride4);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream_split<NV>), dim3(nseg), dim3(128), 0, 0, in, out, nvec, stride4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nseg * (nvec / (NV * 64)) * (NV * 64) * 32.0 * reps;
    printf("copy, loader wave + storer wave NV=%d  %.1f GB/s\n", NV, bytes / (ms * 1e-3) / 1e9);
}

template <int NV, bool WRITE>
static void run(const char* name, f4* in, f4* out, float* sink, int nseg, int nvec, long stride4)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<NV, WRITE>), dim3(nseg), dim3(64), 0, 0, in, out, sink, nvec, stride4);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<NV, WRITE>), dim3(nseg), dim3(64), 0, 0, in, out, sink, nvec, stride4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nseg * nvec * 16.0 * (WRITE ? 2 : 1) * reps;
    printf("%-28s NV=%d  %.1f GB/s\n", name, NV, bytes / (ms * 1e-3) / 1e9);
}

int main()
{
    const int d = 200, nseg = 32768;
    const int tri = d * (d + 1) / 2, nvec = (tri + 3) / 4;
    const long stride4 = nvec;
    f4 *in, *out; float* sink;
    hipMalloc(&in, (size_t)nseg * stride4 * 16);
    hipMalloc(&out, (size_t)nseg * stride4 * 16);
    hipMalloc(&sink, 4);
    hipMemset(in, 0, (size_t)nseg * stride4 * 16);
    hipMemset(out, 0, (size_t)nseg * stride4 * 16);
    run<1, false>("read-only, wave per segment", in, out, sink, nseg, nvec, stride4);
    run<2, false>("read-only, wave per segment", in, out, sink, nseg, nvec, stride4);
    run<4, false>("read-only, wave per segment", in, out, sink, nseg, nvec, stride4);
    run<1, true>("copy, wave per segment", in, out, sink, nseg, nvec, stride4);
    run<2, true>("copy, wave per segment", in, out, sink, nseg, nvec, stride4);
    run<4, true>("copy, wave per segment", in, out, sink, nseg, nvec, stride4);
    run<8, true>("copy, wave per segment", in, out, sink, nseg, nvec, stride4);
    run_dma<1>(in, out, nseg, nvec, stride4);
    run_dma<2>(in, out, nseg, nvec, stride4);
    run_dma<4>(in, out, nseg, nvec, stride4);
    for (int lds : {20000, 13600, 10000, 5000}) {
        run_res<4, false>(in, out, nseg, nvec, stride4, lds);
        run_res<4, true>(in, out, nseg, nvec, stride4, lds);
        run_res<8, false>(in, out, nseg, nvec, stride4, lds);
        run_res<8, true>(in, out, nseg, nvec, stride4, lds);
    }
    run_split<1>(in, out, nseg, nvec, stride4);
    run_split<2>(in, out, nseg, nvec, stride4);
    run_split<4>(in, out, nseg, nvec, stride4);
    run_split<8>(in, out, nseg, nvec, stride4);
    run_fill(out, (long)nseg * stride4);
    run_flat(in, out, (long)nseg * stride4, 2048);
    run_flat(in, out, (long)nseg * stride4, 8192);
    run_flat(in, out, (long)nseg * stride4, (int)(((long)nseg * stride4 + 255) / 256));
    return 0;
}
