// copy_probe2.hip -- round 4 probes of the looping read + write stream that bounds the RAM sweep:
//   (a) does the DISTANCE between a wave's read stream and its write stream matter?  (the sweep's two buffers of a chain sit side by
//       side: the write trails the read by one padded triangle = a fixed distance for every wave -- same HBM channel / bank phase?)
//       k_pair: wave s reads segment s of `buf` and writes segment s of the same array `delta` bytes further on
//   (b) does batching the reads and the writes of a CU into long phases matter (DRAM read/write turnaround)?
//       k_phase: a block of W waves loads P KB into LDS (loads only), barrier, stores P KB (stores only), for its W segments in turn
//   hipcc --offload-arch=gfx950 -O3 -o copy_probe2 copy_probe2.hip && ./copy_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ void __launch_bounds__(64) k_pair(const f4* __restrict__ buf, f4* __restrict__ wbuf, int nvec, long stride4)
{
    const long seg = blockIdx.x;
    const f4* src = buf + seg * stride4;
    f4* dst = wbuf + seg * stride4;
    const int t = threadIdx.x;
    const int nch = nvec / (NV * 64);
    f4 regs[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) regs[v] = src[v * 64 + t];
    for (int k = 0; k < nch; ++k) {
        f4 cur[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[v] = regs[v];
        if (k + 1 < nch) {
#pragma unroll
            for (int v = 0; v < NV; ++v) regs[v] = src[((k + 1) * NV + v) * 64 + t];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[(k * NV + v) * 64 + t] = cur[v] * 1.0001f;
    }
}

// W waves per block; every round the block moves ROUND KB: all waves load their share into LDS, barrier, all waves store it
template <int W, int NV>
__global__ void __launch_bounds__(64 * W) k_phase(const f4* __restrict__ in, f4* __restrict__ out, int nvec, long stride4)
{
    extern __shared__ f4 ring[];                       // [W][NV * 64]
    const int wv = threadIdx.x >> 6, t = threadIdx.x & 63;
    const long seg = (long)blockIdx.x * W + wv;
    const f4* src = in + seg * stride4;
    f4* dst = out + seg * stride4;
    f4* mine = ring + wv * (NV * 64);
    const int nch = nvec / (NV * 64);
    for (int k = 0; k < nch; ++k) {
        f4 r[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) r[v] = src[(k * NV + v) * 64 + t];
#pragma unroll
        for (int v = 0; v < NV; ++v) mine[v * 64 + t] = r[v];
        __syncthreads();                               // every wave of the CU has finished its loads of this round
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[(k * NV + v) * 64 + t] = mine[v * 64 + t] * 1.0001f;
        __syncthreads();                               // ... and its stores have been issued before anyone loads again
    }
}

// the reference points: a one-shot copy (every thread one load, then one store: chip-wide the loads of a launch precede its stores)
__global__ void __launch_bounds__(256) k_oneshot(const f4* __restrict__ in, f4* __restrict__ out, long n4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = in[i] * 1.0001f;
}
struct one_ctx { f4 *in, *out; long n4; };
static void launch_oneshot(void* p)
{
    one_ctx* c = (one_ctx*)p;
    hipLaunchKernelGGL(k_oneshot, dim3((unsigned)((c->n4 + 255) / 256)), dim3(256), 0, 0, c->in, c->out, c->n4);
}

static float timeit(void (*launch)(void*), void* ctx)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) launch(ctx);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) launch(ctx);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10.0f;
}

struct pair_ctx { f4* buf; long delta4; int nseg, nvec; long stride4; };
template <int NV> static void launch_pair(void* p)
{
    pair_ctx* c = (pair_ctx*)p;
    hipLaunchKernelGGL((k_pair<NV>), dim3(c->nseg), dim3(64), 0, 0, c->buf, c->buf + c->delta4, c->nvec, c->stride4);
}
struct phase_ctx { f4 *in, *out; int nseg, nvec; long stride4; };
template <int W, int NV> static void launch_phase(void* p)
{
    phase_ctx* c = (phase_ctx*)p;
    hipLaunchKernelGGL((k_phase<W, NV>), dim3(c->nseg / W), dim3(64 * W), W * NV * 64 * 16, 0, c->in, c->out, c->nvec, c->stride4);
}
template <int W, int NV> static void run_phase(phase_ctx& c)
{
    hipFuncSetAttribute((const void*)k_phase<W, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, W * NV * 64 * 16);
    const float ms = timeit(launch_phase<W, NV>, &c);
    const double bytes = (double)c.nseg * (c.nvec / (NV * 64)) * (NV * 64) * 32.0;
    printf("phases: %2d waves per block x %2d KB per wave = %4d KB per round   %.1f GB/s\n", W, NV, W * NV, bytes / (ms * 1e-3) / 1e9);
}

int main()
{
    const int d = 200, nseg = 32768;
    const int tri = d * (d + 1) / 2, nvec = (tri + 3) / 4;                 // 5025 float4 = 80.4 KB per segment
    const long stride4 = nvec;
    const size_t seg_bytes = (size_t)stride4 * 16, total = (size_t)nseg * seg_bytes;
    f4* big;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    if (hipMalloc(&big, 2 * total + (512 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(big, 0, 2 * total + (512 << 20));
    // (a) read stream at `big`, write stream at big + total + extra: extra sweeps the channel / bank phase between the two streams
    const long extras[] = {0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 65536 + 256, 1 << 20, (1 << 20) + 4096, 3 << 20};
    for (long e : extras) {
        pair_ctx c{big, (long)((total + (size_t)e) / 16), nseg, nvec, stride4};
        const float ms = timeit(launch_pair<8>, &c);
        const double bytes = (double)nseg * (nvec / (8 * 64)) * (8 * 64) * 32.0;
        printf("pair: write stream = read stream + %zu + %-8ld bytes   %.1f GB/s\n", total, e, bytes / (ms * 1e-3) / 1e9);
    }
    // the sweep's own geometry: the write trails the read by ONE segment (the two buffers of a chain side by side): stride = 2 segments
    for (long pad : {0L, 256L, 1024L, 4096L}) {
        const long st4 = 2 * stride4 + pad / 16;
        pair_ctx c{big, stride4 + pad / 32 / 16 * 16, nseg, nvec, st4};
        const float ms = timeit(launch_pair<8>, &c);
        const double bytes = (double)nseg * (nvec / (8 * 64)) * (8 * 64) * 32.0;
        printf("pair: side by side, chain stride 2 x %zu + %-5ld bytes   %.1f GB/s\n", seg_bytes, pad, bytes / (ms * 1e-3) / 1e9);
    }
    {
        one_ctx oc{big, big + total / 16, (long)nseg * stride4};
        const float ms = timeit(launch_oneshot, &oc);
        printf("one-shot copy (a load and a store per thread)   %.1f GB/s\n", (double)oc.n4 * 32.0 / (ms * 1e-3) / 1e9);
    }
    // (b) phases
    phase_ctx pc{big, big + total / 16, nseg, nvec, stride4};
    run_phase<4, 4>(pc);
    run_phase<8, 4>(pc);
    run_phase<16, 4>(pc);
    run_phase<8, 8>(pc);
    run_phase<16, 8>(pc);
    run_phase<16, 4>(pc);
    return 0;
}
