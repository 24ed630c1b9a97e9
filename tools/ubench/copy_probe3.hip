// copy_probe3.hip -- how many bytes must a CU keep in flight for a looping read + write stream?  (round 4; follows copy_probe2:
// the looping copy is not DRAM-limited -- its TCC stall counters are below the one-shot copy's -- it keeps fewer reads in flight)
//   k_deep<NV, DEPTH>: a wave per 80 KB segment; loads run DEPTH chunks of NV KB ahead of the stores through an LDS ring filled by
//   LDS-DMA (global_load_lds_dwordx4: no VGPR staging), hand-counted vmcnt so that the wave never waits for a store
//   hipcc --offload-arch=gfx950 -O3 -o copy_probe3 copy_probe3.hip && ./copy_probe3
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NV, int DEPTH>
__global__ void __launch_bounds__(64) k_deep(const f4* __restrict__ in, f4* __restrict__ out, int nvec, long stride4)
{
    extern __shared__ f4 ring[];                                     // [DEPTH + 1][NV * 64]
    const long seg = blockIdx.x;
    const f4* src = in + seg * stride4;
    f4* dst = out + seg * stride4;
    const int t = threadIdx.x;
    const int nch = nvec / (NV * 64);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    constexpr int SL = DEPTH + 1;
#pragma unroll
    for (int p = 0; p < DEPTH; ++p)
        if (p < nch) {
#pragma unroll
            for (int v = 0; v < NV; ++v) glds16(src + (p * NV + v) * 64 + t, __builtin_amdgcn_readfirstlane(base + (unsigned)((p * NV + v) * 1024)));
        }
    for (int k = 0; k < nch; ++k) {
        const int sl = k % SL;
        if (k + DEPTH < nch) {
            const int nl = (k + DEPTH) % SL;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                glds16(src + ((k + DEPTH) * NV + v) * 64 + t, __builtin_amdgcn_readfirstlane(base + (unsigned)((nl * NV + v) * 1024)));
            // in order: ... S(k-1) [NV], L(k+1..k+DEPTH) -- chunk k has landed when at most NV (stores of k-1) + DEPTH * NV ops remain
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH + 1) * NV > 63 ? 63 : (DEPTH + 1) * NV) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f4 x = ring[(sl * NV + v) * 64 + t];
            dst[(k * NV + v) * 64 + t] = x * 1.0001f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int NV, int DEPTH>
static void run(f4* in, f4* out, int nseg, int nvec, long stride4)
{
    const int lds = (DEPTH + 1) * NV * 1024;
    hipFuncSetAttribute((const void*)k_deep<NV, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_deep<NV, DEPTH>), dim3(nseg), dim3(64), lds, 0, in, out, nvec, stride4);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k_deep<NV, DEPTH>), dim3(nseg), dim3(64), lds, 0, in, out, nvec, stride4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nseg * (nvec / (NV * 64)) * (NV * 64) * 32.0 * 10;
    const int waves = 163840 / lds > 32 ? 32 : 163840 / lds;
    printf("deep ring: %2d KB chunks, loads %d chunks ahead (%3d KB of LDS per wave, %2d waves per CU, %4d KB of loads in flight per CU)   %.1f GB/s\n",
           NV, DEPTH, lds / 1024, waves, waves * DEPTH * NV, bytes / (ms * 1e-3) / 1e9);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int d = 200, nseg = 32768;
    const int tri = d * (d + 1) / 2, nvec = (tri + 3) / 4;
    const long stride4 = nvec;
    const size_t total = (size_t)nseg * stride4 * 16;
    f4 *in, *out;
    if (hipMalloc(&in, total) != hipSuccess || hipMalloc(&out, total) != hipSuccess) return 1;
    hipMemset(in, 0, total); hipMemset(out, 0, total);
    run<1, 1>(in, out, nseg, nvec, stride4);
    run<1, 3>(in, out, nseg, nvec, stride4);
    run<1, 7>(in, out, nseg, nvec, stride4);
    run<2, 1>(in, out, nseg, nvec, stride4);
    run<2, 3>(in, out, nseg, nvec, stride4);
    run<2, 7>(in, out, nseg, nvec, stride4);
    run<4, 1>(in, out, nseg, nvec, stride4);
    run<4, 2>(in, out, nseg, nvec, stride4);
    run<4, 3>(in, out, nseg, nvec, stride4);
    run<4, 7>(in, out, nseg, nvec, stride4);
    run<8, 1>(in, out, nseg, nvec, stride4);
    run<8, 2>(in, out, nseg, nvec, stride4);
    run<8, 3>(in, out, nseg, nvec, stride4);
    run<16, 1>(in, out, nseg, nvec, stride4);
    run<16, 2>(in, out, nseg, nvec, stride4);
    return 0;
}
