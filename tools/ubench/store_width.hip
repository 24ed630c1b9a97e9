// Micro-benchmark (round 4): what does the RECORD of the C2 kernel pay for -- the number of store instructions or their bytes?
// The kernel of DESIGN 6.1 issues 52 buffer_store_dwordx2 per wave-step (2 lanes per chain, 32 chains per wave, rows of the
// [slot][dim+1][chains] tensor) at ONE wave per SIMD and loses 16 % to them.  A 2 x 2 transpose between neighbouring lanes (two
// chains x two rows; v_cndmask_b32_dpp quad_perm:[1,0,3,2]) would turn them into 26 dwordx4 stores of the same bytes.  This
// probe runs a stand-in wave-step -- NV independent fp64 fmas, then the stores -- in five forms:
//   0  no stores                 1  52 x dwordx2 (the kernel's shape)       2  26 x dwordx4, same bytes, no transpose
//   3  26 x dwordx4 + the transposes (4 DPP selects per store)              4  13 x (two dwordx4 per lane pair row)  [n/a]
// hipcc --offload-arch=gfx950 -O3 -o store_width store_width.hip && ./store_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHAINS 65536
#define DIM 100
#define NBL 13
#define STEPS 250
#define NV 1800

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned swap1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ double swapd(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = swap1((unsigned)b), hi = swap1((unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int MODE, bool NT>
__global__ void __launch_bounds__(256) k_step(double* __restrict__ out, const double s, const int steps)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int cw = lane & 31, l = lane >> 5;
    const long c = wave * 32 + cw;
    const long ld = CHAINS;
    double x[NBL][4];
#pragma unroll
    for (int i = 0; i < NBL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[i][j] = (double)(lane + 4 * i + j) * 1e-3;
    if (threadIdx.x == 0) lds[0] = s;
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
        // the stand-in for generation + candidate + accept: NV fmas over the 52 state registers (independent chains of length NV / 52)
#pragma unroll 1
        for (int r = 0; r < NV / 52; ++r) {
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[i][j] = __builtin_fma(x[i][j], s, 1e-9);
        }
        double* slot = out + (long)t * (DIM + 1) * ld;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * (l + 2 * i) + j;
                    if (k < DIM) { if (NT) __builtin_nontemporal_store(x[i][j], slot + (long)k * ld + c); else slot[(long)k * ld + c] = x[i][j]; }
                }
        } else if (MODE == 2 || MODE == 3) {
            const bool odd = (cw & 1) != 0;
            const long c2 = c & ~1L;
#pragma unroll
            for (int i = 0; i < NBL; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    // rows A = (i, 2 jp), B = (i, 2 jp + 1): the even lane writes row A of chains (c, c + 1), the odd lane row B
                    const double a = x[i][2 * jp], b = x[i][2 * jp + 1];
                    d2 v;
                    if (MODE == 3) {
                        const double sb = swapd(b), sa = swapd(a);
                        v.x = odd ? sb : a;
                        v.y = odd ? b : sa;
                    } else {
                        v.x = a; v.y = b;
                    }
                    const int k = 4 * (l + 2 * i) + 2 * jp + (odd ? 1 : 0);
                    if (k < DIM) { if (NT) __builtin_nontemporal_store(v, (d2*)(slot + (long)k * ld + c2)); else *(d2*)(slot + (long)k * ld + c2) = v; }
                }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NBL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += x[i][j];
    if (acc == 123.456) out[c] = acc;
}

template <int MODE, bool NT>
static float run(double* out, int steps)
{
    hipFuncSetAttribute((const void*)k_step<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_step<MODE, NT>), dim3(CHAINS * 2 / 256), dim3(256), 100 * 1024, 0, out, 0.999999, steps);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main()
{
    double* out;
    const size_t bytes = (size_t)STEPS * (DIM + 1) * CHAINS * sizeof(double);
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(out, 0, bytes);
    const float t0 = run<0, true>(out, STEPS), t1 = run<1, true>(out, STEPS), t2 = run<2, true>(out, STEPS), t3 = run<3, true>(out, STEPS);
    const float u1 = run<1, false>(out, STEPS), u3 = run<3, false>(out, STEPS);
    const double gb = (double)STEPS * DIM * CHAINS * 8 / 1e9;
    printf("wave-step = %d fp64 fmas + record of %d x 8 B per chain, %d steps, %d chains, 1 wave per SIMD\n", NV, DIM, STEPS, CHAINS);
    printf("no stores            : %.3f ms\n", t0);
    printf("52 x dwordx2, nt     : %.3f ms  (+%.1f %%, %.2f TB/s)\n", t1, 100.0 * (t1 - t0) / t0, gb / t1);
    printf("26 x dwordx4, nt     : %.3f ms  (+%.1f %%, %.2f TB/s)\n", t2, 100.0 * (t2 - t0) / t0, gb / t2);
    printf("26 x dwordx4+DPP, nt : %.3f ms  (+%.1f %%, %.2f TB/s)\n", t3, 100.0 * (t3 - t0) / t0, gb / t3);
    printf("52 x dwordx2, plain  : %.3f ms  (+%.1f %%, %.2f TB/s)\n", u1, 100.0 * (u1 - t0) / t0, gb / u1);
    printf("26 x dwordx4+DPP, pl.: %.3f ms  (+%.1f %%, %.2f TB/s)\n", u3, 100.0 * (u3 - t0) / t0, gb / u3);
    return 0;
}
