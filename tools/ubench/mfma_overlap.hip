// Micro-benchmark: does the f32-input / f64 MFMA run BESIDE the VALU on gfx950, or do they share the pipe?
// Per wave: (A) NM independent-accumulator MFMAs, (B) NV v_fma, (C) both interleaved in one instruction stream.
// Run at 1, 2 and 4 waves per SIMD.  If time(C) ~ max(A, B) the pipes overlap, if ~ A + B they do not.
// hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2000
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// 4 MFMA (2 accumulators each used twice) + NVAL independent v_fma per iteration
template <int MODE, int NVAL, int KIND>
__global__ void __launch_bounds__(256) k(unsigned long long* out, float seed)
{
    float a = seed + threadIdx.x, b = 1.0f + seed;
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    d4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
    f16v g0 = {0}, g1 = {0};
    double da = a, db = b;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (MODE & 1) {
                if (KIND == 0) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0); }
                if (KIND == 1) { e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, e1, 0, 0, 0); }
                if (KIND == 2) { g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, g0, 0, 0, 0); g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, g1, 0, 0, 0); }
            }
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < NVAL / 2; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = c0[0] + c1[0] + (float)(e0[0] + e1[0]) + g0[0] + g1[0];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[100000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NVAL, int KIND>
static void report(const char* name, unsigned long long* d_out)
{
    for (int w = 1; w <= 4; w *= 2) {
        hipEvent_t e0, e1;
        float ms[3];
        for (int mode = 1; mode <= 3; ++mode) {
            const int grid = 256 * w;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 1) k<1, NVAL, KIND><<<grid, 256>>>(d_out, 1.0f);
                if (mode == 2) k<2, NVAL, KIND><<<grid, 256>>>(d_out, 1.0f);
                if (mode == 3) k<3, NVAL, KIND><<<grid, 256>>>(d_out, 1.0f);
            };
            launch(); launch();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode - 1], e0, e1);
            ms[mode - 1] /= 5;
        }
        // cycles per iteration per SIMD at 2.4 GHz: w waves per SIMD each run ITERS iterations
        const double f = 2.4e6 / ITERS;
        printf("%-28s %d waves/SIMD: 4 MFMA %.1f cyc | %d v_fma %.1f cyc | both %.1f cyc  (sum %.1f, max %.1f) per wave-iteration-slot\n", name, w,
               ms[0] * f / w, NVAL, ms[1] * f / w, ms[2] * f / w, (ms[0] + ms[1]) * f / w, (ms[0] > ms[1] ? ms[0] : ms[1]) * f / w);
    }
}

int main()
{
    unsigned long long* d_out;
    hipMalloc(&d_out, 200000 * sizeof(unsigned long long));
    report<32, 0>("mfma_f32_16x16x4 + 32 fma", d_out);
    report<64, 0>("mfma_f32_16x16x4 + 64 fma", d_out);
    report<64, 2>("mfma_f32_32x32x2 + 64 fma", d_out);
    report<64, 1>("mfma_f64_16x16x4 + 64 fma", d_out);
    report<128, 1>("mfma_f64_16x16x4 + 128 fma", d_out);
    return 0;
}
