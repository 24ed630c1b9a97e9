// Micro-benchmark: cycles per wave64 VALU instruction on gfx950 with ONE wave per SIMD (the
// occupancy the register-resident RWMH kernel runs at).  Each kernel issues 8 independent chains of
// one instruction inside an unrolled loop; cycles/instr = s_memtime delta / instruction count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, DECL, BODY, SINK)                                                    \
    __global__ void __launch_bounds__(64) NAME(unsigned long long* out, unsigned seed)    \
    {                                                                                     \
        DECL                                                                              \
        unsigned long long t0 = __builtin_readcyclecounter();                             \
        for (int i = 0; i < ITERS; ++i) { BODY BODY BODY BODY }                           \
        unsigned long long t1 = __builtin_readcyclecounter();                             \
        SINK                                                                              \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                  \
    }

#define FDECL float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; float b = 1.0001f + threadIdx.x, c = 0.5f;
#define FSINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[1000000] = 1;
#define UDECL unsigned a0 = seed + threadIdx.x, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; unsigned b = 0xD2511F53u + threadIdx.x;
#define USINK if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 12345u) out[1000000] = 1;

#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_fma, FDECL, REP8(FMA), FSINK)
#define FMAK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800347" : "+v"(a##i) : "v"(b));
KERNEL(k_fmaak, FDECL, REP8(FMAK), FSINK)
#define MULF(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul, FDECL, REP8(MULF), FSINK)
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_xor, UDECL, REP8(XOR), USINK)
#define BITOP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a##i) : "v"(b), "s"(seed));
KERNEL(k_bitop3, UDECL, REP8(BITOP), USINK)
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul_lo, UDECL, REP8(MULLO), USINK)
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul_hi, UDECL, REP8(MULHI), USINK)
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul_u24, UDECL, REP8(MUL24), USINK)
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i));
KERNEL(k_sqrt, FDECL, REP8(SQRT), FSINK)
#define CVT(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i));
KERNEL(k_cvt, FDECL, REP8(CVT), FSINK)
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
KERNEL(k_cndmask, UDECL, REP8(CND), USINK)
#define ACCW(i) asm volatile("v_accvgpr_write_b32 a" #i ", %0\n v_accvgpr_read_b32 %0, a" #i : "+v"(a##i) : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
KERNEL(k_acc_wr_rd, UDECL, REP8(ACCW), USINK)

// fp64 (the reference computes in Float64): the f64 engine's VALU ceiling comes from these
#define DDECL double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; double b = 1.0001 + threadIdx.x, c = 0.5;
#define DSINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0) out[1000000] = 1;
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(k_fma64, DDECL, REP8(FMA64), DSINK)
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_mul64, DDECL, REP8(MUL64), DSINK)
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(k_add64, DDECL, REP8(ADD64), DSINK)
#define RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
KERNEL(k_rcp64, DDECL, REP8(RCP64), DSINK)
#define SQRT64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a##i));
KERNEL(k_sqrt64, DDECL, REP8(SQRT64), DSINK)
#define RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a##i));
KERNEL(k_rsq64, DDECL, REP8(RSQ64), DSINK)

// 64-bit products need register pairs: use plain C++ with an opaque barrier instead
__global__ void __launch_bounds__(64) k_mad_u64(unsigned long long* out, unsigned seed)
{
    unsigned a[8];
    for (int j = 0; j < 8; ++j) a[j] = seed + j + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned long long p;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a[j]), "s"(0xD2511F53u) : "vcc");
                a[j] = (unsigned)(p >> 32) ^ (unsigned)p;
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned x = 0;
    for (int j = 0; j < 8; ++j) x ^= a[j];
    if (x == 12345u) out[1000000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// packed fp32
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64) k_pk_fma(unsigned long long* out, unsigned seed)
{
    float2v a[8];
    for (int j = 0; j < 8; ++j) a[j] = float2v{(float)(seed + j), (float)(seed + j + 8)};
    float2v b = {1.0001f + threadIdx.x, 1.0002f}, c = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float x = 0;
    for (int j = 0; j < 8; ++j) x += a[j].x + a[j].y;
    if (x == 12345.f) out[1000000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <class K> static void run(const char* name, K k, int nblocks, unsigned long long* d_out, double extra = 1.0)
{
    std::vector<unsigned long long> h(nblocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d_out, 1u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_out, nblocks * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= nblocks;
    const double ninstr = (double)ITERS * 32 * extra;
    printf("%-14s blocks=%5d  memtime ticks/instr=%7.3f   ns/instr=%7.4f  (kernel %.3f ms)\n", name, nblocks,
           cyc / ninstr, ms * 1e6 / ninstr, ms);
}

int main()
{
    unsigned long long* d_out; hipMalloc(&d_out, 8 * 1000001);
    for (int nb : {1024, 2048, 4096, 8192}) {
        printf("--- %d waves (%d per SIMD)\n", nb, nb / 1024);
        run("v_fma_f32", k_fma, nb, d_out);
        run("v_fmaak_f32", k_fmaak, nb, d_out);
        run("v_mul_f32", k_mul, nb, d_out);
        run("v_pk_fma_f32", k_pk_fma, nb, d_out);
        run("v_xor_b32", k_xor, nb, d_out);
        run("v_bitop3_b32", k_bitop3, nb, d_out);
        run("v_mul_lo_u32", k_mul_lo, nb, d_out);
        run("v_mul_hi_u32", k_mul_hi, nb, d_out);
        run("v_mul_u32_u24", k_mul_u24, nb, d_out);
        run("mad_u64+xor", k_mad_u64, nb, d_out);
        run("v_fma_f64", k_fma64, nb, d_out);
        run("v_mul_f64", k_mul64, nb, d_out);
        run("v_add_f64", k_add64, nb, d_out);
        run("v_rcp_f64", k_rcp64, nb, d_out);
        run("v_sqrt_f64", k_sqrt64, nb, d_out);
        run("v_rsq_f64", k_rsq64, nb, d_out);
        run("v_sqrt_f32", k_sqrt, nb, d_out);
        run("v_cvt_f32_u32", k_cvt, nb, d_out);
        run("v_cndmask", k_cndmask, nb, d_out);
        run("acc wr+rd", k_acc_wr_rd, nb, d_out);
    }
    return 0;
}
