// Prototype: fp64 standard normals by a table ziggurat (N layers, table in LDS, failures handled on the spot under a
// wave-uniform branch) against the engine's Box-Muller, same Philox bits budget (one block per 2 normals).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DMHX_REAL64=1 -I advancedmh.jl_amd/csrc -o zig_proto tools/ubench/zig_proto.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "mhx_device_math.h"
using namespace mhx_f64;

// one candidate from 64 bits: layer = low bits, sign = bit 11, u = top 52 bits; table = x[0..N] (x[N] = 0), 8 bytes per layer
template <int N>
__device__ inline bool zig_try(const double* tab, mhx_u32 lo, mhx_u32 hi, double& x, mhx_u32& layer, double& xi, double& xi1)
{
    layer = lo & (N - 1);
    const mhx_u64 k = (((mhx_u64)hi << 32) | lo) >> 12;
    const double u = mhx_u2d(0x3ff0000000000000ull | k) - 1.0;
    xi = tab[layer]; xi1 = tab[layer + 1];
    const double ax = u * xi;
    x = mhx_u2d(mhx_d2u(ax) ^ ((mhx_u64)(lo & 2048u) << 52));
    return ax < xi1;
}
template <int N>
__device__ __noinline__ double zig_slow(const mhx_philox_key& ks, mhx_u32 id, mhx_u32 step, mhx_u32 blk, const double* tab, double x, mhx_u32 layer, double xi, double xi1)
{
    const double r = tab[1];
    for (mhx_u32 att = 1;; ++att) {
        const mhx_u32x4 v = mhx_philox(ks, id, att << 16, step, blk);
        if (layer == 0) {                                        // the tail beyond r (Marsaglia)
            const double xx = -mhx_log_pos(mhx_u01_open(v.x, v.y)) / r;
            const double yy = -mhx_log_pos(mhx_u01_open(v.z, v.w));
            if (yy + yy >= xx * xx) return mhx_u2d(mhx_d2u(r + xx) | (mhx_d2u(x) & 0x8000000000000000ull));
            continue;
        }
        const double f0 = mhx_exp(-0.5 * (xi * xi - x * x)), f1 = mhx_exp(-0.5 * (xi1 * xi1 - x * x));
        if (f1 + mhx_u01_half(v.z, v.w) * (f0 - f1) < 1.0) return x;
        if (zig_try<N>(tab, v.x, v.y, x, layer, xi, xi1)) return x;
    }
}
template <int N>
__device__ inline void zig_pair(const mhx_philox_key& ks, mhx_u32 id, mhx_u32 step, mhx_u32 blk, const double* tab, double& n0, double& n1)
{
    const mhx_u32x4 w = mhx_philox(ks, id, 0u, step, blk);
    double x0, x1, a0, b0, a1, b1; mhx_u32 l0, l1;
    const bool ok0 = zig_try<N>(tab, w.x, w.y, x0, l0, a0, b0);
    const bool ok1 = zig_try<N>(tab, w.z, w.w, x1, l1, a1, b1);
    if (__ballot(!(ok0 && ok1))) {                               // wave-uniform: some lane left the rectangles
        if (!ok0) x0 = zig_slow<N>(ks, id, step, blk, tab, x0, l0, a0, b0);
        if (!ok1) x1 = zig_slow<N>(ks, id, step, blk | 0x800000u, tab, x1, l1, a1, b1);
    }
    n0 = x0; n1 = x1;
}

template <int MODE, int N>
__global__ void __launch_bounds__(256) k(double* out, const double* gtab, int iters)
{
    __shared__ double tab[N > 0 ? N + 1 : 1];
    if (N > 0) { for (int e = threadIdx.x; e <= N; e += 256) tab[e] = gtab[e]; __syncthreads(); }
    const mhx_philox_key ks = mhx_philox_schedule(0x1234567ull);
    const mhx_u32 id = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 13; ++b) {
            double n[4];
            if (MODE == 0) mhx_normal4(ks, id, 0u, (mhx_u32)it, 0u, (mhx_u32)b, n);
            else { zig_pair<N>(ks, id, it, 2 * b, tab, n[0], n[1]); zig_pair<N>(ks, id, it, 2 * b + 1, tab, n[2], n[3]); }
            acc = fma(n[0], n[1], acc); acc = fma(n[2], n[3], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[id] = acc;
}

static void make_table(int N, std::vector<double>& tab)
{
    // solve for r: equal-area layers, x[N] = 0   (long double bisection is plenty for a timing prototype)
    auto f = [](long double x) { return expl(-0.5L * x * x); };
    auto tailarea = [](long double r) { return sqrtl(acosl(-1.0L) / 2) * erfcl(r / sqrtl(2.0L)); };
    long double lo = 2.0L, hi = 6.0L;
    std::vector<long double> x(N + 1);
    for (int itb = 0; itb < 200; ++itb) {
        long double r = 0.5L * (lo + hi), v = r * f(r) + tailarea(r);
        x[1] = r; bool bad = false;
        for (int i = 1; i < N; ++i) { long double y = v / x[i] + f(x[i]); if (y >= 1.0L) { bad = true; break; } x[i + 1] = sqrtl(-2.0L * logl(y)); }
        if (bad) lo = r; else { long double top = v / x[N - 1] + f(x[N - 1]); (void)top; if (x[N] > 0) hi = r; else lo = r; }
        // crude: want x[N] -> 0: if the recursion ends above 0 r is too large
    }
    long double r = 0.5L * (lo + hi), v = r * f(r) + tailarea(r);
    x[0] = v / f(r); x[1] = r;
    for (int i = 1; i < N; ++i) { long double y = v / x[i] + f(x[i]); x[i + 1] = y < 1.0L ? sqrtl(-2.0L * logl(y)) : 0.0L; }
    x[N] = 0;
    tab.resize(N + 1);
    for (int i = 0; i <= N; ++i) tab[i] = (double)x[i];
    double fastp = 0; for (int i = 0; i < N; ++i) fastp += tab[i + 1] / tab[i] / N;
    printf("N=%d r=%.6Lf fast-path probability %.5f\n", N, r, fastp);
}

template <int MODE, int N>
static void run(const char* name, double* d_out, const double* d_tab)
{
    const int grid = 1024, iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, N><<<grid, 256>>>(d_out, d_tab, 10);
    hipEventRecord(e0);
    k<MODE, N><<<grid, 256>>>(d_out, d_tab, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double normals = (double)grid * 256 * iters * 52;
    printf("%-28s %.3f ms  %.3e normals/s  (%.1f SIMD-cycles per wave per 4 normals)\n", name, ms, normals / (ms * 1e-3),
           ms * 1e-3 * 2.4e9 * 1024 / ((double)grid * 4 * iters * 13));
}

int main()
{
    double* d_out; hipMalloc(&d_out, 1024 * 256 * sizeof(double));
    double* d_tab; hipMalloc(&d_tab, 16400 * sizeof(double));
    run<0, 0>("Box-Muller (engine, fp64)", d_out, d_tab);
    std::vector<double> tab;
    make_table(1024, tab); hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); run<1, 1024>("ziggurat 1024 layers", d_out, d_tab);
    make_table(2048, tab); hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); run<1, 2048>("ziggurat 2048 layers", d_out, d_tab);
    make_table(4096, tab); hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); run<1, 4096>("ziggurat 4096 layers", d_out, d_tab);
    make_table(8192, tab); hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice); run<1, 8192>("ziggurat 8192 layers", d_out, d_tab);
    return 0;
}
