// Micro-benchmark: what does one grid-wide synchronisation cost on MI355X, as a dependent kernel boundary and as an
// in-kernel barrier of a persistent kernel?  (The ensemble sampler needs one per half-step: src/emcee.jl's sweep in its
// parallel half-split form.)  Every phase each block rewrites a 4 KB slab that a block on ANOTHER XCD reads in the next
// phase -- the communication pattern of the stretch move's partner rows -- so the in-kernel barrier needs the agent-scope
// release / acquire (per-XCD L2s are not coherent with each other).
//   launches : N dependent launches of the phase kernel on one stream
//   barrier  : ONE persistent launch, N phases separated by an XCD-hierarchical counter barrier (per-XCD arrival
//              counters, XCD leader -> top counter -> per-XCD generation; relaxed polls, one release before arriving, one
//              acquire after leaving)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct bar_state { unsigned xcd_count[8 * 16]; unsigned xcd_gen[8 * 16]; unsigned top_count; unsigned pad[15]; unsigned timeout; };

__device__ void phase_work(float* slabs, int nblocks, int phase, float* sink)
{
    // read the slab of the block "across" the chip (b + nblocks/2 + 1 lands on another XCD), write our own
    const int b = blockIdx.x, other = (b + nblocks / 2 + 1) % nblocks;
    const float4* src = (const float4*)(slabs + (size_t)((phase & 1) * nblocks + other) * 1024);
    float4* dst = (float4*)(slabs + (size_t)(((phase + 1) & 1) * nblocks + b) * 1024);
    float4 v = src[threadIdx.x];
    v.x += 1.0f; v.y += v.x; v.z += v.y; v.w += v.z;
    dst[threadIdx.x] = v;
    if (v.x == -1.0f) *sink = v.w;
}

__global__ void __launch_bounds__(256) k_phase(float* slabs, int nblocks, int phase, float* sink) { phase_work(slabs, nblocks, phase, sink); }

__device__ void grid_barrier(bar_state* st, const int nblocks, const unsigned epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const int xcd = blockIdx.x & 7;                       // observed placement: block b runs on XCD b % 8 (speed only)
        const unsigned per_xcd = (nblocks + 7 - xcd) / 8;     // blocks with this residue
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(&st->xcd_count[xcd * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == epoch * per_xcd + per_xcd - 1) {             // last of this XCD: go to the top counter
            const unsigned tt = __hip_atomic_fetch_add(&st->top_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tt == epoch * 8 + 7) {                        // last XCD: release every XCD
                for (int x = 0; x < 8; ++x) __hip_atomic_store(&st->xcd_gen[x * 16], epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned spins = 0;
        while (__hip_atomic_load(&st->xcd_gen[xcd * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) { st->timeout = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_persistent(float* slabs, int nblocks, int nphases, bar_state* st, float* sink)
{
    for (int p = 0; p < nphases; ++p) {
        phase_work(slabs, nblocks, p, sink);
        grid_barrier(st, nblocks, (unsigned)p);
        if (__hip_atomic_load(&st->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // never spin twice
    }
}

int main()
{
    const int N = 2000;
    for (int nblocks : {256, 512, 1024}) {
        float *slabs, *sink;
        bar_state* st;
        CHECK(hipMalloc(&slabs, (size_t)2 * nblocks * 1024 * sizeof(float)));
        CHECK(hipMalloc(&sink, 4));
        CHECK(hipMalloc(&st, sizeof(bar_state)));
        CHECK(hipMemset(slabs, 0, (size_t)2 * nblocks * 1024 * sizeof(float)));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms_launch = 0, ms_bar = 0;
        for (int rep = 0; rep < 2; ++rep) {                   // second repetition is the measurement
            CHECK(hipEventRecord(e0));
            for (int p = 0; p < N; ++p) hipLaunchKernelGGL(k_phase, dim3(nblocks), dim3(256), 0, 0, slabs, nblocks, p, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms_launch, e0, e1));
            CHECK(hipMemset(st, 0, sizeof(bar_state)));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_persistent, dim3(nblocks), dim3(256), 0, 0, slabs, nblocks, N, st, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms_bar, e0, e1));
        }
        bar_state h;
        CHECK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
        std::vector<float> chk(4);
        CHECK(hipMemcpy(chk.data(), slabs, 16, hipMemcpyDeviceToHost));
        printf("blocks=%5d  dependent launches: %.3f us per phase   persistent + grid barrier: %.3f us per phase%s   (slab[0].x = %.0f)\n",
               nblocks, ms_launch * 1e3 / N, ms_bar * 1e3 / N, h.timeout ? "  [BARRIER TIMED OUT]" : "", chk[0]);
        hipFree(slabs); hipFree(sink); hipFree(st);
    }
    return 0;
}
