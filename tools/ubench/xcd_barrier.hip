// Micro-benchmark (round 3): the grid barrier of tools/ubench/grid_barrier.hip let EVERY block run the agent-scope release
// (a write-back of its XCD's whole L2) before arriving: 12.5 us at 256 blocks.  MI355X_MICROARCH.md's `barrier-xcd` row
// (4.1 - 4.7 us) releases ONCE per XCD: a block only waits for its own stores to reach the L2 it shares with the other
// blocks of its XCD (vmcnt(0)) and arrives on the XCD's counter; the LAST arriver of the XCD runs the release fence for all
// of them, arrives on the top counter, and the last XCD publishes a generation word per XCD; every block acquires after it
// has seen its XCD's generation.  This file measures that form against dependent launches with the ensemble sampler's
// traffic: each phase every block writes `slab_bytes` (C3 fp64: 8192 moves x 408-byte rows + the 3.3 MB record over 256
// blocks = 26 KB per block) and reads the slab a block on ANOTHER XCD wrote the phase before.
//   hipcc -O3 --offload-arch=gfx950 -o xcd_barrier xcd_barrier.hip && ./xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct bar_state {
    unsigned census[8 * 16];      // blocks resident per XCC (filled once, before the first phase)
    unsigned xcd_count[8 * 16];   // arrivals per XCC, monotonic
    unsigned xcd_gen[8 * 16];     // generation per XCC
    unsigned top_count, pad0[15];
    unsigned flat_count, pad1[15];
    unsigned timeout, nxcc;
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ bool spin_until(unsigned* w, unsigned target, bar_state* st)
{
    unsigned spins = 0;
    while (__hip_atomic_load(w, RLX) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { __hip_atomic_store(&st->timeout, 1u, RLX); return false; }
    }
    return true;
}

// one-time: count the blocks on every XCC, then a flat barrier so that every block reads the final census
__device__ unsigned census_phase(bar_state* st, const int nblocks, const unsigned xcc)
{
    __shared__ unsigned per;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&st->census[xcc * 16], 1u, RLX);
        __hip_atomic_fetch_add(&st->flat_count, 1u, RLX);
        spin_until(&st->flat_count, (unsigned)nblocks, st);
        per = __hip_atomic_load(&st->census[xcc * 16], RLX);
    }
    __syncthreads();
    return per;
}

template <bool LEADER_ONLY>
__device__ void grid_barrier(bar_state* st, const unsigned xcc, const unsigned per_xcc, const unsigned nxcc, const unsigned epoch)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's stores are in the XCD's L2
    __syncthreads();                                           // ... and every other wave's of the block
    if (threadIdx.x == 0) {
        if (!LEADER_ONLY) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned t = __hip_atomic_fetch_add(&st->xcd_count[xcc * 16], 1u, RLX);
        if (t == epoch * per_xcc + per_xcc - 1) {              // last block of this XCC
            if (LEADER_ONLY) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // one L2 write-back for the XCD
            const unsigned tt = __hip_atomic_fetch_add(&st->top_count, 1u, RLX);
            if (tt == epoch * nxcc + nxcc - 1)
                for (int x = 0; x < 8; ++x) __hip_atomic_store(&st->xcd_gen[x * 16], epoch + 1, RLX);
        }
        spin_until(&st->xcd_gen[xcc * 16], epoch + 1, st);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// the phase: read the slab the block across the chip wrote last phase, write our own (float4 per thread, `reps` rounds)
__device__ __forceinline__ void phase_work(float* slabs, const int nblocks, const int slab_f4, const int phase, float* sink)
{
    const int b = blockIdx.x, other = (b + nblocks / 2 + 1) % nblocks;
    const float4* src = (const float4*)slabs + (size_t)((phase & 1) * nblocks + other) * slab_f4;
    float4* dst = (float4*)slabs + (size_t)(((phase + 1) & 1) * nblocks + b) * slab_f4;
    for (int i = threadIdx.x; i < slab_f4; i += blockDim.x) {
        float4 v = src[i];
        v.x += 1.0f; v.y += v.x; v.z += v.y; v.w += v.z;
        dst[i] = v;
        if (v.x == -1.0f) *sink = v.w;
    }
}

__global__ void __launch_bounds__(512) k_phase(float* slabs, int nblocks, int slab_f4, int phase, float* sink)
{
    phase_work(slabs, nblocks, slab_f4, phase, sink);
}

template <bool LEADER_ONLY>
__global__ void __launch_bounds__(512) k_persistent(float* slabs, int nblocks, int slab_f4, int nphases, bar_state* st, float* sink)
{
    const unsigned xcc = xcc_id();
    const unsigned per = census_phase(st, nblocks, xcc);
    unsigned nx = 0;
    for (int x = 0; x < 8; ++x) nx += __hip_atomic_load(&st->census[x * 16], RLX) != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->nxcc = nx;
    for (int p = 0; p < nphases; ++p) {
        phase_work(slabs, nblocks, slab_f4, p, sink);
        grid_barrier<LEADER_ONLY>(st, xcc, per, nx, (unsigned)p);
        if (__hip_atomic_load(&st->timeout, RLX)) return;
    }
}

int main()
{
    const int N = 2000;
    for (int threads : {256, 512})
    for (int nblocks : {256, 512})
    for (int slab_bytes : {4096, 26624}) {
        if (threads == 512 && nblocks == 512) continue;
        const int slab_f4 = slab_bytes / 16;
        float *slabs, *sink;
        bar_state* st;
        const size_t nbytes = (size_t)2 * nblocks * slab_bytes;
        CHECK(hipMalloc(&slabs, nbytes));
        CHECK(hipMalloc(&sink, 4));
        CHECK(hipMalloc(&st, sizeof(bar_state)));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms_launch = 0, ms_all = 0, ms_leader = 0;
        float chk[3] = {0, 0, 0};
        bar_state h;
        for (int rep = 0; rep < 2; ++rep) {                   // second repetition is the measurement
            CHECK(hipMemset(slabs, 0, nbytes));
            CHECK(hipEventRecord(e0));
            for (int p = 0; p < N; ++p) hipLaunchKernelGGL(k_phase, dim3(nblocks), dim3(threads), 0, 0, slabs, nblocks, slab_f4, p, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms_launch, e0, e1));
            CHECK(hipMemcpy(&chk[0], slabs, 4, hipMemcpyDeviceToHost));

            CHECK(hipMemset(slabs, 0, nbytes));
            CHECK(hipMemset(st, 0, sizeof(bar_state)));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_persistent<false>, dim3(nblocks), dim3(threads), 0, 0, slabs, nblocks, slab_f4, N, st, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms_all, e0, e1));
            CHECK(hipMemcpy(&chk[1], slabs, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
            if (h.timeout) { printf("  [every-block-release barrier TIMED OUT]\n"); break; }

            CHECK(hipMemset(slabs, 0, nbytes));
            CHECK(hipMemset(st, 0, sizeof(bar_state)));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_persistent<true>, dim3(nblocks), dim3(threads), 0, 0, slabs, nblocks, slab_f4, N, st, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms_leader, e0, e1));
            CHECK(hipMemcpy(&chk[2], slabs, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
            if (h.timeout) { printf("  [leader-release barrier TIMED OUT]\n"); break; }
        }
        printf("threads=%3d blocks=%4d slab=%5d B  launches %.2f us/phase | barrier, every block releases %.2f | XCD leader releases %.2f"
               "   (xccs %u, census %u..; slab[0].x = %.0f %.0f %.0f, expect %d)\n",
               threads, nblocks, slab_bytes, ms_launch * 1e3 / N, ms_all * 1e3 / N, ms_leader * 1e3 / N, h.nxcc, h.census[0],
               chk[0], chk[1], chk[2], N);
        hipFree(slabs); hipFree(sink); hipFree(st);
    }
    return 0;
}
