// Micro-benchmark: what bounds a chain of N dependent small kernels on one stream -- the host's submission rate or the GPU's
// kernel boundary?  (a) hipLaunchKernelGGL in a loop: host enqueue time (before any sync) and total time; (b) the same chain
// captured ONCE into a hipGraph and replayed; (c) hipModuleLaunchKernel-style launches with a 120-byte argument block.
// hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct args120 { long a[15]; };
__global__ void __launch_bounds__(256) k_small(float* x, args120 a, int phase)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    x[i] = x[(i * 7 + phase) & 131071] * 0.5f + (float)a.a[3];      // reads what another block wrote in the previous launch
}

int main()
{
    const int N = 1000, blocks = 512;
    float* x;
    hipMalloc(&x, 131072 * sizeof(float));
    hipMemset(x, 0, 131072 * sizeof(float));
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    args120 a{};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(e0, s);
        for (int p = 0; p < N; ++p) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, s, x, a, p);
        hipEventRecord(e1, s);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("stream launches : host enqueue %.2f us/launch, GPU events %.2f us/launch, wall %.2f us/launch\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N, ms * 1e3 / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    // the same chain as a graph
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int p = 0; p < N; ++p) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, s, x, a, p);
    hipStreamEndCapture(s, &g);
    auto c0 = std::chrono::steady_clock::now();
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    auto c1 = std::chrono::steady_clock::now();
    printf("graph instantiate: %.1f us for %d nodes\n", std::chrono::duration<double, std::micro>(c1 - c0).count(), N);
    for (int rep = 0; rep < 3; ++rep) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("graph replay    : host enqueue %.2f us/node, GPU events %.2f us/node, wall %.2f us/node\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N, ms * 1e3 / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    return 0;
}
