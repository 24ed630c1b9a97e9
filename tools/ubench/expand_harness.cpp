// expand_harness.cpp -- host-only: synthetic accept-compacted blocks at C2's shape (65 536 chains x 101 rows, fp64, 23.7 % changed)
// through mhx_compact_expand with T threads: what the host side of the return path can write per second on this machine.
// build: g++ -O3 -std=c++17 -pthread -I../../advancedmh.jl_amd/csrc expand_harness.cpp ../../advancedmh.jl_amd/csrc/mhx_host_expand.cpp expand_harness_fail.cpp
#include "../../include/mhx.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <sys/mman.h>
extern "C" int mhx_compact_expand(const void*, size_t, void*, uint8_t*, int64_t, int32_t);
size_t mhx_compact_payload_offset(uint32_t count, uint32_t words, uint32_t nchains);
int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 1;
    const uint32_t n = 65536, d1 = 101, cnt = 5, nblocks = argc > 2 ? (uint32_t)atoi(argv[2]) : 4, words = n / 64;
    const size_t N = (size_t)cnt * nblocks;
    std::mt19937_64 g(1);
    std::vector<std::vector<unsigned char>> blocks;
    for (uint32_t b = 0; b < nblocks; ++b) {
        const size_t nw = (size_t)cnt * words, po = mhx_compact_payload_offset(cnt, words, n);
        std::vector<uint64_t> mask(nw);
        std::vector<uint32_t> rank(nw);
        uint64_t run = 0;
        for (size_t i = 0; i < nw; ++i) {
            uint64_t m = 0;
            if (b == 0 && i < words) m = ~0ull;
            else for (int k = 0; k < 64; ++k) if ((g() & 0xffff) < 0.237 * 65536) m |= 1ull << k;
            mask[i] = m; rank[i] = (uint32_t)run; run += __builtin_popcountll(m);
        }
        mhx_compact_hdr h{};
        h.magic = MHX_COMPACT_MAGIC; h.elem_bytes = 8; h.dim1 = d1; h.nchains = n; h.first_sample = (uint64_t)b * cnt; h.count = cnt; h.words = words;
        h.total_changed = run; h.payload_offset = po; h.block_bytes = po + run * d1 * 8;
        std::vector<unsigned char> blk(h.block_bytes);
        memcpy(blk.data(), &h, 64); memcpy(blk.data() + 64, mask.data(), 8 * nw); memcpy(blk.data() + 64 + 8 * nw, rank.data(), 4 * nw);
        double* p = (double*)(blk.data() + po);
        for (size_t i = 0; i < run * d1; ++i) p[i] = (double)i;
        blocks.push_back(std::move(blk));
    }
    double* out = (double*)aligned_alloc(4096, N * d1 * n * 8);
    if (argc > 3 && atoi(argv[3])) madvise(out, N * d1 * n * 8, MADV_HUGEPAGE);
    if (!(argc > 4 && atoi(argv[4]))) memset(out, 0, N * d1 * n * 8);      // argv[4] = 1: first touch happens inside the expansion
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (auto& b : blocks) if (mhx_compact_expand(b.data(), b.size(), out, nullptr, (int64_t)N, T)) { printf("fail\n"); return 1; }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("T=%d: %.3f s, %.2f GB/s out\n", T, dt, N * d1 * n * 8 / dt / 1e9);
    }
}
