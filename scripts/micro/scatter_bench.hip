// Microbenchmark: random 4-byte scatter into per-block 3.6 MB arrays, alone and next to streaming traffic.
// hipcc -O3 --offload-arch=gfx950 scatter_bench.hip -o scatter_bench ; ./scatter_bench [variant]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
typedef unsigned long long u64;
constexpr int TILE = 8192, THREADS = 512, ITEMS = 16;
// variant 0: scatter only; 1: scatter + stream read 8B + stream write 8B; 2: stream only
__device__ uint32_t LIM = 899999;
template <int V, bool XCD>
__global__ void __launch_bounds__(THREADS) k(const uint32_t *perm, uint32_t *rank, const u64 *src, u64 *dst, uint32_t n, uint32_t S, uint32_t T, uint32_t B)
{
    uint32_t b, tile;
    const uint32_t L = blockIdx.x;
    if (XCD) { const uint32_t slot = L >> 3, kk = slot / T; tile = slot - kk * T; b = kk * 8 + (L & 7); }
    else { b = L / T; tile = L - b * T; }
    if (b >= B) return;
    const size_t base = (size_t)b * S;
#pragma unroll
    for (int kx = 0; kx < ITEMS; kx++) {
        const uint32_t e = tile * TILE + kx * THREADS + threadIdx.x;
        if (e < n) {
            u64 x = 0;
            if (V >= 1) x = src[base + e];
            if (V <= 1) rank[base + perm[base + e] % LIM] = e + (uint32_t)x;
            if (V >= 1) dst[base + e] = x + 1;
        }
    }
}
int main(int argc, char **argv)
{
    const uint32_t n = 899999, S = 900096, B = 113, T = (n + TILE - 1) / TILE;
    std::vector<uint32_t> hp((size_t)B * S);
    std::mt19937 rng(1);
    for (uint32_t b = 0; b < B; b++) {
        std::iota(hp.begin() + (size_t)b * S, hp.begin() + (size_t)b * S + n, 0u);
        std::shuffle(hp.begin() + (size_t)b * S, hp.begin() + (size_t)b * S + n, rng);
    }
    uint32_t *perm, *rank; u64 *src, *dst;
    hipMalloc(&perm, (size_t)B * S * 4); hipMalloc(&rank, (size_t)B * S * 4);
    hipMalloc(&src, (size_t)B * S * 8); hipMalloc(&dst, (size_t)B * S * 8);
    hipMemcpy(perm, hp.data(), (size_t)B * S * 4, hipMemcpyHostToDevice);
    hipMemset(src, 1, (size_t)B * S * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t grid = 8 * ((B + 7) / 8) * T;
    auto run = [&](const char *name, auto kern) {
        for (int it = 0; it < 2; it++) kern<<<grid, THREADS>>>(perm, rank, src, dst, n, S, T, B);
        hipEventRecord(e0);
        for (int it = 0; it < 5; it++) kern<<<grid, THREADS>>>(perm, rank, src, dst, n, S, T, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us per launch (%.1f G elements/s)\n", name, ms * 1e3 / 5, (double)B * n / (ms / 5 * 1e-3) / 1e9);
    };
    for (uint32_t lim : {899999u, 450000u, 225000u, 112000u, 56000u}) {
        hipMemcpyToSymbol(HIP_SYMBOL(LIM), &lim, 4);
        printf("destination range %u entries (%.2f MB per block):\n", lim, lim * 4 / 1e6);
        run("  scatter only, XCD map", k<0, true>);
        run("  scatter + streams, XCD map", k<1, true>);
    }
    { uint32_t lim = 899999u; hipMemcpyToSymbol(HIP_SYMBOL(LIM), &lim, 4); }
    run("scatter only, XCD map", k<0, true>);
    run("scatter only, plain map", k<0, false>);
    run("scatter + streams, XCD map", k<1, true>);
    run("scatter + streams, plain map", k<1, false>);
    run("streams only, XCD map", k<2, true>);
    // fewer workgroups in flight per XCD (dynamic LDS as ballast): how many blocks share an L2 at a time?
    for (uint32_t lds : {40000u, 60000u, 100000u}) {
        auto runl = [&](const char *name, auto kern) {
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
            for (int it = 0; it < 2; it++) kern<<<grid, THREADS, lds>>>(perm, rank, src, dst, n, S, T, B);
            hipEventRecord(e0);
            for (int it = 0; it < 5; it++) kern<<<grid, THREADS, lds>>>(perm, rank, src, dst, n, S, T, B);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s %8.1f us per launch (%.1f G elements/s) with %u B of LDS per workgroup\n", name, ms * 1e3 / 5, (double)B * n / (ms / 5 * 1e-3) / 1e9, lds);
        };
        runl("scatter only, XCD map", k<0, true>);
        runl("scatter + streams, XCD map", k<1, true>);
    }
    // one block per XCD at a time: 15 launches of 8 blocks each
    auto run8 = [&](const char *name, auto kern) {
        const uint32_t g8 = 8 * T;
        auto all = [&]() {
            for (uint32_t b0 = 0; b0 < B; b0 += 8) {
                const uint32_t nb = std::min(8u, B - b0);
                kern<<<g8, THREADS>>>(perm + (size_t)b0 * S, rank + (size_t)b0 * S, src + (size_t)b0 * S, dst + (size_t)b0 * S, n, S, T, nb);
            }
        };
        all(); all();
        hipEventRecord(e0);
        for (int it = 0; it < 5; it++) all();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us per sweep of all blocks (%.1f G elements/s)\n", name, ms * 1e3 / 5, (double)B * n / (ms / 5 * 1e-3) / 1e9);
    };
    run8("8 blocks per launch: scatter only", k<0, true>);
    run8("8 blocks per launch: + streams", k<1, true>);
    return 0;
}
