// copy_rates.hip -- what the host<->device copies of the API paths cost on this box (profiles/r06_copy_rates.txt):
// pageable and pinned H2D in 16 MiB feeds (bzh_stream_feed's copy), the same feed cut into K parts copied by K threads on K
// streams, D2H of a 30 MB stream to pageable and pinned memory.   hipcc -O2 -o copy_rates copy_rates.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t N = 100000000, FEED = 16u << 20, OUT = 30000000;
    uint8_t *d = nullptr, *hp = nullptr, *hq = nullptr;
    hipMalloc(&d, N + FEED);
    uint8_t *pg = (uint8_t *)malloc(N);
    memset(pg, 7, N);
    hipHostMalloc((void **)&hp, N, hipHostMallocDefault);
    memset(hp, 7, N);
    uint8_t *po = (uint8_t *)malloc(OUT);
    memset(po, 1, OUT);
    hipHostMalloc((void **)&hq, OUT, hipHostMallocDefault);
    hipStream_t st[8];
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto feeds = [&](const uint8_t *src, const char *what) {
        double best = 1e9;
        for (int it = 0; it < 5; it++) {
            const double t0 = now_ms();
            for (size_t o = 0; o < N; o += FEED) {
                const size_t k = N - o < FEED ? N - o : FEED;
                hipMemcpyAsync(d + o, src + o, k, hipMemcpyHostToDevice, st[0]);
                hipStreamSynchronize(st[0]);
            }
            best = std::min(best, now_ms() - t0);
        }
        printf("H2D %s, 16 MiB feeds, one thread: %.2f ms for 100 MB = %.1f GB/s\n", what, best, N / best / 1e6);
    };
    feeds(pg, "pageable");
    feeds(hp, "pinned");
    for (int K : {2, 4, 8}) {
        double best = 1e9;
        for (int it = 0; it < 5; it++) {
            const double t0 = now_ms();
            for (size_t o = 0; o < N; o += FEED) {
                const size_t k = N - o < FEED ? N - o : FEED;
                const size_t part = (k + K - 1) / K;
                std::vector<std::thread> th;
                for (int j = 0; j < K; j++)
                    th.emplace_back([&, j]() {
                        const size_t a = j * part, b = std::min(k, a + part);
                        if (a >= b) return;
                        hipMemcpyAsync(d + o + a, pg + o + a, b - a, hipMemcpyHostToDevice, st[j]);
                        hipStreamSynchronize(st[j]);
                    });
                for (auto &t : th) t.join();
            }
            best = std::min(best, now_ms() - t0);
        }
        printf("H2D pageable, 16 MiB feeds cut into %d parts on %d threads/streams: %.2f ms = %.1f GB/s\n", K, K, best, N / best / 1e6);
    }
    // one pageable copy of the whole input
    {
        double best = 1e9;
        for (int it = 0; it < 5; it++) {
            const double t0 = now_ms();
            hipMemcpy(d, pg, N, hipMemcpyHostToDevice);
            best = std::min(best, now_ms() - t0);
        }
        printf("H2D pageable, ONE copy of 100 MB: %.2f ms = %.1f GB/s\n", best, N / best / 1e6);
        best = 1e9;
        for (int it = 0; it < 5; it++) {
            const double t0 = now_ms();
            hipMemcpyAsync(d, hp, N, hipMemcpyHostToDevice, st[0]);
            hipStreamSynchronize(st[0]);
            best = std::min(best, now_ms() - t0);
        }
        printf("H2D pinned, ONE copy of 100 MB: %.2f ms = %.1f GB/s\n", best, N / best / 1e6);
    }
    for (int pinned = 0; pinned < 2; pinned++) {
        double best = 1e9;
        for (int it = 0; it < 5; it++) {
            const double t0 = now_ms();
            hipMemcpyAsync(pinned ? hq : po, d, OUT, hipMemcpyDeviceToHost, st[0]);
            hipStreamSynchronize(st[0]);
            best = std::min(best, now_ms() - t0);
        }
        printf("D2H %s, 30 MB: %.2f ms = %.1f GB/s\n", pinned ? "pinned" : "pageable", best, OUT / best / 1e6);
    }
    return 0;
}
