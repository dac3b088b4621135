// What page-locking a span buffer costs and buys on this box (round 3, loader uploads): hipHostMalloc vs hipHostRegister of a
// transparent-huge-page mapping vs plain pageable memory, 268 MB each, host-to-device copy rate of each.
// hipcc -O2 -o pin_probe pin_probe.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstring>
int main() {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const size_t n = 268u << 20;
    hipFree(nullptr);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *d;
    hipMalloc(&d, n);
    auto map_thp = [&]() {
        void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(p, n, MADV_HUGEPAGE);
        std::memset(p, 1, n);
        return p;
    };
    auto copy_rate = [&](const char *what, void *h) {
        for (int i = 0; i < 4; ++i) {
            auto t0 = clk::now();
            hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
            const double t = ms(t0, clk::now());
            std::printf("%-40s copy %d: %7.2f ms = %5.1f GB/s\n", what, i, t, n / t / 1e6);
        }
    };
    void *pg = map_thp();
    copy_rate("pageable (THP mapping)", pg);
    for (int k = 0; k < 3; ++k) {
        auto t0 = clk::now();
        void *h = nullptr;
        hipHostMalloc(&h, n, hipHostMallocDefault);
        std::printf("hipHostMalloc 268 MB #%d: %.2f ms\n", k, ms(t0, clk::now()));
        t0 = clk::now();
        std::memset(h, 2, n);
        std::printf("  first touch: %.2f ms\n", ms(t0, clk::now()));
        if (k == 0) copy_rate("hipHostMalloc", h);
        t0 = clk::now();
        hipHostFree(h);
        std::printf("  hipHostFree: %.2f ms\n", ms(t0, clk::now()));
    }
    for (int k = 0; k < 3; ++k) {
        void *p = map_thp();
        auto t0 = clk::now();
        hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
        std::printf("hipHostRegister 268 MB THP #%d: %.2f ms (%s)\n", k, ms(t0, clk::now()), hipGetErrorString(e));
        if (k == 0 && e == hipSuccess) copy_rate("registered THP mapping", p);
        t0 = clk::now();
        if (e == hipSuccess) hipHostUnregister(p);
        std::printf("  unregister: %.2f ms\n", ms(t0, clk::now()));
        munmap(p, n);
    }
    // two pageable copies from two host threads' worth of streams at once? (one stream here: the runtime stages through its own pinned chunks)
    return 0;
}
