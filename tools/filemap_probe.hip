// Can the GPU copy straight from the page cache?  (round 3, after the resident server made the host's reads the bound of a queue of
// calls.)  A file is mapped read-only; windows of 268 MB are copied to the device (a) as they are (pageable: the runtime stages them
// through its own pinned chunks), (b) page-locked with hipHostRegister first (default flags, then read-only), next to (c) the
// loader's way: pread into an anonymous buffer, then the copy.  hipcc -O2 -o filemap_probe filemap_probe.hip; ./filemap_probe FILE
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (argc < 2) return 2;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) return 1;
    const size_t n = 268u << 20;
    const size_t nwin = (size_t)sb.st_size / n;
    if (nwin < 3) {
        std::printf("file too small\n");
        return 1;
    }
    (void)hipFree(nullptr);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *d;
    (void)hipMalloc(&d, n);
    char *map = (char *)mmap(nullptr, nwin * n, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) return 1;
    // the file is in the page cache (the caller read it once); fault the mapping's page tables in for the first windows only
    auto copy = [&](const char *what, const void *h) {
        auto t0 = clk::now();
        hipError_t e = hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        const double t = ms(t0, clk::now());
        std::printf("%-56s %7.2f ms = %5.1f GB/s (%s)\n", what, t, n / t / 1e6, hipGetErrorString(e));
    };
    for (int k = 0; k < 2; ++k) copy("mapped window, pageable, untouched", map + (size_t)k * n);
    copy("mapped window, pageable, copied before", map);
    for (unsigned flags : {0u, 8u /* hipHostRegisterReadOnly */}) {
        for (size_t k = 2; k < 2 + 3 && k < nwin; ++k) {
            char *w = map + k * n;
            auto t0 = clk::now();
            hipError_t e = hipHostRegister(w, n, flags);
            const double tr = ms(t0, clk::now());
            std::printf("hipHostRegister(flags %u) of a mapped window #%zu: %.2f ms (%s)\n", flags, k, tr, hipGetErrorString(e));
            if (e == hipSuccess) {
                copy("  registered mapped window", w);
                copy("  registered mapped window, again", w);
                t0 = clk::now();
                (void)hipHostUnregister(w);
                std::printf("  unregister: %.2f ms\n", ms(t0, clk::now()));
            } else (void)hipGetLastError();
        }
    }
    // the loader's way: 16 pread streams into an anonymous buffer (not page-locked), then the copy
    char *buf = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(buf, n, MADV_HUGEPAGE);
    std::memset(buf, 1, n);
    for (size_t k = 0; k < 3 && k < nwin; ++k) {
        auto t0 = clk::now();
        std::vector<std::thread> th;
        const int nt = 16;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                const size_t lo = n / nt * t, hi = t + 1 == nt ? n : n / nt * (t + 1);
                size_t off = lo;
                while (off < hi) {
                    const ssize_t g = pread(fd, buf + off, hi - off, (off_t)(k * n + off));
                    if (g <= 0) break;
                    off += (size_t)g;
                }
            });
        for (auto &t : th) t.join();
        std::printf("pread of window #%zu by 16 threads: %.2f ms\n", k, ms(t0, clk::now()));
        copy("  the buffer, pageable", buf);
    }
    return 0;
}
