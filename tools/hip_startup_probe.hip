// Where a fresh process's first HIP calls spend their time (L2 is bound by this at 1 GB inputs).  hipcc -O2 -o hip_startup_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k() {}
int main() {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto lap = [&](const char *what) {
        auto t1 = clk::now();
        std::printf("%-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    hipInit(0); lap("hipInit");
    int n = 0; hipGetDeviceCount(&n); lap("hipGetDeviceCount");
    hipSetDevice(0); lap("hipSetDevice");
    hipFree(nullptr); lap("hipFree(0) (context)");
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); lap("hipStreamCreate");
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s); hipStreamSynchronize(s); lap("first kernel launch + sync");
    void *a, *b, *c; hipMalloc(&a, 1ull << 30); lap("hipMalloc 1 GiB");
    hipMalloc(&b, 5ull << 29); lap("hipMalloc 2.5 GiB");
    hipMalloc(&c, 5ull << 29); lap("hipMalloc 2.5 GiB (2nd)");
    void *h; hipHostMalloc(&h, 1 << 20, hipHostMallocDefault); lap("hipHostMalloc 1 MiB");
    static char src[1 << 20];
    hipMemcpyAsync(a, src, sizeof src, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); lap("first pageable H2D 1 MiB");
    return 0;
}
