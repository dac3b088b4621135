// What a second, third, fourth HIP stream costs a one-file process: creation, first kernel, first copy in either direction,
// and whether priorities or the null stream change it.  hipcc -O2 --offload-arch=gfx950 -o stream_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void k(int *p) {
    if (p) *p = 1;
}

int main(int argc, char **argv) {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto lap = [&](const char *what, int i = -1) {
        auto t1 = clk::now();
        if (i >= 0) std::printf("%-40s [%d] %8.2f ms\n", what, i, std::chrono::duration<double, std::milli>(t1 - t0).count());
        else std::printf("%-44s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = clk::now();
    };
    const int n = argc > 1 ? std::atoi(argv[1]) : 5;
    hipInit(0);
    hipSetDevice(0);
    hipFree(nullptr);
    lap("hipInit + context");
    void *d = nullptr, *h = nullptr;
    hipMalloc(&d, 64 << 20);
    hipHostMalloc(&h, 16 << 20, hipHostMallocDefault);
    lap("hipMalloc 64 MB + hipHostMalloc 16 MB");
    hipStream_t s[16];
    for (int i = 0; i < n; ++i) {
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        lap("hipStreamCreateWithFlags", i);
    }
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[i], (int *)d);
        hipStreamSynchronize(s[i]);
        lap("first kernel + sync", i);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[i], (int *)d);
        hipStreamSynchronize(s[i]);
        lap("second kernel + sync", i);
    }
    for (int i = 0; i < n; ++i) {
        hipMemcpyAsync(d, h, 16 << 20, hipMemcpyHostToDevice, s[i]);
        hipStreamSynchronize(s[i]);
        lap("first pinned H2D 16 MB + sync", i);
        hipMemcpyAsync(h, d, 16 << 20, hipMemcpyDeviceToHost, s[i]);
        hipStreamSynchronize(s[i]);
        lap("first pinned D2H 16 MB + sync", i);
        hipMemcpyAsync(d, h, 16 << 20, hipMemcpyHostToDevice, s[i]);
        hipStreamSynchronize(s[i]);
        lap("second pinned H2D 16 MB + sync", i);
    }
    {   // four more streams one after the other, then four more from four threads at once: does creation run side by side?
        hipStream_t q[8];
        auto a0 = clk::now();
        for (int i = 0; i < 4; ++i) hipStreamCreateWithFlags(&q[i], hipStreamNonBlocking);
        auto a1 = clk::now();
        std::thread th[4];
        for (int i = 0; i < 4; ++i)
            th[i] = std::thread([&q, i] {
                hipSetDevice(0);
                hipStreamCreateWithFlags(&q[4 + i], hipStreamNonBlocking);
            });
        for (auto &x : th) x.join();
        auto a2 = clk::now();
        std::printf("4 more streams in a row: %.2f ms; 4 more from 4 threads at once: %.2f ms\n", std::chrono::duration<double, std::milli>(a1 - a0).count(),
                    std::chrono::duration<double, std::milli>(a2 - a1).count());
        t0 = clk::now();
    }
    hipEvent_t e;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    lap("hipEventCreate");
    hipEventRecord(e, s[0]);
    hipStreamWaitEvent(s[1 % n], e, 0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[1 % n], (int *)d);
    hipStreamSynchronize(s[1 % n]);
    lap("event record + cross-stream wait + kernel");
    // sync latency: 200 x (tiny kernel + 8-byte D2H + sync)
    for (int rep = 0; rep < 2; ++rep) {
        auto a = clk::now();
        for (int i = 0; i < 200; ++i) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[0], (int *)d);
            hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s[0]);
            hipStreamSynchronize(s[0]);
        }
        std::printf("kernel + 8-byte D2H + sync, per turn: %.1f us\n", std::chrono::duration<double, std::micro>(clk::now() - a).count() / 200);
    }
    {
        auto a = clk::now();
        for (int i = 0; i < 200; ++i) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[0], (int *)d);
            hipStreamSynchronize(s[0]);
        }
        std::printf("kernel + sync, per turn: %.1f us\n", std::chrono::duration<double, std::micro>(clk::now() - a).count() / 200);
        // a value the kernel writes to pinned host memory, polled by the host: no copy, no runtime sync
        volatile int *flag = (volatile int *)h;
        a = clk::now();
        for (int i = 0; i < 200; ++i) {
            *flag = 0;
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[0], (int *)h);
            while (*flag == 0) {
            }
        }
        std::printf("kernel writes pinned host flag, host polls, per turn: %.1f us\n", std::chrono::duration<double, std::micro>(clk::now() - a).count() / 200);
        hipStreamSynchronize(s[0]);
    }
    return 0;
}
