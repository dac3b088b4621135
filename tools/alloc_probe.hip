// What device allocations cost a one-file process (the span loop's inflate_ahead buffers were 0.3 s in round 3): hipMalloc / hipFree
// by size, the first touch, a second allocation of a size just freed, hipMallocAsync from the default pool, and the virtual-memory
// route (reserve once, map physical chunks as the need grows).  hipcc -O2 --offload-arch=gfx950 -o alloc_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void touch(uint32_t *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    hipSetDevice(0);
    hipFree(nullptr);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, (uint32_t *)nullptr, (size_t)0);
    hipStreamSynchronize(s);
    std::printf("%-44s %10s %10s %10s %10s\n", "size", "malloc ms", "touch ms", "touch2 ms", "free ms");
    for (size_t mb : {1, 16, 64, 256, 512, 1024, 2048, 4096}) {
        const size_t bytes = mb << 20;
        void *p = nullptr;
        auto t0 = clk::now();
        hipError_t e = hipMalloc(&p, bytes);
        auto t1 = clk::now();
        if (e != hipSuccess) {
            std::printf("hipMalloc %zu MB failed\n", mb);
            continue;
        }
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (uint32_t *)p, bytes / 4);
        hipStreamSynchronize(s);
        auto t2 = clk::now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (uint32_t *)p, bytes / 4);
        hipStreamSynchronize(s);
        auto t3 = clk::now();
        hipFree(p);
        auto t4 = clk::now();
        char name[64];
        std::snprintf(name, sizeof name, "hipMalloc %zu MB", mb);
        std::printf("%-44s %10.3f %10.3f %10.3f %10.3f\n", name, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
    }
    // a second round: does the runtime keep what was freed?
    for (size_t mb : {256, 1024}) {
        const size_t bytes = mb << 20;
        void *p = nullptr;
        auto t0 = clk::now();
        hipMalloc(&p, bytes);
        auto t1 = clk::now();
        hipFree(p);
        auto t2 = clk::now();
        std::printf("again hipMalloc %4zu MB %36.3f ms, free %.3f ms\n", mb, ms(t0, t1), ms(t1, t2));
    }
    // several allocations in parallel threads?  (the uploader allocates its slots while the caller allocates scan buffers)
    // stream-ordered pool
    {
        hipMemPool_t pool;
        hipDeviceGetDefaultMemPool(&pool, 0);
        uint64_t thr = ~0ull;
        hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
        for (int rep = 0; rep < 2; ++rep)
            for (size_t mb : {256, 1024}) {
                void *p = nullptr;
                auto t0 = clk::now();
                hipError_t e = hipMallocAsync(&p, mb << 20, s);
                hipStreamSynchronize(s);
                auto t1 = clk::now();
                if (e != hipSuccess) {
                    std::printf("hipMallocAsync failed: %s\n", hipGetErrorString(e));
                    break;
                }
                hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (uint32_t *)p, (mb << 20) / 4);
                hipStreamSynchronize(s);
                auto t2 = clk::now();
                hipFreeAsync(p, s);
                hipStreamSynchronize(s);
                auto t3 = clk::now();
                std::printf("hipMallocAsync %4zu MB (round %d) %26.3f ms, touch %.3f ms, freeAsync %.3f ms\n", mb, rep, ms(t0, t1), ms(t1, t2), ms(t2, t3));
            }
    }
    // virtual memory management: reserve 8 GB of address space, back it 256 MB at a time
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        std::printf("vmm granularity: %zu (%s)\n", gran, hipGetErrorString(e));
        void *base = nullptr;
        auto t0 = clk::now();
        e = hipMemAddressReserve(&base, 8ull << 30, 0, nullptr, 0);
        auto t1 = clk::now();
        std::printf("hipMemAddressReserve 8 GB: %.3f ms (%s)\n", ms(t0, t1), hipGetErrorString(e));
        if (e == hipSuccess) {
            const size_t chunk = 256ull << 20;
            std::vector<hipMemGenericAllocationHandle_t> hs;
            for (int i = 0; i < 4; ++i) {
                hipMemGenericAllocationHandle_t h;
                auto a = clk::now();
                e = hipMemCreate(&h, chunk, &prop, 0);
                auto b = clk::now();
                if (e != hipSuccess) {
                    std::printf("hipMemCreate failed: %s\n", hipGetErrorString(e));
                    break;
                }
                e = hipMemMap((char *)base + i * chunk, chunk, 0, h, 0);
                auto c = clk::now();
                hipMemAccessDesc ad = {};
                ad.location = prop.location;
                ad.flags = hipMemAccessFlagsProtReadWrite;
                hipError_t e2 = hipMemSetAccess((char *)base + i * chunk, chunk, &ad, 1);
                auto d = clk::now();
                hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (uint32_t *)((char *)base + i * chunk), chunk / 4);
                hipError_t e3 = hipStreamSynchronize(s);
                auto f = clk::now();
                std::printf("vmm chunk %d: create %.3f map %.3f (%s) access %.3f (%s) touch %.3f (%s)\n", i, ms(a, b), ms(b, c), hipGetErrorString(e), ms(c, d),
                            hipGetErrorString(e2), ms(d, f), hipGetErrorString(e3));
                hs.push_back(h);
            }
        }
    }
    // pinned host memory by size (the rows' staging, span buffers if pinned)
    for (size_t mb : {1, 64, 256}) {
        void *h = nullptr;
        auto t0 = clk::now();
        hipHostMalloc(&h, mb << 20, hipHostMallocDefault);
        auto t1 = clk::now();
        hipHostFree(h);
        auto t2 = clk::now();
        std::printf("hipHostMalloc %4zu MB %10.3f ms, free %.3f ms\n", mb, ms(t0, t1), ms(t1, t2));
    }
    return 0;
}
