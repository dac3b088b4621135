// hbm_read_peak.hip — what a pure streaming read reaches on this MI355X: the measured ceiling next to
// the 8 TB/s spec peak that bench.py prices roofline.frac against (SURVEY.md §8d asks for both).
// Two access shapes over a 2.4 GB buffer (config #3's CIGAR volume):
//   linear : grid-stride, 16 B per lane, whole wave contiguous (the best case)
//   locus  : one wave per 24 KB segment read as 30 chunks of 800 B with 4 loads in flight (the
//            walker's shape), default and non-temporal policy
// build: hipcc -O3 --offload-arch=gfx950 tools/hbm_read_peak.hip -o /tmp/hbm_read_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void read_linear(const u32x4 *p, size_t n16, unsigned *sink) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, (short)0, (int)0x7fffffff, 0x00020000);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    // 2 GiB descriptor window: re-base every 2^27 elements
    for (; i < n16; i += stride) {
        const u32x4 v = AUX ? __builtin_nontemporal_load(&p[i]) : p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    (void)r;
    if (acc == 0x12345678u) *sink = acc;
}

template <int AUX>
__global__ __launch_bounds__(256) void read_locus(const u32x4 *p, size_t n_seg, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const u32x4 *base = p + seg * 1500;  // 24 000 B = 30 chunks of 50 x 16 B
    unsigned acc = 0;
    auto ld = [&](int c) -> u32x4 {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(base + c * 50), (short)0, 800, 0x00020000);
        return __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, AUX);
    };
    u32x4 a = ld(0), b = ld(1), c = ld(2), d = ld(3);
    for (int k = 0; k < 28; k += 4) {
        acc ^= a.x ^ a.w; a = ld(k + 4);
        acc ^= b.x ^ b.w; b = ld(k + 5 < 30 ? k + 5 : 29);
        acc ^= c.x ^ c.w; c = ld(k + 6 < 30 ? k + 6 : 29);
        acc ^= d.x ^ d.w; d = ld(k + 7 < 30 ? k + 7 : 29);
    }
    acc ^= a.x ^ b.x ^ c.x ^ d.x;
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t n_seg = 100000, n16 = n_seg * 1500;  // 2.4 GB
    u32x4 *buf;
    unsigned *sink;
    if (hipMalloc(&buf, n16 * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, n16 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 12; ++rep) {
            (void)hipEventRecord(e0);
            launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;
        }
        std::printf("%-28s %8.1f us  %7.0f GB/s\n", name, best * 1e3, n16 * 16 / (best * 1e-3) / 1e9);
    };
    run("linear 16B/lane default", [&] { hipLaunchKernelGGL(read_linear<0>, dim3(256 * 16), dim3(256), 0, 0, buf, n16, sink); });
    run("linear 16B/lane nt", [&] { hipLaunchKernelGGL(read_linear<1>, dim3(256 * 16), dim3(256), 0, 0, buf, n16, sink); });
    run("locus-shaped default", [&] { hipLaunchKernelGGL(read_locus<0>, dim3((n_seg + 3) / 4), dim3(256), 0, 0, buf, n_seg, sink); });
    run("locus-shaped nt", [&] { hipLaunchKernelGGL(read_locus<2>, dim3((n_seg + 3) / 4), dim3(256), 0, 0, buf, n_seg, sink); });
    return 0;
}
