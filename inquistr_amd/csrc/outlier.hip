// outlier.hip — `inquiSTR outlier` on the GPU: which samples of a cohort carry an outlying
// repeat length at a locus.  Reference: src/outlier.rs (v0.13.0):
//   get_repeat_lengths  :75-95   NaN -> 0, row dropped when max < minsize
//   std_deviation_and_mean :18-31, z_score_outliers :97-110   f32, SEQUENTIAL sums (the order is part of the result)
//   mode :132-145, dbscan_outliers :112-130 + [3P] dbscan 0.3.1  1-D DBSCAN, eps = max(2 * mode, 10), f64 distances
// Rows (loci) are independent.  z-score: one LANE per row, so that every f32 addition happens in the
// reference's order (no FMA contraction: explicit round-to-nearest intrinsics); the kernel is a stream over the
// matrix, three passes per row.  DBSCAN: one WAVE per row with the row in LDS; neighbour counts are O(n^2)
// broadcast reads of LDS.  Noise = neither a core point nor within eps of one (order-independent).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include "../../include/inquistr_hip.h"
#include "front_kernels.h"

namespace inq {

namespace {

__device__ __forceinline__ float clean(float v) { return v != v ? 0.0f : v; }  // :80-83

// values[row][k] -> vt[k][row]: with one lane per row, lane-adjacent rows must be address-adjacent, or every
// 4-byte load drags a whole cache line through HBM (measured: 81 GB/s without this, row-major).
__global__ __launch_bounds__(256) void outlier_transpose_kernel(const float *values, float *vt, uint64_t n_rows, uint32_t stride,
                                                                uint64_t rows_padded) {
    __shared__ float tile[64][65];
    const uint64_t r0 = (uint64_t)blockIdx.x * 64u;
    const uint32_t c0 = blockIdx.y * 64u;
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
    for (uint32_t r = ty; r < 64u; r += 4u)
        tile[r][tx] = (r0 + r < n_rows && c0 + tx < stride) ? values[(r0 + r) * stride + c0 + tx] : 0.0f;
    __syncthreads();
    for (uint32_t c = ty; c < 64u; c += 4u)
        if (c0 + c < stride && r0 + tx < rows_padded) vt[(uint64_t)(c0 + c) * rows_padded + r0 + tx] = tile[tx][c];
}

__global__ __launch_bounds__(256) void outlier_zscore_kernel(OutlierArgs a, const float *vt, uint64_t rows_padded) {
    const uint64_t row = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (row >= a.n_rows) return;
    const uint32_t n = a.row_len[row];
    const float *p = vt + row;  // element k at p[k * rows_padded]
    uint8_t *f = a.flags + row * (uint64_t)a.stride;  // zero-filled by the caller: only the (rare) hits are written
    if (n == 0) {  // get_repeat_lengths: max of an empty vector -> unwrap() on None
        a.keep[row] = INQ_OUTLIER_ROW_EMPTY;
        return;
    }
    float sum = 0.0f, mx = clean(p[0]);
    for (uint32_t k = 0; k < n; ++k) {
        const float v = clean(p[(uint64_t)k * rows_padded]);
        sum = __fadd_rn(sum, v);
        mx = v > mx ? v : mx;
    }
    if (mx < (float)a.minsize) {  // :86-92
        a.keep[row] = INQ_OUTLIER_ROW_SKIP;
        return;
    }
    a.keep[row] = INQ_OUTLIER_ROW_KEEP;
    const float count = (float)n;
    const float mean = __fdiv_rn(sum, count);
    float var = 0.0f;
    for (uint32_t k = 0; k < n; ++k) {
        const float d = __fsub_rn(mean, clean(p[(uint64_t)k * rows_padded]));
        var = __fadd_rn(var, __fmul_rn(d, d));
    }
    const float sd = __fsqrt_rn(__fdiv_rn(var, count));
    for (uint32_t k = 0; k < n; ++k)
        if (__fdiv_rn(__fsub_rn(clean(p[(uint64_t)k * rows_padded]), mean), sd) >= a.zscore_cutoff) f[k] = 1;  // :106
}

constexpr uint32_t kDbscanMaxCols = 8192;

__global__ __launch_bounds__(64) void outlier_dbscan_kernel(OutlierArgs a) {
    __shared__ float v[kDbscanMaxCols];
    __shared__ uint8_t core[kDbscanMaxCols];
    const uint64_t row = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t n = a.row_len[row];
    const float *p = a.values + row * (uint64_t)a.stride;
    uint8_t *f = a.flags + row * (uint64_t)a.stride;
    if (n == 0 || n > kDbscanMaxCols) {
        if (lane == 0) a.keep[row] = n == 0 ? INQ_OUTLIER_ROW_EMPTY : INQ_OUTLIER_ROW_TOO_WIDE;
        return;
    }
    float mx = -INFINITY;
    for (uint32_t k = lane; k < n; k += 64u) {
        const float x = clean(p[k]);
        v[k] = x;
        mx = x > mx ? x : mx;
    }
    for (int off = 32; off; off >>= 1) {
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    __syncthreads();
    if (mx < (float)a.minsize) {
        if (lane == 0) a.keep[row] = INQ_OUTLIER_ROW_SKIP;
        return;
    }
    // mode of `value as usize` over the positive values (:136-139); ties -> the smallest value (the reference
    // leaves them to HashMap order)
    uint32_t best_cnt = 0;
    unsigned long long best_key = ~0ull;
    for (uint32_t i = lane; i < n; i += 64u) {
        const float x = v[i];
        if (!(x > 0.0f)) continue;
        const unsigned long long key = x >= 18446744073709551616.0f ? ~0ull : (unsigned long long)x;
        uint32_t cnt = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const float y = v[j];
            if (y > 0.0f && (y >= 18446744073709551616.0f ? ~0ull : (unsigned long long)y) == key) ++cnt;
        }
        if (cnt > best_cnt || (cnt == best_cnt && key < best_key)) best_cnt = cnt, best_key = key;
    }
    for (int off = 32; off; off >>= 1) {
        const uint32_t oc = __shfl_xor(best_cnt, off);
        const unsigned long long ok = __shfl_xor(best_key, off);
        if (oc > best_cnt || (oc == best_cnt && ok < best_key)) best_cnt = oc, best_key = ok;
    }
    if (best_cnt == 0) {  // "No mode found for repeat"
        if (lane == 0) a.keep[row] = INQ_OUTLIER_ROW_NO_MODE;
        return;
    }
    if (lane == 0) a.keep[row] = INQ_OUTLIER_ROW_KEEP;
    const unsigned long long twice = best_key * 2ull;  // usize arithmetic of the reference (wraps in release builds)
    const double eps = (double)(twice > 10ull ? twice : 10ull);  // :115
    for (uint32_t i = lane; i < n; i += 64u) {
        const double x = (double)v[i];
        uint32_t cnt = 0;
        for (uint32_t j = 0; j < n; ++j) cnt += fabs(x - (double)v[j]) < eps ? 1u : 0u;  // [3P] range_query: distance < eps
        core[i] = cnt >= a.mincluster ? 1 : 0;                                            // [3P] neighbors.len() >= mpt
    }
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64u) {
        bool noise = !core[i];
        if (noise) {
            const double x = (double)v[i];
            for (uint32_t j = 0; j < n && noise; ++j) noise = !(core[j] && fabs(x - (double)v[j]) < eps);
        }
        if (noise) f[i] = 1;  // :126 Classification::Noise (flags are zero-filled by the caller)
    }
}

}  // namespace

void launch_outlier(const OutlierArgs &a, int method, float *transposed, hipStream_t s) {
    if (!a.n_rows) return;
    if (method == INQ_OUTLIER_ZSCORE) {
        const uint64_t rows_padded = outlier_rows_padded(a.n_rows);
        if (a.stride)
            hipLaunchKernelGGL(outlier_transpose_kernel, dim3((uint32_t)((a.n_rows + 63) / 64), (a.stride + 63) / 64), dim3(256), 0, s,
                               a.values, transposed, a.n_rows, a.stride, rows_padded);
        hipLaunchKernelGGL(outlier_zscore_kernel, dim3((uint32_t)((a.n_rows + 255) / 256)), dim3(256), 0, s, a, transposed, rows_padded);
    } else
        hipLaunchKernelGGL(outlier_dbscan_kernel, dim3((uint32_t)a.n_rows), dim3(64), 0, s, a);
}

}  // namespace inq
