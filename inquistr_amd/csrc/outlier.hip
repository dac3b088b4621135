// outlier.hip — `inquiSTR outlier` on the GPU: which samples of a cohort carry an outlying
// repeat length at a locus.  Reference: src/outlier.rs (v0.13.0):
//   get_repeat_lengths  :75-95   NaN -> 0, row dropped when max < minsize
//   std_deviation_and_mean :18-31, z_score_outliers :97-110   f32, SEQUENTIAL sums (the order is part of the result)
//   mode :132-145, dbscan_outliers :112-130 + [3P] dbscan 0.3.1  1-D DBSCAN, eps = max(2 * mode, 10), f64 distances
// Rows (loci) are independent.  z-score: one LANE per row, so that every f32 addition happens in the
// reference's order (no FMA contraction: explicit round-to-nearest intrinsics); the kernel is a stream over the
// matrix, three passes per row.  DBSCAN: one workgroup per row, the row sorted in LDS, neighbour counts by binary
// search (O(n log n)).  Noise = neither a core point nor within eps of one (order-independent).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <type_traits>

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

// The same arithmetic for rows of at most kOutlierTileMaxStride (256) values, ONE read of the matrix instead of five (transpose: read + write,
// three passes over the transposed copy): a wave copies R whole rows - R * stride contiguous floats of the row-major matrix,
// coalesced - into an LDS tile whose row pitch is odd (lane r walks row r: no bank conflicts), then lane r runs the three passes
// over ITS row out of LDS, every f32 operation in the reference's order as above.  R = 64 rows up to 64 values, 32 beyond (16.6 KB of
// LDS per wave = nine waves per CU up to 128 values, 33 KB up to 256), each wave with its whole tile of loads in flight.
template <int R, int MAXS, bool VEC4>
__global__ __launch_bounds__(64) void outlier_zscore_tile_kernel(OutlierArgs a) {
    extern __shared__ float tile[];
    const uint32_t stride = a.stride, pitch = stride | 1u, lane = threadIdx.x;
    const uint64_t row0 = (uint64_t)blockIdx.x * (uint32_t)R;
    const uint32_t rows_here = (uint32_t)(a.n_rows - row0 < (uint64_t)R ? a.n_rows - row0 : (uint64_t)R);
    const float *src = a.values + row0 * (uint64_t)stride;
    const uint32_t total = rows_here * stride;  // <= R * MAXS floats
    if (VEC4) {
        // stride % 4 == 0: rows begin on 16-byte boundaries and no float4 straddles two rows.  ALL of a lane's loads (<= 32 x 16 B, the
        // whole tile = 32 KB per wave) are issued before the first one is waited for: four waves per CU keep 128 KB in flight
        constexpr int kPerLane = R * MAXS / 256;  // the tile's float4s dealt over 64 lanes
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[kPerLane];
        const f4 *src4 = reinterpret_cast<const f4 *>(src);
        const uint32_t total4 = total >> 2;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const uint32_t e4 = (uint32_t)i * 64u + lane;
            if (e4 < total4) v[i] = __builtin_nontemporal_load(src4 + e4);
        }
        const uint32_t stride4 = stride >> 2;
        uint32_t row = lane / stride4, col4 = lane % stride4;
        const uint32_t step_rows = 64u / stride4, step_cols = 64u % stride4;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const uint32_t e4 = (uint32_t)i * 64u + lane;
            if (e4 < total4) {
                float *dst = tile + row * pitch + col4 * 4u;
                dst[0] = v[i].x, dst[1] = v[i].y, dst[2] = v[i].z, dst[3] = v[i].w;
            }
            col4 += step_cols;
            row += step_rows;
            if (col4 >= stride4) col4 -= stride4, ++row;
        }
    } else {
        uint32_t row = lane / stride, col = lane % stride;
        const uint32_t step_rows = 64u / stride, step_cols = 64u % stride;
#pragma unroll 16
        for (uint32_t e = lane; e < total; e += 64u) {
            tile[row * pitch + col] = src[e];
            col += step_cols;
            row += step_rows;
            if (col >= stride) col -= stride, ++row;
        }
    }
    __syncthreads();
    if (lane >= rows_here) return;
    const uint64_t grow = row0 + lane;
    const uint32_t n = a.row_len[grow];
    const float *p = tile + lane * pitch;
    uint8_t *f = a.flags + grow * (uint64_t)stride;  // zero-filled by the caller: only the (rare) hits are written
    if (n == 0) {
        a.keep[grow] = INQ_OUTLIER_ROW_EMPTY;
        return;
    }
    // Every pass is a chain of dependent f32 operations (the reference's order), but the LDS reads that feed it are not: eight values
    // are fetched at a time, then consumed in order - an LDS round trip per eight elements instead of one per element.
    constexpr uint32_t kU = 8;
    float sum = 0.0f, mx = clean(p[0]);
    uint32_t k = 0;
    for (; k + kU <= n; k += kU) {
        float v[kU];
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j) v[j] = clean(p[k + j]);
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j) {
            sum = __fadd_rn(sum, v[j]);
            mx = v[j] > mx ? v[j] : mx;
        }
    }
    for (; k < n; ++k) {
        const float v = clean(p[k]);
        sum = __fadd_rn(sum, v);
        mx = v > mx ? v : mx;
    }
    if (mx < (float)a.minsize) {
        a.keep[grow] = INQ_OUTLIER_ROW_SKIP;
        return;
    }
    a.keep[grow] = INQ_OUTLIER_ROW_KEEP;
    const float count = (float)n;
    const float mean = __fdiv_rn(sum, count);
    float var = 0.0f;
    for (k = 0; k + kU <= n; k += kU) {
        float d[kU];
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j) {
            d[j] = __fsub_rn(mean, clean(p[k + j]));
            d[j] = __fmul_rn(d[j], d[j]);
        }
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j) var = __fadd_rn(var, d[j]);
    }
    for (; k < n; ++k) {
        const float d = __fsub_rn(mean, clean(p[k]));
        var = __fadd_rn(var, __fmul_rn(d, d));
    }
    const float sd = __fsqrt_rn(__fdiv_rn(var, count));
    for (k = 0; k + kU <= n; k += kU) {  // (independent per element: the divisions overlap)
        bool hit[kU];
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j) hit[j] = __fdiv_rn(__fsub_rn(clean(p[k + j]), mean), sd) >= a.zscore_cutoff;
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j)
            if (hit[j]) f[k + j] = 1;
    }
    for (; k < n; ++k)
        if (__fdiv_rn(__fsub_rn(clean(p[k]), mean), sd) >= a.zscore_cutoff) f[k] = 1;
}

constexpr uint32_t kDbscanMaxCols = 8192;

// 1-D DBSCAN of one row per workgroup, O(n log n): the row is sorted in LDS (bitonic, value + original index),
// after which (a) equal `value as usize` keys are contiguous, so the mode is a matter of run lengths, and
// (b) the neighbours of a point - |x - y| < eps, evaluated in f64 exactly as the crate does - are a contiguous
// range found by two binary searches on that same predicate (rounding is monotone, so it has one switch point
// on either side of the point).  A prefix count of the core points then answers "is a core point within eps".
template <int CAP>
struct DbscanLds {
    float v[CAP];
    uint16_t idx[CAP];
    uint16_t pc[CAP + 1];  // pc[i] = core points among sorted positions [0, i)
    unsigned long long wkey[4];  // per-wave partial results of the workgroup's four waves
    uint32_t wcnt[4];
    float wmax[4];
};

__device__ __forceinline__ unsigned long long as_usize(float x) {  // `value as usize`: saturating, for x > 0
    return x >= 18446744073709551616.0f ? ~0ull : (unsigned long long)x;
}

template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void outlier_dbscan_kernel(OutlierArgs a) {
    constexpr int kWaves = THREADS / 64;
    __shared__ DbscanLds<CAP> L;
    const uint64_t row = blockIdx.x;
    const uint32_t t = threadIdx.x;
    const uint32_t n = a.row_len[row];
    const float *p = a.values + row * (uint64_t)a.stride;
    uint8_t *f = a.flags + row * (uint64_t)a.stride;
    if (n == 0 || n > (uint32_t)CAP) {
        if (t == 0) a.keep[row] = n == 0 ? INQ_OUTLIER_ROW_EMPTY : INQ_OUTLIER_ROW_TOO_WIDE;
        return;
    }
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    float mx = -INFINITY;
    for (uint32_t k = t; k < np2; k += (uint32_t)THREADS) {
        const float x = k < n ? clean(p[k]) : INFINITY;  // padding sorts behind everything
        L.v[k] = x;
        L.idx[k] = (uint16_t)k;
        if (k < n) mx = x > mx ? x : mx;
    }
    for (int off = 32; off; off >>= 1) {
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    if ((t & 63u) == 0) L.wmax[t >> 6] = mx;
    __syncthreads();
    mx = L.wmax[0];
    for (int w = 1; w < kWaves; ++w) mx = fmaxf(mx, L.wmax[w]);
    if (mx < (float)a.minsize) {  // :86-92
        if (t == 0) a.keep[row] = INQ_OUTLIER_ROW_SKIP;
        return;
    }
    // bitonic sort ascending by (value, original index)
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < np2; i += (uint32_t)THREADS) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const float x = L.v[i], y = L.v[l];
                    const uint16_t xi = L.idx[i], yi = L.idx[l];
                    const bool up = (i & k) == 0;
                    const bool gt = x > y || (x == y && xi > yi);
                    if (gt == up) {
                        L.v[i] = y, L.v[l] = x;
                        L.idx[i] = yi, L.idx[l] = xi;
                    }
                }
            }
            __syncthreads();
        }
    }
    // mode of `value as usize` over the positive values (:136-139): longest run of equal keys; ties -> the
    // smallest key (the reference leaves them to HashMap order)
    uint32_t best_cnt = 0;
    unsigned long long best_key = ~0ull;
    for (uint32_t i = t; i < n; i += (uint32_t)THREADS) {
        const float x = L.v[i];
        if (!(x > 0.0f)) continue;
        const unsigned long long key = as_usize(x);
        if (i > 0 && L.v[i - 1] > 0.0f && as_usize(L.v[i - 1]) == key) continue;  // not the start of its run
        uint32_t lo = i + 1, hi = n;  // first position behind the run
        while (lo < hi) {
            const uint32_t m = (lo + hi) >> 1;
            if (as_usize(L.v[m]) == key) lo = m + 1;
            else hi = m;
        }
        const uint32_t cnt = lo - i;
        if (cnt > best_cnt || (cnt == best_cnt && key < best_key)) best_cnt = cnt, best_key = key;
    }
    for (int off = 32; off; off >>= 1) {
        const uint32_t oc = __shfl_xor(best_cnt, off);
        const unsigned long long ok = __shfl_xor(best_key, off);
        if (oc > best_cnt || (oc == best_cnt && ok < best_key)) best_cnt = oc, best_key = ok;
    }
    if ((t & 63u) == 0) L.wcnt[t >> 6] = best_cnt, L.wkey[t >> 6] = best_key;
    __syncthreads();
    best_cnt = 0, best_key = ~0ull;
    for (int w = 0; w < kWaves; ++w)
        if (L.wcnt[w] > best_cnt || (L.wcnt[w] == best_cnt && L.wkey[w] < best_key)) best_cnt = L.wcnt[w], best_key = L.wkey[w];
    if (best_cnt == 0) {  // "No mode found for repeat"
        if (t == 0) a.keep[row] = INQ_OUTLIER_ROW_NO_MODE;
        return;
    }
    if (t == 0) a.keep[row] = INQ_OUTLIER_ROW_KEEP;
    const unsigned long long twice = best_key * 2ull;  // usize arithmetic of the reference (wraps in release builds)
    const double eps = (double)(twice > 10ull ? twice : 10ull);  // :115
    auto near = [&](double x, uint32_t j) { return fabs(x - (double)L.v[j]) < eps; };  // [3P] range_query: distance < eps
    auto range_of = [&](uint32_t i, uint32_t &first, uint32_t &behind) {
        const double x = (double)L.v[i];
        if (!isfinite(x)) {  // inf - inf is NaN, and NaN < eps is false: an infinite value is not even its own neighbour
            first = behind = i;
            return;
        }
        uint32_t lo = 0, hi = i;  // first position <= i that is near
        while (lo < hi) {
            const uint32_t m = (lo + hi) >> 1;
            if (near(x, m)) hi = m;
            else lo = m + 1;
        }
        first = lo;
        lo = i + 1, hi = n;  // first position > i that is not near
        while (lo < hi) {
            const uint32_t m = (lo + hi) >> 1;
            if (near(x, m)) lo = m + 1;
            else hi = m;
        }
        behind = lo;
    };
    for (uint32_t i = t; i < n; i += (uint32_t)THREADS) {
        uint32_t first, behind;
        range_of(i, first, behind);
        L.pc[i + 1] = (behind - first) >= a.mincluster ? 1 : 0;  // [3P] neighbors.len() >= mpt (the point itself counts)
    }
    if (t == 0) L.pc[0] = 0;
    __syncthreads();
    if (t < 64u) {  // one wave turns the core flags into prefix counts
        uint32_t carry = 0;
        for (uint32_t base = 0; base < n; base += 64u) {
            const uint32_t i = base + t;
            uint32_t x = i < n ? L.pc[i + 1] : 0u;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(x, off);
                if ((int)t >= off) x += y;
            }
            if (i < n) L.pc[i + 1] = (uint16_t)(carry + x);
            carry += __shfl(x, 63);
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += (uint32_t)THREADS) {
        const bool core = L.pc[i + 1] != L.pc[i];
        if (core) continue;
        uint32_t first, behind;
        range_of(i, first, behind);
        if (L.pc[behind] == L.pc[first]) f[L.idx[i]] = 1;  // no core point within eps: Classification::Noise (:126)
    }
}

}  // namespace

void launch_outlier(const OutlierArgs &a, int method, float *transposed, hipStream_t s, bool use_tile) {
    if (!a.n_rows) return;
    if (method == INQ_OUTLIER_ZSCORE && a.stride && a.stride <= kOutlierTileMaxStride && use_tile) {
        const uint32_t pitch = a.stride | 1u;
        const bool vec4 = a.stride % 4u == 0u && (reinterpret_cast<uintptr_t>(a.values) & 15u) == 0u;
        auto go = [&](auto rows_c, auto maxs_c) {
            constexpr int R = decltype(rows_c)::value, MAXS = decltype(maxs_c)::value;
            const dim3 grid((uint32_t)((a.n_rows + R - 1) / R));
            if (vec4) hipLaunchKernelGGL((outlier_zscore_tile_kernel<R, MAXS, true>), grid, dim3(64), (uint32_t)R * pitch * 4u, s, a);
            else hipLaunchKernelGGL((outlier_zscore_tile_kernel<R, MAXS, false>), grid, dim3(64), (uint32_t)R * pitch * 4u, s, a);
        };
        // rows per wave, measured (profiles/r05_results/outlier_zscore_lds_tile.txt): 16.6 KB of LDS = nine waves per CU up to 128
        // values (the passes are chains of dependent operations: other waves are what hides them); wider rows keep 32 rows per wave
        // (33 KB, four waves per CU: with 16 rows three quarters of the lanes idle through passes twice as long)
        if (a.stride <= 64u) go(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
        else if (a.stride <= 128u) go(std::integral_constant<int, 32>{}, std::integral_constant<int, 128>{});
        else go(std::integral_constant<int, 32>{}, std::integral_constant<int, 256>{});
    } else if (method == INQ_OUTLIER_ZSCORE) {
        const uint64_t rows_padded = outlier_rows_padded(a.n_rows);
        if (a.stride)
            hipLaunchKernelGGL(outlier_transpose_kernel, dim3((uint32_t)((a.n_rows + 63) / 64), (a.stride + 63) / 64), dim3(256), 0, s,
                               a.values, transposed, a.n_rows, a.stride, rows_padded);
        hipLaunchKernelGGL(outlier_zscore_kernel, dim3((uint32_t)((a.n_rows + 255) / 256)), dim3(256), 0, s, a, transposed, rows_padded);
    } else if (a.stride <= 256u)  // a row of <= 256 values keeps one wave busy, not four
        hipLaunchKernelGGL((outlier_dbscan_kernel<256, 64>), dim3((uint32_t)a.n_rows), dim3(64), 0, s, a);
    else if (a.stride <= 2048u)
        hipLaunchKernelGGL((outlier_dbscan_kernel<2048, 256>), dim3((uint32_t)a.n_rows), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((outlier_dbscan_kernel<kDbscanMaxCols, 256>), dim3((uint32_t)a.n_rows), dim3(256), 0, s, a);
}

}  // namespace inq
