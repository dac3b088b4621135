/* outlier_oracle.c — C restatement of `inquiSTR outlier`'s arithmetic (src/outlier.rs:18-31, 75-145), row by row.
 *
 * TEST INFRASTRUCTURE ONLY (CPU baseline of tools/outlier_bench.py and cross-check of oracle/outlier_oracle.py);
 * never linked into the product.  Same shape as the reference: per row a sequential f32 mean / variance, or a
 * 1-D DBSCAN with an O(n^2) neighbour search like [3P] dbscan 0.3.1's range_query.  Built without FP contraction.
 * Pinned by the reference's two unit tests through tests/test_outlier_oracle.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float clean(float v) { return v != v ? 0.0f : v; }

/* keep: 0 = below minsize, 1 = kept, 2 = empty row, 3 = no mode (dbscan); flags[i][k] = 1 where reported */
void orc_outlier_rows(const float *values, const uint32_t *row_len, uint64_t n_rows, uint32_t stride, int method, uint32_t minsize,
                      float cutoff, uint32_t mincluster, uint8_t *flags, uint8_t *keep, int threads) {
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1)
    for (int64_t r = 0; r < (int64_t)n_rows; ++r) {
        const uint32_t n = row_len[r];
        const float *p = values + (uint64_t)r * stride;
        uint8_t *f = flags + (uint64_t)r * stride;
        for (uint32_t k = 0; k < stride; ++k) f[k] = 0;
        if (n == 0) {
            keep[r] = 2;
            continue;
        }
        float mx = clean(p[0]);
        for (uint32_t k = 1; k < n; ++k) mx = clean(p[k]) > mx ? clean(p[k]) : mx;
        if (mx < (float)minsize) {  /* :86-92 */
            keep[r] = 0;
            continue;
        }
        keep[r] = 1;
        if (method == 0) {  /* :18-31, 97-110 */
            float sum = 0.0f;
            for (uint32_t k = 0; k < n; ++k) sum += clean(p[k]);
            const float count = (float)n, mean = sum / count;
            float var = 0.0f;
            for (uint32_t k = 0; k < n; ++k) {
                const float d = mean - clean(p[k]);
                var += d * d;
            }
            const float sd = sqrtf(var / count);
            for (uint32_t k = 0; k < n; ++k) f[k] = (clean(p[k]) - mean) / sd >= cutoff;
            continue;
        }
        /* mode of `value as usize` over the positive values (:132-145); ties: smallest value */
        uint64_t best_key = UINT64_MAX;
        uint32_t best_cnt = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const float x = clean(p[i]);
            if (!(x > 0.0f)) continue;
            const uint64_t key = x >= 18446744073709551616.0f ? UINT64_MAX : (uint64_t)x;
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const float y = clean(p[j]);
                if (y > 0.0f && (y >= 18446744073709551616.0f ? UINT64_MAX : (uint64_t)y) == key) ++cnt;
            }
            if (cnt > best_cnt || (cnt == best_cnt && key < best_key)) best_cnt = cnt, best_key = key;
        }
        if (!best_cnt) {
            keep[r] = 3;
            continue;
        }
        const uint64_t twice = best_key * 2u;
        const double eps = (double)(twice > 10u ? twice : 10u);  /* :115 */
        uint8_t *core = (uint8_t *)malloc(n);
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < n; ++j) cnt += fabs((double)clean(p[i]) - (double)clean(p[j])) < eps;
            core[i] = cnt >= mincluster;
        }
        for (uint32_t i = 0; i < n; ++i) {
            int noise = !core[i];
            for (uint32_t j = 0; j < n && noise; ++j) noise = !(core[j] && fabs((double)clean(p[i]) - (double)clean(p[j])) < eps);
            f[i] = (uint8_t)noise;
        }
        free(core);
    }
}
