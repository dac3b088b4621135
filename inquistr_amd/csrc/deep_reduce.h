// deep_reduce.h - the reduces of a locus with more than 256 offered reads that ONE workgroup does (device functions; the walk that
// leaves the per-read Calls in the ctx scratch is locus_call_mid_walk in kernels.hip, the kernel that calls these is locus_call_tail
// in deep_select.hip):
//   sort_reduce_locus   up to 16 384 reads: 64-bit keys sorted by a bitonic network in LDS
//   reduce_deep_select  beyond that (and Calls that do not fit a key): most-significant-byte-first radix select over the scratch
// Reference semantics: median_str_length src/call.rs:497-522, the unphased split :308-322, the phased bins :341-369.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cigar_walk.h"
#include "kernels.h"

namespace inq {

__device__ __forceinline__ double qnan() { return __builtin_nan(""); }

// Per-read results of one very deep locus (more reads than the LDS sort holds, or a Call beyond the sort key's 47 bits) stay
// in the ctx's global scratch (L2-resident: 9 bytes per read) and are reduced there by ONE workgroup with a most-significant-
// byte-first radix select: eight passes of a 256-bin histogram find the k-th smallest value of any subset, O(n) each.  About
// 60 passes per locus whatever its depth (split of the unphased order, clip threshold and the two middle elements per
// haplotype): a 100 000-read locus costs ~25 000 element visits per thread.
struct DeepStore {
    const int64_t *val;
    unsigned char *meta;
};
struct SelectLds {
    unsigned int hist[256];
    unsigned int scan[256];
    unsigned long long prefix;
    unsigned int k, below, eq, flags;
    unsigned int cnt[8];
};
__device__ __forceinline__ uint64_t order_key(int64_t v) { return (uint64_t)v ^ (1ull << 63); }  // signed order as unsigned order

// k-th smallest (0-based) key among the elements for which pred(e, key) holds, plus `lump_cnt` extra elements of key
// `lump_key`.  Block-uniform result; L.below = elements smaller than it, L.eq = elements equal to it (lump included).
template <class Pred>
__device__ __forceinline__ uint64_t radix_select(const DeepStore &S, uint32_t n, Pred pred, uint32_t k, uint64_t lump_key, uint32_t lump_cnt, SelectLds &L) {
    if (threadIdx.x == 0) L.prefix = 0ull, L.k = k, L.below = 0u;
    for (int pass = 7; pass >= 0; --pass) {
        L.hist[threadIdx.x] = 0u;
        __syncthreads();
        const uint64_t prefix = L.prefix;  // the bytes above `pass`, already decided
        auto upper_matches = [&](uint64_t key) { return pass == 7 || (key >> (8 * (pass + 1))) == prefix; };
        for (uint32_t e = threadIdx.x; e < n; e += 256u) {
            const uint64_t key = order_key(S.val[e]);
            if (upper_matches(key) && pred(e, key)) atomicAdd(&L.hist[(key >> (8 * pass)) & 255u], 1u);
        }
        if (threadIdx.x == 0 && lump_cnt && upper_matches(lump_key)) atomicAdd(&L.hist[(lump_key >> (8 * pass)) & 255u], lump_cnt);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t c = 0, bin = 255u;
            const uint32_t want = L.k;
            for (uint32_t b2 = 0; b2 < 256u; ++b2) {
                if (want < c + L.hist[b2]) {
                    bin = b2;
                    break;
                }
                c += L.hist[b2];
            }
            L.k = want - c;
            L.below += c;
            L.eq = L.hist[bin];
            L.prefix = (prefix << 8) | bin;
        }
        __syncthreads();
    }
    return L.prefix;
}

// median_str_length (src/call.rs:497-522) of haplotype group g (1 / 2) of the store
__device__ __forceinline__ double deep_group_median(const DeepStore &S, uint32_t n, uint32_t g, uint32_t ng, uint32_t ns, uint32_t support, SelectLds &L) {
    if (ng < support) return qnan();  // :498-500
    auto in_group = [&](uint32_t e) {
        const uint32_t me = S.meta[e];
        return (me & PM_KEPT) && ((me >> PM_GRP_SHIFT) & 3u) == g;
    };
    const uint32_t take = ns <= support ? support - ns : 0u;  // :509-513: the largest `take` clipped Calls join the spanning ones
    uint64_t t_key = 0;
    uint32_t lump = 0;
    if (take > 0u) {
        const uint32_t nc = ng - ns;  // >= take because ng >= support
        t_key = radix_select(S, n, [&](uint32_t e, uint64_t) { return in_group(e) && (S.meta[e] & PM_CLIP); }, nc - take, 0ull, 0u, L);
        const uint32_t above = nc - L.below - L.eq;  // clipped Calls larger than the threshold value: all chosen
        lump = take - above;                          // ... and this many equal to it (which ones does not change the values)
        __syncthreads();
    }
    const uint32_t M = ns + take;  // >= 1 because support >= 1
    auto chosen = [&](uint32_t e, uint64_t key) { return in_group(e) && (!(S.meta[e] & PM_CLIP) || (take > 0u && key > t_key)); };
    const int64_t vhi = (int64_t)(radix_select(S, n, chosen, M / 2u, t_key, lump, L) ^ (1ull << 63));
    __syncthreads();
    if (M & 1u) return (double)vhi;  // :520
    const int64_t vlo = (int64_t)(radix_select(S, n, chosen, M / 2u - 1u, t_key, lump, L) ^ (1ull << 63));
    __syncthreads();
    return (double)(vlo + vhi) / 2.0;  // :515-518
}

// The whole reduce of one locus over the global store.
template <bool UNPHASED>
__device__ __forceinline__ void reduce_deep_select(const KArgs &a, uint64_t j, uint64_t p0, uint32_t n, SelectLds &L) {
    DeepStore S{a.sval + p0, (unsigned char *)(a.smeta + p0)};
    bool tie = false;
    if (UNPHASED) {  // src/call.rs:311-313: sort by (value, file order), h1 = the lower mcount / 2, h2 = the rest
        if (threadIdx.x < 8) L.cnt[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t local = 0;
        for (uint32_t e = threadIdx.x; e < n; e += 256u) local += (S.meta[e] & PM_KEPT) ? 1u : 0u;
        if (local) atomicAdd(&L.cnt[0], local);
        __syncthreads();
        const uint32_t mcount = L.cnt[0], ks = mcount / 2u;
        __syncthreads();
        uint64_t split = ~0ull;
        uint32_t r = 0;  // elements equal to the split value that still belong to h1 (the first r in file order)
        if (mcount) {
            split = radix_select(S, n, [&](uint32_t e, uint64_t) { return (S.meta[e] & PM_KEPT) != 0; }, ks < mcount ? ks : mcount - 1u, 0ull, 0u, L);
            r = ks - L.below;
            __syncthreads();
        }
        // groups: each thread owns a contiguous stretch so that "the first r equal ones in file order" is a prefix count
        const uint32_t chunk = (n + 255u) / 256u, e0 = min(n, threadIdx.x * chunk), e1 = min(n, e0 + chunk);
        uint32_t eq = 0;
        for (uint32_t e = e0; e < e1; ++e) eq += ((S.meta[e] & PM_KEPT) && order_key(S.val[e]) == split) ? 1u : 0u;
        L.scan[threadIdx.x] = eq;
        if (threadIdx.x == 0) L.flags = 0u;
        __syncthreads();
        uint32_t eq_before = 0;
        for (uint32_t t = 0; t < threadIdx.x; ++t) eq_before += L.scan[t];
        uint32_t fl = 0;
        for (uint32_t e = e0; e < e1; ++e) {
            uint32_t me = S.meta[e];
            if (!(me & PM_KEPT)) continue;
            const uint64_t key = order_key(S.val[e]);
            uint32_t grp = key < split ? 1u : 2u;
            if (key == split) {
                grp = eq_before < r ? 1u : 2u;
                ++eq_before;
                fl |= (me & PM_CLIP) ? 1u : 2u;
            }
            S.meta[e] = (unsigned char)((me & ~(3u << PM_GRP_SHIFT)) | (grp << PM_GRP_SHIFT));
        }
        if (fl) atomicOr(&L.flags, fl);
        __syncthreads();
        // the split cuts through equal values iff some element equal to the split value went to h1 (:312-314 ambiguity)
        tie = ks >= 1u && ks < mcount && r >= 1u && L.flags == 3u;
    }
    if (threadIdx.x < 8) L.cnt[threadIdx.x] = 0u;
    __syncthreads();
    {
        uint32_t c_ng[3] = {0, 0, 0}, c_ns[3] = {0, 0, 0};
        for (uint32_t e = threadIdx.x; e < n; e += 256u) {
            const uint32_t me = S.meta[e];
            if (!(me & PM_KEPT)) continue;
            const uint32_t g = (me >> PM_GRP_SHIFT) & 3u;
            if (g == 1u || g == 2u) {
                c_ng[g]++;
                if (!(me & PM_CLIP)) c_ns[g]++;
            }
        }
        for (int g = 1; g <= 2; ++g) {
            if (c_ng[g]) atomicAdd(&L.cnt[g], c_ng[g]);
            if (c_ns[g]) atomicAdd(&L.cnt[4 + g], c_ns[g]);
        }
    }
    __syncthreads();
    const uint32_t ng1 = L.cnt[1], ng2 = L.cnt[2], ns1 = L.cnt[5], ns2 = L.cnt[6];
    __syncthreads();
    const double out1 = deep_group_median(S, n, 1u, ng1, ns1, a.support, L);
    const double out2 = deep_group_median(S, n, 2u, ng2, ns2, a.support, L);
    if (threadIdx.x == 0) {
        a.phase1[j] = out1;
        a.phase2[j] = out2;
        if (tie) atomicAdd((unsigned long long *)&a.status->ties, 1ull);
    }
    __syncthreads();
}

// ---- the reduce of a locus of up to CAP reads ------------------------------------------------------
// One workgroup per listed locus: the kept Calls become 64-bit keys
//     [63:62] haplotype group | [61:15] value + 2^46 | [14:1] file-order index | [0] clipped
// sorted once by a bitonic network in LDS (padded to the next power of two of the KEPT Calls, so a shallow
// locus sorts a short array whatever CAP is); both haplotype groups are then contiguous ascending ranges
// and median_str_length's span/clip rule reduces to one prefix count of "spanning" flags.  The file index has
// 14 bits in the key: CAP <= 16384 (128 KB of keys, one workgroup per CU).
constexpr uint64_t kKeyBias = 1ull << 46;
constexpr uint64_t kKeySent = ~0ull;

template <int CAP>
struct SortLds {
    unsigned long long key[CAP];
    unsigned int seg[256];
    unsigned int m, c1, tie_span, tie_clip, overflow;
    long long pick[2];
};

__device__ __forceinline__ int64_t key_value(uint64_t k) { return (int64_t)((k >> 15) & ((1ull << 47) - 1ull)) - (int64_t)kKeyBias; }

// median_str_length (src/call.rs:497-522) of the sorted range key[lo, hi).  Block-uniform result.
template <int CAP>
__device__ __forceinline__ double median_of_sorted_range(SortLds<CAP> &L, uint32_t lo, uint32_t hi, uint32_t support) {
    const uint32_t ng = hi - lo;
    if (ng < support) return qnan();  // :498-500
    const uint32_t t = threadIdx.x;
    const uint32_t seglen = (ng + 255u) / 256u;
    const uint32_t s0 = min(hi, lo + t * seglen), s1 = min(hi, s0 + seglen);
    uint32_t spans = 0;
    for (uint32_t e = s0; e < s1; ++e) spans += (uint32_t)(~L.key[e] & 1ull);
    L.seg[t] = spans;
    __syncthreads();
    uint32_t before_me = 0, ns = 0;
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t c = L.seg[k];
        before_me += k < t ? c : 0u;
        ns += c;
    }
    // chosen = every spanning Call, plus (when there are no more than `support` of them) the largest
    // support - ns clipped ones (:509-513) = the LAST `take` clips of the ascending range; which of several
    // equal clipped values is taken does not change the multiset of values
    const uint32_t nc = ng - ns;
    const uint32_t take = ns <= support ? support - ns : 0u;  // <= nc because ng >= support
    const uint32_t first_clip = nc - take;                     // clips with clip-rank >= first_clip are chosen
    const uint32_t M = ns + take;
    uint32_t span_before = before_me;
    for (uint32_t e = s0; e < s1; ++e) {
        const uint64_t k = L.key[e];
        const bool clip = (k & 1ull) != 0ull;
        const uint32_t clip_before = (e - lo) - span_before;
        const bool chosen = !clip || clip_before >= first_clip;
        if (chosen) {
            const uint32_t r = span_before + (clip_before > first_clip ? clip_before - first_clip : 0u);
            if (r == M / 2u) L.pick[1] = key_value(k);
            if (!(M & 1u) && r == M / 2u - 1u) L.pick[0] = key_value(k);
        }
        span_before += clip ? 0u : 1u;
    }
    __syncthreads();
    const double out = (M & 1u) ? (double)L.pick[1] : (double)(L.pick[0] + L.pick[1]) / 2.0;  // :515-520
    __syncthreads();
    return out;
}

// DEFER: a Call beyond the key's 47 bits is not reduced here - the rows are set to kDeferredRow and locus_call_tail, which looks for
// that pattern at the loci the walk kernel reduces in place, runs the radix select (sh may be null then).  Keeps the select's
// registers out of locus_call_mid_walk, whose medium-depth path wants the occupancy.
constexpr unsigned long long kDeferredRow = 0x7ff8dead00000000ull;  // a quiet NaN no row ever holds (rows are finite or __builtin_nan(""))
template <bool UNPHASED, int CAP, bool DEFER = false>
__device__ __forceinline__ void sort_reduce_locus(const KArgs &a, uint64_t j, uint64_t p0, uint32_t n, SortLds<CAP> &L, SelectLds *shp) {
    if (threadIdx.x == 0) L.m = L.c1 = L.tie_span = L.tie_clip = L.overflow = 0u;
    __syncthreads();
    // keys of the elements that belong to a haplotype group (slot order is fixed by the sort that follows)
    for (uint32_t e = threadIdx.x; e < n; e += 256u) {
        const uint32_t me = a.smeta[p0 + e];
        const uint32_t g = (me >> PM_GRP_SHIFT) & 3u;
        const bool in = UNPHASED ? (me & PM_KEPT) != 0u : ((me & PM_KEPT) && (g == 1u || g == 2u));
        if (in) {
            const int64_t v = a.sval[p0 + e];
            if (v < -(int64_t)kKeyBias || v >= (int64_t)kKeyBias) L.overflow = 1u;
            const uint64_t key = ((uint64_t)(UNPHASED ? 0u : g) << 62) | (((uint64_t)(v + (int64_t)kKeyBias) & ((1ull << 47) - 1ull)) << 15) |
                                 ((uint64_t)e << 1) | ((me & PM_CLIP) ? 1ull : 0ull);
            L.key[atomicAdd(&L.m, 1u)] = key;
            if (!UNPHASED && g == 1u) atomicAdd(&L.c1, 1u);
        }
    }
    __syncthreads();
    if (L.overflow) {  // a Call beyond 47 bits: not representable in the key
        if (DEFER) {
            if (threadIdx.x == 0) a.phase1[j] = __longlong_as_double((long long)kDeferredRow), a.phase2[j] = __longlong_as_double((long long)kDeferredRow);
            __syncthreads();
        } else {
            reduce_deep_select<UNPHASED>(a, j, p0, n, *shp);
        }
        return;
    }
    const uint32_t m = L.m;
    uint32_t N = 1;
    while (N < m) N <<= 1;
    for (uint32_t e = m + threadIdx.x; e < N; e += 256u) L.key[e] = kKeySent;
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t s = k >> 1; s > 0; s >>= 1) {
            for (uint32_t i = threadIdx.x; i < N; i += 256u) {
                const uint32_t l = i ^ s;
                if (l > i) {
                    const uint64_t x = L.key[i], y = L.key[l];
                    if ((y < x) == ((i & k) == 0u)) {
                        L.key[i] = y;
                        L.key[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    uint32_t lo1, hi1, lo2, hi2;
    if (UNPHASED) {  // src/call.rs:311-313: h1 = lower n/2 of the sorted calls, h2 = the rest
        const uint32_t ks = m / 2u;
        lo1 = 0, hi1 = ks, lo2 = ks, hi2 = m;
        if (ks >= 1u && ks < m) {
            const uint64_t va = L.key[ks - 1u] >> 15, vb = L.key[ks] >> 15;  // group bits are 0 here
            if (va == vb) {
                for (uint32_t e = threadIdx.x; e < m; e += 256u) {
                    const uint64_t k2 = L.key[e];
                    if ((k2 >> 15) == va) {
                        if (k2 & 1ull)
                            L.tie_clip = 1u;
                        else
                            L.tie_span = 1u;
                    }
                }
            }
        }
    } else {
        lo1 = 0, hi1 = L.c1, lo2 = L.c1, hi2 = m;
    }
    __syncthreads();
    const double out1 = median_of_sorted_range<CAP>(L, lo1, hi1, a.support);
    const double out2 = median_of_sorted_range<CAP>(L, lo2, hi2, a.support);
    if (threadIdx.x == 0) {
        a.phase1[j] = out1;
        a.phase2[j] = out2;
        if (UNPHASED && L.tie_span && L.tie_clip) atomicAdd((unsigned long long *)&a.status->ties, 1ull);
    }
    __syncthreads();
}

}  // namespace inq
