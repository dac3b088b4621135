// kernels.hip — the two gfx950 kernels of the `inquiSTR call` hot path.
//
//   locus_call_small : one wavefront per locus (<= 64 offered reads).  Walks every read's
//                      CIGAR (cigar_walk.h), keeps the per-read Call in the lane that owns
//                      the read and reduces the locus to its two medians in registers.
//                      Loci with more reads are appended to a work list.
//   locus_call_big   : one 256-thread workgroup per work-list locus; the four waves share the
//                      reads, per-read Calls go through a global scratch, the medians are
//                      found by rank counting over the scratch.
//
// Reference semantics restated here (wdecoster/inquiSTR v0.13.0):
//   genotype_repeat_unphased  src/call.rs:279-327   sort by value, split at n/2
//   genotype_repeat_phased    src/call.rs:329-374   bin by HP
//   median_str_length         src/call.rs:497-522
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cigar_walk.h"
#include "kernels.h"

namespace inq {

__device__ __forceinline__ double qnan() { return __builtin_nan(""); }

// (value, index) strict ordering used for every rank below: ties by file order
__device__ __forceinline__ bool before(int64_t vj, int j, int64_t v, int i) {
    return vj < v || (vj == v && j < i);
}

// XCD-aware block remap: hardware deals consecutive workgroups round-robin over the 8 XCDs;
// give each XCD one contiguous eighth of the loci so neighbouring loci (which share reads in
// real data) meet in the same L2.  Speed only: any placement computes the same result.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t per_xcd) { return (b & 7u) * per_xcd + (b >> 3); }

// A wave holds the per-read Calls of one locus in registers: element (slot s, lane l) is read
// 64*s + l of the locus, E slots per lane (E = 1: up to 64 reads, E = 4: up to 256).
template <int E>
struct Masks {
    uint64_t m[E];
    __device__ __forceinline__ uint32_t count() const {
        uint32_t c = 0;
#pragma unroll
        for (int s = 0; s < E; ++s) c += (uint32_t)__popcll(m[s]);
        return c;
    }
    __device__ __forceinline__ bool any() const {
        uint64_t o = 0;
#pragma unroll
        for (int s = 0; s < E; ++s) o |= m[s];
        return o != 0ull;
    }
};

// Order of the Calls.  Every rank below orders by (value, file order).  When every kept value fits
// the key, (value << IDX_BITS | index) is packed into ONE i32 so a rank step costs one v_readlane + one
// v_cmp per own element; otherwise the i64 value is compared with the index as tie-break.
template <bool FIT, int E>
struct LaneOrder {
    static constexpr int IDX_BITS = (E == 1) ? 6 : 8;
    int64_t val[E];
    int32_t key_asc[E];   // (val << IDX_BITS) | index
    int32_t key_desc[E];  // (val << IDX_BITS) | (max_index - index): "larger value first, then earlier read"
    int lane;
    __device__ __forceinline__ LaneOrder(const int64_t (&v)[E], int l) : lane(l) {
#pragma unroll
        for (int s = 0; s < E; ++s) {
            val[s] = v[s];
            const uint32_t idx = (uint32_t)(s * 64 + l);
            key_asc[s] = (int32_t)(((uint32_t)(int32_t)v[s] << IDX_BITS) | idx);
            key_desc[s] = (int32_t)(((uint32_t)(int32_t)v[s] << IDX_BITS) | ((uint32_t)(E * 64 - 1) - idx));
        }
    }
    // r[t] = number of elements in `mask` that sort before own element t
    __device__ __forceinline__ void rank_asc(const Masks<E> &mask, uint32_t (&r)[E]) const {
#pragma unroll
        for (int t = 0; t < E; ++t) r[t] = 0;
#pragma unroll
        for (int s = 0; s < E; ++s) {
            for (uint64_t mk = mask.m[s]; mk; mk &= mk - 1) {
                const int j = __builtin_ctzll(mk);
                if (FIT) {
                    const int32_t kj = (int32_t)readlane_u32((uint32_t)key_asc[s], j);
#pragma unroll
                    for (int t = 0; t < E; ++t) r[t] += (kj < key_asc[t]) ? 1u : 0u;
                } else {
                    const int64_t vj = readlane_i64(val[s], j);
#pragma unroll
                    for (int t = 0; t < E; ++t) r[t] += before(vj, s * 64 + j, val[t], t * 64 + lane) ? 1u : 0u;
                }
            }
        }
    }
    // r[t] = number of elements in `mask` with a larger value than own element t (ties: earlier read first)
    __device__ __forceinline__ void rank_desc(const Masks<E> &mask, uint32_t (&r)[E]) const {
#pragma unroll
        for (int t = 0; t < E; ++t) r[t] = 0;
#pragma unroll
        for (int s = 0; s < E; ++s) {
            for (uint64_t mk = mask.m[s]; mk; mk &= mk - 1) {
                const int j = __builtin_ctzll(mk);
                if (FIT) {
                    const int32_t kj = (int32_t)readlane_u32((uint32_t)key_desc[s], j);
#pragma unroll
                    for (int t = 0; t < E; ++t) r[t] += (kj > key_desc[t]) ? 1u : 0u;
                } else {
                    const int64_t vj = readlane_i64(val[s], j);
#pragma unroll
                    for (int t = 0; t < E; ++t)
                        r[t] += (vj > val[t] || (vj == val[t] && s * 64 + j < t * 64 + lane)) ? 1u : 0u;
                }
            }
        }
    }
    // value of the element of `mask` whose rank (as computed in r) equals `want`; wave-uniform
    __device__ __forceinline__ int64_t pick(const Masks<E> &mask, const uint32_t (&r)[E], uint32_t want) const {
        int64_t out = 0;
#pragma unroll
        for (int s = 0; s < E; ++s) {
            const uint64_t b = ballot64(((mask.m[s] >> lane) & 1ull) && r[s] == want);
            if (b) out = readlane_i64(val[s], __builtin_ctzll(b));
        }
        return out;
    }
};

// median_str_length (src/call.rs:497-522) for the elements flagged in `g`.  Wave-uniform result.
template <bool FIT, int E>
__device__ __forceinline__ double median_in_lanes(const Masks<E> &g, const Masks<E> &clip, const LaneOrder<FIT, E> &o,
                                                  uint32_t support) {
    const int lane = o.lane;
    const uint32_t ng = g.count();
    if (ng < support) return qnan();  // :498-500
    Masks<E> cm, chosen;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        cm.m[s] = g.m[s] & clip.m[s];
        chosen.m[s] = g.m[s] & ~clip.m[s];
    }
    const uint32_t ns = chosen.count();
    if (ns <= support && cm.any()) {  // :509-513: add the largest (support - ns) clipped values
        const uint32_t take = support - ns;
        if (take > 0u) {
            uint32_t drank[E];
            o.rank_desc(cm, drank);
#pragma unroll
            for (int s = 0; s < E; ++s) chosen.m[s] |= ballot64(((cm.m[s] >> lane) & 1ull) && drank[s] < take);
        }
    }
    const uint32_t M = chosen.count();  // >= 1 because support >= 1
    uint32_t arank[E];
    o.rank_asc(chosen, arank);
    const int64_t vhi = o.pick(chosen, arank, M / 2u);
    if (M & 1u) return (double)vhi;  // :520
    const int64_t vlo = o.pick(chosen, arank, M / 2u - 1u);
    return (double)(vlo + vhi) / 2.0;  // :515-518
}

// The two medians of one locus from the per-read (val, meta) held in registers.
template <bool UNPHASED, bool FIT, int E>
__device__ __forceinline__ void reduce_locus_in_lanes(const int64_t (&val)[E], const uint32_t (&meta)[E], int lane,
                                                      uint32_t support, double &out1, double &out2, bool &tie) {
    const LaneOrder<FIT, E> o(val, lane);
    Masks<E> kept, clip;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        kept.m[s] = ballot64(meta[s] & PM_KEPT);
        clip.m[s] = ballot64((meta[s] & PM_KEPT) && (meta[s] & PM_CLIP));
    }
    tie = false;
    if (UNPHASED) {
        // src/call.rs:311-313: sort by value (ties: file order), h1 = lower n/2, h2 = the rest
        const uint32_t mcount = kept.count();
        const uint32_t ks = mcount / 2u;
        uint32_t rank[E];
        o.rank_asc(kept, rank);
        if (!clip.any()) {
            // no soft-clipped call at this locus: every group member is "spanning", so the
            // within-group order is the global order and the medians can be read off `rank`
            auto med = [&](uint32_t base, uint32_t cnt) -> double {
                if (cnt < support) return qnan();
                if (cnt & 1u) return (double)o.pick(kept, rank, base + cnt / 2u);
                return (double)(o.pick(kept, rank, base + cnt / 2u - 1u) + o.pick(kept, rank, base + cnt / 2u)) / 2.0;
            };
            out1 = med(0u, ks);
            out2 = med(ks, mcount - ks);
        } else {
            Masks<E> g1, g2;
#pragma unroll
            for (int s = 0; s < E; ++s) {
                g1.m[s] = ballot64(((kept.m[s] >> lane) & 1ull) && rank[s] < ks);
                g2.m[s] = kept.m[s] & ~g1.m[s];
            }
            if (ks >= 1u && ks < mcount) {
                const int64_t va = o.pick(kept, rank, ks - 1u), vb = o.pick(kept, rank, ks);
                if (va == vb) {
                    bool has_clip = false, has_span = false;
#pragma unroll
                    for (int s = 0; s < E; ++s) {
                        const uint64_t eq = ballot64(((kept.m[s] >> lane) & 1ull) && val[s] == va);
                        has_clip |= (eq & clip.m[s]) != 0ull;
                        has_span |= (eq & ~clip.m[s]) != 0ull;
                    }
                    tie = has_clip && has_span;
                }
            }
            out1 = median_in_lanes<FIT, E>(g1, clip, o, support);
            out2 = median_in_lanes<FIT, E>(g2, clip, o, support);
        }
    } else {
        Masks<E> g1, g2;
#pragma unroll
        for (int s = 0; s < E; ++s) {
            g1.m[s] = ballot64(((meta[s] >> PM_GRP_SHIFT) & 3u) == 1u);
            g2.m[s] = ballot64(((meta[s] >> PM_GRP_SHIFT) & 3u) == 2u);
        }
        out1 = median_in_lanes<FIT, E>(g1, clip, o, support);  // src/call.rs:367
        out2 = median_in_lanes<FIT, E>(g2, clip, o, support);  // src/call.rs:368
    }
}

// One locus with up to 64*E offered reads, on one wave.
template <bool UNPHASED, int AUX, int E>
__device__ __forceinline__ void wave_locus(const KArgs &a, uint64_t j, uint64_t p0, int n, uint32_t start, uint32_t end,
                                           int lane, WaveLds &L) {
    Window W;
    W.se = start - 10u;
    W.ee = end + 10u;
    W.se1 = W.se + 1u;
    W.width = W.ee - W.se1;
    W.minlen = a.minlen;
    BatchView b{a.cigar4, a.reads, a.pair_read, a.n_reads, a.n_cigar4};
    uint32_t status = 0;
    int64_t val[E];
    uint32_t meta[E];
#pragma unroll
    for (int s = 0; s < E; ++s) {
        val[s] = 0;
        meta[s] = 0;
        const int cnt = min(64, n - s * 64);
        if (cnt > 0) {  // wave-uniform
            bool valid;
            const uint64_t first = p0 + (uint64_t)(s * 64);
            const PairMeta m = load_pair_meta(b, first, cnt, lane, status, valid);
            walk_pairs<UNPHASED, AUX>(b, m, valid, cnt, W, lane, status, L, val[s], meta[s]);
            if (a.pair_call && lane < cnt) a.pair_call[first + lane] = val[s];
            if (a.pair_bits && lane < cnt) a.pair_bits[first + lane] = (uint8_t)(meta[s] & 7u);
        }
    }
    bool tie;
    double out1, out2;
    constexpr int64_t kFit = 1ll << (31 - LaneOrder<true, E>::IDX_BITS);
    bool big_value = false;
#pragma unroll
    for (int s = 0; s < E; ++s) big_value |= (meta[s] & PM_KEPT) && (val[s] < -kFit || val[s] >= kFit);
    if (ballot64(big_value) == 0ull)
        reduce_locus_in_lanes<UNPHASED, true, E>(val, meta, lane, a.support, out1, out2, tie);
    else
        reduce_locus_in_lanes<UNPHASED, false, E>(val, meta, lane, a.support, out1, out2, tie);
    if (lane == 0) {
        a.phase1[j] = out1;
        a.phase2[j] = out2;
        if (tie) atomicAdd((unsigned long long *)&a.status->ties, 1ull);
    }
    if (status) atomicOr(&a.status->err, status);  // per lane: index / phase errors belong to the lane's read
}

constexpr int kMediumSlots = 4;  // locus_call_medium: up to 256 reads per wave

// AUX: cache policy of the CIGAR stream loads (0 = default, 2 = nt: read-once data)
template <bool UNPHASED, int AUX>
__global__ __launch_bounds__(256) void locus_call_small(KArgs a) {
    __shared__ WaveLds lds[4];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lb = xcd_remap(blockIdx.x, a.blocks_per_xcd);
    const uint64_t j = (uint64_t)lb * 4u + wave;
    if (j >= a.n_loci) return;

    const uint64_t p0 = a.locus_pair_off[j], p1 = a.locus_pair_off[j + 1];
    const uint32_t start = a.locus_start[j], end = a.locus_end[j];
    uint32_t status = 0;
    if (p1 < p0 || p1 > a.n_pairs) status |= ST_INDEX;
    if (start < 10u || end < start) status |= ST_LOCUS;  // src/call.rs:285 (u32 underflow), repeats.rs:102
    const uint64_t n64 = p1 - p0;
    if (!status && a.max_reads_hint && n64 > a.max_reads_hint) status |= ST_HINT;  // the caller's promise is broken
    if (status) {
        if (lane == 0) {
            atomicOr(&a.status->err, status);
            a.phase1[j] = qnan();
            a.phase2[j] = qnan();
        }
        return;
    }
    if (n64 > 64ull) {  // deeper locus: list it for locus_call_medium (<= 256 reads) or locus_call_big
        if (lane == 0) {
            const uint32_t kind = n64 > 64ull * kMediumSlots ? 1u : 0u;
            const uint32_t shard = blockIdx.x % kListShards;
            const uint32_t slot = atomicAdd(&a.status->list_count[kind][shard].n, 1u);
            a.worklist[((uint64_t)kind * kListShards + shard) * a.shard_cap + slot] = (uint32_t)j;
        }
        return;
    }
    wave_locus<UNPHASED, AUX, 1>(a, j, p0, (int)n64, start, end, lane, lds[wave]);
}

// Loci with 65..256 offered reads: still one wave per locus, four reads per lane.  Persistent waves
// stride over the work list that locus_call_small filled.
template <bool UNPHASED, int AUX>
__global__ __launch_bounds__(256) void locus_call_medium(KArgs a) {
    __shared__ WaveLds lds[4];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ uint32_t cnt[kListShards];
    if (threadIdx.x < kListShards) cnt[threadIdx.x] = a.status->list_count[0][threadIdx.x].n;
    __syncthreads();
    uint32_t total = 0;
    for (int k = 0; k < kListShards; ++k) total += cnt[k];
    for (uint32_t item = blockIdx.x * 4u + wave; item < total; item += gridDim.x * 4u) {
        uint32_t shard = 0, idx = item;
        while (idx >= cnt[shard]) idx -= cnt[shard++];
        const uint64_t j = a.worklist[(uint64_t)shard * a.shard_cap + idx];
        const uint64_t p0 = a.locus_pair_off[j];
        const int n = (int)(a.locus_pair_off[j + 1] - p0);
        wave_locus<UNPHASED, AUX, kMediumSlots>(a, j, p0, n, a.locus_start[j], a.locus_end[j], lane, lds[wave]);
    }
}

// ---------------------------------------------------------------------------------------------
// Deep loci.  Scratch layout: sval[p] (i64) and smeta[p] (u8) indexed by global pair number.

constexpr int kBigBlock = 64;           // reads one wave walks per block in locus_call_big_walk
constexpr uint32_t kWalkSplit = 16384;  // loci deeper than this are walked by the whole grid

struct BigShared {
    unsigned int cnt_kept, ng[3], ns[3];
    unsigned int tie_span, tie_clip;
    long long med[3][2];
    long long split_lo, split_hi;
};

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
__device__ uint64_t radix_select(const DeepStore &S, uint32_t n, Pred pred, uint32_t k, uint64_t lump_key, uint32_t lump_cnt, SelectLds &L) {
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
__device__ double deep_group_median(const DeepStore &S, uint32_t n, uint32_t g, uint32_t ng, uint32_t ns, uint32_t support, SelectLds &L) {
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
__device__ void reduce_deep_select(const KArgs &a, uint64_t j, uint64_t p0, uint32_t n, SelectLds &L) {
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

// ---- loci with more than 256 reads, stage 1: walk -----------------------------------------------
// One workgroup per listed locus; its four waves take 64-read blocks in turn through the same walker
// and leave (Call, meta) per read in the ctx's global scratch.  The kernel boundary in front of the
// reduce kernel makes the scratch visible: no fences.  The descriptors of a wave's next block are
// fetched while it walks the current one.
template <bool UNPHASED, int AUX>
__global__ __launch_bounds__(256) void locus_call_big_walk(KArgs a) {
    __shared__ WaveLds lds[4];
    __shared__ uint32_t cnt[kListShards];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < kListShards) cnt[threadIdx.x] = a.status->list_count[1][threadIdx.x].n;
    __syncthreads();
    uint32_t total = 0;
    for (int k = 0; k < kListShards; ++k) total += cnt[k];
    BatchView b{a.cigar4, a.reads, a.pair_read, a.n_reads, a.n_cigar4};
    // Every workgroup goes through the whole list.  A locus of up to kWalkSplit reads is walked by ONE workgroup (item modulo
    // grid); a deeper one (amplicon-depth pile-ups) by ALL of them, 64-read blocks dealt round-robin over every wave of the grid,
    // so that a single 100 000-read locus streams at the chip's rate instead of one workgroup's.
    for (uint32_t item = 0; item < total; ++item) {
        uint32_t shard = 0, idx = item;
        while (idx >= cnt[shard]) idx -= cnt[shard++];
        const uint64_t j = a.worklist[((uint64_t)kListShards + shard) * a.shard_cap + idx];
        const uint64_t p0 = a.locus_pair_off[j];
        const uint32_t n = (uint32_t)(a.locus_pair_off[j + 1] - p0);
        const bool shared = n > kWalkSplit;
        if (!shared && item % gridDim.x != blockIdx.x) continue;
        const uint32_t bstep = shared ? 4u * gridDim.x : 4u;  // blocks between two of this wave's
        const uint32_t start = a.locus_start[j], end = a.locus_end[j];
        Window W;
        W.se = start - 10u;
        W.ee = end + 10u;
        W.se1 = W.se + 1u;
        W.width = W.ee - W.se1;
        W.minlen = a.minlen;
        uint32_t status = 0;
        const uint32_t nblk = (n + kBigBlock - 1) / kBigBlock;
        auto blk_cnt = [&](uint32_t blk) { return blk < nblk ? (int)min((uint32_t)kBigBlock, n - blk * kBigBlock) : 0; };
        // software pipeline over this wave's blocks: A = pair index (2 ahead), B = descriptor (1 ahead)
        uint32_t blk = shared ? blockIdx.x * 4u + wave : wave;
        uint32_t ri_b = meta_stage_a(b, p0 + (uint64_t)blk * kBigBlock, blk_cnt(blk), lane);
        uint4 rd = meta_stage_b(b, ri_b);
        uint32_t ri_next = meta_stage_a(b, p0 + (uint64_t)(blk + bstep) * kBigBlock, blk_cnt(blk + bstep), lane);
        for (; blk < nblk; blk += bstep) {
            const int c = blk_cnt(blk);
            const uint64_t first = p0 + (uint64_t)blk * kBigBlock;
            bool valid;
            if (lane < c && (uint64_t)ri_b >= b.n_reads) status |= ST_INDEX;
            const PairMeta m = meta_stage_c(b, rd, (lane < c && (uint64_t)ri_b < b.n_reads) ? c : 0, lane, status, valid);
            // next block: descriptor load now (its index arrived during the previous walk), index load for the one after
            ri_b = ri_next;
            rd = meta_stage_b(b, ri_b);
            ri_next = meta_stage_a(b, p0 + (uint64_t)(blk + 2u * bstep) * kBigBlock, blk_cnt(blk + 2u * bstep), lane);
            int64_t val;
            uint32_t meta;
            walk_pairs<UNPHASED, AUX>(b, m, valid, c, W, lane, status, lds[wave], val, meta);
            if (lane < c) {
                a.sval[first + lane] = val;
                a.smeta[first + lane] = (uint8_t)meta;
                if (a.pair_call) a.pair_call[first + lane] = val;
                if (a.pair_bits) a.pair_bits[first + lane] = (uint8_t)(meta & 7u);
            }
        }
        if (status) atomicOr(&a.status->err, status);
    }
}

// ---- stage 2: reduce ------------------------------------------------------------------------------
// One workgroup per listed locus: the kept Calls become 64-bit keys
//     [63:62] haplotype group | [61:15] value + 2^46 | [14:1] file-order index | [0] clipped
// sorted once by a bitonic network in LDS; both haplotype groups are then contiguous ascending ranges
// and median_str_length's span/clip rule reduces to one prefix count of "spanning" flags.  Two
// instantiations split the list by depth so that shallow-deep loci keep several workgroups per CU:
// CAP = 2048 (16 KB of keys), 8192 (64 KB) and 16384 (128 KB, one workgroup per CU); deeper loci take the
// global rank-counting fallback inside the last launch.  The file index has 14 bits in the key.
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
__device__ double median_of_sorted_range(SortLds<CAP> &L, uint32_t lo, uint32_t hi, uint32_t support) {
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

template <bool UNPHASED, int CAP>
__global__ __launch_bounds__(256) void locus_call_big_reduce(KArgs a) {
    __shared__ SortLds<CAP> L;
    __shared__ SelectLds sh;  // for the loci no LDS sort can hold
    __shared__ uint32_t cnt[kListShards];
    if (threadIdx.x < kListShards) cnt[threadIdx.x] = a.status->list_count[1][threadIdx.x].n;
    __syncthreads();
    uint32_t total = 0;
    for (int k = 0; k < kListShards; ++k) total += cnt[k];

    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
        uint32_t shard = 0, idx = item;
        while (idx >= cnt[shard]) idx -= cnt[shard++];
        const uint64_t j = a.worklist[((uint64_t)kListShards + shard) * a.shard_cap + idx];
        const uint64_t p0 = a.locus_pair_off[j];
        const uint32_t n = (uint32_t)(a.locus_pair_off[j + 1] - p0);
        // depth classes: this launch takes (CAP/4, CAP] reads (the CAP = 2048 launch everything up to 2048);
        // the CAP = 8192 launch also takes what no sort can hold
        constexpr uint32_t kLow = CAP == 2048 ? 0u : CAP == 8192 ? 2048u : 8192u;  // this launch takes (kLow, CAP]
        if (n <= kLow || (CAP != 16384 && n > (uint32_t)CAP)) continue;
        if (n > (uint32_t)CAP) {  // only the last class: deeper than any sort can hold
            if (n > kGridSelectMin) continue;  // ... and deeper than one workgroup should reduce: the grid-wide select behind this launch (deep_select.hip)
            reduce_deep_select<UNPHASED>(a, j, p0, n, sh);
            continue;
        }
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
            reduce_deep_select<UNPHASED>(a, j, p0, n, sh);
            continue;
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
}

// Last kernel of every sequence that launched a deep-locus kernel: the work lists are empty again for the
// next sequence (also when the same sequence is replayed from a hipGraph).
__global__ void clear_lists(KArgs a) {
    if (threadIdx.x < 2 * kListShards) a.status->list_count[threadIdx.x / kListShards][threadIdx.x % kListShards].n = 0u;
}

// ---- launchers (called from capi.hip) ----
template <bool UNPHASED, int AUX>
static void launch_t(const KArgs &a, uint32_t grid_small, uint32_t grid_medium, uint32_t grid_big, hipStream_t s,
                     hipEvent_t ev_mid, void *deep_scratch) {
    if (grid_small) hipLaunchKernelGGL((locus_call_small<UNPHASED, AUX>), dim3(grid_small), dim3(256), 0, s, a);
    if (ev_mid) (void)hipEventRecord(ev_mid, s);
    // launches a promised depth makes pointless are skipped (a broken promise is flagged by locus_call_small)
    const uint32_t h = a.max_reads_hint;
    if (h && h <= 64u) return;  // no locus can be on a work list
    hipLaunchKernelGGL((locus_call_medium<UNPHASED, AUX>), dim3(grid_medium), dim3(256), 0, s, a);
    if (!(h && h <= 64u * kMediumSlots)) {
        hipLaunchKernelGGL((locus_call_big_walk<UNPHASED, AUX>), dim3(grid_big), dim3(256), 0, s, a);
        hipLaunchKernelGGL((locus_call_big_reduce<UNPHASED, 2048>), dim3(grid_big), dim3(256), 0, s, a);
        if (!(h && h <= 2048u)) hipLaunchKernelGGL((locus_call_big_reduce<UNPHASED, 8192>), dim3(grid_big), dim3(256), 0, s, a);
        if (!(h && h <= 8192u)) hipLaunchKernelGGL((locus_call_big_reduce<UNPHASED, 16384>), dim3(256), dim3(256), 0, s, a);
        if (!(h && h <= kGridSelectMin) && deep_scratch) launch_deep_select(a, UNPHASED, deep_scratch, a.n_pairs, s);
    }
    hipLaunchKernelGGL((clear_lists), dim3(1), dim3(64), 0, s, a);
}

void launch_locus_call(const KArgs &a, bool unphased, bool nt_loads, uint32_t grid_small, uint32_t grid_medium,
                       uint32_t grid_big, hipStream_t s, hipEvent_t ev_mid, void *deep_scratch) {
    if (unphased) {
        if (nt_loads)
            launch_t<true, 2>(a, grid_small, grid_medium, grid_big, s, ev_mid, deep_scratch);
        else
            launch_t<true, 0>(a, grid_small, grid_medium, grid_big, s, ev_mid, deep_scratch);
    } else {
        if (nt_loads)
            launch_t<false, 2>(a, grid_small, grid_medium, grid_big, s, ev_mid, deep_scratch);
        else
            launch_t<false, 0>(a, grid_small, grid_medium, grid_big, s, ev_mid, deep_scratch);
    }
}

// An empty launch makes the runtime load this translation unit's code object now (inq_ctx_create, on the
// context thread) instead of in front of the first real launch.
__global__ void preload_locus_kernel() {}
void preload_locus(hipStream_t s) { hipLaunchKernelGGL(preload_locus_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace inq
