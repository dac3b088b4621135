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
#include "deep_reduce.h"
#include "kernels.h"

namespace inq {

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
    if (blockIdx.x == 0 && threadIdx.x == 0)  // the tail kernel's barrier words, whatever an earlier sequence left in them
        a.status->bar_count.v = 0u, a.status->bar_abort.v = 0u, a.status->exit_ticket.v = 0u;
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
    if (n64 > 64ull) {  // deeper locus: listed for locus_call_mid_walk - medium (<= 256 reads), deep, or deeper than kWalkSplit
        if (lane == 0) {
            const uint32_t kind = n64 > kWalkSplit ? 2u : n64 > 64ull * kMediumSlots ? 1u : 0u;
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
__device__ __forceinline__ void medium_part(const KArgs &a, WaveLds (&lds)[4], const uint32_t (&cnt)[kListShards], int lane, uint32_t wave) {
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
    __syncthreads();  // the LDS goes on to the walk part
}

// ---------------------------------------------------------------------------------------------
// Deep loci.  Scratch layout: sval[p] (i64) and smeta[p] (u8) indexed by global pair number.

constexpr int kBigBlock = 64;  // reads one wave walks per block
// (kWalkSplit, kernels.h: loci deeper than this are walked by the whole grid - they are on a work list of their own)

// ---- loci with more than 256 reads: walk (and, up to kReduceInPlace reads, reduce) ------------------------------
// The four waves of a workgroup take the locus' 64-read blocks in turn through the same walker and leave (Call, meta) per read in the
// ctx's global scratch; the descriptors of a wave's next block are fetched while it walks the current one.  n_wg / wg_rank: the blocks
// are dealt over the waves of n_wg workgroups, of which this is number wg_rank (a locus of more than kWalkSplit reads is walked by a
// GROUP of workgroups - the whole grid when it is the only one -: it streams at the chip's rate, not one workgroup's).
template <bool UNPHASED, int AUX>
__device__ __forceinline__ void walk_locus(const KArgs &a, const BatchView &b, uint64_t j, uint64_t p0, uint32_t n, uint32_t n_wg, uint32_t wg_rank, int lane,
                                           uint32_t wave, WaveLds &L) {
    const uint32_t bstep = 4u * n_wg;  // blocks between two of this wave's
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
    uint32_t blk = wg_rank * 4u + wave;
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
        walk_pairs<UNPHASED, AUX>(b, m, valid, c, W, lane, status, L, val, meta);
        if (lane < c) {
            a.sval[first + lane] = val;
            a.smeta[first + lane] = (uint8_t)meta;
            if (a.pair_call) a.pair_call[first + lane] = val;
            if (a.pair_bits) a.pair_bits[first + lane] = (uint8_t)(meta & 7u);
        }
    }
    if (status) atomicOr(&a.status->err, status);
}

// What the walk and the in-place reduce need of LDS, one after the other: the four waves' walk state, then the sort's keys.
union MidLds {
    WaveLds wave[4];
    SortLds<(int)kReduceInPlace> sort;
};

template <bool UNPHASED, int AUX>
__device__ __forceinline__ void walk_part(const KArgs &a, MidLds &lds, const uint32_t (&cnt)[kListShards], const uint32_t (&cnt2)[kListShards], int lane,
                                          uint32_t wave) {
    BatchView b{a.cigar4, a.reads, a.pair_read, a.n_reads, a.n_cigar4};
    // ---- list 1 (257 .. kWalkSplit reads): each locus is ONE workgroup's, the list is dealt over the grid - a workgroup looks at its
    // own items only (round 4 and the first form of this kernel had every workgroup read through the whole list: 10 000 loci of 270
    // reads took 71 ms)
    uint32_t total = 0;
    for (int k = 0; k < kListShards; ++k) total += cnt[k];
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
        uint32_t shard = 0, idx = item;
        while (idx >= cnt[shard]) idx -= cnt[shard++];
        const uint64_t j = a.worklist[((uint64_t)kListShards + shard) * a.shard_cap + idx];
        const uint64_t p0 = a.locus_pair_off[j];
        const uint32_t n = (uint32_t)(a.locus_pair_off[j + 1] - p0);
        walk_locus<UNPHASED, AUX>(a, b, j, p0, n, 1u, 0u, lane, wave, lds.wave[wave]);
        if (n <= kReduceInPlace) {
            // reduced on the spot by the workgroup that walked it: its Calls are this CU's own stores (drained, then a barrier), the
            // sort takes the LDS the walk no longer needs, and thousands of such loci (a targeted panel at 300-fold depth) are
            // reduced by as many workgroups as walked them instead of by the tail kernel's one per CU
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            sort_reduce_locus<UNPHASED, (int)kReduceInPlace, true>(a, j, p0, n, lds.sort, nullptr);  // (a Call beyond 47 bits: left to the tail)
        }
        __syncthreads();  // the LDS goes back to the walk
    }
    // ---- list 2 (more than kWalkSplit reads: amplicon pile-ups): a locus is walked by a GROUP of workgroups - all of them when it
    // is the only one, gridDim / items when there are many (1 000 loci of 18 000 reads: eight workgroups each; one group after the
    // other through all of them took 10.9 ms of a 12 ms sequence)
    total = 0;
    for (int k = 0; k < kListShards; ++k) total += cnt2[k];
    if (total == 0u) return;
    const uint32_t per_item = max(1u, gridDim.x / total), groups = gridDim.x / per_item;
    const uint32_t my_group = blockIdx.x / per_item, my_rank = blockIdx.x % per_item;
    if (my_group >= groups) return;  // (gridDim % per_item workgroups are left over)
    for (uint32_t item = my_group; item < total; item += groups) {
        uint32_t shard = 0, idx = item;
        while (idx >= cnt2[shard]) idx -= cnt2[shard++];
        const uint64_t j = a.worklist[((uint64_t)2 * kListShards + shard) * a.shard_cap + idx];
        const uint64_t p0 = a.locus_pair_off[j];
        const uint64_t n64 = a.locus_pair_off[j + 1] - p0;
        if (n64 > 0xffffffffull) continue;  // (flagged by the tail kernel: outside what the scratch indexing covers)
        walk_locus<UNPHASED, AUX>(a, b, j, p0, (uint32_t)n64, per_item, my_rank, lane, wave, lds.wave[wave]);
    }
}

// ONE kernel behind locus_call_small for everything deeper than 64 reads that needs the CIGARs: the medium loci are called outright,
// the deeper ones walked into the scratch - and reduced there and then up to kReduceInPlace reads; the reduce of the rest is
// locus_call_tail (deep_select.hip).  The lists are mostly empty: a launch that finds them so costs its launch and nothing else -
// round 4 had two launches here (and ~36 more behind them).
template <bool UNPHASED, int AUX>
__global__ __launch_bounds__(256) void locus_call_mid_walk(KArgs a) {
    __shared__ MidLds lds;
    __shared__ uint32_t cnt[kListKinds][kListShards];  // the three lists' lengths, read once (an idle launch is these 96 loads)
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < kListKinds * kListShards) cnt[threadIdx.x / kListShards][threadIdx.x % kListShards] = a.status->list_count[threadIdx.x / kListShards][threadIdx.x % kListShards].n;
    __syncthreads();
    medium_part<UNPHASED, AUX>(a, lds.wave, cnt[0], lane, wave);
    walk_part<UNPHASED, AUX>(a, lds, cnt[1], cnt[2], lane, wave);
}

// Empties the work lists behind a sequence that cannot have filled the deep ones (launch_t).
__global__ void clear_lists(KArgs a) {
    if (threadIdx.x < kListKinds * kListShards) a.status->list_count[threadIdx.x / kListShards][threadIdx.x % kListShards].n = 0u;
}

// ---- launchers (called from capi.hip) ----
// The sequence: locus_call_small, and - unless the caller's depth hint rules deeper loci out - locus_call_mid_walk and the persistent
// locus_call_tail (deep_select.hip), which also empties the work lists for the next sequence (a hint of at most 256 reads: a
// one-wave clear_lists in its place).  Three launches whatever the batch holds; with nothing deep the last two find empty lists
// and leave at once.
template <bool UNPHASED, int AUX>
static void launch_t(const KArgs &a, uint32_t grid_small, uint32_t grid_medium, uint32_t grid_tail, hipStream_t s,
                     hipEvent_t ev_mid, void *deep_scratch) {
    if (grid_small) hipLaunchKernelGGL((locus_call_small<UNPHASED, AUX>), dim3(grid_small), dim3(256), 0, s, a);
    if (ev_mid) (void)hipEventRecord(ev_mid, s);
    // launches a promised depth makes pointless are skipped (a broken promise is flagged by locus_call_small)
    const uint32_t h = a.max_reads_hint;
    if (h && h <= 64u) return;  // no locus can be on a work list
    hipLaunchKernelGGL((locus_call_mid_walk<UNPHASED, AUX>), dim3(grid_medium), dim3(256), 0, s, a);
    if (h && h <= 64u * kMediumSlots) {
        // nothing can be on the deep list: the medium list is emptied by a one-wave kernel instead of the persistent tail, whose
        // workgroups need a whole CU's LDS each and would wait for a CU to drain while a span's inflate (eight 20 KB workgroups
        // per CU, on the ahead stream) is in flight - the CLI's case for data of 65 - 256-fold depth
        hipLaunchKernelGGL(clear_lists, dim3(1), dim3(128), 0, s, a);
        return;
    }
    launch_locus_tail(a, UNPHASED, deep_scratch, a.n_pairs, grid_tail, s);
}

void launch_locus_call(const KArgs &a, bool unphased, bool nt_loads, uint32_t grid_small, uint32_t grid_medium,
                       uint32_t grid_tail, hipStream_t s, hipEvent_t ev_mid, void *deep_scratch) {
    if (unphased) {
        if (nt_loads)
            launch_t<true, 2>(a, grid_small, grid_medium, grid_tail, s, ev_mid, deep_scratch);
        else
            launch_t<true, 0>(a, grid_small, grid_medium, grid_tail, s, ev_mid, deep_scratch);
    } else {
        if (nt_loads)
            launch_t<false, 2>(a, grid_small, grid_medium, grid_tail, s, ev_mid, deep_scratch);
        else
            launch_t<false, 0>(a, grid_small, grid_medium, grid_tail, s, ev_mid, deep_scratch);
    }
}

// An empty launch makes the runtime load this translation unit's code object now (inq_ctx_create, on the
// context thread) instead of in front of the first real launch.
__global__ void preload_locus_kernel() {}
void preload_locus(hipStream_t s) { hipLaunchKernelGGL(preload_locus_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace inq
