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

// Order of the per-read Calls inside one wave (one read per lane).  Every rank below orders by
// (value, file order).  When every kept value fits 25 bits the pair is packed into ONE i32 key
// (value << 6 | lane) so a rank costs one v_readlane + one v_cmp per element; otherwise the
// comparison is done on the i64 value with the lane as tie-break.
template <bool FIT>
struct LaneOrder {
    int64_t val;
    int32_t key_asc;   // (val << 6) | lane
    int32_t key_desc;  // (val << 6) | (63 - lane): "larger value first, then earlier read"
    int lane;
    __device__ __forceinline__ LaneOrder(int64_t v, int l) : val(v), lane(l) {
        key_asc = (int32_t)(((uint32_t)(int32_t)v << 6) | (uint32_t)l);
        key_desc = (int32_t)(((uint32_t)(int32_t)v << 6) | (uint32_t)(63 - l));
    }
    // number of lanes j in `mask` whose (value, j) sorts before this lane's
    __device__ __forceinline__ uint32_t rank_asc(uint64_t mask) const {
        uint32_t r = 0;
        for (uint64_t mk = mask; mk; mk &= mk - 1) {
            const int j = __builtin_ctzll(mk);
            if (FIT) {
                r += ((int32_t)readlane_u32((uint32_t)key_asc, j) < key_asc) ? 1u : 0u;
            } else {
                const int64_t vj = readlane_i64(val, j);
                r += before(vj, j, val, lane) ? 1u : 0u;
            }
        }
        return r;
    }
    // number of lanes j in `mask` with a larger value (ties: earlier read first)
    __device__ __forceinline__ uint32_t rank_desc(uint64_t mask) const {
        uint32_t r = 0;
        for (uint64_t mk = mask; mk; mk &= mk - 1) {
            const int j = __builtin_ctzll(mk);
            if (FIT) {
                r += ((int32_t)readlane_u32((uint32_t)key_desc, j) > key_desc) ? 1u : 0u;
            } else {
                const int64_t vj = readlane_i64(val, j);
                r += (vj > val || (vj == val && j < lane)) ? 1u : 0u;
            }
        }
        return r;
    }
};

// median_str_length (src/call.rs:497-522) for the elements flagged in `gmask`, one element per
// lane.  Wave-uniform result.
template <bool FIT>
__device__ __forceinline__ double median_in_lanes(uint64_t gmask, uint64_t clipmask, const LaneOrder<FIT> &o,
                                                  uint32_t support) {
    const int lane = o.lane;
    const uint32_t ng = (uint32_t)__popcll(gmask);
    if (ng < support) return qnan();  // :498-500
    const uint64_t cm = gmask & clipmask, sm = gmask & ~clipmask;
    const uint32_t ns = (uint32_t)__popcll(sm);
    uint64_t chosen = sm;
    if (ns <= support && cm != 0ull) {  // :509-513: add the largest (support - ns) clipped values
        const uint32_t take = support - ns;
        if (take > 0u) {
            const uint32_t drank = o.rank_desc(cm);
            chosen |= ballot64(((cm >> lane) & 1ull) && drank < take);
        }
    }
    const uint32_t M = (uint32_t)__popcll(chosen);  // >= 1 because support >= 1
    const uint32_t arank = o.rank_asc(chosen);
    const bool mine = (chosen >> lane) & 1ull;
    const int lhi = __builtin_ctzll(ballot64(mine && arank == M / 2u));
    const int64_t vhi = readlane_i64(o.val, lhi);
    if (M & 1u) return (double)vhi;  // :520
    const int llo = __builtin_ctzll(ballot64(mine && arank == M / 2u - 1u));
    const int64_t vlo = readlane_i64(o.val, llo);
    return (double)(vlo + vhi) / 2.0;  // :515-518
}

// The two medians of one locus from the per-read (val, meta) held one per lane.
template <bool UNPHASED, bool FIT>
__device__ __forceinline__ void reduce_locus_in_lanes(int64_t val, uint32_t meta, int lane, uint32_t support,
                                                      double &out1, double &out2, bool &tie) {
    const LaneOrder<FIT> o(val, lane);
    const uint64_t kept = ballot64(meta & PM_KEPT);
    const uint64_t clipmask = ballot64((meta & PM_KEPT) && (meta & PM_CLIP));
    tie = false;
    if (UNPHASED) {
        // src/call.rs:311-313: sort by value (ties: file order), h1 = lower n/2, h2 = the rest
        const uint32_t mcount = (uint32_t)__popcll(kept);
        const uint32_t ks = mcount / 2u;
        const uint32_t rank = o.rank_asc(kept);
        const bool mine = (kept >> lane) & 1ull;
        auto pick = [&](uint32_t r) -> int64_t {
            return readlane_i64(val, __builtin_ctzll(ballot64(mine && rank == r)));
        };
        if (clipmask == 0ull) {
            // no soft-clipped call at this locus: every group member is "spanning", so the
            // within-group order is the global order and the medians can be read off `rank`
            auto med = [&](uint32_t base, uint32_t cnt) -> double {
                if (cnt < support) return qnan();
                if (cnt & 1u) return (double)pick(base + cnt / 2u);
                return (double)(pick(base + cnt / 2u - 1u) + pick(base + cnt / 2u)) / 2.0;
            };
            out1 = med(0u, ks);
            out2 = med(ks, mcount - ks);
        } else {
            const uint64_t g1 = ballot64(mine && rank < ks);
            const uint64_t g2 = kept & ~g1;
            if (ks >= 1u && ks < mcount) {
                const int64_t va = pick(ks - 1u), vb = pick(ks);
                if (va == vb) {
                    const uint64_t eq = ballot64(mine && val == va);
                    tie = (eq & clipmask) != 0ull && (eq & ~clipmask) != 0ull;
                }
            }
            out1 = median_in_lanes<FIT>(g1, clipmask, o, support);
            out2 = median_in_lanes<FIT>(g2, clipmask, o, support);
        }
    } else {
        const uint64_t g1 = ballot64(((meta >> PM_GRP_SHIFT) & 3u) == 1u);
        const uint64_t g2 = ballot64(((meta >> PM_GRP_SHIFT) & 3u) == 2u);
        out1 = median_in_lanes<FIT>(g1, clipmask, o, support);  // src/call.rs:367
        out2 = median_in_lanes<FIT>(g2, clipmask, o, support);  // src/call.rs:368
    }
}

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
    if (status) {
        if (lane == 0) {
            atomicOr(&a.status->err, status);
            a.phase1[j] = qnan();
            a.phase2[j] = qnan();
        }
        return;
    }
    const uint64_t n64 = p1 - p0;
    if (n64 > 64ull) {  // deep locus: hand over to locus_call_big
        if (lane == 0) {
            const uint32_t slot = atomicAdd(&a.status->big_count[a.parity], 1u);
            a.worklist[slot] = (uint32_t)j;
        }
        return;
    }
    const int n = (int)n64;
    Window W;
    W.se = start - 10u;
    W.ee = end + 10u;
    W.se1 = W.se + 1u;
    W.width = W.ee - W.se1;
    W.minlen = a.minlen;

    BatchView b{a.cigar4, a.reads, a.pair_read, a.n_reads, a.n_cigar4};
    bool valid;
    const PairMeta m = load_pair_meta(b, p0, n, lane, status, valid);

    int64_t val;
    uint32_t meta;
    walk_pairs<UNPHASED, AUX>(b, m, valid, n, W, lane, status, lds[wave], val, meta);
    if (a.pair_call && lane < n) a.pair_call[p0 + lane] = val;
    if (a.pair_bits && lane < n) a.pair_bits[p0 + lane] = (uint8_t)(meta & 7u);

    bool tie;
    double out1, out2;
    const bool big_value = (meta & PM_KEPT) && (val < -(1ll << 24) || val >= (1ll << 24));
    if (ballot64(big_value) == 0ull)
        reduce_locus_in_lanes<UNPHASED, true>(val, meta, lane, a.support, out1, out2, tie);
    else
        reduce_locus_in_lanes<UNPHASED, false>(val, meta, lane, a.support, out1, out2, tie);
    if (lane == 0) {
        a.phase1[j] = out1;
        a.phase2[j] = out2;
        if (tie) atomicAdd((unsigned long long *)&a.status->ties, 1ull);
    }
    if (status) atomicOr(&a.status->err, status);  // per lane: index / phase errors belong to the lane's read
}

// ---------------------------------------------------------------------------------------------
// Deep loci.  Scratch layout: sval[p] (i64) and smeta[p] (u8) indexed by global pair number.

constexpr int kBigPairsPerWave = 16;

struct BigShared {
    unsigned int cnt_kept, ng[3], ns[3];
    unsigned int tie_span, tie_clip;
    long long med[3][2];
    long long split_lo, split_hi;
};

// rank counting over the scratch for one group; returns via sh.med[g]
__device__ void big_group_median(const KArgs &a, uint64_t p0, uint32_t n, int g, uint32_t support, BigShared &sh,
                                 double &out) {
    const uint32_t ng = sh.ng[g], ns = sh.ns[g];
    if (ng < support) {  // uniform over the block
        out = qnan();
        return;
    }
    const uint32_t take = (ns <= support) ? support - ns : 0u;
    // chosen = spans of the group, plus the `take` largest clips (src/call.rs:509-513)
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        uint32_t me = a.smeta[p0 + e];
        const bool ing = (me & PM_KEPT) && ((me >> PM_GRP_SHIFT) & 3u) == (uint32_t)g;
        bool ch = false;
        if (ing) {
            if (!(me & PM_CLIP))
                ch = true;
            else if (take > 0u) {
                const int64_t v = a.sval[p0 + e];
                uint32_t drank = 0;
                for (uint32_t jx = 0; jx < n && drank < take; ++jx) {
                    const uint32_t mj = a.smeta[p0 + jx];
                    if ((mj & PM_KEPT) && (mj & PM_CLIP) && ((mj >> PM_GRP_SHIFT) & 3u) == (uint32_t)g) {
                        const int64_t vj = a.sval[p0 + jx];
                        drank += (vj > v || (vj == v && jx < e)) ? 1u : 0u;
                    }
                }
                ch = drank < take;
            }
        }
        // the chosen bit of group 1 must not leak into group 2: it is rewritten per group
        me = ch ? (me | PM_CHOSEN) : (me & ~PM_CHOSEN);
        a.smeta[p0 + e] = (uint8_t)me;
    }
    __threadfence();
    __syncthreads();
    const uint32_t M = (ns > support) ? ns : support;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t me = a.smeta[p0 + e];
        if (!((me & PM_CHOSEN) && ((me >> PM_GRP_SHIFT) & 3u) == (uint32_t)g)) continue;
        const int64_t v = a.sval[p0 + e];
        uint32_t arank = 0;
        for (uint32_t jx = 0; jx < n; ++jx) {
            const uint32_t mj = a.smeta[p0 + jx];
            if ((mj & PM_CHOSEN) && ((mj >> PM_GRP_SHIFT) & 3u) == (uint32_t)g) {
                const int64_t vj = a.sval[p0 + jx];
                arank += before(vj, (int)jx, v, (int)e) ? 1u : 0u;
            }
        }
        if (arank == M / 2u) sh.med[g][1] = v;
        if (!(M & 1u) && arank == M / 2u - 1u) sh.med[g][0] = v;
    }
    __syncthreads();
    if (M & 1u)
        out = (double)sh.med[g][1];
    else
        out = (double)(sh.med[g][0] + sh.med[g][1]) / 2.0;
    __syncthreads();
}

template <bool UNPHASED, int AUX>
__global__ __launch_bounds__(256) void locus_call_big(KArgs a) {
    __shared__ BigShared sh;
    __shared__ WaveLds lds[4];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_big = a.status->big_count[a.parity];
    // the other parity's counter belongs to the next call on this ctx: clear it here, one kernel
    // boundary before locus_call_small of that call increments it
    if (blockIdx.x == 0 && threadIdx.x == 0) a.status->big_count[a.parity ^ 1u] = 0u;

    for (uint32_t item = blockIdx.x; item < n_big; item += gridDim.x) {
        const uint64_t j = a.worklist[item];
        const uint64_t p0 = a.locus_pair_off[j];
        const uint32_t n = (uint32_t)(a.locus_pair_off[j + 1] - p0);
        const uint32_t start = a.locus_start[j], end = a.locus_end[j];
        Window W;
        W.se = start - 10u;
        W.ee = end + 10u;
        W.se1 = W.se + 1u;
        W.width = W.ee - W.se1;
        W.minlen = a.minlen;
        BatchView b{a.cigar4, a.reads, a.pair_read, a.n_reads, a.n_cigar4};
        uint32_t status = 0;
        if (threadIdx.x == 0) {
            sh.cnt_kept = 0;
            for (int g = 0; g < 3; ++g) sh.ng[g] = sh.ns[g] = 0;
            sh.tie_span = sh.tie_clip = 0;
        }
        // ---- walk: the 4 waves take blocks of kBigPairsPerWave reads in turn ----
        const uint32_t nblk = (n + kBigPairsPerWave - 1) / kBigPairsPerWave;
        for (uint32_t blk = wave; blk < nblk; blk += 4u) {
            const uint64_t first = p0 + (uint64_t)blk * kBigPairsPerWave;
            const int cnt = (int)min((uint32_t)kBigPairsPerWave, n - blk * kBigPairsPerWave);
            bool valid;
            const PairMeta m = load_pair_meta(b, first, cnt, lane, status, valid);
                    int64_t val;
            uint32_t meta;
            walk_pairs<UNPHASED, AUX>(b, m, valid, cnt, W, lane, status, lds[wave], val, meta);
            if (lane < cnt) {
                a.sval[first + lane] = val;
                a.smeta[first + lane] = (uint8_t)meta;
                if (a.pair_call) a.pair_call[first + lane] = val;
                if (a.pair_bits) a.pair_bits[first + lane] = (uint8_t)(meta & 7u);
            }
        }
        if (status) atomicOr(&a.status->err, status);
        __threadfence();
        __syncthreads();

        // ---- unphased: global rank -> haplotype group (src/call.rs:311-313) ----
        if (UNPHASED) {
            uint32_t local = 0;
            for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) local += (a.smeta[p0 + e] & PM_KEPT) ? 1u : 0u;
            if (local) atomicAdd(&sh.cnt_kept, local);
            __syncthreads();
            const uint32_t mcount = sh.cnt_kept, ks = mcount / 2u;
            for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
                uint32_t me = a.smeta[p0 + e];
                if (!(me & PM_KEPT)) continue;
                const int64_t v = a.sval[p0 + e];
                uint32_t rank = 0;
                for (uint32_t jx = 0; jx < n; ++jx)
                    if (a.smeta[p0 + jx] & PM_KEPT) rank += before(a.sval[p0 + jx], (int)jx, v, (int)e) ? 1u : 0u;
                const uint32_t grp = rank < ks ? 1u : 2u;
                // group bits live in a second byte plane until every rank is known: other threads
                // still read PM_KEPT of this byte, and KEPT is not modified by this write
                me = (me & ~(3u << PM_GRP_SHIFT)) | (grp << PM_GRP_SHIFT);
                a.smeta[p0 + e] = (uint8_t)me;
                if (ks >= 1u && rank == ks - 1u) sh.split_lo = v;
                if (rank == ks) sh.split_hi = v;
            }
            __threadfence();
            __syncthreads();
            if (ks >= 1u && ks < mcount && sh.split_lo == sh.split_hi) {
                const int64_t vs = sh.split_lo;
                for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
                    const uint32_t me = a.smeta[p0 + e];
                    if ((me & PM_KEPT) && a.sval[p0 + e] == vs) {
                        if (me & PM_CLIP)
                            sh.tie_clip = 1u;
                        else
                            sh.tie_span = 1u;
                    }
                }
            }
            __syncthreads();
        }
        // ---- group sizes ----
        {
            uint32_t c_ng[3] = {0, 0, 0}, c_ns[3] = {0, 0, 0};
            for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
                const uint32_t me = a.smeta[p0 + e];
                if (!(me & PM_KEPT)) continue;
                const uint32_t g = (me >> PM_GRP_SHIFT) & 3u;
                if (g == 1u || g == 2u) {
                    c_ng[g]++;
                    if (!(me & PM_CLIP)) c_ns[g]++;
                }
            }
            for (int g = 1; g <= 2; ++g) {
                if (c_ng[g]) atomicAdd(&sh.ng[g], c_ng[g]);
                if (c_ns[g]) atomicAdd(&sh.ns[g], c_ns[g]);
            }
        }
        __syncthreads();
        double out1, out2;
        big_group_median(a, p0, n, 1, a.support, sh, out1);
        big_group_median(a, p0, n, 2, a.support, sh, out2);
        if (threadIdx.x == 0) {
            a.phase1[j] = out1;
            a.phase2[j] = out2;
            if (UNPHASED && sh.tie_span && sh.tie_clip) atomicAdd((unsigned long long *)&a.status->ties, 1ull);
        }
        __syncthreads();
    }
}

// ---- launchers (called from capi.hip) ----
template <bool UNPHASED, int AUX>
static void launch_t(const KArgs &a, uint32_t grid_small, uint32_t grid_big, hipStream_t s, hipEvent_t ev_mid) {
    if (grid_small) hipLaunchKernelGGL((locus_call_small<UNPHASED, AUX>), dim3(grid_small), dim3(256), 0, s, a);
    if (ev_mid) (void)hipEventRecord(ev_mid, s);
    hipLaunchKernelGGL((locus_call_big<UNPHASED, AUX>), dim3(grid_big), dim3(256), 0, s, a);
}

void launch_locus_call(const KArgs &a, bool unphased, bool nt_loads, uint32_t grid_small, uint32_t grid_big,
                       hipStream_t s, hipEvent_t ev_mid) {
    if (unphased) {
        if (nt_loads)
            launch_t<true, 2>(a, grid_small, grid_big, s, ev_mid);
        else
            launch_t<true, 0>(a, grid_small, grid_big, s, ev_mid);
    } else {
        if (nt_loads)
            launch_t<false, 2>(a, grid_small, grid_big, s, ev_mid);
        else
            launch_t<false, 0>(a, grid_small, grid_big, s, ev_mid);
    }
}

}  // namespace inq
