// deep_select.hip — the reduce of a VERY deep locus (more than kGridSelectMin offered reads: amplicon pile-ups) over the whole grid.
//
// median_str_length (src/call.rs:497-522), the unphased split (:308-322) and the phased bins (:341-369) of such a locus used to run
// on ONE workgroup: ~60 passes over the locus' per-read Calls (9 B per read in the ctx scratch), each pass 256 threads wide - a
// serial tail of tens of milliseconds at 10^6 reads behind a walk that already streams at the chip's rate.  Here every pass is a
// launch of its own over the whole grid: a workgroup histograms its slice of the Calls in LDS and adds the bins it met to the
// locus' global histogram of that pass; the NEXT launch starts by deriving, in every workgroup alike, which bin the wanted rank
// fell into (a 256-wide scan per finished pass: microseconds), i.e. no workgroup ever waits for another inside a kernel and the
// kernel boundary is the only synchronisation.  A most-significant-byte-first radix select of a 64-bit key is 8 such launches;
// the order statistics a locus needs:
//     unphased: the split value (rank mcount / 2 of the kept Calls) and, among Calls equal to it, the first r in FILE ORDER
//               (a prefix count over per-slice counts) - then, per haplotype group as in the phased case:
//     the clip threshold (rank nc - take of the group's clipped Calls, only when spanning <= support),
//     the upper median (rank M / 2 of the chosen Calls), and the lower one, which is the upper one again unless exactly M / 2
//     chosen Calls lie below it - then it is their maximum: one more pass, not eight.
// Every kernel loops over ALL very deep loci of the work list (their states lie side by side), so the number of launches does not
// depend on how many there are: 31 unphased / 21 phased, ~4 us each when there is nothing to do (the launch sequence is skipped
// altogether when the caller's depth hint rules such loci out).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cigar_walk.h"
#include "kernels.h"
#include "wave_primitives.h"

namespace inq {

namespace {

__device__ __forceinline__ double qnan_d() { return __builtin_nan(""); }
__device__ __forceinline__ uint64_t okey(int64_t v) { return (uint64_t)v ^ (1ull << 63); }  // signed order as unsigned order
__device__ __forceinline__ int64_t okey_inv(uint64_t k) { return (int64_t)(k ^ (1ull << 63)); }

constexpr uint32_t kSliceReads = 4096;  // Calls one workgroup histograms per pass and locus

struct Sel {  // one radix select: the histograms of its eight passes
    uint32_t hist[8][256];
};
struct SelOut {
    uint64_t key;
    uint32_t k, below, eq;
};

// Rank k (0-based) among the histogrammed elements plus `lump_cnt` copies of `lump_key`: the state after the histograms of passes
// 7 .. down_to + 1 have been applied.  Every thread of the 256-thread workgroup calls it; the result is uniform.
struct ChainLds {
    uint32_t wave_tot[4];
    uint32_t bin, c, h;
};
__device__ SelOut chain(const Sel &S, int down_to, uint32_t k0, uint64_t lump_key, uint32_t lump_cnt, ChainLds &L) {
    SelOut o{0ull, k0, 0u, 0u};
    for (int q = 7; q > down_to; --q) {
        uint32_t h = S.hist[q][threadIdx.x];
        if (lump_cnt && (q == 7 || (lump_key >> (8 * (q + 1))) == o.key) && ((lump_key >> (8 * q)) & 255u) == threadIdx.x) h += lump_cnt;
        const uint32_t inc = wave_inclusive_scan_u32(h);
        __syncthreads();  // L free again
        if ((threadIdx.x & 63u) == 63u) L.wave_tot[threadIdx.x >> 6] = inc;
        if (threadIdx.x == 0) L.bin = 255u, L.c = 0xffffffffu, L.h = 0u;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += L.wave_tot[w];
        const uint32_t cum_incl = base + inc, cum_excl = cum_incl - h;
        if (h && o.k >= cum_excl && o.k < cum_incl) L.bin = threadIdx.x, L.c = cum_excl, L.h = h;  // exactly one thread (or none: rank beyond the elements)
        __syncthreads();
        if (L.c == 0xffffffffu) {  // cannot happen for a rank below the element count; keeps the walk defined
            const uint32_t tot = L.wave_tot[0] + L.wave_tot[1] + L.wave_tot[2] + L.wave_tot[3];
            o.below += tot;
            o.k = 0;
            o.eq = 0;
            o.key = (o.key << 8) | 255u;
        } else {
            o.k -= L.c;
            o.below += L.c;
            o.eq = L.h;
            o.key = (o.key << 8) | L.bin;
        }
    }
    return o;
}

// state of one very deep locus
constexpr uint32_t kMaxSlices = 4096;  // 16.7 million reads per locus
struct DeepLocus {
    uint64_t j, p0;
    uint32_t n, n_slices;
    uint32_t mcount;        // unphased: kept Calls
    uint32_t ng[3], ns[3];  // per haplotype group: Calls, spanning Calls
    uint32_t flags;         // unphased tie: bit 0 a clipped, bit 1 a spanning Call equal to the split value
    unsigned long long lo_max[3];  // per group: largest chosen key below the upper median
    uint32_t slice_eq[kMaxSlices];  // unphased: kept Calls equal to the split value, per slice (file order)
    Sel split, thr[3], hi[3];
};
struct DeepHead {
    uint32_t n_deep;
    uint32_t pad[3];
};

struct DeepArgs {
    KArgs k;
    DeepHead *head;
    DeepLocus *loci;
    uint32_t cap;  // loci the scratch holds
};

// what a group's selects look for (src/call.rs:497-513), from its counts
struct GroupPlan {
    bool live;      // ng >= support
    uint32_t take;  // clipped Calls that join the spanning ones
    uint32_t nc, M;
};
__device__ __forceinline__ GroupPlan plan_of(const DeepLocus &D, uint32_t g, uint32_t support) {
    GroupPlan p;
    const uint32_t ng = D.ng[g], ns = D.ns[g];
    p.live = ng >= support;
    p.take = ns <= support ? support - ns : 0u;
    p.nc = ng - ns;
    p.M = ns + p.take;
    return p;
}

// ---------------------------------------------------------------- kernels
// 1 workgroup: the very deep loci of the big list (in any order: their states are independent)
__global__ __launch_bounds__(256) void deep_collect(DeepArgs a) {
    __shared__ uint32_t cnt[kListShards];
    __shared__ uint32_t nd;
    if (threadIdx.x < kListShards) cnt[threadIdx.x] = a.k.status->list_count[1][threadIdx.x].n;
    if (threadIdx.x == 0) nd = 0u;
    __syncthreads();
    uint32_t total = 0;
    for (int k = 0; k < kListShards; ++k) total += cnt[k];
    for (uint32_t item = threadIdx.x; item < total; item += 256u) {
        uint32_t shard = 0, idx = item;
        while (idx >= cnt[shard]) idx -= cnt[shard++];
        const uint64_t j = a.k.worklist[((uint64_t)kListShards + shard) * a.k.shard_cap + idx];
        const uint64_t p0 = a.k.locus_pair_off[j];
        const uint64_t n = a.k.locus_pair_off[j + 1] - p0;
        if (n <= kGridSelectMin) continue;
        const uint32_t slot = n > (uint64_t)kMaxSlices * kSliceReads ? 0xffffffffu : atomicAdd(&nd, 1u);
        if (slot >= a.cap) {  // (the scratch is sized from n_pairs: cannot happen below 16.7 million reads; such a locus stays NaN and is flagged)
            atomicOr(&a.k.status->err, ST_RANGE);
            a.k.phase1[j] = qnan_d();
            a.k.phase2[j] = qnan_d();
            continue;
        }
        DeepLocus &D = a.loci[slot];
        D.j = j, D.p0 = p0, D.n = (uint32_t)n, D.n_slices = (uint32_t)((n + kSliceReads - 1) / kSliceReads);
    }
    __syncthreads();
    if (threadIdx.x == 0) a.head->n_deep = nd < a.cap ? nd : a.cap;
}

// grid: everything behind the geometry of a locus' state back to zero
__global__ __launch_bounds__(256) void deep_zero(DeepArgs a) {
    const uint32_t nd = a.head->n_deep;
    constexpr size_t kSkip = offsetof(DeepLocus, mcount);
    const size_t words = (sizeof(DeepLocus) - kSkip) / 4;
    for (uint32_t d = 0; d < nd; ++d) {
        uint32_t *w = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(&a.loci[d]) + kSkip);
        for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (size_t)gridDim.x * 256u) w[i] = 0u;
    }
}

// slices of a locus are dealt to the workgroups round-robin
#define FOR_MY_SLICES(D) for (uint32_t sl = blockIdx.x; sl < (D).n_slices; sl += gridDim.x)
#define FOR_SLICE_ELEMS(D, sl, e) \
    for (uint32_t e = (sl) * kSliceReads + threadIdx.x, e##_end = min((D).n, ((sl) + 1u) * kSliceReads); e < e##_end; e += 256u)

// unphased: how many Calls are kept
__global__ __launch_bounds__(256) void deep_count_kept(DeepArgs a) {
    const uint32_t nd = a.head->n_deep;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        uint32_t local = 0;
        FOR_MY_SLICES(D) FOR_SLICE_ELEMS(D, sl, e) local += (a.k.smeta[D.p0 + e] & PM_KEPT) ? 1u : 0u;
        for (int off = 32; off; off >>= 1) local += __shfl_xor(local, off);
        if ((threadIdx.x & 63u) == 0u && local) atomicAdd(&D.mcount, local);
    }
}

struct PassLds {
    uint32_t hist[256];
    ChainLds ch;
};
// adds this workgroup's LDS histogram to the locus' global one (only the bins that were met)
__device__ __forceinline__ void flush_hist(PassLds &L, uint32_t *global_hist) {
    __syncthreads();
    const uint32_t v = L.hist[threadIdx.x];
    if (v) atomicAdd(&global_hist[threadIdx.x], v);
    __syncthreads();
}

// WHICH: 0 = the unphased split (kept Calls), 1 = a group's clip threshold (its clipped Calls), 2 = a group's upper median (chosen Calls)
template <int WHICH>
__global__ __launch_bounds__(256) void deep_select_pass(DeepArgs a, int pass) {
    __shared__ PassLds L;
    const uint32_t nd = a.head->n_deep;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        for (uint32_t g = (WHICH == 0 ? 0u : 1u); g <= (WHICH == 0 ? 0u : 2u); ++g) {
            uint32_t k0 = 0, lump = 0;
            uint64_t t_key = 0;
            bool use_t = false;
            Sel *S;
            if (WHICH == 0) {
                if (D.mcount == 0u) continue;
                const uint32_t ks = D.mcount / 2u;
                k0 = ks < D.mcount ? ks : D.mcount - 1u;
                S = &D.split;
            } else {
                const GroupPlan P = plan_of(D, g, a.k.support);
                if (!P.live) continue;
                if (WHICH == 1) {
                    if (P.take == 0u) continue;
                    k0 = P.nc - P.take;
                    S = &D.thr[g];
                } else {
                    if (P.take > 0u) {  // the threshold select is through: its value, and how many Calls equal to it are taken
                        const SelOut T = chain(D.thr[g], -1, P.nc - P.take, 0ull, 0u, L.ch);
                        t_key = T.key;
                        lump = P.take - (P.nc - T.below - T.eq);
                        use_t = true;
                    }
                    k0 = P.M / 2u;
                    S = &D.hi[g];
                }
            }
            const SelOut st = chain(*S, pass, k0, t_key, lump, L.ch);
            const uint64_t prefix = st.key;
            L.hist[threadIdx.x] = 0u;
            __syncthreads();
            FOR_MY_SLICES(D) FOR_SLICE_ELEMS(D, sl, e) {
                const uint32_t me = a.k.smeta[D.p0 + e];
                if (!(me & PM_KEPT)) continue;
                if (WHICH != 0 && ((me >> PM_GRP_SHIFT) & 3u) != g) continue;
                const uint64_t key = okey(a.k.sval[D.p0 + e]);
                if (WHICH == 1 && !(me & PM_CLIP)) continue;
                if (WHICH == 2 && (me & PM_CLIP) && !(use_t && key > t_key)) continue;
                if (pass != 7 && (key >> (8 * (pass + 1))) != prefix) continue;
                atomicAdd(&L.hist[(key >> (8 * pass)) & 255u], 1u);
            }
            flush_hist(L, S->hist[pass]);
        }
    }
}

// unphased, behind the split select: Calls equal to the split value, per slice
__global__ __launch_bounds__(256) void deep_split_eq(DeepArgs a) {
    __shared__ ChainLds ch;
    const uint32_t nd = a.head->n_deep;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        if (D.mcount == 0u) continue;
        const uint32_t ks = D.mcount / 2u;
        const uint64_t split = chain(D.split, -1, ks < D.mcount ? ks : D.mcount - 1u, 0ull, 0u, ch).key;
        FOR_MY_SLICES(D) {
            uint32_t local = 0;
            FOR_SLICE_ELEMS(D, sl, e) local += ((a.k.smeta[D.p0 + e] & PM_KEPT) && okey(a.k.sval[D.p0 + e]) == split) ? 1u : 0u;
            for (int off = 32; off; off >>= 1) local += __shfl_xor(local, off);
            if ((threadIdx.x & 63u) == 0u && local) atomicAdd(&D.slice_eq[sl], local);
        }
    }
}

// the haplotype groups and their counts.  UNPHASED: src/call.rs:311-313 - Calls below the split value go to h1, above to h2, and of
// those equal to it the first r in file order to h1 (r = what h1 still lacks): a prefix over the slices' counts, then over the
// threads' within the slice.  Phased: the groups are there already (HP), only the counts are taken.
template <bool UNPHASED>
__global__ __launch_bounds__(256) void deep_groups(DeepArgs a) {
    __shared__ ChainLds ch;
    __shared__ uint32_t th_eq[256];
    __shared__ uint32_t cnt[8];
    const uint32_t nd = a.head->n_deep;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        if (threadIdx.x < 8) cnt[threadIdx.x] = 0u;
        uint64_t split = ~0ull;
        uint32_t r = 0;
        if (UNPHASED && D.mcount) {
            const uint32_t ks = D.mcount / 2u;
            const SelOut so = chain(D.split, -1, ks < D.mcount ? ks : D.mcount - 1u, 0ull, 0u, ch);
            split = so.key;
            r = ks - so.below;  // ks >= below: the split value is the ks-th smallest
        }
        __syncthreads();
        uint32_t c_ng[3] = {0, 0, 0}, c_ns[3] = {0, 0, 0}, fl = 0;
        FOR_MY_SLICES(D) {
            if (UNPHASED) {
                // equal Calls in front of this slice, then in front of this thread's part of it (contiguous parts: file order)
                uint32_t before = 0;
                for (uint32_t s2 = threadIdx.x; s2 < sl; s2 += 256u) before += D.slice_eq[s2];
                for (int off = 32; off; off >>= 1) before += __shfl_xor(before, off);
                __syncthreads();
                if ((threadIdx.x & 63u) == 0u) th_eq[threadIdx.x >> 6] = before;
                __syncthreads();
                const uint32_t eq_front = th_eq[0] + th_eq[1] + th_eq[2] + th_eq[3];
                __syncthreads();
                const uint32_t s_lo = sl * kSliceReads, s_hi = min(D.n, s_lo + kSliceReads);
                const uint32_t part = (s_hi - s_lo + 255u) / 256u, e0 = min(s_hi, s_lo + threadIdx.x * part), e1 = min(s_hi, e0 + part);
                uint32_t mine = 0;
                for (uint32_t e = e0; e < e1; ++e) mine += ((a.k.smeta[D.p0 + e] & PM_KEPT) && okey(a.k.sval[D.p0 + e]) == split) ? 1u : 0u;
                th_eq[threadIdx.x] = mine;
                __syncthreads();
                uint32_t eq_before = eq_front;
                for (uint32_t t = 0; t < threadIdx.x; ++t) eq_before += th_eq[t];
                __syncthreads();
                for (uint32_t e = e0; e < e1; ++e) {
                    const uint32_t me = a.k.smeta[D.p0 + e];
                    if (!(me & PM_KEPT)) continue;
                    const uint64_t key = okey(a.k.sval[D.p0 + e]);
                    uint32_t grp = key < split ? 1u : 2u;
                    if (key == split) {
                        grp = eq_before < r ? 1u : 2u;
                        ++eq_before;
                        fl |= (me & PM_CLIP) ? 1u : 2u;
                    }
                    a.k.smeta[D.p0 + e] = (uint8_t)((me & ~(3u << PM_GRP_SHIFT)) | (grp << PM_GRP_SHIFT));
                    c_ng[grp]++;
                    if (!(me & PM_CLIP)) c_ns[grp]++;
                }
            } else {
                FOR_SLICE_ELEMS(D, sl, e) {
                    const uint32_t me = a.k.smeta[D.p0 + e];
                    if (!(me & PM_KEPT)) continue;
                    const uint32_t grp = (me >> PM_GRP_SHIFT) & 3u;
                    if (grp == 1u || grp == 2u) {
                        c_ng[grp]++;
                        if (!(me & PM_CLIP)) c_ns[grp]++;
                    }
                }
            }
        }
        for (int g = 1; g <= 2; ++g) {
            if (c_ng[g]) atomicAdd(&cnt[g], c_ng[g]);
            if (c_ns[g]) atomicAdd(&cnt[4 + g], c_ns[g]);
        }
        if (fl) atomicOr(&cnt[0], fl);
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int g = 1; g <= 2; ++g) {
                if (cnt[g]) atomicAdd(&D.ng[g], cnt[g]);
                if (cnt[4 + g]) atomicAdd(&D.ns[g], cnt[4 + g]);
            }
            if (cnt[0]) atomicOr(&D.flags, cnt[0]);
        }
        __syncthreads();
    }
}

// the lower median when it is not the upper one: the largest chosen key below it
__global__ __launch_bounds__(256) void deep_lower(DeepArgs a) {
    __shared__ ChainLds ch;
    __shared__ unsigned long long best;
    const uint32_t nd = a.head->n_deep;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        for (uint32_t g = 1; g <= 2; ++g) {
            const GroupPlan P = plan_of(D, g, a.k.support);
            if (!P.live || (P.M & 1u)) continue;
            uint64_t t_key = 0;
            uint32_t lump = 0;
            if (P.take > 0u) {
                const SelOut T = chain(D.thr[g], -1, P.nc - P.take, 0ull, 0u, ch);
                t_key = T.key;
                lump = P.take - (P.nc - T.below - T.eq);
            }
            const SelOut H = chain(D.hi[g], -1, P.M / 2u, t_key, lump, ch);
            if (H.below < P.M / 2u) continue;  // rank M / 2 - 1 holds the same value
            if (threadIdx.x == 0) best = 0ull;
            __syncthreads();
            unsigned long long mine = 0ull;
            FOR_MY_SLICES(D) FOR_SLICE_ELEMS(D, sl, e) {
                const uint32_t me = a.k.smeta[D.p0 + e];
                if (!(me & PM_KEPT) || ((me >> PM_GRP_SHIFT) & 3u) != g) continue;
                const uint64_t key = okey(a.k.sval[D.p0 + e]);
                if ((me & PM_CLIP) && !(P.take > 0u && key > t_key)) continue;
                if (key < H.key && key + 1ull > mine) mine = key + 1ull;  // + 1: 0 means "none"
            }
            if (blockIdx.x == 0 && lump && t_key < H.key && t_key + 1ull > mine) mine = t_key + 1ull;  // the taken Calls equal to the threshold
            if (mine) atomicMax(&best, mine);
            __syncthreads();
            if (threadIdx.x == 0 && best) atomicMax(&D.lo_max[g], best);
            __syncthreads();
        }
    }
}

// 1 workgroup per locus: the two rows
template <bool UNPHASED>
__global__ __launch_bounds__(256) void deep_final(DeepArgs a) {
    __shared__ ChainLds ch;
    const uint32_t nd = a.head->n_deep;
    for (uint32_t d = blockIdx.x; d < nd; d += gridDim.x) {
        DeepLocus &D = a.loci[d];
        double out[3] = {qnan_d(), qnan_d(), qnan_d()};
        for (uint32_t g = 1; g <= 2; ++g) {
            const GroupPlan P = plan_of(D, g, a.k.support);
            if (!P.live) continue;  // :498-500
            uint64_t t_key = 0;
            uint32_t lump = 0;
            if (P.take > 0u) {
                const SelOut T = chain(D.thr[g], -1, P.nc - P.take, 0ull, 0u, ch);
                t_key = T.key;
                lump = P.take - (P.nc - T.below - T.eq);
            }
            const SelOut H = chain(D.hi[g], -1, P.M / 2u, t_key, lump, ch);
            const int64_t vhi = okey_inv(H.key);
            if (P.M & 1u) out[g] = (double)vhi;  // :520
            else {
                const int64_t vlo = H.below < P.M / 2u ? vhi : okey_inv(D.lo_max[g] - 1ull);
                out[g] = (double)(vlo + vhi) / 2.0;  // :515-518: i64 add, then f64
            }
        }
        if (threadIdx.x == 0) {
            a.k.phase1[D.j] = out[1];
            a.k.phase2[D.j] = out[2];
        }
        if (UNPHASED && D.mcount) {  // the split cuts through equal values of mixed kind (:312-314 ambiguity)
            const uint32_t ks = D.mcount / 2u;
            const SelOut so = chain(D.split, -1, ks < D.mcount ? ks : D.mcount - 1u, 0ull, 0u, ch);
            const uint32_t r = ks - so.below;
            if (threadIdx.x == 0 && ks >= 1u && ks < D.mcount && r >= 1u && D.flags == 3u) atomicAdd((unsigned long long *)&a.k.status->ties, 1ull);
        }
        __syncthreads();
    }
}

}  // namespace

size_t deep_select_scratch_bytes(uint64_t n_pairs) {
    const uint64_t cap = n_pairs / kGridSelectMin + 1u;
    return sizeof(DeepHead) + (size_t)cap * sizeof(DeepLocus);
}

void launch_deep_select(const KArgs &k, bool unphased, void *scratch, uint64_t n_pairs, hipStream_t s) {
    DeepArgs a;
    a.k = k;
    a.head = reinterpret_cast<DeepHead *>(scratch);
    a.loci = reinterpret_cast<DeepLocus *>(reinterpret_cast<char *>(scratch) + sizeof(DeepHead));
    a.cap = (uint32_t)(n_pairs / kGridSelectMin + 1u);
    constexpr uint32_t G = 512;
    hipLaunchKernelGGL(deep_collect, dim3(1), dim3(256), 0, s, a);
    hipLaunchKernelGGL(deep_zero, dim3(64), dim3(256), 0, s, a);
    if (unphased) {
        hipLaunchKernelGGL(deep_count_kept, dim3(G), dim3(256), 0, s, a);
        for (int pass = 7; pass >= 0; --pass) hipLaunchKernelGGL(deep_select_pass<0>, dim3(G), dim3(256), 0, s, a, pass);
        hipLaunchKernelGGL(deep_split_eq, dim3(G), dim3(256), 0, s, a);
        hipLaunchKernelGGL(deep_groups<true>, dim3(G), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(deep_groups<false>, dim3(G), dim3(256), 0, s, a);
    }
    for (int pass = 7; pass >= 0; --pass) hipLaunchKernelGGL(deep_select_pass<1>, dim3(G), dim3(256), 0, s, a, pass);
    for (int pass = 7; pass >= 0; --pass) hipLaunchKernelGGL(deep_select_pass<2>, dim3(G), dim3(256), 0, s, a, pass);
    hipLaunchKernelGGL(deep_lower, dim3(G), dim3(256), 0, s, a);
    if (unphased) hipLaunchKernelGGL(deep_final<true>, dim3(64), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(deep_final<false>, dim3(64), dim3(256), 0, s, a);
}

}  // namespace inq
