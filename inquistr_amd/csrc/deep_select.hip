// deep_select.hip — locus_call_tail: the LAST kernel of the locus sequence, persistent, one workgroup per compute unit.  It reduces
// every locus of more than 256 offered reads from the per-read Calls locus_call_mid_walk left in the ctx scratch, and empties the
// work lists for the next sequence.  With nothing on the list it is one launch that reads 32 counters and leaves (round 4 had ~36
// launches here, ~0.15 ms of idle launches behind a 0.38 ms kernel whenever the caller gave no depth hint).
//
//   phase A  every workgroup takes listed loci in turn: up to 16 384 reads an LDS sort, up to kGridSelectMin reads (and beyond
//            16.7 million) one workgroup's radix select over the scratch (deep_reduce.h).  No barrier: the walk is a kernel behind.
//   phase B  only when a locus of more than kGridSelectMin reads is listed (amplicon pile-ups): those are reduced by the WHOLE grid.
//            median_str_length (src/call.rs:497-522), the unphased split (:308-322) and the phased bins (:341-369) of such a locus
//            are order statistics of its Calls; a most-significant-byte-first radix select finds one in 8 passes, and in every pass
//            each workgroup histograms its slice of the locus (one slice per workgroup) in LDS and adds the bins it met to the locus' global histogram of that
//            pass.  Between two passes the grid meets at a BARRIER (a device-scope arrival counter; lane 0 of each workgroup
//            releases its stores, arrives, polls, acquires: 5 - 10 us, MI355X_MICROARCH.md "barrier-counter"), then every workgroup
//            derives, alike, which bin the wanted rank fell into (a 256-wide scan per finished pass).  Round 4 ran each pass as a
//            launch of its own: 31 / 21 launches whether or not such a locus existed.  Order statistics per locus:
//     unphased: the split value (rank mcount / 2 of the kept Calls) and, among Calls equal to it, the first r in FILE ORDER
//               (a prefix count over per-slice counts) - then, per haplotype group as in the phased case:
//     the clip threshold (rank nc - take of the group's clipped Calls, only when spanning <= support),
//     the upper median (rank M / 2 of the chosen Calls), and the lower one, which is the upper one again unless exactly M / 2
//     chosen Calls lie below it - then it is their maximum: one more pass, not eight.
//   Keys are rebased to the smallest Call of the locus, so a select takes as many passes as the spread of its Calls has bytes (3 below
//   2^24).  All very deep loci share the passes (their states lie side by side): 14 barriers unphased / 9 phased at 3 passes,
//   29 / 19 at most, however many such loci there are.
//
// The barrier needs every workgroup resident: the grid is one workgroup per compute unit (132 KB of LDS each: no second one fits) and
// never larger than the device's CU count (capi.hip).  Every spin is bounded: a barrier that waits ~2 s raises ST_INTERNAL, sets the
// abort word and every workgroup leaves - the grid always drains.  Which workgroup gets which loci / slices depends on blockIdx and
// gridDim only, both the same in every phase: what a workgroup rewrote (the group bits of smeta) it reads back itself.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cigar_walk.h"
#include "deep_reduce.h"
#include "kernels.h"
#include "wave_primitives.h"

namespace inq {

namespace {

__device__ __forceinline__ double qnan_d() { return __builtin_nan(""); }
__device__ __forceinline__ uint64_t okey(int64_t v) { return (uint64_t)v ^ (1ull << 63); }  // signed order as unsigned order
__device__ __forceinline__ int64_t okey_inv(uint64_t k) { return (int64_t)(k ^ (1ull << 63)); }

// device-scope loads of words other workgroups wrote with atomics (they bypass this CU's L1; the barrier's acquire has dealt with it
// already: belt and braces for the few control words everything else hangs on)
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


struct Sel {  // one radix select: the histograms of its (up to) eight passes
    uint32_t hist[8][256];
};
struct SelOut {
    uint64_t key;
    uint32_t k, below, eq;
};

// Rank k (0-based) among the histogrammed elements plus `lump_cnt` copies of `lump_key`: the state after the histograms of passes
// top .. down_to + 1 have been applied.  Every thread of the 256-thread workgroup calls it; the result is uniform.  Keys are REBASED
// (key - smallest key of the locus' Calls, below): they have no bits above byte `top`, so the select starts there - three passes for
// Calls that spread over less than 2^24, not eight.
struct ChainLds {
    uint32_t wave_tot[4];
    uint32_t bin, c, h;
};
__device__ __forceinline__ SelOut chain(const Sel &S, int top, int down_to, uint32_t k0, uint64_t lump_key, uint32_t lump_cnt, ChainLds &L) {
    SelOut o{0ull, k0, 0u, 0u};
    for (int q = top; q > down_to; --q) {
        uint32_t h = ld_u32(&S.hist[q][threadIdx.x]);
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
constexpr uint32_t kMaxSlices = 1024;  // >= the grid: a locus is cut into at most one slice per workgroup
// What is written once and read by everyone (the geometry) and what the grid accumulates with atomics lie on different 128-byte
// lines, and so does every pass' histogram: a line is either final or in the making, never both.
struct DeepLocus {
    uint64_t j, p0;
    uint32_t n, n_slices, slice;   // slice = reads per slice: ceil(n / grid) rounded up to 256, so every workgroup takes at most one
    alignas(128) uint32_t mcount;  // unphased: kept Calls
    uint32_t ng[3], ns[3];         // per haplotype group: Calls, spanning Calls
    uint32_t flags;                // unphased tie: bit 0 a clipped, bit 1 a spanning Call equal to the split value
    unsigned long long lo_max[3];  // per group: largest chosen key below the upper median (+ 1; 0 = none)
    unsigned long long kmax, kmin_inv;  // largest key / ~smallest key among the Calls the selects look at (atomicMax both)
    alignas(128) uint32_t slice_eq[kMaxSlices];  // unphased: kept Calls equal to the split value, per slice (file order)
    alignas(128) Sel split;
    Sel thr[3], hi[3];
};
static_assert(sizeof(Sel) % 128 == 0 && offsetof(DeepLocus, split) % 128 == 0 && sizeof(DeepLocus) % 128 == 0, "histograms on lines of their own");
struct DeepHead {
    uint32_t pad[32];
};

struct DeepArgs {
    KArgs k;
    DeepHead *head;
    DeepLocus *loci;  // may be null: no locus of the batch can be that deep
    uint32_t cap;     // loci the scratch holds
};

// the rebased key space of a locus: key' = key - kmin, and the most significant byte any key' has a bit in
struct KeySpace {
    uint64_t kmin;
    int top;
};
__device__ __forceinline__ KeySpace key_space(const DeepLocus &D) {
    const uint64_t kmax = ld_u64(&D.kmax), kmin = ~ld_u64(&D.kmin_inv);
    KeySpace ks{kmin, 0};
    if (kmax > kmin) ks.top = (63 - __builtin_clzll(kmax - kmin)) >> 3;
    if (kmax < kmin) ks.kmin = 0;  // no Call at all: nothing will be histogrammed
    return ks;
}

// what a group's selects look for (src/call.rs:497-513), from its counts
struct GroupPlan {
    bool live;      // ng >= support
    uint32_t take;  // clipped Calls that join the spanning ones
    uint32_t nc, M;
};
__device__ __forceinline__ GroupPlan plan_of(const DeepLocus &D, uint32_t g, uint32_t support) {
    GroupPlan p;
    const uint32_t ng = ld_u32(&D.ng[g]), ns = ld_u32(&D.ns[g]);
    p.live = ng >= support;
    p.take = ns <= support ? support - ns : 0u;
    p.nc = ng - ns;
    p.M = ns + p.take;
    return p;
}

// ---------------------------------------------------------------- the phases (device functions of locus_call_tail)
// grid: everything behind the geometry of a locus' state back to zero - with device-scope (write-through) stores: no XCD's L2 is left
// holding a zeroed line that atomics from the other XCDs then change at the memory side
__device__ __forceinline__ void deep_zero(const DeepArgs &a, const uint32_t nd) {
    constexpr size_t kSkip = offsetof(DeepLocus, mcount);
    const size_t words = (sizeof(DeepLocus) - kSkip) / 4;
    for (uint32_t d = 0; d < nd; ++d) {
        uint32_t *w = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(&a.loci[d]) + kSkip);
        for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (size_t)gridDim.x * 256u)
            __hip_atomic_store(&w[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A workgroup's slice of a locus, four elements per thread in flight: f(e, meta, value) for every element (the loads of a round are
// issued before any of them is looked at: a pass is a chain of dependent global loads otherwise).
#define FOR_MY_SLICES(D) for (uint32_t sl = blockIdx.x; sl < (D).n_slices; sl += gridDim.x)
template <class F>
__device__ __forceinline__ void for_slice_elems(const DeepArgs &a, const DeepLocus &D, uint32_t sl, F f) {
    const uint32_t lo = sl * D.slice, hi = min(D.n, lo + D.slice);
    const uint8_t *const meta = a.k.smeta + D.p0;
    const int64_t *const val = a.k.sval + D.p0;
    for (uint32_t e0 = lo + threadIdx.x; e0 < hi; e0 += 1024u) {
        uint32_t me[4];
        int64_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = e0 + 256u * (uint32_t)u;
            const bool in = e < hi;
            me[u] = in ? (uint32_t)meta[e] : 0u;
            v[u] = in ? val[e] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + 256u * (uint32_t)u < hi) f(e0 + 256u * (uint32_t)u, me[u], v[u]);
    }
}
// smallest and largest key of the Calls a workgroup met -> the locus' key space (wave-reduced first)
__device__ __forceinline__ void publish_key_range(DeepLocus &D, uint64_t kmin, uint64_t kmax, bool any) {
    uint64_t inv = any ? ~kmin : 0ull, mx = any ? kmax : 0ull;
    for (int off = 32; off; off >>= 1) {
        const uint64_t oi = __shfl_xor(inv, off), om = __shfl_xor(mx, off);
        inv = oi > inv ? oi : inv;
        mx = om > mx ? om : mx;
    }
    if ((threadIdx.x & 63u) == 0u && inv) {  // (inv == 0 <=> no Call in this wave: ~key is never 0 for a key below 2^64 - 1... and a Call of that key would still be ordered right)
        atomicMax(&D.kmin_inv, (unsigned long long)inv);
        atomicMax(&D.kmax, (unsigned long long)mx);
    }
}

// unphased: how many Calls are kept, and the range of their keys
__device__ __forceinline__ void deep_count_kept(const DeepArgs &a, const uint32_t nd) {
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        uint32_t local = 0;
        uint64_t kmin = ~0ull, kmax = 0ull;
        FOR_MY_SLICES(D) for_slice_elems(a, D, sl, [&](uint32_t, uint32_t me, int64_t v) {
            if (!(me & PM_KEPT)) return;
            ++local;
            const uint64_t key = okey(v);
            kmin = key < kmin ? key : kmin;
            kmax = key > kmax ? key : kmax;
        });
        publish_key_range(D, kmin, kmax, local != 0u);
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
__device__ __forceinline__ void deep_select_pass(const DeepArgs &a, const uint32_t nd, int pass) {
    __shared__ PassLds L;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        const KeySpace ks = key_space(D);
        if (pass > ks.top) continue;  // this locus' keys have no such byte
        for (uint32_t g = (WHICH == 0 ? 0u : 1u); g <= (WHICH == 0 ? 0u : 2u); ++g) {
            uint32_t k0 = 0, lump = 0;
            uint64_t t_key = 0;
            bool use_t = false;
            Sel *S;
            if (WHICH == 0) {
                const uint32_t mcount = ld_u32(&D.mcount);
                if (mcount == 0u) continue;
                const uint32_t kh = mcount / 2u;
                k0 = kh < mcount ? kh : mcount - 1u;
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
                        const SelOut T = chain(D.thr[g], ks.top, -1, P.nc - P.take, 0ull, 0u, L.ch);
                        t_key = T.key;
                        lump = P.take - (P.nc - T.below - T.eq);
                        use_t = true;
                    }
                    k0 = P.M / 2u;
                    S = &D.hi[g];
                }
            }
            const SelOut st = chain(*S, ks.top, pass, k0, t_key, lump, L.ch);
            const uint64_t prefix = st.key;
            L.hist[threadIdx.x] = 0u;
            __syncthreads();
            FOR_MY_SLICES(D) for_slice_elems(a, D, sl, [&](uint32_t, uint32_t me, int64_t v) {
                if (!(me & PM_KEPT)) return;
                if (WHICH != 0 && ((me >> PM_GRP_SHIFT) & 3u) != g) return;
                const uint64_t key = okey(v) - ks.kmin;
                if (WHICH == 1 && !(me & PM_CLIP)) return;
                if (WHICH == 2 && (me & PM_CLIP) && !(use_t && key > t_key)) return;
                if (pass != ks.top && (key >> (8 * (pass + 1))) != prefix) return;
                atomicAdd(&L.hist[(key >> (8 * pass)) & 255u], 1u);
            });
            flush_hist(L, S->hist[pass]);
        }
    }
}

// unphased, behind the split select: Calls equal to the split value, per slice
__device__ __forceinline__ void deep_split_eq(const DeepArgs &a, const uint32_t nd) {
    __shared__ ChainLds ch;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        const uint32_t mcount = ld_u32(&D.mcount);
        if (mcount == 0u) continue;
        const KeySpace ks = key_space(D);
        const uint32_t kh = mcount / 2u;
        const uint64_t split = chain(D.split, ks.top, -1, kh < mcount ? kh : mcount - 1u, 0ull, 0u, ch).key + ks.kmin;
        FOR_MY_SLICES(D) {
            uint32_t local = 0;
            for_slice_elems(a, D, sl, [&](uint32_t, uint32_t me, int64_t v) { local += ((me & PM_KEPT) && okey(v) == split) ? 1u : 0u; });
            for (int off = 32; off; off >>= 1) local += __shfl_xor(local, off);
            if ((threadIdx.x & 63u) == 0u && local) atomicAdd(&D.slice_eq[sl], local);
        }
    }
}

// the haplotype groups and their counts.  UNPHASED: src/call.rs:311-313 - Calls below the split value go to h1, above to h2, and of
// those equal to it the first r in file order to h1 (r = what h1 still lacks): a prefix over the slices' counts, then over the
// threads' within the slice.  Phased: the groups are there already (HP); the counts are taken, and the range of the groups' keys.
template <bool UNPHASED>
__device__ __forceinline__ void deep_groups(const DeepArgs &a, const uint32_t nd) {
    __shared__ ChainLds ch;
    __shared__ uint32_t th_eq[256];
    __shared__ uint32_t cnt[8];
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        if (threadIdx.x < 8) cnt[threadIdx.x] = 0u;
        uint64_t split = ~0ull;
        uint32_t r = 0;
        const uint32_t mcount = UNPHASED ? ld_u32(&D.mcount) : 0u;
        if (UNPHASED && mcount) {
            const KeySpace ks = key_space(D);
            const uint32_t kh = mcount / 2u;
            const SelOut so = chain(D.split, ks.top, -1, kh < mcount ? kh : mcount - 1u, 0ull, 0u, ch);
            split = so.key + ks.kmin;
            r = kh - so.below;  // kh >= below: the split value is the kh-th smallest
        }
        __syncthreads();
        uint32_t c_ng[3] = {0, 0, 0}, c_ns[3] = {0, 0, 0}, fl = 0;
        uint64_t kmin = ~0ull, kmax = 0ull;
        bool any = false;
        FOR_MY_SLICES(D) {
            if (UNPHASED) {
                // equal Calls in front of this slice, then in front of this thread's part of it (contiguous parts: file order)
                uint32_t before = 0;
                for (uint32_t s2 = threadIdx.x; s2 < sl; s2 += 256u) before += ld_u32(&D.slice_eq[s2]);
                for (int off = 32; off; off >>= 1) before += __shfl_xor(before, off);
                __syncthreads();
                if ((threadIdx.x & 63u) == 0u) th_eq[threadIdx.x >> 6] = before;
                __syncthreads();
                const uint32_t eq_front = th_eq[0] + th_eq[1] + th_eq[2] + th_eq[3];
                __syncthreads();
                const uint32_t s_lo = sl * D.slice, s_hi = min(D.n, s_lo + D.slice);
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
                for_slice_elems(a, D, sl, [&](uint32_t, uint32_t me, int64_t v) {
                    if (!(me & PM_KEPT)) return;
                    const uint32_t grp = (me >> PM_GRP_SHIFT) & 3u;
                    if (grp == 1u || grp == 2u) {
                        c_ng[grp]++;
                        if (!(me & PM_CLIP)) c_ns[grp]++;
                        const uint64_t key = okey(v);
                        kmin = key < kmin ? key : kmin;
                        kmax = key > kmax ? key : kmax;
                        any = true;
                    }
                });
            }
        }
        if (!UNPHASED) publish_key_range(D, kmin, kmax, any);  // (unphased: deep_count_kept did, over all kept Calls)
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
__device__ __forceinline__ void deep_lower(const DeepArgs &a, const uint32_t nd) {
    __shared__ ChainLds ch;
    __shared__ unsigned long long best;
    for (uint32_t d = 0; d < nd; ++d) {
        DeepLocus &D = a.loci[d];
        const KeySpace ks = key_space(D);
        for (uint32_t g = 1; g <= 2; ++g) {
            const GroupPlan P = plan_of(D, g, a.k.support);
            if (!P.live || (P.M & 1u)) continue;
            uint64_t t_key = 0;
            uint32_t lump = 0;
            if (P.take > 0u) {
                const SelOut T = chain(D.thr[g], ks.top, -1, P.nc - P.take, 0ull, 0u, ch);
                t_key = T.key;
                lump = P.take - (P.nc - T.below - T.eq);
            }
            const SelOut H = chain(D.hi[g], ks.top, -1, P.M / 2u, t_key, lump, ch);
            if (H.below < P.M / 2u) continue;  // rank M / 2 - 1 holds the same value
            if (threadIdx.x == 0) best = 0ull;
            __syncthreads();
            unsigned long long mine = 0ull;
            FOR_MY_SLICES(D) for_slice_elems(a, D, sl, [&](uint32_t, uint32_t me, int64_t v) {
                if (!(me & PM_KEPT) || ((me >> PM_GRP_SHIFT) & 3u) != g) return;
                const uint64_t key = okey(v) - ks.kmin;
                if ((me & PM_CLIP) && !(P.take > 0u && key > t_key)) return;
                if (key < H.key && key + 1ull > mine) mine = key + 1ull;  // + 1: 0 means "none"
            });
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
__device__ __forceinline__ void deep_final(const DeepArgs &a, const uint32_t nd) {
    __shared__ ChainLds ch;
    for (uint32_t d = blockIdx.x; d < nd; d += gridDim.x) {
        DeepLocus &D = a.loci[d];
        const KeySpace ks = key_space(D);
        double out[3] = {qnan_d(), qnan_d(), qnan_d()};
        for (uint32_t g = 1; g <= 2; ++g) {
            const GroupPlan P = plan_of(D, g, a.k.support);
            if (!P.live) continue;  // :498-500
            uint64_t t_key = 0;
            uint32_t lump = 0;
            if (P.take > 0u) {
                const SelOut T = chain(D.thr[g], ks.top, -1, P.nc - P.take, 0ull, 0u, ch);
                t_key = T.key;
                lump = P.take - (P.nc - T.below - T.eq);
            }
            const SelOut H = chain(D.hi[g], ks.top, -1, P.M / 2u, t_key, lump, ch);
            const int64_t vhi = okey_inv(H.key + ks.kmin);
            if (P.M & 1u) out[g] = (double)vhi;  // :520
            else {
                const int64_t vlo = H.below < P.M / 2u ? vhi : okey_inv(ld_u64(&D.lo_max[g]) - 1ull + ks.kmin);
                out[g] = (double)(vlo + vhi) / 2.0;  // :515-518: i64 add, then f64
            }
        }
        if (threadIdx.x == 0) {
            a.k.phase1[D.j] = out[1];
            a.k.phase2[D.j] = out[2];
        }
        const uint32_t mcount = UNPHASED ? ld_u32(&D.mcount) : 0u;
        if (UNPHASED && mcount) {  // the split cuts through equal values of mixed kind (:312-314 ambiguity)
            const uint32_t kh = mcount / 2u;
            const SelOut so = chain(D.split, ks.top, -1, kh < mcount ? kh : mcount - 1u, 0ull, 0u, ch);
            const uint32_t r = kh - so.below;
            if (threadIdx.x == 0 && kh >= 1u && kh < mcount && r >= 1u && ld_u32(&D.flags) == 3u) atomicAdd((unsigned long long *)&a.k.status->ties, 1ull);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- the grid barrier
// Arrival counter, monotonic inside a launch (locus_call_small zeroes it in front of every sequence): barrier number e is passed
// when the counter reaches e * gridDim.x.  Every wave drains its own memory operations, the workgroup meets, lane 0 releases what the
// workgroup wrote to device scope (buffer_wbl2 sc1), arrives, polls with device-scope loads and a sleep between them, acquires
// (buffer_inv sc1: this CU's L1, and what this XCD's L2 holds of other XCDs' lines), and the workgroup meets again.
// Returns false when the wait was given up (~2 s): ST_INTERNAL is raised and every workgroup leaves at its next barrier.
__device__ __forceinline__ bool grid_barrier(DevStatus *st, uint32_t &epoch) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ uint32_t ok;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the fence's own wait: MI355X_MICROARCH.md, compiler hazard)
        ++epoch;
        const uint32_t target = epoch * gridDim.x;
        atomicAdd(&st->bar_count.v, 1u);
        uint32_t good = 1u;
        for (uint32_t spins = 0;; ++spins) {
            if (__hip_atomic_load(&st->bar_count.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            if (__hip_atomic_load(&st->bar_abort.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                good = 0u;
                break;
            }
            if (spins > (1u << 21)) {  // ~1 us per round: two seconds
                atomicOr(&st->err, ST_INTERNAL);
                __hip_atomic_store(&st->bar_abort.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok = good;
    } else {
        ++epoch;
    }
    __syncthreads();
    return ok != 0u;
}

}  // namespace

// ---------------------------------------------------------------- the kernel
template <bool UNPHASED>
__global__ __launch_bounds__(256) void locus_call_tail(DeepArgs a) {
    __shared__ SortLds<16384> sortL;
    __shared__ SelectLds selL;
    __shared__ uint32_t cnt[kListShards], cnt2[kListShards], cnt_med[kListShards];
    __shared__ uint32_t n_deep_sh;
    DevStatus *const st = a.k.status;
    if (threadIdx.x < kListShards) {
        cnt[threadIdx.x] = st->list_count[1][threadIdx.x].n;
        cnt2[threadIdx.x] = st->list_count[2][threadIdx.x].n;
        cnt_med[threadIdx.x] = st->list_count[0][threadIdx.x].n;
    }
    if (threadIdx.x == 0) n_deep_sh = 0u;
    __syncthreads();
    uint32_t total1 = 0, total2 = 0, total_med = 0;
    for (int k = 0; k < kListShards; ++k) total1 += cnt[k], total2 += cnt2[k], total_med += cnt_med[k];
    const uint32_t total = total1 + total2;
    // the last workgroup out empties the work lists (every other one has read the counters by then) and tidies the barrier words
    auto leave = [&]() {
        __syncthreads();
        if (threadIdx.x == 0 && (total | total_med)) {
            if (atomicAdd(&st->exit_ticket.v, 1u) == gridDim.x - 1u) {
                for (int k = 0; k < kListKinds * kListShards; ++k) st->list_count[k / kListShards][k % kListShards].n = 0u;
                st->bar_count.v = 0u, st->exit_ticket.v = 0u;
            }
        }
    };
    if (total == 0u) return leave();  // nothing deeper than 256 reads: this is all the launch costs

    // item < total1: list 1 (257 .. kWalkSplit reads; up to kReduceInPlace the walk has reduced them already); the rest: list 2
    auto item_locus = [&](uint32_t item, uint64_t &j, uint64_t &p0, uint64_t &n) {
        const bool second = item >= total1;
        const uint32_t *const c = second ? cnt2 : cnt;
        uint32_t shard = 0, idx = second ? item - total1 : item;
        while (idx >= c[shard]) idx -= c[shard++];
        j = a.k.worklist[((uint64_t)(second ? 2 : 1) * kListShards + shard) * a.k.shard_cap + idx];
        p0 = a.k.locus_pair_off[j];
        n = a.k.locus_pair_off[j + 1] - p0;
    };
    auto for_grid = [&](uint64_t n) { return n > kGridSelectMin && n <= 0xffffffffull && a.loci != nullptr && gridDim.x <= kMaxSlices; };

    // ---- the very deep loci of the list: every workgroup counts them the same way (list order), so all agree on how many there are
    // and which state is whose without a word being exchanged; workgroup 0 writes the geometry down for the passes
    {
        uint32_t seen = 0;  // very deep loci in front of this chunk (uniform)
        for (uint32_t base = total1; base < total; base += 256u) {  // (they are all on list 2: kGridSelectMin > kWalkSplit)
            const uint32_t item = base + threadIdx.x;
            uint64_t j = 0, p0 = 0, n = 0;
            bool deep = false;
            if (item < total) {
                item_locus(item, j, p0, n);
                deep = for_grid(n);
            }
            const uint64_t bal = ballot64(deep);
            const uint32_t in_wave = (uint32_t)__popcll(bal & ((1ull << (threadIdx.x & 63u)) - 1ull));
            __syncthreads();
            if ((threadIdx.x & 63u) == 0u) selL.scan[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = seen;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += selL.scan[w];
            const uint32_t slot = before + in_wave;
            if (deep && blockIdx.x == 0) {
                if (slot < a.cap) {
                    DeepLocus &D = a.loci[slot];
                    const uint32_t slice = (uint32_t)(((n + gridDim.x - 1) / gridDim.x + 255u) / 256u * 256u);  // at most one slice per workgroup
                    D.j = j, D.p0 = p0, D.n = (uint32_t)n, D.slice = slice, D.n_slices = (uint32_t)((n + slice - 1) / slice);
                } else {  // (the scratch is sized from n_pairs: cannot happen; such a locus stays NaN and is flagged)
                    atomicOr(&st->err, ST_RANGE);
                    a.k.phase1[j] = qnan_d();
                    a.k.phase2[j] = qnan_d();
                }
            }
            seen += selL.scan[0] + selL.scan[1] + selL.scan[2] + selL.scan[3];
        }
        if (threadIdx.x == 0) n_deep_sh = seen < a.cap ? seen : a.cap;
        __syncthreads();
    }
    const uint32_t nd = n_deep_sh;  // (every workgroup counted the same)

    // ---- phase A: the loci one workgroup reduces
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
        uint64_t j, p0, n;
        item_locus(item, j, p0, n);
        if (for_grid(n)) continue;  // the grid's (phase B)
        if (n <= kReduceInPlace) {  // reduced by the workgroup that walked it - unless a Call did not fit its sort key
            if ((unsigned long long)__double_as_longlong(a.k.phase1[j]) == kDeferredRow) reduce_deep_select<UNPHASED>(a.k, j, p0, (uint32_t)n, selL);
            continue;
        }
        if (n <= 16384u) sort_reduce_locus<UNPHASED, 16384>(a.k, j, p0, (uint32_t)n, sortL, &selL);
        else if (n <= 0xffffffffull) reduce_deep_select<UNPHASED>(a.k, j, p0, (uint32_t)n, selL);
        else if (threadIdx.x == 0) {  // 2^32 reads at one locus: outside what the scratch indexing covers
            atomicOr(&st->err, ST_RANGE);
            a.k.phase1[j] = qnan_d();
            a.k.phase2[j] = qnan_d();
        }
    }
    if (nd == 0u) return leave();

    // ---- phase B: the very deep loci, by the whole grid
    uint32_t epoch = 0;
#define INQ_BARRIER()                  \
    if (!grid_barrier(st, epoch)) {    \
        leave();                       \
        return;                        \
    }
    deep_zero(a, nd);
    INQ_BARRIER();  // geometry (workgroup 0) and zeroes are there
    // the most significant byte any locus' rebased keys reach: the selects start there (Calls that spread over less than 2^24: 3 passes)
    auto top_of_all = [&]() {
        int top = 0;
        for (uint32_t d = 0; d < nd; ++d) top = max(top, key_space(a.loci[d]).top);
        return top;
    };
    int top;
    if (UNPHASED) {
        deep_count_kept(a, nd);
        INQ_BARRIER();
        top = top_of_all();
        for (int pass = top; pass >= 0; --pass) {
            deep_select_pass<0>(a, nd, pass);
            INQ_BARRIER();
        }
        deep_split_eq(a, nd);
        INQ_BARRIER();
        deep_groups<true>(a, nd);
        INQ_BARRIER();
    } else {
        deep_groups<false>(a, nd);
        INQ_BARRIER();
        top = top_of_all();
    }
    for (int pass = top; pass >= 0; --pass) {
        deep_select_pass<1>(a, nd, pass);
        INQ_BARRIER();
    }
    for (int pass = top; pass >= 0; --pass) {
        deep_select_pass<2>(a, nd, pass);
        INQ_BARRIER();
    }
    deep_lower(a, nd);
    INQ_BARRIER();
    deep_final<UNPHASED>(a, nd);
#undef INQ_BARRIER
    leave();
}

size_t deep_select_scratch_bytes(uint64_t n_pairs) {
    const uint64_t cap = n_pairs / kGridSelectMin + 1u;
    return sizeof(DeepHead) + (size_t)cap * sizeof(DeepLocus);
}

void launch_locus_tail(const KArgs &k, bool unphased, void *scratch, uint64_t n_pairs, uint32_t grid, hipStream_t s) {
    DeepArgs a;
    a.k = k;
    a.head = reinterpret_cast<DeepHead *>(scratch);
    a.loci = scratch ? reinterpret_cast<DeepLocus *>(reinterpret_cast<char *>(scratch) + sizeof(DeepHead)) : nullptr;
    a.cap = scratch ? (uint32_t)(n_pairs / kGridSelectMin + 1u) : 0u;
    if (grid < 1u) grid = 1u;
    if (unphased) hipLaunchKernelGGL(locus_call_tail<true>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(locus_call_tail<false>, dim3(grid), dim3(256), 0, s, a);
}

}  // namespace inq
