// bgzf_inflate_wg.hip — DEFLATE (RFC 1951) of BGZF blocks, ONE WORKGROUP PER BLOCK, every lane decoding.
//
// Replaces, for the device front end, the zlib inflate htslib runs under bam.fetch()/rc_records()
// (reference call sites src/call.rs:288,294,338,345; [3P] htslib bgzf.c).  The lane-per-block kernel
// (bgzf_inflate.hip) is bound by the latency of one lane's serial decode (~36 ms per 64 KB block, whatever
// the number of blocks), so a file of 20 000 blocks keeps a quarter of the chip busy for 36 ms.  Here a
// workgroup of T lanes decodes ONE block together:
//
//   * Huffman decode is parallel INSIDE the deflate stream.  A round hands every lane a 32-byte segment of
//     the compressed bits.  Lane 0 knows where its first symbol starts; the others guess (their segment's first
//     bit) and decode anyway: a Huffman decoder started at a wrong bit re-synchronises with the true symbol
//     chain after a few symbols.  Every lane reports where its chain left its segment; a lane whose left
//     neighbour ended somewhere else than it started decodes again from there, until every start equals the
//     neighbour's end (lane 0's is true by construction, so after k repeats lanes 0..k are: correctness never
//     depends on the guess, only speed does; typically 2-3 repeats).  These passes only COUNT (bits, output
//     bytes, matches).
//   * A prefix sum of the byte counts gives every segment its place in the output.  The round is then COMMITTED in
//     stretches of segments whose bytes fit the root array, two jobs per segment dealt over all lanes (the counting pass
//     notes where a chain enters its segment's second half; any lane can decode any half: the bits are staged, starts
//     and places are shared).  A job decodes once more and records, for every output byte, its ROOT in LDS: a literal's
//     value, or the position a match byte is ultimately copied from.  Inside the job's own bytes the root is known at
//     once (root[p] = root[p - distance], the job walks its bytes in order); a source in front of them is left as a
//     pointer.  Pointer jumping (root[p] = root[root[p]], all bytes at once) then shortens every chain to a literal or
//     to bytes of earlier stretches in a few sweeps - the chains CIGAR-like data produces (3-byte matches at distance 4
//     or 8, each feeding the next) would take thousands of ordered copy sweeps otherwise - and ONE coalesced pass stores
//     the stretch as whole dwords.
//   * Symbols are decoded by table: 10 bits index a 1024-entry table in LDS (literal / length base + extra-bit count,
//     and the code length), 8 bits a distance table; end of block, "not a code" and codes longer than the index share
//     one flag bit, so the loop's fast path tests once; longer codes (rare by construction) take the canonical
//     limits-and-base path.  The bit cursor refills without a branch.  Tables are built by the whole workgroup (count,
//     rank by ballot, scatter, one canonical decode per table entry); so are the code lengths of a dynamic block's
//     header, a Huffman stream of their own, decoded 32 bits per lane with the same guess-and-confirm scheme.
//   * The output window is the output itself in global memory (L2-resident while the block is in flight), so a
//     workgroup needs 20 KB of LDS and eight of them share a CU.
// Every access is bounded (LDS indices masked or checked, output by ISIZE, distances by the bytes produced,
// every loop by a bit count that strictly grows), so a corrupt stream ends in a per-block status, never in a
// fault or a hang.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/inquistr_hip.h"
#include "front_kernels.h"
#include "wave_primitives.h"

namespace inq {

namespace {

#ifndef INQ_WG_T
#define INQ_WG_T 128
#endif
#ifndef INQ_WG_LITBITS
#define INQ_WG_LITBITS 10
#endif
#ifndef INQ_WG_DISTBITS
#define INQ_WG_DISTBITS 8
#endif
constexpr int kLitBits = INQ_WG_LITBITS, kDistBits = INQ_WG_DISTBITS;
#ifndef INQ_WG_LITFIRST
#define INQ_WG_LITFIRST 1
#endif
#ifndef INQ_WG_STRETCH_SEGS
#define INQ_WG_STRETCH_SEGS (INQ_WG_T / 2)
#endif
#ifndef INQ_WG_SEGBITS
#define INQ_WG_SEGBITS 256
#endif
constexpr uint32_t kSegBits = INQ_WG_SEGBITS;  // compressed bits per lane and round (a multiple of 32, > the longest symbol's 48)
constexpr int kMaxLit = 288, kMaxDist = 32;

constexpr uint32_t E_LIT = 0u, E_LEN = 1u, E_EOB = 2u, E_LONG = 3u;
// table entry: bits 0-3 code length (0 = not a code), 4-5 type, 6 = everything the decode loop leaves its fast path for
// (end of block, not a code, a code longer than the table's index), and
//   32-bit form (INQ_WG_LUT16 = 0): 8-11 extra bits, 16-31 literal / base value
//   16-bit form (INQ_WG_LUT16 = 1): 7-15 a literal's byte, or a length / distance symbol's INDEX - extra bits and base are
//     arithmetic in the index (RFC 1951 3.2.5), a handful of operations on the match path in exchange for 2.5 KB of LDS
//     (17.9 instead of 20.4 KB per workgroup: nine BGZF blocks in flight per CU instead of eight)
#ifndef INQ_WG_LUT16
#define INQ_WG_LUT16 0
#endif
constexpr uint32_t kSpecial = 0x40u;
constexpr uint32_t kLongEntry = (E_LONG << 4) | kSpecial;  // code longer than the table's index: canonical path
constexpr uint32_t kNoCode = kSpecial;
#if INQ_WG_LUT16
typedef uint16_t lut_t;
constexpr uint32_t kValShift = 7u;
__device__ __forceinline__ uint32_t mk_entry(uint32_t n, uint32_t type, uint32_t, uint32_t val) { return n | (type << 4) | (val << kValShift); }
__device__ __forceinline__ uint32_t len_xbits(uint32_t e) {
    const uint32_t s = e >> kValShift;
    return (s < 8u || s == 28u) ? 0u : (s - 4u) >> 2;
}
__device__ __forceinline__ uint32_t len_base(uint32_t e, uint32_t xb) {
    const uint32_t s = e >> kValShift;
    return s < 8u ? 3u + s : s == 28u ? 258u : 3u + ((4u + (s & 3u)) << xb);
}
__device__ __forceinline__ uint32_t dist_xbits(uint32_t d) {
    const uint32_t s = d >> kValShift;
    return s < 4u ? 0u : (s - 2u) >> 1;
}
__device__ __forceinline__ uint32_t dist_base(uint32_t d, uint32_t xb) {
    const uint32_t s = d >> kValShift;
    return s < 4u ? 1u + s : 1u + ((2u + (s & 1u)) << xb);
}
#else
typedef uint32_t lut_t;
constexpr uint32_t kValShift = 16u;
__device__ __forceinline__ uint32_t mk_entry(uint32_t n, uint32_t type, uint32_t xb, uint32_t val) {
    return n | (type << 4) | (xb << 8) | (val << kValShift);
}
__device__ __forceinline__ uint32_t len_xbits(uint32_t e) { return (e >> 8) & 15u; }
__device__ __forceinline__ uint32_t len_base(uint32_t e, uint32_t) { return e >> kValShift; }
__device__ __forceinline__ uint32_t dist_xbits(uint32_t d) { return (d >> 8) & 15u; }
__device__ __forceinline__ uint32_t dist_base(uint32_t d, uint32_t) { return d >> kValShift; }
#endif
// RFC 1951 3.2.5: literal/length symbol -> entry
__device__ __forceinline__ uint32_t ll_entry(uint32_t sym, uint32_t n) {
    if (sym < 256u) return mk_entry(n, E_LIT, 0u, sym);
    if (sym == 256u) return mk_entry(n, E_EOB, 0u, 0u) | kSpecial;
    const uint32_t s = sym - 257u;
    if (s >= 29u) return kNoCode;  // 286, 287 take part in the fixed code but never appear in valid data
#if INQ_WG_LUT16
    return mk_entry(n, E_LEN, 0u, s);
#else
    if (s < 8u) return mk_entry(n, E_LEN, 0u, 3u + s);
    if (s == 28u) return mk_entry(n, E_LEN, 0u, 258u);
    const uint32_t xb = (s - 4u) >> 2;
    return mk_entry(n, E_LEN, xb, 3u + ((4u + (s & 3u)) << xb));
#endif
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t n) {
    if (sym >= 30u) return kNoCode;
#if INQ_WG_LUT16
    return mk_entry(n, E_LEN, 0u, sym);
#else
    if (sym < 4u) return mk_entry(n, E_LEN, 0u, 1u + sym);
    const uint32_t xb = (sym - 2u) >> 1;
    return mk_entry(n, E_LEN, xb, 1u + ((2u + (sym & 1u)) << xb));
#endif
}

template <int T>
struct WgLds {
    // dwords: T segments + look-ahead; never less than a dynamic block's header (<= ~600 bytes), which is parsed from it too
    static constexpr int kStage = (T * (int)kSegBits / 32 + 8) > 264 ? (T * (int)kSegBits / 32 + 8) : 264;
#ifndef INQ_WG_CAP
#define INQ_WG_CAP 4096
#endif
    static constexpr int kRoundCap = INQ_WG_CAP;    // output bytes per round (rounds are cut at the lane that would exceed it)
    static_assert(T % 64 == 0 && T >= 64 && T <= 512, "whole waves");
    static_assert((kRoundCap & (kRoundCap - 1)) == 0 && kRoundCap >= 1024 && kRoundCap <= 16384, "root indices are masked with kRoundCap - 1; 32768 + kRoundCap stays below the literal range");
    static_assert(kSegBits % 64 == 0 && kSegBits >= 128, "segments are whole dwords, and half a segment is longer than the longest symbol (48 bits)");
    static_assert(T * kSegBits + 64 <= 65536, "bit positions of a round fit 16 bits (start_sh, mid_sh)");
    lut_t lut_ll[1 << kLitBits];
    lut_t lut_d[1 << kDistBits];
    uint32_t stage[kStage];
    // root of every output byte of the stretch being committed: a literal's value, or where a match byte is copied from
    // as position - (stretch start - 32768)
    // (the arrays that only live while no stretch is being committed share its storage: header parse, table build, counting)
    union {
        uint16_t root[kRoundCap];
        struct {
            uint32_t end_bit[T];   // per lane: where its chain left its segment, | kFlagBit if it stopped (EOB / not a code)
            uint32_t cnt[2][16], run[2][16];
            uint8_t lens[kMaxLit + kMaxDist];
            uint8_t cl_lut[128];  // 7 bits of the stream -> code-length symbol | code length << 5 (0 = not a code)
            uint8_t cl_len[20];
            uint8_t last_sh[T];  // header: the last code length a lane's symbols leave behind (kNoLast = all of them copy their predecessor)
        };
    };
    uint32_t off_sh[T + 1];  // per lane: first output byte of its chain, relative to the round's; [T] = the round's bytes
    // what a commit job needs of a segment besides off_sh: where its chain starts, and its first symbol start in the
    // segment's second half with the bytes produced before it (position | bytes << 16, kNoMid = the chain ended before)
    uint16_t start_sh[T];
    uint32_t mid_sh[T];
    uint16_t ntok_sh[T], midtok_sh[T];  // tokens the segment's chain left behind (kTokOverflow: none usable), of which in front of mid
    uint16_t sorted[kMaxLit + kMaxDist];  // symbols by (code length, value): literal/length, then distance
    uint32_t limit[2][16], base[2][16];
    // block-uniform state
    uint32_t P, out, status, last, type, eob, hlit, hdist, flag;
    uint32_t red[T / 64], red2[T / 64];
#ifdef INQ_WG_PAD  // experiments only (tools/inflate_occupancy.sh): bytes of LDS nobody uses, so that fewer workgroups share a CU
    uint32_t pad[INQ_WG_PAD / 4];
#endif
};

// root values: < 32768 a byte of an earlier stretch (deflate distances are <= 32768), 32768 .. 32768 + kRoundCap a byte of
// this stretch, >= kRootLit a literal (low byte)
constexpr uint32_t kRootLit = 0xff00u;
// tokens (the symbols of a segment as its last counting pass decoded them, kept in global memory for the commit): a literal is
// kTokLit | byte, a match length << 16 | distance - 1.  Segment s of block b keeps its i-th token at
// tokens[((b * kTokCap + i) * T + s]: the lanes of a wave, all at the same i, store and load 256 contiguous bytes.
#ifndef INQ_WG_TOKCAP
#define INQ_WG_TOKCAP 64
#endif
constexpr uint32_t kTokCap = INQ_WG_TOKCAP;  // a segment with more symbols (256 bits of codes under 4 bits) is decoded again by the commit
constexpr uint32_t kTokLit = 0x80000000u;
constexpr uint32_t kTokOverflow = 0xffffu;
constexpr uint32_t kNoMid = 0xffffffffu;
constexpr uint32_t kStopped = 0x80000000u;  // in end_bit: the chain met EOB or a pattern that is no code
constexpr uint32_t kStopEob = 0x40000000u;  // ... and it was EOB

__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    return w;
}
__device__ __forceinline__ void st_u32(uint8_t *p, uint32_t w) { __builtin_memcpy(p, &w, 4); }

// ---- LZ77 copy in global memory (the window is the output): <= 16 bytes per load/store group, loads first
struct Quad {
    uint32_t w0, w1, w2, w3;
};
__device__ __forceinline__ Quad load_quad(const uint8_t *src) { return Quad{ld_u32(src), ld_u32(src + 4), ld_u32(src + 8), ld_u32(src + 12)}; }
__device__ __forceinline__ void store_quad(uint8_t *dst, const Quad &q, uint32_t n) {
    if (n == 16u) {
        st_u32(dst, q.w0);
        st_u32(dst + 4, q.w1);
        st_u32(dst + 8, q.w2);
        st_u32(dst + 12, q.w3);
        return;
    }
    uint32_t k = 0;
    if (n >= 4u) st_u32(dst, q.w0), k = 4u;
    if (n >= 8u) st_u32(dst + 4, q.w1), k = 8u;
    if (n >= 12u) st_u32(dst + 8, q.w2), k = 12u;
    uint32_t t = k == 0u ? q.w0 : k == 4u ? q.w1 : k == 8u ? q.w2 : q.w3;
    for (; k < n; ++k, t >>= 8) dst[k] = (uint8_t)t;
}
__device__ __forceinline__ void copy_match(uint8_t *dst, uint32_t dd, uint32_t len) {
    const uint8_t *src = dst - dd;
    if (dd >= 16u || dd >= len) {  // a 16-byte group never reads what it writes
        for (uint32_t k = 0; k < len; k += 16u) {
            const Quad q = load_quad(src + k);
            store_quad(dst + k, q, len - k < 16u ? len - k : 16u);
        }
        return;
    }
    // short period (dd < 16, dd < len): the output is the last dd bytes repeated
    const uint64_t lo = (uint64_t)ld_u32(src) | ((uint64_t)ld_u32(src + 4) << 32);
    const uint64_t hi = (uint64_t)ld_u32(src + 8) | ((uint64_t)ld_u32(src + 12) << 32);
    if (dd == 1u) {
        const uint32_t w = ((uint32_t)lo & 0xffu) * 0x01010101u;
        uint32_t k = 0;
        for (; k + 4u <= len; k += 4u) st_u32(dst + k, w);
        for (; k < len; ++k) dst[k] = (uint8_t)w;
        return;
    }
    uint32_t idx = 0;
    for (uint32_t k = 0; k < len; ++k) {
        dst[k] = (uint8_t)(idx < 8u ? lo >> (8u * idx) : hi >> (8u * (idx - 8u)));
        if (++idx == dd) idx = 0;
    }
}

// ---- bit cursor over the staged compressed dwords (LDS); positions are relative to stage bit 0
struct SegBits {
    const uint32_t *st;
    uint32_t lo, hi, nxt, wi, pos;
    __device__ __forceinline__ void init(const uint32_t *stage, uint32_t p) {
        st = stage;
        pos = p;
        wi = p >> 5;
        lo = st[wi];
        hi = st[wi + 1];
        nxt = st[wi + 2];
    }
    __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, pos & 31u); }  // >= 33 valid bits behind pos
    __device__ __forceinline__ void consume(uint32_t n) {  // n < 32
        // no branch: the dword two behind the cursor's is fetched every time (the same one again while the cursor stays
        // inside a dword) and nothing waits for it before the cursor has crossed into the next dword
        const uint32_t np = pos + n;
        const bool cross = ((np ^ pos) & ~31u) != 0u;
        lo = cross ? hi : lo;
        hi = cross ? nxt : hi;
        nxt = st[(np >> 5) + 2u];
        pos = np;
    }
};

// canonical decode of the code at the cursor (first stream bit = top of v15): {length, sorted index} or 0 length
template <int T>
__device__ __forceinline__ uint32_t canon_entry(const WgLds<T> &L, int tbl, uint32_t v15, uint32_t from_len) {
    uint32_t len = from_len;
    for (; len <= 15u; ++len)
        if (v15 < L.limit[tbl][len]) break;
    if (len > 15u) return kNoCode;
    const uint32_t idx = L.base[tbl][len] + (v15 >> (15u - len));
    if (tbl == 0) return idx < (uint32_t)kMaxLit ? ll_entry(L.sorted[idx], len) : kNoCode;
    return idx < (uint32_t)kMaxDist ? dist_entry(L.sorted[kMaxLit + idx], len) : kNoCode;
}

// Roots of the `len` bytes of a match at distance `dist`, written by the lane that owns bytes [own, lane_end) of the stretch
// (all positions relative to the stretch's first byte); p = position of the match's first byte.
template <int T>
__device__ __forceinline__ void root_match(WgLds<T> &L, uint32_t p, uint32_t own, uint32_t lane_end, uint32_t len, uint32_t dist) {
    // the lane's own bytes are rooted already (it walks them in order; LDS operations of a wave execute in
    // order); a source in front of the lane's stretch stays a pointer for the jumping sweeps.  Roots go four
    // at a time (8 bytes); a group may write beyond the match's end as long as it stays inside the lane's own
    // bytes: the symbols behind it write those entries again, later.
    const int32_t sp0 = (int32_t)(p - dist);  // may lie in front of the stretch (negative)
    auto put4 = [&](uint32_t at, uint64_t w, uint32_t cnt) {
        if (at + 4u <= lane_end) __builtin_memcpy(&L.root[at], &w, 8);
        else
            for (uint32_t j = 0; j < cnt && j < 4u; ++j, w >>= 16) L.root[at + j] = (uint16_t)w;
    };
    if (sp0 + (int32_t)len <= (int32_t)own) {
        // the whole source lies in front of this lane's stretch: the roots are consecutive pointers
        const uint32_t v = (uint32_t)(sp0 + 32768);
        for (uint32_t k = 0; k < len; k += 4u) {
            const uint32_t a = (v + k) | ((v + k + 1u) << 16), b2 = (v + k + 2u) | ((v + k + 3u) << 16);
            put4(p + k, (uint64_t)a | ((uint64_t)b2 << 32), len - k);
        }
    } else if (sp0 >= (int32_t)own) {
        // the whole source lies in the lane's own stretch, already rooted: copy roots (a group never reads what
        // it writes; later groups may read what earlier ones wrote: LDS is in order).  A period below four
        // (runs) gives the first group from the period's entries and the rest from a multiple of it.
        uint32_t back = dist, k = 0;
        if (dist < 4u) {
            uint64_t w;
            __builtin_memcpy(&w, &L.root[(uint32_t)sp0], 8);  // entries sp0 .. sp0 + 3 <= p + 2: inside the match (len >= 3)
            const uint64_t e0 = w & 0xffffu, e1 = (w >> 16) & 0xffffu, e2 = (w >> 32) & 0xffffu;
            const uint64_t g = dist == 1u ? e0 * 0x0001000100010001ull
                               : dist == 2u ? (e0 | e1 << 16) * 0x0000000100000001ull
                                            : (e0 | e1 << 16 | e2 << 32 | e0 << 48);
            put4(p, g, len);
            back = dist == 3u ? 6u : 4u;
            k = 4u;
        }
        for (; k < len; k += 4u) {
            uint64_t w;
            __builtin_memcpy(&w, &L.root[p + k - back], 8);
            put4(p + k, w, len - k);
        }
    } else {  // the source straddles the start of the lane's stretch
        for (uint32_t k = 0; k < len; ++k) {
            const int32_t sp = sp0 + (int32_t)k;
            L.root[p + k] = sp >= (int32_t)own ? L.root[(uint32_t)sp] : (uint16_t)(sp + 32768);
        }
    }
}

// Decodes the symbols that START in [start, seg_end) of one lane's segment.  MODE 0: counts only.  MODE 1 (commit):
// literals go to out[o...] and every byte gets its root (r0 = first output byte of the round).  MODE 2 (a lone lane
// whose output exceeds the round's root array): literals and matches go straight to the output, in order.
// Returns the position behind the last decoded symbol (| kStopped / kStopEob).  `bad` collects INQ_INFLATE_* bits.
// MODE 0 with tok != null also leaves the symbols behind as tokens (tok = this segment's column of the block's token scratch,
// stride T words): *ntok = how many the chain produced (only the first kTokCap are stored), *mid_tok = how many before `mid`.
template <int T, int MODE, int FORM>
__device__ __forceinline__ uint32_t decode_segment(WgLds<T> &L, uint32_t start, uint32_t seg_end, uint32_t &nbytes, uint8_t *out, uint32_t o,
                                                   uint32_t r0, uint32_t &bad, uint32_t *mid = nullptr, uint32_t *tok = nullptr,
                                                   uint32_t *ntok = nullptr, uint32_t *mid_tok = nullptr) {
    SegBits b;
    b.init(L.stage, start);
    uint32_t nb = 0, stop = 0, nt = 0;
    const uint32_t lane_end = MODE == 1 ? o - r0 + nbytes : 0u;  // MODE 1: nbytes comes in as the job's counted bytes
    // MODE 0 walks the segment's two halves one after the other and notes where the chain enters the second
    if (MODE == 0) *mid = kNoMid;
    for (int half = MODE == 0 ? 0 : 1; half < 2; ++half) {
    const uint32_t lim = half == 0 ? seg_end - kSegBits / 2u : seg_end;
    if (MODE == 0 && half == 1) {
        if (stop) break;
        *mid = b.pos | (nb << 16);
        if (tok) *mid_tok = nt;
    }
    while (b.pos < lim) {
        const uint32_t bits = b.peek();
        uint32_t e = L.lut_ll[bits & ((1u << kLitBits) - 1u)];
#if INQ_WG_LITFIRST
        // a plain literal - the table's most frequent answer in sequence / quality bytes - is recognised by ONE test; everything
        // else (a length, or one of the three rare cases behind kSpecial) takes the second
        bool lit = !(e & (kSpecial | (E_LEN << 4)));
        if (!lit && (e & kSpecial)) {
#else
        bool lit;
        if (e & kSpecial) {  // one test keeps the three rare cases out of the loop's fast path
#endif
            if (e == kLongEntry) e = canon_entry<T>(L, 0, __brev(bits) >> 17, kLitBits + 1);
            if (e & kSpecial) {
                if (e & 15u) {  // end of block (a code has a length; "not a code" has none)
                    b.consume(e & 15u);
                    stop = kStopped | kStopEob;
                } else stop = kStopped;
                break;
            }
#if INQ_WG_LITFIRST
            lit = !(e & (E_LEN << 4));
#endif
        }
#if !INQ_WG_LITFIRST
        lit = !(e & (E_LEN << 4));
#endif
#if INQ_WG_LUT16
        const uint32_t n = e & 15u;  // (extra bits: computed on the match path only, below)
#else
        const uint32_t n = e & 15u, xb = len_xbits(e);  // a literal has no extra bits
#endif
        constexpr bool PAIR = FORM == 1;
        if constexpr (PAIR) {
            if (lit) {
                // a literal: look at the symbol behind it in the bits already peeked (>= 32 - 15 of them are left) - when that is a
                // plain literal too (the common case in base-quality and sequence bytes, what a BAM mostly is) and starts in front of
                // the limit, both go out with one move of the cursor and one turn of the loop.  Measured (20 000 blocks, zlib level
                // 1 / 6): quality-like bytes 17.4 -> 13.5 / 16.1 -> 12.2 ms, nanopore-like 14.0 -> 12.1 / 13.8 -> 11.3, packed bases
                // 11.9 -> 10.9 / 11.2 -> 10.8, CIGAR-only blocks 11.4 -> 11.95 / 8.1 -> 8.15 (the look is wasted when a match
                // follows).  A third literal from the same peek, a "only behind a literal" predictor and a per-block switch (pairs
                // only where literals hold most of the block's code space) all lost: any condition on the second look costs more
                // than the look (profiles/r03_results/inflate_literal_runs_five_builds.txt, inflate_pairs_per_block_switch.txt).
                const uint32_t e2 = L.lut_ll[(bits >> n) & ((1u << kLitBits) - 1u)];
                const bool two = !(e2 & (kSpecial | (E_LEN << 4))) && b.pos + n < lim;
                b.consume(n + (two ? (e2 & 15u) : 0u));
                if (MODE == 2) {
                    out[o + nb] = (uint8_t)(e >> kValShift);
                    if (two) out[o + nb + 1u] = (uint8_t)(e2 >> kValShift);
                }
                if (MODE == 1) {  // the bytes themselves: stored by the gather, coalesced
                    L.root[o + nb - r0] = (uint16_t)(kRootLit | (e >> kValShift));
                    if (two) L.root[o + nb + 1u - r0] = (uint16_t)(kRootLit | (e2 >> kValShift));
                }
                if (MODE == 0 && tok) {
                    if (nt < kTokCap) tok[nt * T] = kTokLit | (e >> kValShift);
                    ++nt;
                    if (two) {
                        if (nt < kTokCap) tok[nt * T] = kTokLit | (e2 >> kValShift);
                        ++nt;
                    }
                }
                nb += two ? 2u : 1u;
                continue;
            }
        } else {
#if INQ_WG_LUT16
            if (lit) b.consume(n);
#else
            b.consume(n + xb);
#endif
            if (lit) {
                if (MODE == 2) out[o + nb] = (uint8_t)(e >> kValShift);
                if (MODE == 1) L.root[o + nb - r0] = (uint16_t)(kRootLit | (e >> kValShift));  // the byte itself: stored by the gather, coalesced
                if (MODE == 0 && tok) {
                    if (nt < kTokCap) tok[nt * T] = kTokLit | (e >> kValShift);
                    ++nt;
                }
                ++nb;
                continue;
            }
        }
#if INQ_WG_LUT16
        const uint32_t xb = len_xbits(e);
        b.consume(n + xb);
#else
        if constexpr (PAIR) b.consume(n + xb);
#endif
        const uint32_t len = len_base(e, xb) + __builtin_amdgcn_ubfe(bits, n, xb);
        const uint32_t dbits = b.peek();
        uint32_t d = L.lut_d[dbits & ((1u << kDistBits) - 1u)];
        if (d & kSpecial) {
            if (d == kLongEntry) d = canon_entry<T>(L, 1, __brev(dbits) >> 17, kDistBits + 1);
            if (d & kSpecial) {
                stop = kStopped;
                break;
            }
        }
        const uint32_t dn = d & 15u;
        const uint32_t dxb = dist_xbits(d);
        const uint32_t dist = dist_base(d, dxb) + __builtin_amdgcn_ubfe(dbits, dn, dxb);
        b.consume(dn + dxb);
        if (MODE == 0 && tok) {
            if (nt < kTokCap) tok[nt * T] = (len << 16) | (dist - 1u);
            ++nt;
        }
        if (MODE) {
            if (dist > o + nb) {
                bad |= INQ_INFLATE_BAD_DISTANCE;
                stop = kStopped;
                break;
            }
            if (MODE == 1) {
                root_match<T>(L, o + nb - r0, o - r0, lane_end, len, dist);
            } else {
                copy_match(out + o + nb, dist, len);
            }
        }
        nb += len;
    }
    }
    nbytes = nb;
    if (MODE == 0 && tok) *ntok = nt;
    return b.pos | stop;
}

// The commit of a job from its tokens: no Huffman decode, no bit cursor.  tok = the segment's token column, tokens [i0, i1);
// the job's bytes start at output position o (the round's first byte at r0 is the stretch's origin) and are `nbytes` long.
template <int T>
__device__ __forceinline__ void commit_tokens(WgLds<T> &L, const uint32_t *tok, uint32_t i0, uint32_t i1, uint32_t o, uint32_t r0, uint32_t nbytes,
                                              uint32_t &bad) {
    const uint32_t own = o - r0, lane_end = own + nbytes;
    uint32_t p = own;
    // four tokens in flight ahead of the one being written out (the loads are independent of everything the loop computes)
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    auto fetch = [&](uint32_t i) -> uint32_t { return tok[(i < i1 ? i : i1 - 1u) * T]; };
    if (i0 < i1) q0 = fetch(i0), q1 = fetch(i0 + 1u), q2 = fetch(i0 + 2u), q3 = fetch(i0 + 3u);
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t t = q0;
        q0 = q1, q1 = q2, q2 = q3, q3 = fetch(i + 4u);
        if (t & kTokLit) {
            L.root[p] = (uint16_t)(kRootLit | (t & 0xffu));
            ++p;
            continue;
        }
        const uint32_t len = t >> 16, dist = (t & 0xffffu) + 1u;
        if (dist > o - own + p || p + len > lane_end) {  // (o - own = r0: the bytes in front of the stretch; a token is what this kernel wrote, the second test only keeps LDS writes inside the job)
            bad |= INQ_INFLATE_BAD_DISTANCE;
            break;
        }
        root_match<T>(L, p, own, lane_end, len, dist);
        p += len;
    }
}

// ---- workgroup-wide helpers (T lanes, T / 64 waves); every lane must call them
// first lane of the workgroup for which p holds (0xffff: none), for two predicates at once: a ballot and a find-first-bit per
// wave, one exchange through LDS for both
template <int T>
__device__ __forceinline__ void wg_first2(bool pa, bool pb, uint32_t &fa, uint32_t &fb, uint32_t *red, uint32_t *red2) {
    const uint64_t ma = ballot64(pa), mb = ballot64(pb);
    const uint32_t w0 = threadIdx.x & ~63u;
    const uint32_t a = ma ? w0 + (uint32_t)__ffsll((unsigned long long)ma) - 1u : 0xffffu;
    const uint32_t b = mb ? w0 + (uint32_t)__ffsll((unsigned long long)mb) - 1u : 0xffffu;
    if (T == 64) {
        fa = a, fb = b;
        return;
    }
    __syncthreads();  // red[] free again
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = a, red2[threadIdx.x >> 6] = b;
    __syncthreads();
    uint32_t ra = red[0], rb = red2[0];
    for (int w = 1; w < T / 64; ++w) {
        ra = red[w] < ra ? red[w] : ra;
        rb = red2[w] < rb ? red2[w] : rb;
    }
    fa = ra, fb = rb;
}
template <int T>
__device__ __forceinline__ uint32_t wg_first(bool p, uint32_t *red, uint32_t *red2) {
    uint32_t fa, fb;
    wg_first2<T>(p, false, fa, fb, red, red2);
    return fa;
}
// exclusive prefix sums of two values at once; totals through tot_a / tot_b
template <int T>
__device__ __forceinline__ void wg_scan2(uint32_t a, uint32_t b, uint32_t &ex_a, uint32_t &ex_b, uint32_t &tot_a, uint32_t &tot_b,
                                         uint32_t *red, uint32_t *red2) {
    const uint32_t ia = wave_inclusive_scan_u32(a), ib = wave_inclusive_scan_u32(b);
    __syncthreads();
    if ((threadIdx.x & 63u) == 63u) red[threadIdx.x >> 6] = ia, red2[threadIdx.x >> 6] = ib;
    __syncthreads();
    uint32_t ca = 0, cb = 0, ta = 0, tb = 0;
    for (int w = 0; w < T / 64; ++w) {
        if (w < (int)(threadIdx.x >> 6)) ca += red[w], cb += red2[w];
        ta += red[w];
        tb += red2[w];
    }
    ex_a = ca + ia - a;
    ex_b = cb + ib - b;
    tot_a = ta;
    tot_b = tb;
}

// ---- header of a dynamic block (RFC 1951 3.2.7) from the staged bits, in three steps: lane 0 reads the counts and the
// code-length code's lengths; lanes 0..18 build that code's 7-bit decode table (one symbol each); the workgroup decodes the
// literal/length and distance code lengths through it into L.lens (header_lengths_wg).
template <int T>
__device__ uint32_t header_counts(WgLds<T> &L, SegBits &b) {  // lane 0; the code-length code's lengths: header_cl_lens, by lanes 0..18
    const uint32_t bits = b.peek();
    const uint32_t hlit = (bits & 31u) + 257u, hdist = ((bits >> 5) & 31u) + 1u, hclen = ((bits >> 10) & 15u) + 4u;
    b.consume(14u);
    if (hlit > 286u || hdist > 30u) return INQ_INFLATE_BAD_HEADER;  // zlib: "too many length or distance symbols"
    for (uint32_t left = 3u * hclen; left;) {  // (consume takes < 32 bits)
        const uint32_t n = left < 30u ? left : 30u;
        b.consume(n);
        left -= n;
    }
    L.hlit = hlit;
    L.hdist = hdist;
    return 0u;
}
// lane i < 19 of a dynamic block's header at stage bit p0: the i-th 3-bit field behind the 17 bits of type and counts is the
// length of code-length symbol order[i] (RFC 1951 3.2.7); fields beyond HCLEN are zero
template <int T>
__device__ __forceinline__ void header_cl_lens(WgLds<T> &L, uint32_t p0, uint32_t i) {
    auto field = [&](uint32_t at, uint32_t mask) { return __builtin_amdgcn_alignbit(L.stage[(at >> 5) + 1u], L.stage[at >> 5], at & 31u) & mask; };
    const uint32_t hclen = field(p0 + 13u, 15u) + 4u;
    const uint64_t order = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                           5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    const uint32_t sy = i < 12u ? (uint32_t)(order >> (5u * i)) & 31u : (uint32_t)(order_hi >> (5u * (i - 12u))) & 31u;
    L.cl_len[sy] = (uint8_t)(i < hclen ? field(p0 + 17u + 3u * i, 7u) : 0u);
}

template <int T>
__device__ void header_cl_table(WgLds<T> &L, int tid) {  // lanes 0..18; cl_lut zero-filled by the caller
    const uint32_t n = L.cl_len[tid];
    uint64_t cnt = 0;  // 5 bits per length
    uint32_t rank = 0;
    for (int t = 0; t < 19; ++t) {
        const uint32_t l = L.cl_len[t];
        cnt += 1ull << (5u * l);
        rank += (l == n && t < tid) ? 1u : 0u;
    }
    if (tid == 0) {  // zlib: an over-subscribed or incomplete code-length code is an error
        int left = 1;
        for (int len = 1; len <= 7; ++len) {
            left = (left << 1) - (int)((cnt >> (5 * len)) & 31u);
            if (left < 0) break;
        }
        if (left != 0) atomicOr(&L.status, (uint32_t)INQ_INFLATE_BAD_HEADER);
    }
    if (n == 0u) return;
    uint32_t code = 0;
    for (uint32_t len = 1; len < n; ++len) code = (code + (uint32_t)((cnt >> (5u * len)) & 31u)) << 1;
    code += rank;
    if (code >> n) return;  // over-subscribed: reported by lane 0
    const uint32_t rev = __brev(code) >> (32u - n);
    for (uint32_t k = 0; k < (1u << (7u - n)); ++k) L.cl_lut[rev | (k << n)] = (uint8_t)((uint32_t)tid | (n << 5));
}

// The literal/length and distance code lengths, by the whole workgroup (every lane calls it; L.status == 0 and the
// code-length code is complete, so every 7-bit pattern is a code).  The code lengths are a Huffman stream of their own: lane k takes the symbols that start in bits
// [hp + 32 k, hp + 32 k + 32) - 128 lanes cover 4096 bits, a valid sequence of <= 316 lengths ends within 2212 - and, like
// the rounds of the block body, guesses its first symbol start, counts, and decodes again from where its left neighbour's
// chain ended until the lanes in front of the sequence's end agree.  A last pass, with every lane's place in lens[] known,
// writes the lengths and applies zlib's checks (inflate.c CODELENS: "invalid bit length repeat").
constexpr uint32_t kNoLast = 0xffu;
template <int T, bool FINAL>
__device__ __forceinline__ uint32_t cl_walk(WgLds<T> &L, uint32_t from, uint32_t seg_hi, uint32_t idx, uint32_t prev, uint32_t total, uint32_t &cnt,
                                            uint32_t &last, uint32_t &bad, uint32_t &end_at_total) {
    SegBits b;
    b.init(L.stage, from);
    uint32_t c = 0, own_last = kNoLast;
    const uint32_t hlit = L.hlit;
    while (b.pos < seg_hi) {
        if (FINAL && idx >= total) break;
        const uint32_t bits = b.peek();
        const uint32_t e = L.cl_lut[bits & 127u];
        const uint32_t n = e >> 5, sy = e & 31u;
        if (n == 0u) {  // cannot happen with a complete code; keeps the loop finite whatever the table holds
            if (FINAL) bad |= INQ_INFLATE_BAD_CODE;
            break;
        }
        uint32_t len, rep, adv;
        if (sy < 16u) len = sy, rep = 1u, adv = n, own_last = sy;
        else if (sy == 16u) {
            if (FINAL && idx == 0u) {
                bad |= INQ_INFLATE_BAD_HEADER;
                break;
            }
            len = prev, rep = 3u + ((bits >> n) & 3u), adv = n + 2u;
        } else if (sy == 17u) len = 0u, rep = 3u + ((bits >> n) & 7u), adv = n + 3u, own_last = 0u;
        else len = 0u, rep = 11u + ((bits >> n) & 127u), adv = n + 7u, own_last = 0u;
        b.consume(adv);
        c += rep;
        if (FINAL) {
            if (idx + rep > total) {
                bad |= INQ_INFLATE_BAD_HEADER;
                break;
            }
            prev = len;
            if (len)  // literal/length lengths at lens[0 .. hlit), distance lengths at lens[kMaxLit .. kMaxLit + hdist); zero-filled before
                for (uint32_t r = 0; r < rep; ++r) L.lens[idx + r < hlit ? idx + r : kMaxLit + (idx + r - hlit)] = (uint8_t)len;
            idx += rep;
            if (idx == total) end_at_total = b.pos;
        }
    }
    cnt = c;
    last = own_last;
    return b.pos;
}

template <int T>
__device__ void header_lengths_wg(WgLds<T> &L, uint32_t hp, uint32_t stage_bit0, int tid) {
    // 286 + 30 code lengths of <= 7 bits: a valid sequence ends within 2212 bits behind the 14 + 3 * 19 bits in front of it;
    // with fewer lanes than that the loop below could end without any lane reaching the sequence's end
    static_assert(T * 32 >= 2212 + 14, "one 32-bit slice per lane must cover the longest code-length sequence (T >= 128)");
    const uint32_t total = L.hlit + L.hdist;
    const uint32_t seg_hi = hp + 32u * ((uint32_t)tid + 1u);
    uint32_t start = hp + 32u * (uint32_t)tid;  // lane 0's is true
    uint32_t cnt = 0, last = kNoLast, bad = 0, dummy = 0;
    uint32_t end = cl_walk<T, false>(L, start, seg_hi, 0u, 0u, total, cnt, last, bad, dummy);
    uint32_t idx0 = 0;
    for (int it = 0; it <= T; ++it) {
        L.end_bit[tid] = end;
        __syncthreads();
        const uint32_t left = tid == 0 ? start : L.end_bit[tid - 1];
        const bool mismatch = left != start;
        uint32_t tot, d0, d1;
        wg_scan2<T>(cnt, 0u, idx0, d0, tot, d1, L.red, L.red2);
        const uint32_t first_bad = wg_first<T>(mismatch, L.red, L.red2);
        // the lanes in front of first_bad are final: do their symbols reach the end of the sequence?
        if (tid == 0) L.flag = first_bad == 0xffffu ? tot : 0u;
        __syncthreads();
        if ((uint32_t)tid == first_bad) L.flag = idx0;
        __syncthreads();
        if (L.flag >= total) break;  // (128 lanes x 32 bits hold >= 585 lengths: with no lane left to fix, the sum is beyond total)
        if (mismatch) {
            start = left;
            end = cl_walk<T, false>(L, start, seg_hi, 0u, 0u, total, cnt, last, bad, dummy);
        }
        __syncthreads();  // end_bit[] and flag read by everyone before they are rewritten
    }
    // the length in front of a lane's first symbol (a "copy the previous length" there needs it): the nearest lane to the
    // left that leaves one behind
    L.last_sh[tid] = (uint8_t)last;
    __syncthreads();
    uint32_t end_at_total = 0xffffffffu;
    if (idx0 < total) {
        uint32_t prev = 0;
        for (int j = tid - 1; j >= 0; --j) {
            const uint32_t v = L.last_sh[j];
            if (v != kNoLast) {
                prev = v;
                break;
            }
        }
        (void)cl_walk<T, true>(L, start, seg_hi, idx0, prev, total, cnt, last, bad, end_at_total);
    }
    if (bad) atomicOr(&L.status, bad);
    if (end_at_total != 0xffffffffu) L.P = stage_bit0 + end_at_total;
    // no lane saw the last length: the sequence runs out of the staged bits (cannot happen in a valid stream, see the
    // static_assert) - never continue from a stale L.P
    const int reached = __syncthreads_or(end_at_total != 0xffffffffu ? 1 : 0);
    if (tid == 0 && L.status == 0u && (!reached || L.lens[256] == 0)) L.status = INQ_INFLATE_BAD_HEADER;  // zlib: "missing end-of-block"
}

// ---- code construction by the whole workgroup from L.lens (zeroed beyond hlit / hdist)
template <int T>
__device__ void build_tables(WgLds<T> &L, int tid) {
    if (tid < 32) L.cnt[tid >> 4][tid & 15] = 0u;
    __syncthreads();
    for (int s = tid; s < kMaxLit + kMaxDist; s += T) {
        const uint32_t n = L.lens[s];
        if (n) atomicAdd(&L.cnt[s >= kMaxLit][n], 1u);
    }
    __syncthreads();
    if (tid < 2) {  // limits, bases, insertion slots; zlib's inflate_table rejects over-subscribed sets and
        const int tbl = tid;  // incomplete ones with any code longer than one bit
        int left = 1, maxlen = 0;
        uint32_t off = 0, first = 0;
        bool ok = true;
        L.limit[tbl][0] = 0u;
        for (int len = 1; len <= 15; ++len) {
            const uint32_t n = L.cnt[tbl][len];
            left = (left << 1) - (int)n;
            ok &= left >= 0;
            if (n) maxlen = len;
            L.limit[tbl][len] = ok ? (first + n) << (15 - len) : 0u;
            L.base[tbl][len] = off - first;
            L.run[tbl][len] = off;
            off += n;
            first = (first + n) << 1;
        }
        if (!(ok && (left == 0 || maxlen <= 1))) atomicOr(&L.status, (uint32_t)INQ_INFLATE_BAD_HEADER);
        if (!ok)
            for (int len = 1; len <= 15; ++len) L.limit[tbl][len] = 0u;  // nothing decodes
    }
    __syncthreads();
    // rank inside a length by ballot, 64 symbols at a time: wave 0 sorts the literal/length symbols, the last wave
    // the distance symbols (the same wave when T == 64)
    const int wave = tid >> 6, lane = tid & 63;
    for (int tbl = 0; tbl < 2; ++tbl) {
        if (wave != (tbl ? T / 64 - 1 : 0)) continue;
        const int nsym = tbl ? kMaxDist : kMaxLit, sbase = tbl ? kMaxLit : 0;
        for (int c0 = 0; c0 < nsym; c0 += 64) {
            const int s = c0 + lane;
            const uint32_t n = s < nsym ? L.lens[sbase + s] : 0u;
            // the lanes whose symbol has this lane's code length, from four ballots (one per bit of the length) instead of a
            // turn of the loop per length: every length of the chunk is ranked at once
            uint64_t m = ~0ull;
#pragma unroll
            for (uint32_t bit = 0; bit < 4u; ++bit) {
                const uint64_t bal = ballot64(((n >> bit) & 1u) != 0u);
                m &= ((n >> bit) & 1u) ? bal : ~bal;
            }
            const uint32_t r0 = L.run[tbl][n & 15u];
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (n) L.sorted[sbase + r0 + rank] = (uint16_t)s;
            __builtin_amdgcn_wave_barrier();  // every lane has read run[] before the first lane of a length moves it on
            if (n && rank == 0u) L.run[tbl][n] = r0 + (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // an entry's code length is the number of limits its pattern is not below (they grow with the length): the limits are read once,
    // the count has no dependent LDS read in it
    auto fill = [&](auto tbl_c, lut_t *lut, int bits) {
        constexpr int tbl = decltype(tbl_c)::value;
        uint32_t lim[16];
#pragma unroll
        for (int k = 1; k <= 15; ++k) lim[k] = L.limit[tbl][k];
        for (int e = tid; e < (1 << bits); e += T) {
            const uint32_t v15 = __brev((uint32_t)e) >> 17;
            uint32_t len = 1u;
#pragma unroll
            for (int k = 1; k < 15; ++k) len += (k < bits && v15 >= lim[k]) ? 1u : 0u;
            uint32_t ent = v15 < lim[15] ? kLongEntry : kNoCode;
            if (v15 < lim[bits]) {
                const uint32_t idx = L.base[tbl][len] + (v15 >> (15u - len));
                if (tbl == 0) ent = idx < (uint32_t)kMaxLit ? ll_entry(L.sorted[idx], len) : kNoCode;
                else ent = idx < (uint32_t)kMaxDist ? dist_entry(L.sorted[kMaxLit + idx], len) : kNoCode;
            }
            lut[e] = (lut_t)ent;
        }
    };
    fill(std::integral_constant<int, 0>{}, L.lut_ll, kLitBits);
    fill(std::integral_constant<int, 1>{}, L.lut_d, kDistBits);
    __syncthreads();
}

template <int T>
__device__ __forceinline__ void stage_load(WgLds<T> &L, const uint8_t *payload, const uint8_t *hard, uint32_t base_dw, int tid) {
    for (int k = tid; k < WgLds<T>::kStage; k += T) {
        const uint8_t *p = payload + 4ull * ((uint64_t)base_dw + (uint64_t)k);
        L.stage[k] = ld_u32(p < hard ? p : hard);  // behind the payload: only a corrupt stream consumes it
    }
}

}  // namespace

#ifndef INQ_WG_WAVES
#define INQ_WG_WAVES 4
#endif
template <int T, int FORM>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(INQ_WG_WAVES, INQ_WG_WAVES))) void bgzf_inflate_wg_kernel(InflateArgs a) {
    __shared__ WgLds<T> L;
    constexpr uint32_t LDS_STAGE_BYTES = 4u * (uint32_t)WgLds<T>::kStage;
    static_assert(LDS_STAGE_BYTES / 128u + 2u <= (uint32_t)T, "one lane per line of the next round's bits");
    const int tid = (int)threadIdx.x;
    const uint64_t bi = blockIdx.x;
    const inq_bgzf_block_t blk = a.blocks[bi];
    // host-checked, re-checked: the block's extents lie inside the buffers
    const bool extents_ok = !(blk.comp_off > a.comp_bytes || (uint64_t)blk.comp_len > a.comp_bytes - blk.comp_off || blk.out_off > a.out_bytes ||
                              (uint64_t)blk.isize > a.out_bytes - blk.out_off || blk.isize > 65536u);
    const uint8_t *payload = a.comp + blk.comp_off;
    const uint8_t *hard = a.comp + a.comp_bytes + 60;  // the buffer carries 64 bytes of padding
    uint8_t *out = a.out + blk.out_off;
    const uint32_t isize = blk.isize;
    const uint32_t payload_bits = blk.comp_len * 8u;
    if (tid == 0) {
        L.P = 0u;
        L.out = 0u;
        L.status = extents_ok ? 0u : (uint32_t)INQ_INFLATE_BAD_HEADER;
        L.last = 0u;
    }
#ifdef INQ_INFLATE_DEBUG_ENV
    // dev-time probe (debug_flags & 8): per block {deflate blocks, rounds, count passes, match sweeps, kilo-cycles in
    // header+tables, counting, commit, matches} into block_status[8 * bi ..] for bi < n_blocks / 8
    uint32_t dbg_n[5] = {0, 0, 0, 0, 0};
    uint64_t dbg_c[6] = {0, 0, 0, 0, 0, 0}, dbg_t = clock64();
#define DBG_N(i) ++dbg_n[i]
#define DBG_LAP(i) { const uint64_t now_ = clock64(); dbg_c[i] += now_ - dbg_t; dbg_t = now_; }
#else
#define DBG_N(i)
#define DBG_LAP(i)
#endif
    __syncthreads();
    while (L.status == 0u) {
        // ================= block header
        const uint32_t P0 = L.P;
        __syncthreads();  // everyone has read P (and the loop condition) before lane 0 moves on
        stage_load<T>(L, payload, hard, P0 >> 5, tid);
        for (int s = tid; s < kMaxLit + kMaxDist; s += T) L.lens[s] = 0;
        __syncthreads();
        SegBits hb;
        if (tid == 0) {
            hb.init(L.stage, P0 & 31u);
            uint32_t st = 0;
            if (P0 + 3u > payload_bits) st = INQ_INFLATE_INPUT_OVERRUN;
            else {
                const uint32_t bits = hb.peek();
                hb.consume(3u);
                L.last = bits & 1u;
                const uint32_t type = (bits >> 1) & 3u;
                L.type = type;
                if (type == 3u) st = INQ_INFLATE_BAD_HEADER;
                else if (type == 2u) st = header_counts<T>(L, hb);
                else if (type == 1u) {
                    L.hlit = 288u;
                    L.hdist = 32u;
                }
            }
            L.P = (P0 & ~31u) + hb.pos;
            if (st) L.status = st;
        }
        // (whatever the block's type is: the lengths are only used behind a dynamic block's counts)
        if (tid >= 64 && tid < 64 + 19) header_cl_lens<T>(L, P0 & 31u, (uint32_t)tid - 64u);
        for (int i = tid; i < 128; i += T) L.cl_lut[i] = 0;
        __syncthreads();
        if (L.status) break;
        if (L.type == 2u) {
            if (tid < 19) header_cl_table<T>(L, tid);
            __syncthreads();
            if (L.status == 0u) header_lengths_wg<T>(L, L.P - (P0 & ~31u), P0 & ~31u, tid);  // uniform: status was written before the barrier
            __syncthreads();
            if (L.status) break;
        }
        const uint32_t type = L.type;
        if (type == 0u) {  // stored: byte-align, LEN, NLEN, bytes
            const uint32_t q = (L.P + 7u) >> 3;  // byte offset of LEN
            const uint32_t o0 = L.out;
            uint32_t st = 0, len = 0;
            if ((uint64_t)q + 4u > blk.comp_len) st = INQ_INFLATE_INPUT_OVERRUN;
            else {
                const uint32_t w = ld_u32(payload + q);
                len = w & 0xffffu;
                if ((len ^ (w >> 16)) != 0xffffu) st = INQ_INFLATE_BAD_STORED;
                else if ((uint64_t)q + 4u + len > blk.comp_len) st = INQ_INFLATE_INPUT_OVERRUN;
                else if (len > isize - o0) st = INQ_INFLATE_OUTPUT_SIZE;
            }
            __syncthreads();
            if (st == 0u) {
                for (uint32_t k = (uint32_t)tid; k < len; k += T) out[o0 + k] = payload[q + 4u + k];
            }
            if (tid == 0) {
                if (st) L.status = st;
                L.P = (q + 4u + len) * 8u;
                L.out = o0 + len;
            }
            __syncthreads();
            if (L.status || L.last) break;
            continue;
        }
        if (type == 1u) {  // RFC 1951 3.2.6
            for (int s = tid; s < kMaxLit; s += T) L.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
            // 32 five-bit distance codes make the set complete; 30 and 31 never appear in valid data (dist_entry: not a code)
            for (int s = tid; s < 32; s += T) L.lens[kMaxLit + s] = 5;
            __syncthreads();
        }
        DBG_LAP(0);
        build_tables<T>(L, tid);
        DBG_N(0);
        DBG_LAP(4);
        if (L.status) break;

        // ================= rounds: T segments of 256 compressed bits each
        for (;;) {
            const uint32_t P = L.P, out0 = L.out;
            const uint32_t base_dw = P >> 5;
            __syncthreads();
            stage_load<T>(L, payload, hard, base_dw, tid);
            __syncthreads();
            const uint32_t seg_end = ((uint32_t)tid + 1u) * kSegBits;
            uint32_t start = tid == 0 ? (P & 31u) : (uint32_t)tid * kSegBits;
            uint32_t nbytes = 0, bad = 0, mid = kNoMid, ntok = 0, mid_tok = 0;
            uint32_t *const tok = a.tokens ? a.tokens + (bi * kTokCap) * (uint64_t)T + (uint32_t)tid : nullptr;
            uint32_t end = decode_segment<T, 0, FORM>(L, start, seg_end, nbytes, nullptr, 0u, 0u, bad, &mid, tok, &ntok, &mid_tok);
            DBG_N(1);
            DBG_N(2);
            // ---- until every chain starts where its left neighbour's ended (lanes behind the first stop do not matter)
            uint32_t ncommit = T;
            for (int it = 0; it <= T; ++it) {
                L.end_bit[tid] = end;
                __syncthreads();
                const uint32_t left = tid == 0 ? start : L.end_bit[tid - 1];
                const bool left_stopped = tid != 0 && (left & kStopped);
                const bool mismatch = !left_stopped && (left & 0x3fffffffu) != start;
                // first lane that has to decode again, first lane that stopped
                uint32_t first_bad, first_stop;
                wg_first2<T>(mismatch, (end & kStopped) != 0u, first_bad, first_stop, L.red, L.red2);
                if (first_bad == 0xffffu || first_stop < first_bad) {
                    ncommit = first_stop == 0xffffu ? (uint32_t)T : first_stop + 1u;
                    break;
                }
                if (mismatch) {
                    start = left & 0x3fffffffu;
                    end = decode_segment<T, 0, FORM>(L, start, seg_end, nbytes, nullptr, 0u, 0u, bad, &mid, tok, &ntok, &mid_tok);
                }
                DBG_N(2);
                __syncthreads();  // end_bit[] read by everyone before it is rewritten
            }
            // ---- places in the output
            if ((uint32_t)tid >= ncommit) nbytes = 0u;
            uint32_t off_b, tot_b, dummy0, dummy1;
            wg_scan2<T>(nbytes, 0u, off_b, dummy0, tot_b, dummy1, L.red, L.red2);
            L.off_sh[tid] = off_b;
            L.start_sh[tid] = (uint16_t)start;
            L.mid_sh[tid] = mid;
            L.ntok_sh[tid] = (uint16_t)(tok && ntok <= kTokCap ? ntok : kTokOverflow);
            L.midtok_sh[tid] = (uint16_t)mid_tok;
            if (tid == 0) L.off_sh[T] = tot_b;
            if (tid == (int)ncommit - 1) {  // the last chain of the round: where the next round starts, and why this one ended
                const uint32_t stop = end & (kStopped | kStopEob);
                if (stop == kStopped) bad |= INQ_INFLATE_BAD_CODE;
                L.P = (base_dw << 5) + (end & 0x3fffffffu);
                L.eob = stop == (kStopped | kStopEob);
                L.out = out0 + tot_b;
            }
            if (tot_b > isize - out0) bad |= INQ_INFLATE_OUTPUT_SIZE;  // uniform
            if (bad) atomicOr(&L.status, bad);
            __syncthreads();
            DBG_LAP(1);
            if (L.status) break;
            // where the next round (or the next block's header) will stage its bits from is known now: one dword of each of its
            // 128-byte lines is asked for here, so that the lines are in L2 when the commit is over and stage_load wants them
            // (the value is looked at behind the commit, which keeps the load; nothing depends on it)
            uint32_t warm = 0;
            if (tid < (int)(LDS_STAGE_BYTES / 128u + 2u)) {
                const uint8_t *p = payload + 4ull * (uint64_t)(L.P >> 5) + 128ull * (uint32_t)tid;
                warm = ld_u32(p < hard ? p : hard);
            }
            // ---- commit, in stretches of lanes whose bytes fit the root array: the round is counted once, whatever it inflates to
            uint32_t k0 = 0;
            while (k0 < ncommit) {
                const uint32_t r_lo = L.off_sh[k0];
                // k1 = first lane at or behind k0 whose bytes end beyond the array
                // ... or, on data that hardly inflates, the first one that would make the stretch longer than T / 2 segments: two
                // jobs per segment are then one job per lane, where a longer stretch has some lanes do two in a row while the
                // others wait
                const bool over = (uint32_t)tid >= k0 && (uint32_t)tid < ncommit &&
                                  (off_b + nbytes - r_lo > (uint32_t)WgLds<T>::kRoundCap || (uint32_t)tid >= k0 + (uint32_t)INQ_WG_STRETCH_SEGS);
                uint32_t k1 = wg_first<T>(over, L.red, L.red2);
                const bool lone = k1 == k0;  // a single lane exceeds the array: it writes its literals and matches in order by itself
                if (k1 == 0xffffu) k1 = ncommit;
                if (lone) k1 = k0 + 1u;
                const uint32_t r_hi = k1 < ncommit ? L.off_sh[k1] : tot_b;
                const uint32_t r0 = out0 + r_lo, nbytes_s = r_hi - r_lo;
                uint32_t cbad = 0;
                DBG_N(4);
                if (lone) {
                    if ((uint32_t)tid == k0) {
                        uint32_t nb2 = nbytes;
                        (void)decode_segment<T, 2, FORM>(L, start, seg_end, nb2, out, out0 + off_b, r0, cbad);
                    }
                } else {
                    // two jobs per segment, dealt over all lanes: the first half of its chain, and the rest from where the chain
                    // enters the second half (any lane can decode any segment: the bits are staged, the places are shared)
                    const uint32_t njobs = 2u * (k1 - k0);
                    for (uint32_t j = (uint32_t)tid; j < njobs; j += T) {
                        const uint32_t seg = k0 + (j >> 1), second = j & 1u;
                        const uint32_t s0 = L.start_sh[seg], m = L.mid_sh[seg], ob = L.off_sh[seg], oe = L.off_sh[seg + 1u];
                        const uint32_t send = (seg + 1u) * kSegBits;
                        uint32_t js, je, jo, jn;  // bits [js, je), bytes [jo, jo + jn) of the round
                        if (m == kNoMid) js = s0, je = send, jo = ob, jn = oe - ob;
                        else if (!second) js = s0, je = m & 0xffffu, jo = ob, jn = m >> 16;
                        else js = m & 0xffffu, je = send, jo = ob + (m >> 16), jn = oe - ob - (m >> 16);
                        if (m == kNoMid && second) continue;
                        const uint32_t nt_seg = L.ntok_sh[seg];
                        if (nt_seg != kTokOverflow) {  // the symbols are there already: write their roots
                            const uint32_t mt = L.midtok_sh[seg];
                            const uint32_t i0 = (m != kNoMid && second) ? mt : 0u, i1 = (m != kNoMid && !second) ? mt : nt_seg;
                            commit_tokens<T>(L, a.tokens + (bi * kTokCap) * (uint64_t)T + seg, i0, i1, out0 + jo, r0, jn, cbad);
                            continue;
                        }
                        uint32_t nb2 = jn;
                        (void)decode_segment<T, 1, FORM>(L, js, je, nb2, out, out0 + jo, r0, cbad);
                    }
                }
                if (cbad) atomicOr(&L.status, cbad);
                __syncthreads();  // literals (global) and roots (LDS) are visible to the workgroup
                DBG_LAP(2);
                if (L.status) break;
                // ---- matches: shorten every chain to its root by pointer jumping, then one gather
                if (!lone) {
                    // four consecutive bytes per lane and step (one 8-byte LDS access for their roots)
                    const uint32_t n4 = (nbytes_s + 3u) >> 2;
                    for (int sweep = 0; sweep < 16; ++sweep) {  // chains hop to an earlier lane's bytes each time: <= log2(T) + 1 sweeps
                        if (tid == 0) L.flag = 0u;
                        __syncthreads();
                        bool changed = false;
                        for (uint32_t g = (uint32_t)tid; g < n4; g += T) {
                            uint64_t w;
                            __builtin_memcpy(&w, &L.root[4u * g], 8);  // the array is a multiple of 4 long
                            const uint32_t r0_ = (uint32_t)w & 0xffffu, r1_ = (uint32_t)(w >> 16) & 0xffffu, r2_ = (uint32_t)(w >> 32) & 0xffffu,
                                           r3_ = (uint32_t)(w >> 48);
                            // a pointer into this stretch takes over what its target holds (a literal, an earlier byte, or a pointer
                            // further back); bytes behind nbytes_s hold stale roots of an earlier stretch: following them is
                            // harmless (bounded index), and nobody reads them
                            uint32_t a0 = r0_ - 32768u < kRootLit - 32768u ? L.root[(r0_ - 32768u) & (WgLds<T>::kRoundCap - 1)] : r0_;
                            uint32_t a1 = r1_ - 32768u < kRootLit - 32768u ? L.root[(r1_ - 32768u) & (WgLds<T>::kRoundCap - 1)] : r1_;
                            uint32_t a2 = r2_ - 32768u < kRootLit - 32768u ? L.root[(r2_ - 32768u) & (WgLds<T>::kRoundCap - 1)] : r2_;
                            uint32_t a3 = r3_ - 32768u < kRootLit - 32768u ? L.root[(r3_ - 32768u) & (WgLds<T>::kRoundCap - 1)] : r3_;
                            const uint64_t nw = (uint64_t)a0 | ((uint64_t)a1 << 16) | ((uint64_t)a2 << 32) | ((uint64_t)a3 << 48);
                            if (nw != w) __builtin_memcpy(&L.root[4u * g], &nw, 8);
                            // another sweep only while a pointer into this stretch is left (not "while something changed": the
                            // sweep that resolves the last pointers is the last one, nobody has to look again to find nothing:
                            // 1 - 3.5 % on all kinds of data; a second hop per sweep where the first lands on a pointer: +-1 %, not kept;
                            // profiles/r03_results/inflate_sweep_end.txt, inflate_sweep_hop2.txt)
                            changed |= (a0 - 32768u < kRootLit - 32768u) | (a1 - 32768u < kRootLit - 32768u) | (a2 - 32768u < kRootLit - 32768u) |
                                       (a3 - 32768u < kRootLit - 32768u);
                        }
                        if (changed) L.flag = 1u;
                        DBG_N(3);
                        __syncthreads();
                        if (L.flag == 0u) break;
                        __syncthreads();
                    }
                    DBG_LAP(5);
                    const uint8_t *from = out + r0 - 32768;  // root r lives at from[r]; never dereferenced in front of the block (distance check)
                    uint8_t *dstb = out + r0;
                    for (uint32_t g = (uint32_t)tid; g < n4; g += T) {
                        const uint32_t q = 4u * g;
                        uint64_t w;
                        __builtin_memcpy(&w, &L.root[q], 8);
                        const uint32_t r0_ = (uint32_t)w & 0xffffu, r1_ = (uint32_t)(w >> 16) & 0xffffu, r2_ = (uint32_t)(w >> 32) & 0xffffu,
                                       r3_ = (uint32_t)(w >> 48);
                        // every root is final now: a literal, or a byte of an earlier stretch (a pointer the sweeps left over cannot
                        // exist; it would be taken for a literal: no access depends on it)
                        if (q + 4u <= nbytes_s) {
                            uint32_t val;
                            if ((r0_ & r1_ & r2_ & r3_) >= kRootLit) {
                                val = (r0_ & 0xffu) | (r1_ & 0xffu) << 8 | (r2_ & 0xffu) << 16 | r3_ << 24;
                            } else if (r3_ < 32768u && r1_ == r0_ + 1u && r2_ == r0_ + 2u && r3_ == r0_ + 3u) {
                                val = ld_u32(from + r0_);  // four bytes of one match
                            } else {
                                const uint32_t b0 = r0_ < 32768u ? from[r0_] : r0_ & 0xffu, b1 = r1_ < 32768u ? from[r1_] : r1_ & 0xffu;
                                const uint32_t b2 = r2_ < 32768u ? from[r2_] : r2_ & 0xffu, b3 = r3_ < 32768u ? from[r3_] : r3_ & 0xffu;
                                val = b0 | b1 << 8 | b2 << 16 | b3 << 24;
                            }
                            st_u32(dstb + q, val);
                        } else {  // the stretch's last bytes: the roots behind them are stale, never followed
                            if (q < nbytes_s) dstb[q] = (uint8_t)(r0_ < 32768u ? from[r0_] : r0_);
                            if (q + 1u < nbytes_s) dstb[q + 1u] = (uint8_t)(r1_ < 32768u ? from[r1_] : r1_);
                            if (q + 2u < nbytes_s) dstb[q + 2u] = (uint8_t)(r2_ < 32768u ? from[r2_] : r2_);
                        }
                    }
                }
                __syncthreads();  // the stretch is final before the next one roots into it
                DBG_LAP(3);
                k0 = k1;
            }
            if (warm == 0x9e3779b9u) atomicOr(&L.flag, 0u);  // (no effect: it only makes the early load's value used)
            if (L.status) break;
            if (L.eob) break;
            if (L.P > payload_bits) {  // a round that ran off the payload without meeting end-of-block
                if (tid == 0) L.status = INQ_INFLATE_INPUT_OVERRUN;
                __syncthreads();
                break;
            }
        }
        if (L.status || L.last) break;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t st = L.status;
        if (st == 0u) {
            if (L.out != isize) st = INQ_INFLATE_OUTPUT_SIZE;
            else if (L.P > payload_bits) st = INQ_INFLATE_INPUT_OVERRUN;
        }
#ifdef INQ_INFLATE_DEBUG_ENV
        if (!(a.debug_flags & 8u))
#endif
        if (a.block_status) a.block_status[bi] = st;
        if (st) atomicOr(a.err, st);
    }
#ifdef INQ_INFLATE_DEBUG_ENV
    if ((a.debug_flags & 8u) && a.block_status) {  // {deflate blocks, rounds, count passes, kcyc header parse, tables, counting, commit, matches}
        __syncthreads();
        if (tid == 0 && bi < a.n_blocks / 8) {
            uint32_t *o = a.block_status + 8 * bi;
            o[0] = dbg_n[0] | dbg_n[4] << 8 | dbg_n[3] << 20, o[1] = dbg_n[1] | (uint32_t)(dbg_c[5] >> 10) << 8, o[2] = dbg_n[2];  // deflate blocks | stretches | sweeps; rounds | kcyc in the sweeps
            o[3] = (uint32_t)(dbg_c[0] >> 10), o[4] = (uint32_t)(dbg_c[4] >> 10), o[5] = (uint32_t)(dbg_c[1] >> 10), o[6] = (uint32_t)(dbg_c[2] >> 10),
            o[7] = (uint32_t)(dbg_c[3] >> 10);
        }
        if (tid == 0 && bi >= a.n_blocks / 8) (void)0;
    }
#endif
}

#ifndef INQ_WG_T
#define INQ_WG_T 128
#endif
uint64_t inflate_token_words(uint64_t n_blocks) { return n_blocks * (uint64_t)kTokCap * INQ_WG_T; }

void launch_bgzf_inflate_wg(const InflateArgs &a, hipStream_t s) {
    if (!a.n_blocks) return;
#ifndef INQ_WG_T
#define INQ_WG_T 128
#endif
    constexpr int T = INQ_WG_T;
    // two forms of the symbol loop: a second literal decoded from the same peek (literal-heavy data: sequence / quality
    // bytes, +18 - 24 %), or not (match-heavy data, CIGAR-only records: the second look costs 4.6 % there).  The caller says
    // which (InflateArgs::lit_pairs, from the code lengths of a few sampled block headers: deflate_probe.h); any condition
    // INSIDE the loop costs more than it saves.  (A third form for match-heavy data - a literal and the length code behind it
    // in one turn - was worth 1.4 % on CIGAR-only blocks at zlib level 1 and nothing at level 6: not kept,
    // profiles/r03_results/inflate_literal_length_fusion_ab.txt.)
    if (a.lit_pairs) hipLaunchKernelGGL((bgzf_inflate_wg_kernel<T, 1>), dim3((uint32_t)a.n_blocks), dim3(T), 0, s, a);
    else hipLaunchKernelGGL((bgzf_inflate_wg_kernel<T, 0>), dim3((uint32_t)a.n_blocks), dim3(T), 0, s, a);
}

}  // namespace inq
