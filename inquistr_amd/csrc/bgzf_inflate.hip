// bgzf_inflate.hip — DEFLATE (RFC 1951) of BGZF blocks on the GPU.
//
// Replaces, for the device front end, the zlib inflate htslib runs under bam.fetch()/rc_records()
// (reference call sites src/call.rs:288,294,338,345; [3P] htslib bgzf.c).  BGZF blocks are
// independent (<= 64 KB of output each, no shared history), so the unit of parallelism is the
// block: ONE LANE PER BLOCK.  A 30x long-read BAM has millions of them; decoding is bit-serial
// inside a block whatever one does, so the chip is filled across blocks, not inside one.
//
// Per lane:
//   * a 64-bit bit buffer; the compressed stream is fetched 16 bytes at a time, one fetch ahead of its use;
//   * canonical Huffman decoding (RFC 1951 3.2.2) without a per-length loop: the code length at the cursor
//     comes from 15 independent compares of the left-aligned next bits against per-length limits held
//     in REGISTERS (see Code); only the symbol permutation and 16 per-length bases live in LDS,
//     lane-interleaved ([entry][lane], 16-bit), so lanes reading different entries hit different banks;
//   * dynamic headers are decoded TWICE (pass 1 counts lengths, pass 2 places symbols), which
//     removes the 320-entry per-lane length array: the header is <1 % of a block's symbols;
//   * length/distance bases are computed arithmetically, no constant tables;
//   * output goes straight to global memory (the history window is the output itself); a match issues all
//     its loads before its first store (one memory round trip per <= 64 bytes, see copy_group).
// Every access is bounded (input by the block's extent, output by ISIZE, distances by the bytes
// produced), so a corrupt stream ends in a per-block status, never in a fault.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inquistr_hip.h"
#include "front_kernels.h"
#include "wave_primitives.h"

namespace inq {

namespace {

constexpr int kLanes = 64;
#ifndef INQ_LITERAL_RUN
#define INQ_LITERAL_RUN 4
#endif
constexpr int kLiteralRun = INQ_LITERAL_RUN;
constexpr int kLitSyms = 286, kDistSyms = 30;  // HLIT <= 286, HDIST <= 30 (larger headers are rejected like zlib does)

// 32 KB per wave, so that five waves share a CU's 160 KB.  All tables are lane-interleaved ([entry][lane]).
//   sym : the symbols of both codes sorted by (length, value), ONE BYTE each: value & 0xff.  Within a length
//         the literals (< 256) come before the length codes (>= 256), so one threshold per length (thr, the
//         sorted index of the first length code of that length) recovers the ninth bit.
//   cnt : construction scratch (per-length counts, then insertion slots), rows = length - 1 (+15: distances)
//   base: low half = base[len] of Code (signed), high half = thr[len]; one read serves both
struct InflateLds {
    uint8_t sym[kLitSyms + kDistSyms][kLanes];
    uint16_t cnt[30][kLanes];
    uint32_t base[30][kLanes];
    uint4 stage[kLanes];  // the next 16 compressed bytes of every lane, written by an LDS-DMA load
};
static_assert(sizeof(InflateLds) <= 32768, "five waves per CU need 32 KB per wave");

__device__ __forceinline__ uint32_t load_u32(const uint8_t *p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);  // unaligned-access-mode: one global_load_dword
    return w;
}
__device__ __forceinline__ void store_u32(uint8_t *p, uint32_t w) { __builtin_memcpy(p, &w, 4); }

// Bit cursor over the payload.  The compressed bytes are fetched 16 at a time, one fetch AHEAD of their
// use (cur = being consumed, nxt = in flight), so the decoder waits for a load issued ~10 symbols ago.
struct BitReader {
    const uint8_t *p;     // first byte NOT yet handed to the bit buffer (drives overrun / stored-block math)
    const uint8_t *end;   // end of this block's deflate payload
    const uint8_t *hard;  // last address a 16-byte fetch may start at (inside the padding of the whole buffer)
    uint64_t bb;
    uint32_t bc;
    const uint8_t *fetch;  // address of the 16 bytes in flight / waiting in LDS
    uint64_t cur_lo, cur_hi;
    uint32_t cur_n;        // dwords left in cur
    InflateLds *lds;       // stage[lane] receives the prefetch
    int lane;
    // Unconditional load from a clamped address: what lies behind the payload (trailer, next block) is
    // only ever consumed by a corrupt stream, which overrun() then reports.
    __device__ __forceinline__ const uint8_t *clamp(const uint8_t *q) const { return q < hard ? q : hard; }
    __device__ __forceinline__ void load16(const uint8_t *q, uint64_t &lo, uint64_t &hi) const {
        q = clamp(q);
        lo = (uint64_t)load_u32(q) | ((uint64_t)load_u32(q + 4) << 32);
        hi = (uint64_t)load_u32(q + 8) | ((uint64_t)load_u32(q + 12) << 32);
    }
    // The prefetch is a global -> LDS load (no register destination): a register destination carried around
    // the decode loop makes the compiler stage the load through temporaries and wait for it on the spot,
    // which turns the prefetch into a blocking read (measured: 20 % of the kernel).  The LDS-DMA is tracked
    // by the compiler as a pending LDS write, so the wait lands in front of the ds_read that consumes it,
    // 16 input bytes later.  Destination = wave-uniform base + lane * 16.
    __device__ __forceinline__ void prefetch16(const uint8_t *q) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)clamp(q),
                                         (__attribute__((address_space(3))) void *)&lds->stage[0], 16, 0, 0);
    }
    __device__ __forceinline__ void start(const uint8_t *from) {  // (re)position at a byte
        p = from;
        bb = 0ull;
        bc = 0u;
        load16(from, cur_lo, cur_hi);
        fetch = from + 16;
        cur_n = 4u;
        prefetch16(fetch);
    }
    __device__ __forceinline__ void refill() {  // afterwards bc > 32
        if (bc <= 32u) {
            if (cur_n == 0u) {
                const uint4 v = lds->stage[lane];
                cur_lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
                cur_hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
                fetch += 16;
                prefetch16(fetch);
                cur_n = 4u;
            }
            const uint32_t w = (uint32_t)cur_lo;
            cur_lo = (cur_lo >> 32) | (cur_hi << 32);
            cur_hi >>= 32;
            --cur_n;
            bb |= (uint64_t)w << bc;
            p += 4;
            bc += 32u;
        }
    }
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)bb; }
    __device__ __forceinline__ void drop(uint32_t n) {
        bb >>= n;
        bc -= n;
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {  // n <= 16, caller has refilled
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    // bits consumed beyond the payload?
    __device__ __forceinline__ bool overrun(const uint8_t *start_) const {
        const int64_t used = (int64_t)(p - start_) * 8 - (int64_t)bc;
        return used > (int64_t)(end - start_) * 8;
    }
};

// Copies n <= 16 bytes whose source and destination do not overlap: the four loads are issued before the
// first store, so the lane pays ONE memory round trip (on gfx9 a load's data is only usable once every
// earlier store of the wave has been acknowledged: load/store ping-pong costs a round trip per element).
struct Quad {
    uint32_t w0, w1, w2, w3;
};
__device__ __forceinline__ Quad load_quad(const uint8_t *src) {  // reads 16 bytes: inside the padded buffers
    return Quad{load_u32(src), load_u32(src + 4), load_u32(src + 8), load_u32(src + 12)};
}
__device__ __forceinline__ void store_quad(uint8_t *dst, const Quad &q, uint32_t n) {  // the first n <= 16 bytes of q
    const uint32_t w0 = q.w0, w1 = q.w1, w2 = q.w2, w3 = q.w3;
    if (n == 16u) {
        store_u32(dst, w0);
        store_u32(dst + 4, w1);
        store_u32(dst + 8, w2);
        store_u32(dst + 12, w3);
        return;
    }
    uint32_t k = 0;
    if (n >= 4u) store_u32(dst, w0), k = 4u;
    if (n >= 8u) store_u32(dst + 4, w1), k = 8u;
    if (n >= 12u) store_u32(dst + 8, w2), k = 12u;
    uint32_t t = k == 0u ? w0 : k == 4u ? w1 : k == 8u ? w2 : w3;
    for (; k < n; ++k, t >>= 8) dst[k] = (uint8_t)t;
}
__device__ __forceinline__ void copy_quad(uint8_t *dst, const uint8_t *src, uint32_t n) {
    const Quad q = load_quad(src);
    store_quad(dst, q, n);
}

__device__ __forceinline__ void copy_forward(uint8_t *dst, const uint8_t *src, uint32_t n) {
    for (uint32_t k = 0; k < n; k += 16u) copy_quad(dst + k, src + k, n - k < 16u ? n - k : 16u);
}

// LZ77 match: len bytes from dd bytes back; source and destination overlap when dd < len.
__device__ __forceinline__ void copy_match(uint8_t *dst, uint32_t dd, uint32_t len) {
    const uint8_t *src = dst - dd;
    if (dd >= len || dd >= 16u) {  // a quad never reads what it writes
        for (uint32_t k = 0; k < len; k += 16u) copy_quad(dst + k, src + k, len - k < 16u ? len - k : 16u);
        return;
    }
    // short period (dd < 16, dd < len): the output is the last dd bytes repeated; take them once
    const uint64_t lo = (uint64_t)load_u32(src) | ((uint64_t)load_u32(src + 4) << 32);
    const uint64_t hi = (uint64_t)load_u32(src + 8) | ((uint64_t)load_u32(src + 12) << 32);
    if (dd == 1u) {
        const uint32_t w = ((uint32_t)lo & 0xffu) * 0x01010101u;
        uint32_t k = 0;
        for (; k + 4u <= len; k += 4u) store_u32(dst + k, w);
        for (; k < len; ++k) dst[k] = (uint8_t)w;
        return;
    }
    uint32_t idx = 0;
    for (uint32_t k = 0; k < len; ++k) {
        dst[k] = (uint8_t)(idx < 8u ? lo >> (8u * idx) : hi >> (8u * (idx - 8u)));
        if (++idx == dd) idx = 0;
    }
}

// One canonical code, ready to decode without a per-length loop.  Left-align the next 15 bits MSB-first
// (v); codes of length L occupy [first[L], first[L] + count[L]) << (15 - L), shorter codes below longer
// ones, so with limit[L] = (first[L] + count[L]) << (15 - L) (non-decreasing in L) the length of the code
// at the cursor is 1 + #{L : v >= limit[L]}: fifteen independent compares against register constants, no
// divergence between lanes.  The symbol is sym[base[len] + (v >> (15 - len))] with
// base[len] = (first sorted slot of length len) - first[len], kept per lane in LDS.
typedef short short2v __attribute__((ext_vector_type(2)));

struct Code {
    // limit[L] - 1 as a signed 16-bit field, two lengths per register (L = 2k in the low half, 2k + 1 in the
    // high half; L = 0 holds 0x7fff so it never counts): v >= limit[L]  <=>  (limit[L] - 1) - v < 0, which
    // packed 16-bit subtract / arithmetic shift / add evaluate for two lengths per instruction.
    uint32_t lim[8];
};

// Returns the symbol or -1 (bit pattern outside the code).  `tbl` = 0 literal/length, 1 distance.
__device__ __forceinline__ int decode_sym(BitReader &b, const Code &c, const InflateLds &L, int tbl, int lane) {
    const uint32_t v = __brev(b.peek()) >> 17;  // next 15 bits, first bit of the stream on top
    const short2v vv = {(short)v, (short)v};
    short2v acc = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        short2v lm;
        __builtin_memcpy(&lm, &c.lim[k], 4);
        acc += (lm - vv) >> 15;  // -1 for every length whose limit v has reached
    }
    const uint32_t len = 1u - (uint32_t)(int)(acc.x + acc.y);
    if (len > 15u) return -1;
    const uint32_t e = L.base[tbl * 15 + (int)len - 1][lane];
    const int idx = (int)(short)(e & 0xffffu) + (int)(v >> (15u - len));
    b.drop(len);
    const int s8 = (int)L.sym[(tbl ? kLitSyms : 0) + idx][lane];
    return s8 + ((uint32_t)idx >= (e >> 16) ? 256 : 0);  // thr = 0xffff for distance codes
}

// the code-length code: 19 symbols of <= 7 bits, everything in registers
struct ClCode {
    uint32_t count;  // count[len] for len 1..7 in 4-bit... up to 19 needs 5 bits: 7 x 5 = 35 bits -> two words
    uint32_t count_hi;
    uint64_t syms_lo, syms_hi;  // sorted symbols, 5 bits each (12 in lo, 7 in hi)
    __device__ __forceinline__ uint32_t cnt(int len) const {  // len 1..7
        const int sh = (len - 1) * 5;
        return sh < 30 ? (count >> sh) & 31u : (count_hi >> (sh - 30)) & 31u;
    }
    __device__ __forceinline__ uint32_t sym(int i) const {
        return i < 12 ? (uint32_t)(syms_lo >> (5 * i)) & 31u : (uint32_t)(syms_hi >> (5 * (i - 12))) & 31u;
    }
};

__device__ __forceinline__ int decode_cl(BitReader &b, const ClCode &c) {
    uint32_t bits = b.peek();
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 7; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)c.cnt(len);
        if (code - count < first) {
            b.drop((uint32_t)len);
            return (int)c.sym(index + (code - first));
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// Per-length counts in L.cnt[base .. base + 15] -> limits (registers), base[] (LDS), and L.cnt becomes the
// running insertion slot of each length for the symbol placement.  False for the sets zlib's inflate_table
// rejects: over-subscribed, or incomplete with any code longer than one bit.
__device__ __forceinline__ bool build_code(Code &c, InflateLds &L, int tbl, int lane) {
    int left = 1, maxlen = 0;
    uint32_t off = 0, first = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) c.lim[i] = 0u;
    c.lim[0] = 0x7fffu;  // length 0 never counts
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        const uint32_t n = L.cnt[tbl * 15 + len - 1][lane];
        left = (left << 1) - (int)n;
        ok &= left >= 0;
        if (n) maxlen = len;
        const uint32_t lim = ok ? (first + n) << (15 - len) : 0u;  // <= 1 << 15 while not over-subscribed
        c.lim[len >> 1] |= ((lim - 1u) & 0xffffu) << ((len & 1) * 16);  // limit 0 -> -1: every v has reached it
        // thr starts at "no length code of this length"; place_symbol() lowers it when symbol 256 comes by
        L.base[tbl * 15 + len - 1][lane] = (((uint32_t)((int)off - (int)first)) & 0xffffu) | 0xffff0000u;
        L.cnt[tbl * 15 + len - 1][lane] = (uint16_t)off;
        off += n;
        first = (first + n) << 1;
    }
    return ok && (left == 0 || maxlen <= 1);
}

// Symbols arrive in ascending order (first the literal/length code, then the distance code).  Right before
// symbol 256 is placed, every length's insertion slot is the sorted index of its first length code.
__device__ __forceinline__ void place_symbol(InflateLds &L, int tbl, int sym, int len, int lane) {
    if (tbl == 0 && sym == 256) {
        for (int l = 0; l < 15; ++l) L.base[l][lane] = (L.base[l][lane] & 0xffffu) | ((uint32_t)L.cnt[l][lane] << 16);
    }
    const uint32_t at = L.cnt[tbl * 15 + len - 1][lane];
    L.cnt[tbl * 15 + len - 1][lane] = (uint16_t)(at + 1);
    L.sym[(tbl ? kLitSyms : 0) + at][lane] = (uint8_t)sym;
}

__device__ void build_fixed(InflateLds &L, Code &lit, Code &dist, int lane) {
    // RFC 1951 3.2.6: literal/length lengths 8 (0..143), 9 (144..255), 7 (256..279), 8 (280..287).  The codes
    // of 286 and 287 take part in the code construction (152 codes of length 8) but never appear in valid data
    // and have no room in the 286-entry table: the length-9 region is moved two slots down over their slots.
    // A stream that uses them reads the first two length-9 literals there, above thr[8], i.e. as 400 and 401:
    // not a length code, reported like any other invalid symbol.  Distance codes 30, 31 are left out (5 bits,
    // 30 codes): they decode as "not a code".
    for (int i = 0; i < 30; ++i) L.cnt[i][lane] = 0;
    L.cnt[7 - 1][lane] = 24;
    L.cnt[8 - 1][lane] = 152;
    L.cnt[9 - 1][lane] = 112;
    L.cnt[15 + 5 - 1][lane] = 30;
    (void)build_code(lit, L, 0, lane);
    (void)build_code(dist, L, 1, lane);
    L.base[9 - 1][lane] = (L.base[9 - 1][lane] & 0xffff0000u) | ((L.base[9 - 1][lane] - 2u) & 0xffffu);
    L.cnt[9 - 1][lane] = (uint16_t)(L.cnt[9 - 1][lane] - 2);
    for (int sidx = 0; sidx < 286; ++sidx) place_symbol(L, 0, sidx, sidx < 144 ? 8 : sidx < 256 ? 9 : sidx < 280 ? 7 : 8, lane);
    for (int sidx = 0; sidx < 30; ++sidx) place_symbol(L, 1, sidx, 5, lane);
}

// status bits per block
constexpr uint32_t kBadHeader = INQ_INFLATE_BAD_HEADER, kBadCode = INQ_INFLATE_BAD_CODE, kInputOverrun = INQ_INFLATE_INPUT_OVERRUN,
                   kOutputSize = INQ_INFLATE_OUTPUT_SIZE, kBadDistance = INQ_INFLATE_BAD_DISTANCE, kBadStored = INQ_INFLATE_BAD_STORED;

__device__ uint32_t read_dynamic_header(BitReader &b, InflateLds &L, Code &lit, Code &dist, int lane) {
    b.refill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return kBadHeader;  // zlib: "too many length or distance symbols"
    // code-length code lengths, 3 bits each, in the order of RFC 1951 3.2.7
    const uint64_t order = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                           10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    uint64_t cl = 0;  // 3 bits per symbol
    for (int i = 0; i < hclen; ++i) {
        b.refill();
        const uint32_t s = i < 12 ? (uint32_t)(order >> (5 * i)) & 31u : (uint32_t)(order_hi >> (5 * (i - 12))) & 31u;
        cl |= (uint64_t)b.take(3) << (3 * s);
    }
    ClCode cc;
    cc.count = cc.count_hi = 0;
    cc.syms_lo = cc.syms_hi = 0;
    {
        int n = 0, left = 1;
        for (int len = 1; len <= 7; ++len) {
            uint32_t k = 0;
            for (int s = 0; s < 19; ++s)
                if (((cl >> (3 * s)) & 7u) == (uint64_t)len) {
                    if (n < 12) cc.syms_lo |= (uint64_t)s << (5 * n);
                    else cc.syms_hi |= (uint64_t)s << (5 * (n - 12));
                    ++n;
                    ++k;
                }
            const int sh = (len - 1) * 5;
            if (sh < 30) cc.count |= k << sh;
            else cc.count_hi |= k << (sh - 30);
            left = (left << 1) - (int)k;
            if (left < 0) return kBadHeader;
        }
        if (left != 0) return kBadHeader;  // zlib: an incomplete code-length code is an error
    }
    const BitReader mark = b;  // pass 2 starts here again (the staged 16 bytes are re-fetched on restore)
    for (int i = 0; i < 30; ++i) L.cnt[i][lane] = 0;
    const int total = hlit + hdist;
    bool has_eob = false;
    for (int pass = 0; pass < 2; ++pass) {
        int idx = 0, prev = 0;
        while (idx < total) {
            b.refill();
            const int s = decode_cl(b, cc);
            if (s < 0) return kBadCode;
            int len, rep;
            if (s < 16) len = s, rep = 1;
            else if (s == 16) {
                if (idx == 0) return kBadHeader;
                len = prev, rep = 3 + (int)b.take(2);
            } else if (s == 17) len = 0, rep = 3 + (int)b.take(3);
            else len = 0, rep = 11 + (int)b.take(7);
            if (idx + rep > total) return kBadHeader;
            prev = len;
            if (len == 0) {
                idx += rep;
                continue;
            }
            for (int r = 0; r < rep; ++r, ++idx) {
                const bool is_dist = idx >= hlit;
                if (pass == 0) {
                    const int slot = (is_dist ? 15 : 0) + len - 1;
                    L.cnt[slot][lane] = (uint16_t)(L.cnt[slot][lane] + 1);
                    has_eob |= idx == 256;
                } else
                    place_symbol(L, is_dist ? 1 : 0, is_dist ? idx - hlit : idx, len, lane);
            }
        }
        if (pass == 0) {
            if (!has_eob) return kBadHeader;  // zlib: "missing end-of-block"
            if (!build_code(lit, L, 0, lane) || !build_code(dist, L, 1, lane)) return kBadHeader;
            b = mark;
            b.prefetch16(b.fetch);  // the LDS slot holds a later fetch of pass 1
        }
    }
    return 0u;
}

}  // namespace

__global__ __launch_bounds__(kLanes) void bgzf_inflate_kernel(InflateArgs a) {
    __shared__ InflateLds L;
    const int lane = (int)threadIdx.x;
    const uint64_t bi = (uint64_t)blockIdx.x * kLanes + (uint64_t)lane;
    if (bi >= a.n_blocks) return;
    const inq_bgzf_block_t blk = a.blocks[bi];
    uint32_t st = 0;
    const uint64_t clk0 = (a.debug_flags & 4u) ? clock64() : 0ull;
    // host-checked, re-checked: the block's extents lie inside the buffers
    if (blk.comp_off > a.comp_bytes || (uint64_t)blk.comp_len > a.comp_bytes - blk.comp_off || blk.out_off > a.out_bytes ||
        (uint64_t)blk.isize > a.out_bytes - blk.out_off) {
        st = kBadHeader;
    } else {
        const uint8_t *start = a.comp + blk.comp_off;
        BitReader b;
        b.lds = &L;
        b.lane = lane;
        b.end = start + blk.comp_len;
        b.hard = a.comp + a.comp_bytes + 32;  // the buffer carries 64 bytes of padding
        b.start(start);
        uint8_t *out = a.out + blk.out_off;
        const uint32_t isize = blk.isize;
        uint32_t o = 0;
        Code lit, dist;
        bool last = false;
        // A match of <= 16 bytes is split in time: its loads are issued when it is decoded, its stores when
        // the NEXT match (or the end of the block) comes up, so the memory round trip overlaps the decoding
        // of the symbols in between.  Literals in between go to other addresses; a later match that reads
        // these bytes issues its loads after these stores, which is all same-lane ordering needs.
        Quad pend = {0u, 0u, 0u, 0u};
        uint8_t *pend_dst = out;
        uint32_t pend_n = 0;
        while (!last && st == 0u) {
            b.refill();
            if (b.overrun(start)) {
                st = kInputOverrun;
                break;
            }
            last = b.take(1) != 0u;
            const uint32_t type = b.take(2);
            if (type == 0u) {  // stored: byte-align, LEN, NLEN, bytes
                b.drop(b.bc & 7u);
                b.refill();
                const uint8_t *q = b.p - (b.bc >> 3);  // byte position of the bit cursor
                if (q + 4 > b.end) {
                    st = kInputOverrun;
                    break;
                }
                const uint32_t w = load_u32(q);
                const uint32_t len = w & 0xffffu;
                if ((len ^ (w >> 16)) != 0xffffu) {
                    st = kBadStored;
                    break;
                }
                q += 4;
                if (q + len > b.end) {
                    st = kInputOverrun;
                    break;
                }
                if (len > isize - o) {
                    st = kOutputSize;
                    break;
                }
                copy_forward(out + o, q, len);
                o += len;
                b.start(q + len);
                continue;
            }
            if (type == 3u) {
                st = kBadHeader;
                break;
            }
            if (type == 1u) build_fixed(L, lit, dist, lane);
            else if ((st = read_dynamic_header(b, L, lit, dist, lane)) != 0u) break;
            for (;;) {
                // Up to kLiteralRun literals in a tight inner loop before the wave looks at lengths and
                // distances: with 64 lanes some lane has a match in almost every round, so the (long) match
                // path below is executed by the wave every time it is reached - once per run of literals
                // instead of once per symbol.  A lane that meets a non-literal waits for the others here.
                int s = 0;
                bool pending = false;  // s holds a non-literal symbol (or an error) to deal with
#pragma unroll 1
                for (int k = 0; k < kLiteralRun; ++k) {
                    b.refill();
                    s = decode_sym(b, lit, L, 0, lane);
                    if (s >= 256 || s < 0 || o >= isize) {
                        pending = true;
                        break;
                    }
                    if (!(a.debug_flags & 1u)) out[o] = (uint8_t)s;
                    ++o;
                }
                if (!pending) continue;
                if (s < 0) {
                    st = kBadCode;
                    break;
                }
                if (s < 256) {  // a literal with no room left
                    st = kOutputSize;
                    break;
                }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) {
                    st = kBadCode;
                    break;
                }
                // RFC 1951 3.2.5 length: 3..10 plain, then 4 codes per extra-bit count, 258 for the last
                uint32_t len;
                if (s < 8) len = 3u + (uint32_t)s;
                else if (s == 28) len = 258u;
                else {
                    const uint32_t eb = ((uint32_t)s - 4u) >> 2;
                    len = 3u + ((4u + ((uint32_t)s & 3u)) << eb) + b.take(eb);
                }
                b.refill();
                const int d = decode_sym(b, dist, L, 1, lane);
                if (d < 0 || d >= 30) {
                    st = kBadCode;
                    break;
                }
                uint32_t dd;
                if (d < 4) dd = 1u + (uint32_t)d;
                else {
                    const uint32_t eb = ((uint32_t)d - 2u) >> 1;
                    dd = 1u + ((2u + ((uint32_t)d & 1u)) << eb) + b.take(eb);
                }
                if (dd > o) {
                    st = kBadDistance;
                    break;
                }
                if (len > isize - o) {
                    st = kOutputSize;
                    break;
                }
                if (pend_n) {
                    store_quad(pend_dst, pend, pend_n);
                    pend_n = 0;
                }
                if (!(a.debug_flags & 2u)) {
                    if (len <= 16u && dd >= len) {
                        pend = load_quad(out + o - dd);
                        pend_dst = out + o;
                        pend_n = len;
                    } else
                        copy_match(out + o, dd, len);
                }
                o += len;
            }
        }
        if (pend_n) store_quad(pend_dst, pend, pend_n);
        if (st == 0u) {
            if (o != isize) st = kOutputSize;
            else if (b.overrun(start)) st = kInputOverrun;
        }
    }
    if (a.block_status) a.block_status[bi] = st;
    if (st) atomicOr(a.err, st);
    if ((a.debug_flags & 4u) && a.block_status) a.block_status[bi] = (uint32_t)((clock64() - clk0) >> 10);  // shader kilo-cycles of this lane
}

// ---------------------------------------------------------------- CRC32 of the inflated blocks
// htslib checks every block against the CRC32 in its trailer ([3P] bgzf.c check_header / inflate_block);
// a mismatch is a read error, i.e. a panic in the reference (src/call.rs:295,346).  One WAVE per block (round 1: one lane
// per block, 1.4 ms for any number of blocks).  With the register starting at 0 the CRC is linear in the bytes, so every lane
// can run over its own 16-byte granules (granule g belongs to lane g mod 64: a wave-instruction reads 1 KB of contiguous
// memory) as if all other bytes were zero: between two of its granules the register just travels through 1008 zero bytes,
// which is one lookup in four 256-entry tables, like a data word.  At the end lane i's register is moved through the
// zero bytes between its last granule and the end of the block and the 64 registers are XORed.  Moving a register through
// 2^k zero bytes is a 32 x 32 bit matrix over GF(2) (kCrcShift, squared up from the one-bit operator at compile time, as
// zlib's crc32_combine does at run time); the initial 0xffffffff enters as its own journey through the block's length.
struct CrcShift {
    uint32_t m[17][32];  // m[k] = the register moved through 2^k zero bytes, column by column
    uint32_t skip[32];   // ... through 1008 = 16 + 32 + ... + 512 zero bytes (from one granule of a lane to its next)
};
constexpr void gf2_square(const uint32_t (&in)[32], uint32_t (&out)[32]) {
    for (int j = 0; j < 32; ++j) {
        uint32_t v = in[j], sum = 0;
        for (int i = 0; v; ++i, v >>= 1)
            if (v & 1u) sum ^= in[i];
        out[j] = sum;
    }
}
constexpr CrcShift make_crc_shift() {
    CrcShift t{};
    uint32_t cur[32] = {}, nxt[32] = {};
    cur[0] = 0xEDB88320u;  // one zero BIT: x -> (x >> 1) ^ (poly if x & 1)
    for (int j = 1; j < 32; ++j) cur[j] = 1u << (j - 1);
    for (int sq = 0; sq < 3 + 17; ++sq) {  // squaring: 1 bit -> 2 -> 4 -> 8 bits = 1 byte = m[0], then m[k + 1] = m[k]^2
        if (sq >= 3)
            for (int j = 0; j < 32; ++j) t.m[sq - 3][j] = cur[j];
        gf2_square(cur, nxt);
        for (int j = 0; j < 32; ++j) cur[j] = nxt[j];
    }
    // skip = m[9] o m[8] o ... o m[4]: the image of every basis vector under the six operators in turn (they commute)
    for (int j = 0; j < 32; ++j) {
        uint32_t v = 1u << j;
        for (int k = 4; k <= 9; ++k) {
            uint32_t sum = 0, w = v;
            for (int i = 0; w; ++i, w >>= 1)
                if (w & 1u) sum ^= t.m[k][i];
            v = sum;
        }
        t.skip[j] = v;
    }
    return t;
}
__constant__ CrcShift kCrcShift = make_crc_shift();

__device__ __forceinline__ uint32_t crc_shift(uint32_t v, int k) {  // v moved through 2^k zero bytes
    uint32_t sum = 0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) sum ^= kCrcShift.m[k][i] & (0u - ((v >> i) & 1u));
    return sum;
}

__global__ __launch_bounds__(256) void bgzf_crc32_kernel(InflateArgs a) {
    __shared__ uint32_t T[4][256];  // slice-by-4: a data word
    __shared__ uint32_t S[4][256];  // the same shape for "1008 zero bytes"
    {
        const int i = (int)threadIdx.x;
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        T[0][i] = c;
        for (int t = 0; t < 4; ++t) {  // S[t][i] = skip applied to i << 8t: eight columns
            uint32_t sum = 0;
            for (int bit = 0; bit < 8; ++bit) sum ^= kCrcShift.skip[8 * t + bit] & (0u - (((uint32_t)i >> bit) & 1u));
            S[t][i] = sum;
        }
    }
    __syncthreads();
    {
        const int i = (int)threadIdx.x;
        uint32_t c = T[0][i];
        for (int t = 1; t < 4; ++t) {
            c = T[0][c & 0xffu] ^ (c >> 8);
            T[t][i] = c;
        }
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t bi = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (bi >= a.n_blocks) return;
    const inq_bgzf_block_t blk = a.blocks[bi];
    // blocks the inflate kernel rejected keep their status; extents were checked there
    if (a.block_status && a.block_status[bi]) return;
    if (blk.comp_off > a.comp_bytes || (uint64_t)blk.comp_len + 8u > a.comp_bytes - blk.comp_off || blk.out_off > a.out_bytes ||
        (uint64_t)blk.isize > a.out_bytes - blk.out_off || blk.isize > 65536u)
        return;
    const uint8_t *p = a.out + blk.out_off;
    const uint32_t n = blk.isize;
    uint32_t crc = 0u, at = 16u * lane, done_to = 0u;  // done_to = end of this lane's last granule
    auto word = [&](uint32_t w) {
        crc ^= w;
        crc = T[3][crc & 0xffu] ^ T[2][(crc >> 8) & 0xffu] ^ T[1][(crc >> 16) & 0xffu] ^ T[0][crc >> 24];
    };
    auto skip = [&]() { crc = S[0][crc & 0xffu] ^ S[1][(crc >> 8) & 0xffu] ^ S[2][(crc >> 16) & 0xffu] ^ S[3][crc >> 24]; };
    for (; at + 16u <= n; at += 1024u) {
        const uint32_t w0 = load_u32(p + at), w1 = load_u32(p + at + 4), w2 = load_u32(p + at + 8), w3 = load_u32(p + at + 12);
        if (done_to) skip();
        word(w0);
        word(w1);
        word(w2);
        word(w3);
        done_to = at + 16u;
    }
    if (at < n) {  // the block's last, partial granule
        if (done_to) skip();
        for (uint32_t k = at; k < n; ++k) crc = T[0][(crc ^ p[k]) & 0xffu] ^ (crc >> 8);
        done_to = n;
    }
    // through the zero bytes behind the lane's last granule (fewer than 1024), then XOR over the wave
    {
        const uint32_t rest = done_to ? n - done_to : 0u;
        for (int bit = 0; bit < 10; ++bit)
            if (ballot64(((rest >> bit) & 1u) != 0u))  // wave-uniform branch around the 32-step product
                crc = ((rest >> bit) & 1u) ? crc_shift(crc, bit) : crc;
    }
    for (int off = 32; off; off >>= 1) crc ^= (uint32_t)__shfl_xor((int)crc, off);
    // the register starts at 0xffffffff: its journey through n bytes, by the binary digits of n
    uint32_t init = 0xffffffffu;
    for (int bit = 0; bit < 17; ++bit)
        if ((n >> bit) & 1u) init = crc_shift(init, bit);
    crc = ~(crc ^ init);
    if (lane == 0u) {
        const uint32_t want = load_u32(a.comp + blk.comp_off + blk.comp_len);
        if (crc != want) {
            if (a.block_status) a.block_status[bi] = INQ_INFLATE_BAD_CRC;
            atomicOr(a.err, INQ_INFLATE_BAD_CRC);
        }
    }
}

void launch_bgzf_inflate(const InflateArgs &a, hipStream_t s) {
    if (!a.n_blocks) return;
    const uint64_t grid = (a.n_blocks + kLanes - 1) / kLanes;
    // Two kernels, one job.  A workgroup per block (bgzf_inflate_wg.hip) costs 0.54-0.82 ms per 1000 blocks and has no floor; a
    // lane per block (this file) takes 36-56 ms for anything up to ~65 000 blocks and then doubles.  Since the round-2 work on
    // the workgroup kernel it is the quicker one at every size measured (profiles/r02_front/README.md: 80 000 blocks 47 / 84 ms),
    // so "auto" means it; the lane-per-block kernel stays selectable and goes through the same tests.
    const bool wg = a.algo != 1u;
    if (wg) launch_bgzf_inflate_wg(a, s);
    else hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((uint32_t)grid), dim3(kLanes), 0, s, a);
    if (a.verify_crc) hipLaunchKernelGGL(bgzf_crc32_kernel, dim3((uint32_t)((a.n_blocks + 3) / 4)), dim3(256), 0, s, a);
}

// An empty launch makes the runtime load this translation unit's code object now (inq_ctx_create, on the
// context thread) instead of in front of the first real launch.
__global__ void preload_inflate_kernel() {}
void preload_inflate(hipStream_t s) { hipLaunchKernelGGL(preload_inflate_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace inq
