// bgzf_inflate.hip — DEFLATE (RFC 1951) of BGZF blocks on the GPU.
//
// Replaces, for the device front end, the zlib inflate htslib runs under bam.fetch()/rc_records()
// (reference call sites src/call.rs:288,294,338,345; [3P] htslib bgzf.c).  BGZF blocks are
// independent (<= 64 KB of output each, no shared history), so the unit of parallelism is the
// block: ONE LANE PER BLOCK.  A 30x long-read BAM has millions of them; decoding is bit-serial
// inside a block whatever one does, so the chip is filled across blocks, not inside one.
//
// Per lane:
//   * a 64-bit bit buffer refilled with (unaligned) dword loads from the compressed stream;
//   * canonical Huffman decoding by first-code / count per length (the textbook method of RFC 1951
//     3.2.2): the 15 per-length counts of each code live in REGISTERS (static indexing in an unrolled
//     loop), only the symbol permutation lives in LDS, lane-interleaved ([entry][lane], u16), so
//     that lanes reading different entries still hit different banks pairwise;
//   * dynamic headers are decoded TWICE (pass 1 counts lengths, pass 2 places symbols), which
//     removes the 320-entry per-lane length array: the header is <1 % of a block's symbols;
//   * length/distance bases are computed arithmetically, no constant tables;
//   * output goes straight to global memory (the history window is the output itself); matches with
//     distance >= 4 are copied a dword at a time.
// Every access is bounded (input by the block's extent, output by ISIZE, distances by the bytes
// produced), so a corrupt stream ends in a per-block status, never in a fault.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inquistr_hip.h"
#include "front_kernels.h"

namespace inq {

namespace {

constexpr int kLanes = 64;
constexpr int kLitSyms = 288, kDistSyms = 32;

struct InflateLds {
    uint16_t sym[kLitSyms + kDistSyms][kLanes];  // sorted symbols: [0,288) literal/length, [288,320) distance
    uint16_t cnt[32][kLanes];                    // construction scratch: [0,16) literal/length, [16,32) distance
};

__device__ __forceinline__ uint32_t load_u32(const uint8_t *p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);  // unaligned-access-mode: one global_load_dword
    return w;
}
__device__ __forceinline__ void store_u32(uint8_t *p, uint32_t w) { __builtin_memcpy(p, &w, 4); }

struct BitReader {
    const uint8_t *p;    // next byte to load
    const uint8_t *end;  // end of this block's deflate payload
    uint64_t bb;
    uint32_t bc;
    __device__ __forceinline__ void refill() {  // afterwards bc > 32 (zeros behind the end of the payload)
        if (bc <= 32u) {
            const uint32_t w = p < end ? load_u32(p) : 0u;  // the buffer carries >= 4 bytes of padding
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
    __device__ __forceinline__ bool overrun(const uint8_t *start) const {
        const int64_t used = (int64_t)(p - start) * 8 - (int64_t)bc;
        return used > (int64_t)(end - start) * 8;
    }
};

// counts of one canonical code: count[len] for len 1..15, two 16-bit fields per register
struct Counts {
    uint32_t r[8];
    __device__ __forceinline__ uint32_t get(int len) const { return (r[len >> 1] >> ((len & 1) * 16)) & 0xffffu; }
};

// One symbol of a canonical code: walk the lengths, keeping the first code and the first symbol index of
// each length.  Returns the symbol or -1 (code not in the set).  `base` = first LDS entry of the code.
__device__ __forceinline__ int decode_sym(BitReader &b, const Counts &c, const InflateLds &L, int base, int lane) {
    uint32_t bits = b.peek();
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)c.get(len);
        if (code - count < first) {
            b.drop((uint32_t)len);
            return (int)L.sym[base + index + (code - first)][lane];
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
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

__device__ __forceinline__ void load_counts(Counts &c, const InflateLds &L, int base, int lane) {
#pragma unroll
    for (int k = 0; k < 8; ++k) c.r[k] = (uint32_t)L.cnt[base + 2 * k][lane] | ((uint32_t)L.cnt[base + 2 * k + 1][lane] << 16);
    c.r[0] &= 0xffff0000u;  // length 0 = "no code"
}

// counts -> running offsets (first sorted slot of each length).  False for the sets zlib's inflate_table
// rejects: over-subscribed, or incomplete with any code longer than one bit.
__device__ __forceinline__ bool counts_to_offsets(InflateLds &L, int base, int lane) {
    int left = 1, maxlen = 0;
    uint32_t off = 0;
    bool ok = true;
    for (int len = 1; len <= 15; ++len) {
        const uint32_t n = L.cnt[base + len][lane];
        left = (left << 1) - (int)n;
        ok &= left >= 0;
        if (n) maxlen = len;
        L.cnt[base + len][lane] = (uint16_t)off;
        off += n;
    }
    return ok && (left == 0 || maxlen <= 1);
}

__device__ void build_fixed(InflateLds &L, Counts &lit, Counts &dist, int lane) {
    // RFC 1951 3.2.6: literal/length lengths 8 (0..143), 9 (144..255), 7 (256..279), 8 (280..287)
    int k = 0;
    for (int s = 256; s < 280; ++s) L.sym[k++][lane] = (uint16_t)s;
    for (int s = 0; s < 144; ++s) L.sym[k++][lane] = (uint16_t)s;
    for (int s = 280; s < 288; ++s) L.sym[k++][lane] = (uint16_t)s;
    for (int s = 144; s < 256; ++s) L.sym[k++][lane] = (uint16_t)s;
    for (int s = 0; s < 30; ++s) L.sym[kLitSyms + s][lane] = (uint16_t)s;  // codes 30, 31 never decode
#pragma unroll
    for (int i = 0; i < 8; ++i) lit.r[i] = 0u, dist.r[i] = 0u;
    lit.r[3] = 24u << 16;             // len 7
    lit.r[4] = 152u | (112u << 16);   // len 8, len 9
    dist.r[2] = 30u << 16;            // len 5
}

// status bits per block
constexpr uint32_t kBadHeader = INQ_INFLATE_BAD_HEADER, kBadCode = INQ_INFLATE_BAD_CODE, kInputOverrun = INQ_INFLATE_INPUT_OVERRUN,
                   kOutputSize = INQ_INFLATE_OUTPUT_SIZE, kBadDistance = INQ_INFLATE_BAD_DISTANCE, kBadStored = INQ_INFLATE_BAD_STORED;

__device__ uint32_t read_dynamic_header(BitReader &b, InflateLds &L, Counts &lit, Counts &dist, int lane) {
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
    const BitReader mark = b;  // pass 2 starts here again
    for (int i = 0; i < 32; ++i) L.cnt[i][lane] = 0;
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
                const int slot = (is_dist ? 16 : 0) + len;
                if (pass == 0) {
                    L.cnt[slot][lane] = (uint16_t)(L.cnt[slot][lane] + 1);
                    has_eob |= idx == 256;
                } else {
                    const uint32_t at = L.cnt[slot][lane];
                    L.cnt[slot][lane] = (uint16_t)(at + 1);
                    L.sym[(is_dist ? kLitSyms : 0) + at][lane] = (uint16_t)(is_dist ? idx - hlit : idx);
                }
            }
        }
        if (pass == 0) {
            if (!has_eob) return kBadHeader;  // zlib: "missing end-of-block"
            load_counts(lit, L, 0, lane);
            load_counts(dist, L, 16, lane);
            if (!counts_to_offsets(L, 0, lane) || !counts_to_offsets(L, 16, lane)) return kBadHeader;
            b = mark;
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
    // host-checked, re-checked: the block's extents lie inside the buffers
    if (blk.comp_off > a.comp_bytes || (uint64_t)blk.comp_len > a.comp_bytes - blk.comp_off || blk.out_off > a.out_bytes ||
        (uint64_t)blk.isize > a.out_bytes - blk.out_off) {
        st = kBadHeader;
    } else {
        const uint8_t *start = a.comp + blk.comp_off;
        BitReader b{start, start + blk.comp_len, 0ull, 0u};
        uint8_t *out = a.out + blk.out_off;
        const uint32_t isize = blk.isize;
        uint32_t o = 0;
        Counts lit, dist;
        bool last = false;
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
                for (uint32_t k = 0; k < len; ++k) out[o + k] = q[k];
                o += len;
                b.p = q + len;
                b.bb = 0ull;
                b.bc = 0u;
                continue;
            }
            if (type == 3u) {
                st = kBadHeader;
                break;
            }
            if (type == 1u) build_fixed(L, lit, dist, lane);
            else if ((st = read_dynamic_header(b, L, lit, dist, lane)) != 0u) break;
            for (;;) {
                b.refill();
                int s = decode_sym(b, lit, L, 0, lane);
                if (s < 256) {
                    if (s < 0) {
                        st = kBadCode;
                        break;
                    }
                    if (o >= isize) {
                        st = kOutputSize;
                        break;
                    }
                    out[o++] = (uint8_t)s;
                    continue;
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
                const int d = decode_sym(b, dist, L, kLitSyms, lane);
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
                uint8_t *dst = out + o;
                const uint8_t *src = dst - dd;
                uint32_t k = 0;
                if (dd >= 4u)
                    for (; k + 4u <= len; k += 4u) store_u32(dst + k, load_u32(src + k));
                for (; k < len; ++k) dst[k] = src[k];
                o += len;
            }
        }
        if (st == 0u) {
            if (o != isize) st = kOutputSize;
            else if (b.overrun(start)) st = kInputOverrun;
        }
    }
    if (a.block_status) a.block_status[bi] = st;
    if (st) atomicOr(a.err, st);
}

void launch_bgzf_inflate(const InflateArgs &a, hipStream_t s) {
    if (!a.n_blocks) return;
    const uint64_t grid = (a.n_blocks + kLanes - 1) / kLanes;
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((uint32_t)grid), dim3(kLanes), 0, s, a);
}

}  // namespace inq
