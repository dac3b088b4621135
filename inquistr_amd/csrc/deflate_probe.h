// deflate_probe.h — how literal-heavy is a span?  Host side, a few hundred bytes per sampled BGZF block.
//
// The workgroup inflate has two forms of its symbol loop (bgzf_inflate_wg.hip): one looks for a second literal behind every
// literal (+18 - 24 % on sequence / quality bytes), one does not (match-heavy data: CIGAR-only records lose 4.6 % to the wasted
// look).  Which data a span holds is written in the first dynamic-Huffman header of its blocks (RFC 1951 3.2.7): a literal's share
// of the symbols is 2^-length of its code, so the 256 literal code lengths sum up to the literals' share of the code space.
// Quality-like bytes: 0.93, nanopore-like records 0.72 - 0.87, packed bases 0.45 - 0.48, CIGAR-only records 0.35 - 0.39.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/inquistr_hip.h"

namespace inq {

// share of the code space the 256 literals hold in the first deflate block of `payload`, in units of 2^-15; -1 if that block is
// not a dynamic-Huffman block or its header does not parse (stored / fixed blocks, damaged data: the inflate itself will say)
inline int deflate_literal_mass(const uint8_t *payload, size_t len) {
    uint64_t pos = 0;
    const uint64_t nbits = (uint64_t)len * 8;
    auto bits = [&](int k) -> int {  // -1 behind the end
        if (pos + (uint64_t)k > nbits) return -1;
        uint32_t v = 0;
        for (int i = 0; i < k; ++i, ++pos) v |= (uint32_t)((payload[pos >> 3] >> (pos & 7)) & 1u) << i;
        return (int)v;
    };
    const int hdr = bits(3);
    if (hdr < 0 || ((hdr >> 1) & 3) != 2) return -1;
    const int hlit = bits(5), hdist = bits(5), hclen = bits(4);
    if (hlit < 0 || hdist < 0 || hclen < 0) return -1;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen + 4; ++i) {
        const int v = bits(3);
        if (v < 0) return -1;
        cl[order[i]] = (uint8_t)v;
    }
    // canonical code of the code-length alphabet (<= 7 bits): (length, code) -> symbol by search; 19 symbols, a handful of lookups
    uint16_t code_of[19] = {0};
    {
        int count[8] = {0}, next[8] = {0};
        for (int s = 0; s < 19; ++s) ++count[cl[s]];
        count[0] = 0;
        int c = 0;
        for (int l = 1; l <= 7; ++l) {
            c = (c + count[l - 1]) << 1;
            next[l] = c;
        }
        for (int s = 0; s < 19; ++s)
            if (cl[s]) code_of[s] = (uint16_t)next[cl[s]]++;
    }
    auto decode = [&]() -> int {
        int code = 0;
        for (int l = 1; l <= 7; ++l) {
            const int b = bits(1);
            if (b < 0) return -1;
            code = (code << 1) | b;
            for (int s = 0; s < 19; ++s)
                if (cl[s] == l && code_of[s] == code) return s;
        }
        return -1;
    };
    const int n_lit = hlit + 257;
    int idx = 0, prev = 0;
    uint32_t mass = 0;
    while (idx < 256 && idx < n_lit) {
        const int s = decode();
        if (s < 0) return -1;
        int rep = 1, val = s;
        if (s == 16) {
            if (idx == 0) return -1;
            const int x = bits(2);
            if (x < 0) return -1;
            rep = 3 + x, val = prev;
        } else if (s == 17) {
            const int x = bits(3);
            if (x < 0) return -1;
            rep = 3 + x, val = 0;
        } else if (s == 18) {
            const int x = bits(7);
            if (x < 0) return -1;
            rep = 11 + x, val = 0;
        }
        for (int r = 0; r < rep && idx < 256; ++r, ++idx)
            if (val) mass += 1u << (15 - val);
        prev = val;
    }
    return (int)mass;
}

// 1 = literal-heavy (or unknown): the pair form of the symbol loop; 0 = match-heavy.  Up to 9 blocks spread over the table are read.
inline uint32_t inflate_wants_literal_pairs(const uint8_t *comp, uint64_t comp_bytes, const inq_bgzf_block_t *blocks, uint64_t n_blocks) {
    if (!comp || !blocks || !n_blocks) return 1u;
    constexpr int kSamples = 9;
    constexpr int kThreshold = 13800;  // of 32768: CIGAR-only records 11 600 - 12 900, packed bases 14 800 - 15 900, quality bytes 30 000+
    int masses[kSamples], n = 0;
    for (int k = 0; k < kSamples; ++k) {
        const uint64_t i = n_blocks <= (uint64_t)kSamples ? (uint64_t)k : (n_blocks - 1) * (uint64_t)k / (kSamples - 1);
        if (i >= n_blocks) break;
        const inq_bgzf_block_t &b = blocks[i];
        if (b.comp_off > comp_bytes || b.comp_len > comp_bytes - b.comp_off) continue;
        const int m = deflate_literal_mass(comp + b.comp_off, b.comp_len);
        if (m >= 0) masses[n++] = m;
    }
    if (!n) return 1u;
    for (int a = 1; a < n; ++a)  // median by insertion sort
        for (int c = a; c > 0 && masses[c] < masses[c - 1]; --c) {
            const int t = masses[c];
            masses[c] = masses[c - 1];
            masses[c - 1] = t;
        }
    return masses[n / 2] >= kThreshold ? 1u : 0u;
}

}  // namespace inq
