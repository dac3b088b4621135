// cigar_walk.h — the per-read CIGAR walk of inquiSTR `call`, one wavefront per read.
//
// Restates, for a whole wave at once, what the reference does one op at a time:
//   call_from_cigar          src/call.rs:377-413   (indel / soft-clip sum inside the window)
//   Record::reference_end    src/call.rs:298,351   ([3P] htslib bam_endpos: second CIGAR walk)
//   read filters             src/call.rs:297-302 (unphased), 349-355 (phased)
//   fetch() overlap rule     src/call.rs:288,338   ([3P] htslib iterator)
// Both CIGAR walks of the reference are fused into ONE pass over the packed ops.
//
// Data path: each lane loads 4 packed ops (one dwordx4, 16 B/lane, 1 KiB per wave
// instruction, fully coalesced); the reference position of every op comes from a
// 4-wide in-lane prefix + a DPP wave scan + a scalar carry between 256-op chunks.
#pragma once
#include "wave_primitives.h"

namespace inq {

// device status word bits (mapped to INQ_ERR_* by the host, same precedence as the oracle)
constexpr uint32_t ST_INDEX = 1u, ST_CIGAR_OP = 2u, ST_RANGE = 4u, ST_PHASE = 8u, ST_LOCUS = 16u;

// per-pair meta byte: low 3 bits are the public INQ_PAIR_* bits
constexpr uint32_t PM_CLIP = 1u, PM_FETCHED = 2u, PM_KEPT = 4u;
constexpr int PM_GRP_SHIFT = 4;  // bits 4-5: haplotype group 0 (none) / 1 / 2
constexpr uint32_t PM_CHOSEN = 64u;

constexpr uint32_t RB_UNMAPPED = 1u, RB_REVERSE = 2u, RB_HAS_HP = 4u, RB_IS_2D = 8u;

constexpr int kPrefetch = 4;  // first-chunk loads kept in flight per wave

struct Window {
    uint32_t se;     // start_ext = start - 10            (src/call.rs:285,335)
    uint32_t ee;     // end_ext   = end + 10              (src/call.rs:286,336)
    uint32_t se1;    // se + 1
    uint32_t width;  // ee - se1: an op at refpos counts iff (refpos - se1) <u width  ==  se < refpos < ee
    uint32_t minlen;
};

struct BatchView {
    const uint4 *cigar4;  // packed ops viewed as 16-byte groups
    const uint4 *reads;   // inq_read_t as uint4: x=cigar_off4 y=n_cigar z=pos w=mapq|bits<<8|phase<<16
    const uint32_t *pair_read;
    uint64_t n_reads;
    uint64_t n_cigar4;  // n_cigar_words / 4
};

// Per-lane descriptor of the pair this lane "owns" inside a block of <= 64 pairs.
struct PairMeta {
    uint32_t off4, nc, pos, misc;
};

// Loads descriptors for pairs [first, first+cnt): lane l < cnt owns pair first+l.
// A descriptor that points outside the buffers is replaced by an empty read and flagged.
__device__ __forceinline__ PairMeta load_pair_meta(const BatchView &b, uint64_t first, int cnt, int lane,
                                                    uint32_t &status, bool &valid) {
    PairMeta m{0u, 0u, 0u, 0u};
    valid = false;
    if (lane < cnt) {
        uint32_t ri = b.pair_read[first + (uint64_t)lane];
        if ((uint64_t)ri < b.n_reads) {
            uint4 r = b.reads[ri];
            uint64_t n4 = ((uint64_t)r.y + 3u) >> 2;
            if (r.y < 0x80000000u && (uint64_t)r.x + n4 <= b.n_cigar4) {
                m.off4 = r.x;
                m.nc = r.y;
                m.pos = r.z;
                m.misc = r.w;
                valid = true;
            } else {
                status |= ST_INDEX;
            }
        } else {
            status |= ST_INDEX;
        }
    }
    return m;
}

// One dwordx4 of chunk c of a read: lane -> ops [256c + 4*lane, +4).  Lanes past the read's
// (4-padded) end get zeros = four `0M`, which are no-ops for every rule below.
__device__ __forceinline__ uint4 load_chunk(const BatchView &b, uint32_t off4, uint32_t nc, uint32_t c, int lane) {
    uint32_t i4 = c * 64u + (uint32_t)lane;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (i4 * 4u < nc) w = b.cigar4[(uint64_t)off4 + i4];
    return w;
}

struct LaneAcc {
    int64_t sum;      // this lane's share of `call`
    uint32_t clip;    // a soft clip was counted
    uint32_t maxop;   // largest op code seen
    uint32_t range;   // bit 31 of any running reference position
};

// Processes 256 ops (4 per lane).  `carry` (wave-uniform) is the reference position before the
// first op of the chunk, i.e. reference_position of src/call.rs:380 advanced over earlier chunks.
__device__ __forceinline__ void walk_chunk(const uint4 w, const Window &W, uint32_t cand_mask, uint32_t &carry,
                                           LaneAcc &A) {
    const uint32_t op0 = w.x & 15u, op1 = w.y & 15u, op2 = w.z & 15u, op3 = w.w & 15u;
    const uint32_t l0 = w.x >> 4, l1 = w.y >> 4, l2 = w.z >> 4, l3 = w.w >> 4;
    // ops that consume the reference: M D N = X  -> bits 0,2,3,7,8 (src/call.rs:384-392,404)
    constexpr uint32_t kConsume = 0x18Du;
    const uint32_t a0 = ((kConsume >> op0) & 1u) ? l0 : 0u;
    const uint32_t a1 = ((kConsume >> op1) & 1u) ? l1 : 0u;
    const uint32_t a2 = ((kConsume >> op2) & 1u) ? l2 : 0u;
    const uint32_t a3 = ((kConsume >> op3) & 1u) ? l3 : 0u;
    const uint32_t e1 = a0, e2 = e1 + a1, e3 = e2 + a2, tot = e3 + a3;
    const uint32_t incl = wave_inclusive_scan_u32(tot);
    const uint32_t base = carry + (incl - tot);  // reference position at this lane's first op
    A.range |= (carry + incl);
    A.maxop = max(max(A.maxop, max(op0, op1)), max(op2, op3));
    int32_t s = 0;
    uint32_t clip = 0;
#define INQ_OP(op, len, e)                                                                    \
    {                                                                                         \
        const bool hit = ((cand_mask >> (op)) & 1u) && (len) > W.minlen &&                    \
                         ((base + (e)) - W.se1) < W.width;                                     \
        const int32_t v = ((op) == 2u) ? -(int32_t)(len) : (int32_t)(len);                    \
        s += hit ? v : 0;                                                                     \
        clip |= (hit && (op) == 4u) ? 1u : 0u;                                                \
    }
    INQ_OP(op0, l0, 0u)
    INQ_OP(op1, l1, e1)
    INQ_OP(op2, l2, e2)
    INQ_OP(op3, l3, e3)
#undef INQ_OP
    A.sum += (int64_t)s;  // |s| < 4 * 2^28
    A.clip |= clip;
    carry += readlane_u32(incl, 63);
}

// Wave-uniform outcome of one (locus, read) pair
struct PairOut {
    int64_t call;   // Call value                              src/call.rs:67-71
    uint32_t meta;  // PM_CLIP | PM_FETCHED | PM_KEPT | group
};

// Walks the reads of pairs [0, cnt) described by `m` (lane k owns pair k) and hands each
// pair's wave-uniform result to sink(k, PairOut).  First chunks of the next kPrefetch reads
// are kept in flight while the current one is reduced.
template <bool UNPHASED, class Sink>
__device__ __forceinline__ void walk_pairs(const BatchView &b, const PairMeta &m, uint64_t valid_mask, int cnt,
                                           const Window &W, int lane, uint32_t &status, Sink sink) {
    uint4 q0, q1, q2, q3;
    {
        auto first = [&](int k) -> uint4 {
            if (k < cnt) return load_chunk(b, readlane_u32(m.off4, k), readlane_u32(m.nc, k), 0u, lane);
            return make_uint4(0u, 0u, 0u, 0u);
        };
        q0 = first(0);
        q1 = first(1);
        q2 = first(2);
        q3 = first(3);
    }
    for (int k = 0; k < cnt; ++k) {
        uint4 w = q0;
        q0 = q1;
        q1 = q2;
        q2 = q3;
        {
            const int kn = k + kPrefetch;
            q3 = make_uint4(0u, 0u, 0u, 0u);
            if (kn < cnt) q3 = load_chunk(b, readlane_u32(m.off4, kn), readlane_u32(m.nc, kn), 0u, lane);
        }
        const uint32_t off4 = readlane_u32(m.off4, k);
        const uint32_t nc = readlane_u32(m.nc, k);
        const uint32_t pos = readlane_u32(m.pos, k);
        const uint32_t misc = readlane_u32(m.misc, k);
        const uint32_t mapq = misc & 0xffu, bits = (misc >> 8) & 0xffu, phase = (misc >> 16) & 0xffu;
        const bool pvalid = (valid_mask >> k) & 1ull;
        // soft clips of an accidental-2D read never count (src/call.rs:394): drop S from the candidates
        const uint32_t cand_mask = (bits & RB_IS_2D) ? 0x06u : 0x16u;  // I=1, D=2, S=4

        uint32_t carry = pos + 1u;  // (reference_start + 1) as u32, src/call.rs:380
        LaneAcc A{0, 0u, 0u, carry};
        const uint32_t nchunks = (nc + 255u) >> 8;
        uint4 wn = make_uint4(0u, 0u, 0u, 0u);
        if (nchunks > 1u) wn = load_chunk(b, off4, nc, 1u, lane);
        for (uint32_t c = 0;;) {
            walk_chunk(w, W, cand_mask, carry, A);
            ++c;
            if (c >= nchunks) break;
            w = wn;
            if (c + 1u < nchunks) wn = load_chunk(b, off4, nc, c + 1u, lane);
        }
        // ---- wave-uniform epilogue of the pair ----
        const uint64_t nz = ballot64(A.sum != 0);
        const int64_t call = nz ? wave_reduce_add_i64(A.sum) : 0;
        const bool clipped = ballot64(A.clip != 0u) != 0ull;
        if (ballot64(A.maxop > 8u)) status |= ST_CIGAR_OP;  // rust-htslib cigar() would panic
        if (ballot64((A.range >> 31) != 0u)) status |= ST_RANGE;
        // [3P] bam_endpos: rlen = unmapped ? 0 : sum(ref-consuming); rlen == 0 -> 1
        uint32_t rlen = carry - (pos + 1u);
        if ((bits & RB_UNMAPPED) || rlen == 0u) rlen = 1u;
        const uint32_t rend = pos + rlen;  // reference_end() as u32
        // fetch(): pos < end_ext && endpos > start_ext (signed pos, pos >= -1 inside the domain)
        const bool fetched = pvalid && ((int32_t)pos < 0 || pos < W.ee) && rend > W.se;
        bool skip;
        if (UNPHASED)
            skip = W.se < pos || rend < W.ee || mapq <= 10u;  // src/call.rs:297-302
        else
            skip = !(bits & RB_HAS_HP) || (W.se < pos && rend < W.ee) || mapq <= 10u;  // :349-355
        const bool kept = fetched && !skip;
        uint32_t grp = 0u;
        if (!UNPHASED && kept) {
            if (phase > 2u)
                status |= ST_PHASE;  // calls.get_mut(&phase).unwrap() panics, src/call.rs:358
            else
                grp = phase;
        }
        PairOut o;
        o.call = pvalid ? call : 0;
        o.meta = pvalid ? ((clipped ? PM_CLIP : 0u) | (fetched ? PM_FETCHED : 0u) | (kept ? PM_KEPT : 0u) |
                           (grp << PM_GRP_SHIFT))
                        : 0u;
        sink(k, o);
    }
}

}  // namespace inq
