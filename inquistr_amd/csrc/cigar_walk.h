// cigar_walk.h — the per-read CIGAR walk of inquiSTR `call`, one wavefront per read.
//
// Restates, for a whole wave at once, what the reference does one op at a time:
//   call_from_cigar          src/call.rs:377-413   (indel / soft-clip sum inside the window)
//   Record::reference_end    src/call.rs:298,351   ([3P] htslib bam_endpos: second CIGAR walk)
//   read filters             src/call.rs:297-302 (unphased), 349-355 (phased)
//   fetch() overlap rule     src/call.rs:288,338   ([3P] htslib iterator)
// Both CIGAR walks of the reference are fused into ONE pass over the packed ops.
//
// Data path (per wave):
//   * ops stream in as 256-op chunks: one buffer_load_dwordx4 per lane (16 B/lane, 1 KiB per wave
//     instruction, coalesced).  The buffer descriptor's range check returns 0 (= `0M`, a no-op)
//     past the read's end, so the load is never predicated.  Four chunk loads are always in
//     flight per wave (flattened over reads and over the chunks of long reads).
//   * reference positions: 4-wide in-lane prefix + DPP wave scan + scalar carry between chunks.
//   * only the few lanes whose ops can start inside [start_ext, end_ext) matter for the call:
//     they are compacted into an LDS queue and evaluated 64 at a time (one queue entry per
//     lane), their signed lengths land in the owning read's LDS accumulator (ds_add_u64).
//     The whole wave therefore pays the per-op window/minlen/sign logic once per ~64 window
//     lanes instead of once per chunk.
#pragma once
#include "wave_primitives.h"

namespace inq {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// device status word bits (mapped to INQ_ERR_* by the host, same precedence as the oracle)
constexpr uint32_t ST_INDEX = 1u, ST_CIGAR_OP = 2u, ST_RANGE = 4u, ST_PHASE = 8u, ST_LOCUS = 16u, ST_HINT = 32u, ST_AUX = 64u,
                   ST_INTERNAL = 128u;  // a grid barrier of locus_call_tail gave up waiting (never seen; the grid drains all the same)

// per-pair meta byte: low 3 bits are the public INQ_PAIR_* bits
constexpr uint32_t PM_CLIP = 1u, PM_FETCHED = 2u, PM_KEPT = 4u;
constexpr int PM_GRP_SHIFT = 4;  // bits 4-5: haplotype group 0 (none) / 1 / 2
constexpr uint32_t PM_CHOSEN = 64u;

constexpr uint32_t RB_UNMAPPED = 1u, RB_REVERSE = 2u, RB_HAS_HP = 4u, RB_IS_2D = 8u, RB_SA_PANIC = 16u;

constexpr int kQueueCap = 128;  // window-lane queue entries per wave

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

// LDS owned by one wave
struct QueueEntry {
    u32x4 w;     // the lane's 4 packed ops
    uint2 info;  // .x = reference position of the first op minus (start_ext+1); .y = read slot | is2d<<6
    uint2 pad;   // 32-byte stride: one address register serves both stores
};
struct WaveLds {
    QueueEntry q[kQueueCap];
    unsigned long long acc[64];  // per read slot: the Call value (two's complement)
    unsigned int flags[64];      // per read slot: bit0 = a soft clip was counted
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
            if (r.y < 0x10000000u && (uint64_t)r.x + n4 <= b.n_cigar4) {
                m.off4 = r.x;
                m.nc = r.y | ((r.w & (RB_IS_2D << 8)) ? 0x80000000u : 0u);  // bit 31 = is_accidental_2d
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

// The same in two stages, so a wave can fetch the descriptors of its NEXT block of reads while it walks
// the current one: stage A (pair index) one block ahead of stage B (descriptor), checks at use time.
__device__ __forceinline__ uint32_t meta_stage_a(const BatchView &b, uint64_t first, int cnt, int lane) {
    return lane < cnt ? b.pair_read[first + (uint64_t)lane] : 0xffffffffu;
}
__device__ __forceinline__ uint4 meta_stage_b(const BatchView &b, uint32_t ri) {
    // ri comes from a completed stage A; out-of-range indices load nothing and are flagged in stage C
    return (uint64_t)ri < b.n_reads ? b.reads[ri] : make_uint4(0u, 0xffffffffu, 0u, 0u);
}
__device__ __forceinline__ PairMeta meta_stage_c(const BatchView &b, const uint4 r, int cnt, int lane, uint32_t &status,
                                                  bool &valid) {
    PairMeta m{0u, 0u, 0u, 0u};
    valid = false;
    if (lane < cnt) {
        const uint64_t n4 = ((uint64_t)r.y + 3u) >> 2;
        if (r.y < 0x10000000u && (uint64_t)r.x + n4 <= b.n_cigar4) {
            m.off4 = r.x;
            m.nc = r.y | ((r.w & (RB_IS_2D << 8)) ? 0x80000000u : 0u);
            m.pos = r.z;
            m.misc = r.w;
            valid = true;
        } else {
            status |= ST_INDEX;
        }
    }
    return m;
}

// ops that consume the reference: M D N = X -> bits 0,2,3,7,8 (src/call.rs:384-392,404)
constexpr uint32_t kConsume = 0x18Du;

// len if the op consumes the reference, else 0 (v_bfe_i32 + v_and)
__device__ __forceinline__ uint32_t ref_advance(uint32_t op, uint32_t len) {
    return len & (uint32_t)__builtin_amdgcn_sbfe((int)kConsume, op, 1u);
}
// Same from the raw packed word, without extracting the op: v_bfe_i32 takes its bit offset from
// the low FIVE bits of the word (op | len&1 << 4), so the 9-entry table is laid down twice.
constexpr uint32_t kConsume32 = kConsume | (kConsume << 16);
__device__ __forceinline__ uint32_t ref_advance_raw(uint32_t w) {
    return (w >> 4) & (uint32_t)__builtin_amdgcn_sbfe((int)kConsume32, w, 1u);
}
// bit 0 of (kBadOp32 >> (w & 31)) is set iff the op code is 9..15 (rust-htslib cigar() panics)
constexpr uint32_t kBadOp32 = 0xFE00FE00u;

// Evaluates the queued window lanes: entry e -> lane e.  src/call.rs:387-403 for 4 ops per lane.
__device__ __forceinline__ void drain_queue(WaveLds &L, uint32_t &qcount, const Window &W, int lane) {
    // single-wave LDS traffic: DS instructions of one wave execute in issue order, so the queue
    // writes above are visible to the reads below without a fence; wave_barrier only pins the
    // compiler's schedule
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < qcount; base += 64u) {
        const uint32_t e = base + (uint32_t)lane;
        if (e < qcount) {
            const u32x4 w = L.q[e].w;
            const uint2 info = L.q[e].info;
            const uint32_t rel = info.x;
            const uint32_t slot = info.y & 63u;
            // soft clips of an accidental-2D read never count (src/call.rs:394): drop S from the candidates
            const uint32_t cand_mask = (info.y & 64u) ? 0x06u : 0x16u;  // I=1, D=2, S=4
            const uint32_t op0 = w.x & 15u, op1 = w.y & 15u, op2 = w.z & 15u, op3 = w.w & 15u;
            const uint32_t l0 = w.x >> 4, l1 = w.y >> 4, l2 = w.z >> 4, l3 = w.w >> 4;
            const uint32_t e1 = ref_advance(op0, l0), e2 = e1 + ref_advance(op1, l1), e3 = e2 + ref_advance(op2, l2);
            int32_t s = 0;
            uint32_t clip = 0;
#define INQ_OP(op, len, ee_)                                                                              \
    {                                                                                                     \
        const bool hit = ((cand_mask >> (op)) & 1u) && (len) > W.minlen && (rel + (ee_)) < W.width;        \
        const int32_t v = ((op) == 2u) ? -(int32_t)(len) : (int32_t)(len);                                \
        s += hit ? v : 0;                                                                                 \
        clip |= (hit && (op) == 4u) ? 1u : 0u;                                                            \
    }
            INQ_OP(op0, l0, 0u)
            INQ_OP(op1, l1, e1)
            INQ_OP(op2, l2, e2)
            INQ_OP(op3, l3, e3)
#undef INQ_OP
            if (s != 0) atomicAdd(&L.acc[slot], (unsigned long long)(long long)s);  // |s| < 4 * 2^28
            if (clip) atomicOr(&L.flags[slot], 1u);
        }
    }
    qcount = 0;
    __builtin_amdgcn_wave_barrier();
}

// Walks the reads of pairs [0, cnt) described by `m` (lane k owns pair k).  On return lane k holds
// the pair's Call (src/call.rs:67-71) in `val` and PM_CLIP | PM_FETCHED | PM_KEPT | group in `meta`.
template <bool UNPHASED, int AUX>
__device__ __forceinline__ void walk_pairs(const BatchView &b, const PairMeta &m, bool valid, int cnt,
                                           const Window &W, int lane, uint32_t &status, WaveLds &L, int64_t &val,
                                           uint32_t &meta) {
    L.acc[lane] = 0ull;
    L.flags[lane] = 0u;
    uint32_t qcount = 0;
    uint32_t lane_range = 0, lane_bad = 0;
    uint32_t end_carry = 0;  // lane k: reference_position after the last op of read k

    // ---- load cursor: runs 4 chunk loads ahead of the compute cursor ----
    int hk = 0;
    uint32_t hc = 0, h_nchunks = 1, h_off4 = 0, h_n4 = 0;
    auto head_load = [&]() {
        if (hk < cnt) {
            h_off4 = readlane_u32(m.off4, hk);
            const uint32_t nc = readlane_u32(m.nc, hk) & 0x7fffffffu;
            h_n4 = (nc + 3u) >> 2;
            h_nchunks = max(1u, (nc + 255u) >> 8);
        } else {
            h_off4 = 0;
            h_n4 = 0;
            h_nchunks = 1;
        }
    };
    auto issue = [&]() -> u32x4 {
        // raw buffer load: offsets at or beyond num_records return 0, no exec masking needed
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(b.cigar4 + h_off4), (short)0, (int)(h_n4 * 16u), 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((hc * 64u + (uint32_t)lane) * 16u), 0, AUX);
        ++hc;
        if (hc >= h_nchunks) {
            ++hk;
            hc = 0;
            head_load();
        }
        return v;
    };
    head_load();
    u32x4 qa = issue(), qb = issue(), qc = issue(), qd = issue();

    // ---- compute cursor ----
    int tk = 0;
    uint32_t tc = 0, t_nchunks = 1, t_nc = 0, t_info = 0, carry = 0;
    auto tail_load = [&]() {
        if (tk < cnt) {
            const uint32_t ncp = readlane_u32(m.nc, tk);
            t_nc = ncp & 0x7fffffffu;
            t_nchunks = max(1u, (t_nc + 255u) >> 8);
            carry = readlane_u32(m.pos, tk) + 1u;  // (reference_start + 1) as u32, src/call.rs:380
            t_info = (uint32_t)tk | ((ncp >> 31) << 6);
            lane_range |= carry;
        }
    };
    tail_load();

    auto step = [&](const u32x4 w) {
        const uint32_t e1 = ref_advance_raw(w.x);
        const uint32_t e2 = e1 + ref_advance_raw(w.y);
        const uint32_t e3 = e2 + ref_advance_raw(w.z);
        const uint32_t tot = e3 + ref_advance_raw(w.w);
        const uint32_t incl = wave_inclusive_scan_u32(tot);
        lane_bad |= (kBadOp32 >> (w.x & 31u)) | (kBadOp32 >> (w.y & 31u)) | (kBadOp32 >> (w.z & 31u)) |
                    (kBadOp32 >> (w.w & 31u));
        lane_range |= carry + incl;
        // x = position after this lane's ops, relative to start_ext + 1.  One of the lane's ops can start
        // inside the window only if 0 <= x and x - tot < width  <=>  x <u width + tot  (all < 2^31 inside
        // the parity domain).  Zero-filled lanes behind a read that ends inside the window pass the test too;
        // they queue four `0M` that contribute nothing, which is cheaper than a second compare on every chunk.
        const uint32_t x = (carry - W.se1) + incl;
        const bool inw = x < W.width + tot;
        const uint64_t mask = ballot64(inw);
        if (mask) {
            QueueEntry *const tail = &L.q[qcount];  // wave-uniform
            if (inw) {
                const uint32_t idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                tail[idx].w = w;
                tail[idx].info = make_uint2(x - tot, t_info);
            }
            qcount += (uint32_t)__popcll(mask);
            if (qcount > 64u) drain_queue(L, qcount, W, lane);
        }
        carry += readlane_u32(incl, 63);
        ++tc;
        if (tc >= t_nchunks) {
            end_carry = writelane_u32(end_carry, carry, tk);  // lane tk <- carry
            ++tk;
            tc = 0;
            tail_load();
        }
    };

    // four named buffers, no register rotation: each step waits only for its own load (vmcnt(3))
    while (tk < cnt) {
        step(qa);
        qa = issue();
        if (tk < cnt) step(qb);
        qb = issue();
        if (tk < cnt) step(qc);
        qc = issue();
        if (tk < cnt) step(qd);
        qd = issue();
    }
    if (qcount) drain_queue(L, qcount, W, lane);
    if (ballot64((lane_bad & 1u) != 0u)) status |= ST_CIGAR_OP;  // rust-htslib cigar() would panic
    if (ballot64((lane_range >> 31) != 0u)) status |= ST_RANGE;

    // ---- per-read epilogue, one read per lane ----
    val = 0;
    meta = 0;
    if (lane < cnt && valid) {
        const uint32_t pos = m.pos;
        const uint32_t mapq = m.misc & 0xffu, bits = (m.misc >> 8) & 0xffu, phase = (m.misc >> 16) & 0xffu;
        // [3P] bam_endpos: rlen = unmapped ? 0 : sum(ref-consuming); rlen == 0 -> 1
        uint32_t rlen = end_carry - (pos + 1u);
        if ((bits & RB_UNMAPPED) || rlen == 0u) rlen = 1u;
        const uint32_t rend = pos + rlen;  // reference_end() as u32
        // fetch(): pos < end_ext && endpos > start_ext (signed pos, pos >= -1 inside the domain)
        const bool fetched = ((int32_t)pos < 0 || pos < W.ee) && rend > W.se;
        bool skip;
        if (UNPHASED)
            skip = W.se < pos || rend < W.ee || mapq <= 10u;  // src/call.rs:297-302
        else
            skip = !(bits & RB_HAS_HP) || (W.se < pos && rend < W.ee) || mapq <= 10u;  // :349-355
        const bool kept = fetched && !skip;
        uint32_t grp = 0u;
        bool bad_phase = false;
        if (!UNPHASED && kept) {
            if (phase > 2u)
                bad_phase = true;  // calls.get_mut(&phase).unwrap() panics, src/call.rs:358
            else
                grp = phase;
        }
        if (bad_phase) status |= ST_PHASE;
        // is_accidental_2d panics on this read's SA, but the reference only calls it from call_from_cigar,
        // i.e. for reads that passed the filter (src/call.rs:303,357 -> :394)
        if (kept && (bits & RB_SA_PANIC)) status |= ST_AUX;
        val = (int64_t)L.acc[lane];
        meta = ((L.flags[lane] & 1u) ? PM_CLIP : 0u) | (fetched ? PM_FETCHED : 0u) | (kept ? PM_KEPT : 0u) |
               (grp << PM_GRP_SHIFT);
    }
}

}  // namespace inq
