// kernels.h — kernel argument block shared by kernels.hip and capi.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace inq {

// Device-resident status block (one per ctx).
constexpr int kListShards = 32;
// work lists: 0 = 65 .. 256 offered reads, 1 = 257 .. kWalkSplit, 2 = deeper (walked by the whole grid)
constexpr int kListKinds = 3;
constexpr uint32_t kWalkSplit = 16384;
constexpr uint32_t kReduceInPlace = 2048;  // a listed locus of up to this many reads is reduced by the workgroup that walked it (locus_call_mid_walk)

struct DevStatus {
    unsigned int err;           // ST_* bits OR-ed by the kernels
    unsigned int pad;
    // work-list lengths [kind: 0 = medium (65..256 reads), 1 = deep, 2 = deeper than kWalkSplit][shard], zero between launch sequences
    // (the last kernel of a sequence that may have filled them clears them).  Every counter
    // sits on its own 128-byte line: returning atomics on one line serialise at ~11 ns each.
    struct alignas(128) Counter {
        unsigned int n;
    } list_count[kListKinds][kListShards];
    unsigned long long ties;    // unphased loci whose split cuts mixed Span/Clip ties
    // locus_call_tail (deep_select.hip), each word on a line of its own: the grid barrier's arrival counter (monotonic inside a launch),
    // its abort word (a barrier that waited too long: every workgroup leaves), the exit ticket (the last workgroup out empties the
    // work lists).  locus_call_small puts all three back to zero in front of every sequence.
    struct alignas(128) Word {
        unsigned int v;
    } bar_count, bar_abort, exit_ticket;
};

struct KArgs {
    // batch (device pointers)
    const uint4 *cigar4;
    const uint4 *reads;
    const uint32_t *pair_read;
    const uint64_t *locus_pair_off;
    const uint32_t *locus_start;
    const uint32_t *locus_end;
    uint64_t n_reads, n_cigar4, n_pairs, n_loci;
    uint32_t minlen, support;
    // results
    double *phase1, *phase2;
    int64_t *pair_call;  // may be null
    uint8_t *pair_bits;  // may be null
    // ctx scratch
    DevStatus *status;
    uint32_t *worklist;  // [kind][kListShards][shard_cap]
    int64_t *sval;       // [n_pairs]
    uint8_t *smeta;      // [n_pairs]
    uint32_t blocks_per_xcd;  // grid_small / 8
    uint32_t shard_cap;       // loci one shard can list: every locus whose block has blockIdx % kListShards == shard
    uint32_t max_reads_hint;  // caller's promise (0 = none): no locus is offered more reads than this
};

// deep_scratch: deep_select_scratch_bytes(n_pairs) of ctx scratch for the loci the grid-wide select takes (may be null when the
// depth hint rules them out, or no locus can be that deep); grid_tail: workgroups of the persistent locus_call_tail - they wait
// for one another at grid barriers, so all of them must be resident at once: at most one per compute unit of the device
void launch_locus_call(const KArgs &a, bool unphased, bool nt_loads, uint32_t grid_small, uint32_t grid_medium,
                       uint32_t grid_tail, hipStream_t s, hipEvent_t ev_mid, void *deep_scratch);

// deep_select.hip: loci with more offered reads than this are reduced by the whole grid instead of one workgroup
constexpr uint32_t kGridSelectMin = 65536;
size_t deep_select_scratch_bytes(uint64_t n_pairs);
void launch_locus_tail(const KArgs &k, bool unphased, void *scratch, uint64_t n_pairs, uint32_t grid, hipStream_t s);

}  // namespace inq
