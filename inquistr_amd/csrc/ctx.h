// ctx.h — the library context behind inq_ctx_t, shared by the translation units that implement the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/inquistr_hip.h"
#include "kernels.h"

namespace inq {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct EvTriple {
    hipEvent_t e0, e1, e2;
};

struct SpanState;  // device front end (span.hip)

}  // namespace inq

struct inq_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string backend, last_err;
    inq::DevStatus *d_status = nullptr;
    inq::DevStatus *h_status = nullptr;  // pinned mirror for the host-buffer entry
    inq::DevBuf worklist, sval, smeta, deep;
    // staging for the host-buffer entry
    inq::DevBuf cigar, reads, pair_read, off, lstart, lend, p1, p2, pcall, pbits;
    inq::DevBuf ovalues, olen, oflags, okeep, otrans;  // inq_outlier_rows
    uint32_t n_cus = 0;        // compute units of the device
    uint32_t grid_tail = 256;  // workgroups of the persistent locus_call_tail: they meet at grid barriers, so never more than n_cus
    uint32_t grid_medium = 4096;  // (8 192: + 3 us for the launch that finds the lists empty, 0 - 4 % quicker where they are full)
    uint32_t max_reads_hint = 0;  // 0 = unknown; else the caller's bound on reads per locus
    uint32_t call_hint = 0;       // set by the host-buffer entry, which sees the offsets, for its own launch
    int nt_loads = -1;  // -1 auto: non-temporal when no read is shared between loci
    bool timing = false;
    bool verify_crc = true;  // device front end: check inflated blocks against their CRC32
    // workgroup inflate: the counting passes leave the symbols behind for the commit (option "inflate_tokens").  Round 3 switched it on
    // for match-heavy spans (+3 ... 4 % on CIGAR-only records); the counters then showed what it moves: 32 KB of scratch per block
    // written in EVERY counting pass, 7.5 GB of HBM traffic for 1.79 GB of algorithmic bytes on the 501 MB file (profiles/r03_front/).
    // With the inflate of span k + 1 now running beside the scan / gather / locus kernels of span k that traffic is no longer free:
    // off unless asked for (1 = on, -1 = on where the sampled block headers say match-heavy, 0 = off)
    int inflate_tokens = 0;
    int inflate_lit_pairs = -1;  // workgroup inflate's symbol loop: 1 = a second literal from the same peek, 0 = not, -1 = by the data (deflate_probe.h)
    // a staged span (inq_span_stage_begin) is inflated right behind its upload, on a stream of its own: the inflate of span k + 1 runs
    // beside span k's record scan, gather and join.  A staging slot then holds an inflated buffer of its own.  Round 3 left this off
    // for "+0.3 s for a one-file process"; round 4 found that figure to be the driver's teardown of the PREVIOUS process in the
    // timing loop, not the allocations (hipMalloc: 10 - 200 us whatever the size): on
    int inflate_ahead = 1;
    int outlier_tile = 1;   // z-score of rows of <= 256 values through an LDS tile (0: the transposed-copy kernel for every width)
    int blocking_sync = 0;  // waits give the core back (hipDeviceScheduleBlockingSync, blocking events): for a caller short of cores
    // the gather's stores bypass the caches (bam_scan.hip): the batch it builds is read by a later launch, not by this one; the locus
    // kernels behind it run at 5.4 - 6.2 instead of 4.9 - 5.3 TB/s of algorithmic bytes (profiles/r04_results/locus_kernels_in_the_cli.txt)
    bool gather_nt = true;
    uint64_t batch_loci_hint = 0;  // inq_call_span_deferred: loci the caller lets a batch collect before it flushes (0 = no word)
    uint32_t inflate_algo = 2;  // 0 = workgroup per BGZF block, 1 = lane per block, 2 = the quicker one (0 since round 2)
    // Buffers that were outgrown.  Growing one used to mean hipDeviceSynchronize + hipFree + hipMalloc on the spot; both calls wait for
    // EVERY stream of the device - in the span loop that is the 5 ms upload of the next span on the copy stream, ten times per file
    // (profiles/r04_results/span_loop_growth_stalls.txt).  hipMalloc alone costs 10 - 200 us whatever the size (tools/alloc_probe.hip),
    // so the old buffer is parked here - earlier launches may still read it - and given back at a point where the device is idle
    // anyway (the end of a flush / of a host-buffer call, the destruction of the ctx) or when more than 16 GB wait.
    std::mutex retired_mu;  // the uploader thread (inq_span_stage) grows its slots while the caller grows the scan buffers
    std::vector<std::pair<void *, size_t>> retired;
    size_t retired_bytes = 0;
    size_t retired_limit = 16ull << 30;  // option "retired_limit_mb"
    // an allocation that fails for lack of memory gives the parked buffers back and tries again (capi.hip device_alloc)
    uint32_t test_fail_allocs = 0;  // option "test_fail_allocs": the next N allocations fail at their first attempt
    uint64_t alloc_retries = 0;     // how often that second attempt was needed (inq_ctx_alloc_retries)
    std::vector<inq::EvTriple> ev_pool;
    size_t ev_used = 0;
    inq::SpanState *span = nullptr;  // created on first use by the device front end
};

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (ctx)->last_err = std::string(#expr) + ": " + hipGetErrorString(_e);               \
            return _e == hipErrorOutOfMemory ? INQ_ERR_NOMEM : INQ_ERR_HIP;                    \
        }                                                                                      \
    } while (0)

namespace inq {
// The one place the library could read an experiment's switch from the environment - and it does only when built with
// -DINQ_DEBUG_ENV (make DEBUG_ENV=1).  What a host can set is inq_ctx_set_option / inq_default_option, nothing else; INQ_TIMING
// (stage clocks on stderr) is the only variable the shipped library looks at.
#ifdef INQ_DEBUG_ENV
inline const char *debug_env(const char *name) { return std::getenv(name); }
#else
inline const char *debug_env(const char *) { return nullptr; }
#endif
// grows b to at least `bytes` (with headroom); the old buffer is retired, not freed: no synchronisation, contents NOT kept
int ensure(inq_ctx *c, DevBuf &b, size_t bytes);
int device_alloc(inq_ctx *c, void **out, size_t want, size_t exact, size_t *got);
void retire(inq_ctx *c, void *p, size_t bytes);
void purge_retired(inq_ctx *c);  // hipFree of everything retired (waits for the device: call where it is idle)
// enqueue-only launch sequence of the locus kernels over a device-resident batch
int call_batch_device_impl(inq_ctx *c, const inq_batch_t *b, inq_result_t *r, void *hip_stream);
int status_to_code(uint32_t st);
void span_state_destroy(SpanState *s);
int span_state_init(inq_ctx *c);       // device front end state, the part staging needs (streams, slots); called by inq_ctx_create
int span_state_init_rest(inq_ctx *c);  // ... and the part the calls need
void preload_locus(hipStream_t s);
void preload_inflate(hipStream_t s);
void preload_scan(hipStream_t s);
double span_last_inflate_ms(SpanState *s);  // kernel time of the last inflate launch, < 0 if none
}  // namespace inq
