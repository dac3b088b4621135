// capi.hip — implementation of the C ABI in include/inquistr_hip.h on top of kernels.hip.
// HIP only: there is no CPU fallback; without a gfx950 device every entry fails loudly.
#include <hip/hip_runtime.h>

#include <mutex>

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/inquistr_hip.h"
#include "cigar_walk.h"
#include "ctx.h"
#include "front_kernels.h"
#include "kernels.h"

using namespace inq;

namespace inq {

void retire(inq_ctx *c, void *p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> g(c->retired_mu);
    c->retired.emplace_back(p, bytes);
    c->retired_bytes += bytes;
}

void purge_retired(inq_ctx *c) {
    std::vector<std::pair<void *, size_t>> gone;
    {
        std::lock_guard<std::mutex> g(c->retired_mu);
        gone.swap(c->retired);
        c->retired_bytes = 0;
    }
    for (auto &e : gone) (void)hipFree(e.first);  // hipFree waits for the device: work that still reads the buffer ends first
}

// hipMalloc that gives the parked buffers back before it gives up: retired buffers (ctx.h) are memory the context could free at any
// moment, so "out of memory" with some of them waiting is not yet out of memory.  `exact` (< want, or 0): what the caller really
// needs, tried last without the headroom.  ("test_fail_allocs" = N makes the next N first attempts fail: the retry's test seam.)
int device_alloc(inq_ctx *c, void **out, size_t want, size_t exact, size_t *got) {
    *out = nullptr;
    *got = want;
    hipError_t e = hipSuccess;
    if (c->test_fail_allocs > 0) {
        --c->test_fail_allocs;
        ++c->alloc_retries;
        e = hipErrorOutOfMemory;
    } else {
        e = hipMalloc(out, want);
    }
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();  // clear the sticky error
        purge_retired(c);         // waits for the device: whatever still read those buffers is through
        ++c->alloc_retries;
        e = hipMalloc(out, want);
        if (e == hipErrorOutOfMemory && exact && exact < want) {
            (void)hipGetLastError();
            e = hipMalloc(out, exact);
            *got = exact;
        }
    }
    if (e != hipSuccess) {
        *out = nullptr;
        c->last_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? INQ_ERR_NOMEM : INQ_ERR_HIP;
    }
    return INQ_OK;
}

int ensure(inq_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return INQ_OK;
    if (b.p) {
        // earlier launches (possibly on the caller's stream) may still use the old buffer: it is parked, not freed (ctx.h)
        retire(c, b.p, b.cap);
        b.p = nullptr;
        b.cap = 0;
        bool too_much;
        {
            std::lock_guard<std::mutex> g(c->retired_mu);
            too_much = c->retired_bytes > c->retired_limit;
        }
        if (too_much) purge_retired(c);
    }
    // grow with headroom: batches of a sweep vary in size, reallocating for each new maximum would put a malloc in front of most
    // calls.  Large buffers get little of it: device memory a process holds is wiped by the driver when the process leaves, at
    // ~20 GB/s, and while that goes on the NEXT process's hipInit and hipMalloc wait (0.1 -> 0.3 s of start-up, 50 ms per GB
    // allocated: profiles/r04_results/back_to_back_vs_paused_processes.txt) - spans of a file are the same size to a percent
    const size_t floor_bytes = bytes < 256 ? 256 : bytes;
    size_t want = bytes < 256 ? 256 : bytes < (32u << 20) ? bytes + bytes / 2 + (1u << 16) : bytes + bytes / 16 + (1u << 20);
    size_t got = 0;
    const int rc = device_alloc(c, &b.p, want, floor_bytes, &got);
    if (rc != INQ_OK) return rc;
    b.cap = got;
    return INQ_OK;
}

int status_to_code(uint32_t st) {
    if (st & ST_HINT) return INQ_ERR_ARG;
    if (st & ST_LOCUS) return INQ_ERR_LOCUS;
    if (st & ST_INDEX) return INQ_ERR_INDEX;
    if (st & ST_CIGAR_OP) return INQ_ERR_CIGAR_OP;
    if (st & ST_RANGE) return INQ_ERR_RANGE;
    if (st & ST_PHASE) return INQ_ERR_PHASE;
    if (st & ST_AUX) return INQ_ERR_AUX;
    if (st & ST_INTERNAL) return INQ_ERR_HIP;  // a grid barrier gave up waiting (the grid drained; the rows of very deep loci are not there)
    return INQ_OK;
}

}  // namespace inq

extern "C" {

int inq_abi_version(void) { return INQ_ABI_VERSION; }

const char *inq_strerror(int code) {
    switch (code) {
    case INQ_OK: return "ok";
    case INQ_ERR_ARG: return "invalid argument (null pointer, inconsistent sizes or non-monotone offsets)";
    case INQ_ERR_SUPPORT_ZERO: return "support must be >= 1";
    case INQ_ERR_PHASE: return "a read passing the phased filter has HP outside {0,1,2}";
    case INQ_ERR_CIGAR_OP: return "CIGAR op code above 8";
    case INQ_ERR_LOCUS: return "locus start < 10 or end < start";
    case INQ_ERR_RANGE: return "read position plus reference span does not fit 31 bits";
    case INQ_ERR_INDEX: return "pair or CIGAR index outside the batch buffers";
    case INQ_ERR_HIP: return "HIP runtime error";
    case INQ_ERR_NOMEM: return "out of device memory";
    case INQ_ERR_INFLATE: return "a BGZF block does not inflate to its recorded size (corrupt or truncated BAM)";
    case INQ_ERR_BAM: return "corrupt BAM record chain, or records not coordinate-sorted";
    case INQ_ERR_AUX: return "HP aux of a fetched read is neither C nor i, or the SA aux of a kept read with a soft clip cannot be parsed (the reference panics)";
    case INQ_ERR_NO_DEVICE: return "no gfx950 (MI355X) device available; this library has no CPU fallback";
    default: return "unknown error";
    }
}

static void apply_default_options(inq_ctx *c);
static bool default_option(const char *key, int64_t *value);

// The context in two steps for a caller that is in a hurry: *out is set and *stage_ready raised as soon as the staging entry points
// (inq_span_stage_begin / _wait / inq_span_stage) may be used - the runtime is up, the copy and inflate streams and the staging
// slots exist - while the rest (status buffers, code objects: ~30 ms) is still being made; a process whose spans are already in
// memory starts uploading that much earlier.  Everything else may be called once the function has returned INQ_OK.  Whatever it
// returns, a context it has published stays valid until inq_ctx_destroy (the caller destroys it, also after a failure).
int inq_ctx_create_early(int device_id, inq_ctx_t **out, volatile int *stage_ready) {
    if (!out) return INQ_ERR_ARG;
    *out = nullptr;
    if (stage_ready) *stage_ready = 0;
    int n = 0;
    // INQ_TIMING=2: where the start-up goes (the runtime's own initialisation is most of a short run)
    const char *tenv = std::getenv("INQ_TIMING");
    const bool verbose = tenv && tenv[0] == '2';
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[inq ctx] %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return INQ_ERR_NO_DEVICE;
    lap("hipGetDeviceCount (runtime init)");
    if (device_id < 0 || device_id >= n) return INQ_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return INQ_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return INQ_ERR_NO_DEVICE;
    inq_ctx *c = new (std::nothrow) inq_ctx();
    if (!c) return INQ_ERR_NOMEM;
    c->device = device_id;
    c->n_cus = (uint32_t)std::max(prop.multiProcessorCount, 1);
    c->grid_tail = std::min<uint32_t>(c->grid_tail, c->n_cus);  // a partitioned device (CPX: 32 CUs) takes a smaller grid
    c->backend = std::string("hip:") + prop.gcnArchName + ":" + prop.name;
    bool published = false;
    auto fail = [&](int code) {
        if (!published) inq_ctx_destroy(c);  // (a published context is the caller's to destroy: another thread may be using it)
        return code;
    };
    lap("device properties");
    {   // "blocking_sync" has to be known before the first stream and event are made
        int64_t v = 0;
        if (default_option("blocking_sync", &v) && v) {
            c->blocking_sync = 1;
            (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);  // (an error here - flags fixed by an earlier user of the device - is not ours to report)
            (void)hipGetLastError();
        }
    }
    if (hipSetDevice(device_id) != hipSuccess) return fail(INQ_ERR_HIP);
    // the streams of the staging path first (the process's first stream costs 18 - 170 ms, every further one 8): uploads and the
    // inflates behind them start while the rest of the context is still being made
    if (span_state_init(c) != INQ_OK) return fail(INQ_ERR_HIP);
    apply_default_options(c);
    lap("hipSetDevice + copy / inflate streams, slots");
    *out = c;
    if (stage_ready) {
        published = true;
        __atomic_store_n(stage_ready, 1, __ATOMIC_RELEASE);
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(INQ_ERR_HIP);
    if (span_state_init_rest(c) != INQ_OK) return fail(INQ_ERR_HIP);
    lap("main stream, span state");
    if (hipMalloc((void **)&c->d_status, sizeof(DevStatus)) != hipSuccess) return fail(INQ_ERR_NOMEM);
    if (hipMemset(c->d_status, 0, sizeof(DevStatus)) != hipSuccess) return fail(INQ_ERR_HIP);
    if (hipHostMalloc((void **)&c->h_status, sizeof(DevStatus), hipHostMallocDefault) != hipSuccess) return fail(INQ_ERR_NOMEM);
    lap("status buffers");
    // load the code objects while the caller is still busy opening its input
    preload_locus(c->stream);
    preload_inflate(c->stream);
    preload_scan(c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(INQ_ERR_HIP);
    lap("code objects (3 empty launches)");
    return INQ_OK;
}

int inq_ctx_create(int device_id, inq_ctx_t **out) {
    const int rc = inq_ctx_create_early(device_id, out, nullptr);
    if (rc != INQ_OK && out) *out = nullptr;  // (nothing was published: the context is gone already)
    return rc;
}

int inq_ctx_create_multi(const int *device_ids, int n, inq_ctx_t **ctxs) {
    if (!device_ids || !ctxs || n <= 0 || n > 64) return INQ_ERR_ARG;
    for (int i = 0; i < n; ++i) ctxs[i] = nullptr;
    try {
        std::vector<int> rc((size_t)n, INQ_OK);
        std::vector<std::thread> th;
        for (int i = 0; i < n; ++i) th.emplace_back([&, i] { rc[(size_t)i] = inq_ctx_create(device_ids[i], &ctxs[i]); });
        for (auto &t : th) t.join();
        for (int i = 0; i < n; ++i)
            if (rc[(size_t)i] != INQ_OK) {
                for (int k = 0; k < n; ++k) {
                    inq_ctx_destroy(ctxs[k]);
                    ctxs[k] = nullptr;
                }
                return rc[(size_t)i];
            }
        return INQ_OK;
    } catch (...) {
        for (int k = 0; k < n; ++k) {
            inq_ctx_destroy(ctxs[k]);
            ctxs[k] = nullptr;
        }
        return INQ_ERR_NOMEM;
    }
}

void inq_ctx_destroy(inq_ctx_t *c) {
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->span) span_state_destroy(c->span);
    purge_retired(c);
    for (DevBuf *b : {&c->worklist, &c->sval, &c->smeta, &c->deep, &c->cigar, &c->reads, &c->pair_read, &c->off, &c->lstart,
                      &c->lend, &c->p1, &c->p2, &c->pcall, &c->pbits, &c->ovalues, &c->olen, &c->oflags, &c->okeep, &c->otrans})
        if (b->p) (void)hipFree(b->p);
    for (auto &e : c->ev_pool) {
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
        (void)hipEventDestroy(e.e2);
    }
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int inq_ctx_numa_node(const inq_ctx_t *c) {
    if (!c) return -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) != hipSuccess) return -1;
    for (char *q = bus; *q; ++q)
        if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');  // sysfs spells the address in lower case
    char path[160];
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = std::fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

const char *inq_backend_name(const inq_ctx_t *c) { return c ? c->backend.c_str() : "none"; }
const char *inq_last_error(const inq_ctx_t *c) { return c ? c->last_err.c_str() : ""; }

static int check_scalars(const inq_batch_t *b, const inq_result_t *r) {
    if (!b || !r) return INQ_ERR_ARG;
    if (b->n_loci && (!r->phase1 || !r->phase2)) return INQ_ERR_ARG;
    if (b->n_loci && (!b->locus_pair_off || !b->locus_start || !b->locus_end)) return INQ_ERR_ARG;
    if (b->n_pairs && (!b->pair_read || !b->reads || !b->n_reads)) return INQ_ERR_ARG;
    if (b->n_cigar_words && !b->cigar) return INQ_ERR_ARG;
    if (b->n_cigar_words % 4 != 0 || b->reserved != 0 || b->unphased > 1) return INQ_ERR_ARG;
    if (!b->n_loci && b->n_pairs) return INQ_ERR_ARG;
    if (b->n_loci >= 0xfffffff0ull || b->n_pairs >= (1ull << 40)) return INQ_ERR_ARG;
    if (b->support == 0) return INQ_ERR_SUPPORT_ZERO;
    return INQ_OK;
}

static int enqueue_batch(inq_ctx_t *c, const inq_batch_t *b, inq_result_t *r, void *hip_stream) {
    if (!c) return INQ_ERR_ARG;
    int rc = check_scalars(b, r);
    if (rc != INQ_OK) return rc;
    if (((uintptr_t)b->cigar & 15u) || ((uintptr_t)b->reads & 15u)) return INQ_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const uint64_t blocks = (b->n_loci + 3) / 4;
    const uint32_t per_xcd = (uint32_t)((blocks + 7) / 8);
    const uint32_t grid_small = per_xcd * 8u;
    const uint32_t shard_cap = (grid_small / kListShards + 1u) * 4u;
    if ((rc = ensure(c, c->worklist, (size_t)shard_cap * kListShards * kListKinds * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, c->sval, (size_t)b->n_pairs * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, c->smeta, (size_t)b->n_pairs)) != INQ_OK) return rc;
    // loci of more than kGridSelectMin reads are reduced over the whole grid (deep_select.hip): a state of 74 KB each, and there
    // cannot be more of them than n_pairs / kGridSelectMin; nothing is allocated or launched when the depth hint rules them out
    const uint32_t hint0 = c->call_hint ? c->call_hint : c->max_reads_hint;
    const bool deep_possible = b->n_pairs > kGridSelectMin && !(hint0 && hint0 <= kGridSelectMin);
    if (deep_possible && (rc = ensure(c, c->deep, deep_select_scratch_bytes(b->n_pairs))) != INQ_OK) return rc;

    KArgs a;
    a.cigar4 = (const uint4 *)b->cigar;
    a.reads = (const uint4 *)b->reads;
    a.pair_read = b->pair_read;
    a.locus_pair_off = b->locus_pair_off;
    a.locus_start = b->locus_start;
    a.locus_end = b->locus_end;
    a.n_reads = b->n_reads;
    a.n_cigar4 = b->n_cigar_words / 4;
    a.n_pairs = b->n_pairs;
    a.n_loci = b->n_loci;
    a.minlen = b->minlen;
    a.support = b->support;
    a.phase1 = r->phase1;
    a.phase2 = r->phase2;
    a.pair_call = r->pair_call;
    a.pair_bits = r->pair_bits;
    a.status = c->d_status;
    a.worklist = (uint32_t *)c->worklist.p;
    a.sval = (int64_t *)c->sval.p;
    a.smeta = (uint8_t *)c->smeta.p;
    a.blocks_per_xcd = per_xcd;
    a.shard_cap = shard_cap;
    const uint32_t hint = c->call_hint ? c->call_hint : c->max_reads_hint;
    c->call_hint = 0;
    a.max_reads_hint = hint;

    // CIGAR words of a read referenced by one locus only are read exactly once: stream them past the
    // caches (nt).  Reads shared by neighbouring loci keep the default policy so the second locus hits L2.
    const bool nt = c->nt_loads < 0 ? (b->n_pairs <= b->n_reads) : (c->nt_loads != 0);
    EvTriple *ev = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev_pool.size()) {
            EvTriple t;
            HIP_TRY(c, hipEventCreate(&t.e0));
            HIP_TRY(c, hipEventCreate(&t.e1));
            HIP_TRY(c, hipEventCreate(&t.e2));
            c->ev_pool.push_back(t);
        }
        ev = &c->ev_pool[c->ev_used++];
        HIP_TRY(c, hipEventRecord(ev->e0, s));
    }
    launch_locus_call(a, b->unphased != 0, nt, grid_small, c->grid_medium, c->grid_tail, s, ev ? ev->e1 : nullptr, deep_possible ? c->deep.p : nullptr);
    HIP_TRY(c, hipGetLastError());
    if (ev) HIP_TRY(c, hipEventRecord(ev->e2, s));
    return INQ_OK;
}

int inq_call_batch_device(inq_ctx_t *c, const inq_batch_t *b, inq_result_t *r, void *hip_stream) {
    try {  // nothing may unwind across the C ABI
        return enqueue_batch(c, b, r, hip_stream);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_ctx_status(inq_ctx_t *c, uint64_t *n_tie_loci) {
    if (!c) return INQ_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    DevStatus h;
    HIP_TRY(c, hipMemcpy(&h, c->d_status, sizeof h, hipMemcpyDeviceToHost));
    if (n_tie_loci) *n_tie_loci = h.ties;
    // clear err and ties, keep the work-list counters
    HIP_TRY(c, hipMemsetAsync(&c->d_status->err, 0, sizeof(unsigned int), c->stream));
    HIP_TRY(c, hipMemsetAsync(&c->d_status->ties, 0, sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return status_to_code(h.err);
}

static int call_batch_impl(inq_ctx_t *c, const inq_batch_t *b, inq_result_t *r) {
    if (!c) return INQ_ERR_ARG;
    int rc = check_scalars(b, r);
    if (rc != INQ_OK) return rc;
    // host-side shape checks: everything the grid and the kernels' indexing assume
    uint64_t max_reads = 0;
    if (b->n_loci) {
        if (b->locus_pair_off[0] != 0 || b->locus_pair_off[b->n_loci] != b->n_pairs) return INQ_ERR_ARG;
        for (uint64_t j = 0; j < b->n_loci; ++j) {
            if (b->locus_pair_off[j] > b->locus_pair_off[j + 1]) return INQ_ERR_ARG;
            if (b->locus_start[j] < 10 || b->locus_end[j] < b->locus_start[j]) return INQ_ERR_LOCUS;
            max_reads = std::max<uint64_t>(max_reads, b->locus_pair_off[j + 1] - b->locus_pair_off[j]);
        }
    }
    r->n_tie_loci = 0;
    if (b->n_loci == 0) return INQ_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    struct Up {
        DevBuf *d;
        const void *h;
        size_t bytes;
    } ups[] = {
        {&c->cigar, b->cigar, (size_t)b->n_cigar_words * 4},
        {&c->reads, b->reads, (size_t)b->n_reads * sizeof(inq_read_t)},
        {&c->pair_read, b->pair_read, (size_t)b->n_pairs * 4},
        {&c->off, b->locus_pair_off, (size_t)(b->n_loci + 1) * 8},
        {&c->lstart, b->locus_start, (size_t)b->n_loci * 4},
        {&c->lend, b->locus_end, (size_t)b->n_loci * 4},
    };
    for (auto &u : ups) {
        if ((rc = ensure(c, *u.d, u.bytes)) != INQ_OK) return rc;
        if (u.bytes) HIP_TRY(c, hipMemcpyAsync(u.d->p, u.h, u.bytes, hipMemcpyHostToDevice, s));
    }
    if ((rc = ensure(c, c->p1, (size_t)b->n_loci * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, c->p2, (size_t)b->n_loci * 8)) != INQ_OK) return rc;
    if (r->pair_call && (rc = ensure(c, c->pcall, (size_t)b->n_pairs * 8)) != INQ_OK) return rc;
    if (r->pair_bits && (rc = ensure(c, c->pbits, (size_t)b->n_pairs)) != INQ_OK) return rc;

    inq_batch_t db = *b;
    db.cigar = (const uint32_t *)c->cigar.p;
    db.reads = (const inq_read_t *)c->reads.p;
    db.pair_read = (const uint32_t *)c->pair_read.p;
    db.locus_pair_off = (const uint64_t *)c->off.p;
    db.locus_start = (const uint32_t *)c->lstart.p;
    db.locus_end = (const uint32_t *)c->lend.p;
    inq_result_t dr;
    dr.phase1 = (double *)c->p1.p;
    dr.phase2 = (double *)c->p2.p;
    dr.pair_call = r->pair_call ? (int64_t *)c->pcall.p : nullptr;
    dr.pair_bits = r->pair_bits ? (uint8_t *)c->pbits.p : nullptr;
    dr.n_tie_loci = 0;
    c->call_hint = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(max_reads, 1), 0xffffffffull);  // skips the deep-locus launches when no locus needs them
    if ((rc = enqueue_batch(c, &db, &dr, s)) != INQ_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(r->phase1, dr.phase1, (size_t)b->n_loci * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(r->phase2, dr.phase2, (size_t)b->n_loci * 8, hipMemcpyDeviceToHost, s));
    if (r->pair_call && b->n_pairs)
        HIP_TRY(c, hipMemcpyAsync(r->pair_call, dr.pair_call, (size_t)b->n_pairs * 8, hipMemcpyDeviceToHost, s));
    if (r->pair_bits && b->n_pairs)
        HIP_TRY(c, hipMemcpyAsync(r->pair_bits, dr.pair_bits, (size_t)b->n_pairs, hipMemcpyDeviceToHost, s));
    // status travels with the results: one synchronisation per call; the device copy is cleared on the
    // same stream, i.e. before the next host-entry call's kernels
    HIP_TRY(c, hipMemcpyAsync(c->h_status, c->d_status, sizeof(DevStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemsetAsync(&c->d_status->err, 0, sizeof(unsigned int), s));
    HIP_TRY(c, hipMemsetAsync(&c->d_status->ties, 0, sizeof(unsigned long long), s));
    HIP_TRY(c, hipStreamSynchronize(s));
    purge_retired(c);  // the stream is idle: buffers this call outgrew go back now
    r->n_tie_loci = c->h_status->ties;
    return status_to_code(c->h_status->err);
}

int inq_call_batch(inq_ctx_t *c, const inq_batch_t *b, inq_result_t *r) {
    try {
        return call_batch_impl(c, b, r);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

static int outlier_rows_impl(inq_ctx_t *c, const float *values, const uint32_t *row_len, uint64_t n_rows, uint32_t stride,
                             int method, uint32_t minsize, float cutoff, uint32_t mincluster, uint8_t *flags, uint8_t *keep) {
    if (!c || (method != INQ_OUTLIER_ZSCORE && method != INQ_OUTLIER_DBSCAN)) return INQ_ERR_ARG;
    if (n_rows && (!row_len || !keep || (stride && (!values || !flags)))) return INQ_ERR_ARG;
    if (n_rows >= 0x7fffffffull) return INQ_ERR_ARG;
    for (uint64_t i = 0; i < n_rows; ++i)
        if (row_len[i] > stride) return INQ_ERR_ARG;
    if (!n_rows) return INQ_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t cells = (size_t)n_rows * stride;
    int rc;
    if ((rc = ensure(c, c->ovalues, cells * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, c->olen, n_rows * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, c->oflags, cells)) != INQ_OK) return rc;
    if ((rc = ensure(c, c->okeep, n_rows)) != INQ_OK) return rc;
    const bool tile = c->outlier_tile && stride <= kOutlierTileMaxStride;
    if (method == INQ_OUTLIER_ZSCORE && !tile && (rc = ensure(c, c->otrans, outlier_rows_padded(n_rows) * stride * 4)) != INQ_OK) return rc;
    if (cells) HIP_TRY(c, hipMemcpyAsync(c->ovalues.p, values, cells * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->olen.p, row_len, n_rows * 4, hipMemcpyHostToDevice, s));
    if (cells) HIP_TRY(c, hipMemsetAsync(c->oflags.p, 0, cells, s));
    OutlierArgs a;
    a.values = (const float *)c->ovalues.p;
    a.row_len = (const uint32_t *)c->olen.p;
    a.n_rows = n_rows;
    a.stride = stride;
    a.minsize = minsize;
    a.zscore_cutoff = cutoff;
    a.mincluster = mincluster;
    a.flags = (uint8_t *)c->oflags.p;
    a.keep = (uint8_t *)c->okeep.p;
    EvTriple *ev = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev_pool.size()) {
            EvTriple t;
            HIP_TRY(c, hipEventCreate(&t.e0));
            HIP_TRY(c, hipEventCreate(&t.e1));
            HIP_TRY(c, hipEventCreate(&t.e2));
            c->ev_pool.push_back(t);
        }
        ev = &c->ev_pool[c->ev_used++];
        HIP_TRY(c, hipEventRecord(ev->e0, s));
    }
    launch_outlier(a, method, (float *)c->otrans.p, s, tile);
    HIP_TRY(c, hipGetLastError());
    if (ev) {
        HIP_TRY(c, hipEventRecord(ev->e1, s));
        HIP_TRY(c, hipEventRecord(ev->e2, s));
    }
    if (cells) HIP_TRY(c, hipMemcpyAsync(flags, c->oflags.p, cells, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(keep, c->okeep.p, n_rows, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return INQ_OK;
}

int inq_outlier_rows(inq_ctx_t *c, const float *values, const uint32_t *row_len, uint64_t n_rows, uint32_t stride, int method,
                     uint32_t minsize, float zscore_cutoff, uint32_t mincluster, uint8_t *flags, uint8_t *keep) {
    try {
        return outlier_rows_impl(c, values, row_len, n_rows, stride, method, minsize, zscore_cutoff, mincluster, flags, keep);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_ctx_timing_enable(inq_ctx_t *c, int on) {
    if (!c) return INQ_ERR_ARG;
    c->timing = on != 0;
    return INQ_OK;
}

int inq_ctx_timing_reset(inq_ctx_t *c) {
    if (!c) return INQ_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    c->ev_used = 0;
    return INQ_OK;
}

int inq_ctx_timing_read(inq_ctx_t *c, int which, double *total_ms, uint64_t *launches) {
    if (!c || which < 0 || which > 2) return INQ_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (which == 2) {  // the last BGZF inflate launch (device front end)
        HIP_TRY(c, hipDeviceSynchronize());
        const double ms = span_last_inflate_ms(c->span);
        if (total_ms) *total_ms = ms < 0 ? 0.0 : ms;
        if (launches) *launches = ms < 0 ? 0u : 1u;
        return INQ_OK;
    }
    double tot = 0.0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        EvTriple &e = c->ev_pool[i];
        HIP_TRY(c, hipEventSynchronize(e.e2));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, e.e0, which == 0 ? e.e2 : e.e1));
        tot += (double)ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (uint64_t)c->ev_used;
    return INQ_OK;
}

// options every context made from now on starts with (inq_default_option): applied inside inq_ctx_create_early BEFORE the context is
// published for staging, so that an uploader thread never sees the built-in value of e.g. "inflate_ahead"
static std::mutex g_defaults_mu;
static std::vector<std::pair<std::string, int64_t>> g_defaults;

int inq_default_option(const char *key, int64_t value) {
    if (!key) return INQ_ERR_ARG;
    try {
        inq_ctx probe;  // the key and the value are checked on a context that is never opened
        probe.n_cus = 256;
        const int rc = inq_ctx_set_option(&probe, key, value);
        if (rc != INQ_OK) return rc;
        std::lock_guard<std::mutex> g(g_defaults_mu);
        for (auto &kv : g_defaults)
            if (kv.first == key) {
                kv.second = value;
                return INQ_OK;
            }
        g_defaults.emplace_back(key, value);
        return INQ_OK;
    } catch (...) {
        return INQ_ERR_NOMEM;
    }
}

static bool default_option(const char *key, int64_t *value) {
    std::lock_guard<std::mutex> g(g_defaults_mu);
    for (auto &kv : g_defaults)
        if (kv.first == key) {
            if (value) *value = kv.second;
            return true;
        }
    return false;
}
int inq_default_option_get(const char *key, int64_t *value) {
    if (!key) return INQ_ERR_ARG;
    try {
        return default_option(key, value) ? INQ_OK : INQ_ERR_ARG;
    } catch (...) {
        return INQ_ERR_NOMEM;
    }
}

static void apply_default_options(inq_ctx *c) {
    std::vector<std::pair<std::string, int64_t>> d;
    {
        std::lock_guard<std::mutex> g(g_defaults_mu);
        d = g_defaults;
    }
    for (auto &kv : d) (void)inq_ctx_set_option(c, kv.first.c_str(), kv.second);
}

int inq_ctx_set_option(inq_ctx_t *c, const char *key, int64_t value) {
    if (!c || !key) return INQ_ERR_ARG;
    if (std::strcmp(key, "grid_tail") == 0 || std::strcmp(key, "grid_big") == 0) {  // ("grid_big": the name up to ABI v4)
        if (value < 1 || value > 65535) return INQ_ERR_ARG;
        c->grid_tail = std::min<uint32_t>((uint32_t)value, c->n_cus ? c->n_cus : 1u);  // all of them must be resident at once
        return INQ_OK;
    }
    if (std::strcmp(key, "grid_medium") == 0) {
        if (value < 1 || value > 65535) return INQ_ERR_ARG;
        c->grid_medium = (uint32_t)value;
        return INQ_OK;
    }
    if (std::strcmp(key, "max_reads_hint") == 0) {
        if (value < 0 || value > 0xffffffffll) return INQ_ERR_ARG;
        c->max_reads_hint = (uint32_t)value;
        return INQ_OK;
    }
    if (std::strcmp(key, "verify_crc") == 0) {
        c->verify_crc = value != 0;
        return INQ_OK;
    }
    if (std::strcmp(key, "inflate_algo") == 0) {
        if (value < 0 || value > 2) return INQ_ERR_ARG;
        c->inflate_algo = (uint32_t)value;
        return INQ_OK;
    }
    if (std::strcmp(key, "inflate_lit_pairs") == 0) {
        c->inflate_lit_pairs = value < 0 ? -1 : (value != 0);
        return INQ_OK;
    }
    if (std::strcmp(key, "outlier_tile") == 0) {
        c->outlier_tile = value != 0;
        return INQ_OK;
    }
    if (std::strcmp(key, "blocking_sync") == 0) {
        // as a default option (set BEFORE the context is made) it also makes the context's events blocking ones; on a live context it
        // switches the device's wait mode only
        c->blocking_sync = value != 0;
        if (c->device >= 0 && c->stream) {
            (void)hipSetDevice(c->device);
            (void)hipSetDeviceFlags(value ? hipDeviceScheduleBlockingSync : hipDeviceScheduleSpin);
            (void)hipGetLastError();
        }
        return INQ_OK;
    }
    if (std::strcmp(key, "inflate_ahead") == 0) {
        c->inflate_ahead = value != 0;
        return INQ_OK;
    }
    if (std::strcmp(key, "batch_loci_hint") == 0) {
        c->batch_loci_hint = value < 0 ? 0u : (uint64_t)value;
        return INQ_OK;
    }
    if (std::strcmp(key, "gather_nt") == 0) {
        c->gather_nt = value != 0;
        return INQ_OK;
    }
    if (std::strcmp(key, "inflate_tokens") == 0) {
        c->inflate_tokens = value < 0 ? -1 : (value != 0);
        return INQ_OK;
    }
    if (std::strcmp(key, "retired_limit_mb") == 0) {  // outgrown buffers parked before they are given back (default 16384)
        if (value < 0 || value > (1ll << 20)) return INQ_ERR_ARG;
        c->retired_limit = (size_t)value << 20;
        return INQ_OK;
    }
    if (std::strcmp(key, "test_fail_allocs") == 0) {  // test seam: the next N device allocations see "out of memory" at their first attempt
        if (value < 0 || value > 1000000) return INQ_ERR_ARG;
        c->test_fail_allocs = (uint32_t)value;
        return INQ_OK;
    }
    if (std::strcmp(key, "nt_loads") == 0) {
        c->nt_loads = value < 0 ? -1 : (value != 0);
        return INQ_OK;
    }
    return INQ_ERR_ARG;
}

uint64_t inq_ctx_alloc_retries(const inq_ctx_t *c) { return c ? c->alloc_retries : 0; }

int inq_pin_host(void *p, size_t bytes) {
    if (!p || !bytes) return INQ_ERR_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? INQ_OK : INQ_ERR_HIP;
}

void inq_unpin_host(void *p) {
    if (p) (void)hipHostUnregister(p);
}

int inq_alloc_pinned(size_t bytes, void **out) {
    if (!out) return INQ_ERR_ARG;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return INQ_ERR_NOMEM;
    if (e != hipSuccess) return INQ_ERR_NO_DEVICE;
    return INQ_OK;
}

void inq_free_pinned(void *p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"

int inq::call_batch_device_impl(inq_ctx *c, const inq_batch_t *b, inq_result_t *r, void *hip_stream) {
    return enqueue_batch(c, b, r, hip_stream);
}

