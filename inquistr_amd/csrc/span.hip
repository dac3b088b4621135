// span.hip — the device front end behind inq_bgzf_inflate() / inq_call_span() (include/inquistr_hip.h):
// sequences bgzf_inflate.hip and bam_scan.hip, then hands the device-resident batch to the locus kernels.
// Three small readbacks per span (record count, CIGAR size, pair count) size the buffers of the next stage.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "ctx.h"
#include "deflate_probe.h"
#include "front_kernels.h"

namespace inq {

struct SpanState {
    DevBuf tok;  // the workgroup inflate's token scratch
    DevBuf comp, blocks, u, block_status, anchors, anchor_cnt, anchor_base, rec_off, reads, info, key, endkey, pmax, cig_off, cigar,
        anchor_stop, ltid, lstart, lend, locus_cnt, locus_off, pair_read, p1, p2, tmp;
    FrontStatus *d_st = nullptr;
    struct Host {  // pinned readback area
        FrontStatus st;
        DevStatus ks;
        uint64_t val[4];
        unsigned long long init_n_valid;
    } *h = nullptr;
    // pinned staging of the rows: the caller's result arrays are ordinary memory, and the FIRST copy to or from pageable
    // memory costs the runtime 7-20 ms (it builds its own staging then): rows go through this buffer instead
    double *h_rows = nullptr;
    size_t h_rows_cap = 0;  // in doubles
    bool copy_path_warm = false;  // the runtime sets up its device-to-host copy path at the first copy of some size (~8 ms)
    hipEvent_t ev[6] = {};
    bool have_ev = false;
    // inq_span_stage: compressed bytes + block table + anchors of up to three spans, uploaded on their own
    // stream (possibly by another host thread) while an earlier span is being inflated
    struct Stage {
        DevBuf comp, blocks, anchors, anchor_stop;
        const void *host_comp = nullptr;
        uint64_t comp_bytes = 0, n_blocks = 0, n_anchors = 0;
        bool valid = false;
        // ... and inflated there too (option "inflate_ahead"), on a stream of its own: the inflate of span k + 1 runs next to span
        // k's record scan, gather and join - kernels that leave most of the chip idle - and next to the tail of span k's inflate
        DevBuf u, tok;
        unsigned int *d_err = nullptr;  // the INQ_INFLATE_* bits of this slot's blocks
        hipEvent_t ev_up = nullptr, ev_inf0 = nullptr, ev_inf1 = nullptr;
        bool inflated = false;
        // block table + anchors + anchor stops, copied here (page-locked) when the span is staged: ONE small DMA behind the compressed
        // bytes instead of three copies from pageable memory (each of those costs the copy stream 20 - 30 us of staging between two
        // spans' 268 MB), and the caller's tables are free again at once
        uint8_t *h_tab = nullptr;
        size_t h_tab_cap = 0;
        DevBuf tab;  // device side of it: blocks | anchors | anchor_stop, each 16-byte aligned
        bool pending = false;  // inq_span_stage_begin went through, inq_span_stage_wait has not
    } stage[INQ_SPAN_SLOTS];  // two sets of four: the spans of the NEXT file of a cohort are staged while this file's are still being called
    hipStream_t copy_stream = nullptr;
    hipStream_t ahead_stream = nullptr;  // the inflates launched at staging time
    uint8_t *h_warm = nullptr;  // page-locked target of the one warm-up copy (call_span_impl)
    // the batch the last inq_call_span built
    uint64_t n_reads = 0, n_cigar_words = 0, n_pairs = 0, n_loci = 0;
    bool last_from_acc = false;
    // inq_call_span_deferred: the batches of several spans appended to one (CIGAR units, reads, pairs, loci so far), called
    // together by inq_call_flush - a span of SEQ-bearing records holds a few hundred loci, far too few to fill the chip
    struct Acc {
        DevBuf cigar, reads, pair_read, off, lstart, lend;
        uint64_t n_units = 0, n_reads = 0, n_pairs = 0, n_loci = 0, n_spans = 0;
        uint32_t minlen = 0, support = 0, unphased = 0, max_reads = 0;
    } acc;
};

double span_last_inflate_ms(SpanState *S) {
    float ms = 0.f;
    if (!S || !S->have_ev || hipEventElapsedTime(&ms, S->ev[1], S->ev[2]) != hipSuccess) return -1.0;
    return (double)ms;
}

void span_state_destroy(SpanState *S) {
    if (!S) return;
    for (DevBuf *b : {&S->comp, &S->blocks, &S->u, &S->block_status, &S->anchors, &S->anchor_cnt, &S->anchor_base, &S->rec_off,
                      &S->reads, &S->info, &S->key, &S->endkey, &S->pmax, &S->cig_off, &S->cigar, &S->anchor_stop, &S->ltid, &S->lstart, &S->lend, &S->locus_cnt,
                      &S->locus_off, &S->pair_read, &S->p1, &S->p2, &S->tmp, &S->tok})
        if (b->p) (void)hipFree(b->p);
    for (auto &g : S->stage) {
        for (DevBuf *b : {&g.comp, &g.u, &g.tok, &g.tab})  // (blocks / anchors / anchor_stop are views into tab)
            if (b->p) (void)hipFree(b->p);
        if (g.d_err) (void)hipFree(g.d_err);
        if (g.h_tab) (void)hipHostFree(g.h_tab);
        for (hipEvent_t e : {g.ev_up, g.ev_inf0, g.ev_inf1})
            if (e) (void)hipEventDestroy(e);
    }
    if (S->ahead_stream) (void)hipStreamDestroy(S->ahead_stream);
    for (DevBuf *b : {&S->acc.cigar, &S->acc.reads, &S->acc.pair_read, &S->acc.off, &S->acc.lstart, &S->acc.lend})
        if (b->p) (void)hipFree(b->p);
    if (S->copy_stream) (void)hipStreamDestroy(S->copy_stream);
    if (S->h_warm) (void)hipHostFree(S->h_warm);
    if (S->d_st) (void)hipFree(S->d_st);
    if (S->h) (void)hipHostFree(S->h);
    if (S->h_rows) (void)hipHostFree(S->h_rows);
    if (S->have_ev)
        for (auto &e : S->ev) (void)hipEventDestroy(e);
    delete S;
}

}  // namespace inq

using namespace inq;

int inq::span_state_init(inq_ctx *c) {
    SpanState *S = new (std::nothrow) SpanState();
    if (!S) return INQ_ERR_NOMEM;
    c->span = S;
    // what STAGING needs, and nothing else: inq_ctx_create_early publishes the context right behind this function
    HIP_TRY(c, hipStreamCreateWithFlags(&S->copy_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&S->ahead_stream, hipStreamNonBlocking));
    // ("inflate_ahead", "gather_nt": options, inq_ctx_set_option / inq_default_option - the environment is not read)
    for (auto &g : S->stage) {
        HIP_TRY(c, hipMalloc((void **)&g.d_err, sizeof(unsigned int)));
        // (ev_up is waited for by the uploader thread: a blocking event gives its core back to the readers while a span crosses the link)
        HIP_TRY(c, hipEventCreateWithFlags(&g.ev_up, c->blocking_sync ? hipEventBlockingSync : hipEventDefault));
        HIP_TRY(c, hipEventCreate(&g.ev_inf0));
        HIP_TRY(c, hipEventCreate(&g.ev_inf1));
    }
    return INQ_OK;
}

// the rest of the span state: what the calls (not the staging) need
int inq::span_state_init_rest(inq_ctx *c) {
    SpanState *S = c->span;
    HIP_TRY(c, hipMalloc((void **)&S->d_st, sizeof(FrontStatus)));
    HIP_TRY(c, hipHostMalloc((void **)&S->h, sizeof(SpanState::Host), hipHostMallocDefault));
    for (auto &e : S->ev) HIP_TRY(c, hipEventCreate(&e));
    S->have_ev = true;
    return INQ_OK;
}

namespace {

int span_state(inq_ctx *c, SpanState **out) {
    if (!c->span) return INQ_ERR_HIP;  // created with the ctx (span_state_init): two host threads may come here at once
    *out = c->span;
    return INQ_OK;
}

// grows b to `bytes`, keeping its first `used` bytes (the accumulated batch): the copy is ordered on the stream behind everything
// that wrote the old buffer and in front of everything that will read the new one; the old buffer is retired (ctx.h), nothing waits
int ensure_keep(inq_ctx *c, DevBuf &b, size_t bytes, size_t used, hipStream_t s, size_t reserve = 0) {
    if (bytes <= b.cap && b.p) return INQ_OK;
    void *np = nullptr;
    const size_t want = std::max(bytes < (32u << 20) ? bytes + bytes / 2 + (1u << 20) : bytes + bytes / 8 + (1u << 20), reserve);
    // (the speculative reserve - the whole batch's size guessed from its first span - and the headroom are given up before the call is:
    // device_alloc frees the parked buffers, then falls back to the bytes really needed)
    size_t got = 0;
    const int rc = device_alloc(c, &np, want, bytes, &got);
    if (rc != INQ_OK) return rc;
    if (b.p) {
        if (used) {
            const hipError_t e = hipMemcpyAsync(np, b.p, used, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) {
                (void)hipFree(np);
                c->last_err = std::string("growing the deferred batch: ") + hipGetErrorString(e);
                return INQ_ERR_HIP;
            }
        }
        retire(c, b.p, b.cap);
    }
    b.p = np;
    b.cap = got;
    return INQ_OK;
}

// host-side shape checks of the block table: everything the inflate grid assumes
int check_blocks(const uint8_t *comp, uint64_t comp_bytes, const inq_bgzf_block_t *blocks, uint64_t n_blocks, uint64_t out_bytes,
                 bool dense) {
    if (n_blocks && (!blocks || !comp)) return INQ_ERR_ARG;
    uint64_t next = 0;
    for (uint64_t i = 0; i < n_blocks; ++i) {
        const inq_bgzf_block_t &b = blocks[i];
        // payload + the 8-byte CRC32 / ISIZE trailer of the block
        if (b.comp_off > comp_bytes || (uint64_t)b.comp_len + 8u > comp_bytes - b.comp_off) return INQ_ERR_ARG;
        if (b.isize > 65536u || b.out_off > out_bytes || b.isize > out_bytes - b.out_off) return INQ_ERR_ARG;
        if (dense && b.out_off != next) return INQ_ERR_ARG;
        next = b.out_off + b.isize;
    }
    return INQ_OK;
}

// uploads the compressed bytes and the block table, clears the front status, inflates into S->u
int upload_and_inflate(inq_ctx *c, SpanState *S, const uint8_t *comp, uint64_t comp_bytes, const inq_bgzf_block_t *blocks,
                       uint64_t n_blocks, uint64_t out_bytes, bool want_block_status, hipStream_t s,
                       const SpanState::Stage *staged = nullptr) {
    int rc;
    constexpr size_t kPad = 64;
    const void *d_comp, *d_blocks;
    if (staged) {  // already on the device (inq_span_stage)
        d_comp = staged->comp.p;
        d_blocks = staged->blocks.p;
    } else {
        if ((rc = ensure(c, S->comp, comp_bytes + kPad)) != INQ_OK) return rc;
        if ((rc = ensure(c, S->blocks, n_blocks * sizeof(inq_bgzf_block_t))) != INQ_OK) return rc;
        if (comp_bytes) HIP_TRY(c, hipMemcpyAsync(S->comp.p, comp, comp_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemsetAsync((uint8_t *)S->comp.p + comp_bytes, 0, kPad, s));
        if (n_blocks) HIP_TRY(c, hipMemcpyAsync(S->blocks.p, blocks, n_blocks * sizeof(inq_bgzf_block_t), hipMemcpyHostToDevice, s));
        d_comp = S->comp.p;
        d_blocks = S->blocks.p;
    }
    HIP_TRY(c, hipMemsetAsync(S->d_st, 0, sizeof(FrontStatus), s));
    HIP_TRY(c, hipMemsetAsync(&S->d_st->first_bad, 0xff, sizeof(unsigned long long), s));
    if (staged && staged->inflated) {  // inflated when it was staged: wait for that, take over its status word
        HIP_TRY(c, hipEventRecord(S->ev[1], s));
        HIP_TRY(c, hipStreamWaitEvent(s, staged->ev_inf1, 0));
        HIP_TRY(c, hipMemcpyAsync(&S->d_st->inflate, staged->d_err, sizeof(unsigned int), hipMemcpyDeviceToDevice, s));
        return INQ_OK;
    }
    if ((rc = ensure(c, S->u, out_bytes + kPad)) != INQ_OK) return rc;
    if (want_block_status && (rc = ensure(c, S->block_status, n_blocks * 4)) != INQ_OK) return rc;
    HIP_TRY(c, hipMemsetAsync((uint8_t *)S->u.p + out_bytes, 0, kPad, s));
    HIP_TRY(c, hipEventRecord(S->ev[1], s));
    InflateArgs ia;
    ia.comp = (const uint8_t *)d_comp;
    ia.comp_bytes = comp_bytes;
    ia.blocks = (const inq_bgzf_block_t *)d_blocks;
    ia.n_blocks = n_blocks;
    ia.out = (uint8_t *)S->u.p;
    ia.out_bytes = out_bytes;
    ia.block_status = want_block_status ? (uint32_t *)S->block_status.p : nullptr;
    ia.err = &S->d_st->inflate;
    ia.verify_crc = c->verify_crc ? 1u : 0u;
    ia.debug_flags = 0u;
    // timing experiments only (drops stores: wrong bytes); reads nothing unless built with -DINQ_DEBUG_ENV
    if (const char *dbg = debug_env("INQ_INFLATE_DEBUG")) ia.debug_flags = (uint32_t)std::atoi(dbg);
    ia.algo = c->inflate_algo;
    // literal-heavy or match-heavy?  (the host still has the compressed bytes: a few block headers are read; option
    // "inflate_lit_pairs" = 0 / 1 forces a form, -1 = look)
    ia.lit_pairs = c->inflate_lit_pairs < 0 ? inflate_wants_literal_pairs(comp, comp_bytes, blocks, n_blocks) : (uint32_t)c->inflate_lit_pairs;
    ia.tokens = nullptr;
    if (ia.algo != 1u && (c->inflate_tokens < 0 ? ia.lit_pairs == 0u : c->inflate_tokens != 0)) {
        if ((rc = ensure(c, S->tok, inflate_token_words(n_blocks) * 4)) != INQ_OK) return rc;
        ia.tokens = (uint32_t *)S->tok.p;
    }
    launch_bgzf_inflate(ia, s);
    HIP_TRY(c, hipGetLastError());
    return INQ_OK;
}

int bgzf_inflate_impl(inq_ctx *c, const uint8_t *comp, uint64_t comp_bytes, const inq_bgzf_block_t *blocks, uint64_t n_blocks,
                      uint8_t *out, uint64_t out_bytes, uint32_t *block_status) {
    if (!c || (out_bytes && !out)) return INQ_ERR_ARG;
    int rc = check_blocks(comp, comp_bytes, blocks, n_blocks, out_bytes, false);
    if (rc != INQ_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    SpanState *S;
    if ((rc = span_state(c, &S)) != INQ_OK) return rc;
    hipStream_t s = c->stream;
    if ((rc = upload_and_inflate(c, S, comp, comp_bytes, blocks, n_blocks, out_bytes, true, s)) != INQ_OK) return rc;
    HIP_TRY(c, hipEventRecord(S->ev[2], s));
    if (out_bytes) HIP_TRY(c, hipMemcpyAsync(out, S->u.p, out_bytes, hipMemcpyDeviceToHost, s));
    if (block_status && n_blocks) HIP_TRY(c, hipMemcpyAsync(block_status, S->block_status.p, n_blocks * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&S->h->st, S->d_st, sizeof(FrontStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return S->h->st.inflate ? INQ_ERR_INFLATE : INQ_OK;
}

// defer: the span's batch is appended to S->acc instead of being called (r may be null then)
int call_span_impl(inq_ctx *c, const inq_span_t *sp, inq_result_t *r, inq_span_stats_t *stats, int slot, bool defer = false) {
    if (!c || !sp || (!r && !defer) || slot >= INQ_SPAN_SLOTS) return INQ_ERR_ARG;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (sp->reserved || sp->unphased > 1) return INQ_ERR_ARG;
    if (sp->n_loci && (!sp->locus_tid || !sp->locus_start || !sp->locus_end || (!defer && (!r->phase1 || !r->phase2)))) return INQ_ERR_ARG;
    if (sp->n_anchors && (!sp->anchors || !sp->anchor_stop)) return INQ_ERR_ARG;
    if (sp->n_loci >= 0xfffffff0ull) return INQ_ERR_ARG;
    if (sp->support == 0) return INQ_ERR_SUPPORT_ZERO;
    const uint64_t nb = sp->n_blocks;
    const uint64_t u_bytes = nb ? sp->blocks[nb - 1].out_off + sp->blocks[nb - 1].isize : 0;
    int rc = check_blocks(sp->comp, sp->comp_bytes, sp->blocks, nb, u_bytes, true);
    if (rc != INQ_OK) return rc;
    for (uint64_t i = 0; i < sp->n_anchors; ++i) {
        const uint64_t stop = sp->anchor_stop[i] & ~INQ_ANCHOR_SEGMENT_END;
        if (sp->anchors[i] > u_bytes || (i && sp->anchors[i] <= sp->anchors[i - 1])) return INQ_ERR_ARG;
        if (stop < sp->anchors[i] || stop > u_bytes || (i + 1 < sp->n_anchors && stop > sp->anchors[i + 1])) return INQ_ERR_ARG;
    }
    for (uint64_t j = 0; j < sp->n_loci; ++j) {
        if (sp->locus_tid[j] < 0) return INQ_ERR_ARG;
        if (sp->locus_start[j] < 10 || sp->locus_end[j] < sp->locus_start[j]) return INQ_ERR_LOCUS;
    }
    if (r) r->n_tie_loci = 0;
    if (sp->n_loci == 0) return INQ_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    SpanState *S;
    if ((rc = span_state(c, &S)) != INQ_OK) return rc;
    hipStream_t s = c->stream;
    const uint64_t nl = sp->n_loci, na = sp->n_anchors;
    SpanState::Acc &A = S->acc;
    if (defer && A.n_spans && (A.minlen != sp->minlen || A.support != sp->support || A.unphased != sp->unphased)) return INQ_ERR_ARG;
    const SpanState::Stage *staged = nullptr;
    if (slot >= 0) {  // the span inq_span_stage put there, and nothing else
        const SpanState::Stage &g = S->stage[slot];
        if (!g.valid || g.host_comp != sp->comp || g.comp_bytes != sp->comp_bytes || g.n_blocks != nb || g.n_anchors != na) return INQ_ERR_ARG;
        staged = &g;
    }

    // ---- stage 1: upload, inflate, count the records
    const bool verbose = std::getenv("INQ_TIMING") && std::getenv("INQ_TIMING")[0] == '2';
    using clk = std::chrono::steady_clock;
    const auto w0 = clk::now();
    auto wall = [&](const char *what) {
        if (verbose) std::fprintf(stderr, "[inq span host] %-28s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(clk::now() - w0).count());
    };
    HIP_TRY(c, hipEventRecord(S->ev[0], s));
    if (!staged) {
        if ((rc = ensure(c, S->anchors, na * 8)) != INQ_OK) return rc;
        if ((rc = ensure(c, S->anchor_stop, na * 8)) != INQ_OK) return rc;
    }
    if ((rc = ensure(c, S->ltid, nl * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->anchor_cnt, na * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->anchor_base, (na + 1) * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->lstart, nl * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->lend, nl * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->locus_cnt, nl * 4)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->locus_off, (nl + 1) * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->p1, nl * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->p2, nl * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->tmp, scan_tmp_words(std::max<uint64_t>(std::max(na, nl), 1)) * 8)) != INQ_OK) return rc;
    if (na && !staged) HIP_TRY(c, hipMemcpyAsync(S->anchors.p, sp->anchors, na * 8, hipMemcpyHostToDevice, s));
    if (na && !staged) HIP_TRY(c, hipMemcpyAsync(S->anchor_stop.p, sp->anchor_stop, na * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(S->ltid.p, sp->locus_tid, nl * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(S->lstart.p, sp->locus_start, nl * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(S->lend.p, sp->locus_end, nl * 4, hipMemcpyHostToDevice, s));
    if ((rc = upload_and_inflate(c, S, sp->comp, sp->comp_bytes, sp->blocks, nb, u_bytes, false, s, staged)) != INQ_OK) return rc;
    HIP_TRY(c, hipEventRecord(S->ev[2], s));
    wall("buffers + uploads enqueued");

    ScanArgs a;
    std::memset(&a, 0, sizeof a);
    const bool ahead = staged && staged->inflated;
    a.u = (const uint8_t *)(ahead ? staged->u.p : S->u.p);
    a.u_bytes = u_bytes;
    a.anchors = (const uint64_t *)(staged ? staged->anchors.p : S->anchors.p);
    a.n_anchors = na;
    a.anchor_cnt = (uint32_t *)S->anchor_cnt.p;
    a.anchor_base = (uint64_t *)S->anchor_base.p;
    a.anchor_stop = (const uint64_t *)(staged ? staged->anchor_stop.p : S->anchor_stop.p);
    a.locus_tid = (const int32_t *)S->ltid.p;
    a.unphased = sp->unphased;
    a.locus_start = (const uint32_t *)S->lstart.p;
    a.locus_end = (const uint32_t *)S->lend.p;
    a.n_loci = nl;
    a.locus_cnt = (uint32_t *)S->locus_cnt.p;
    a.locus_pair_off = (uint64_t *)S->locus_off.p;
    a.st = S->d_st;
    launch_chain_count(a, s);
    launch_scan_u32_to_u64(a.anchor_cnt, a.anchor_base, na, (uint64_t *)S->tmp.p, s);
    HIP_TRY(c, hipGetLastError());
    if (S->h_rows_cap < 2 * nl) {
        if (S->h_rows) (void)hipHostFree(S->h_rows);
        S->h_rows = nullptr;
        S->h_rows_cap = 0;
        const size_t want = std::max<size_t>(2 * nl + nl / 2 + 1024, 1u << 16);
        HIP_TRY(c, hipHostMalloc((void **)&S->h_rows, want * sizeof(double), hipHostMallocDefault));
        S->h_rows_cap = want;
    }
    if (!S->copy_path_warm && u_bytes >= (512u << 10)) {
        // the first device-to-host copy of this size costs the host ~8 ms inside the runtime; spent here, on the copy stream (the
        // other direction of the uploads), it hides behind the inflate that was just enqueued instead of sitting behind the last
        // kernel of the span.  It lands in a page-locked scratch of its own: nothing waits for it, no row can be overwritten by it
        // (round 4 put it on the ahead stream into the row staging: the first flush then waited for inflates queued behind it)
        if (!S->h_warm) HIP_TRY(c, hipHostMalloc((void **)&S->h_warm, 512u << 10, hipHostMallocDefault));
        HIP_TRY(c, hipMemcpyAsync(S->h_warm, ahead ? staged->u.p : S->u.p, 512u << 10, hipMemcpyDeviceToHost, S->copy_stream));
        S->copy_path_warm = true;
    }
    HIP_TRY(c, hipMemcpyAsync(&S->h->val[0], a.anchor_base + na, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&S->h->st, S->d_st, sizeof(FrontStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    auto front_fail = [&](const FrontStatus &st) -> int {
        if (stats) {
            stats->front_status = st.err | (st.inflate << 16);
            stats->first_bad_record = st.first_bad;
        }
        if (st.inflate) return INQ_ERR_INFLATE;
        if (st.err & (FS_CHAIN | FS_RECORD | FS_UNSORTED | FS_TOO_BIG)) return INQ_ERR_BAM;
        if (st.err & (FS_HP_TYPE | FS_SA_TYPE | FS_SA_FORMAT)) return INQ_ERR_AUX;
        return INQ_OK;
    };
    if ((rc = front_fail(S->h->st)) != INQ_OK) return rc;
    const uint64_t n_rec = S->h->val[0];
    wall("inflate + chain count done");

    // ---- stage 2: record offsets, fields, CIGAR sizes
    if ((rc = ensure(c, S->rec_off, n_rec * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->reads, n_rec * sizeof(inq_read_t))) != INQ_OK) return rc;
    if ((rc = ensure(c, S->info, n_rec * sizeof(RecInfo))) != INQ_OK) return rc;
    if ((rc = ensure(c, S->key, n_rec * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->endkey, n_rec * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->pmax, n_rec * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->cig_off, (n_rec + 1) * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->tmp, scan_tmp_words(std::max<uint64_t>(std::max(std::max(na, nl), n_rec), 1)) * 8)) != INQ_OK) return rc;
    a.rec_off = (uint64_t *)S->rec_off.p;
    a.n_records = n_rec;
    a.reads = (inq_read_t *)S->reads.p;
    a.info = (RecInfo *)S->info.p;
    a.key = (int64_t *)S->key.p;
    a.endkey = (int64_t *)S->endkey.p;
    a.pmax = (int64_t *)S->pmax.p;
    a.cig_off = (uint64_t *)S->cig_off.p;
    S->h->init_n_valid = n_rec;
    HIP_TRY(c, hipMemcpyAsync(&S->d_st->n_valid, &S->h->init_n_valid, 8, hipMemcpyHostToDevice, s));
    launch_chain_fill(a, s);
    launch_record_parse(a, s);
    launch_scan_cigar_units(a.reads, &S->d_st->n_valid, a.cig_off, n_rec, (uint64_t *)S->tmp.p, s);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&S->h->val[1], a.cig_off + n_rec, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&S->h->st, S->d_st, sizeof(FrontStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if ((rc = front_fail(S->h->st)) != INQ_OK) return rc;
    const uint64_t n_valid = S->h->st.n_valid, n_units = S->h->val[1];
    wall("parse done");
    if (n_valid > n_rec || n_valid >= 0xfffffff0ull || n_units >= 0xffffffffull) {
        if (stats) stats->front_status = FS_TOO_BIG;
        return INQ_ERR_BAM;
    }

    // ---- stage 3: CIGAR gather, reference spans, overlap join (count)
    if (defer) {  // behind the CIGARs and reads of the spans already deferred
        if (A.n_units + n_units >= 0xffffffffull || A.n_reads + n_valid >= 0xfffffff0ull) {
            if (stats) stats->front_status = FS_TOO_BIG;
            return INQ_ERR_BAM;
        }
        // The first span of a batch says what a locus of this file weighs: with the caller's word on how many loci a batch will
        // hold ("batch_loci_hint": the driver flushes at that many) the buffers are made for the whole batch at once, instead of
        // being outgrown span after span - every growth copies what is there (0.7 GB for a CIGAR-only span) with ordinary stores
        // right in front of the locus kernels that stream the batch (profiles/r04_results/locus_kernels_in_the_cli.txt)
        size_t res_units = 0, res_reads = 0;
        if (A.n_spans == 0 && c->batch_loci_hint > nl && nl) {
            const double spans = std::min(64.0, std::ceil((double)c->batch_loci_hint / (double)nl)) * 1.04;  // the flush comes with the span that reaches the hint
            res_units = (size_t)std::min((double)(12ull << 30), (double)n_units * 16.0 * spans);
            res_reads = (size_t)std::min((double)(4ull << 30), (double)n_valid * (double)sizeof(inq_read_t) * spans);
        }
        if ((rc = ensure_keep(c, A.cigar, (A.n_units + n_units) * 16, A.n_units * 16, s, res_units)) != INQ_OK) return rc;
        wall("  batch CIGAR buffer ready");
        if ((rc = ensure_keep(c, A.reads, (A.n_reads + n_valid) * sizeof(inq_read_t), A.n_reads * sizeof(inq_read_t), s, res_reads)) != INQ_OK) return rc;
        wall("  batch read buffer ready");
        a.cigar = (uint32_t *)A.cigar.p + A.n_units * 4;
        a.unit_base = (uint32_t)A.n_units;
        a.read_base = (uint32_t)A.n_reads;
    } else {
        if ((rc = ensure(c, S->cigar, n_units * 16)) != INQ_OK) return rc;
        a.cigar = (uint32_t *)S->cigar.p;
    }
    a.n_cigar_units = n_units;
    a.gather_nt = c->gather_nt ? 1u : 0u;
    launch_cigar_gather(a, n_valid, s);
    if (defer && n_valid)  // the descriptors, CIGAR offsets already counted from the start of the accumulated buffer
        HIP_TRY(c, hipMemcpyAsync((inq_read_t *)A.reads.p + A.n_reads, a.reads, n_valid * sizeof(inq_read_t), hipMemcpyDeviceToDevice, s));
    launch_scan_max_i64(a.endkey, a.pmax, n_valid, (uint64_t *)S->tmp.p, s);
    HIP_TRY(c, hipEventRecord(S->ev[3], s));
    launch_join_count(a, n_valid, s);
    launch_scan_u32_to_u64(a.locus_cnt, a.locus_pair_off, nl, (uint64_t *)S->tmp.p, s);
    HIP_TRY(c, hipGetLastError());
    wall("  gather + join count enqueued");
    HIP_TRY(c, hipMemcpyAsync(&S->h->val[2], a.locus_pair_off + nl, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&S->h->st, S->d_st, sizeof(FrontStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if ((rc = front_fail(S->h->st)) != INQ_OK) return rc;
    const uint64_t n_pairs = S->h->val[2];
    wall("gather + join count done");
    if (n_pairs >= (1ull << 40)) return INQ_ERR_ARG;

    // ---- stage 4: pairs, then the locus kernels on the device-resident batch
    auto fill_stats = [&]() {
        if (!stats) return;
        stats->n_records = n_rec;
        stats->n_reads = n_valid;
        stats->n_pairs = n_pairs;
        stats->n_cigar_words = n_units * 4;
        stats->inflated_bytes = u_bytes;
        stats->max_reads = S->h->st.max_reads;
        auto el = [&](int i, int j) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, S->ev[i], S->ev[j]);
            return (double)ms;
        };
        stats->ms_upload = el(0, 1);
        stats->ms_inflate = el(1, 2);
        if (ahead) {  // the inflate ran on its own stream while the span before was scanned: its own events
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, staged->ev_inf0, staged->ev_inf1);
            stats->ms_inflate = (double)ms;
        }
        stats->ms_scan = el(2, 3);
        stats->ms_join = el(3, 4);
        stats->ms_call = defer ? 0.0 : el(4, 5);
    };
    if (defer) {
        if (A.n_pairs + n_pairs >= (1ull << 40) || A.n_loci + nl >= 0xfffffff0ull) return INQ_ERR_ARG;
        size_t res_pairs = 0;
        if (A.n_spans == 0 && c->batch_loci_hint > nl && nl)
            res_pairs = (size_t)std::min((double)(4ull << 30), (double)n_pairs * 4.0 * std::min(64.0, std::ceil((double)c->batch_loci_hint / (double)nl)) * 1.04);
        if ((rc = ensure_keep(c, A.pair_read, (A.n_pairs + n_pairs) * 4, A.n_pairs * 4, s, res_pairs)) != INQ_OK) return rc;
        if ((rc = ensure_keep(c, A.off, (A.n_loci + nl + 1) * 8, (A.n_loci + 1) * 8, s)) != INQ_OK) return rc;
        if ((rc = ensure_keep(c, A.lstart, (A.n_loci + nl) * 4, A.n_loci * 4, s)) != INQ_OK) return rc;
        if ((rc = ensure_keep(c, A.lend, (A.n_loci + nl) * 4, A.n_loci * 4, s)) != INQ_OK) return rc;
        a.pair_read = (uint32_t *)A.pair_read.p + A.n_pairs;
        launch_join_fill(a, n_valid, s);
        launch_offset_copy((uint64_t *)A.off.p + A.n_loci, a.locus_pair_off, nl + 1, A.n_pairs, s);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync((uint32_t *)A.lstart.p + A.n_loci, a.locus_start, nl * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemcpyAsync((uint32_t *)A.lend.p + A.n_loci, a.locus_end, nl * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipEventRecord(S->ev[4], s));
        HIP_TRY(c, hipMemcpyAsync(&S->h->st, S->d_st, sizeof(FrontStatus), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));  // the span's slot and the scan buffers are free again when this returns
        fill_stats();
        if ((rc = front_fail(S->h->st)) != INQ_OK) return rc;
        if (A.n_spans == 0) A.minlen = sp->minlen, A.support = sp->support, A.unphased = sp->unphased, A.max_reads = 0;
        A.max_reads = std::max<uint32_t>(A.max_reads, S->h->st.max_reads);
        A.n_units += n_units, A.n_reads += n_valid, A.n_pairs += n_pairs, A.n_loci += nl, ++A.n_spans;
        wall("appended to the deferred batch");
        return INQ_OK;
    }
    if ((rc = ensure(c, S->pair_read, n_pairs * 4)) != INQ_OK) return rc;
    a.pair_read = (uint32_t *)S->pair_read.p;
    launch_join_fill(a, n_valid, s);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(S->ev[4], s));
    inq_batch_t db;
    std::memset(&db, 0, sizeof db);
    db.n_reads = n_valid;
    db.n_cigar_words = n_units * 4;
    db.n_pairs = n_pairs;
    db.n_loci = nl;
    db.cigar = a.cigar;
    db.reads = a.reads;
    db.pair_read = a.pair_read;
    db.locus_pair_off = a.locus_pair_off;
    db.locus_start = a.locus_start;
    db.locus_end = a.locus_end;
    db.minlen = sp->minlen;
    db.support = sp->support;
    db.unphased = sp->unphased;
    inq_result_t dr;
    dr.phase1 = (double *)S->p1.p;
    dr.phase2 = (double *)S->p2.p;
    dr.pair_call = nullptr;
    dr.pair_bits = nullptr;
    dr.n_tie_loci = 0;
    c->call_hint = std::max<uint32_t>(S->h->st.max_reads, 1u);
    if ((rc = call_batch_device_impl(c, &db, &dr, s)) != INQ_OK) return rc;
    HIP_TRY(c, hipEventRecord(S->ev[5], s));
    HIP_TRY(c, hipMemcpyAsync(S->h_rows, dr.phase1, nl * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(S->h_rows + nl, dr.phase2, nl * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&S->h->ks, c->d_status, sizeof(DevStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&S->h->st, S->d_st, sizeof(FrontStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemsetAsync(&c->d_status->err, 0, sizeof(unsigned int), s));
    HIP_TRY(c, hipMemsetAsync(&c->d_status->ties, 0, sizeof(unsigned long long), s));
    HIP_TRY(c, hipStreamSynchronize(s));
    std::memcpy(r->phase1, S->h_rows, nl * 8);
    std::memcpy(r->phase2, S->h_rows + nl, nl * 8);
    wall("call done");
    S->n_reads = n_valid;
    S->n_cigar_words = n_units * 4;
    S->n_pairs = n_pairs;
    S->n_loci = nl;
    S->last_from_acc = false;
    fill_stats();
    r->n_tie_loci = S->h->ks.ties;
    if ((rc = front_fail(S->h->st)) != INQ_OK) return rc;
    return status_to_code(S->h->ks.err);
}

// The locus kernels over everything inq_call_span_deferred appended; rows in the order the loci were appended.
// dev: the rows stay on the device - row j of the flush goes to dev->d1[dev->index[j]], dev->d2[...] - and r only receives the tie count
struct FlushToDevice {
    double *d1, *d2;
    const uint32_t *index;  // HOST, n_loci entries
    uint64_t cap;           // entries of d1 / d2
};
int call_flush_impl(inq_ctx *c, inq_result_t *r, uint64_t n_loci, double *ms_call, const FlushToDevice *dev = nullptr) {
    if (!c || !r) return INQ_ERR_ARG;
    SpanState *S;
    int rc;
    if ((rc = span_state(c, &S)) != INQ_OK) return rc;
    SpanState::Acc &A = S->acc;
    r->n_tie_loci = 0;
    if (ms_call) *ms_call = 0.0;
    if (n_loci != A.n_loci) return INQ_ERR_ARG;
    if (A.n_loci == 0) {
        A = SpanState::Acc{A.cigar, A.reads, A.pair_read, A.off, A.lstart, A.lend};
        return INQ_OK;
    }
    if (!dev && (!r->phase1 || !r->phase2)) return INQ_ERR_ARG;
    if (dev) {
        if (!dev->d1 || !dev->d2 || !dev->index) return INQ_ERR_ARG;
        for (uint64_t j = 0; j < n_loci; ++j)
            if ((uint64_t)dev->index[j] >= dev->cap) return INQ_ERR_ARG;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint64_t nl = A.n_loci;
    if ((rc = ensure(c, S->p1, nl * 8)) != INQ_OK) return rc;
    if ((rc = ensure(c, S->p2, nl * 8)) != INQ_OK) return rc;
    if (S->h_rows_cap < 2 * nl) {
        if (S->h_rows) (void)hipHostFree(S->h_rows);
        S->h_rows = nullptr;
        S->h_rows_cap = 0;
        const size_t want = std::max<size_t>(2 * nl + nl / 2 + 1024, 1u << 16);
        HIP_TRY(c, hipHostMalloc((void **)&S->h_rows, want * sizeof(double), hipHostMallocDefault));
        S->h_rows_cap = want;
    }
    inq_batch_t db;
    std::memset(&db, 0, sizeof db);
    db.n_reads = A.n_reads;
    db.n_cigar_words = A.n_units * 4;
    db.n_pairs = A.n_pairs;
    db.n_loci = nl;
    db.cigar = (const uint32_t *)A.cigar.p;
    db.reads = (const inq_read_t *)A.reads.p;
    db.pair_read = (const uint32_t *)A.pair_read.p;
    db.locus_pair_off = (const uint64_t *)A.off.p;
    db.locus_start = (const uint32_t *)A.lstart.p;
    db.locus_end = (const uint32_t *)A.lend.p;
    db.minlen = A.minlen;
    db.support = A.support;
    db.unphased = A.unphased;
    inq_result_t dr;
    dr.phase1 = (double *)S->p1.p;
    dr.phase2 = (double *)S->p2.p;
    dr.pair_call = nullptr;
    dr.pair_bits = nullptr;
    dr.n_tie_loci = 0;
    c->call_hint = std::max<uint32_t>(A.max_reads, 1u);
    // whatever happens from here on, the batch is spent: a failed flush must not leave its spans for the next file of a session
    struct Spent {
        SpanState::Acc &a;
        ~Spent() { a = SpanState::Acc{a.cigar, a.reads, a.pair_read, a.off, a.lstart, a.lend}; }
    } spent{A};
    const uint64_t n_reads_all = A.n_reads, n_units_all = A.n_units, n_pairs_all = A.n_pairs;
    HIP_TRY(c, hipEventRecord(S->ev[4], s));
    if ((rc = call_batch_device_impl(c, &db, &dr, s)) != INQ_OK) return rc;
    HIP_TRY(c, hipEventRecord(S->ev[5], s));
    // measurement builds only (make DEBUG_ENV=1; tools/profile_cli_locus.sh): the shipped library reads nothing here
    if (const char *again = debug_env("INQ_CALL_AGAIN"); again && again[0] == '1') {
        // measurement only: the same launch sequence once more on the same batch (same rows), to tell what a launch that comes
        // cold behind the gather pays (caches, translations, write-back) from what the batch's layout costs
        HIP_TRY(c, hipEventRecord(S->ev[0], s));
        if ((rc = call_batch_device_impl(c, &db, &dr, s)) != INQ_OK) return rc;
        HIP_TRY(c, hipEventRecord(S->ev[1], s));
        HIP_TRY(c, hipStreamSynchronize(s));
        float ms2 = 0.f;
        (void)hipEventElapsedTime(&ms2, S->ev[0], S->ev[1]);
        std::fprintf(stderr, "[inq call] the same %llu loci again: locus kernels %.3f ms\n", (unsigned long long)nl, (double)ms2);
    }
    if (dev) {  // the rows' places go up (4 B per locus, through the page-locked row staging), the rows stay where they are
        if ((rc = ensure(c, S->tmp, std::max<size_t>(nl * 4, 64))) != INQ_OK) return rc;
        std::memcpy(S->h_rows, dev->index, nl * 4);
        HIP_TRY(c, hipMemcpyAsync(S->tmp.p, S->h_rows, nl * 4, hipMemcpyHostToDevice, s));
        launch_scatter_rows(dr.phase1, dr.phase2, (const uint32_t *)S->tmp.p, dev->d1, dev->d2, nl, dev->cap, s);
        HIP_TRY(c, hipGetLastError());
    } else {
        HIP_TRY(c, hipMemcpyAsync(S->h_rows, dr.phase1, nl * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(S->h_rows + nl, dr.phase2, nl * 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipMemcpyAsync(&S->h->ks, c->d_status, sizeof(DevStatus), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemsetAsync(&c->d_status->err, 0, sizeof(unsigned int), s));
    HIP_TRY(c, hipMemsetAsync(&c->d_status->ties, 0, sizeof(unsigned long long), s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (!dev) {
        std::memcpy(r->phase1, S->h_rows, nl * 8);
        std::memcpy(r->phase2, S->h_rows + nl, nl * 8);
    }
    {
        bool much;
        {
            std::lock_guard<std::mutex> g(c->retired_mu);
            much = c->retired_bytes > (2ull << 30);
        }
        if (much) purge_retired(c);
    }
    if (ms_call) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, S->ev[4], S->ev[5]);
        *ms_call = (double)ms;
    }
    S->n_reads = n_reads_all, S->n_cigar_words = n_units_all * 4, S->n_pairs = n_pairs_all, S->n_loci = nl;
    S->last_from_acc = true;  // inq_span_fetch_batch reads the accumulated buffers (their contents stay until the next append)
    r->n_tie_loci = S->h->ks.ties;
    return status_to_code(S->h->ks.err);  // (`spent` puts the counters back to zero, the buffers stay)
}

// Runs on whatever host thread calls it, on the copy stream; touches only stage[slot] (and ctx->last_err on failure).
// begin: everything is ENQUEUED (compressed bytes, tables, the inflate behind them on the ahead stream) and the call returns; the
// caller's block table and anchors are copied to page-locked memory on the spot, the compressed bytes must stay until
// span_stage_wait(slot) has returned.  Two spans may be begun before the first is waited for: the copy engine then goes from one
// span's bytes straight to the next one's, instead of idling while the host learns that a copy is over and issues the next
// (0.25 - 0.35 ms per 268 MB span, measured: 5.17 ms per span in the loop against 4.8 ms for the copy alone).
int span_stage_begin_impl(inq_ctx *c, const inq_span_t *sp, int slot) {
    if (!c || !sp || slot < 0 || slot >= INQ_SPAN_SLOTS) return INQ_ERR_ARG;
    const uint64_t nb = sp->n_blocks, na = sp->n_anchors;
    const uint64_t u_bytes = nb ? sp->blocks[nb - 1].out_off + sp->blocks[nb - 1].isize : 0;
    int rc = check_blocks(sp->comp, sp->comp_bytes, sp->blocks, nb, u_bytes, true);
    if (rc != INQ_OK) return rc;
    if (na && (!sp->anchors || !sp->anchor_stop)) return INQ_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    SpanState *S;
    if ((rc = span_state(c, &S)) != INQ_OK) return rc;
    SpanState::Stage &g = S->stage[slot];
    if (g.pending) {  // begun and never waited for (a run that was torn down half-way): that upload is long over or about to be
        (void)hipEventSynchronize(g.ev_up);
        g.pending = false;
    }
    g.valid = false;
    constexpr size_t kPad = 64;
    hipStream_t s = S->copy_stream;
    const bool verbose = std::getenv("INQ_TIMING") && std::getenv("INQ_TIMING")[0] == '2';
    const auto w0 = std::chrono::steady_clock::now();
    auto wall = [&](const char *what) {
        if (verbose) std::fprintf(stderr, "[inq stage host] slot %d %-28s at %.2f ms\n", slot, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    };
    auto al16 = [](size_t x) { return (x + 15u) & ~(size_t)15u; };
    const size_t off_blocks = 0, off_anch = al16(nb * sizeof(inq_bgzf_block_t)), off_stop = off_anch + al16(na * 8), tab_bytes = off_stop + al16(na * 8);
    if ((rc = ensure(c, g.comp, sp->comp_bytes + kPad)) != INQ_OK) return rc;
    if ((rc = ensure(c, g.tab, tab_bytes + 16)) != INQ_OK) return rc;
    if (g.h_tab_cap < tab_bytes) {
        if (g.h_tab) (void)hipHostFree(g.h_tab);
        g.h_tab = nullptr, g.h_tab_cap = 0;
        const size_t want = std::max<size_t>(tab_bytes + tab_bytes / 2, 1u << 20);
        HIP_TRY(c, hipHostMalloc((void **)&g.h_tab, want, hipHostMallocDefault));
        g.h_tab_cap = want;
    }
    if (nb) std::memcpy(g.h_tab + off_blocks, sp->blocks, nb * sizeof(inq_bgzf_block_t));
    if (na) std::memcpy(g.h_tab + off_anch, sp->anchors, na * 8), std::memcpy(g.h_tab + off_stop, sp->anchor_stop, na * 8);
    // the views the rest of the code reads (no buffers of their own any more)
    g.blocks.p = (uint8_t *)g.tab.p + off_blocks, g.blocks.cap = 0;
    g.anchors.p = (uint8_t *)g.tab.p + off_anch, g.anchors.cap = 0;
    g.anchor_stop.p = (uint8_t *)g.tab.p + off_stop, g.anchor_stop.cap = 0;
    // From here on the copy engine may be reading the caller's span buffer and this slot's table: an error exit must not hand them back
    // while it does (the slot stays invalid, `pending` is not raised, so nobody will wait for this upload later).
    struct DrainOnError {
        hipStream_t s;
        bool armed = true;
        ~DrainOnError() {
            if (armed) (void)hipStreamSynchronize(s);
        }
    } drain{s};
    if (sp->comp_bytes) HIP_TRY(c, hipMemcpyAsync(g.comp.p, sp->comp, sp->comp_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemsetAsync((uint8_t *)g.comp.p + sp->comp_bytes, 0, kPad, s));
    if (tab_bytes) HIP_TRY(c, hipMemcpyAsync(g.tab.p, g.h_tab, tab_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipEventRecord(g.ev_up, s));
    wall("copies enqueued");
    g.inflated = false;
    if (c->inflate_ahead && nb) {
        // the inflate behind the upload, on the ahead stream (the copy stream goes on with the next span's bytes)
        if ((rc = ensure(c, g.u, u_bytes + kPad)) != INQ_OK) return rc;
        hipStream_t sa = S->ahead_stream;
        HIP_TRY(c, hipStreamWaitEvent(sa, g.ev_up, 0));
        HIP_TRY(c, hipMemsetAsync((uint8_t *)g.u.p + u_bytes, 0, kPad, sa));
        HIP_TRY(c, hipMemsetAsync(g.d_err, 0, sizeof(unsigned int), sa));
        HIP_TRY(c, hipEventRecord(g.ev_inf0, sa));
        InflateArgs ia;
        ia.comp = (const uint8_t *)g.comp.p;
        ia.comp_bytes = sp->comp_bytes;
        ia.blocks = (const inq_bgzf_block_t *)g.blocks.p;
        ia.n_blocks = nb;
        ia.out = (uint8_t *)g.u.p;
        ia.out_bytes = u_bytes;
        ia.block_status = nullptr;
        ia.err = g.d_err;
        ia.verify_crc = c->verify_crc ? 1u : 0u;
        ia.debug_flags = 0u;
        ia.algo = c->inflate_algo;
        ia.lit_pairs = c->inflate_lit_pairs < 0 ? inflate_wants_literal_pairs(sp->comp, sp->comp_bytes, sp->blocks, nb) : (uint32_t)c->inflate_lit_pairs;
        ia.tokens = nullptr;
        if (ia.algo != 1u && (c->inflate_tokens < 0 ? ia.lit_pairs == 0u : c->inflate_tokens != 0)) {
            if ((rc = ensure(c, g.tok, inflate_token_words(nb) * 4)) != INQ_OK) return rc;
            ia.tokens = (uint32_t *)g.tok.p;
        }
        launch_bgzf_inflate(ia, sa);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(g.ev_inf1, sa));
        g.inflated = true;
        wall("inflate enqueued");
    }
    g.host_comp = sp->comp;
    g.comp_bytes = sp->comp_bytes;
    g.n_blocks = nb;
    g.n_anchors = na;
    g.pending = true;
    drain.armed = false;
    return INQ_OK;
}

// the upload of the slot's span is over: the caller's compressed bytes are free again (the inflate, if any, goes on)
int span_stage_wait_impl(inq_ctx *c, int slot) {
    if (!c || slot < 0 || slot >= INQ_SPAN_SLOTS) return INQ_ERR_ARG;
    SpanState *S;
    int rc;
    if ((rc = span_state(c, &S)) != INQ_OK) return rc;
    SpanState::Stage &g = S->stage[slot];
    if (!g.pending) return INQ_ERR_ARG;
    g.pending = false;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(g.ev_up));
    g.valid = true;
    return INQ_OK;
}

int fetch_batch_impl(inq_ctx *c, uint32_t *cigar, inq_read_t *reads, uint32_t *pair_read, uint64_t *locus_pair_off) {
    if (!c || !c->span) return INQ_ERR_ARG;
    SpanState *S = c->span;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const bool acc = S->last_from_acc;
    if (cigar && S->n_cigar_words) HIP_TRY(c, hipMemcpy(cigar, acc ? S->acc.cigar.p : S->cigar.p, S->n_cigar_words * 4, hipMemcpyDeviceToHost));
    if (reads && S->n_reads) HIP_TRY(c, hipMemcpy(reads, acc ? S->acc.reads.p : S->reads.p, S->n_reads * sizeof(inq_read_t), hipMemcpyDeviceToHost));
    if (pair_read && S->n_pairs) HIP_TRY(c, hipMemcpy(pair_read, acc ? S->acc.pair_read.p : S->pair_read.p, S->n_pairs * 4, hipMemcpyDeviceToHost));
    if (locus_pair_off && S->n_loci) HIP_TRY(c, hipMemcpy(locus_pair_off, acc ? S->acc.off.p : S->locus_off.p, (S->n_loci + 1) * 8, hipMemcpyDeviceToHost));
    return INQ_OK;
}

}  // namespace

extern "C" {

int inq_bgzf_inflate(inq_ctx_t *c, const uint8_t *comp, uint64_t comp_bytes, const inq_bgzf_block_t *blocks, uint64_t n_blocks,
                     uint8_t *out, uint64_t out_bytes, uint32_t *block_status) {
    try {  // nothing may unwind across the C ABI
        return bgzf_inflate_impl(c, comp, comp_bytes, blocks, n_blocks, out, out_bytes, block_status);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_span_stage_begin(inq_ctx_t *c, const inq_span_t *span, int slot) {
    try {
        return span_stage_begin_impl(c, span, slot);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_span_stage_wait(inq_ctx_t *c, int slot) {
    try {
        return span_stage_wait_impl(c, slot);
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_span_stage(inq_ctx_t *c, const inq_span_t *span, int slot) {
    const int rc = inq_span_stage_begin(c, span, slot);
    return rc != INQ_OK ? rc : inq_span_stage_wait(c, slot);
}

int inq_call_span_staged(inq_ctx_t *c, const inq_span_t *span, int slot, inq_result_t *result, inq_span_stats_t *stats) {
    try {
        return slot < 0 ? INQ_ERR_ARG : call_span_impl(c, span, result, stats, slot);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_call_span(inq_ctx_t *c, const inq_span_t *span, inq_result_t *result, inq_span_stats_t *stats) {
    try {
        return call_span_impl(c, span, result, stats, -1);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_call_span_deferred(inq_ctx_t *c, const inq_span_t *span, int slot, inq_span_stats_t *stats) {
    try {
        return call_span_impl(c, span, nullptr, stats, slot < 0 ? -1 : slot, true);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_call_flush(inq_ctx_t *c, inq_result_t *result, uint64_t n_loci, double *ms_call) {
    try {
        return call_flush_impl(c, result, n_loci, ms_call);
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

int inq_call_flush_device(inq_ctx_t *c, double *d_phase1, double *d_phase2, uint64_t cap, const uint32_t *index, uint64_t n_loci, uint64_t *n_tie_loci,
                          double *ms_call) {
    try {
        inq_result_t r;
        std::memset(&r, 0, sizeof r);
        const FlushToDevice dev{d_phase1, d_phase2, index, cap};
        const int rc = call_flush_impl(c, &r, n_loci, ms_call, &dev);
        if (n_tie_loci) *n_tie_loci = r.n_tie_loci;
        return rc;
    } catch (const std::bad_alloc &) {
        return INQ_ERR_NOMEM;
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

// device memory for a caller that does not link the runtime itself (the host library's device-resident row arrays)
int inq_dev_alloc_rows(inq_ctx_t *c, uint64_t n, double **out) {
    if (!c || !out) return INQ_ERR_ARG;
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    void *p = nullptr;
    HIP_TRY(c, hipMalloc(&p, std::max<size_t>((size_t)n * 8, 64)));
    launch_fill_f64((double *)p, n, __builtin_nan(""), c->stream);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(p);
        return INQ_ERR_HIP;
    }
    *out = (double *)p;
    return INQ_OK;
}
void inq_dev_free_rows(inq_ctx_t *c, double *p) {
    if (!c || !p) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(p);
}
int inq_dev_write_rows(inq_ctx_t *c, double *dst, const double *src_host, uint64_t n) {
    if (!c || (n && (!dst || !src_host))) return INQ_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n) HIP_TRY(c, hipMemcpy(dst, src_host, (size_t)n * 8, hipMemcpyHostToDevice));
    return INQ_OK;
}
int inq_dev_read_rows(inq_ctx_t *c, double *dst_host, const double *src, uint64_t n) {
    if (!c || (n && (!dst_host || !src))) return INQ_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n) HIP_TRY(c, hipMemcpy(dst_host, src, (size_t)n * 8, hipMemcpyDeviceToHost));
    return INQ_OK;
}

uint64_t inq_call_deferred_loci(const inq_ctx_t *c) { return c && c->span ? c->span->acc.n_loci : 0; }

void inq_call_discard(inq_ctx_t *c) {
    if (!c || !c->span) return;
    SpanState::Acc &A = c->span->acc;
    A = SpanState::Acc{A.cigar, A.reads, A.pair_read, A.off, A.lstart, A.lend};
}

int inq_span_fetch_batch(inq_ctx_t *c, uint32_t *cigar, inq_read_t *reads, uint32_t *pair_read, uint64_t *locus_pair_off) {
    try {
        return fetch_batch_impl(c, cigar, reads, pair_read, locus_pair_off);
    } catch (...) {
        return INQ_ERR_HIP;
    }
}

}  // extern "C"
