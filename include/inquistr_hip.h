/*
 * inquistr_hip.h — C ABI of the MI355X-native `inquiSTR call` hot path.
 *
 * What this boundary replaces in the reference (wdecoster/inquiSTR @ v0.13.0):
 * the reference has no FFI; the per-locus seam is
 *     genotype_repeat_phased(&mut IndexedReader, RepeatInterval, minlen, support)   src/call.rs:329-374
 *     genotype_repeat_unphased(...)                                                  src/call.rs:279-327
 * called once per locus from the serial loop (src/call.rs:150-157) or from a rayon
 * worker (src/call.rs:115-136).  A per-locus call cannot feed a GPU, so the ABI is
 * batch level: the host decodes BAM records (rust-htslib in the reference, the C++
 * front end in this repo), packs them into the buffers below and asks for the two
 * per-haplotype medians of every locus in one call.  Everything from "is this
 * record yielded by fetch()" (src/call.rs:288,338) through call_from_cigar
 * (src/call.rs:377-413) to median_str_length (src/call.rs:497-522) runs on the GPU.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Caller owns every buffer;
 * the library keeps no pointer after a call returns.
 */
#ifndef INQUISTR_HIP_H
#define INQUISTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INQ_ABI_VERSION 5

/* ---- error codes (0 = ok, negative = failure; never throws / aborts) ---- */
enum {
    INQ_OK = 0,
    INQ_ERR_ARG = -1,          /* NULL pointer, inconsistent sizes, non-monotone CSR               */
    INQ_ERR_SUPPORT_ZERO = -2, /* support == 0: reference underflows `len/2 - 1` (src/call.rs:516) */
    INQ_ERR_PHASE = -3,        /* a read that passes the phased filter has HP > 2:
                                  reference `calls.get_mut(&phase).unwrap()` panics (src/call.rs:358) */
    INQ_ERR_CIGAR_OP = -4,     /* CIGAR op code > 8: rust-htslib's cigar() panics (src/call.rs:382)  */
    INQ_ERR_LOCUS = -5,        /* locus start < 10 (u32 underflow, src/call.rs:285,335) or end < start
                                  (src/repeats.rs:102-104)                                           */
    INQ_ERR_RANGE = -6,        /* pos + reference span of a read does not fit 31 bits (not a valid BAM) */
    INQ_ERR_INDEX = -7,        /* pair_read[] or a read's CIGAR extent points outside the buffers    */
    INQ_ERR_HIP = -8,          /* HIP runtime failure; see inq_last_error()                         */
    INQ_ERR_NOMEM = -9,
    INQ_ERR_NO_DEVICE = -10,   /* no gfx950 device visible: the library has NO CPU fallback          */
    INQ_ERR_INFLATE = -11,     /* device front end: a BGZF block does not inflate to its ISIZE (htslib: read error,
                                  `r.expect("Error reading BAM file")` src/call.rs:295,346 panics)     */
    INQ_ERR_BAM = -12,         /* device front end: record chain / field lengths corrupt, or records of the
                                  contig not coordinate-sorted                                          */
    INQ_ERR_AUX = -13          /* an aux field the reference panics on: HP of a FETCHED read is neither `C`
                                  nor `i` in phased mode (get_phase runs before the filter, src/call.rs:349,
                                  482-491; device front end), or a KEPT read carries INQ_READ_SA_PANIC
                                  (is_accidental_2d is only reached from call_from_cigar, i.e. for reads
                                  that passed the filter: src/call.rs:303,357 -> :394 -> :429-451)        */
};

/* ---- read descriptor: one 16-byte record per decoded BAM record ---------
 * Replaces rust-htslib bam::Record accessors used on the path:
 *   reference_start() src/call.rs:297,351,380   -> pos
 *   mapq()            src/call.rs:299,352       -> mapq
 *   cigar()           src/call.rs:382           -> cigar_off4 / n_cigar into inq_batch_t.cigar
 *   aux(b"HP")        src/call.rs:482-491       -> bits&INQ_READ_HAS_HP, phase (value `as u8`)
 *   is_accidental_2d  src/call.rs:415-459       -> bits&INQ_READ_IS_2D / INQ_READ_SA_PANIC (the SA string is evaluated
 *                                                  by whoever builds the descriptors: host sweep or cigar_gather)
 *   flag 0x4          (bam_endpos rule)         -> bits&INQ_READ_UNMAPPED
 */
typedef struct inq_read {
    uint32_t cigar_off4; /* first CIGAR word of this read, in units of 4 words (16 bytes)     */
    uint32_t n_cigar;    /* number of CIGAR ops                                               */
    int32_t pos;         /* BAM core.pos, 0-based leftmost                                    */
    uint8_t mapq;
    uint8_t bits;        /* INQ_READ_* */
    uint8_t phase;       /* HP value truncated to u8; meaningful iff INQ_READ_HAS_HP          */
    uint8_t reserved;    /* must be 0                                                         */
} inq_read_t;

#define INQ_READ_UNMAPPED 0x01u /* BAM flag 0x4   */
#define INQ_READ_REVERSE 0x02u  /* BAM flag 0x10  */
#define INQ_READ_HAS_HP 0x04u   /* HP aux present */
#define INQ_READ_IS_2D 0x08u    /* is_accidental_2d(record) == true */
#define INQ_READ_SA_PANIC 0x10u /* the CIGAR has an S op AND is_accidental_2d(record) would panic (SA aux not `Z`,
                                   no entry, < 4 fields, POS or CIGAR unparsable: src/call.rs:429-451,469).  The
                                   reference only gets there for a read that passed the filter (:303,357), so the
                                   device raises INQ_ERR_AUX iff such a read is KEPT; a filtered-out one is ignored */

/* ---- one batch of loci ----------------------------------------------------
 * cigar   : BAM-native packed ops, `len << 4 | op`, op in 0..8 = MIDNSHP=X, all reads
 *           concatenated.  Every read starts on a 4-word boundary; the 0..3 words of
 *           padding behind a read must be 0 (`0M`, a no-op for every rule on the path).
 *           n_cigar_words counts the padding and is a multiple of 4.
 * pairs   : CSR over loci.  pair_read[locus_pair_off[j] .. locus_pair_off[j+1]) are the
 *           reads offered to locus j IN FILE ORDER (the order rc_records() yields them,
 *           src/call.rs:294,345).  The list may be a superset of what fetch() would
 *           yield: the device applies htslib's overlap rule itself
 *           (pos < end_ext && bam_endpos > start_ext), so extra candidates change nothing.
 * loci    : un-extended BED coordinates (src/repeats.rs:75-79); the ±10 extension
 *           (src/call.rs:285-286,335-336) is applied on the device.
 */
typedef struct inq_batch {
    uint64_t n_reads;
    uint64_t n_cigar_words;
    uint64_t n_pairs;
    uint64_t n_loci;
    const uint32_t *cigar;          /* [n_cigar_words], 16-byte aligned */
    const inq_read_t *reads;        /* [n_reads]                        */
    const uint32_t *pair_read;      /* [n_pairs] index into reads       */
    const uint64_t *locus_pair_off; /* [n_loci + 1]                     */
    const uint32_t *locus_start;    /* [n_loci]                         */
    const uint32_t *locus_end;      /* [n_loci]                         */
    uint32_t minlen;                /* -m, src/main.rs:43               */
    uint32_t support;               /* -s, src/main.rs:47  (>= 1)       */
    uint32_t unphased;              /* -u, src/main.rs:55  (0 / 1)      */
    uint32_t reserved;              /* must be 0                        */
} inq_batch_t;

/* ---- results --------------------------------------------------------------
 * phase1/phase2 : Genotype.phase1/.phase2 (src/call.rs:27-31): exact integers or
 *                 halves, quiet NaN where the reference returns NAN (src/call.rs:499).
 * pair_call     : optional (may be NULL) per-pair Call value (src/call.rs:67-71)
 * pair_bits     : optional (may be NULL) per-pair INQ_PAIR_* bits
 * n_tie_loci    : unphased only: loci whose median split (src/call.rs:312-314) cuts
 *                 through equal values of mixed Span/Clip, where Rust's unstable sort
 *                 makes the reference itself ambiguous; this library orders ties by
 *                 file order (exact for <= 20 reads).
 */
typedef struct inq_result {
    double *phase1;     /* [n_loci] */
    double *phase2;     /* [n_loci] */
    int64_t *pair_call; /* [n_pairs] or NULL */
    uint8_t *pair_bits; /* [n_pairs] or NULL */
    uint64_t n_tie_loci;
} inq_result_t;

#define INQ_PAIR_CLIP 0x01u    /* Call::Clip (a soft clip was counted)            */
#define INQ_PAIR_FETCHED 0x02u /* yielded by fetch(): pos < end_ext && endpos > start_ext */
#define INQ_PAIR_KEPT 0x04u    /* survived the read filter (src/call.rs:297-302 / 349-355) */

typedef struct inq_ctx inq_ctx_t;

/* Opens HIP device `device_id` (must be gfx950), creates the library's stream and
 * scratch.  One ctx per device; a ctx serves one caller at a time. */
int inq_ctx_create(int device_id, inq_ctx_t **out);
/* The same in a hurry: *ctx is set and *stage_ready raised (release store) as soon as inq_span_stage / _begin / _wait may be called
 * on it from another thread, ~30 ms before the function returns; every other entry point only after it has returned INQ_OK.  A
 * context that was published stays valid until inq_ctx_destroy, whatever the return value (the caller destroys it). */
int inq_ctx_create_early(int device_id, inq_ctx_t **ctx, volatile int *stage_ready);
/* The list form ("inq_ctx_create(device_ids, n, &ctx)"; the reference's counterpart is rayon's pool of workers,
 * src/call.rs:104-118): one context per entry of device_ids - an ordinal may repeat -, made concurrently (each on a thread of
 * its own: the runtime's start-up is paid once, the per-device parts side by side); ctxs[0 .. n) receive them.  All or nothing: on
 * failure every context that was made is destroyed again, ctxs[] is all NULL and the first failing entry's code is returned.  Each
 * context is then driven by its own host thread (a ctx serves one caller at a time; distinct ctxs are independent). */
int inq_ctx_create_multi(const int *device_ids, int n, inq_ctx_t **ctxs);
void inq_ctx_destroy(inq_ctx_t *ctx);

/* Host-buffer entry: batch and result point to HOST memory (pinned for best H2D).
 * Validates the batch, uploads, runs the kernels, downloads, synchronises. */
int inq_call_batch(inq_ctx_t *ctx, const inq_batch_t *batch, inq_result_t *result);

/* Device-resident entry: every pointer in batch/result is a DEVICE pointer on the
 * ctx's device; hip_stream is a hipStream_t (NULL = the ctx's own stream).  Enqueues
 * only — no host synchronisation, no allocation.  result->n_tie_loci is not written;
 * fetch it, and any domain error the kernels flagged, with inq_ctx_status() after
 * the stream has been synchronised. */
int inq_call_batch_device(inq_ctx_t *ctx, const inq_batch_t *batch, inq_result_t *result,
                          void *hip_stream);

/* Device-side status of the launches since the last inq_ctx_status() call: returns
 * INQ_OK or the first domain error (INQ_ERR_PHASE / _CIGAR_OP / _LOCUS / _RANGE /
 * _INDEX) and the tie count.  Synchronises the device. */
int inq_ctx_status(inq_ctx_t *ctx, uint64_t *n_tie_loci);

/* Kernel timing with HIP events on the launch stream (for bench.py's roofline block).
 * which: 0 = whole inq_call_batch_device launch sequence, 1 = the CIGAR-walk kernel,
 * 2 = the BGZF inflate kernel of the last inq_bgzf_inflate / inq_call_span (launches = 1).
 * Returns accumulated milliseconds and launch count since the last reset. */
int inq_ctx_timing_enable(inq_ctx_t *ctx, int on);
int inq_ctx_timing_read(inq_ctx_t *ctx, int which, double *total_ms, uint64_t *launches);
int inq_ctx_timing_reset(inq_ctx_t *ctx);

/* Tuning knobs.  key: "grid_medium" = workgroups of the kernel that takes the 65..256-read loci and walks the deeper ones
 * (default 4096); "grid_tail" ("grid_big" up to ABI v4) = workgroups of the persistent kernel that reduces the deeper ones (default
 * 256, never more than the device's compute units: they meet at grid barriers); "max_reads_hint" = N > 0 promises that no locus of
 * the following batches is offered more than N reads (N <= 64 skips the two launches behind the first kernel - which cost a few
 * microseconds when nothing deep is there; a violated promise is reported as INQ_ERR_ARG), 0 (default) = unknown;
 * "verify_crc" = 0 skips the CRC32 check of the device front end (default 1);
 * "inflate_algo" = 0 inflates with one workgroup per BGZF block (no latency floor, 0.54-0.82 ms per 1000 blocks), 1 with one
 * lane per block (36-56 ms for up to ~65 000 blocks), 2 (default) = the quicker one, which is 0 at every size measured;
 * "inflate_lit_pairs" = 1 / 0: the workgroup inflate's symbol loop decodes a second literal from the same 32 bits as the first
 * (+18 - 24 % on sequence / quality bytes) or not (CIGAR-only records lose 4.6 % to the wasted look); -1 (default) = decided per
 * call from the code lengths in a few sampled block headers;
 * "inflate_tokens" = 1: the workgroup inflate keeps the symbols its counting passes decode (32 KB of device scratch per BGZF
 * block) so that its commit step does not decode them again (+3 - 4 % on match-heavy blocks, -2 ... -5 % on sequence / quality
 * bytes, and four times the HBM traffic of the inflate: the stores cost what the second decode did), 0 (default since round 4) =
 * it decodes again, -1 = on for spans whose sampled block headers say match-heavy (round 3's default);
 * "blocking_sync" = 1: every wait of this context gives its core back (hipDeviceScheduleBlockingSync; as a process default - set
 *   before the context is made - also a blocking upload event) instead of spinning.  Default 0: spinning costs two cores (the caller's
 *   thread during a span's inflate, the uploader's during its copy) and answers a few microseconds sooner; libinquistr_host.so sets 1
 *   by itself when a caller's share of the granted cores is below 8 (eight ranks on a 16-core grant), where those two cores are the
 *   readers'.
 * "inflate_ahead" = 1 (default since round 4): inq_span_stage inflates a span right behind its upload, on a stream of its own, so
 * that the inflate of span k + 1 runs next to span k's record scan, gather and join (costs one inflated buffer per staging slot:
 * device allocations cost microseconds, tools/alloc_probe.hip); 0 = the span is inflated when it is called;
 * "gather_nt" = 1 (default) / 0: the device front end's CIGAR gather stores with the non-temporal policy (the batch it builds is
 * read by a later launch); "batch_loci_hint" = N: inq_call_span_deferred sizes the batch's buffers for N loci from its first span;
 * "retired_limit_mb" / "test_fail_allocs": see inq_ctx_alloc_retries;
 * "nt_loads" = 1 / 0 forces the non-temporal cache policy for the CIGAR stream on / off; -1 (default)
 * picks it when no read is shared between loci (n_pairs <= n_reads). */
int inq_ctx_set_option(inq_ctx_t *ctx, const char *key, int64_t value);
/* How often a device allocation of this context met "out of memory", gave the buffers it had outgrown and parked back (options
 * "retired_limit_mb": how many MB may be parked, default 16384) and tried again - and, for the deferred batch's speculative
 * reserve, fell back to the bytes really needed.  ("test_fail_allocs" = N makes the next N first attempts fail: the test seam.) */
uint64_t inq_ctx_alloc_retries(const inq_ctx_t *ctx);
/* The same for every context made FROM NOW ON in this process (any thread): key and value are checked at once, the option is applied
 * inside inq_ctx_create / _early / _multi before the context is handed out.  For a host that does not make the contexts itself
 * (libinquistr_host.so does: `inquistr call --ctx-option key=value`).  The library reads NO option from the environment. */
int inq_default_option(const char *key, int64_t value);
/* INQ_OK and *value if the process default of `key` has been set (by inq_default_option), INQ_ERR_ARG otherwise: lets a host library
 * choose a default of its own ("blocking_sync" when its share of the cores is small) without overriding what its user asked for. */
int inq_default_option_get(const char *key, int64_t *value);

/* Page-locks caller-owned host memory in place (hipHostRegister: 0.5 ms for 268 MB of huge-page-backed memory, against 50 - 60 ms
 * for inq_alloc_pinned of the same size), so that copies from it are plain DMA.  The runtime must be up (a ctx exists). */
int inq_pin_host(void *p, size_t bytes);
void inq_unpin_host(void *p);

/* Pinned host allocations for batch buffers. */
int inq_alloc_pinned(size_t bytes, void **out);
void inq_free_pinned(void *p);


/* ==== device front end: BGZF inflate + BAM record scan + overlap join on the GPU =====================
 * Replaces, for one span of the file, what `bam.fetch((tid, start_ext, end_ext))` + `rc_records()` and the
 * record accessors do per locus in the reference (src/call.rs:288,294,297-299,338,345,351-352; [3P]
 * htslib bgzf.c / sam.c / hts.c): the host only reads the compressed bytes and walks the 18-byte BGZF
 * headers and the .bai; inflating, finding the records, decoding their fixed fields and HP / SA / CG
 * tags, and matching reads to loci with htslib's overlap rule all happen on the device, and the batch
 * the locus kernels consume never exists in host memory.
 */
typedef struct inq_bgzf_block {
    uint64_t comp_off; /* offset of the block's DEFLATE payload (behind its gzip header) in `comp`    */
    uint32_t comp_len; /* payload bytes, without the 8-byte CRC32 / ISIZE trailer                      */
    uint32_t isize;    /* inflated size from the trailer (<= 65536)                                    */
    uint64_t out_off;  /* where the block's data goes in the inflated byte string                      */
} inq_bgzf_block_t;

/* per-block inflate status bits (0 = the block inflated to exactly isize bytes) */
#define INQ_INFLATE_BAD_HEADER 0x01u    /* reserved block type, bad code-length set, extents outside the buffers */
#define INQ_INFLATE_BAD_CODE 0x02u      /* a bit pattern that is no code of the current Huffman set            */
#define INQ_INFLATE_INPUT_OVERRUN 0x04u /* the stream needs more bits than the payload holds                   */
#define INQ_INFLATE_OUTPUT_SIZE 0x08u   /* more or fewer bytes than isize                                      */
#define INQ_INFLATE_BAD_DISTANCE 0x10u  /* a match reaches in front of the block                               */
#define INQ_INFLATE_BAD_STORED 0x20u    /* stored block LEN / NLEN mismatch                                    */
#define INQ_INFLATE_BAD_CRC 0x40u       /* inflated bytes do not match the CRC32 of the block's trailer        */

/* Inflates n_blocks BGZF payloads; every pointer is HOST memory (the call uploads, runs one workgroup or one
 * lane per block - ctx option "inflate_algo" -, downloads, synchronises).  block_status may be NULL.  Returns INQ_ERR_INFLATE if any block
 * failed.  `comp` holds whole BGZF blocks: the 8 bytes behind every payload are its CRC32 / ISIZE trailer,
 * and the inflated bytes are checked against that CRC32 as htslib does (ctx option "verify_crc", default 1). */
int inq_bgzf_inflate(inq_ctx_t *ctx, const uint8_t *comp, uint64_t comp_bytes, const inq_bgzf_block_t *blocks,
                     uint64_t n_blocks, uint8_t *out, uint64_t out_bytes, uint32_t *block_status);

/* One span: whole BGZF blocks of a coordinate-sorted BAM, as one or more SEGMENTS (runs of consecutive
 * blocks, ascending and disjoint in the file), plus loci whose overlapping records all start inside those
 * segments.  Segments let one call serve loci that are far apart without inflating what lies between. */
#define INQ_ANCHOR_SEGMENT_END (1ull << 63)
typedef struct inq_span {
    const uint8_t *comp;             /* HOST (pageable is fine): the compressed bytes of the blocks,
                                        segment after segment, each payload followed by its trailer      */
    uint64_t comp_bytes;
    const inq_bgzf_block_t *blocks;  /* file order; out_off dense and ascending from 0                   */
    uint64_t n_blocks;
    const uint64_t *anchors;         /* ascending offsets into the inflated bytes at which a BAM record
                                        is known to start (virtual offsets of the .bai: chunk begins and
                                        linear-index entries); every segment opens with one.  Records
                                        are found by following block_size from each anchor, one lane per
                                        anchor.                                                           */
    const uint64_t *anchor_stop;     /* [n_anchors] where chain i ends: the next anchor (the chain must
                                        land on it), or `end of the segment | INQ_ANCHOR_SEGMENT_END`
                                        (a record cut by the segment's end is dropped)                    */
    uint64_t n_anchors;
    const int32_t *locus_tid;        /* [n_loci] header().tid(chrom), src/call.rs:287,337                 */
    const uint32_t *locus_start;     /* [n_loci] un-extended BED coordinates, any order                   */
    const uint32_t *locus_end;
    uint64_t n_loci;
    uint32_t minlen, support, unphased, reserved;
} inq_span_t;

typedef struct inq_span_stats {
    uint64_t n_records;      /* records found in the span                                   */
    uint64_t n_reads;        /* of which placed on a contig (unplaced records close the file) */
    uint64_t n_pairs;        /* (locus, read) pairs = records fetch() would yield, summed    */
    uint64_t n_cigar_words;  /* gathered CIGAR words incl. padding                           */
    uint64_t inflated_bytes;
    uint32_t max_reads;      /* deepest locus                                                */
    uint32_t front_status;   /* raw status bits of the scan kernels (diagnostics)            */
    uint64_t first_bad_record;
    double ms_upload, ms_inflate, ms_scan, ms_join, ms_call; /* HIP-event times of the stages */
} inq_span_stats_t;

/* Runs the whole path for one span; result->phase1/phase2 (HOST, [n_loci]) receive the rows in the order
 * of locus_start/locus_end; pair_call / pair_bits are not produced (pass NULL).  stats may be NULL. */
int inq_call_span(inq_ctx_t *ctx, const inq_span_t *span, inq_result_t *result, inq_span_stats_t *stats);

/* Two-step form for files of many spans: inq_span_stage uploads the compressed bytes, the block table and the
 * anchors of a span into one of INQ_SPAN_SLOTS device-side slots (0 .. 7) on the library's copy stream and returns when they are
 * there; inq_call_span_staged then runs the span from that slot without uploading.  inq_span_stage may be
 * called from another host thread while an inq_call_span / inq_call_span_staged of an EARLIER span is in
 * progress on the same ctx (different slot): the upload of span k+1 then overlaps the inflate of span k.
 * A slot may be staged again once the call that used it has returned.  span->comp must be the same pointer
 * and sizes in both calls. */
#define INQ_SPAN_SLOTS 8 /* two sets of four: a session stages the next file's spans while this file's are called */
int inq_span_stage(inq_ctx_t *ctx, const inq_span_t *span, int slot);
/* The same in two steps, so that the copy engine goes from one span's bytes straight to the next one's: _begin enqueues the
 * upload (and the inflate behind it) and returns - the block table and the anchors are copied on the spot, span->comp must stay
 * until _wait(slot) has returned; a second span may be begun (other slot) before the first is waited for.  inq_span_stage =
 * _begin + _wait. */
int inq_span_stage_begin(inq_ctx_t *ctx, const inq_span_t *span, int slot);
int inq_span_stage_wait(inq_ctx_t *ctx, int slot);
int inq_call_span_staged(inq_ctx_t *ctx, const inq_span_t *span, int slot, inq_result_t *result,
                         inq_span_stats_t *stats);

/* Deferred form, for spans that hold too few loci to fill the chip on their own (a span of 256 MB of SEQ / QUAL-bearing records
 * holds under a thousand loci; the locus kernels want tens of thousands per launch): inq_call_span_deferred does everything
 * inq_call_span / inq_call_span_staged (slot >= 0) does up to the overlap join and APPENDS the span's batch - CIGARs, read
 * descriptors, pairs, loci - to a batch that stays on the device; inq_call_flush runs the locus kernels once over everything
 * appended since the last flush and returns the rows in the order the loci were appended (n_loci must be their number;
 * ms_call, if not NULL, receives the HIP-event time of the launch sequence).  minlen / support / unphased must be the same in
 * every span of one batch.  The reference's counterpart is still the per-locus loop of src/call.rs:115-136,150-157: which loci
 * share a launch is not observable in the rows. */
int inq_call_span_deferred(inq_ctx_t *ctx, const inq_span_t *span, int slot /* -1: not staged */, inq_span_stats_t *stats);
int inq_call_flush(inq_ctx_t *ctx, inq_result_t *result, uint64_t n_loci, double *ms_call);
/* inq_call_flush with the rows left ON THE DEVICE: row j of the flush goes to d_phase1[index[j]], d_phase2[index[j]] (DEVICE arrays of
 * `cap` entries on the ctx's device; index: HOST, n_loci entries, every one < cap), only the tie count and the status come back.  For a
 * caller whose rows travel on from device memory - one process per GPU: the RCCL gather of the per-shard rows to rank 0 (north_star;
 * inquistr_amd/call_dist.py) reads them where the kernels' rows already are. */
int inq_call_flush_device(inq_ctx_t *ctx, double *d_phase1, double *d_phase2, uint64_t cap, const uint32_t *index, uint64_t n_loci,
                          uint64_t *n_tie_loci, double *ms_call);
/* Row arrays in device memory for a host that does not link the HIP runtime itself: n f64, filled with quiet NaN (a locus no span
 * holds prints NaN NaN); plain synchronous copies in and out. */
int inq_dev_alloc_rows(inq_ctx_t *ctx, uint64_t n, double **out);
void inq_dev_free_rows(inq_ctx_t *ctx, double *p);
int inq_dev_write_rows(inq_ctx_t *ctx, double *dst_device, const double *src_host, uint64_t n);
int inq_dev_read_rows(inq_ctx_t *ctx, double *dst_host, const double *src_device, uint64_t n);
uint64_t inq_call_deferred_loci(const inq_ctx_t *ctx); /* loci appended since the last flush */
void inq_call_discard(inq_ctx_t *ctx); /* forgets what was appended (a run that failed between two flushes; a flush, failed or not, does it too) */

/* Test / debug: copies the batch the last inq_call_span (or inq_call_flush) built on the device into caller-allocated HOST
 * arrays sized from that call's stats: cigar[n_cigar_words], reads[n_reads], pair_read[n_pairs],
 * locus_pair_off[n_loci + 1].  Any pointer may be NULL. */
int inq_span_fetch_batch(inq_ctx_t *ctx, uint32_t *cigar, inq_read_t *reads, uint32_t *pair_read, uint64_t *locus_pair_off);

/* ==== `inquiSTR outlier` (src/outlier.rs): outlying samples per locus of a combined .inq ====
 * values: HOST, row-major [n_rows][stride] f32, row i holding row_len[i] <= stride parsed numbers (NaN allowed:
 * get_repeat_lengths maps it to 0, src/outlier.rs:80-83).  method: z-score (std_deviation_and_mean + z_score_outliers,
 * :18-31, 97-110; f32 with the reference's sequential summation order) or DBSCAN (dbscan_outliers + mode, :112-145;
 * eps = max(2 * mode, 10), min points = mincluster, at most 8192 values per row).  flags[i][k] = 1 where
 * samples[k] is reported for row i; keep[i] tells what became of the row. */
#define INQ_OUTLIER_ZSCORE 0
#define INQ_OUTLIER_DBSCAN 1
#define INQ_OUTLIER_ROW_SKIP 0     /* max < minsize: get_repeat_lengths returns None, the row is not looked at      */
#define INQ_OUTLIER_ROW_KEEP 1
#define INQ_OUTLIER_ROW_EMPTY 2    /* no values: the reference panics (unwrap on None, src/outlier.rs:90)            */
#define INQ_OUTLIER_ROW_NO_MODE 3  /* DBSCAN, no value > 0: the reference panics ("No mode found for repeat", :144) */
#define INQ_OUTLIER_ROW_TOO_WIDE 4 /* DBSCAN row with more than 8192 values: not supported                           */
int inq_outlier_rows(inq_ctx_t *ctx, const float *values, const uint32_t *row_len, uint64_t n_rows, uint32_t stride,
                     int method, uint32_t minsize, float zscore_cutoff, uint32_t mincluster, uint8_t *flags,
                     uint8_t *keep);

const char *inq_strerror(int code);
const char *inq_backend_name(const inq_ctx_t *ctx); /* "hip:gfx950:<device name>" */
const char *inq_last_error(const inq_ctx_t *ctx);   /* detail of the last INQ_ERR_HIP */
int inq_abi_version(void);
/* NUMA node of the host the context's GPU hangs off (its PCI address through sysfs), -1 if unknown: a host buffer the GPU's
 * copy engine reads is best placed there (268 MB from the other socket: 5.6 - 6.1 ms instead of 4.9 ms on the boxes measured). */
int inq_ctx_numa_node(const inq_ctx_t *ctx);

#ifdef __cplusplus
}
#endif
#endif /* INQUISTR_HIP_H */
