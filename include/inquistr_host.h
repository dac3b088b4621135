/*
 * inquistr_host.h — C ABI of the host side of `inquiSTR call` (libinquistr_host.so).
 *
 * Mirrors the reference's one entry point for this path,
 *     call::genotype_repeats(bamp, region, region_file, minlen, support, threads, unphased,
 *                            sample_name, reference)                      src/call.rs:76-86
 * with the same argument meaning and the same observable behaviour: `.inq` text on the output
 * stream, diagnostics on stderr, and an exit status (0, 1 for the reference's explicit
 * `exit(1)` paths, 101 where the reference panics).  The hot path itself is delegated to
 * libinquistr_hip.so (include/inquistr_hip.h); there is no CPU implementation of it here.
 *
 * The front-end entry points below expose the BAM -> batch stage on its own (what
 * bam.fetch()/rc_records() + record accessors do in the reference, src/call.rs:288-299,338-352)
 * so that it can be tested without a GPU and driven by another host.
 */
#ifndef INQUISTR_HOST_H
#define INQUISTR_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "inquistr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* clap arguments of `inquiSTR call`, src/main.rs:27-64 */
typedef struct inq_call_args {
    const char *bam;         /* positional */
    const char *region;      /* -r, NULL if absent */
    const char *region_file; /* -R, NULL if absent */
    uint32_t minlen;         /* -m, default 5 */
    uint64_t support;        /* -s, default 3 */
    uint64_t threads;        /* -t, default 1: 1 = rows in BED order, >1 = rows sorted (src/call.rs:141) */
    int32_t unphased;        /* -u */
    const char *sample_name; /* --sample-name, NULL if absent */
    const char *reference;   /* --reference (CRAM only; CRAM is not supported here) */
    int32_t device;          /* HIP device ordinal (not a reference argument) */
    int32_t reserved;        /* front end: 0 = auto (env INQ_FRONTEND=host|device, else device when the loci need >= 1 MiB of BAM per host thread), 1 = host sweep, 2 = device spans */
} inq_call_args_t;

#define INQ_EXIT_OK 0
#define INQ_EXIT_ERROR 1   /* the reference's std::process::exit(1) paths          */
#define INQ_EXIT_PANIC 101 /* the reference's panic!/expect/unwrap paths          */

/* Runs the whole command; writes header + rows to out_fd.  Returns the exit status; a message
 * for non-zero statuses is copied to errbuf (and printed to stderr by the CLI).
 * Side effect worth knowing in a long-lived host process: the threads this call creates for reading, uploading and starting the
 * HIP runtime (and the helper threads the runtime spawns from the latter) are kept on the CPUs of the GPU's NUMA node and prefer
 * its memory (uploads run at the PCIe rate only from there); the calling thread is left alone.  INQ_NUMA_NODE=-1 in the
 * environment switches that off, INQ_NUMA_CPUS=0 keeps the memory preference only. */
int inq_genotype_repeats(const inq_call_args_t *args, int out_fd, char *errbuf, size_t errcap);

/* One process per GPU (inquistr_amd/call_dist.py; the reference's counterpart is the rayon loop over loci, src/call.rs:115-136,
 * whose workers share nothing but the output Vec): the same command restricted to the targets target_index[0 .. n_index)
 * (positions in the target list -r / -R give, in list order), rows as numbers instead of text: phase1[k], phase2[k] = the row
 * of target target_index[k] (NaN where the reference prints NaN).  Nothing is written anywhere.  Same exit statuses. */
int inq_genotype_repeats_rows(const inq_call_args_t *args, const uint32_t *target_index, uint64_t n_index, double *phase1, double *phase2,
                              char *errbuf, size_t errcap);
/* The work split for `world` such processes: order[0 .. *n_targets) = the targets in file order (contig of the BAM header,
 * start, end), cuts[0 .. world] = cut points into order[] chosen so that every part needs about the same number of
 * compressed BAM bytes (.bai linear index).  Part r = order[cuts[r] .. cuts[r + 1]).  No GPU involved. */
int inq_host_partition(const inq_call_args_t *args, uint64_t world, uint32_t *order, uint64_t order_cap, uint64_t *cuts, uint64_t *n_targets,
                       char *errbuf, size_t errcap);

/* ---- several GPUs of one node from ONE process (north_star: "Partition loci across the 8 GPUs of one node ... a trivial gather of
 * per-shard .inq rows"; the reference's counterpart is the rayon loop over loci, src/call.rs:103-145).  The command of
 * inq_genotype_repeats on the HIP devices device_ids[0 .. n_devices): the targets, in file order, are cut into n_devices contiguous
 * parts of about equal compressed BAM bytes (as inq_run_partition), part r runs whole - its own reader pool, uploads, spans, locus
 * kernels - on a thread and a device context of its own on device_ids[r], the rows land in this process's arrays (the gather is a
 * scatter in host memory: no collective, no torch) and the calling thread writes the ordered .inq to out_fd.  An ordinal may appear
 * more than once (a rehearsal of N parts on one GPU).  args->device is ignored.  stats (may be NULL): [n_devices], one entry per part.
 * Byte for byte the output of inq_genotype_repeats; same exit statuses (the first failing part's, its message prefixed "part r of N").
 * The host's cores are dealt among the parts: each reader pool takes granted cores / n_devices threads (at least 2). */
typedef struct inq_part_stats {
    int32_t device;          /* device_ids[r]                                                          */
    int32_t status;          /* this part's exit status                                                */
    uint64_t loci;           /* targets of the part                                                    */
    uint64_t spans;          /* spans its device front end was handed (0: host sweep, or nothing to read) */
    uint64_t bam_bytes_read; /* compressed BAM bytes of those spans                                    */
    double rows_s;           /* the part's thread, start to rows                                       */
    double span_loop_s;      /* first span handed to the device -> last flush                          */
    double wait_loader_s;    /* of which waiting for the loader / uploader                             */
    double device_calls_s;   /* ... and inside inq_call_span_deferred / inq_call_flush                 */
    int32_t front;           /* 1 = host sweep, 2 = device spans                                       */
    int32_t io_threads;      /* reader threads of its span pipeline                                    */
} inq_part_stats_t;
int inq_genotype_repeats_devices(const inq_call_args_t *args, const int32_t *device_ids, size_t n_devices, int out_fd, inq_part_stats_t *stats,
                                 char *errbuf, size_t errcap);
/* How many processes (or device parts) share this host's cores with this one, and which of them it is (0 .. sharers - 1): sizes the
 * reader pool of every later call (granted cores / sharers, bound to L3 domains from a different start per sharer).  Default:
 * LOCAL_WORLD_SIZE / LOCAL_RANK of the environment (torch.distributed.run exports them), else 1 / 0.  sharers <= 0 = that default. */
void inq_host_set_local_share(int sharers, int index);
/* what the last call that ended in this process did (spans, BAM bytes, span loop / loader / device seconds, front end, reader
 * threads; device, status, loci and rows_s are not filled): a rank of the one-process-per-GPU run reports it next to its rows */
void inq_host_last_call_stats(inq_part_stats_t *out);
/* An option (include/inquistr_hip.h: inq_ctx_set_option's keys) for every device context this library makes from now on: the host
 * library creates its contexts itself, so this is how a caller of inq_genotype_repeats & co sets "inflate_ahead", "verify_crc",
 * "grid_tail", ... (`inquistr call --ctx-option key=value`).  0, or 1 for an unknown key / a value out of range. */
int inq_host_ctx_option(const char *key, int64_t value);
int inq_host_granted_cpus(void); /* CPUs of the affinity mask, cut by the cgroup's CPU quota */
int inq_host_span_io_threads(uint64_t threads, int sharers); /* the reader pool size a call with -t threads would take */
/* tests (no GPU involved): the control flow of inq_genotype_repeats_devices with rows that name their target and their part; a device
 * context that fails in a chosen way (host/multi_device.cc) */
int inq_host_devices_selftest(const inq_call_args_t *args, size_t n_parts, int fail_part, int out_fd, uint64_t *cuts, char *errbuf, size_t errcap);
void inq_host_test_ctx_creator(int mode, long timeout_ms);

/* ---- a prepared run: what get_targets + get_bam_reader leave behind (src/call.rs:146-147,182-202), kept open ----
 * The BAM header, its index and the target list are read ONCE; the handle then serves the work split, this process's rows and
 * the output stage of a multi-process run (inquistr_amd/call_dist.py), so that rank 0 neither re-opens the BAM nor re-parses
 * the BED for the names, and the ordered .inq text (src/call.rs:137-157) is written by the same code as in inq_genotype_repeats. */
typedef struct inq_run inq_run_t;
int inq_run_open(const inq_call_args_t *args, inq_run_t **out, char *errbuf, size_t errcap); /* statuses as inq_genotype_repeats */
uint64_t inq_run_n_targets(const inq_run_t *run);
const char *inq_run_sample(const inq_run_t *run);
int inq_run_target(const inq_run_t *run, uint64_t i, const char **chrom, uint32_t *start, uint32_t *end);
/* as inq_host_partition; order[] holds inq_run_n_targets() entries, cuts[] world + 1.  No GPU involved. */
int inq_run_partition(inq_run_t *run, uint64_t world, uint32_t *order, uint64_t *cuts, char *errbuf, size_t errcap);
/* as inq_genotype_repeats_rows (runs on the GPU) */
int inq_run_rows(inq_run_t *run, const uint32_t *target_index, uint64_t n_index, double *phase1, double *phase2, char *errbuf, size_t errcap);
/* as inq_run_rows, with the rows left in DEVICE memory: *d_phase1 / *d_phase2 receive arrays of `width` >= n_index f64 on args->device
 * (row k = target target_index[k]; entries from n_index on are NaN; the two lie back to back: *d_phase2 = *d_phase1 + width, one
 * [2][width] buffer), owned by the run and valid until the next call on it or
 * inq_run_close.  For one process per GPU, whose RCCL gather of the per-shard rows (north_star) then reads them where the kernels'
 * rows already are - `width` = the largest part of the work split, so that every rank's buffer has the collective's one shape.  The
 * run keeps its device context from the first such call on. */
int inq_run_rows_device(inq_run_t *run, const uint32_t *target_index, uint64_t n_index, uint64_t width, void **d_phase1, void **d_phase2,
                        char *errbuf, size_t errcap);
/* header + one row per target (phase1[i], phase2[i] = row of target i of the list; n_rows must equal inq_run_n_targets) to
 * out_fd: BED order for -t 1, (human_compare(chrom), start) order for -t >= 2 (src/call.rs:33-38,141).  No GPU involved. */
int inq_run_write_inq(inq_run_t *run, const double *phase1, const double *phase2, uint64_t n_rows, int out_fd, char *errbuf, size_t errcap);
void inq_run_close(inq_run_t *run);

/* ---- a session: many BAMs on one device context ----
 * A cohort is called sample by sample with the same targets and pasted together by `combine` (src/combine.rs).  One process per
 * sample pays the HIP runtime's start-up (0.1 - 0.3 s: all of a 1 GB file's time) every time; a session pays it once.
 * inq_session_call is inq_genotype_repeats on the session's context.  inq_session_call_many runs args[0 .. n) in order, file
 * k + 1 being opened, planned, read and uploaded while file k is called; out_fds[k] receives file k's .inq - byte for byte what
 * its own `inquistr call` prints -, statuses[k] (may be NULL) its exit status; a failing file does not stop the others.
 * Returns 0, or the status of the first failing file (its message in errbuf).  The per-BAM surface of the reference is
 * unchanged: this is the loop a caller would otherwise write around `inquiSTR call`. */
typedef struct inq_session inq_session_t;
int inq_session_open(int32_t device, inq_session_t **out); /* returns at once; the runtime starts on a thread of its own */
int inq_session_call(inq_session_t *s, const inq_call_args_t *args, int out_fd, char *errbuf, size_t errcap);
int inq_session_call_many(inq_session_t *s, const inq_call_args_t *args, size_t n, const int *out_fds, int *statuses, char *errbuf, size_t errcap);
void inq_session_close(inq_session_t *s);
/* The two halves of inq_session_call, for a caller that receives its files one by one (`inquistr serve`): inq_session_stage opens the
 * BAM, validates the targets, plans the spans and starts reading and uploading them - it may run on another thread while the file
 * before is inside inq_session_run -; inq_session_run does the calling and writes the .inq (and reports what staging found wrong),
 * and frees the handle.  At most ONE file may be staged ahead of the one that is running; files run in the order they were staged.
 * inq_session_discard drops a staged file that will not be run. */
typedef struct inq_staged inq_staged_t;
int inq_session_stage(inq_session_t *s, const inq_call_args_t *args, inq_staged_t **out);
int inq_session_run(inq_session_t *s, inq_staged_t *staged, int out_fd, char *errbuf, size_t errcap);
void inq_session_discard(inq_staged_t *staged);
/* A prepared run ON a session (inq_run_* above): its inq_run_rows / inq_run_rows_device calls use the session's device context, its
 * span buffers and its BED cache instead of making their own - what a resident rank of a one-process-per-GPU job
 * (inquistr_amd/call_dist.py under torch.distributed.run) calls file after file, or pass after pass: the context (streams, device
 * slots: tens of milliseconds even in a process whose runtime is up) and a GB of touched span buffers are made once per process, not
 * once per file.  args->device is ignored (the session's device is used).  The session must outlive the run; one call at a time per
 * session, as for every other inq_session_* entry. */
int inq_session_run_open(inq_session_t *s, const inq_call_args_t *args, inq_run_t **out, char *errbuf, size_t errcap);

/* ---- BAM -> batch front end (no GPU involved) ---- */
typedef struct inq_frontend inq_frontend_t;
int inq_frontend_open(const inq_call_args_t *args, inq_frontend_t **out, char *errbuf, size_t errcap);
uint64_t inq_frontend_n_targets(const inq_frontend_t *fe);
int inq_frontend_target(const inq_frontend_t *fe, uint64_t i, const char **chrom, uint32_t *start, uint32_t *end);
const char *inq_frontend_sample(const inq_frontend_t *fe);
/* Next batch: 1 = *batch filled (host pointers, valid until the next call), 0 = no more, <0 = exit
 * status negated.  locus_index[j] = position of batch locus j in the target list. */
int inq_frontend_next(inq_frontend_t *fe, inq_batch_t *batch, const uint32_t **locus_index, char *errbuf,
                      size_t errcap);
void inq_frontend_set_batch_words(inq_frontend_t *fe, uint64_t max_cigar_words);
void inq_frontend_close(inq_frontend_t *fe);

/* ---- spans: host half of the DEVICE front end (inq_call_span in inquistr_hip.h), no GPU involved ----
 * Cuts the targets into spans (loci of one contig + the whole BGZF blocks holding every record that
 * overlaps them, found through the .bai), reads the compressed bytes and builds the block table and the
 * record-start anchors.  max_comp_bytes = 0 takes the default (2 GiB, env INQ_SPAN_MB). */
typedef struct inq_spans inq_spans_t;
int inq_spans_open(const inq_call_args_t *args, uint64_t max_comp_bytes, inq_spans_t **out, char *errbuf, size_t errcap);
uint64_t inq_spans_n_targets(const inq_spans_t *s);
/* Next span: 1 = *span filled (host pointers, valid until the next call), 0 = no more, <0 = exit status
 * negated.  locus_index[j] = position of span locus j in the target list; targets that appear in no span
 * have no record anywhere near (rows NaN NaN).  file_begin = file offset of span->comp[0]. */
int inq_spans_next(inq_spans_t *s, inq_span_t *span, const uint32_t **locus_index, uint64_t *file_begin, char *errbuf,
                   size_t errcap);
void inq_spans_close(inq_spans_t *s);

/* `inquiSTR combine` (src/combine.rs:27-59): column-wise paste of N .inq files — every line of the first
 * file, then columns 4.. of the same line of each other file, tab-joined.  Files ending in ".gz" are
 * read through gzip (src/combine.rs:10-25).  Returns 0, or 101 where the reference panics (missing
 * file, a later file with fewer lines than the first). */
int inq_combine(const char *const *files, size_t n_files, int out_fd, char *errbuf, size_t errcap);

/* `inquiSTR outlier` (src/outlier.rs:33-73; arguments src/main.rs:75-99): loci of a combined .inq (plain or gzip)
 * with outlying samples, one line `chrom begin end s1,s2,...` each, after the header `chrom begin end outliers`.
 * The arithmetic (z-score / DBSCAN per locus) runs on the GPU (inq_outlier_rows).  Returns 0, 1 (no GPU) or 101
 * where the reference panics (missing file, -s with -S, a number that does not parse, ...). */
typedef struct inq_outlier_args {
    const char *combined;    /* positional                                   */
    uint32_t minsize;        /* --minsize, default 10                        */
    float zscore;            /* -z / --zscore, default 3.0                   */
    int32_t method;          /* --method: INQ_OUTLIER_ZSCORE (default) / INQ_OUTLIER_DBSCAN */
    const char *sample;      /* -s / --sample, NULL if absent                */
    const char *subset_file; /* -S / --subset, NULL if absent                */
    int32_t device;          /* HIP device ordinal (not a reference argument) */
    int32_t reserved;
} inq_outlier_args_t;
int inq_outlier(const inq_outlier_args_t *args, int out_fd, char *errbuf, size_t errcap);

/* ---- text side (src/call.rs:27-65, 91-101) ---- */
size_t inq_host_format_f64(double v, char *buf, size_t cap);
size_t inq_host_format_row(const char *chrom, uint32_t start, uint32_t end, double p1, double p2, char *buf, size_t cap);
size_t inq_host_format_header(const char *sample, char *buf, size_t cap);
size_t inq_host_sample_name(const char *bam_path, char *buf, size_t cap);
int inq_host_human_compare(const char *a, const char *b);
/* RepeatIntervalIterator::from_string against a one-contig length table; 0 ok, 101 panic */
int inq_host_parse_region(const char *reg, const char *chrom_name, uint64_t chrom_len, char *chrom_out, size_t cap,
                          uint32_t *start, uint32_t *end);
/* BAI facts (for fixtures): number of references, and mapped/unmapped counts of one */
/* File offset (compressed bytes) from which a forward scan sees every record of `tid` overlapping positions
 * >= pos, from the .bai linear index; 0 if the contig has nothing there.  Lets a multi-process run split the
 * targets by the amount of BAM each rank has to read rather than by locus count. */
uint64_t inq_host_bai_file_offset(const char *bai_path, int32_t tid, int64_t pos);
/* The same as a full virtual offset (compressed offset << 16 | offset in the inflated block); bai_path may name a .csi.
 * This is where both front ends start reading for a window that begins at pos (the reference: bam.fetch(), src/call.rs:288,338). */
uint64_t inq_host_bai_scan_start(const char *bai_path, int32_t tid, int64_t pos);
/* The device front end's plan on its own (host/span_planner.cc): for the targets of args, every segment [vo_begin, vo_limit) of
 * every span (seg_span[k] = number of the span segment k belongs to; *n_segs = how many there are, also when seg_cap is smaller)
 * and, per target, the span it is called in (0xffffffff: no record can overlap it).  Reads the BAM's header and index only. */
int inq_host_plan_spans(const inq_call_args_t *args, uint64_t max_comp_bytes, uint64_t *seg_vo_begin, uint64_t *seg_vo_limit, uint32_t *seg_span,
                        uint64_t seg_cap, uint64_t *n_segs, uint32_t *target_span, uint64_t target_cap, char *errbuf, size_t errcap);
int inq_host_bam_tid(const char *bam_path, const char *contig);
int inq_host_bai_stats(const char *bai_path, uint32_t *n_ref, int32_t tid, uint64_t *n_mapped, uint64_t *n_unmapped,
                       uint64_t *n_bins, uint64_t *n_intv);
/* Compressed BAM bytes this process has handed to the device front end so far (all calls, all files): what a rank of the
 * one-process-per-GPU run read for its share of the targets (inq_run_partition promises every rank about the same). */
uint64_t inq_host_span_bytes_read(void);
/* Self-check of the reader-thread pool the span loader copies file bytes with (host/span_planner.h IoPool): `rounds` runs of `n_jobs`
 * jobs on `n_threads` threads (bound to the L3 domains of NUMA node `numa_node`, -1 = the machine, when pin != 0); every job adds
 * its index + 1 to a sum.  Returns the sum over all rounds (= rounds * n_jobs * (n_jobs + 1) / 2 when every job ran exactly once). */
uint64_t inq_host_iopool_selftest(int n_threads, int numa_node, int pin, uint64_t n_jobs, uint64_t rounds);

#ifdef __cplusplus
}
#endif
#endif
