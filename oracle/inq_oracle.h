/*
 * inq_oracle.h — CPU restatement of inquiSTR's `call` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (inquistr_amd/, include/) never links, imports or executes it.
 *
 * PARITY UNPINNED: the reference is Rust (no rustc/cargo in the image), its only BAM
 * fixture is absent and none of its tests asserts a number on this path (SURVEY.md §4,
 * §8c).  This restatement is pinned only by the hand-derived known-answer vectors in
 * tests/golden/ and by a second, independent Python restatement (oracle/pyoracle.py).
 *
 * Each function cites the reference lines (wdecoster/inquiSTR v0.13.0) it follows.
 */
#ifndef INQ_ORACLE_H
#define INQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/inquistr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* "panic" outcomes of the reference, reported instead of aborting */
enum {
    ORC_OK = 0,
    ORC_PANIC_CIGAR_OP = 1,   /* rust-htslib cigar(): unknown op code                       */
    ORC_PANIC_SA_TYPE = 2,    /* src/call.rs:428-431  SA aux not a string                   */
    ORC_PANIC_SA_FORMAT = 3,  /* src/call.rs:433-450  index / unwrap / parse failures       */
    ORC_PANIC_HP_TYPE = 4,    /* src/call.rs:487      HP aux neither U8 nor I32             */
    ORC_PANIC_PHASE_KEY = 5,  /* src/call.rs:358      phase not in {0,1,2}                  */
    ORC_PANIC_SUPPORT = 6,    /* src/call.rs:516      support == 0 with empty input         */
    ORC_PANIC_LOCUS = 7,      /* src/repeats.rs:102-114, src/call.rs:285 (start < 10)       */
    ORC_PANIC_PARSE = 8       /* region / BED field does not parse                          */
};

/* Call::Span(v) / Call::Clip(v), src/call.rs:67-71 */
typedef struct orc_call {
    int64_t value;
    int clipped;
} orc_call_t;

/* The slice of a BAM record the path touches */
typedef struct orc_record {
    int32_t tid;
    int64_t pos;     /* core.pos, 0-based */
    uint16_t flag;   /* 0x4 unmapped, 0x10 reverse */
    uint8_t mapq;
    uint32_t n_cigar;
    const uint32_t *cigar; /* len<<4|op */
    char hp_type;          /* 0 = HP absent, else BAM aux type char: c C s S i I ... */
    int64_t hp_value;
    char sa_type;   /* 0 = SA absent, 'Z' = string, else other aux type */
    const char *sa; /* NUL-terminated when sa_type == 'Z' */
    int is2d_given; /* -1: evaluate is_accidental_2d from SA; 0/1: use this value */
} orc_record_t;

/* htslib bam_endpos (reached via Record::reference_end, src/call.rs:298,351,449) */
int64_t orc_bam_endpos(const orc_record_t *r);
/* src/call.rs:461-477 */
int64_t orc_cigar_to_rlen(const char *cigar, int *panic);
/* src/call.rs:415-459 */
int orc_is_accidental_2d(const orc_record_t *r, int *panic);
/* src/call.rs:482-491; returns 1 if Some(phase) */
int orc_get_phase(const orc_record_t *r, uint8_t *phase, int *panic);
/* src/call.rs:377-413 */
orc_call_t orc_call_from_cigar(const orc_record_t *r, uint32_t minlen, uint32_t start, uint32_t end,
                               int *panic);
/* src/call.rs:497-522 */
double orc_median_str_length(const orc_call_t *calls, size_t n, size_t support, int *panic);

/* src/call.rs:329-374 / 279-327 on an in-memory record list in file order.  The list may
 * hold records fetch() would not yield; htslib's iterator rule (tid match, pos < end_ext,
 * endpos > start_ext) is applied here.  *tie (may be NULL) is set when the unphased split
 * cuts through equal values of mixed Span/Clip. */
int orc_genotype_repeat_phased(const orc_record_t *recs, size_t n, int32_t tid, uint32_t start,
                               uint32_t end, uint32_t minlen, size_t support, double *phase1,
                               double *phase2);
int orc_genotype_repeat_unphased(const orc_record_t *recs, size_t n, int32_t tid, uint32_t start,
                                 uint32_t end, uint32_t minlen, size_t support, double *phase1,
                                 double *phase2, int *tie);

/* Batch form over the C-ABI structs (same contract as inq_call_batch, host pointers).
 * Returns an INQ_* code.  n_threads > 1 spreads loci over OpenMP threads (the reference's
 * rayon par_bridge over loci, src/call.rs:115-118). */
int orc_call_batch(const inq_batch_t *batch, inq_result_t *result, int n_threads);

/* ---- text side of the path ---- */
/* Rust `{}` of f64 for the values this path can produce (integers, halves, NaN) */
size_t orc_format_f64(double v, char *buf, size_t cap);
/* Genotype Display, src/call.rs:57-65 */
size_t orc_format_row(const char *chrom, uint32_t start, uint32_t end, double p1, double p2, char *buf,
                      size_t cap);
/* header, src/call.rs:101 */
size_t orc_format_header(const char *sample, char *buf, size_t cap);
/* sample name from the BAM path, src/call.rs:91-100 */
size_t orc_sample_name(const char *bam_path, char *buf, size_t cap);
/* human_sort::compare as used by Genotype::cmp, src/call.rs:33-38 */
int orc_human_compare(const char *a, const char *b);
/* RepeatIntervalIterator::from_string, src/repeats.rs:13-29 (chrom copied into chrom_buf) */
int orc_parse_region(const char *reg, char *chrom_buf, size_t cap, uint32_t *start, uint32_t *end);
/* RepeatInterval::new_interval, src/repeats.rs:96-115: chrom_len < 0 means "not in header" */
int orc_check_interval(uint32_t start, uint32_t end, int64_t chrom_len);

#ifdef __cplusplus
}
#endif
#endif
