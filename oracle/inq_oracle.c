/*
 * inq_oracle.c — CPU restatement of inquiSTR's `call` hot path.  TEST INFRASTRUCTURE ONLY
 * (see inq_oracle.h: parity unpinned; never linked into the product).
 *
 * Written from the reference's text, function by function; every block names the lines of
 * wdecoster/inquiSTR v0.13.0 it restates.  Facts that live in un-vendored third parties
 * (htslib bam_endpos and iterator overlap rule, rust-htslib aux typing, Rust integer parsing
 * and f64 Display, human-sort) are restated from their published behaviour and marked [3P].
 */
#include "inq_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* BAM op codes: M I D N S H P = X */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

/* [3P] htslib bam_cigar_type(op) & 2: the op consumes the reference */
static int op_consumes_ref(uint32_t op) {
    return op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X;
}

/* [3P] htslib sam.c bam_endpos: rlen = unmapped ? 0 : bam_cigar2rlen(); if (rlen == 0) rlen = 1 */
int64_t orc_bam_endpos(const orc_record_t *r) {
    int64_t rlen = 0;
    if (!(r->flag & 0x4)) {
        for (uint32_t i = 0; i < r->n_cigar; i++) {
            uint32_t op = r->cigar[i] & 0xf, len = r->cigar[i] >> 4;
            if (op_consumes_ref(op)) rlen += len;
        }
    }
    if (rlen == 0) rlen = 1;
    return r->pos + rlen;
}

/* [3P] Rust `str::parse::<i64>()`: optional sign, at least one ASCII digit, nothing else */
static int parse_i64(const char *s, size_t n, int64_t *out) {
    size_t i = 0;
    int neg = 0;
    if (n == 0) return 0;
    if (s[0] == '+' || s[0] == '-') {
        neg = s[0] == '-';
        i = 1;
    }
    if (i == n) return 0;
    __int128 v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 0;
        v = v * 10 + (s[i] - '0');
        if (v > ((__int128)1 << 63)) return 0;
    }
    if (neg) v = -v;
    if (v > INT64_MAX || v < INT64_MIN) return 0;
    *out = (int64_t)v;
    return 1;
}

/* [3P] Rust `str::parse::<u32>()`: optional '+', digits, no overflow */
static int parse_u32(const char *s, size_t n, uint32_t *out) {
    size_t i = 0;
    if (n == 0) return 0;
    if (s[0] == '+') i = 1;
    if (i == n) return 0;
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 0;
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > UINT32_MAX) return 0;
    }
    *out = (uint32_t)v;
    return 1;
}

/* src/call.rs:461-477: digits accumulate, any other char closes a number; M = X D N add */
int64_t orc_cigar_to_rlen(const char *cigar, int *panic) {
    int64_t rlen = 0;
    const char *num = cigar;
    size_t numlen = 0;
    for (const char *c = cigar; *c; c++) {
        if (*c >= '0' && *c <= '9') {
            if (numlen == 0) num = c;
            numlen++;
        } else {
            int64_t n;
            if (!parse_i64(num, numlen, &n)) { /* :469 unwrap on an empty / overflowing number */
                if (panic) *panic = ORC_PANIC_SA_FORMAT;
                return 0;
            }
            if (*c == 'M' || *c == '=' || *c == 'X' || *c == 'D' || *c == 'N') rlen += n;
            numlen = 0;
        }
    }
    return rlen;
}

/* src/call.rs:415-459 */
int orc_is_accidental_2d(const orc_record_t *r, int *panic) {
    if (r->is2d_given >= 0) return r->is2d_given;
    char read_strand = (r->flag & 0x10) ? '-' : '+'; /* :422 */
    if (r->sa_type == 0) return 0;                   /* :425-427 */
    if (r->sa_type != 'Z') {                         /* :429-432 */
        if (panic) *panic = ORC_PANIC_SA_TYPE;
        return 0;
    }
    /* :434 split on ';', drop empty entries */
    const char *s = r->sa;
    const char *first = NULL;
    size_t first_len = 0;
    int n_entries = 0;
    while (1) {
        const char *e = strchr(s, ';');
        size_t len = e ? (size_t)(e - s) : strlen(s);
        if (len > 0) {
            if (n_entries == 0) {
                first = s;
                first_len = len;
            }
            n_entries++;
        }
        if (!e) break;
        s = e + 1;
    }
    if (n_entries > 1) return 0; /* :436-438 */
    if (n_entries == 0) {        /* :439 sa_entries[0] out of bounds */
        if (panic) *panic = ORC_PANIC_SA_FORMAT;
        return 0;
    }
    /* :439 split the entry on ',' : rname,POS,strand,CIGAR,mapQ,NM */
    const char *fld[8];
    size_t flen[8];
    int nf = 0;
    {
        const char *p = first, *end = first + first_len;
        const char *start = p;
        for (;; p++) {
            if (p == end || *p == ',') {
                if (nf < 8) {
                    fld[nf] = start;
                    flen[nf] = (size_t)(p - start);
                }
                nf++;
                start = p + 1;
                if (p == end) break;
            }
        }
    }
    if (nf < 3 || flen[2] == 0) { /* :441 sa_entry[2].chars().next().unwrap() */
        if (panic) *panic = ORC_PANIC_SA_FORMAT;
        return 0;
    }
    if (read_strand == fld[2][0]) return 0; /* :441-443 (first byte; strand is ASCII) */
    int64_t start = r->pos;                 /* :448 */
    int64_t end = orc_bam_endpos(r);        /* :449 */
    int64_t sa_start;
    if (!parse_i64(fld[1], flen[1], &sa_start)) { /* :450 */
        if (panic) *panic = ORC_PANIC_SA_FORMAT;
        return 0;
    }
    if (nf < 4) { /* :451 sa_entry[3] */
        if (panic) *panic = ORC_PANIC_SA_FORMAT;
        return 0;
    }
    char cig[4096];
    char *cigp = cig;
    if (flen[3] + 1 > sizeof cig) cigp = (char *)malloc(flen[3] + 1);
    memcpy(cigp, fld[3], flen[3]);
    cigp[flen[3]] = 0;
    int p2 = 0;
    int64_t sa_end = sa_start + orc_cigar_to_rlen(cigp, &p2); /* :451 */
    if (cigp != cig) free(cigp);
    if (p2) {
        if (panic) *panic = p2;
        return 0;
    }
    int64_t lo = start > sa_start ? start : sa_start;
    int64_t hi = end < sa_end ? end : sa_end;
    return lo < hi; /* :454-458 */
}

/* src/call.rs:482-491; [3P] rust-htslib maps aux type C->U8, i->I32, c->I8, s->I16, S->U16, I->U32 */
int orc_get_phase(const orc_record_t *r, uint8_t *phase, int *panic) {
    if (r->hp_type == 0) return 0; /* Err(_) => None */
    if (r->hp_type == 'C') {       /* Aux::U8(v) => Some(v) */
        *phase = (uint8_t)r->hp_value;
        return 1;
    }
    if (r->hp_type == 'i') { /* Aux::I32(v) => Some(v as u8): truncation */
        *phase = (uint8_t)(uint32_t)(int32_t)r->hp_value;
        return 1;
    }
    if (panic) *panic = ORC_PANIC_HP_TYPE; /* :487 */
    return 0;
}

/* src/call.rs:377-413 */
orc_call_t orc_call_from_cigar(const orc_record_t *r, uint32_t minlen, uint32_t start, uint32_t end,
                               int *panic) {
    orc_call_t out = {0, 0};
    /* :382 r.cigar() decodes every op first; [3P] an op code above 8 panics there */
    for (uint32_t i = 0; i < r->n_cigar; i++) {
        if ((r->cigar[i] & 0xf) > 8) {
            if (panic) *panic = ORC_PANIC_CIGAR_OP;
            return out;
        }
    }
    int64_t call = 0;                                   /* :378 */
    uint32_t reference_position = (uint32_t)(r->pos + 1); /* :380, `as u32` truncates */
    int clipped = 0;                                    /* :381 */
    for (uint32_t i = 0; i < r->n_cigar; i++) {
        uint32_t op = r->cigar[i] & 0xf, len = r->cigar[i] >> 4;
        switch (op) {
        case OP_M:
        case OP_EQ:
        case OP_X:
            reference_position += len; /* :384-386, release build: wrapping */
            break;
        case OP_D: /* :387-392 */
            if (len > minlen && start < reference_position && reference_position < end) call -= (int64_t)len;
            reference_position += len;
            break;
        case OP_S: { /* :393-398; `!is_accidental_2d(&r)` is the FIRST operand of the && chain */
            int p2 = 0;
            int is2d = orc_is_accidental_2d(r, &p2);
            if (p2) {
                if (panic) *panic = p2;
                return out;
            }
            if (!is2d && len > minlen && start < reference_position && reference_position < end) {
                call += (int64_t)len;
                clipped = 1;
            }
            break;
        }
        case OP_I: /* :399-403 */
            if (len > minlen && start < reference_position && reference_position < end) call += (int64_t)len;
            break;
        case OP_N:
            reference_position += len; /* :404 */
            break;
        default: /* :405 HardClip, Pad */
            break;
        }
    }
    out.value = call;
    out.clipped = clipped; /* :408-412 */
    return out;
}

static int cmp_i64_asc(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}
static int cmp_i64_desc(const void *a, const void *b) { return cmp_i64_asc(b, a); }

/* src/call.rs:497-522 */
double orc_median_str_length(const orc_call_t *calls, size_t n, size_t support, int *panic) {
    if (n < support) return NAN; /* :498-500 */
    int64_t *spanning = (int64_t *)malloc((n + support + 1) * sizeof(int64_t));
    int64_t *clipped = (int64_t *)malloc((n + 1) * sizeof(int64_t));
    size_t ns = 0, nc = 0;
    for (size_t i = 0; i < n; i++) { /* :503-508 */
        if (calls[i].clipped)
            clipped[nc++] = calls[i].value;
        else
            spanning[ns++] = calls[i].value;
    }
    if (ns <= support) { /* :509-513 */
        qsort(clipped, nc, sizeof(int64_t), cmp_i64_desc);
        size_t take = support - ns; /* n >= support guarantees take <= nc */
        for (size_t i = 0; i < take; i++) spanning[ns + i] = clipped[i];
        ns += take;
    }
    double res;
    if (ns == 0) { /* only reachable with support == 0: `0/2 - 1` underflows at :516 */
        if (panic) *panic = ORC_PANIC_SUPPORT;
        res = NAN;
    } else {
        qsort(spanning, ns, sizeof(int64_t), cmp_i64_asc); /* :514 */
        if (ns % 2 == 0)
            res = (double)(spanning[ns / 2 - 1] + spanning[ns / 2]) / 2.0; /* :515-518 */
        else
            res = (double)spanning[ns / 2]; /* :520 */
    }
    free(spanning);
    free(clipped);
    return res;
}

/* [3P] htslib iterator: yield iff tid matches, pos < end, bam_endpos > beg */
static int fetch_yields(const orc_record_t *r, int32_t tid, uint32_t beg, uint32_t end) {
    return r->tid == tid && r->pos < (int64_t)end && orc_bam_endpos(r) > (int64_t)beg;
}

/* src/call.rs:329-374 */
int orc_genotype_repeat_phased(const orc_record_t *recs, size_t n, int32_t tid, uint32_t start,
                               uint32_t end, uint32_t minlen, size_t support, double *phase1,
                               double *phase2) {
    if (start < 10) return ORC_PANIC_LOCUS; /* :335 u32 underflow ⇒ fetch fails */
    uint32_t start_ext = start - 10, end_ext = end + 10; /* :335-336 */
    orc_call_t *calls[3];
    size_t cnt[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) calls[k] = (orc_call_t *)malloc((n + 1) * sizeof(orc_call_t));
    int panic = 0;
    for (size_t i = 0; i < n && !panic; i++) {
        const orc_record_t *r = &recs[i];
        if (!fetch_yields(r, tid, start_ext, end_ext)) continue; /* :338,345 */
        uint8_t phase = 0;
        int has = orc_get_phase(r, &phase, &panic); /* :349 */
        if (panic) break;
        uint32_t rs = (uint32_t)r->pos, re = (uint32_t)orc_bam_endpos(r);
        if (!has || (start_ext < rs && re < end_ext) || r->mapq <= 10) continue; /* :350-355 */
        orc_call_t c = orc_call_from_cigar(r, minlen, start_ext, end_ext, &panic); /* :357 */
        if (panic) break;
        if (phase > 2) { /* :358 get_mut(&phase).unwrap() */
            panic = ORC_PANIC_PHASE_KEY;
            break;
        }
        calls[phase][cnt[phase]++] = c;
    }
    if (!panic) {
        *phase1 = orc_median_str_length(calls[1], cnt[1], support, &panic); /* :367 */
        *phase2 = orc_median_str_length(calls[2], cnt[2], support, &panic); /* :368 */
    }
    for (int k = 0; k < 3; k++) free(calls[k]);
    return panic;
}

/* src/call.rs:279-327 */
int orc_genotype_repeat_unphased(const orc_record_t *recs, size_t n, int32_t tid, uint32_t start,
                                 uint32_t end, uint32_t minlen, size_t support, double *phase1,
                                 double *phase2, int *tie) {
    if (start < 10) return ORC_PANIC_LOCUS;
    uint32_t start_ext = start - 10, end_ext = end + 10; /* :285-286 */
    orc_call_t *calls = (orc_call_t *)malloc((n + 1) * sizeof(orc_call_t));
    size_t m = 0;
    int panic = 0;
    for (size_t i = 0; i < n; i++) {
        const orc_record_t *r = &recs[i];
        if (!fetch_yields(r, tid, start_ext, end_ext)) continue; /* :288,294 */
        uint32_t rs = (uint32_t)r->pos, re = (uint32_t)orc_bam_endpos(r);
        if (start_ext < rs || re < end_ext || r->mapq <= 10) continue; /* :297-302 */
        calls[m++] = orc_call_from_cigar(r, minlen, start_ext, end_ext, &panic); /* :303-304 */
        if (panic) break;
    }
    if (!panic) {
        /* :311 sort_unstable_by_key(value).  Build rule for equal keys: file order (insertion
         * sort = what std uses for <= 20 elements; a defined choice above that). */
        for (size_t i = 1; i < m; i++) {
            orc_call_t x = calls[i];
            size_t j = i;
            while (j > 0 && calls[j - 1].value > x.value) {
                calls[j] = calls[j - 1];
                j--;
            }
            calls[j] = x;
        }
        size_t k = m / 2; /* :313 split_at(len/2) */
        if (tie) {
            *tie = 0;
            if (k >= 1 && k < m && calls[k - 1].value == calls[k].value) {
                int has_span = 0, has_clip = 0;
                for (size_t i = 0; i < m; i++)
                    if (calls[i].value == calls[k].value) {
                        if (calls[i].clipped)
                            has_clip = 1;
                        else
                            has_span = 1;
                    }
                *tie = has_span && has_clip;
            }
        }
        *phase1 = orc_median_str_length(calls, k, support, &panic);         /* :319 */
        *phase2 = orc_median_str_length(calls + k, m - k, support, &panic); /* :320 */
    }
    free(calls);
    return panic;
}

/* ---- batch form over the C-ABI structs ---- */

static int panic_to_inq(int panic) {
    switch (panic) {
    case ORC_OK: return INQ_OK;
    case ORC_PANIC_CIGAR_OP: return INQ_ERR_CIGAR_OP;
    case ORC_PANIC_PHASE_KEY: return INQ_ERR_PHASE;
    case ORC_PANIC_SUPPORT: return INQ_ERR_SUPPORT_ZERO;
    case ORC_PANIC_LOCUS: return INQ_ERR_LOCUS;
    case ORC_PANIC_SA_TYPE:
    case ORC_PANIC_SA_FORMAT:
    case ORC_PANIC_HP_TYPE: return INQ_ERR_AUX;
    default: return INQ_ERR_ARG;
    }
}

/* error precedence shared with the HIP library: INDEX > CIGAR_OP > RANGE > PHASE > AUX */
static int err_rank(int code) {
    switch (code) {
    case INQ_ERR_INDEX: return 5;
    case INQ_ERR_CIGAR_OP: return 4;
    case INQ_ERR_RANGE: return 3;
    case INQ_ERR_PHASE: return 2;
    case INQ_ERR_AUX: return 1;
    default: return 0;
    }
}

int orc_call_batch(const inq_batch_t *b, inq_result_t *res, int n_threads) {
    if (!b || !res || !res->phase1 || !res->phase2) return INQ_ERR_ARG;
    if (b->n_loci && (!b->locus_pair_off || !b->locus_start || !b->locus_end)) return INQ_ERR_ARG;
    if (b->n_pairs && (!b->pair_read || !b->reads)) return INQ_ERR_ARG;
    if (b->n_cigar_words && !b->cigar) return INQ_ERR_ARG;
    if (b->n_cigar_words % 4 != 0 || b->reserved != 0 || b->unphased > 1) return INQ_ERR_ARG;
    if (b->support == 0) return INQ_ERR_SUPPORT_ZERO;
    if (b->n_loci) {
        if (b->locus_pair_off[0] != 0 || b->locus_pair_off[b->n_loci] != b->n_pairs) return INQ_ERR_ARG;
    } else if (b->n_pairs != 0)
        return INQ_ERR_ARG;
    for (uint64_t j = 0; j < b->n_loci; j++) {
        if (b->locus_pair_off[j] > b->locus_pair_off[j + 1]) return INQ_ERR_ARG;
        if (b->locus_start[j] < 10 || b->locus_end[j] < b->locus_start[j]) return INQ_ERR_LOCUS;
    }
    int worst = INQ_OK;
    uint64_t ties = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : ties)
#endif
    for (int64_t j = 0; j < (int64_t)b->n_loci; j++) {
        uint64_t p0 = b->locus_pair_off[j], p1 = b->locus_pair_off[j + 1];
        size_t n = (size_t)(p1 - p0);
        orc_record_t *recs = (orc_record_t *)calloc(n + 1, sizeof(orc_record_t));
        int code = INQ_OK;
        for (size_t i = 0; i < n; i++) {
            uint32_t ri = b->pair_read[p0 + i];
            if (ri >= b->n_reads) {
                code = err_rank(INQ_ERR_INDEX) > err_rank(code) ? INQ_ERR_INDEX : code;
                recs[i].tid = -1; /* never yielded */
                continue;
            }
            const inq_read_t *rd = &b->reads[ri];
            uint64_t off = (uint64_t)rd->cigar_off4 * 4;
            if (off + rd->n_cigar > b->n_cigar_words) {
                code = err_rank(INQ_ERR_INDEX) > err_rank(code) ? INQ_ERR_INDEX : code;
                recs[i].tid = -1;
                continue;
            }
            orc_record_t *r = &recs[i];
            r->tid = 0;
            r->pos = rd->pos;
            r->flag = (uint16_t)(((rd->bits & INQ_READ_UNMAPPED) ? 0x4 : 0) |
                                 ((rd->bits & INQ_READ_REVERSE) ? 0x10 : 0));
            r->mapq = rd->mapq;
            r->n_cigar = rd->n_cigar;
            r->cigar = b->cigar + off;
            r->hp_type = (rd->bits & INQ_READ_HAS_HP) ? 'C' : 0;
            r->hp_value = rd->phase;
            r->sa_type = 0;
            r->sa = NULL;
            r->is2d_given = (rd->bits & INQ_READ_IS_2D) ? 1 : 0;
            if (rd->bits & INQ_READ_SA_PANIC) {
                /* "is_accidental_2d would panic on this record": stands in as an SA aux that is not a
                 * string, so the panic fires where the reference's does - at the first soft-clip op of a
                 * read that passed the filter (src/call.rs:303,357 -> :394 -> :429-432) - and nowhere else */
                r->sa_type = 'i';
                r->is2d_given = -1;
            }
            /* domain checks the device also makes for every offered read */
            int64_t rlen = 0;
            int bad_op = 0;
            for (uint32_t k = 0; k < r->n_cigar; k++) {
                uint32_t op = r->cigar[k] & 0xf;
                if (op > 8) bad_op = 1;
                if (op_consumes_ref(op)) rlen += r->cigar[k] >> 4;
            }
            if (bad_op && err_rank(INQ_ERR_CIGAR_OP) > err_rank(code)) code = INQ_ERR_CIGAR_OP;
            if ((r->pos < -1 || r->pos + 1 + rlen >= ((int64_t)1 << 31)) &&
                err_rank(INQ_ERR_RANGE) > err_rank(code))
                code = INQ_ERR_RANGE;
        }
        double p1v = NAN, p2v = NAN;
        int tie = 0, panic;
        if (b->unphased)
            panic = orc_genotype_repeat_unphased(recs, n, 0, b->locus_start[j], b->locus_end[j], b->minlen,
                                                 b->support, &p1v, &p2v, &tie);
        else
            panic = orc_genotype_repeat_phased(recs, n, 0, b->locus_start[j], b->locus_end[j], b->minlen,
                                               b->support, &p1v, &p2v);
        int pc = panic_to_inq(panic);
        if (err_rank(pc) > err_rank(code)) code = pc;
        res->phase1[j] = p1v;
        res->phase2[j] = p2v;
        ties += (uint64_t)tie;
        if (res->pair_call || res->pair_bits) {
            uint32_t start_ext = b->locus_start[j] - 10, end_ext = b->locus_end[j] + 10;
            for (size_t i = 0; i < n; i++) {
                const orc_record_t *r = &recs[i];
                uint8_t bits = 0;
                int64_t val = 0;
                if (r->tid == 0) {
                    int pp = 0;
                    orc_record_t q = *r; /* debug value of a read flagged SA_PANIC: as if is_accidental_2d were false */
                    if (q.is2d_given < 0) q.sa_type = 0, q.is2d_given = 0;
                    orc_call_t c = orc_call_from_cigar(&q, b->minlen, start_ext, end_ext, &pp);
                    val = c.value;
                    if (c.clipped) bits |= INQ_PAIR_CLIP;
                    if (fetch_yields(r, 0, start_ext, end_ext)) {
                        bits |= INQ_PAIR_FETCHED;
                        uint32_t rs = (uint32_t)r->pos, re = (uint32_t)orc_bam_endpos(r);
                        int skip;
                        if (b->unphased)
                            skip = start_ext < rs || re < end_ext || r->mapq <= 10;
                        else
                            skip = r->hp_type == 0 || (start_ext < rs && re < end_ext) || r->mapq <= 10;
                        if (!skip) bits |= INQ_PAIR_KEPT;
                    }
                }
                if (res->pair_call) res->pair_call[p0 + i] = val;
                if (res->pair_bits) res->pair_bits[p0 + i] = bits;
            }
        }
        free(recs);
        if (code != INQ_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            if (err_rank(code) > err_rank(worst)) worst = code;
        }
    }
    res->n_tie_loci = ties;
    return worst;
}

/* ---- text side ---- */

/* [3P] Rust f64 Display: shortest round-trip digits, never exponent form, "NaN" for NaN.
 * The path only produces integers and halves (src/call.rs:515-520). */
size_t orc_format_f64(double v, char *buf, size_t cap) {
    /* [3P] Rust's Display for f64 (core::fmt::float, flt2dec shortest): the shortest decimal digits that read back as v, written
     * positionally - never an exponent: 2^60 prints as 1152921504606847000 -, "NaN", "inf", "-0".  What the path produces
     * (integers and halves below 2^53, src/call.rs:518) are their own shortest digits. */
    if (isnan(v)) return (size_t)snprintf(buf, cap, "NaN");
    if (isinf(v)) return (size_t)snprintf(buf, cap, v < 0 ? "-inf" : "inf");
    double a = fabs(v);
    char e[40];
    int prec = 0;
    for (; prec <= 16; prec++) {
        snprintf(e, sizeof e, "%.*e", prec, a);
        if (strtod(e, NULL) == a) break;
    }
    if (prec > 16) snprintf(e, sizeof e, "%.16e", a);
    char digits[24];
    int nd = 0, exp10 = 0;
    for (const char *q = e; *q; q++) {
        if (*q >= '0' && *q <= '9') digits[nd++] = *q;
        else if (*q == 'e') {
            exp10 = atoi(q + 1);
            break;
        }
    }
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    size_t n = 0;
#define PUT(c)                      \
    do {                            \
        if (n + 1 < cap) buf[n] = (c); \
        n++;                        \
    } while (0)
    if (signbit(v)) PUT('-');
    if (exp10 >= nd - 1) {
        for (int i = 0; i < nd; i++) PUT(digits[i]);
        for (int i = 0; i < exp10 - (nd - 1); i++) PUT('0');
    } else if (exp10 >= 0) {
        for (int i = 0; i <= exp10; i++) PUT(digits[i]);
        PUT('.');
        for (int i = exp10 + 1; i < nd; i++) PUT(digits[i]);
    } else {
        PUT('0');
        PUT('.');
        for (int i = 0; i < -exp10 - 1; i++) PUT('0');
        for (int i = 0; i < nd; i++) PUT(digits[i]);
    }
#undef PUT
    if (cap) buf[n < cap ? n : cap - 1] = 0;
    return n;
}

/* src/call.rs:57-65 */
size_t orc_format_row(const char *chrom, uint32_t start, uint32_t end, double p1, double p2, char *buf,
                      size_t cap) {
    char a[64], b[64];
    orc_format_f64(p1, a, sizeof a);
    orc_format_f64(p2, b, sizeof b);
    return (size_t)snprintf(buf, cap, "%s\t%u\t%u\t%s\t%s", chrom, start, end, a, b);
}

/* src/call.rs:101 */
size_t orc_format_header(const char *sample, char *buf, size_t cap) {
    return (size_t)snprintf(buf, cap, "chromosome\tbegin\tend\t%s_H1\t%s_H2", sample, sample);
}

/* Rust str::replace: all non-overlapping matches, left to right */
static void replace_all(char *s, const char *pat) {
    size_t pl = strlen(pat);
    char *w = s;
    for (char *r = s; *r;) {
        if (strncmp(r, pat, pl) == 0)
            r += pl;
        else
            *w++ = *r++;
    }
    *w = 0;
}

/* src/call.rs:91-100: Path::file_stem [3P: name up to the last '.', unless that '.' is the first
 * byte], then every ".bam" and every ".cram" removed */
size_t orc_sample_name(const char *bam_path, char *buf, size_t cap) {
    size_t plen = strlen(bam_path);
    while (plen > 1 && bam_path[plen - 1] == '/') plen--; /* Path ignores trailing separators */
    size_t s = plen;
    while (s > 0 && bam_path[s - 1] != '/') s--;
    size_t nlen = plen - s;
    if (nlen + 1 > cap) nlen = cap - 1;
    memcpy(buf, bam_path + s, nlen);
    buf[nlen] = 0;
    if (strcmp(buf, "..") != 0) {
        char *dot = strrchr(buf, '.');
        if (dot && dot != buf) *dot = 0;
    }
    replace_all(buf, ".bam");
    replace_all(buf, ".cram");
    return strlen(buf);
}

/* [3P] human-sort 0.2.2 compare: walk both strings; when both cursors sit on digits compare the
 * two digit runs by numeric value, otherwise compare one char by code point; the shorter string
 * is smaller when one runs out. */
int orc_human_compare(const char *a, const char *b) {
    while (*a && *b) {
        int da = *a >= '0' && *a <= '9', db = *b >= '0' && *b <= '9';
        if (da && db) {
            unsigned __int128 x = 0, y = 0;
            while (*a >= '0' && *a <= '9') x = x * 10 + (unsigned)(*a++ - '0');
            while (*b >= '0' && *b <= '9') y = y * 10 + (unsigned)(*b++ - '0');
            if (x != y) return x < y ? -1 : 1;
        } else {
            unsigned char ca = (unsigned char)*a, cb = (unsigned char)*b;
            if (ca != cb) return ca < cb ? -1 : 1;
            a++;
            b++;
        }
    }
    if (*a) return 1;
    if (*b) return -1;
    return 0;
}

/* src/repeats.rs:13-29: split(':')[0], split(':')[1], then split('-')[0] / [1] parsed as u32 */
int orc_parse_region(const char *reg, char *chrom_buf, size_t cap, uint32_t *start, uint32_t *end) {
    const char *c1 = strchr(reg, ':');
    if (!c1) return ORC_PANIC_PARSE; /* index 1 out of bounds */
    size_t cl = (size_t)(c1 - reg);
    if (cl + 1 > cap) return ORC_PANIC_PARSE;
    memcpy(chrom_buf, reg, cl);
    chrom_buf[cl] = 0;
    const char *iv = c1 + 1;
    const char *c2 = strchr(iv, ':');
    size_t ivlen = c2 ? (size_t)(c2 - iv) : strlen(iv);
    const char *d1 = memchr(iv, '-', ivlen);
    if (!d1) return ORC_PANIC_PARSE; /* split('-')[1] out of bounds (start parses first, but both panic) */
    size_t l0 = (size_t)(d1 - iv);
    const char *s1 = d1 + 1;
    size_t rem = ivlen - l0 - 1;
    const char *d2 = memchr(s1, '-', rem);
    size_t l1 = d2 ? (size_t)(d2 - s1) : rem;
    if (!parse_u32(iv, l0, start)) return ORC_PANIC_PARSE;
    if (!parse_u32(s1, l1, end)) return ORC_PANIC_PARSE;
    return ORC_OK;
}

/* src/repeats.rs:96-115 */
int orc_check_interval(uint32_t start, uint32_t end, int64_t chrom_len) {
    if (end < start) return ORC_PANIC_LOCUS;                               /* :102-104 */
    if (chrom_len >= 0 && (int64_t)end < chrom_len) return ORC_OK;          /* :108-110 */
    return ORC_PANIC_LOCUS;                                                /* :112-114 */
}
