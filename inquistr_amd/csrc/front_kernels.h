// front_kernels.h — argument blocks and launchers of the device front end (bgzf_inflate.hip, bam_scan.hip),
// shared with span.hip which sequences them behind inq_call_span().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inquistr_hip.h"

namespace inq {

struct InflateArgs {
    const uint8_t *comp;  // whole BGZF blocks; >= 64 readable bytes behind comp_bytes
    uint64_t comp_bytes;
    const inq_bgzf_block_t *blocks;
    uint64_t n_blocks;
    uint8_t *out;  // >= 16 writable/readable bytes behind out_bytes
    uint64_t out_bytes;
    uint32_t *block_status;  // [n_blocks] or null
    unsigned int *err;       // OR of the INQ_INFLATE_* bits of all blocks
    uint32_t debug_flags;    // timing experiments only: 1 = drop literal stores, 2 = drop match copies, 4 = block_status receives shader kilo-cycles
    uint32_t verify_crc;     // also check every block against the CRC32 of its trailer (comp holds whole blocks)
    uint32_t algo;           // 0 = one workgroup per block (bgzf_inflate_wg.hip), 1 = one lane per block (bgzf_inflate.hip), 2 = by block count
    // workgroup kernel: scratch for the symbols of a round as decoded by the counting passes (inflate_token_words(n_blocks)
    // u32), so that the commit does not decode them again; null = the commit decodes (round 2's form)
    uint32_t *tokens;
    uint32_t lit_pairs;  // workgroup kernel: 1 = the symbol loop looks for a second literal behind a literal (literal-heavy data)
};
uint64_t inflate_token_words(uint64_t n_blocks);
void launch_bgzf_inflate(const InflateArgs &a, hipStream_t s);
void launch_bgzf_inflate_wg(const InflateArgs &a, hipStream_t s);  // the inflate alone, no CRC pass

// scan-side status bits (FrontStatus::err)
constexpr uint32_t FS_CHAIN = 0x01u;      // a record chain does not land on the next index anchor / record shorter than its fixed part
constexpr uint32_t FS_RECORD = 0x02u;     // field lengths of a record exceed its block_size
constexpr uint32_t FS_UNSORTED = 0x04u;   // (contig, position) decreases along the file
constexpr uint32_t FS_HP_TYPE = 0x08u;    // HP aux of a read offered to a locus is neither C nor i (src/call.rs:482-491 panics)
constexpr uint32_t FS_SA_TYPE = 0x10u;    // SA aux of such a read is not Z (src/call.rs:429-432 panics)
constexpr uint32_t FS_SA_FORMAT = 0x20u;  // SA string the reference cannot index / parse (src/call.rs:439-451 panics)
constexpr uint32_t FS_TOO_BIG = 0x40u;    // more CIGAR than 32-bit offsets in units of 16 bytes can address

struct FrontStatus {
    unsigned int err;         // FS_* bits
    unsigned int inflate;     // INQ_INFLATE_* bits
    unsigned long long n_valid;      // placed records (tid >= 0): a prefix of the record list
    unsigned long long first_bad;    // smallest record index that raised an FS_* bit (for the message)
    unsigned int max_reads;   // deepest locus
    unsigned int pad;
};

// per record, written by the parse kernel
struct RecInfo {
    uint64_t cigar_src;  // offset of the CIGAR words in the inflated bytes (the CG:B,I payload for long CIGARs)
    uint64_t sa_off;     // offset of the SA value, 0 = no SA tag
    uint32_t sa_type;    // BAM aux type char of SA
    uint32_t err;        // FS_HP_TYPE / FS_SA_* bits this read raises if a locus is offered it
};

struct ScanArgs {
    const uint8_t *u;  // inflated bytes, >= 8 readable bytes of padding
    uint64_t u_bytes;
    const uint64_t *anchors;
    const uint64_t *anchor_stop;  // where the chain of anchor i ends; INQ_ANCHOR_SEGMENT_END set = end of a segment
    uint64_t n_anchors;
    uint32_t *anchor_cnt;   // [n_anchors]
    uint64_t *anchor_base;  // [n_anchors + 1] exclusive scan of anchor_cnt
    uint64_t *rec_off;      // [n_records]
    uint64_t n_records;
    inq_read_t *reads;      // [n_records]
    RecInfo *info;          // [n_records]
    int64_t *key;           // [n_records] (tid << 32) | (pos + 1): ascending in a coordinate-sorted file
    int64_t *endkey;        // [n_records] (tid << 32) + (endpos + 1)
    int64_t *pmax;          // [n_records] running maximum of endkey
    uint64_t *cig_off;      // [n_records + 1] exclusive scan of CIGAR sizes in units of 4 words
    uint32_t *cigar;        // gathered CIGAR words
    uint64_t n_cigar_units;
    uint32_t unphased;
    const int32_t *locus_tid;
    const uint32_t *locus_start, *locus_end;
    uint64_t n_loci;
    uint32_t *locus_cnt;       // [n_loci]
    uint64_t *locus_pair_off;  // [n_loci + 1]
    uint32_t *pair_read;
    FrontStatus *st;
    // a span appended to the batch of earlier spans (inq_call_span_deferred): the CIGAR units / reads that are already there
    uint32_t unit_base, read_base;
    uint32_t gather_nt;  // cigar_gather_kernel stores with the non-temporal policy
};

void launch_chain_count(const ScanArgs &a, hipStream_t s);
void launch_chain_fill(const ScanArgs &a, hipStream_t s);
void launch_record_parse(const ScanArgs &a, hipStream_t s);
void launch_cigar_gather(const ScanArgs &a, uint64_t n_valid, hipStream_t s);
void launch_join_count(const ScanArgs &a, uint64_t n_valid, hipStream_t s);
void launch_join_fill(const ScanArgs &a, uint64_t n_valid, hipStream_t s);
// dst[i] = src[i] + base, i < n (the CSR offsets of an appended span)
void launch_offset_copy(uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t base, hipStream_t s);

// d1[index[i]] = p1[i], d2[index[i]] = p2[i] for i < n (index[i] < cap); p[i] = v
void launch_scatter_rows(const double *p1, const double *p2, const uint32_t *index, double *d1, double *d2, uint64_t n, uint64_t cap, hipStream_t s);
void launch_fill_f64(double *p, uint64_t n, double v, hipStream_t s);

// exclusive prefix sums (out[n] = total) and the running maximum; `tmp` holds ceil(n / 4096) + 1 u64
void launch_scan_u32_to_u64(const uint32_t *in, uint64_t *out, uint64_t n, uint64_t *tmp, hipStream_t s);
// CIGAR sizes: units(i) = ceil(n_cigar / 4) for i < *n_valid, else 0
void launch_scan_cigar_units(const inq_read_t *reads, const unsigned long long *n_valid, uint64_t *out, uint64_t n, uint64_t *tmp,
                             hipStream_t s);
void launch_scan_max_i64(const int64_t *in, int64_t *out, uint64_t n, uint64_t *tmp, hipStream_t s);
inline uint64_t scan_tmp_words(uint64_t n) { return (n + 4095) / 4096 + 2; }

// outlier.hip
struct OutlierArgs {
    const float *values;      // [n_rows][stride], row i holds row_len[i] values
    const uint32_t *row_len;  // [n_rows]
    uint64_t n_rows;
    uint32_t stride;
    uint32_t minsize;
    float zscore_cutoff;
    uint32_t mincluster;
    uint8_t *flags;  // [n_rows][stride]: 1 = outlier
    uint8_t *keep;   // [n_rows]: INQ_OUTLIER_ROW_*
};
// z-score: rows of at most kOutlierTileMaxStride values go through an LDS tile (one read of the matrix; use_tile), wider ones work on
// a transposed copy ([stride][rows_padded] floats, caller's scratch); flags must be zero-filled
constexpr uint32_t kOutlierTileMaxStride = 256;
inline uint64_t outlier_rows_padded(uint64_t n_rows) { return (n_rows + 63) / 64 * 64; }
void launch_outlier(const OutlierArgs &a, int method, float *transposed, hipStream_t s, bool use_tile = true);

}  // namespace inq
