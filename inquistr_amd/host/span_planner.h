// span_planner.h — host side of the device front end: which bytes of the BAM a group of loci needs.
//
// The reference asks htslib for every locus separately (bam.fetch((tid, start-10, end+10)),
// src/call.rs:288,338); here the loci of a contig, sorted by start, are cut into SPANS: runs of loci plus
// the range of whole BGZF blocks that holds every record overlapping any of them.  The host never
// inflates: it reads the compressed bytes, walks the 18-byte BGZF headers for the block table and takes
// from the .bai (a) where to start (linear index), (b) where it can stop (the first chunk of a bin that
// starts behind the last window: everything at smaller positions lies in front of it in a coordinate-
// sorted file) and (c) record-start anchors inside the range (chunk begins + linear index entries).
// The rest - inflate, record scan, overlap join, calling - is inq_call_span() on the GPU.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/inquistr_hip.h"
#include "bam_reader.h"
#include "targets.h"

namespace inqhost {

// per-contig views of the .bai, built on first use
struct BaiAnchors {
    struct PerRef {
        bool built = false;
        std::vector<uint64_t> anchors;                   // sorted unique virtual offsets that are record starts
        std::vector<std::pair<int64_t, uint64_t>> bins;  // (first position of the bin, suffix-minimum of chunk begins)
    };
    explicit BaiAnchors(const BaiIndex &idx) : idx_(idx), refs_(idx.refs.size()) {}
    const PerRef &ref(int tid);
    // a virtual offset (a record start, or the end of the contig's data) that every record of `tid` with
    // pos < x lies in front of
    uint64_t limit_after(int tid, int64_t x);

private:
    const BaiIndex &idx_;
    std::vector<PerRef> refs_;
};

struct SpanPlan {
    int tid = -1;
    std::vector<uint32_t> locus_index;  // into the target list
    std::vector<uint32_t> locus_start, locus_end;
    uint64_t vo_begin = 0;  // first record to look at
    uint64_t vo_limit = 0;  // record start (or end of data) behind everything needed
};

class SpanPlanner {
public:
    SpanPlanner(const BamFile &bam, const std::vector<RepeatInterval> &targets, uint64_t max_comp_bytes);
    // false when no span is left.  Loci without any record at or behind them never appear in a span
    // (their rows stay NaN).
    bool next(SpanPlan &out);
    BaiAnchors &anchors() { return anch_; }

private:
    struct Locus {
        uint32_t start, end, index;
    };
    struct Group {
        int tid;
        std::vector<Locus> loci;
    };
    const BamFile &bam_;
    BaiAnchors anch_;
    uint64_t max_comp_;
    std::vector<Group> groups_;
    size_t g_ = 0, j_ = 0;
};

// One loaded span: compressed bytes (caller's buffer), block table, anchors.
struct SpanData {
    std::vector<inq_bgzf_block_t> blocks;
    std::vector<uint64_t> anchors;
    uint64_t comp_bytes = 0;
    uint64_t file_begin = 0;  // file offset of comp[0]
};

class SpanLoader {
public:
    SpanLoader() = default;
    ~SpanLoader();
    bool open(const std::string &path, std::string *err);
    uint64_t file_size() const { return size_; }
    // Byte range [begin, end) of whole BGZF blocks a plan needs (reads one block header at the limit).
    bool extent(const SpanPlan &p, uint64_t *begin, uint64_t *end, std::string *err) const;
    // Reads [begin, end) into buf with n_threads preads, walks the block headers, maps the plan's .bai
    // anchors to offsets in the inflated byte string.
    bool load(const SpanPlan &p, BaiAnchors &anch, uint64_t begin, uint64_t end, uint8_t *buf, int n_threads, SpanData &out,
              std::string *err) const;

private:
    int fd_ = -1;
    uint64_t size_ = 0;
};

}  // namespace inqhost
