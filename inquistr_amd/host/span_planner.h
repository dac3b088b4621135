// span_planner.h — host side of the device front end: which bytes of the BAM a group of loci needs.
//
// The reference asks htslib for every locus separately (bam.fetch((tid, start-10, end+10)),
// src/call.rs:288,338); here the loci, sorted by (contig, start), are cut into SPANS: runs of loci plus the
// whole BGZF blocks that hold every record overlapping any of them, as one or more segments of the file
// (loci far apart get their own segment, so what lies between is neither read nor inflated).  The host never
// inflates: it reads the compressed bytes, walks the 18-byte BGZF headers for the block table and takes
// from the .bai (a) where to start (linear index), (b) where it can stop (the first chunk of a bin that
// starts behind the last window: everything at smaller positions lies in front of it in a coordinate-
// sorted file) and (c) record-start anchors inside the range (chunk begins + linear index entries).
// The rest - inflate, record scan, overlap join, calling - is inq_call_span() on the GPU.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
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
    std::mutex mu_;  // a contig's anchors are built on first use, by the planner or the loader, whoever comes first
};

// A run of consecutive BGZF blocks: from the block of vo_begin to the block of vo_limit.
struct Segment {
    uint64_t vo_begin = 0;  // first record to look at
    uint64_t vo_limit = 0;  // record start (or end of data) behind everything needed
    int tid_first = -1, tid_last = -1;  // contigs whose records lie in the segment
};

struct SpanPlan {
    std::vector<Segment> segs;  // ascending, disjoint
    std::vector<uint32_t> locus_index;  // into the target list
    std::vector<int32_t> locus_tid;
    std::vector<uint32_t> locus_start, locus_end;
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
        int tid;
        uint32_t start, end, index;
    };
    const BamFile &bam_;
    BaiAnchors anch_;
    uint64_t max_comp_;
    uint64_t gap_ = 128ull << 10;  // a gap of compressed bytes up to this is read through rather than skipped
    std::vector<Locus> loci_;  // sorted by (tid, start)
    size_t j_ = 0;
};

// One loaded span: compressed bytes (caller's buffer), block table, anchors.
struct SpanData {
    std::vector<inq_bgzf_block_t> blocks;
    std::vector<uint64_t> anchors, anchor_stop;
    uint64_t comp_bytes = 0;
    uint64_t file_begin = 0;  // file offset of the first segment
    double ms_read = 0, ms_tables = 0;  // where load() spent its time: the parallel copy, the block table + anchors (INQ_TIMING=2 prints them)
};

// CPU seconds of a pipeline's threads by kind (INQ_TIMING prints them with the process's total: what is left is the runtime's own
// threads) - who uses the cores of a caller that has few
struct CpuByKind {
    std::atomic<uint64_t> readers_us{0}, loader_us{0}, uploader_us{0};
};
CpuByKind &cpu_by_kind();
uint64_t thread_cpu_us();  // CPU time of the calling thread so far

// The threads that copy file bytes into a span buffer (pread from the page cache): made once per file, not once per span, and
// spread over the L3 domains (CCDs) of the NUMA node the GPU hangs on.  Why spread: a copy is bound by memory bandwidth, and on
// the two-socket EPYC hosts measured a CCD's link to memory carries a fraction of the socket's; left to the scheduler, sixteen
// fresh threads per span land where the loader thread is.  Measured on the 12.8 GB file (profiles/r04_results/
// span_loop_12.8GB_after_changes.txt): span loop 51 - 54 GB/s bound, 37 - 53 (median 45) unbound.  Threads are bound to a CCD's
// CPUs, not to one CPU.  (Round 3's "one slow run in five" is NOT this: it is the second read of a freshly written file)
class IoPool {
public:
    // numa_node < 0: every CPU this process may use; pin = false: plain threads (INQ_IO_PIN=0)
    // group_offset: thread i is bound to L3 domain (i + group_offset * n_threads) modulo their number, so that several pools on one
    // host (one per rank / device part) start on different domains
    IoPool(int n_threads, int numa_node, bool pin, int group_offset = 0);
    ~IoPool();
    IoPool(const IoPool &) = delete;
    // fn(k) for every k < n_jobs, on the pool's threads and the caller's; returns when all have run
    void run(size_t n_jobs, const std::function<void(size_t)> &fn);
    int threads() const { return (int)workers_.size() + 1; }
    const std::string &layout() const { return layout_; }  // for INQ_TIMING=2
    std::string last_cpus() const;  // the CPU every thread found itself on when the last run() began (INQ_TIMING=2)

private:
    void work(int id);
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    std::atomic<size_t> next_{0};
    size_t n_jobs_ = 0;
    uint64_t generation_ = 0;
    int busy_ = 0;
    bool stop_ = false;
    std::string layout_;
    std::vector<int> cpu_at_;  // [threads()]
};

class SpanLoader {
public:
    SpanLoader() = default;
    ~SpanLoader();
    bool open(const std::string &path, std::string *err);
    uint64_t file_size() const { return size_; }
    // Byte range [begin, end) of whole BGZF blocks a segment needs (reads one block header at the limit).
    bool extent(const Segment &g, uint64_t *begin, uint64_t *end, std::string *err) const;
    // Total compressed bytes of a plan (sum of its segments' extents).
    bool total_bytes(const SpanPlan &p, uint64_t *bytes, std::string *err) const;
    // Reads every segment into buf (back to back) with n_threads preads, walks the block headers, maps the
    // .bai anchors inside each segment to offsets in the inflated byte string.
    // (pool may be null: n_threads threads are started for this one call)
    bool load(const SpanPlan &p, BaiAnchors &anch, uint8_t *buf, int n_threads, SpanData &out, std::string *err, IoPool *pool = nullptr) const;

private:
    int fd_ = -1;
    uint64_t size_ = 0;
};

}  // namespace inqhost
