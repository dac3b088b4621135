// span_pipeline.h - host half of the device front end: span buffers, the loader and uploader threads (span_pipeline.cc)
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/inquistr_host.h"
#include "span_planner.h"

namespace inqhost {

constexpr int kNumaUnknown = -2;
int guess_gpu_numa_node(int device);
void prefer_gpu_node_for_this_thread(int device);
void prefer_numa_node(void *p, size_t len, int node);
double stamp_ms();  // milliseconds since the library was loaded (INQ_TIMING=2 stamps)

// Span buffers that outlive one file: a cohort run (inq_session) hands the buffers of file k to file k + 2 instead of unmapping
// and re-faulting a GB of pages per file.
struct HostBufPool {
    struct B {
        uint8_t *p = nullptr;
        size_t cap = 0;
        bool pinned = false;
        int node = -1;  // NUMA node the mapping prefers, -1 = none
    };
    std::mutex mu;
    std::vector<B> free_list;
    bool take(size_t bytes, bool pinned, B *out) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].cap >= bytes && free_list[i].pinned == pinned) {
                *out = free_list[i];
                free_list.erase(free_list.begin() + (long)i);
                return true;
            }
        return false;
    }
    void give(const B &b) {
        std::lock_guard<std::mutex> g(mu);
        free_list.push_back(b);
    }
    ~HostBufPool() {
        for (auto &b : free_list) {
            if (b.pinned) inq_free_pinned(b.p);
            else ::munmap(b.p, b.cap);
        }
    }
};

// Device front end, host half: a loader thread plans the spans, reads their compressed bytes (parallel pread;
// into pageable memory by default: pinning a few hundred MB costs more than the staged copy it saves,
// INQ_SPAN_PINNED=1 switches) and builds block tables and anchors, two spans ahead
// of the caller, who feeds inq_call_span().  (INQ_SPAN_PINNED exists only in builds with -DINQ_DEBUG_ENV.)
class SpanPipeline {
public:
    // span buffers (and device staging slots) per file: one being read, one being uploaded, one being inflated / scanned, and one of
    // slack - read (4 ms), upload (4.9 ms) and device (4.5 ms) of a 268 MB span are so close that with three any jitter stalled all
    static constexpr int kSlotsPerSet = INQ_SPAN_SLOTS / 2;
    struct Item {
        SpanPlan plan;
        SpanData data;
        uint8_t *buf = nullptr;
        size_t cap = 0;
        bool pinned = false;
        int node = -1;       // NUMA node the buffer was mapped for
        bool registered = false;  // page-locked in place (inq_pin_host) by the uploader
        int slot = 0;        // index of this item: also its device-side staging slot
        bool staged = false; // the loader already uploaded it (inq_span_stage)
    };
    // stage: called on the loader thread for every loaded span with the filled inq_span_t; returns true when the
    // span now sits in device slot `slot` (the upload then overlaps the caller's work on earlier spans)
    using StageFn = std::function<bool(const inq_span_t &, int slot)>;
    // ... in two steps (inq_span_stage_begin / _wait): with `wait` given, `stage` only enqueues, and the uploader keeps two spans
    // enqueued so that the copy engine never waits for the host between them
    using WaitFn = std::function<bool(int slot)>;
    // slot_base: 0 or kSlotsPerSet, the set of device-side staging slots this pipeline uploads into; pool: where span buffers come from
    // and go back to (may be null: mapped and unmapped by the pipeline)
    SpanPipeline(const std::string &bam_path, const BamFile &hdr, const std::vector<RepeatInterval> &targets,
                 uint64_t max_comp_bytes, int n_threads, bool pinned, StageFn stage = nullptr, int slot_base = 0, HostBufPool *pool = nullptr,
                 std::function<void()> gate = nullptr, std::function<int()> numa_query = nullptr, int device = 0,
                 std::function<void()> runtime_gate = nullptr, WaitFn stage_wait = nullptr, int io_group_offset = 0)
        : path_(bam_path), planner_(hdr, targets, max_comp_bytes), n_threads_(std::max(n_threads, 1)), pinned_(pinned),
          stage_(std::move(stage)), stage_wait_(std::move(stage_wait)), pool_(pool), gate_(std::move(gate)), gate_registered_(std::move(runtime_gate)),
          numa_query_(std::move(numa_query)), device_(device), io_group_offset_(io_group_offset) {
        for (int i = 0; i < kSlotsPerSet; ++i) slots_[i].slot = slot_base + i;
        int use = kSlotsPerSet;
#ifdef INQ_DEBUG_ENV
        if (const char *e = std::getenv("INQ_SPAN_BUFFERS")) use = std::min(kSlotsPerSet, std::max(2, std::atoi(e)));  // A/B only
#endif
        for (int i = 0; i < use; ++i) free_.push_back(&slots_[i]);
        th_ = std::thread([this] { run(); });
        if (stage_) up_ = std::thread([this] { run_uploads(); });  // span k uploads while span k + 1 is being read
    }
    ~SpanPipeline() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_free_.notify_all();
        cv_loaded_.notify_all();
        th_.join();
        if (up_.joinable()) up_.join();
        for (auto &it : slots_) release_buf(it);
    }
    static void fill_span(const Item &it, inq_span_t *sp) {  // the data part; the caller adds minlen / support / unphased
        std::memset(sp, 0, sizeof *sp);
        sp->comp = it.buf;
        sp->comp_bytes = it.data.comp_bytes;
        sp->blocks = it.data.blocks.data();
        sp->n_blocks = it.data.blocks.size();
        sp->anchors = it.data.anchors.data();
        sp->anchor_stop = it.data.anchor_stop.data();
        sp->n_anchors = it.data.anchors.size();
        sp->locus_tid = it.plan.locus_tid.data();
        sp->locus_start = it.plan.locus_start.data();
        sp->locus_end = it.plan.locus_end.data();
        sp->n_loci = it.plan.locus_start.size();
    }
    // 1 = item, 0 = done, -1 = error
    int next(Item *&out, std::string *err) {
        std::unique_lock<std::mutex> g(mu_);
        cv_item_.wait(g, [&] { return !ready_.empty() || done_ || failed_; });
        if (!ready_.empty()) {
            out = ready_.front();
            ready_.pop_front();
            return 1;
        }
        if (failed_) {
            *err = err_;
            return -1;
        }
        return 0;
    }
    void release(Item *it) {
        std::lock_guard<std::mutex> g(mu_);
        free_.push_back(it);
        cv_free_.notify_one();
    }
    int io_threads() const { return n_threads_; }

private:
    void release_buf(Item &it);
    bool fit(Item &it, size_t bytes);
    void fail(const std::string &m) {
        std::lock_guard<std::mutex> g(mu_);
        failed_ = true;
        err_ = m;
        cv_item_.notify_all();
        cv_loaded_.notify_all();
    }
    void run();
    // uploads in file order, one span behind the reader
    void run_uploads();

    std::string path_;
    SpanPlanner planner_;
    int n_threads_;
    bool pinned_;
    StageFn stage_;
    WaitFn stage_wait_;
    HostBufPool *pool_ = nullptr;
    std::function<void()> gate_;
#ifdef INQ_DEBUG_ENV
    bool register_ = std::getenv("INQ_SPAN_REGISTER") && std::getenv("INQ_SPAN_REGISTER")[0] == '1';
#else
    bool register_ = false;  // (page-locking the span buffers in place: an experiment of round 3)
#endif
    std::function<void()> gate_registered_;  // waits for the runtime before the first registration
    std::function<int()> numa_query_;  // the GPU's NUMA node, kNumaUnknown while the context is not there yet, -1 = do not place
    int device_ = 0;
    int io_group_offset_ = 0;  // which L3 domain the first reader thread is bound to: sharers of one host start at different ones
    bool verbose_ = std::getenv("INQ_TIMING") && std::getenv("INQ_TIMING")[0] == '2';
    Item slots_[kSlotsPerSet];
    std::vector<Item *> free_;
    std::deque<Item *> ready_, loaded_;
    std::thread th_, up_;
    std::mutex mu_;
    std::condition_variable cv_item_, cv_free_, cv_loaded_;
    bool stop_ = false, done_ = false, failed_ = false, load_done_ = false;
    std::string err_;
};

uint64_t span_bytes_from_env();


}  // namespace inqhost
