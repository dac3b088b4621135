// driver.cc — `call::genotype_repeats` (src/call.rs:76-159) on top of the front end and the HIP library,
// plus the C ABI of include/inquistr_host.h.
#include <sys/mman.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/inquistr_host.h"
#include "front_end.h"
#include "inq_text.h"
#include "sa2d.h"
#include "span_planner.h"
#include "targets.h"

using namespace inqhost;

// auto front-end choice: the device front end inflates a BGZF block per GPU lane, which takes ~40 ms however
// few blocks there are; the CPU sweep inflates ~70 MB/s of BAM per thread.  Below this many compressed bytes
// per host thread the sweep is as quick (16 MiB at -t 16, 1 MiB at -t 1; round 1's lane-per-block inflate put the line at 3 MiB:
// profiles/r02_results/front_end_choice.txt).
static constexpr uint64_t kDeviceFrontMinBytesPerThread = 1ull << 20;

namespace {

// INQ_TIMING=2 stamps every stage with milliseconds since the library was loaded (about the start of the process)
const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
double stamp_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_t0).count(); }

void set_err(char *buf, size_t cap, const std::string &m) {
    if (buf && cap) std::snprintf(buf, cap, "%s", m.c_str());
}

bool starts_with(const std::string &s, const char *p) { return s.compare(0, std::strlen(p), p) == 0; }
bool ends_with(const std::string &s, const char *p) {
    size_t n = std::strlen(p);
    return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}

bool is_file(const std::string &p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

struct Prepared {
    std::unique_ptr<BamFile> bam;
    std::vector<RepeatInterval> targets;
    std::string sample;
};

// src/call.rs:87-102 + get_targets :182-202.  Returns an exit status.
// A cohort is called with ONE BED: inside a session (inquistr cohort / serve) the parsed and validated target list of the last BED is
// kept and taken again when the file is the same (device, inode, size, modification time) and the BAM's contigs are (names and lengths
// decide every check of from_bed, src/repeats.rs:96-115).  100 000 targets: 12 - 25 ms of a 70 ms call.
struct BedCache {
    std::mutex mu;
    std::string path;
    uint64_t dev = 0, ino = 0, size = 0;
    int64_t mtime_ns = 0;
    std::map<std::string, uint64_t> lengths;
    TargetsResult tr;
    bool valid = false;
};

int prepare(const inq_call_args_t *a, Prepared &P, std::string &msg, BedCache *bed_cache = nullptr) {
    if (!a || !a->bam) {
        msg = "no BAM given";
        return INQ_EXIT_ERROR;
    }
    const std::string bamp = a->bam;
    const bool remote = starts_with(bamp, "s3") || starts_with(bamp, "https://");
    if (!is_file(bamp) && !remote) {  // :87-90
        msg = "ERROR: path to bam file " + bamp + " is not valid!";
        return INQ_EXIT_ERROR;
    }
    if (remote) {  // :227-240 needs libcurl + htslib network code: not in this build
        msg = "remote inputs (s3://, https://) are not supported by this build";
        return INQ_EXIT_ERROR;
    }
    if (ends_with(bamp, ".cram")) {  // :245-259 needs htslib's CRAM codecs: not in this build
        msg = "CRAM input is not supported by this build (BAM + .bai only)";
        return INQ_EXIT_ERROR;
    }
    P.sample = a->sample_name ? std::string(a->sample_name) : sample_name_from_path(bamp);  // :91-100
    // get_chrom_lengths_from_bam_header opens the BAM before the target arguments are looked at (:187)
    const auto t_open = std::chrono::steady_clock::now();
    P.bam.reset(new BamFile(1));  // header + index; with -t > 1 every sweep worker opens its own reader
    std::string e;
    if (!P.bam->open(bamp, &e)) {
        msg = "Error opening local BAM: " + e;  // :242-243
        return INQ_EXIT_PANIC;
    }
    auto lengths = P.bam->sq_lengths(&e);
    if (!e.empty()) {
        msg = e;
        return INQ_EXIT_PANIC;
    }
    const auto t_targets = std::chrono::steady_clock::now();
    TargetsResult tr;
    if (a->region && !a->region_file)
        tr = targets_from_string(a->region, lengths);  // :190
    else if (!a->region && a->region_file) {
        struct stat sb;
        const bool have_stat = bed_cache && ::stat(a->region_file, &sb) == 0;
        bool hit = false;
        if (have_stat) {
            std::lock_guard<std::mutex> lk(bed_cache->mu);
            hit = bed_cache->valid && bed_cache->path == a->region_file && bed_cache->dev == (uint64_t)sb.st_dev && bed_cache->ino == (uint64_t)sb.st_ino &&
                  bed_cache->size == (uint64_t)sb.st_size &&
                  bed_cache->mtime_ns == (int64_t)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec && bed_cache->lengths == lengths;
            if (hit) tr = bed_cache->tr;
        }
        if (!hit) {
            tr = targets_from_bed(a->region_file, lengths);  // :192-195
            if (have_stat) {
                std::lock_guard<std::mutex> lk(bed_cache->mu);
                bed_cache->path = a->region_file, bed_cache->dev = (uint64_t)sb.st_dev, bed_cache->ino = (uint64_t)sb.st_ino;
                bed_cache->size = (uint64_t)sb.st_size, bed_cache->mtime_ns = (int64_t)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec;
                bed_cache->lengths = lengths, bed_cache->tr = tr, bed_cache->valid = true;
            }
        }
    } else {
        msg = "ERROR: Specify a region string (-r) or a region_file (-R)!";  // :197-200
        return INQ_EXIT_ERROR;
    }
    if (tr.panicked) {
        msg = tr.message;
        return INQ_EXIT_PANIC;
    }
    for (const auto &t : tr.data) {
        if (t.start < 10) {  // src/call.rs:285,335: `repeat.start - 10` underflows u32 -> fetch fails -> expect() panics
            msg = "Failed to fetch region (" + t.chrom + ":" + std::to_string(t.start) + "-" + std::to_string(t.end) +
                  ": start - 10 underflows)";
            return INQ_EXIT_PANIC;
        }
    }
    P.targets.swap(tr.data);
    if (const char *tm = std::getenv("INQ_TIMING"); tm && tm[0] == '2') {
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[inq prepare] header + index %.2f ms, targets (%zu) %.2f ms\n",
                     std::chrono::duration<double, std::milli>(t_targets - t_open).count(), P.targets.size(),
                     std::chrono::duration<double, std::milli>(now - t_targets).count());
    }
    return INQ_EXIT_OK;
}

bool write_all(int fd, const char *data, size_t len) {
    size_t off = 0;
    while (off < len) {
        ssize_t w = ::write(fd, data + off, len - off);
        if (w <= 0) return false;
        off += (size_t)w;
    }
    return true;
}
bool write_all(int fd, const std::string &s) { return write_all(fd, s.data(), s.size()); }

}  // namespace

// Several sweep workers, each with its own reader, over contiguous slices of the position-sorted
// target list; batches flow to the caller through a bounded queue.  Replaces the reference's
// rayon par_bridge over loci (src/call.rs:115-118) on the decode side.
class ParallelFrontEnd {
public:
    struct Item {
        HostBatch batch;
        std::vector<uint32_t> index;  // position of each batch locus in the full target list
    };
    ParallelFrontEnd(const std::string &bam_path, BamFile &hdr, const std::vector<RepeatInterval> &targets, bool unphased,
                     int n_workers, uint64_t max_words = 0)
        : path_(bam_path), targets_(targets), unphased_(unphased), max_words_(max_words) {
        std::vector<uint32_t> order(targets.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            int ta = hdr.tid(targets[a].chrom), tb = hdr.tid(targets[b].chrom);
            if (ta != tb) return ta < tb;
            return targets[a].start < targets[b].start;
        });
        const size_t n = order.size();
        const size_t n_slices = std::max<size_t>(1, std::min<size_t>(n, (size_t)n_workers * 3));
        for (size_t k = 0; k < n_slices; ++k) {
            size_t lo = n * k / n_slices, hi = n * (k + 1) / n_slices;
            if (hi > lo) slices_.emplace_back(order.begin() + lo, order.begin() + hi);
        }
        live_ = n_workers;
        for (int w = 0; w < n_workers; ++w) pool_.emplace_back([this] { work(); });
    }
    ~ParallelFrontEnd() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_space_.notify_all();
        for (auto &t : pool_) t.join();
    }
    // hand a consumed item back so its vectors' capacity is reused (no mmap/munmap churn while the
    // HIP runtime is pinning pages on another thread)
    void recycle(Item &&it) {
        std::lock_guard<std::mutex> g(mu_);
        if (free_.size() < 16) free_.push_back(std::move(it));
    }
    // 1 = item, 0 = done, -1 = error
    int next(Item &out, std::string *err, bool *panic) {
        std::unique_lock<std::mutex> g(mu_);
        cv_item_.wait(g, [&] { return !q_.empty() || live_ == 0 || failed_; });
        if (failed_) {
            *err = err_;
            *panic = panic_;
            return -1;
        }
        if (q_.empty()) return 0;
        out = std::move(q_.front());
        q_.pop_front();
        cv_space_.notify_one();
        return 1;
    }

private:
    void work() {
        BamFile bam(1);
        std::string e;
        bool ok = bam.open(path_, &e);
        for (;;) {
            size_t k = next_slice_.fetch_add(1);
            if (!ok || k >= slices_.size()) break;
            std::vector<RepeatInterval> sub;
            sub.reserve(slices_[k].size());
            for (uint32_t i : slices_[k]) sub.push_back(targets_[i]);
            FrontEnd fe(bam, sub, unphased_);
            if (max_words_) fe.set_max_batch_words(max_words_);
            for (;;) {
                Item it;
                {
                    std::lock_guard<std::mutex> g(mu_);
                    if (!free_.empty()) {
                        it = std::move(free_.back());
                        free_.pop_back();
                    }
                }
                bool panic = false;
                int rc = fe.next(it.batch, &e, &panic);
                if (rc < 0) {
                    std::lock_guard<std::mutex> g(mu_);
                    if (!failed_) failed_ = true, err_ = e, panic_ = panic;
                    ok = false;
                    break;
                }
                if (rc == 0) break;
                it.index.resize(it.batch.locus_index.size());
                for (size_t j = 0; j < it.index.size(); ++j) it.index[j] = slices_[k][it.batch.locus_index[j]];
                std::unique_lock<std::mutex> g(mu_);
                cv_space_.wait(g, [&] { return q_.size() < 8 || stop_; });
                if (stop_) return;
                q_.push_back(std::move(it));
                cv_item_.notify_one();
            }
            if (!ok) break;
        }
        if (!ok && !e.empty()) {
            std::lock_guard<std::mutex> g(mu_);
            if (!failed_) failed_ = true, err_ = e, panic_ = true;
        }
        std::lock_guard<std::mutex> g(mu_);
        --live_;
        cv_item_.notify_all();
    }

    std::string path_;
    const std::vector<RepeatInterval> &targets_;
    bool unphased_;
    uint64_t max_words_ = 0;
    std::vector<std::vector<uint32_t>> slices_;
    std::atomic<size_t> next_slice_{0};
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_item_, cv_space_;
    std::deque<Item> q_;
    std::vector<Item> free_;
    int live_ = 0;
    bool stop_ = false, failed_ = false, panic_ = false;
    std::string err_;
};

// NUMA.  A span buffer is read by the GPU's copy engine over PCIe, which hangs off ONE socket, and a pageable upload passes
// through the runtime on the thread that issues it.  On the two-socket hosts measured (profiles/r03_results/loader_numa_binding.txt,
// l2_seq_12.8GB_numa_modes.txt) an upload of 268 MB takes 4.86 - 4.97 ms when the issuing thread runs on the GPU's node and its
// memory comes from there, 5.5 - 6.3 ms otherwise - preferring the node for the memory alone (mbind of the span buffers, or
// MPOL_PREFERRED for the threads) changed nothing: it is the CPU side of the copy that has to be near.  That is the difference
// between an upload-bound and a device-bound span loop on SEQ-bearing files (loader wait 12 - 17 % -> 2 - 4 % of the loop, 39 - 41 ->
// 43 - 46.5 GB/s of compressed bytes).  So the threads that read, upload and start the runtime run on the CPUs of the GPU's node
// (cut with the mask they were given; left alone if that leaves nothing) and prefer its memory.  The node must be known before the
// runtime is up: it is read from sysfs for the device-th render node this process can really open.  INQ_NUMA_NODE=n overrides,
// -1 switches all of it off; INQ_NUMA_CPUS=0 keeps the memory preference but lets the threads run anywhere.
constexpr int kNumaUnknown = -2;

static int guess_gpu_numa_node(int device) {
    static std::mutex mu;
    static std::map<int, int> memo;
    std::lock_guard<std::mutex> g(mu);
    auto it = memo.find(device);
    if (it != memo.end()) return it->second;
    int node = -1;
    if (const char *e = std::getenv("INQ_NUMA_NODE")) node = std::atoi(e);
    else {
        int seen = 0;
        for (int minor = 128; minor < 128 + 64 && node == -1; ++minor) {
            char dev[64], path[128];
            std::snprintf(dev, sizeof dev, "/dev/dri/renderD%d", minor);
            const int fd = ::open(dev, O_RDWR | O_CLOEXEC);  // the device cgroup, not the permission bits, says which GPU is ours
            if (fd < 0) continue;
            ::close(fd);
            if (seen++ != device) continue;
            std::snprintf(path, sizeof path, "/sys/class/drm/renderD%d/device/numa_node", minor);
            if (FILE *f = std::fopen(path, "r")) {
                if (std::fscanf(f, "%d", &node) != 1) node = -1;
                std::fclose(f);
            }
            break;
        }
    }
    memo[device] = node;
    return node;
}

static void prefer_gpu_node_for_this_thread(int device) {
    const int node = guess_gpu_numa_node(device);
    if (node < 0 || node >= 1024) return;
    unsigned long mask[16] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    (void)::syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, sizeof mask * 8);
    const char *c = std::getenv("INQ_NUMA_CPUS");
    if (!(c && c[0] == '0')) {
        char path[128], buf[4096] = {0};
        std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE *f = std::fopen(path, "r");
        if (!f) return;
        if (!std::fgets(buf, sizeof buf, f)) buf[0] = 0;
        std::fclose(f);
        cpu_set_t set, have;
        CPU_ZERO(&set);
        CPU_ZERO(&have);
        if (sched_getaffinity(0, sizeof have, &have) != 0) return;
        for (char *p = buf; *p;) {  // "0-63,128-191"
            char *q;
            long a = std::strtol(p, &q, 10), b = a;
            if (q == p) break;
            if (*q == '-') b = std::strtol(q + 1, &q, 10);
            for (long k = a; k <= b && k < CPU_SETSIZE; ++k)
                if (CPU_ISSET((int)k, &have)) CPU_SET((int)k, &set);  // never beyond what the process was given
            if (*q != ',') break;
            p = q + 1;
        }
        if (CPU_COUNT(&set) >= 4) (void)sched_setaffinity(0, sizeof set, &set);
    }
}

static void prefer_numa_node(void *p, size_t len, int node) {
    if (node < 0 || node >= 1024) return;
    unsigned long mask[16] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    (void)::syscall(SYS_mbind, p, len, 1 /* MPOL_PREFERRED */, mask, sizeof mask * 8, 0);  // best effort: placement only
}

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
// of the caller, who feeds inq_call_span().
class SpanPipeline {
public:
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
    // slot_base: 0 or 3, the set of device-side staging slots this pipeline uploads into; pool: where span buffers come from
    // and go back to (may be null: mapped and unmapped by the pipeline)
    SpanPipeline(const std::string &bam_path, const BamFile &hdr, const std::vector<RepeatInterval> &targets,
                 uint64_t max_comp_bytes, int n_threads, bool pinned, StageFn stage = nullptr, int slot_base = 0, HostBufPool *pool = nullptr,
                 std::function<void()> gate = nullptr, std::function<int()> numa_query = nullptr, int device = 0,
                 std::function<void()> runtime_gate = nullptr)
        : path_(bam_path), planner_(hdr, targets, max_comp_bytes), n_threads_(std::max(n_threads, 1)), pinned_(pinned),
          stage_(std::move(stage)), pool_(pool), gate_(std::move(gate)), gate_registered_(std::move(runtime_gate)),
          numa_query_(std::move(numa_query)), device_(device) {
        for (int i = 0; i < 3; ++i) slots_[i].slot = slot_base + i;
        for (auto &it : slots_) free_.push_back(&it);
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

private:
    void release_buf(Item &it) {
        if (it.buf && it.registered) {
            inq_unpin_host(it.buf);
            it.registered = false;
        }
        if (it.buf) {
            if (pool_ && (it.pinned || it.node >= 0 || !numa_query_ || numa_query_() < 0)) pool_->give(HostBufPool::B{it.buf, it.cap, it.pinned, it.node});
            else if (it.pinned) inq_free_pinned(it.buf);
            else ::munmap(it.buf, it.cap);
        }
        it.buf = nullptr;
        it.cap = 0;
    }
    bool fit(Item &it, size_t bytes) {
        const int node = numa_query_ ? numa_query_() : -1;
        // a buffer mapped before the GPU's node was known is given up for one on that node (the slot has been uploaded by now)
        if (it.buf && !it.pinned && node >= 0 && it.node != node) release_buf(it);
        if (bytes <= it.cap && it.buf) return true;
        release_buf(it);
        HostBufPool::B got;
        if (pool_ && pool_->take(bytes, pinned_, &got)) {
            it.buf = got.p, it.cap = got.cap, it.pinned = got.pinned, it.node = got.node;
            if (!(node >= 0 && !it.pinned && it.node != node)) return true;
            release_buf(it);  // from before the node was known: not taken
        }
        const size_t want = bytes + bytes / 4 + (1u << 20);
        void *p = nullptr;
        if (pinned_ && inq_alloc_pinned(want, &p) == INQ_OK) it.pinned = true;
        else {
            // anonymous mapping with transparent huge pages where the kernel offers them: a span is hundreds of
            // MB written once by pread; 4 KB pages cost a fault each on the way in and a free on the way out
            const size_t huge = 2u << 20;
            const size_t len = (want + huge - 1) / huge * huge;
            p = ::mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (p == MAP_FAILED) p = nullptr;
            else {
                (void)::madvise(p, len, MADV_HUGEPAGE);
                prefer_numa_node(p, len, node);
                it.node = node >= 0 ? node : -1;
            }
            it.pinned = false;
            it.buf = (uint8_t *)p;
            it.cap = p ? len : 0;
            return p != nullptr;
        }
        it.buf = (uint8_t *)p;
        it.cap = p ? want : 0;
        return p != nullptr;
    }
    void fail(const std::string &m) {
        std::lock_guard<std::mutex> g(mu_);
        failed_ = true;
        err_ = m;
        cv_item_.notify_all();
        cv_loaded_.notify_all();
    }
    void run() {
        prefer_gpu_node_for_this_thread(device_);
        SpanLoader loader;
        std::string e;
        if (!loader.open(path_, &e)) return fail(e);
        // span k + 1 is planned (index searches for some 25 000 loci: ~4 ms) on a helper thread while span k is being read
        SpanPlan ahead;
        bool have = planner_.next(ahead);
        // INQ_GATE_READS=1: the first read waits for the device context.  Tried because the runtime's start-up looked longer
        // while the loader was reading (two boxes, 102 vs 221 ms); five runs each way on a third box showed the start-up
        // varying between 108 and 431 ms with and without early reads alike (profiles/r03_results/loader_gate_ab.txt): it
        // is hipInit itself that varies (57 - 224 ms), so reads start at once - the spans are there when the context is.
        if (have && gate_) gate_();
        while (have) {
            Item *it = nullptr;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_free_.wait(g, [&] { return !free_.empty() || stop_; });
                if (stop_) return;
                it = free_.back();
                free_.pop_back();
            }
            const auto t0 = std::chrono::steady_clock::now();
            std::swap(it->plan, ahead);
            std::future<bool> more = std::async(std::launch::async, [&] { return planner_.next(ahead); });  // joined by get() or by its destructor
            uint64_t nbytes = 0;
            const auto t1 = std::chrono::steady_clock::now();
            if (!loader.total_bytes(it->plan, &nbytes, &e)) return fail(e);
            if (nbytes > (64ull << 30)) return fail("a span of the BAM exceeds 64 GiB (index without usable bins)");
            if (!fit(*it, (size_t)nbytes + 64)) return fail("cannot allocate the span buffer");
            const auto t2 = std::chrono::steady_clock::now();
            if (!loader.load(it->plan, planner_.anchors(), it->buf, n_threads_, it->data, &e)) return fail(e);
            it->staged = false;
            if (verbose_) {
                const auto t3 = std::chrono::steady_clock::now();
                auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                std::fprintf(stderr, "[inq loader] @%.1f slot %d: plan %.2f ms, buffer %.2f ms (%s), read+tables %.2f ms for %.1f MB, %zu segments, %zu anchors\n",
                             stamp_ms(), it->slot, ms(t0, t1), ms(t1, t2), it->pinned ? "pinned" : "pageable", ms(t2, t3), nbytes / 1e6,
                             it->plan.segs.size(), it->data.anchors.size());
            }
            have = more.get();
            std::lock_guard<std::mutex> g(mu_);
            if (stage_) {
                loaded_.push_back(it);
                cv_loaded_.notify_one();
            } else {
                ready_.push_back(it);
                cv_item_.notify_one();
            }
        }
        std::lock_guard<std::mutex> g(mu_);
        if (stage_) {
            load_done_ = true;
            cv_loaded_.notify_all();
        } else {
            done_ = true;
            cv_item_.notify_all();
        }
    }
    // uploads in file order, one span behind the reader
    void run_uploads() {
        prefer_gpu_node_for_this_thread(device_);  // the runtime's staging chunks are allocated by the thread that first copies
        for (;;) {
            Item *it = nullptr;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_loaded_.wait(g, [&] { return !loaded_.empty() || load_done_ || stop_ || failed_; });
                if (stop_ || failed_) return;
                if (loaded_.empty()) {  // the reader is through
                    done_ = true;
                    cv_item_.notify_all();
                    return;
                }
                it = loaded_.front();
                loaded_.pop_front();
            }
            const auto t0 = std::chrono::steady_clock::now();
            inq_span_t sp;
            fill_span(*it, &sp);
            if (register_ && !it->registered && !it->pinned && it->buf) {
                // INQ_SPAN_REGISTER=1 (experiment): the buffer is page-locked where it lies before its first upload
                if (gate_registered_) gate_registered_();
                it->registered = inq_pin_host(it->buf, it->cap) == INQ_OK;
            }
            it->staged = stage_(sp, it->slot);
            if (verbose_)
                std::fprintf(stderr, "[inq loader] @%.1f slot %d: upload %.2f ms for %.1f MB%s\n", stamp_ms(), it->slot,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), it->data.comp_bytes / 1e6,
                             it->staged ? "" : " (not staged)");
            std::lock_guard<std::mutex> g(mu_);
            ready_.push_back(it);
            cv_item_.notify_one();
        }
    }

    std::string path_;
    SpanPlanner planner_;
    int n_threads_;
    bool pinned_;
    StageFn stage_;
    HostBufPool *pool_ = nullptr;
    std::function<void()> gate_;
    bool register_ = std::getenv("INQ_SPAN_REGISTER") && std::getenv("INQ_SPAN_REGISTER")[0] == '1';
    std::function<void()> gate_registered_;  // waits for the runtime before the first registration
    std::function<int()> numa_query_;  // the GPU's NUMA node, kNumaUnknown while the context is not there yet, -1 = do not place
    int device_ = 0;
    bool verbose_ = std::getenv("INQ_TIMING") && std::getenv("INQ_TIMING")[0] == '2';
    Item slots_[3];
    std::vector<Item *> free_;
    std::deque<Item *> ready_, loaded_;
    std::thread th_, up_;
    std::mutex mu_;
    std::condition_variable cv_item_, cv_free_, cv_loaded_;
    bool stop_ = false, done_ = false, failed_ = false, load_done_ = false;
    std::string err_;
};

static uint64_t span_bytes_from_env() {
    const char *e = std::getenv("INQ_SPAN_MB");
    const long v = e ? std::atol(e) : 0;
    // ~10 000 BGZF blocks per span: the workgroup-per-block inflate has no latency floor (1.1 ms per 1000 blocks), so small
    // spans cost nothing and the device starts on the first one while the loader still reads and uploads the next ones
    // (1 GB file: 256 MB spans 0.28 s median start to exit, one 1 GB span 0.37 s; profiles/r02_results/l2_span_size.txt)
    return v > 0 ? (uint64_t)v << 20 : (256ull << 20);
}

struct inq_spans {
    Prepared P;
    std::unique_ptr<SpanPipeline> pipe;
    SpanPipeline::Item *cur = nullptr;
    uint32_t minlen = 5, support = 3;
    bool unphased = false;
};

struct inq_frontend {
    Prepared P;
    std::unique_ptr<FrontEnd> fe;           // threads <= 1: one sweep on the caller's thread
    std::unique_ptr<ParallelFrontEnd> pfe;  // threads  > 1: the same worker pool the CLI driver uses
    ParallelFrontEnd::Item item;
    HostBatch batch;
    std::string bam_path;
    uint32_t minlen = 5, support = 3;
    uint64_t threads = 1, max_words = 0;
    bool unphased = false;
};

extern "C" {

static int inq_frontend_open_impl(const inq_call_args_t *args, inq_frontend_t **out, char *errbuf, size_t errcap) {
    if (!out) return INQ_EXIT_ERROR;
    *out = nullptr;
    std::unique_ptr<inq_frontend> F(new inq_frontend());
    std::string msg;
    int rc = prepare(args, F->P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    F->minlen = args->minlen;
    F->support = (uint32_t)std::min<uint64_t>(args->support, 0xffffffffull);
    F->unphased = args->unphased != 0;
    F->threads = args->threads;
    F->bam_path = args->bam;
    if (F->threads <= 1) F->fe.reset(new FrontEnd(*F->P.bam, F->P.targets, F->unphased));
    *out = F.release();
    return INQ_EXIT_OK;
}

uint64_t inq_frontend_n_targets(const inq_frontend_t *fe) { return fe ? fe->P.targets.size() : 0; }

int inq_frontend_target(const inq_frontend_t *fe, uint64_t i, const char **chrom, uint32_t *start, uint32_t *end) {
    if (!fe || i >= fe->P.targets.size()) return -1;
    if (chrom) *chrom = fe->P.targets[i].chrom.c_str();
    if (start) *start = fe->P.targets[i].start;
    if (end) *end = fe->P.targets[i].end;
    return 0;
}

const char *inq_frontend_sample(const inq_frontend_t *fe) { return fe ? fe->P.sample.c_str() : ""; }

void inq_frontend_set_batch_words(inq_frontend_t *fe, uint64_t w) {
    if (!fe) return;
    fe->max_words = w;
    if (fe->fe) fe->fe->set_max_batch_words(w);
}

static int inq_frontend_next_impl(inq_frontend_t *fe, inq_batch_t *batch, const uint32_t **locus_index, char *errbuf,
                      size_t errcap) {
    if (!fe || !batch) return -INQ_EXIT_ERROR;
    std::string err;
    bool panic = false;
    if (fe->threads > 1) {  // batches arrive in completion order; locus_index says where each row belongs
        if (!fe->pfe)
            fe->pfe.reset(new ParallelFrontEnd(fe->bam_path, *fe->P.bam, fe->P.targets, fe->unphased,
                                               (int)std::min<uint64_t>(fe->threads, 64), fe->max_words));
        fe->pfe->recycle(std::move(fe->item));
        fe->item = ParallelFrontEnd::Item();
        int rc = fe->pfe->next(fe->item, &err, &panic);
        if (rc < 0) {
            set_err(errbuf, errcap, err);
            return -INQ_EXIT_PANIC;
        }
        if (rc == 0) return 0;
        fe->item.batch.view(batch, fe->minlen, fe->support, fe->unphased);
        if (locus_index) *locus_index = fe->item.index.data();
        return 1;
    }
    int rc = fe->fe->next(fe->batch, &err, &panic);
    if (rc < 0) {
        set_err(errbuf, errcap, err);
        return -INQ_EXIT_PANIC;  // read errors are expect()/unwrap() panics too (:294,346)
    }
    if (rc == 0) return 0;
    fe->batch.view(batch, fe->minlen, fe->support, fe->unphased);
    if (locus_index) *locus_index = fe->batch.locus_index.data();
    return 1;
}

void inq_frontend_close(inq_frontend_t *fe) { delete fe; }

// The device context is created on its own thread from the first instruction of the command: HIP start-up
// (0.1 - 0.3 s) is the longest fixed cost of a run and overlaps opening the BAM, the BED, the .bai and the
// first reads of the file.
struct AsyncCtx {
    inq_ctx_t *ctx = nullptr;
    int hrc = INQ_OK;
    int numa_node = -1;
    std::atomic<bool> ready{false};  // ctx / hrc / numa_node are final
    std::thread th;
    bool leak = false;  // set after a clean run when the process is about to exit (INQ_FAST_EXIT)
    void start(int device) {
        th = std::thread([this, device] {
            prefer_gpu_node_for_this_thread(device);  // what the runtime allocates while it starts
            const double a = stamp_ms();
            hrc = inq_ctx_create(device, &ctx);
            numa_node = hrc == INQ_OK ? inq_ctx_numa_node(ctx) : -1;
            ready.store(true);
            const char *e = std::getenv("INQ_TIMING");
            if (e && e[0] == '2') std::fprintf(stderr, "[inq ctx] @%.1f device context ready (inq_ctx_create %.1f ms), GPU on NUMA node %d\n", stamp_ms(), stamp_ms() - a, numa_node);
        });
    }
    std::mutex mu;
    bool wait() {  // any thread
        std::lock_guard<std::mutex> g(mu);
        if (th.joinable()) th.join();
        return hrc == INQ_OK;
    }
    ~AsyncCtx() {
        if (th.joinable()) th.join();
        if (!leak) inq_ctx_destroy(ctx);
    }
};

// front end selection: args->reserved 1 = host sweep (BGZF inflate + record decode on CPU threads),
// 2 = device (inq_call_span); 0 = INQ_FRONTEND=host|device, else by the amount of BAM the loci need
static bool use_device_front(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets) {
    if (args->reserved == 1) return false;
    if (args->reserved == 2) return true;
    const char *e = std::getenv("INQ_FRONTEND");
    if (e && std::strcmp(e, "host") == 0) return false;
    if (e && std::strcmp(e, "device") == 0) return true;
    // auto: plan without reading anything and count the compressed bytes the loci need
    SpanPlanner planner(bam, targets, ~0ull >> 1);
    SpanPlan plan;
    uint64_t bytes = 0;
    while (planner.next(plan))
        for (const Segment &g : plan.segs) bytes += (g.vo_limit >> 16) - (g.vo_begin >> 16) + 32768;
    return bytes >= kDeviceFrontMinBytesPerThread * std::max<uint64_t>(1, std::min<uint64_t>(args->threads, 16));
}

// fills p1 / p2 through the device front end; returns an exit status
// what one call works on: the opened BAM (header + index), the targets it was asked for, the options
struct CallView {
    BamFile &bam;
    const std::vector<RepeatInterval> &targets;
    const std::string &sample;
    uint32_t minlen, support;
    bool unphased;
};

// what a session adds to one call: a pipeline that was started ahead of it (its loader has been reading and uploading while the
// previous file was being called), the buffer pool, the set of device staging slots
struct SessionHooks {
    SpanPipeline *early_pipe = nullptr;
    HostBufPool *pool = nullptr;
    int slot_base = 0;
    int front = 0;  // 0 = decide here, 1 = host sweep, 2 = device spans (decided when the pipeline was started)
};

static int span_io_threads(const inq_call_args_t *args) {
    // -t counts the reference's calling workers; here the host only copies file bytes, which a few pread
    // streams do best whatever -t says (bounded by the machine)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::min<uint64_t>(std::max<uint64_t>(args->threads, 8), std::min<uint64_t>(hw, 32));
}

static SpanPipeline *start_span_pipeline(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets,
                                         AsyncCtx &actx, int slot_base, HostBufPool *pool) {
    const char *pin_env = std::getenv("INQ_SPAN_PINNED");
    // the loader uploads every span it has read (waiting for the context the first time), so that the upload of span k+1
    // overlaps the inflate of span k
    return new SpanPipeline(args->bam, bam, targets, span_bytes_from_env(), span_io_threads(args), pin_env ? pin_env[0] == '1' : false,
                            [&actx](const inq_span_t &sp, int slot) { return actx.wait() && inq_span_stage(actx.ctx, &sp, slot) == INQ_OK; },
                            slot_base, pool, std::getenv("INQ_GATE_READS") ? std::function<void()>([&actx] { (void)actx.wait(); }) : std::function<void()>(),
                            [&actx, dev = args->device]() -> int {
                                if (const char *e = std::getenv("INQ_NUMA_NODE")) return std::atoi(e);
                                return actx.ready.load() ? actx.numa_node : guess_gpu_numa_node(dev);
                            },
                            args->device, [&actx] { (void)actx.wait(); });
}

static int run_device_front(const inq_call_args_t *args, const CallView &V, AsyncCtx &actx, std::vector<double> &p1,
                            std::vector<double> &p2, char *errbuf, size_t errcap, double *t_front, double *t_dev,
                            const SessionHooks &hooks = SessionHooks()) {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const int timing = std::getenv("INQ_TIMING") ? (std::getenv("INQ_TIMING")[0] == '2' ? 2 : 1) : 0;
    const auto t_begin = clk::now();
    bool &leak_all = actx.leak;
    inq_ctx_t *&ctx = actx.ctx;
    int &hrc = actx.hrc;
    std::vector<double> b1, b2;
    {
        // the CLI sets INQ_FAST_EXIT: it is about to leave the process, so the span buffers (unmapping a GB
        // of touched pages takes ~0.1 s) and the device context are left to the operating system
        const char *fast_env = std::getenv("INQ_FAST_EXIT");
        const bool fast_exit = fast_env && fast_env[0] == '1';
        struct PipeHolder {
            SpanPipeline *p;
            const bool &leak;
            bool owned;
            ~PipeHolder() {
                if (owned && !leak) delete p;
            }
        } holder{hooks.early_pipe ? hooks.early_pipe : start_span_pipeline(args, V.bam, V.targets, actx, hooks.slot_base, hooks.pool), leak_all,
                 hooks.early_pipe == nullptr};
        SpanPipeline &pipe = *holder.p;
        bool joined = false;
        // loci whose batches wait on the device (inq_call_span_deferred), in the order they were appended
        // 50 000 loci per launch of the locus kernels: 0.69 of the HBM peak in the CLI's trace (53 500 loci, 240 us), 0.73 at
        // 107 000 (452 us) against 0.83 for the same kernel in bench.py's steady loop - a launch here comes cold behind the
        // gather that has just written its CIGARs - and twice the device memory to tear down at exit for 100 000
        const size_t kFlushLoci = std::getenv("INQ_FLUSH_LOCI") ? (size_t)std::max(1l, std::atol(std::getenv("INQ_FLUSH_LOCI"))) : 50000;
        constexpr uint64_t kFlushWords = 1ull << 31;    // ... or 8 GB of gathered CIGARs
        std::vector<uint32_t> pending;
        uint64_t pending_words = 0;
        auto flush = [&]() -> int {
            if (pending.empty()) return INQ_EXIT_OK;
            const auto f0 = clk::now();
            b1.assign(pending.size(), NAN);
            b2.assign(pending.size(), NAN);
            inq_result_t res;
            std::memset(&res, 0, sizeof res);
            res.phase1 = b1.data();
            res.phase2 = b2.data();
            double ms_call = 0;
            int rc2 = inq_call_flush(ctx, &res, pending.size(), &ms_call);
            *t_dev += secs(f0, clk::now());
            if (timing == 2)
                std::fprintf(stderr, "[inq call] @%.1f %zu loci, %.1f MB of CIGARs: locus kernels %.3f ms | wall %.2f ms\n", stamp_ms(), pending.size(),
                             pending_words * 4 / 1e6, ms_call, secs(f0, clk::now()) * 1e3);
            if (rc2 != INQ_OK) {
                std::string m = std::string("device call failed: ") + inq_strerror(rc2);
                if (rc2 == INQ_ERR_HIP) m += std::string(" [") + inq_last_error(ctx) + "]";
                set_err(errbuf, errcap, m);
                return (rc2 == INQ_ERR_HIP || rc2 == INQ_ERR_NOMEM || rc2 == INQ_ERR_NO_DEVICE) ? INQ_EXIT_ERROR : INQ_EXIT_PANIC;
            }
            for (size_t j = 0; j < pending.size(); ++j) {
                p1[pending[j]] = b1[j];
                p2[pending[j]] = b2[j];
            }
            pending.clear();
            pending_words = 0;
            return INQ_EXIT_OK;
        };
        for (;;) {
            SpanPipeline::Item *it = nullptr;
            std::string ferr;
            auto ta = clk::now();
            int nb = pipe.next(it, &ferr);
            auto tb = clk::now();
            *t_front += secs(ta, tb);
            if (nb < 0) {
                set_err(errbuf, errcap, ferr);
                return INQ_EXIT_PANIC;  // read errors are expect()/unwrap() panics in the reference (:294,346)
            }
            if (nb == 0) break;
            if (!joined) {
                actx.wait();
                joined = true;
                if (hrc == INQ_OK) inq_call_discard(ctx);  // a session's context: nothing of a file that failed half-way stays behind
            }
            if (hrc != INQ_OK) {
                set_err(errbuf, errcap, std::string("cannot open HIP device: ") + inq_strerror(hrc));
                return INQ_EXIT_ERROR;
            }
            inq_span_t sp;
            SpanPipeline::fill_span(*it, &sp);
            sp.minlen = V.minlen;
            sp.support = V.support;
            sp.unphased = V.unphased ? 1u : 0u;
            // the span's batch is appended to the one on the device; the locus kernels run once enough loci wait (a span of
            // SEQ-bearing records holds a few hundred loci, a launch wants tens of thousands) or the file is through
            inq_span_stats_t stt;
            int rc2 = inq_call_span_deferred(ctx, &sp, it->staged ? it->slot : -1, &stt);
            *t_dev += secs(tb, clk::now());
            if (timing == 2)
                std::fprintf(stderr,
                             "[inq span] @%.1f waited %.2f ms | loci %llu comp %.1f MB -> %.1f MB, %llu records, %llu pairs | upload %.2f inflate %.2f scan %.2f "
                             "join %.2f ms | wall %.2f ms\n",
                             stamp_ms(), secs(ta, tb) * 1e3, (unsigned long long)sp.n_loci, sp.comp_bytes / 1e6, stt.inflated_bytes / 1e6,
                             (unsigned long long)stt.n_records, (unsigned long long)stt.n_pairs, stt.ms_upload, stt.ms_inflate,
                             stt.ms_scan, stt.ms_join, secs(tb, clk::now()) * 1e3);
            if (rc2 != INQ_OK) {
                std::string m = std::string("device call failed: ") + inq_strerror(rc2);
                if (rc2 == INQ_ERR_HIP) m += std::string(" [") + inq_last_error(ctx) + "]";
                if (rc2 == INQ_ERR_BAM || rc2 == INQ_ERR_AUX || rc2 == INQ_ERR_INFLATE)
                    m += " (status " + std::to_string(stt.front_status) + ", record " + std::to_string(stt.first_bad_record) +
                         " of the span at file offset " + std::to_string(it->data.file_begin) + ")";
                set_err(errbuf, errcap, m);
                return (rc2 == INQ_ERR_HIP || rc2 == INQ_ERR_NOMEM || rc2 == INQ_ERR_NO_DEVICE) ? INQ_EXIT_ERROR : INQ_EXIT_PANIC;
            }
            pending.insert(pending.end(), it->plan.locus_index.begin(), it->plan.locus_index.end());
            pending_words += stt.n_cigar_words;
            pipe.release(it);
            if (pending.size() >= kFlushLoci || pending_words >= kFlushWords) {
                int frc = flush();
                if (frc != INQ_EXIT_OK) return frc;
            }
            continue;
        }
        if (!joined) actx.wait();
        if (hrc == INQ_OK) {
            int frc = flush();
            if (frc != INQ_EXIT_OK) return frc;
        }
        leak_all = fast_exit;  // only after a clean run: error paths tear down normally
        if (const char *probe = std::getenv("INQ_EXIT_PROBE")) {
            // experiment: what does the process's exit pay for?  1 = unmap the span buffers here (the pipeline's destructor) and
            // time it, 2 = also destroy the device context (every hipFree) and time that; then the fast exit as usual
            const auto e0 = clk::now();
            leak_all = false;
            if (holder.owned) {
                delete holder.p;
                holder.owned = false;
            }
            const auto e1 = clk::now();
            std::fprintf(stderr, "[inq exit probe] span pipeline torn down (loader joined, host buffers unmapped): %.2f ms\n", secs(e0, e1) * 1e3);
            if (probe[0] == '2') {
                inq_ctx_destroy(ctx);
                ctx = nullptr;
                std::fprintf(stderr, "[inq exit probe] device context destroyed: %.2f ms\n", secs(e1, clk::now()) * 1e3);
            }
            leak_all = true;
        }
        if (timing) std::fprintf(stderr, "[inq timing] spans done at %.3fs after the start of the device path\n", secs(t_begin, clk::now()));
    }
    if (timing) std::fprintf(stderr, "[inq timing] loader joined at %.3fs\n", secs(t_begin, clk::now()));
    if (hrc != INQ_OK) {  // no GPU is an error even for an empty target list
        set_err(errbuf, errcap, std::string("cannot open HIP device: ") + inq_strerror(hrc));
        return INQ_EXIT_ERROR;
    }
    return INQ_EXIT_OK;
}

static int write_rows(uint64_t threads, const std::vector<RepeatInterval> &targets, const std::string &sample, const double *p1,
                      const double *p2, int out_fd, char *errbuf, size_t errcap);

// rows instead of text: the targets named by idx[] (positions in the parsed target list) are called, their rows go to p1 / p2
struct RowsOut {
    const uint32_t *idx = nullptr;
    uint64_t n = 0;
    double *p1 = nullptr, *p2 = nullptr;
    bool active = false;
};

// the call on an opened BAM + parsed targets, on a device context that may outlive it (a session calls many BAMs on one)
static int genotype_prepared(const inq_call_args_t *args, AsyncCtx &actx, Prepared &P, int out_fd, char *errbuf, size_t errcap,
                             const RowsOut &rows, std::chrono::steady_clock::time_point t_start, const SessionHooks &hooks = SessionHooks()) {
    using clk = std::chrono::steady_clock;
    const bool timing = std::getenv("INQ_TIMING") != nullptr;
    double t_front = 0, t_dev = 0;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    std::vector<RepeatInterval> sub;
    if (rows.active) {  // this caller's share of the targets (one process per GPU: inquistr_amd/call_dist.py)
        sub.reserve(rows.n);
        for (uint64_t k = 0; k < rows.n; ++k) {
            if (rows.idx[k] >= P.targets.size()) {
                set_err(errbuf, errcap, "target index outside the target list");
                return INQ_EXIT_ERROR;
            }
            sub.push_back(P.targets[rows.idx[k]]);
        }
    }
    const CallView V{*P.bam, rows.active ? sub : P.targets, P.sample, args->minlen,
                     (uint32_t)std::min<uint64_t>(args->support, 0xffffffffull), args->unphased != 0};
    const size_t n = V.targets.size();
    std::vector<double> p1(n, NAN), p2(n, NAN);
    auto emit = [&]() -> int {
        if (!rows.active) return write_rows(args->threads, V.targets, V.sample, p1.data(), p2.data(), out_fd, errbuf, errcap);
        if (n) std::memcpy(rows.p1, p1.data(), n * sizeof(double)), std::memcpy(rows.p2, p2.data(), n * sizeof(double));
        return INQ_EXIT_OK;
    };

    const auto t_open = clk::now();
    const bool device_front = hooks.front ? hooks.front == 2 : use_device_front(args, V.bam, V.targets);
    if (device_front) {
        const auto t_choice = clk::now();
        int drc = run_device_front(args, V, actx, p1, p2, errbuf, errcap, &t_front, &t_dev, hooks);
        if (drc != INQ_EXIT_OK) return drc;
        const auto t_run = clk::now();
        drc = emit();
        if (timing)
            std::fprintf(stderr,
                         "[inq timing] device front end: open+targets %.3fs  front-end choice %.3fs  spans %.3fs (waiting for the loader "
                         "%.3fs, device calls %.3fs)  output %.3fs  total %.3fs\n",
                         secs(t_start, t_open), secs(t_open, t_choice), secs(t_choice, t_run), t_front, t_dev, secs(t_run, clk::now()),
                         secs(t_start, clk::now()));
        return drc;
    }

    auto t_prep = clk::now();
    inq_ctx_t *&ctx = actx.ctx;
    int &hrc = actx.hrc;
    bool ctx_ready = false;
    auto need_ctx = [&]() -> bool {
        if (!ctx_ready) {
            actx.wait();
            ctx_ready = true;
        }
        if (hrc != INQ_OK) {
            set_err(errbuf, errcap, std::string("cannot open HIP device: ") + inq_strerror(hrc));
            return false;
        }
        return true;
    };

    // -t N: N-1 sweep workers + this thread (which feeds the GPU and spins in the HIP runtime while waiting)
    // pinned staging: the device copies come from page-locked memory this thread fills, never from the
    // workers' pageable vectors (on-the-fly pinning contends with their page faults: 25 ms stalls)
    struct Pinned {
        void *p = nullptr;
        size_t cap = 0;
        ~Pinned() { inq_free_pinned(p); }
        void *fit(size_t bytes) {
            if (bytes > cap) {
                inq_free_pinned(p);
                p = nullptr;
                cap = bytes + bytes / 2 + (1u << 20);
                if (inq_alloc_pinned(cap, &p) != INQ_OK) p = nullptr, cap = 0;
            }
            return p;
        }
    } pin;
    const int n_workers = (int)std::max<uint64_t>(1, std::min<uint64_t>(args->threads > 1 ? args->threads - 1 : 1, 64));
    ParallelFrontEnd pfe(args->bam, V.bam, V.targets, V.unphased, n_workers);
    auto t_ctx = clk::now();
    std::vector<double> b1, b2;
    for (;;) {
        ParallelFrontEnd::Item item;
        std::string ferr;
        bool fpanic = false;
        auto ta = clk::now();
        int nb = pfe.next(item, &ferr, &fpanic);
        auto tb = clk::now();
        t_front += secs(ta, tb);
        if (nb < 0) {
            set_err(errbuf, errcap, ferr);
            return INQ_EXIT_PANIC;  // read errors are expect()/unwrap() panics in the reference (:294,346)
        }
        if (nb == 0) break;
        if (!need_ctx()) return INQ_EXIT_ERROR;
        inq_batch_t batch;
        item.batch.view(&batch, V.minlen, V.support, V.unphased);
        {
            auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
            const size_t s0 = al(batch.n_cigar_words * 4), s1 = al(batch.n_reads * sizeof(inq_read_t)),
                         s2 = al(batch.n_pairs * 4), s3 = al((batch.n_loci + 1) * 8), s4 = al(batch.n_loci * 4);
            char *base = (char *)pin.fit(s0 + s1 + s2 + s3 + 2 * s4);
            if (!base) {
                set_err(errbuf, errcap, "cannot allocate pinned host memory");
                return INQ_EXIT_ERROR;
            }
            auto put = [&](const void *src, size_t bytes, size_t &off, size_t slot) {
                void *dst = base + off;
                if (bytes) std::memcpy(dst, src, bytes);
                off += slot;
                return dst;
            };
            size_t off = 0;
            batch.cigar = (const uint32_t *)put(batch.cigar, batch.n_cigar_words * 4, off, s0);
            batch.reads = (const inq_read_t *)put(batch.reads, batch.n_reads * sizeof(inq_read_t), off, s1);
            batch.pair_read = (const uint32_t *)put(batch.pair_read, batch.n_pairs * 4, off, s2);
            batch.locus_pair_off = (const uint64_t *)put(batch.locus_pair_off, (batch.n_loci + 1) * 8, off, s3);
            batch.locus_start = (const uint32_t *)put(batch.locus_start, batch.n_loci * 4, off, s4);
            batch.locus_end = (const uint32_t *)put(batch.locus_end, batch.n_loci * 4, off, s4);
        }
        b1.assign(batch.n_loci, NAN);
        b2.assign(batch.n_loci, NAN);
        inq_result_t res;
        std::memset(&res, 0, sizeof res);
        res.phase1 = b1.data();
        res.phase2 = b2.data();
        auto tc = clk::now();
        int rc2 = inq_call_batch(ctx, &batch, &res);
        t_dev += secs(tb, clk::now());
        if (timing && std::getenv("INQ_TIMING")[0] == '2')
            std::fprintf(stderr, "[inq batch] loci %llu pairs %llu cigar %.1f MB  wait-ctx %.2f ms  call %.2f ms\n",
                         (unsigned long long)batch.n_loci, (unsigned long long)batch.n_pairs, batch.n_cigar_words * 4 / 1e6,
                         secs(tb, tc) * 1e3, secs(tc, clk::now()) * 1e3);
        if (rc2 != INQ_OK) {
            std::string m = std::string("device call failed: ") + inq_strerror(rc2);
            if (rc2 == INQ_ERR_HIP) m += std::string(" [") + inq_last_error(ctx) + "]";
            set_err(errbuf, errcap, m);
            // domain errors are the reference's panics (HP > 2, bad CIGAR op, ...)
            return (rc2 == INQ_ERR_HIP || rc2 == INQ_ERR_NOMEM || rc2 == INQ_ERR_NO_DEVICE) ? INQ_EXIT_ERROR : INQ_EXIT_PANIC;
        }
        for (uint64_t j = 0; j < batch.n_loci; ++j) {
            p1[item.index[j]] = b1[j];
            p2[item.index[j]] = b2[j];
        }
        pfe.recycle(std::move(item));
    }
    if (!need_ctx()) return INQ_EXIT_ERROR;  // no GPU is an error even for an empty target list

    {
        int wrc = emit();
        if (wrc != INQ_EXIT_OK) return wrc;
    }
    {  // the CLI is about to leave the process: the device context is left to the operating system (see run_device_front)
        const char *fast_env = std::getenv("INQ_FAST_EXIT");
        actx.leak = fast_env && fast_env[0] == '1';
    }
    if (timing)
        std::fprintf(stderr, "[inq timing] open+targets %.3fs  hip ctx %.3fs  front end %.3fs  device calls %.3fs  total %.3fs\n",
                     secs(t_start, t_prep), secs(t_prep, t_ctx), t_front, t_dev, secs(t_start, clk::now()));
    return INQ_EXIT_OK;
}

// the output stage, src/call.rs:137-157
static int write_rows(uint64_t threads, const std::vector<RepeatInterval> &targets, const std::string &sample, const double *p1,
                      const double *p2, int out_fd, char *errbuf, size_t errcap) {
    const size_t n = targets.size();
    const bool timing = std::getenv("INQ_TIMING") != nullptr;
    const auto t_w0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    if (threads > 1) {
        // genotypes_vec.sort_unstable() with Ord = (human_compare(chrom), start), :33-38,141.  Equal keys
        // are in completion order in the reference (nondeterministic); BED order is kept here.
        // The contig names are ranked once (a BED has few distinct ones), the rows sorted on integers.
        std::map<std::string, uint32_t> rank;
        for (size_t i = 0; i < n; ++i)
            if (i == 0 || targets[i].chrom != targets[i - 1].chrom) rank.emplace(targets[i].chrom, 0u);
        std::vector<const std::string *> names;
        for (auto &kv : rank) names.push_back(&kv.first);
        std::stable_sort(names.begin(), names.end(), [](const std::string *a, const std::string *b) { return human_compare(*a, *b) < 0; });
        for (size_t i = 0, r = 0; i < names.size(); ++i) {
            if (i && human_compare(*names[i - 1], *names[i]) != 0) ++r;  // names that compare equal share a rank
            rank[*names[i]] = (uint32_t)r;
        }
        std::vector<uint64_t> key(n);
        const std::string *last = nullptr;  // neighbouring targets mostly share the contig: one map lookup per run of them
        uint32_t last_rank = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!last || *last != targets[i].chrom) last = &targets[i].chrom, last_rank = rank[*last];
            key[i] = ((uint64_t)last_rank << 32) | targets[i].start;
        }
        if (!std::is_sorted(key.begin(), key.end())) {
            // stable LSD radix sort of the row numbers on the 64-bit key, 16 bits a pass; digits all keys share are skipped
            // (a BED has few contigs and starts below 2^28: two or three passes instead of n log n compares)
            uint64_t all_or = 0, all_and = ~0ull;
            for (uint64_t k : key) all_or |= k, all_and &= k;
            std::vector<uint32_t> tmp(n);
            std::vector<uint32_t> cnt(65536);
            for (int shift = 0; shift < 64; shift += 16) {
                if ((((all_or ^ all_and) >> shift) & 0xffffu) == 0) continue;
                std::fill(cnt.begin(), cnt.end(), 0u);
                for (size_t i = 0; i < n; ++i) ++cnt[(key[order[i]] >> shift) & 0xffffu];
                uint32_t run = 0;
                for (auto &c : cnt) {
                    const uint32_t v = c;
                    c = run;
                    run += v;
                }
                for (size_t i = 0; i < n; ++i) tmp[cnt[(key[order[i]] >> shift) & 0xffffu]++] = order[i];
                order.swap(tmp);
            }
        }
    }
    const auto t_w1 = std::chrono::steady_clock::now();
    // the text: rows formatted by a few threads into their own stretches of one buffer (sized from an upper bound per row,
    // written through a bare pointer: no per-character capacity checks), written in order.  Four threads at most: 500 000
    // rows are 17 MB of text, ~20 ms on one core, and starting a thread costs up to 2 ms on virtualised hosts.
    const size_t n_parts = n < 65536 ? 1 : std::min<size_t>(4, std::max(1u, std::thread::hardware_concurrency()));
    const std::string header = format_header(sample) + "\n";
    std::vector<size_t> part_off(n_parts + 1, 0), part_len(n_parts, 0);
    auto value_bound = [](double v) -> size_t { return std::fabs(v) < 9007199254740992.0 || std::isnan(v) ? 20 : 330; };
    for (size_t k = 0; k < n_parts; ++k) {
        const size_t lo = n * k / n_parts, hi = n * (k + 1) / n_parts;
        size_t cap = (k == 0 ? header.size() : 0) + (hi - lo) * (2 * 10 + 5) + 128;
        for (size_t j = lo; j < hi; ++j) {
            const uint32_t i = order[j];
            cap += targets[i].chrom.size() + value_bound(p1[i]) + value_bound(p2[i]);
        }
        part_off[k + 1] = part_off[k] + ((cap + 63) & ~(size_t)63);
    }
    const auto t_w1b = std::chrono::steady_clock::now();
    const size_t huge = 2u << 20, map_len = (part_off[n_parts] + huge - 1) / huge * huge;
    char *const base = (char *)::mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == (char *)MAP_FAILED) {
        set_err(errbuf, errcap, "cannot allocate the output buffer");
        return INQ_EXIT_ERROR;
    }
    auto format_part = [&](size_t k) {
        const size_t lo = n * k / n_parts, hi = n * (k + 1) / n_parts;
        char *p = base + part_off[k];
        if (k == 0) std::memcpy(p, header.data(), header.size()), p += header.size();
        for (size_t j = lo; j < hi; ++j) {
            const uint32_t i = order[j];
            const RepeatInterval &t = targets[i];
            p = write_row(p, t.chrom, t.start, t.end, p1[i], p2[i]);
            *p++ = '\n';
        }
        part_len[k] = (size_t)(p - (base + part_off[k]));
    };
    {
        std::vector<std::thread> th;
        for (size_t k = 1; k < n_parts; ++k) th.emplace_back(format_part, k);
        format_part(0);
        for (auto &x : th) x.join();
    }
    const auto t_w2 = std::chrono::steady_clock::now();
    bool wrote = true;
    for (size_t k = 0; k < n_parts && wrote; ++k) wrote = write_all(out_fd, base + part_off[k], part_len[k]);
    ::munmap(base, map_len);
    if (!wrote) {
        set_err(errbuf, errcap, "Failed writing the result.");
        return INQ_EXIT_PANIC;
    }
    if (timing) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[inq output] %zu rows: order %.2f ms, bounds %.2f ms, text %.2f ms (%zu threads), write %.2f ms\n", n, ms(t_w0, t_w1), ms(t_w1, t_w1b), ms(t_w1b, t_w2),
                     n_parts, ms(t_w2, std::chrono::steady_clock::now()));
    }
    return INQ_EXIT_OK;
}

static int inq_genotype_repeats_impl(const inq_call_args_t *args, int out_fd, char *errbuf, size_t errcap, const RowsOut &rows = RowsOut()) {
    const auto t_start = std::chrono::steady_clock::now();
    AsyncCtx actx;
    if (args) actx.start(args->device);
    Prepared P;
    std::string msg;
    int rc = prepare(args, P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    return genotype_prepared(args, actx, P, out_fd, errbuf, errcap, rows, t_start);
}

// ---- combine, src/combine.rs ----
namespace {
struct LineReader {
    gzFile gz = nullptr;
    FILE *fp = nullptr;
    bool open(const std::string &path, std::string *err) {
        const bool is_gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;  // Path::extension() == "gz"
        if (is_gz) {
            gz = gzopen(path.c_str(), "rb");
            if (gz) gzbuffer(gz, 128 * 1024);
        } else {
            fp = std::fopen(path.c_str(), "rb");
        }
        if (!gz && !fp) {
            *err = "couldn't open " + path;
            return false;
        }
        return true;
    }
    // BufRead::lines(): splits on '\n', strips a trailing "\r\n" or "\n"; false at end of input
    bool next(std::string &line) {
        line.clear();
        char buf[1 << 16];
        bool got = false;
        for (;;) {
            char *r = gz ? gzgets(gz, buf, sizeof buf) : std::fgets(buf, sizeof buf, fp);
            if (!r) break;
            got = true;
            size_t n = std::strlen(buf);
            line.append(buf, n);
            if (n && buf[n - 1] == '\n') break;
        }
        if (!got) return false;
        if (!line.empty() && line.back() == '\n') line.pop_back();
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
    ~LineReader() {
        if (gz) gzclose(gz);
        if (fp) std::fclose(fp);
    }
};
}  // namespace

static int inq_combine_impl(const char *const *files, size_t n_files, int out_fd, char *errbuf, size_t errcap) {
    if (!files || n_files == 0) {
        set_err(errbuf, errcap, "no input files");
        return INQ_EXIT_ERROR;
    }
    for (size_t i = 0; i < n_files; ++i) {  // src/combine.rs:29-33
        struct stat st;
        if (::stat(files[i], &st) != 0) {
            set_err(errbuf, errcap, std::string("File ") + files[i] + " does not exist!");
            return INQ_EXIT_PANIC;
        }
    }
    std::vector<std::unique_ptr<LineReader>> rd;
    for (size_t i = 0; i < n_files; ++i) {
        rd.emplace_back(new LineReader());
        std::string e;
        if (!rd.back()->open(files[i], &e)) {
            set_err(errbuf, errcap, e);
            return INQ_EXIT_PANIC;
        }
    }
    std::string line, other, out;
    while (rd[0]->next(line)) {  // :42
        out += line;
        for (size_t i = 1; i < n_files; ++i) {
            if (!rd[i]->next(other)) {  // :49 file2.next().unwrap() on None
                set_err(errbuf, errcap, std::string("called `Option::unwrap()` on a `None` value (") + files[i] +
                                            " has fewer lines than " + files[0] + ")");
                return INQ_EXIT_PANIC;
            }
            // :52-55 split('\t').skip(3): every field after the third
            size_t p = 0;
            int tabs = 0;
            while (tabs < 3) {
                size_t t = other.find('\t', p);
                if (t == std::string::npos) {
                    p = std::string::npos;
                    break;
                }
                p = t + 1;
                ++tabs;
            }
            if (p != std::string::npos) {
                out += '\t';
                out.append(other, p, std::string::npos);
            }
        }
        out += '\n';
        if (out.size() > (1u << 20)) {
            if (!write_all(out_fd, out)) return INQ_EXIT_PANIC;
            out.clear();
        }
    }
    if (!write_all(out_fd, out)) return INQ_EXIT_PANIC;
    return INQ_EXIT_OK;
}

// public entries: no C++ exception may unwind across the C ABI
#define INQ_GUARD(expr, errbuf, errcap)                                   \
    try {                                                                  \
        return expr;                                                       \
    } catch (const std::exception &e) {                                    \
        set_err(errbuf, errcap, std::string("internal error: ") + e.what()); \
        return INQ_EXIT_ERROR;                                             \
    } catch (...) {                                                        \
        set_err(errbuf, errcap, "internal error");                         \
        return INQ_EXIT_ERROR;                                             \
    }
int inq_frontend_open(const inq_call_args_t *args, inq_frontend_t **out, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_frontend_open_impl(args, out, errbuf, errcap), errbuf, errcap)
}
int inq_frontend_next(inq_frontend_t *fe, inq_batch_t *batch, const uint32_t **locus_index, char *errbuf, size_t errcap) {
    try {
        return inq_frontend_next_impl(fe, batch, locus_index, errbuf, errcap);
    } catch (...) {
        set_err(errbuf, errcap, "internal error");
        return -INQ_EXIT_ERROR;
    }
}
int inq_genotype_repeats(const inq_call_args_t *args, int out_fd, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_genotype_repeats_impl(args, out_fd, errbuf, errcap), errbuf, errcap)
}
int inq_combine(const char *const *files, size_t n_files, int out_fd, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_combine_impl(files, n_files, out_fd, errbuf, errcap), errbuf, errcap)
}

// ---- one process per GPU: this process's share of the targets, rows as numbers ----
static int inq_genotype_repeats_rows_impl(const inq_call_args_t *args, const uint32_t *target_index, uint64_t n_index, double *phase1,
                                          double *phase2, char *errbuf, size_t errcap) {
    if (n_index && (!target_index || !phase1 || !phase2)) {
        set_err(errbuf, errcap, "null argument");
        return INQ_EXIT_ERROR;
    }
    RowsOut r;
    r.idx = target_index;
    r.n = n_index;
    r.p1 = phase1;
    r.p2 = phase2;
    r.active = true;
    return inq_genotype_repeats_impl(args, -1, errbuf, errcap, r);
}
int inq_genotype_repeats_rows(const inq_call_args_t *args, const uint32_t *target_index, uint64_t n_index, double *phase1, double *phase2,
                              char *errbuf, size_t errcap) {
    INQ_GUARD(inq_genotype_repeats_rows_impl(args, target_index, n_index, phase1, phase2, errbuf, errcap), errbuf, errcap)
}

// The targets in file order (contig of the BAM header, start, end, position in the list) and `world` + 1 cut points into
// that order, so that every part needs about the same number of compressed BAM bytes: cost of a target = bytes between its
// scan start in the .bai's linear index and the next target's, capped so that one far-away locus does not own a contig.
static int partition_prepared(Prepared &P, uint64_t world, uint32_t *order, uint64_t *cuts) {
    const size_t n = P.targets.size();
    std::vector<int> tid(n);
    {
        std::map<std::string, int> memo;
        for (size_t i = 0; i < n; ++i) {
            auto it = memo.find(P.targets[i].chrom);
            if (it == memo.end()) it = memo.emplace(P.targets[i].chrom, P.bam->tid(P.targets[i].chrom)).first;
            tid[i] = it->second;
        }
    }
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order, order + n, [&](uint32_t a, uint32_t b) {
        if (tid[a] != tid[b]) return tid[a] < tid[b];
        if (P.targets[a].start != P.targets[b].start) return P.targets[a].start < P.targets[b].start;
        return P.targets[a].end < P.targets[b].end;
    });
    std::vector<double> off(n), cost(n, 1.0);
    for (size_t k = 0; k < n; ++k) {
        const RepeatInterval &t = P.targets[order[k]];
        off[k] = (double)(P.bam->index().scan_start(tid[order[k]], t.start >= 10 ? (int64_t)t.start - 10 : 0) >> 16);
    }
    std::vector<double> d;
    for (size_t k = 0; k + 1 < n; ++k)
        if (tid[order[k]] == tid[order[k + 1]] && off[k + 1] > off[k]) d.push_back(off[k + 1] - off[k]);
    if (!d.empty()) {
        std::vector<double> ds = d;
        std::sort(ds.begin(), ds.end());
        const double cap = ds[std::min(ds.size() - 1, (size_t)(0.99 * (double)ds.size()))] * 4 + 1, med = ds[ds.size() / 2];
        for (size_t k = 0; k < n; ++k) {
            const bool same = k + 1 < n && tid[order[k]] == tid[order[k + 1]] && off[k + 1] > off[k];
            cost[k] += same ? std::min(off[k + 1] - off[k], cap) : med;  // last target of a contig: a typical gap
        }
    }
    std::vector<double> csum(n + 1, 0.0);
    for (size_t k = 0; k < n; ++k) csum[k + 1] = csum[k] + cost[k];
    cuts[0] = 0;
    for (uint64_t r = 1; r < world; ++r) {
        const double want = csum[n] * (double)r / (double)world;
        uint64_t k = (uint64_t)(std::lower_bound(csum.begin(), csum.end(), want) - csum.begin());
        cuts[r] = std::min<uint64_t>(std::max<uint64_t>(k, cuts[r - 1]), n);
    }
    cuts[world] = n;
    return INQ_EXIT_OK;
}
static int inq_host_partition_impl(const inq_call_args_t *args, uint64_t world, uint32_t *order, uint64_t order_cap, uint64_t *cuts,
                                   uint64_t *n_targets, char *errbuf, size_t errcap) {
    if (!world || !cuts || !n_targets) {
        set_err(errbuf, errcap, "null argument");
        return INQ_EXIT_ERROR;
    }
    Prepared P;
    std::string msg;
    int rc = prepare(args, P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    const size_t n = P.targets.size();
    *n_targets = n;
    if (n > order_cap || (n && !order)) {
        set_err(errbuf, errcap, "order[] too small for the target list");
        return INQ_EXIT_ERROR;
    }
    return partition_prepared(P, world, order, cuts);
}
int inq_host_partition(const inq_call_args_t *args, uint64_t world, uint32_t *order, uint64_t order_cap, uint64_t *cuts, uint64_t *n_targets,
                       char *errbuf, size_t errcap) {
    INQ_GUARD(inq_host_partition_impl(args, world, order, order_cap, cuts, n_targets, errbuf, errcap), errbuf, errcap)
}

// ---- a prepared run: BAM header + index + targets opened once and used for the split, this process's rows and the output ----
struct OwnedArgs {
    inq_call_args_t a;
    std::string bam, region, region_file, sample_name, reference;
    explicit OwnedArgs(const inq_call_args_t &src) : a(src) {
        auto own = [](const char *&p, std::string &keep) {
            if (p) keep = p, p = keep.c_str();
        };
        own(a.bam, bam), own(a.region, region), own(a.region_file, region_file), own(a.sample_name, sample_name), own(a.reference, reference);
    }
    OwnedArgs(const OwnedArgs &) = delete;
};

struct inq_run {
    std::unique_ptr<OwnedArgs> args;
    Prepared P;
};

static int partition_prepared(Prepared &P, uint64_t world, uint32_t *order, uint64_t *cuts);

static int inq_run_open_impl(const inq_call_args_t *args, inq_run_t **out, char *errbuf, size_t errcap) {
    if (!out || !args) return INQ_EXIT_ERROR;
    *out = nullptr;
    std::unique_ptr<inq_run> R(new inq_run());
    R->args.reset(new OwnedArgs(*args));
    std::string msg;
    int rc = prepare(&R->args->a, R->P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    *out = R.release();
    return INQ_EXIT_OK;
}
int inq_run_open(const inq_call_args_t *args, inq_run_t **out, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_run_open_impl(args, out, errbuf, errcap), errbuf, errcap)
}
uint64_t inq_run_n_targets(const inq_run_t *r) { return r ? r->P.targets.size() : 0; }
const char *inq_run_sample(const inq_run_t *r) { return r ? r->P.sample.c_str() : ""; }
int inq_run_target(const inq_run_t *r, uint64_t i, const char **chrom, uint32_t *start, uint32_t *end) {
    if (!r || i >= r->P.targets.size()) return -1;
    if (chrom) *chrom = r->P.targets[i].chrom.c_str();
    if (start) *start = r->P.targets[i].start;
    if (end) *end = r->P.targets[i].end;
    return 0;
}
int inq_run_partition(inq_run_t *r, uint64_t world, uint32_t *order, uint64_t *cuts, char *errbuf, size_t errcap) {
    if (!r || !world || !cuts || (!order && !r->P.targets.empty())) {
        set_err(errbuf, errcap, "null argument");
        return INQ_EXIT_ERROR;
    }
    INQ_GUARD(partition_prepared(r->P, world, order, cuts), errbuf, errcap)
}
static int inq_run_rows_impl(inq_run_t *r, const uint32_t *target_index, uint64_t n_index, double *phase1, double *phase2, char *errbuf,
                             size_t errcap) {
    if (!r || (n_index && (!target_index || !phase1 || !phase2))) {
        set_err(errbuf, errcap, "null argument");
        return INQ_EXIT_ERROR;
    }
    const auto t_start = std::chrono::steady_clock::now();
    AsyncCtx actx;
    actx.start(r->args->a.device);
    RowsOut ro;
    ro.idx = target_index, ro.n = n_index, ro.p1 = phase1, ro.p2 = phase2, ro.active = true;
    return genotype_prepared(&r->args->a, actx, r->P, -1, errbuf, errcap, ro, t_start);
}
int inq_run_rows(inq_run_t *r, const uint32_t *target_index, uint64_t n_index, double *phase1, double *phase2, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_run_rows_impl(r, target_index, n_index, phase1, phase2, errbuf, errcap), errbuf, errcap)
}
int inq_run_write_inq(inq_run_t *r, const double *phase1, const double *phase2, uint64_t n_rows, int out_fd, char *errbuf, size_t errcap) {
    if (!r || n_rows != r->P.targets.size() || (n_rows && (!phase1 || !phase2))) {
        set_err(errbuf, errcap, "row count does not match the target list");
        return INQ_EXIT_ERROR;
    }
    INQ_GUARD(write_rows(r->args->a.threads, r->P.targets, r->P.sample, phase1, phase2, out_fd, errbuf, errcap), errbuf, errcap)
}
void inq_run_close(inq_run_t *r) { delete r; }

// ---- a session: many BAMs on ONE device context (a cohort is called sample by sample with the same BED, then combined:
// src/combine.rs).  The HIP runtime's start-up (0.1 - 0.3 s, the whole cost of a 1 GB file) is paid once; span buffers are
// reused; and while file k is being called, file k + 1 is opened, its targets parsed, its spans planned, read and uploaded
// into the other set of device staging slots.  Each file's output is byte for byte that of its own `inquistr call`.
struct inq_session {
    AsyncCtx actx;
    HostBufPool pool;
    BedCache bed_cache;
    uint64_t n_staged = 0;  // inq_session_stage: which of the two sets of device slots the next file takes
};

namespace {
struct StagedFile {
    std::unique_ptr<OwnedArgs> args;
    Prepared P;
    int rc = INQ_EXIT_OK;
    std::string msg;
    std::unique_ptr<SpanPipeline> pipe;
    int slot_base = 0, front = 0;
    std::chrono::steady_clock::time_point t_start;
};

void stage_file(inq_session *S, const inq_call_args_t *a, int slot_base, StagedFile &out) {
    out.t_start = std::chrono::steady_clock::now();
    try {
        out.args.reset(new OwnedArgs(*a));
        out.slot_base = slot_base;
        out.rc = prepare(&out.args->a, out.P, out.msg, &S->bed_cache);
        if (out.rc != INQ_EXIT_OK) return;
        out.front = use_device_front(&out.args->a, *out.P.bam, out.P.targets) ? 2 : 1;
        if (out.front == 2) out.pipe.reset(start_span_pipeline(&out.args->a, *out.P.bam, out.P.targets, S->actx, slot_base, &S->pool));
    } catch (const std::exception &e) {
        out.rc = INQ_EXIT_ERROR;
        out.msg = std::string("internal error: ") + e.what();
    }
}

int run_staged(inq_session *S, StagedFile &f, int out_fd, char *errbuf, size_t errcap) {
    if (f.rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, f.msg);
        return f.rc;
    }
    SessionHooks hooks;
    hooks.early_pipe = f.pipe.get();
    hooks.pool = &S->pool;
    hooks.slot_base = f.slot_base;
    hooks.front = f.front;
    const bool keep_leak = S->actx.leak;
    int rc = genotype_prepared(&f.args->a, S->actx, f.P, out_fd, errbuf, errcap, RowsOut(), f.t_start, hooks);
    S->actx.leak = keep_leak;  // the context belongs to the session, whatever the single-call path decided
    f.pipe.reset();            // joins the loader, hands the span buffers back to the pool
    return rc;
}
}  // namespace

int inq_session_open(int32_t device, inq_session_t **out) {
    if (!out) return INQ_EXIT_ERROR;
    *out = nullptr;
    try {
        inq_session *S = new inq_session();
        S->actx.start(device);  // returns at once: the runtime starts on its own thread
        *out = S;
        return INQ_EXIT_OK;
    } catch (...) {
        return INQ_EXIT_ERROR;
    }
}

static int inq_session_call_impl(inq_session_t *S, const inq_call_args_t *args, int out_fd, char *errbuf, size_t errcap) {
    if (!S || !args) return INQ_EXIT_ERROR;
    StagedFile f;
    stage_file(S, args, 0, f);
    return run_staged(S, f, out_fd, errbuf, errcap);
}
int inq_session_call(inq_session_t *S, const inq_call_args_t *args, int out_fd, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_session_call_impl(S, args, out_fd, errbuf, errcap), errbuf, errcap)
}

static int inq_session_call_many_impl(inq_session_t *S, const inq_call_args_t *args, size_t n, const int *out_fds, int *statuses, char *errbuf,
                                      size_t errcap) {
    if (!S || (n && (!args || !out_fds))) return INQ_EXIT_ERROR;
    const bool timing = std::getenv("INQ_TIMING") != nullptr;
    std::vector<StagedFile> st(n);
    int worst = INQ_EXIT_OK;
    bool have_msg = false;
    if (n) stage_file(S, &args[0], 0, st[0]);
    for (size_t k = 0; k < n; ++k) {
        // file k + 1 is staged (opened, planned, read, uploaded into the other set of device slots) while file k is called
        std::future<void> next;
        if (k + 1 < n) next = std::async(std::launch::async, [&, k] { stage_file(S, &args[k + 1], 3 * (int)((k + 1) & 1), st[k + 1]); });
        char msg[1024] = {0};
        const auto t0 = std::chrono::steady_clock::now();
        int rc;
        try {
            rc = run_staged(S, st[k], out_fds[k], msg, sizeof msg);
        } catch (const std::exception &e) {
            rc = INQ_EXIT_ERROR;
            std::snprintf(msg, sizeof msg, "internal error: %s", e.what());
        }
        if (timing)
            std::fprintf(stderr, "[inq session] @%.1f file %zu (%s): status %d, %.1f ms since the previous file finished\n", stamp_ms(), k,
                         args[k].bam ? args[k].bam : "?", rc, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (statuses) statuses[k] = rc;
        if (rc != INQ_EXIT_OK) {
            if (!have_msg) set_err(errbuf, errcap, std::string(args[k].bam ? args[k].bam : "?") + ": " + msg), have_msg = true;
            if (worst == INQ_EXIT_OK || rc == INQ_EXIT_PANIC) worst = rc;
        }
        StagedFile done;
        std::swap(done, st[k]);  // header, index, targets of file k go now, not at the end of the cohort
        if (next.valid()) next.get();
    }
    return worst;
}
int inq_session_call_many(inq_session_t *S, const inq_call_args_t *args, size_t n, const int *out_fds, int *statuses, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_session_call_many_impl(S, args, n, out_fds, statuses, errbuf, errcap), errbuf, errcap)
}

struct inq_staged {
    StagedFile f;
};
int inq_session_stage(inq_session_t *S, const inq_call_args_t *args, inq_staged_t **out) {
    if (!S || !args || !out) return INQ_EXIT_ERROR;
    *out = nullptr;
    try {
        std::unique_ptr<inq_staged> st(new inq_staged());
        stage_file(S, args, 3 * (int)(S->n_staged++ & 1u), st->f);  // what it finds wrong is reported by inq_session_run
        *out = st.release();
        return INQ_EXIT_OK;
    } catch (...) {
        return INQ_EXIT_ERROR;
    }
}
static int inq_session_run_impl(inq_session_t *S, inq_staged_t *st, int out_fd, char *errbuf, size_t errcap) {
    if (!S || !st) return INQ_EXIT_ERROR;
    std::unique_ptr<inq_staged> own(st);
    return run_staged(S, own->f, out_fd, errbuf, errcap);
}
int inq_session_run(inq_session_t *S, inq_staged_t *st, int out_fd, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_session_run_impl(S, st, out_fd, errbuf, errcap), errbuf, errcap)
}
void inq_session_discard(inq_staged_t *st) { delete st; }

void inq_session_close(inq_session_t *S) {
    if (!S) return;
    const char *fast_env = std::getenv("INQ_FAST_EXIT");
    S->actx.leak = fast_env && fast_env[0] == '1';  // the CLI is about to leave the process (see run_device_front)
    if (S->actx.leak) {
        S->actx.wait();
        S->pool.free_list.clear();  // left to the operating system as well
    }
    delete S;
}

// ---- spans: the host half of the device front end, on its own (no GPU involved) ----
static int inq_spans_open_impl(const inq_call_args_t *args, uint64_t max_comp_bytes, inq_spans_t **out, char *errbuf, size_t errcap) {
    if (!out) return INQ_EXIT_ERROR;
    *out = nullptr;
    std::unique_ptr<inq_spans> S(new inq_spans());
    std::string msg;
    int rc = prepare(args, S->P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    S->minlen = args->minlen;
    S->support = (uint32_t)std::min<uint64_t>(args->support, 0xffffffffull);
    S->unphased = args->unphased != 0;
    S->pipe.reset(new SpanPipeline(args->bam, *S->P.bam, S->P.targets, max_comp_bytes ? max_comp_bytes : span_bytes_from_env(),
                                   (int)std::max<uint64_t>(1, std::min<uint64_t>(args->threads, 32)), false));
    *out = S.release();
    return INQ_EXIT_OK;
}

static int inq_spans_next_impl(inq_spans_t *S, inq_span_t *sp, const uint32_t **locus_index, uint64_t *file_begin, char *errbuf,
                               size_t errcap) {
    if (!S || !sp) return -INQ_EXIT_ERROR;
    if (S->cur) S->pipe->release(S->cur);
    S->cur = nullptr;
    std::string err;
    int rc = S->pipe->next(S->cur, &err);
    if (rc < 0) {
        set_err(errbuf, errcap, err);
        return -INQ_EXIT_PANIC;
    }
    if (rc == 0) return 0;
    SpanPipeline::Item *it = S->cur;
    SpanPipeline::fill_span(*it, sp);
    sp->minlen = S->minlen;
    sp->support = S->support;
    sp->unphased = S->unphased ? 1u : 0u;
    if (locus_index) *locus_index = it->plan.locus_index.data();
    if (file_begin) *file_begin = it->data.file_begin;
    return 1;
}

int inq_spans_open(const inq_call_args_t *args, uint64_t max_comp_bytes, inq_spans_t **out, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_spans_open_impl(args, max_comp_bytes, out, errbuf, errcap), errbuf, errcap)
}
int inq_spans_next(inq_spans_t *S, inq_span_t *span, const uint32_t **locus_index, uint64_t *file_begin, char *errbuf, size_t errcap) {
    try {
        return inq_spans_next_impl(S, span, locus_index, file_begin, errbuf, errcap);
    } catch (...) {
        set_err(errbuf, errcap, "internal error");
        return -INQ_EXIT_ERROR;
    }
}
uint64_t inq_spans_n_targets(const inq_spans_t *S) { return S ? S->P.targets.size() : 0; }
void inq_spans_close(inq_spans_t *S) {
    if (S && S->cur) S->pipe->release(S->cur);
    delete S;
}

size_t inq_host_format_f64(double v, char *buf, size_t cap) { return (size_t)std::snprintf(buf, cap, "%s", format_f64(v).c_str()); }
size_t inq_host_format_row(const char *chrom, uint32_t start, uint32_t end, double p1, double p2, char *buf, size_t cap) {
    return (size_t)std::snprintf(buf, cap, "%s", format_row(chrom, start, end, p1, p2).c_str());
}
size_t inq_host_format_header(const char *sample, char *buf, size_t cap) {
    return (size_t)std::snprintf(buf, cap, "%s", format_header(sample).c_str());
}
size_t inq_host_sample_name(const char *p, char *buf, size_t cap) {
    return (size_t)std::snprintf(buf, cap, "%s", sample_name_from_path(p).c_str());
}
int inq_host_human_compare(const char *a, const char *b) { return human_compare(a, b); }

int inq_host_parse_region(const char *reg, const char *chrom_name, uint64_t chrom_len, char *chrom_out, size_t cap,
                          uint32_t *start, uint32_t *end) {
    std::map<std::string, uint64_t> lens;
    if (chrom_name) lens[chrom_name] = chrom_len;
    TargetsResult r = targets_from_string(reg, lens);
    if (r.panicked) {
        if (chrom_out && cap) std::snprintf(chrom_out, cap, "%s", r.message.c_str());
        return INQ_EXIT_PANIC;
    }
    if (chrom_out && cap) std::snprintf(chrom_out, cap, "%s", r.data[0].chrom.c_str());
    if (start) *start = r.data[0].start;
    if (end) *end = r.data[0].end;
    return INQ_EXIT_OK;
}

uint64_t inq_host_bai_file_offset(const char *bai_path, int32_t tid, int64_t pos) {
    try {
        static std::mutex mu;
        static std::string cached_path;
        static BaiIndex cached;
        std::lock_guard<std::mutex> g(mu);
        if (cached_path != bai_path) {
            std::string e;
            BaiIndex idx;
            if (!idx.load(bai_path, &e)) return 0;
            cached = std::move(idx);
            cached_path = bai_path;
        }
        return cached.scan_start(tid, pos) >> 16;
    } catch (...) {
        return 0;
    }
}

uint64_t inq_host_bai_scan_start(const char *bai_path, int32_t tid, int64_t pos) {
    try {
        BaiIndex idx;
        std::string e;
        const std::string p = bai_path;
        const bool is_csi = p.size() > 4 && p.compare(p.size() - 4, 4, ".csi") == 0;
        if (!(is_csi ? idx.load_csi(p, &e) : idx.load(p, &e))) return 0;
        return idx.scan_start(tid, pos);
    } catch (...) {
        return 0;
    }
}

// The plan alone: every segment of every span the device front end would read for these targets, nothing read from the BAM
// beyond its header and index.
static int inq_host_plan_spans_impl(const inq_call_args_t *args, uint64_t max_comp_bytes, uint64_t *seg_vo_begin, uint64_t *seg_vo_limit,
                                    uint32_t *seg_span, uint64_t seg_cap, uint64_t *n_segs, uint32_t *target_span, uint64_t target_cap,
                                    char *errbuf, size_t errcap) {
    if (!n_segs) return INQ_EXIT_ERROR;
    Prepared P;
    std::string msg;
    int rc = prepare(args, P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    if (target_span) {
        if (target_cap < P.targets.size()) {
            set_err(errbuf, errcap, "target_span[] too small for the target list");
            return INQ_EXIT_ERROR;
        }
        for (size_t i = 0; i < P.targets.size(); ++i) target_span[i] = 0xffffffffu;
    }
    SpanPlanner planner(*P.bam, P.targets, max_comp_bytes ? max_comp_bytes : span_bytes_from_env());
    SpanPlan plan;
    uint64_t n = 0;
    uint32_t span = 0;
    while (planner.next(plan)) {
        for (const Segment &g : plan.segs) {
            if (n < seg_cap && seg_vo_begin && seg_vo_limit) {
                seg_vo_begin[n] = g.vo_begin;
                seg_vo_limit[n] = g.vo_limit;
                if (seg_span) seg_span[n] = span;
            }
            ++n;
        }
        if (target_span)
            for (uint32_t i : plan.locus_index) target_span[i] = span;
        ++span;
    }
    *n_segs = n;
    return INQ_EXIT_OK;
}
int inq_host_plan_spans(const inq_call_args_t *args, uint64_t max_comp_bytes, uint64_t *seg_vo_begin, uint64_t *seg_vo_limit, uint32_t *seg_span,
                        uint64_t seg_cap, uint64_t *n_segs, uint32_t *target_span, uint64_t target_cap, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_host_plan_spans_impl(args, max_comp_bytes, seg_vo_begin, seg_vo_limit, seg_span, seg_cap, n_segs, target_span, target_cap, errbuf,
                                       errcap),
              errbuf, errcap)
}

int inq_host_bam_tid(const char *bam_path, const char *contig) {
    try {
        BamFile b(1);
        std::string e;
        if (!b.open(bam_path, &e)) return -2;
        return b.tid(contig);
    } catch (...) {
        return -2;
    }
}

int inq_host_bai_stats(const char *bai_path, uint32_t *n_ref, int32_t tid, uint64_t *n_mapped, uint64_t *n_unmapped,
                       uint64_t *n_bins, uint64_t *n_intv) {
    BaiIndex idx;
    std::string e;
    if (!idx.load(bai_path, &e)) return -1;
    if (n_ref) *n_ref = (uint32_t)idx.refs.size();
    if (tid >= 0 && (size_t)tid < idx.refs.size()) {
        if (n_mapped) *n_mapped = idx.refs[tid].n_mapped;
        if (n_unmapped) *n_unmapped = idx.refs[tid].n_unmapped;
        if (n_bins) *n_bins = idx.refs[tid].bins.size();
        if (n_intv) *n_intv = idx.refs[tid].ioffset.size();
    }
    return 0;
}

}  // extern "C"
