// span_pipeline.cc - the device front end as the host sees it: NUMA placement of the threads that read and upload, span buffers,
// the loader / uploader pipeline, the loop that feeds inq_call_span_deferred (src/call.rs:288,294,338,345 are what it replaces),
// and the inq_spans_* entry points (the host half on its own).
#include "driver_internal.h"

using namespace inqhost;

namespace inqhost {

static std::atomic<uint64_t> g_span_bytes_read{0};  // inq_host_span_bytes_read

// INQ_TIMING=2 stamps every stage with milliseconds since the library was loaded (about the start of the process)
const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
double stamp_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_t0).count(); }

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
int guess_gpu_numa_node(int device) {
    // (never destroyed: a context thread that was given up on - AsyncCtx - may still come through here while the process exits)
    static std::mutex &mu = *new std::mutex();
    static std::map<int, int> &memo = *new std::map<int, int>();
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

void prefer_gpu_node_for_this_thread(int device) {
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

void prefer_numa_node(void *p, size_t len, int node) {
    if (node < 0 || node >= 1024) return;
    unsigned long mask[16] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    (void)::syscall(SYS_mbind, p, len, 1 /* MPOL_PREFERRED */, mask, sizeof mask * 8, 0);  // best effort: placement only
}

void SpanPipeline::release_buf(Item &it) {
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

bool SpanPipeline::fit(Item &it, size_t bytes) {
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

CpuByKind &cpu_by_kind() {
    static CpuByKind &k = *new CpuByKind();
    return k;
}
uint64_t thread_cpu_us() {
    struct timespec t;
    if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t) != 0) return 0;
    return (uint64_t)t.tv_sec * 1000000ull + (uint64_t)t.tv_nsec / 1000ull;
}
namespace {
struct CpuLap {  // adds the calling thread's CPU time between construction and destruction to a counter
    std::atomic<uint64_t> &to;
    uint64_t t0 = thread_cpu_us();
    ~CpuLap() { to.fetch_add(thread_cpu_us() - t0, std::memory_order_relaxed); }
};
}  // namespace

void SpanPipeline::run() {
    CpuLap cpu_lap{cpu_by_kind().loader_us};
    prefer_gpu_node_for_this_thread(device_);
    SpanLoader loader;
    std::string e;
    if (!loader.open(path_, &e)) return fail(e);
    // the reader threads of this file (span_planner.h: spread over the L3 domains of the GPU's NUMA node)
    const char *pin_env = debug_env("INQ_IO_PIN");
    IoPool pool(n_threads_, guess_gpu_numa_node(device_), !(pin_env && pin_env[0] == '0'), io_group_offset_);
    if (verbose_) std::fprintf(stderr, "[inq loader] @%.1f %s\n", stamp_ms(), pool.layout().c_str());
    // span k + 1 is planned (index searches for some 25 000 loci: ~4 ms) on a helper thread while span k is being read
    SpanPlan ahead;
    bool have = planner_.next(ahead);
    // INQ_GATE_READS=1: the first read waits for the device context.  Tried because the runtime's start-up looked longer
    // while the loader was reading (two boxes, 102 vs 221 ms); five runs each way on a third box showed the start-up
    // varying between 108 and 431 ms with and without early reads alike (profiles/r03_results/loader_gate_ab.txt): it
    // is hipInit itself that varies (57 - 224 ms: round 4 found the cause - the driver still taking the PREVIOUS process of the timing
    // loop apart), so reads start at once - the spans are there when the context is.
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
        if (!loader.load(it->plan, planner_.anchors(), it->buf, n_threads_, it->data, &e, &pool)) return fail(e);
        it->staged = false;
        if (verbose_) {
            const auto t3 = std::chrono::steady_clock::now();
            {   // where the buffer's pages lie (a sample, asked of the kernel), and on which CPUs the readers ran
                const size_t ps = 4096, n_s = 64;
                void *pages[n_s];
                int status[n_s];
                for (size_t k = 0; k < n_s; ++k) pages[k] = it->buf + ((size_t)nbytes / n_s * k) / ps * ps;
                long rc = ::syscall(SYS_move_pages, 0, (unsigned long)n_s, pages, nullptr, status, 0);
                int on[4] = {0, 0, 0, 0};
                if (rc >= 0)
                    for (size_t k = 0; k < n_s; ++k)
                        if (status[k] >= 0 && status[k] < 4) on[status[k]]++;
                std::fprintf(stderr, "[inq loader] slot %d buffer pages by node (64 samples): %d %d %d %d; reader CPUs: %s\n", it->slot, on[0], on[1], on[2], on[3],
                             pool.last_cpus().c_str());
            }
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[inq loader] @%.1f slot %d: plan %.2f ms, buffer %.2f ms (%s), read+tables %.2f ms (copy %.2f, block table + anchors %.2f) for %.1f MB, %zu segments, %zu anchors\n",
                         stamp_ms(), it->slot, ms(t0, t1), ms(t1, t2), it->pinned ? "pinned" : "pageable", ms(t2, t3), it->data.ms_read, it->data.ms_tables, nbytes / 1e6,
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

// uploads in file order behind the reader; up to two spans enqueued at a time (when the staging has a wait step), so that the copy
// engine goes from one span's bytes straight to the next one's
void SpanPipeline::run_uploads() {
    CpuLap cpu_lap{cpu_by_kind().uploader_us};
    prefer_gpu_node_for_this_thread(device_);  // the runtime's staging chunks are allocated by the thread that first copies
    struct Flight {
        Item *it;
        bool begun;
        std::chrono::steady_clock::time_point t0;
    };
    std::deque<Flight> inflight;
    const size_t depth = stage_wait_ ? 2 : 1;
    // whichever way this thread leaves (the file is through, a failure elsewhere, the pipeline torn down early): no upload may still
    // be reading a span buffer when the buffers are given back
    struct Drain {
        std::deque<Flight> &q;
        WaitFn &wait;
        ~Drain() {
            for (auto &f : q)
                if (f.begun && wait) (void)wait(f.it->slot);
        }
    } drain{inflight, stage_wait_};
    auto finish_oldest = [&] {
        Flight f = inflight.front();
        inflight.pop_front();
        f.it->staged = f.begun && (!stage_wait_ || stage_wait_(f.it->slot));
        if (verbose_)
            std::fprintf(stderr, "[inq loader] @%.1f slot %d: upload %.2f ms for %.1f MB%s\n", stamp_ms(), f.it->slot,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f.t0).count(), f.it->data.comp_bytes / 1e6,
                         f.it->staged ? "" : " (not staged)");
        std::lock_guard<std::mutex> g(mu_);
        ready_.push_back(f.it);
        cv_item_.notify_one();
    };
    for (;;) {
        Item *it = nullptr;
        {
            std::unique_lock<std::mutex> g(mu_);
            // with a span in flight only what is there already is taken: its upload is waited for rather than the reader
            if (inflight.empty()) cv_loaded_.wait(g, [&] { return !loaded_.empty() || load_done_ || stop_ || failed_; });
            if (stop_ || failed_) return;
            if (!loaded_.empty() && inflight.size() < depth) {
                it = loaded_.front();
                loaded_.pop_front();
            } else if (inflight.empty()) {  // the reader is through
                done_ = true;
                cv_item_.notify_all();
                return;
            }
        }
        if (it) {
            const auto t0 = std::chrono::steady_clock::now();
            inq_span_t sp;
            fill_span(*it, &sp);
            if (register_ && !it->registered && !it->pinned && it->buf) {
                // INQ_SPAN_REGISTER=1 (experiment): the buffer is page-locked where it lies before its first upload
                if (gate_registered_) gate_registered_();
                it->registered = inq_pin_host(it->buf, it->cap) == INQ_OK;
            }
            inflight.push_back(Flight{it, stage_(sp, it->slot), t0});
            if (inflight.size() < depth) continue;  // a second one, if the reader has it
        }
        if (!inflight.empty()) finish_oldest();
    }
}

uint64_t span_bytes_from_env() {
    const char *e = std::getenv("INQ_SPAN_MB");
    const long v = e ? std::atol(e) : 0;
    // ~10 000 BGZF blocks per span: the workgroup-per-block inflate has no latency floor (1.1 ms per 1000 blocks), so small
    // spans cost nothing and the device starts on the first one while the loader still reads and uploads the next ones
    // (1 GB file: 256 MB spans 0.28 s median start to exit, one 1 GB span 0.37 s; profiles/r02_results/l2_span_size.txt)
    return v > 0 ? (uint64_t)v << 20 : (256ull << 20);
}


// auto front-end choice: the device front end inflates a BGZF block per GPU lane, which takes ~40 ms however
// few blocks there are; the CPU sweep inflates ~70 MB/s of BAM per thread.  Below this many compressed bytes
// per host thread the sweep is as quick (16 MiB at -t 16, 1 MiB at -t 1; round 1's lane-per-block inflate put the line at 3 MiB:
// profiles/r02_results/front_end_choice.txt).
static constexpr uint64_t kDeviceFrontMinBytesPerThread = 1ull << 20;

// front end selection: args->reserved 1 = host sweep (BGZF inflate + record decode on CPU threads),
// 2 = device (inq_call_span); 0 = INQ_FRONTEND=host|device, else by the amount of BAM the loci need
bool use_device_front(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets) {
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

static std::atomic<int> g_sharers{0}, g_share_index{0};  // inq_host_set_local_share; 0 = not set

// ---- the device context's thread: its time-out, what creates the context (tests put a failing creator there)
static std::atomic<long> g_ctx_timeout_ms{-1};
static std::atomic<CtxCreateFn> g_ctx_create{nullptr};
double ctx_timeout_s() {
    const long forced = g_ctx_timeout_ms.load();
    if (forced >= 0) return forced / 1e3;
    if (const char *e = std::getenv("INQ_CTX_TIMEOUT_S")) {
        const double v = std::atof(e);
        if (v > 0) return v;
    }
    return 60.0;
}
CtxCreateFn ctx_create_fn() {
    CtxCreateFn f = g_ctx_create.load();
    return f ? f : &inq_ctx_create_early;
}
void set_ctx_creator_for_tests(CtxCreateFn f, long timeout_ms) {
    g_ctx_create.store(f);
    g_ctx_timeout_ms.store(timeout_ms);
}
std::string ctx_failure_message(AsyncCtx &actx) {
    if (actx.timed_out()) {
        char b[160];
        std::snprintf(b, sizeof b, "cannot open HIP device: the device context did not come up within %.0f s (its thread is left behind)", ctx_timeout_s());
        return b;
    }
    return std::string("cannot open HIP device: ") + inq_strerror(actx.hrc);
}
void set_local_share(int sharers, int index) {
    g_sharers.store(sharers);
    g_share_index.store(index);
}

// fills p1 / p2 through the device front end; returns an exit status

// The cores this process may really use: its affinity mask, cut by the cgroup's CPU quota when one is set (a container on a 256-thread
// host is typically granted 16: std::thread::hardware_concurrency() says 256 there).
int granted_cpus() {
    cpu_set_t have;
    CPU_ZERO(&have);
    int n = sched_getaffinity(0, sizeof have, &have) == 0 ? CPU_COUNT(&have) : (int)std::max(1u, std::thread::hardware_concurrency());
    auto read_pair = [](const char *path, long long *a, long long *b) -> bool {
        FILE *f = std::fopen(path, "r");
        if (!f) return false;
        char x[64] = {0}, y[64] = {0};
        const int got = std::fscanf(f, "%63s %63s", x, y);
        std::fclose(f);
        if (got < 1 || std::strcmp(x, "max") == 0) return false;
        *a = std::atoll(x);
        *b = got == 2 ? std::atoll(y) : 0;
        return true;
    };
    long long q = 0, per = 0;
    if (read_pair("/sys/fs/cgroup/cpu.max", &q, &per) && q > 0 && per > 0) n = (int)std::min<long long>(n, std::max<long long>(1, q / per));
    else if (read_pair("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q, &per) && q > 0) {
        long long p2 = 0, dummy = 0;
        if (read_pair("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &p2, &dummy) && p2 > 0) n = (int)std::min<long long>(n, std::max<long long>(1, q / p2));
    }
    return std::max(1, n);
}

void local_share(int *sharers, int *index) {
    int n = g_sharers.load(), k = g_share_index.load();
    if (n <= 0) {  // one process per GPU under torch.distributed.run: it exports both
        const char *w = std::getenv("LOCAL_WORLD_SIZE"), *r = std::getenv("LOCAL_RANK");
        n = w ? std::atoi(w) : 1;
        k = r ? std::atoi(r) : 0;
    }
    if (n < 1) n = 1;
    if (k < 0 || k >= n) k = 0;
    *sharers = n;
    *index = k;
}

void choose_wait_mode(int sharers) {
    int64_t v = 0;
    if (inq_default_option_get("blocking_sync", &v) == INQ_OK) return;  // the user's word
    int idx = 0;
    if (sharers <= 0) local_share(&sharers, &idx);
    if (granted_cpus() / std::max(1, sharers) < 8) (void)inq_default_option("blocking_sync", 1);
}

int span_io_threads(const inq_call_args_t *args, int sharers) {
    // -t counts the reference's calling workers; here the host only copies file bytes, which a few pread streams do best whatever -t
    // says - bounded by this caller's SHARE of the cores the process was granted: eight ranks on one host must not start 8 x 32
    // readers on the cores of one socket, and a container's quota is not the machine (hardware_concurrency())
    int idx = 0;
    if (sharers <= 0) local_share(&sharers, &idx);
    const int share = std::max(2, granted_cpus() / std::max(1, sharers));
    return (int)std::min<uint64_t>(std::max<uint64_t>(args->threads, 8), (uint64_t)std::min(share, 32));
}

SpanPipeline *start_span_pipeline(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets,
                                         AsyncCtx &actx, int slot_base, HostBufPool *pool, int sharers, int share_index) {
    if (sharers <= 0) local_share(&sharers, &share_index);
    bool pinned = false;
    std::function<void()> gate;
    // A/B switches of rounds 2 - 3, read only by a build with -DINQ_DEBUG_ENV
    if (const char *pin_env = debug_env("INQ_SPAN_PINNED")) pinned = pin_env[0] == '1';
    if (debug_env("INQ_GATE_READS")) gate = [&actx] { (void)actx.wait(); };
    // A file (or this caller's share of it) of less than 2 GB is cut into ~8 spans, not into 256 MB ones: with three spans the device
    // idles through the first upload and the last inflate (0.8 GB CIGAR-only file on a resident context: 53.4 ms at 256 MB, 48.4 at
    // 128, 49.8 at 64, 55.6 at 32; files of 1.9 GB and more: 256 MB is as quick or quicker - profiles/r05_results/span_size_for_small_files.txt).
    // INQ_SPAN_MB fixes the size.
    // Only for a device that is READY (a session's later files, a resident rank, a server): a process that has just started reads four
    // spans ahead while its context is being made, and the more bytes those hold the better (the CLI's loop on that file: 21.5 ms
    // with 256 MB spans, 26.8 with 112 MB ones).
    uint64_t span_bytes = span_bytes_from_env();
    if (!std::getenv("INQ_SPAN_MB") && actx.ready.load()) {
        struct stat st;
        if (::stat(args->bam, &st) == 0 && st.st_size > 0) {
            const uint64_t share = (uint64_t)st.st_size / (uint64_t)std::max(1, sharers);
            const uint64_t want = ((share / 8u) + (16ull << 20) - 1) / (16ull << 20) * (16ull << 20);
            span_bytes = std::min<uint64_t>(span_bytes, std::max<uint64_t>(want, 64ull << 20));
        }
    }
    // the loader uploads every span it has read (waiting for the context the first time), so that the upload of span k+1
    // overlaps the inflate of span k
    return new SpanPipeline(args->bam, bam, targets, span_bytes, span_io_threads(args, sharers), pinned,
                            [&actx](const inq_span_t &sp, int slot) { return actx.wait_stage() && inq_span_stage_begin(actx.ctx, &sp, slot) == INQ_OK; },
                            slot_base, pool, gate,
                            [&actx, dev = args->device]() -> int {
                                if (const char *e = std::getenv("INQ_NUMA_NODE")) return std::atoi(e);
                                return actx.ready.load() ? actx.numa_node : guess_gpu_numa_node(dev);
                            },
                            args->device, [&actx] { (void)actx.wait(); },
                            [&actx](int slot) { return !actx.timed_out() && inq_span_stage_wait(actx.ctx, slot) == INQ_OK; }, share_index);
}

int run_device_front(const inq_call_args_t *args, const CallView &V, AsyncCtx &actx, std::vector<double> &p1,
                            std::vector<double> &p2, char *errbuf, size_t errcap, double *t_front, double *t_dev,
                            const SessionHooks &hooks) {
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
        } holder{hooks.early_pipe ? hooks.early_pipe
                                  : start_span_pipeline(args, V.bam, V.targets, actx, hooks.slot_base, hooks.pool, hooks.sharers, hooks.share_index),
                 leak_all, hooks.early_pipe == nullptr};
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
        // the span loop by itself: from the moment the first span is handed to the device (the context is there, the span read and
        // uploaded) to the last flush - what the file costs once the process's fixed costs are behind it
        clk::time_point t_loop0{};
        uint64_t loop_spans = 0, loop_comp_bytes = 0;
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
            // (rows that travel on from device memory stay there: pending[j] is the row's place in the caller's device arrays)
            int rc2 = hooks.dev_p1 ? inq_call_flush_device(ctx, hooks.dev_p1, hooks.dev_p2, hooks.dev_cap, pending.data(), pending.size(), nullptr, &ms_call)
                                   : inq_call_flush(ctx, &res, pending.size(), &ms_call);
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
            if (!hooks.dev_p1)
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
                if (hrc == INQ_OK) (void)inq_ctx_set_option(ctx, "batch_loci_hint", (int64_t)std::min<size_t>(kFlushLoci, V.targets.size()));
            }
            if (hrc != INQ_OK) {
                set_err(errbuf, errcap, ctx_failure_message(actx));
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
            if (loop_spans++ == 0) t_loop0 = clk::now();
            loop_comp_bytes += sp.comp_bytes;
            g_span_bytes_read.fetch_add(sp.comp_bytes, std::memory_order_relaxed);
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
        if (hooks.stats && loop_spans) {
            hooks.stats->spans = loop_spans, hooks.stats->comp_bytes = loop_comp_bytes;
            hooks.stats->span_loop_s = secs(t_loop0, clk::now());
            hooks.stats->wait_loader_s = *t_front, hooks.stats->device_calls_s = *t_dev;
            hooks.stats->io_threads = pipe.io_threads();
        }
        if (timing && loop_spans) {
            const double loop_s = secs(t_loop0, clk::now());
            std::fprintf(stderr, "[inq timing] span loop: %llu spans, %.1f MB compressed, %.4f s from the first span's call to the last flush = %.2f GB/s\n",
                         (unsigned long long)loop_spans, loop_comp_bytes / 1e6, loop_s, loop_comp_bytes / 1e9 / std::max(loop_s, 1e-9));
        }
        leak_all = fast_exit;  // only after a clean run: error paths tear down normally
        if (const char *probe = debug_env("INQ_EXIT_PROBE")) {
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
        if (timing) {
            struct timespec pt;
            const double proc = clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &pt) == 0 ? (double)pt.tv_sec + (double)pt.tv_nsec * 1e-9 : 0.0;
            const CpuByKind &k = cpu_by_kind();
            std::fprintf(stderr,
                         "[inq timing] cpu seconds so far: process %.3f | this thread (device calls) %.3f | reader pool %.3f (the loader's own share of the copies is "
                         "under 'loader' until it ends) | loader %.3f, uploader %.3f (counted when they end) | the rest: the runtime's threads, the context thread\n",
                         proc, thread_cpu_us() * 1e-6, k.readers_us.load() * 1e-6, k.loader_us.load() * 1e-6, k.uploader_us.load() * 1e-6);
        }
    }
    if (timing) std::fprintf(stderr, "[inq timing] loader joined at %.3fs\n", secs(t_begin, clk::now()));
    if (hrc != INQ_OK) {  // no GPU is an error even for an empty target list
        set_err(errbuf, errcap, ctx_failure_message(actx));
        return INQ_EXIT_ERROR;
    }
    return INQ_EXIT_OK;
}

}  // namespace inqhost

struct inq_spans {
    Prepared P;
    std::unique_ptr<SpanPipeline> pipe;
    SpanPipeline::Item *cur = nullptr;
    uint32_t minlen = 5, support = 3;
    bool unphased = false;
};

extern "C" {

uint64_t inq_host_span_bytes_read(void) { return inqhost::g_span_bytes_read.load(std::memory_order_relaxed); }

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


}  // extern "C"
