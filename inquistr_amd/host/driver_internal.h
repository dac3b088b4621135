// driver_internal.h - what the pieces of the host driver share (run.cc, span_pipeline.cc, session.cc, combine.cc,
// hostapi_probes.cc): the prepared call, the asynchronously created device context, the hooks a session adds to a call.
#pragma once
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
#include "front_pool.h"
#include "inq_text.h"
#include "sa2d.h"
#include "span_pipeline.h"
#include "span_planner.h"
#include "targets.h"

namespace inqhost {

// Experiments' switches (INQ_SPAN_REGISTER, INQ_SPAN_BUFFERS, INQ_IO_PIN, INQ_EXIT_PROBE, INQ_GATE_READS, INQ_SPAN_PINNED) are read from
// the environment only by a build with -DINQ_DEBUG_ENV (make DEBUG_ENV=1).  What the shipped host library reads is listed in
// INQ_TIMING, INQ_FRONTEND, INQ_SPAN_MB, INQ_SPAN_GAP_BYTES, INQ_SPAN_JOB_BYTES, INQ_FLUSH_LOCI, INQ_NUMA_NODE, INQ_NUMA_CPUS,
// INQ_CTX_TIMEOUT_S, INQ_FAST_EXIT, INQ_SERVER, INQ_SERVER_IDLE, INQ_HOST_LIB, and torch.distributed.run's LOCAL_WORLD_SIZE / LOCAL_RANK.
#ifdef INQ_DEBUG_ENV
inline const char *debug_env(const char *name) { return std::getenv(name); }
#else
inline const char *debug_env(const char *) { return nullptr; }
#endif

inline void set_err(char *buf, size_t cap, const std::string &m) {
    if (buf && cap) std::snprintf(buf, cap, "%s", m.c_str());
}
inline bool starts_with(const std::string &s, const char *p) { return s.compare(0, std::strlen(p), p) == 0; }
inline bool ends_with(const std::string &s, const char *p) {
    size_t n = std::strlen(p);
    return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}
inline bool is_file(const std::string &p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
bool write_all(int fd, const char *data, size_t len);
inline bool write_all(int fd, const std::string &s) { return write_all(fd, s.data(), s.size()); }

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

int prepare(const inq_call_args_t *a, Prepared &P, std::string &msg, BedCache *bed_cache = nullptr);


// The device context is created on its own thread from the first instruction of the command: HIP start-up
// (0.1 - 0.3 s) is the longest fixed cost of a run and overlaps opening the BAM, the BED, the .bai and the
// first reads of the file.
//
// Every wait for that thread is BOUNDED (ctx_timeout_s(): 60 s, INQ_CTX_TIMEOUT_S overrides).  Round 4 lost a GPU box to the case it
// was not: a profiler's tool library aborted (glog FATAL -> SIGABRT) inside the runtime's first call, ON the context thread
// (gpurun_out/prof_cli_locus/p2.log: inq_ctx_create_early <- AsyncCtx::start), its signal handler then "finalized" there for ever, and
// the process stayed: the uploader polled `stage_ready` without end, the caller sat in SpanPipeline::next() behind it, and a join of the
// context thread would have sat there too.  Now the thread's state lives in a block it shares with its owner; whoever waits gives up
// after the time-out, the context counts as failed (INQ_ERR_HIP, "did not come up"), the thread is left behind (detached: it may
// be stuck inside a signal handler or the driver) and the call ends with INQ_EXIT_ERROR.
double ctx_timeout_s();
// A caller short of cores (its share of the granted cores below 8: eight ranks on a 16-core grant have 2 each) cannot afford the two
// cores that SPIN while a span inflates (the caller's thread in hipStreamSynchronize) and crosses the link (the uploader's in
// hipEventSynchronize): they are the readers' (measured with the process confined to 4 CPUs: tools/few_cores.sh).  The context is then
// made with blocking waits ("blocking_sync"), unless the user has set the option either way (--ctx-option, inq_default_option).
// sharers: 0 = local_share().
void choose_wait_mode(int sharers);
typedef int (*CtxCreateFn)(int device, inq_ctx_t **ctx, volatile int *stage_ready);
CtxCreateFn ctx_create_fn();  // inq_ctx_create_early, or what inq_host_test_ctx_creator put in its place (tests)
struct AsyncCtx {
    struct State {
        inq_ctx_t *ctx = nullptr;
        int hrc = INQ_OK;
        int numa_node = -1;
        std::atomic<bool> ready{false};  // ctx / hrc / numa_node are final
        // raised by inq_ctx_create_early as soon as spans may be STAGED on ctx (uploads, inflates), ~30 ms before the context is
        // complete: the uploader thread starts on the spans the loader has read by then
        volatile int stage_ready = 0;
        std::atomic<bool> gave_up{false};  // a waiter ran into the time-out: hrc is the waiter's verdict, the thread's result is ignored
        std::mutex mu;
        std::condition_variable cv;
    };
    std::shared_ptr<State> st = std::make_shared<State>();
    inq_ctx_t *&ctx = st->ctx;
    int &hrc = st->hrc;
    int &numa_node = st->numa_node;
    std::atomic<bool> &ready = st->ready;
    std::thread th;
    bool started = false;
    bool leak = false;  // set after a clean run when the process is about to exit (INQ_FAST_EXIT)
    std::chrono::steady_clock::time_point t_start;
    AsyncCtx() = default;
    AsyncCtx(const AsyncCtx &) = delete;
    void start(int device, int sharers = 0) {
        t_start = std::chrono::steady_clock::now();
        started = true;
        choose_wait_mode(sharers);
        std::shared_ptr<State> s = st;  // the thread keeps the block alive, whatever becomes of this object
        CtxCreateFn create = ctx_create_fn();
        th = std::thread([s, device, create] {
            prefer_gpu_node_for_this_thread(device);  // what the runtime allocates while it starts
            const double a = stamp_ms();
            inq_ctx_t *c = nullptr;
            // (the early form publishes through these two words; they are read by wait_stage() only)
            const int rc = create(device, &s->ctx, &s->stage_ready);
            c = s->ctx;
            const int node = rc == INQ_OK ? inq_ctx_numa_node(c) : -1;
            {
                std::lock_guard<std::mutex> g(s->mu);
                if (!s->gave_up.load()) s->hrc = rc, s->numa_node = node;
                s->ready.store(true);
            }
            s->cv.notify_all();
            const char *e = std::getenv("INQ_TIMING");
            if (e && e[0] == '2') std::fprintf(stderr, "[inq ctx] @%.1f device context ready (inq_ctx_create %.1f ms), GPU on NUMA node %d\n", stamp_ms(), stamp_ms() - a, node);
        });
    }
    // marks the context as failed because its thread did not finish in time; the thread is left to itself
    void give_up_locked() {
        if (st->gave_up.exchange(true)) return;
        st->hrc = INQ_ERR_HIP;
        st->numa_node = -1;
        if (th.joinable()) th.detach();
    }
    double seconds_left() const {
        return ctx_timeout_s() - std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    }
    // any thread: true once spans may be staged on ctx (false: the context could not be made, or not in time)
    bool wait_stage() {
        while (!__atomic_load_n(&st->stage_ready, __ATOMIC_ACQUIRE)) {
            if (ready.load()) return hrc == INQ_OK && !st->gave_up.load() && __atomic_load_n(&st->stage_ready, __ATOMIC_ACQUIRE) != 0;
            if (st->gave_up.load()) return false;
            if (seconds_left() <= 0) {
                std::lock_guard<std::mutex> g(st->mu);
                if (!ready.load()) give_up_locked();
                return false;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        return ctx != nullptr && !st->gave_up.load();
    }
    std::mutex join_mu;
    bool wait() {  // any thread: the context is complete (true) or there is none (false); never longer than the time-out
        if (!started) return false;
        {
            std::unique_lock<std::mutex> g(st->mu);
            while (!ready.load() && !st->gave_up.load()) {
                const double left = seconds_left();
                if (left <= 0) {
                    give_up_locked();
                    break;
                }
                st->cv.wait_for(g, std::chrono::duration<double>(std::min(left, 0.25)));
            }
        }
        std::lock_guard<std::mutex> g(join_mu);
        if (!st->gave_up.load() && th.joinable()) th.join();
        return hrc == INQ_OK && !st->gave_up.load();
    }
    bool timed_out() const { return st->gave_up.load(); }
    ~AsyncCtx() {
        if (started) (void)wait();
        // a context whose thread was given up on is never touched again (the thread may still be inside its constructor)
        if (!leak && !st->gave_up.load()) inq_ctx_destroy(ctx);
    }
};

}  // namespace inqhost

// a session: many BAMs on ONE device context (session.cc; a prepared run may be opened on one: run.cc inq_session_run_open)
struct inq_session {
    inqhost::AsyncCtx actx;
    inqhost::HostBufPool pool;
    inqhost::BedCache bed_cache;
    int32_t device = 0;
    uint64_t n_staged = 0;  // inq_session_stage: which of the two sets of device slots the next file takes
};

namespace inqhost {

// How many processes / device parts share this host's cores with the caller, and which of them it is: the reader pool of a span
// pipeline takes the granted cores (sched_getaffinity, cut by the cgroup's CPU quota) divided by that number, and binds its threads to
// L3 domains starting at a different one per sharer.  Default: LOCAL_WORLD_SIZE / LOCAL_RANK as torch.distributed.run exports them
// (one process per GPU), else 1 / 0; inq_host_set_local_share overrides; the one-process --devices entry passes its own per part.
int granted_cpus();
void local_share(int *sharers, int *index);
void set_local_share(int sharers, int index);
void set_ctx_creator_for_tests(CtxCreateFn f, long timeout_ms);  // f = nullptr: the real one; timeout_ms < 0: the default

// what a device part of a multi-device call reports (inq_part_stats_t of include/inquistr_host.h is filled from it)
struct PartStats {
    uint64_t spans = 0, comp_bytes = 0;
    double span_loop_s = 0, wait_loader_s = 0, device_calls_s = 0;
    int front = 0;  // 1 = host sweep, 2 = device spans
    int io_threads = 0;
};

void publish_last_stats(const PartStats &s);
PartStats last_stats();  // of the last call that ended in this process (any thread)

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
    int sharers = 0, share_index = 0;  // > 0: this call is one of `sharers` device parts of one process (else: local_share())
    PartStats *stats = nullptr;        // may be null
    // rows stay on the device (inq_run_rows_device): row of target k of the call -> dev_p1[k], dev_p2[k]
    double *dev_p1 = nullptr, *dev_p2 = nullptr;
    uint64_t dev_cap = 0;
};

// rows instead of text: the targets named by idx[] (positions in the parsed target list) are called, their rows go to p1 / p2
struct RowsOut {
    const uint32_t *idx = nullptr;
    uint64_t n = 0;
    double *p1 = nullptr, *p2 = nullptr;  // host arrays (null when the rows stay on the device)
    bool active = false;
    double *d1 = nullptr, *d2 = nullptr;  // DEVICE arrays of dcap entries on the call's device: the rows are left there
    uint64_t dcap = 0;
};

bool use_device_front(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets);
std::string ctx_failure_message(AsyncCtx &actx);
int span_io_threads(const inq_call_args_t *args, int sharers);
SpanPipeline *start_span_pipeline(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets, AsyncCtx &actx,
                                  int slot_base, HostBufPool *pool, int sharers = 0, int share_index = 0);
int run_device_front(const inq_call_args_t *args, const CallView &V, AsyncCtx &actx, std::vector<double> &p1, std::vector<double> &p2,
                     char *errbuf, size_t errcap, double *t_front, double *t_dev, const SessionHooks &hooks = SessionHooks());
int genotype_prepared(const inq_call_args_t *args, AsyncCtx &actx, Prepared &P, int out_fd, char *errbuf, size_t errcap, const RowsOut &rows,
                      std::chrono::steady_clock::time_point t_start, const SessionHooks &hooks = SessionHooks());
int write_rows(uint64_t threads, const std::vector<RepeatInterval> &targets, const std::string &sample, const double *p1, const double *p2,
               int out_fd, char *errbuf, size_t errcap);
int partition_prepared(Prepared &P, uint64_t world, uint32_t *order, uint64_t *cuts);

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

}  // namespace inqhost

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
