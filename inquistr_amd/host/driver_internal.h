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
struct AsyncCtx {
    inq_ctx_t *ctx = nullptr;
    int hrc = INQ_OK;
    int numa_node = -1;
    std::atomic<bool> ready{false};  // ctx / hrc / numa_node are final
    // raised by inq_ctx_create_early as soon as spans may be STAGED on ctx (uploads, inflates), ~30 ms before the context is
    // complete: the uploader thread starts on the spans the loader has read by then
    volatile int stage_ready = 0;
    std::thread th;
    bool leak = false;  // set after a clean run when the process is about to exit (INQ_FAST_EXIT)
    void start(int device) {
        th = std::thread([this, device] {
            prefer_gpu_node_for_this_thread(device);  // what the runtime allocates while it starts
            const double a = stamp_ms();
            hrc = inq_ctx_create_early(device, &ctx, &stage_ready);
            numa_node = hrc == INQ_OK ? inq_ctx_numa_node(ctx) : -1;
            ready.store(true);
            const char *e = std::getenv("INQ_TIMING");
            if (e && e[0] == '2') std::fprintf(stderr, "[inq ctx] @%.1f device context ready (inq_ctx_create %.1f ms), GPU on NUMA node %d\n", stamp_ms(), stamp_ms() - a, numa_node);
        });
    }
    // any thread: true once spans may be staged on ctx (false: the context could not be made)
    bool wait_stage() {
        while (!__atomic_load_n(&stage_ready, __ATOMIC_ACQUIRE)) {
            if (ready.load()) return hrc == INQ_OK || __atomic_load_n(&stage_ready, __ATOMIC_ACQUIRE) != 0;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        return true;
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

// rows instead of text: the targets named by idx[] (positions in the parsed target list) are called, their rows go to p1 / p2
struct RowsOut {
    const uint32_t *idx = nullptr;
    uint64_t n = 0;
    double *p1 = nullptr, *p2 = nullptr;
    bool active = false;
};

bool use_device_front(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets);
int span_io_threads(const inq_call_args_t *args);
SpanPipeline *start_span_pipeline(const inq_call_args_t *args, const BamFile &bam, const std::vector<RepeatInterval> &targets, AsyncCtx &actx,
                                  int slot_base, HostBufPool *pool);
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
