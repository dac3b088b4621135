// multi_device.cc - `inquiSTR call` on several GPUs of one node from ONE process (include/inquistr_host.h: inq_genotype_repeats_devices;
// CLI: --devices 0,1,...).  The reference's counterpart is the rayon loop over loci (src/call.rs:103-145): workers that share nothing
// but the output Vec.  Here a worker is a device: the targets, in file order, are cut into as many contiguous parts as there are
// devices so that every part needs about the same number of compressed BAM bytes (partition_prepared, from the index); every part
// gets a thread and a device context of its own and runs the whole path on its GPU - its own loader, reader pool, uploader, spans,
// locus kernels -, and its rows land in this process's row arrays: the "gather of per-shard .inq rows" of north_star is a scatter
// through `order[]` in host memory here, no collective at all.  Rank 0's job of the one-process-per-GPU form (inquistr_amd/call_dist.py:
// the ordered .inq text, src/call.rs:137-157) is the calling thread's.  One opened BAM header + index and one parsed target list serve
// all parts (read-only); the host's cores and L3 domains are dealt among the parts' reader pools (span_io_threads, IoPool).
#include "driver_internal.h"

using namespace inqhost;

namespace inqhost {

// part r of `world`: the targets order[cuts[r] .. cuts[r + 1]); fn fills its rows (k-th row = target order[cuts[r] + k]) and returns an
// exit status.  Every part runs on a thread of its own; the rows are scattered into p1 / p2 (size = number of targets) afterwards.
// Returns the status of the first failing part (lowest number), its message prefixed with the part.
using PartFn = std::function<int(size_t part, const uint32_t *idx, uint64_t n, double *p1, double *p2, char *err, size_t errcap)>;

int run_parts(Prepared &P, size_t world, const PartFn &fn, std::vector<double> &p1, std::vector<double> &p2, std::vector<uint32_t> &order,
              std::vector<uint64_t> &cuts, std::vector<int> *part_status, std::vector<double> *part_seconds, char *errbuf, size_t errcap) {
    const size_t n = P.targets.size();
    order.assign(n, 0);
    cuts.assign(world + 1, 0);
    int rc = partition_prepared(P, world, order.data(), cuts.data());
    if (rc != INQ_EXIT_OK) return rc;
    p1.assign(n, NAN);
    p2.assign(n, NAN);
    struct Part {
        std::vector<double> a, b;
        char err[1024] = {0};
        int rc = INQ_EXIT_OK;
        double seconds = 0;
    };
    std::vector<Part> parts(world);
    std::vector<std::thread> th;
    for (size_t r = 0; r < world; ++r) {
        const uint64_t lo = cuts[r], hi = cuts[r + 1];
        parts[r].a.assign(hi - lo, NAN);
        parts[r].b.assign(hi - lo, NAN);
        th.emplace_back([&, r, lo, hi] {
            const auto t0 = std::chrono::steady_clock::now();
            try {
                parts[r].rc = fn(r, order.data() + lo, hi - lo, parts[r].a.data(), parts[r].b.data(), parts[r].err, sizeof parts[r].err);
            } catch (const std::exception &e) {
                parts[r].rc = INQ_EXIT_ERROR;
                std::snprintf(parts[r].err, sizeof parts[r].err, "internal error: %s", e.what());
            } catch (...) {
                parts[r].rc = INQ_EXIT_ERROR;
                std::snprintf(parts[r].err, sizeof parts[r].err, "internal error");
            }
            parts[r].seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    for (auto &t : th) t.join();
    if (part_status) part_status->assign(world, 0);
    if (part_seconds) part_seconds->assign(world, 0.0);
    int first_bad = INQ_EXIT_OK;
    for (size_t r = 0; r < world; ++r) {
        if (part_status) (*part_status)[r] = parts[r].rc;
        if (part_seconds) (*part_seconds)[r] = parts[r].seconds;
        if (parts[r].rc != INQ_EXIT_OK && first_bad == INQ_EXIT_OK) {
            first_bad = parts[r].rc;
            set_err(errbuf, errcap, "part " + std::to_string(r) + " of " + std::to_string(world) + ": " + parts[r].err);
        }
    }
    if (first_bad != INQ_EXIT_OK) return first_bad;
    // the gather of the one-process-per-GPU form, here: rows to their targets' places
    for (size_t r = 0; r < world; ++r)
        for (uint64_t k = cuts[r]; k < cuts[r + 1]; ++k) p1[order[k]] = parts[r].a[k - cuts[r]], p2[order[k]] = parts[r].b[k - cuts[r]];
    return INQ_EXIT_OK;
}

}  // namespace inqhost

extern "C" {

static int genotype_devices_impl(const inq_call_args_t *args, const int32_t *device_ids, size_t n_devices, int out_fd, inq_part_stats_t *stats,
                                 char *errbuf, size_t errcap) {
    if (!args || !device_ids || n_devices == 0 || n_devices > 64) {
        set_err(errbuf, errcap, "device list: between 1 and 64 HIP device ordinals");
        return INQ_EXIT_ERROR;
    }
    for (size_t r = 0; r < n_devices; ++r)
        if (device_ids[r] < 0) {
            set_err(errbuf, errcap, "device list: negative device ordinal");
            return INQ_EXIT_ERROR;
        }
    const auto t_start = std::chrono::steady_clock::now();
    // every device's runtime context starts now, each on its own thread, while the BAM header, the index and the BED are read
    std::vector<std::unique_ptr<AsyncCtx>> actx(n_devices);
    for (size_t r = 0; r < n_devices; ++r) {
        actx[r].reset(new AsyncCtx());
        actx[r]->start(device_ids[r], (int)n_devices);
    }
    Prepared P;
    std::string msg;
    int rc = prepare(args, P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    std::vector<PartStats> pst(n_devices);
    std::vector<double> p1, p2, part_s;
    std::vector<uint32_t> order;
    std::vector<uint64_t> cuts;
    std::vector<int> part_rc;
    const PartFn fn = [&](size_t r, const uint32_t *idx, uint64_t n, double *a, double *b, char *err, size_t cap) -> int {
        OwnedArgs oa(*args);
        oa.a.device = device_ids[r];
        RowsOut ro;
        ro.idx = idx, ro.n = n, ro.p1 = a, ro.p2 = b, ro.active = true;
        SessionHooks hooks;
        hooks.sharers = (int)n_devices, hooks.share_index = (int)r, hooks.stats = &pst[r];
        // (the runtime threads of this part prefer the NUMA node of ITS device: genotype_prepared's helpers bind per thread)
        return genotype_prepared(&oa.a, *actx[r], P, -1, err, cap, ro, t_start, hooks);
    };
    rc = run_parts(P, n_devices, fn, p1, p2, order, cuts, &part_rc, &part_s, errbuf, errcap);
    if (stats)
        for (size_t r = 0; r < n_devices && r < cuts.size() - 1; ++r) {
            inq_part_stats_t &o = stats[r];
            std::memset(&o, 0, sizeof o);
            o.device = device_ids[r];
            o.status = part_rc.empty() ? rc : part_rc[r];
            o.loci = cuts[r + 1] - cuts[r];
            o.spans = pst[r].spans, o.bam_bytes_read = pst[r].comp_bytes;
            o.rows_s = part_s.empty() ? 0.0 : part_s[r];
            o.span_loop_s = pst[r].span_loop_s, o.wait_loader_s = pst[r].wait_loader_s, o.device_calls_s = pst[r].device_calls_s;
            o.front = pst[r].front, o.io_threads = pst[r].io_threads;
        }
    if (rc != INQ_EXIT_OK) return rc;
    return write_rows(args->threads, P.targets, P.sample, p1.data(), p2.data(), out_fd, errbuf, errcap);
}

int inq_genotype_repeats_devices(const inq_call_args_t *args, const int32_t *device_ids, size_t n_devices, int out_fd, inq_part_stats_t *stats,
                                 char *errbuf, size_t errcap) {
    INQ_GUARD(genotype_devices_impl(args, device_ids, n_devices, out_fd, stats, errbuf, errcap), errbuf, errcap)
}

// The same control flow without any GPU (tests): the partition, one thread per part, the scatter and the ordered .inq text of
// inq_genotype_repeats_devices, with every part's rows COMPUTED as a function of what it was handed - phase1 = the target's position in
// the list, phase2 = the part that called it - so that a row in the wrong place, a target called twice or not at all, or a part
// that saw another part's slice shows in the text.  fail_part >= 0: that part reports exit status 101 instead of rows.
static int devices_selftest_impl(const inq_call_args_t *args, size_t n_parts, int fail_part, int out_fd, uint64_t *cuts_out, char *errbuf, size_t errcap) {
    if (!args || n_parts == 0 || n_parts > 64) return INQ_EXIT_ERROR;
    Prepared P;
    std::string msg;
    int rc = prepare(args, P, msg);
    if (rc != INQ_EXIT_OK) {
        set_err(errbuf, errcap, msg);
        return rc;
    }
    std::vector<double> p1, p2;
    std::vector<uint32_t> order;
    std::vector<uint64_t> cuts;
    const PartFn fn = [&](size_t r, const uint32_t *idx, uint64_t n, double *a, double *b, char *err, size_t cap) -> int {
        if ((int)r == fail_part) {
            std::snprintf(err, cap, "injected failure");
            return INQ_EXIT_PANIC;
        }
        for (uint64_t k = 0; k < n; ++k) a[k] = (double)idx[k], b[k] = (double)r;
        return INQ_EXIT_OK;
    };
    rc = run_parts(P, n_parts, fn, p1, p2, order, cuts, nullptr, nullptr, errbuf, errcap);
    if (cuts_out)
        for (size_t r = 0; r < cuts.size(); ++r) cuts_out[r] = cuts[r];
    if (rc != INQ_EXIT_OK) return rc;
    return write_rows(args->threads, P.targets, P.sample, p1.data(), p2.data(), out_fd, errbuf, errcap);
}
int inq_host_devices_selftest(const inq_call_args_t *args, size_t n_parts, int fail_part, int out_fd, uint64_t *cuts, char *errbuf, size_t errcap) {
    INQ_GUARD(devices_selftest_impl(args, n_parts, fail_part, out_fd, cuts, errbuf, errcap), errbuf, errcap)
}

void inq_host_last_call_stats(inq_part_stats_t *out) {
    if (!out) return;
    const PartStats s = last_stats();
    std::memset(out, 0, sizeof *out);
    out->device = -1;
    out->spans = s.spans, out->bam_bytes_read = s.comp_bytes;
    out->span_loop_s = s.span_loop_s, out->wait_loader_s = s.wait_loader_s, out->device_calls_s = s.device_calls_s;
    out->front = s.front, out->io_threads = s.io_threads;
}
void inq_host_set_local_share(int sharers, int index) { set_local_share(sharers, index); }
int inq_host_ctx_option(const char *key, int64_t value) { return inq_default_option(key, value) == INQ_OK ? INQ_EXIT_OK : INQ_EXIT_ERROR; }
int inq_host_granted_cpus(void) { return granted_cpus(); }
int inq_host_span_io_threads(uint64_t threads, int sharers) {
    inq_call_args_t a;
    std::memset(&a, 0, sizeof a);
    a.threads = threads;
    return span_io_threads(&a, sharers);
}

// ---- tests: a device context that fails in a chosen way (no GPU needed) ----
// mode 0 = the real constructor again; 1 = returns INQ_ERR_NO_DEVICE at once (nothing published); 2 = never returns, nothing
// published (a runtime that hangs in its first call); 3 = publishes a null context + stage_ready, then never returns (dies between the
// two halves of inq_ctx_create_early); 4 = publishes a null context + stage_ready, then returns INQ_ERR_HIP.  timeout_ms: what every
// wait for the context thread is bounded by while the mode is set (< 0: the default, 60 s).  Threads of modes 2 / 3 sleep until the mode
// is set back to 0.
static std::atomic<int> g_test_mode{0};
static int failing_ctx_create(int, inq_ctx_t **ctx, volatile int *stage_ready) {
    const int mode = g_test_mode.load();
    if (ctx) *ctx = nullptr;
    if (mode == 1) return INQ_ERR_NO_DEVICE;
    if (mode == 3 || mode == 4) {
        if (stage_ready) __atomic_store_n(stage_ready, 1, __ATOMIC_RELEASE);
        if (mode == 4) return INQ_ERR_HIP;
    }
    while (g_test_mode.load() == mode) std::this_thread::sleep_for(std::chrono::milliseconds(20));
    return INQ_ERR_HIP;
}
void inq_host_test_ctx_creator(int mode, long timeout_ms) {
    g_test_mode.store(mode);
    set_ctx_creator_for_tests(mode == 0 ? nullptr : &failing_ctx_create, mode == 0 ? -1 : timeout_ms);
}

}  // extern "C"
