// session.cc - many BAMs on ONE device context (inq_session_*; `inquistr cohort` and `inquistr serve` sit on it).
#include "driver_internal.h"

using namespace inqhost;

// ---- a session: many BAMs on ONE device context (a cohort is called sample by sample with the same BED, then combined:
// src/combine.rs).  The HIP runtime's start-up (0.1 - 0.3 s, the whole cost of a 1 GB file) is paid once; span buffers are
// reused; and while file k is being called, file k + 1 is opened, its targets parsed, its spans planned, read and uploaded
// into the other set of device staging slots.  Each file's output is byte for byte that of its own `inquistr call`.
// (struct inq_session: driver_internal.h)

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
        S->device = device;
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
        if (k + 1 < n) next = std::async(std::launch::async, [&, k] { stage_file(S, &args[k + 1], SpanPipeline::kSlotsPerSet * (int)((k + 1) & 1), st[k + 1]); });
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
        stage_file(S, args, SpanPipeline::kSlotsPerSet * (int)(S->n_staged++ & 1u), st->f);  // what it finds wrong is reported by inq_session_run
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

