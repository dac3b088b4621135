// run.cc - `call::genotype_repeats` (src/call.rs:76-159) on top of the front ends and the HIP library: the prepared call, the host-sweep
// loop, the output stage, and the entry points of include/inquistr_host.h that run one call (text or rows), split it, or hold it open.
#include "driver_internal.h"

using namespace inqhost;

namespace inqhost {


int prepare(const inq_call_args_t *a, Prepared &P, std::string &msg, BedCache *bed_cache) {
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

// the call on an opened BAM + parsed targets, on a device context that may outlive it (a session calls many BAMs on one)
int genotype_prepared(const inq_call_args_t *args, AsyncCtx &actx, Prepared &P, int out_fd, char *errbuf, size_t errcap,
                             const RowsOut &rows, std::chrono::steady_clock::time_point t_start, const SessionHooks &hooks) {
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
    const bool rows_on_device = rows.active && rows.d1 && rows.d2;
    if (rows_on_device && rows.dcap < n) {
        set_err(errbuf, errcap, "device row arrays smaller than the target list");
        return INQ_EXIT_ERROR;
    }
    auto emit = [&]() -> int {
        if (!rows.active) return write_rows(args->threads, V.targets, V.sample, p1.data(), p2.data(), out_fd, errbuf, errcap);
        if (rows_on_device) return INQ_EXIT_OK;  // (device front end: they are there; host sweep: written below)
        if (n) std::memcpy(rows.p1, p1.data(), n * sizeof(double)), std::memcpy(rows.p2, p2.data(), n * sizeof(double));
        return INQ_EXIT_OK;
    };

    const auto t_open = clk::now();
    const bool device_front = hooks.front ? hooks.front == 2 : use_device_front(args, V.bam, V.targets);
    // what the call did, for whoever asks afterwards (a device part's inq_part_stats_t, inq_host_last_call_stats)
    PartStats local_stats;
    PartStats *ps = hooks.stats ? hooks.stats : &local_stats;
    struct Publish {
        PartStats *p;
        ~Publish() { publish_last_stats(*p); }
    } publish{ps};
    ps->front = device_front ? 2 : 1;
    if (device_front) {
        const auto t_choice = clk::now();
        SessionHooks dh = hooks;
        dh.stats = ps;
        if (rows_on_device) dh.dev_p1 = rows.d1, dh.dev_p2 = rows.d2, dh.dev_cap = rows.dcap;
        int drc = run_device_front(args, V, actx, p1, p2, errbuf, errcap, &t_front, &t_dev, dh);
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
            set_err(errbuf, errcap, ctx_failure_message(actx));
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
    if (rows_on_device && n) {  // the sweep's rows came back batch by batch (small inputs): one copy up
        if (inq_dev_write_rows(ctx, rows.d1, p1.data(), n) != INQ_OK || inq_dev_write_rows(ctx, rows.d2, p2.data(), n) != INQ_OK) {
            set_err(errbuf, errcap, std::string("device call failed: ") + inq_last_error(ctx));
            return INQ_EXIT_ERROR;
        }
    }

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


static std::mutex g_last_mu;
static PartStats g_last_stats;
void publish_last_stats(const PartStats &s) {
    std::lock_guard<std::mutex> g(g_last_mu);
    g_last_stats = s;
}
PartStats last_stats() {
    std::lock_guard<std::mutex> g(g_last_mu);
    return g_last_stats;
}

int write_rows(uint64_t threads, const std::vector<RepeatInterval> &targets, const std::string &sample, const double *p1,
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

}  // namespace inqhost

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

}  // extern "C"

namespace inqhost {

// The targets in file order (contig of the BAM header, start, end, position in the list) and `world` + 1 cut points into
// that order, so that every part needs about the same number of compressed BAM bytes: cost of a target = bytes between its
// scan start in the .bai's linear index and the next target's, capped so that one far-away locus does not own a contig.
int partition_prepared(Prepared &P, uint64_t world, uint32_t *order, uint64_t *cuts) {
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

}  // namespace inqhost

extern "C" {

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


struct inq_run {
    std::unique_ptr<OwnedArgs> args;
    Prepared P;
    // inq_run_rows_device: the context outlives the call (its rows are read from device memory afterwards)
    std::unique_ptr<AsyncCtx> actx;
    // inq_session_run_open: context, span buffers and BED cache are the session's (which outlives the run)
    inq_session *sess = nullptr;
    double *d1 = nullptr, *d2 = nullptr;
    uint64_t dcap = 0;
    AsyncCtx *ctx_of_rows() { return sess ? &sess->actx : actx.get(); }
    ~inq_run() {
        AsyncCtx *c = ctx_of_rows();
        if (d1 && c && c->wait()) inq_dev_free_rows(c->ctx, d1);  // (d2 points into the same allocation)
    }
};

static int inq_run_open_impl(const inq_call_args_t *args, inq_run_t **out, char *errbuf, size_t errcap, inq_session *sess = nullptr) {
    if (!out || !args) return INQ_EXIT_ERROR;
    *out = nullptr;
    std::unique_ptr<inq_run> R(new inq_run());
    R->args.reset(new OwnedArgs(*args));
    R->sess = sess;
    if (sess) R->args->a.device = sess->device;
    std::string msg;
    int rc = prepare(&R->args->a, R->P, msg, sess ? &sess->bed_cache : nullptr);
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
int inq_session_run_open(inq_session_t *s, const inq_call_args_t *args, inq_run_t **out, char *errbuf, size_t errcap) {
    if (!s) {
        set_err(errbuf, errcap, "null session");
        return INQ_EXIT_ERROR;
    }
    INQ_GUARD(inq_run_open_impl(args, out, errbuf, errcap, s), errbuf, errcap)
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
    RowsOut ro;
    ro.idx = target_index, ro.n = n_index, ro.p1 = phase1, ro.p2 = phase2, ro.active = true;
    if (r->sess) {  // the session's context and span buffers: nothing is made or torn down per call
        SessionHooks hooks;
        hooks.pool = &r->sess->pool;
        const bool keep_leak = r->sess->actx.leak;
        const int rc = genotype_prepared(&r->args->a, r->sess->actx, r->P, -1, errbuf, errcap, ro, t_start, hooks);
        r->sess->actx.leak = keep_leak;  // the context belongs to the session
        return rc;
    }
    AsyncCtx actx;
    actx.start(r->args->a.device);
    return genotype_prepared(&r->args->a, actx, r->P, -1, errbuf, errcap, ro, t_start);
}
int inq_run_rows(inq_run_t *r, const uint32_t *target_index, uint64_t n_index, double *phase1, double *phase2, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_run_rows_impl(r, target_index, n_index, phase1, phase2, errbuf, errcap), errbuf, errcap)
}
static int inq_run_rows_device_impl(inq_run_t *r, const uint32_t *target_index, uint64_t n_index, uint64_t width, void **d_phase1, void **d_phase2,
                                    char *errbuf, size_t errcap) {
    if (!r || !d_phase1 || !d_phase2 || width < n_index || (n_index && !target_index)) {
        set_err(errbuf, errcap, "null argument, or width below the number of targets");
        return INQ_EXIT_ERROR;
    }
    *d_phase1 = *d_phase2 = nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    if (!r->sess && !r->actx) {
        r->actx.reset(new AsyncCtx());
        r->actx->start(r->args->a.device);
    }
    AsyncCtx &actx = *r->ctx_of_rows();
    if (!actx.wait()) {
        set_err(errbuf, errcap, ctx_failure_message(actx));
        return INQ_EXIT_ERROR;
    }
    // one array of 2 x width: phase1 row, phase2 row (one collective's buffer); made afresh per call - every entry starts as NaN, a
    // device allocation costs microseconds (tools/alloc_probe.hip)
    const uint64_t w = std::max<uint64_t>(width, 1);
    inq_dev_free_rows(actx.ctx, r->d1);
    r->d1 = r->d2 = nullptr, r->dcap = 0;
    if (inq_dev_alloc_rows(actx.ctx, 2 * w, &r->d1) != INQ_OK) {
        set_err(errbuf, errcap, "cannot allocate the device row arrays");
        return INQ_EXIT_ERROR;
    }
    r->d2 = r->d1 + w, r->dcap = w;
    RowsOut ro;
    ro.idx = target_index, ro.n = n_index, ro.active = true, ro.d1 = r->d1, ro.d2 = r->d2, ro.dcap = r->dcap;
    SessionHooks hooks;
    if (r->sess) hooks.pool = &r->sess->pool;
    const bool keep_leak = actx.leak;
    const int rc = genotype_prepared(&r->args->a, actx, r->P, -1, errbuf, errcap, ro, t_start, hooks);
    actx.leak = keep_leak;  // the context belongs to the run (or its session)
    if (rc != INQ_EXIT_OK) return rc;
    *d_phase1 = r->d1, *d_phase2 = r->d2;
    return INQ_EXIT_OK;
}
int inq_run_rows_device(inq_run_t *r, const uint32_t *target_index, uint64_t n_index, uint64_t width, void **d_phase1, void **d_phase2, char *errbuf,
                        size_t errcap) {
    INQ_GUARD(inq_run_rows_device_impl(r, target_index, n_index, width, d_phase1, d_phase2, errbuf, errcap), errbuf, errcap)
}
int inq_run_write_inq(inq_run_t *r, const double *phase1, const double *phase2, uint64_t n_rows, int out_fd, char *errbuf, size_t errcap) {
    if (!r || n_rows != r->P.targets.size() || (n_rows && (!phase1 || !phase2))) {
        set_err(errbuf, errcap, "row count does not match the target list");
        return INQ_EXIT_ERROR;
    }
    INQ_GUARD(write_rows(r->args->a.threads, r->P.targets, r->P.sample, phase1, phase2, out_fd, errbuf, errcap), errbuf, errcap)
}
void inq_run_close(inq_run_t *r) { delete r; }

}  // extern "C"
