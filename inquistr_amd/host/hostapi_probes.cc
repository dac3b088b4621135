// hostapi_probes.cc - small entry points of include/inquistr_host.h that expose one host-side rule each to tests and bindings:
// row / header text, sample name, human sort, region parsing, index lookups, the span plan.
#include "driver_internal.h"

using namespace inqhost;

extern "C" {

uint64_t inq_host_iopool_selftest(int n_threads, int numa_node, int pin, uint64_t n_jobs, uint64_t rounds) {
    IoPool pool(n_threads, numa_node, pin != 0);
    std::atomic<uint64_t> sum{0};
    for (uint64_t r = 0; r < rounds; ++r) pool.run((size_t)n_jobs, [&](size_t k) { sum.fetch_add((uint64_t)k + 1u, std::memory_order_relaxed); });
    return sum.load();
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
