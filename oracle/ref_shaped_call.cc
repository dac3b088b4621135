// ref_shaped_call.cc — the reference's CONTROL FLOW around the CPU oracle: a CPU-only `inquiSTR call`
// shaped like src/call.rs:76-159.  TEST / MEASUREMENT INFRASTRUCTURE ONLY (never linked into the
// product); it is (i) an end-to-end cross-check of the product CLI on large BAMs and (ii) the
// CPU baseline of BASELINE.md §3, labelled "CPU restatement of the reference, not the Rust binary".
//
//   mode A  faithful -t N: worker pool pulling loci from one shared iterator (par_bridge, :115-118);
//           EVERY locus re-opens the BAM, re-parses the header and reloads the .bai (:217), then
//           index-fetches.  Thread count = what the caller passes (the reference ends up on rayon's
//           global pool = all cores, because the pool it builds is dropped, :104-107).
//   mode B  charitable: one reader per worker, index fetch per locus.
//   mode C  -t 1: one reader, BED order (:146-157).
// BAM/BAI decoding uses the repo's own reader (inquistr_amd/host/bam_reader.*); the arithmetic is
// oracle/inq_oracle.c.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../inquistr_amd/host/bam_reader.h"
#include "../inquistr_amd/host/inq_text.h"
#include "../inquistr_amd/host/targets.h"
#include "inq_oracle.h"

using namespace inqhost;

struct Rec {
    orc_record_t r;
    std::vector<uint32_t> cigar;
    std::string sa;
};

// bam.fetch((tid, beg, end)) + rc_records(): every record the index query yields, full decode
static bool fetch_records(BamFile &bam, int tid, uint32_t beg, uint32_t end, std::vector<Rec> &out, std::string *err) {
    out.clear();
    auto chunks = bam.index().query(tid, beg, end);
    BamRec rec;
    for (auto &c : chunks) {
        if (!bam.seek(c.first, err)) return false;
        for (;;) {
            int rc = bam.next(rec, err);
            if (rc < 0) return false;
            if (rc == 0) break;
            if (rec.voffset >= c.second) break;  // chunk exhausted (the next chunk starts at or after here)
            if (rec.tid != tid || rec.pos >= (int64_t)end) {
                if (rec.tid == tid || rec.tid > tid || rec.tid < 0) return true;  // past the region: iterator done
                continue;
            }
            if (bam_endpos(rec) > (int64_t)beg) {
                out.emplace_back();
                Rec &R = out.back();
                R.cigar.assign(rec.cigar, rec.cigar + rec.n_cigar);
                if (rec.sa_type == 'Z') R.sa = rec.sa;
                std::memset(&R.r, 0, sizeof R.r);
                R.r.tid = rec.tid;
                R.r.pos = rec.pos;
                R.r.flag = rec.flag;
                R.r.mapq = rec.mapq;
                R.r.n_cigar = rec.n_cigar;
                R.r.hp_type = rec.hp_type;
                R.r.hp_value = rec.hp_value;
                R.r.sa_type = rec.sa_type;
                R.r.is2d_given = -1;
            }
        }
    }
    return true;
}

int main(int argc, char **argv) {
    if (argc < 8) {
        std::fprintf(stderr, "usage: ref_shaped_call <bam> <bed> <mode A|B|C> <threads> <unphased 0|1> <minlen> <support> [sample]\n");
        return 2;
    }
    const std::string bamp = argv[1], bed = argv[2];
    const char mode = argv[3][0];
    int threads = std::max(1, atoi(argv[4]));
    const bool unphased = atoi(argv[5]) != 0;
    const uint32_t minlen = (uint32_t)atoi(argv[6]);
    const size_t support = (size_t)atoi(argv[7]);
    const std::string sample = argc > 8 ? argv[8] : sample_name_from_path(bamp);
    if (mode == 'C') threads = 1;
    auto t0 = std::chrono::steady_clock::now();

    std::string err;
    BamFile first(1);
    if (!first.open(bamp, &err)) {
        std::fprintf(stderr, "Error opening local BAM: %s\n", err.c_str());
        return 101;
    }
    auto lengths = first.sq_lengths(&err);
    TargetsResult tr = targets_from_bed(bed, lengths);
    if (tr.panicked) {
        std::fprintf(stderr, "%s\n", tr.message.c_str());
        return 101;
    }
    const size_t n = tr.data.size();
    std::vector<double> p1(n, NAN), p2(n, NAN);
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&](int /*wid*/) {
        std::unique_ptr<BamFile> mine;
        if (mode != 'A') {
            mine.reset(new BamFile(1));
            std::string e;
            if (!mine->open(bamp, &e)) {
                failed = 101;
                return;
            }
        }
        std::vector<Rec> recs;
        std::vector<orc_record_t> view;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n || failed) return;
            const RepeatInterval &t = tr.data[i];
            std::unique_ptr<BamFile> per_locus;
            BamFile *bam = mine.get();
            std::string e;
            if (mode == 'A') {  // get_bam_reader() per locus, src/call.rs:217
                per_locus.reset(new BamFile(1));
                if (!per_locus->open(bamp, &e)) {
                    failed = 101;
                    return;
                }
                bam = per_locus.get();
            }
            if (t.start < 10) {
                failed = 101;
                return;
            }
            const int tid = bam->tid(t.chrom);
            const uint32_t se = t.start - 10, ee = t.end + 10;
            if (!fetch_records(*bam, tid, se, ee, recs, &e)) {
                failed = 101;
                return;
            }
            view.resize(recs.size());
            for (size_t k = 0; k < recs.size(); ++k) {
                view[k] = recs[k].r;
                view[k].cigar = recs[k].cigar.data();
                view[k].sa = recs[k].r.sa_type == 'Z' ? recs[k].sa.c_str() : nullptr;
            }
            int panic, tie = 0;
            if (unphased)
                panic = orc_genotype_repeat_unphased(view.data(), view.size(), tid, t.start, t.end, minlen, support,
                                                     &p1[i], &p2[i], &tie);
            else
                panic = orc_genotype_repeat_phased(view.data(), view.size(), tid, t.start, t.end, minlen, support, &p1[i],
                                                   &p2[i]);
            if (panic) {
                failed = 101;
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int w = 1; w < threads; ++w) pool.emplace_back(work, w);
    work(0);
    for (auto &th : pool) th.join();
    if (failed) {
        std::fprintf(stderr, "panicked\n");
        return failed;
    }
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    if (mode != 'C' && threads > 1)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            int c = human_compare(tr.data[x].chrom, tr.data[y].chrom);
            if (c) return c < 0;
            return tr.data[x].start < tr.data[y].start;
        });
    std::string text = format_header(sample) + "\n";
    for (uint32_t i : order) text += format_row(tr.data[i].chrom, tr.data[i].start, tr.data[i].end, p1[i], p2[i]) + "\n";
    std::fwrite(text.data(), 1, text.size(), stdout);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "ref_shaped mode=%c threads=%d loci=%zu seconds=%.3f loci_per_s=%.1f\n", mode, threads, n, dt, n / dt);
    return 0;
}
