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
// Shares NO code with the product: BGZF / BAM / BAI decoding is oracle/minibam.h (a second reader, written from the SAM
// specification on plain zlib), the arithmetic and the text are oracle/inq_oracle.c, the BED is read right here.  A `.inq`
// from this program that equals the product's therefore cross-checks the product's decoders (host C++ reader, GPU inflate +
// record scan) as well as its kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "inq_oracle.h"
#include "minibam.h"

struct Target {
    std::string chrom;
    uint32_t start, end;
};

// RepeatIntervalIterator::from_bed (src/repeats.rs:30-45, 87-115) for well-formed BEDs: tab-separated, `#` comments and empty
// lines skipped ([3P] rust-bio / csv), columns 2-3 unsigned; end < start, an unknown contig or end >= LN panic.
static bool read_bed(const std::string &path, const minibam::Reader &bam, std::vector<Target> &out, std::string *err) {
    std::ifstream f(path);
    if (!f) return *err = "cannot open " + path, false;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        const size_t t1 = line.find('\t'), t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1);
        if (t2 == std::string::npos) return *err = "BED line with fewer than 3 fields", false;
        const size_t t3 = line.find('\t', t2 + 1);
        Target t;
        t.chrom = line.substr(0, t1);
        char *e1 = nullptr, *e2 = nullptr;
        const std::string a = line.substr(t1 + 1, t2 - t1 - 1), b2 = line.substr(t2 + 1, t3 == std::string::npos ? std::string::npos : t3 - t2 - 1);
        const unsigned long long s0 = std::strtoull(a.c_str(), &e1, 10), e0 = std::strtoull(b2.c_str(), &e2, 10);
        if (a.empty() || b2.empty() || *e1 || *e2 || s0 > 0xffffffffull || e0 > 0xffffffffull) return *err = "BED coordinates do not parse", false;
        t.start = (uint32_t)s0, t.end = (uint32_t)e0;
        const int tid = bam.tid(t.chrom);
        if (orc_check_interval(t.start, t.end, tid < 0 ? -1 : bam.refs[tid].second) != ORC_OK) return *err = "interval outside the contigs of the BAM header", false;
        out.push_back(t);
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
    std::string sample;
    if (argc > 8) sample = argv[8];
    else {
        char sb[1024];
        orc_sample_name(bamp.c_str(), sb, sizeof sb);
        sample = sb;
    }
    if (mode == 'C') threads = 1;
    auto t0 = std::chrono::steady_clock::now();

    std::string err;
    minibam::Reader first;
    if (!first.open(bamp)) {
        std::fprintf(stderr, "Error opening local BAM (or its .bai): %s\n", bamp.c_str());
        return 101;
    }
    std::vector<Target> targets;
    if (!read_bed(bed, first, targets, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 101;
    }
    const size_t n = targets.size();
    std::vector<double> p1(n, NAN), p2(n, NAN);
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&](int /*wid*/) {
        std::unique_ptr<minibam::Reader> mine;
        if (mode != 'A') {
            mine.reset(new minibam::Reader());
            if (!mine->open(bamp)) {
                failed = 101;
                return;
            }
        }
        std::vector<minibam::Record> recs;
        std::vector<orc_record_t> view;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n || failed) return;
            const Target &t = targets[i];
            std::unique_ptr<minibam::Reader> per_locus;
            minibam::Reader *bam = mine.get();
            if (mode == 'A') {  // get_bam_reader() per locus, src/call.rs:217
                per_locus.reset(new minibam::Reader());
                if (!per_locus->open(bamp)) {
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
            if (tid < 0 || !bam->fetch(tid, se, ee, recs)) {  // bam.fetch() + rc_records(), :288,294,338,345
                failed = 101;
                return;
            }
            view.resize(recs.size());
            for (size_t k = 0; k < recs.size(); ++k) {
                orc_record_t &v = view[k];
                std::memset(&v, 0, sizeof v);
                v.tid = recs[k].tid;
                v.pos = recs[k].pos;
                v.flag = recs[k].flag;
                v.mapq = recs[k].mapq;
                v.n_cigar = (uint32_t)recs[k].cigar.size();
                v.cigar = recs[k].cigar.data();
                v.hp_type = recs[k].hp_type;
                v.hp_value = recs[k].hp_value;
                v.sa_type = recs[k].sa_type;
                v.sa = recs[k].sa_type == 'Z' ? recs[k].sa.c_str() : nullptr;
                v.is2d_given = -1;
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
            int c = orc_human_compare(targets[x].chrom.c_str(), targets[y].chrom.c_str());
            if (c) return c < 0;
            return targets[x].start < targets[y].start;
        });
    char line[4096];
    orc_format_header(sample.c_str(), line, sizeof line);
    std::string text = std::string(line) + "\n";
    for (uint32_t i : order) {
        orc_format_row(targets[i].chrom.c_str(), targets[i].start, targets[i].end, p1[i], p2[i], line, sizeof line);
        text += line;
        text += '\n';
    }
    std::fwrite(text.data(), 1, text.size(), stdout);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "ref_shaped mode=%c threads=%d loci=%zu seconds=%.3f loci_per_s=%.1f\n", mode, threads, n, dt, n / dt);
    return 0;
}
