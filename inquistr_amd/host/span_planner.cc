#include "span_planner.h"

#include <chrono>

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <thread>

namespace inqhost {

static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ---------------------------------------------------------------- reader threads
namespace {

// "0-63,128-191" -> CPUs
std::vector<int> parse_cpulist(const char *s) {
    std::vector<int> out;
    for (const char *p = s; *p;) {
        char *q;
        long a = std::strtol(p, &q, 10), b = a;
        if (q == p) break;
        if (*q == '-') b = std::strtol(q + 1, &q, 10);
        for (long k = a; k <= b && k < CPU_SETSIZE; ++k) out.push_back((int)k);
        if (*q != ',') break;
        p = q + 1;
    }
    return out;
}

bool read_line(const char *path, char *buf, size_t cap) {
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    const bool ok = std::fgets(buf, (int)cap, f) != nullptr;
    std::fclose(f);
    return ok;
}

// the CPUs this process may use on `node` (all of them for node < 0), grouped by the L3 cache they share
std::vector<std::vector<int>> l3_groups(int node) {
    cpu_set_t have;
    CPU_ZERO(&have);
    if (sched_getaffinity(0, sizeof have, &have) != 0) return {};
    std::vector<int> cpus;
    char path[160], buf[4096];
    if (node >= 0) {
        std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        if (read_line(path, buf, sizeof buf)) cpus = parse_cpulist(buf);
    }
    if (cpus.empty())
        for (int k = 0; k < CPU_SETSIZE; ++k)
            if (CPU_ISSET(k, &have)) cpus.push_back(k);
    std::map<long, std::vector<int>> by_l3;
    for (int k : cpus) {
        if (!CPU_ISSET(k, &have)) continue;  // never beyond what the process was given
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/id", k);
        long id = -1;
        if (read_line(path, buf, sizeof buf)) id = std::strtol(buf, nullptr, 10);
        by_l3[id].push_back(k);
    }
    std::vector<std::vector<int>> out;
    for (auto &kv : by_l3) out.push_back(std::move(kv.second));
    return out;
}

}  // namespace

IoPool::IoPool(int n_threads, int numa_node, bool pin, int group_offset) {
    const int n = std::max(n_threads, 1);
    std::vector<std::vector<int>> groups;
    if (pin) groups = l3_groups(numa_node);
    if (groups.size() < 2) groups.clear();  // one domain (or no topology to read): nothing to spread over
    layout_ = std::to_string(n) + " reader threads";
    if (!groups.empty()) layout_ += ", bound in turn to the " + std::to_string(groups.size()) + " L3 domains of " + (numa_node >= 0 ? "NUMA node " + std::to_string(numa_node) : std::string("the machine"));
    else layout_ += pin ? ", not bound (one L3 domain or no topology in sysfs)" : ", not bound (INQ_IO_PIN=0)";
    cpu_at_.assign((size_t)n, -1);
    for (int i = 1; i < n; ++i) {
        workers_.emplace_back([this, i] { work(i); });
        if (!groups.empty()) {
            const std::vector<int> &g = groups[((size_t)i + (size_t)std::max(group_offset, 0) * (size_t)n) % groups.size()];
            cpu_set_t set;
            CPU_ZERO(&set);
            for (int k : g) CPU_SET(k, &set);
            (void)pthread_setaffinity_np(workers_.back().native_handle(), sizeof set, &set);  // best effort: placement only
        }
    }
}

IoPool::~IoPool() {
    {
        std::lock_guard<std::mutex> g(mu_);
        stop_ = true;
    }
    cv_go_.notify_all();
    for (auto &t : workers_) t.join();
}

std::string IoPool::last_cpus() const {
    std::string s;
    for (int c : cpu_at_) s += (s.empty() ? "" : ",") + std::to_string(c);
    return s;
}

void IoPool::work(int id) {
    struct Lap {
        uint64_t t0 = thread_cpu_us();
        ~Lap() { cpu_by_kind().readers_us.fetch_add(thread_cpu_us() - t0, std::memory_order_relaxed); }
    } lap;
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(size_t)> *fn;
        size_t n;
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_go_.wait(g, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            fn = fn_;
            n = n_jobs_;
        }
        cpu_at_[(size_t)id] = sched_getcpu();
        for (;;) {
            const size_t k = next_.fetch_add(1);
            if (k >= n) break;
            (*fn)(k);
        }
        std::lock_guard<std::mutex> g(mu_);
        if (--busy_ == 0) cv_done_.notify_one();
    }
}

void IoPool::run(size_t n_jobs, const std::function<void(size_t)> &fn) {
    if (n_jobs == 0) return;
    {
        std::lock_guard<std::mutex> g(mu_);
        fn_ = &fn;
        n_jobs_ = n_jobs;
        next_.store(0);
        busy_ = (int)workers_.size();
        ++generation_;
    }
    cv_go_.notify_all();
    cpu_at_[0] = sched_getcpu();
    for (;;) {  // the caller takes jobs too
        const size_t k = next_.fetch_add(1);
        if (k >= n_jobs) break;
        fn(k);
    }
    std::unique_lock<std::mutex> g(mu_);
    cv_done_.wait(g, [&] { return busy_ == 0; });
    fn_ = nullptr;
}

// ---------------------------------------------------------------- .bai views
const BaiAnchors::PerRef &BaiAnchors::ref(int tid) {
    std::lock_guard<std::mutex> g(mu_);  // built once, never changed afterwards: readers need no lock behind this call
    PerRef &P = refs_[tid];
    if (P.built) return P;
    const BaiRef &R = idx_.refs[tid];
    // UCSC binning (.bai: min_shift 14, depth 5; .csi: as the file says): first bin id and width (as a shift) of every level
    for (const auto &kv : R.bins) {
        const uint32_t bin = kv.first;
        if (bin >= idx_.level_first(idx_.depth + 1) || kv.second.empty()) continue;
        int l = 0;
        while (bin >= idx_.level_first(l + 1)) ++l;
        const int64_t start = (int64_t)(bin - idx_.level_first(l)) << idx_.level_shift(l);
        uint64_t beg = kv.second[0].first;
        for (const auto &c : kv.second) {
            beg = std::min(beg, c.first);
            P.anchors.push_back(c.first);  // a chunk begins at a record
        }
        P.bins.emplace_back(start, beg);
    }
    for (uint64_t v : R.ioffset)
        if (v) P.anchors.push_back(v);  // the first record overlapping a 16 kb window
    for (const auto &kv : R.loff)
        if (kv.second) P.anchors.push_back(kv.second);  // .csi: the same for the first window of a bin
    std::sort(P.anchors.begin(), P.anchors.end());
    P.anchors.erase(std::unique(P.anchors.begin(), P.anchors.end()), P.anchors.end());
    std::sort(P.bins.begin(), P.bins.end());
    for (size_t i = P.bins.size(); i-- > 1;) P.bins[i - 1].second = std::min(P.bins[i - 1].second, P.bins[i].second);
    P.built = true;
    return P;
}

uint64_t BaiAnchors::limit_after(int tid, int64_t x) {
    const PerRef &P = ref(tid);
    const BaiRef &R = idx_.refs[tid];
    // records of a bin that starts at or behind x have pos >= x; the file is coordinate-sorted, so every
    // record with pos < x lies in front of the first of them
    auto it = std::lower_bound(P.bins.begin(), P.bins.end(), std::make_pair(x, (uint64_t)0));
    if (it != P.bins.end()) return it->second;
    return R.max_offset;  // end of the contig's last chunk
}

// ---------------------------------------------------------------- planner
SpanPlanner::SpanPlanner(const BamFile &bam, const std::vector<RepeatInterval> &targets, uint64_t max_comp_bytes)
    : bam_(bam), anch_(bam.index()), max_comp_(max_comp_bytes ? max_comp_bytes : (2048ull << 20)) {
    for (uint32_t i = 0; i < targets.size(); ++i) {
        const int tid = bam_.tid(targets[i].chrom);
        if (tid < 0 || (size_t)tid >= bam_.index().refs.size()) continue;  // no index entry: nothing to fetch
        loci_.push_back({tid, targets[i].start, targets[i].end, i});
    }
    if (const char *e = std::getenv("INQ_SPAN_GAP_BYTES")) gap_ = (uint64_t)std::strtoull(e, nullptr, 10);  // tests: force segments
    std::stable_sort(loci_.begin(), loci_.end(), [](const Locus &a, const Locus &b) {
        if (a.tid != b.tid) return a.tid < b.tid;
        return a.start < b.start;
    });
}

bool SpanPlanner::next(SpanPlan &out) {
    const uint64_t kGap = gap_;  // compressed bytes worth reading through rather than opening a segment
    const BaiIndex &idx = bam_.index();
    out = SpanPlan();
    uint64_t bytes = 0;  // compressed bytes of the closed segments
    auto seg_bytes = [](const Segment &g) { return (g.vo_limit >> 16) - (g.vo_begin >> 16) + 65536; };
    for (; j_ < loci_.size(); ++j_) {
        const Locus &L = loci_[j_];
        // src/call.rs:285-286,335-336 (start >= 10 was checked by the driver)
        const uint64_t vo = idx.scan_start(L.tid, (int64_t)L.start - 10);
        const uint64_t lim = vo ? anch_.limit_after(L.tid, (int64_t)L.end + 10) : 0;
        if (vo == 0 || lim <= vo) continue;  // no record can overlap this window: the row stays NaN
        bool extend = false;
        if (!out.segs.empty()) {
            const Segment &S = out.segs.back();
            if (vo < S.vo_begin) break;  // index not monotone: start over with a new span
            extend = (vo >> 16) <= (S.vo_limit >> 16) + kGap;
            Segment grown = S;
            if (extend) grown.vo_limit = std::max(S.vo_limit, lim);
            const uint64_t after = extend ? bytes + seg_bytes(grown) : bytes + seg_bytes(S) + (lim >> 16) - (vo >> 16) + 65536;
            if (after > max_comp_) break;  // span full (it holds at least one locus)
        }
        if (extend) {
            Segment &S = out.segs.back();
            S.vo_limit = std::max(S.vo_limit, lim);
            S.tid_last = L.tid;
        } else {
            if (!out.segs.empty()) bytes += seg_bytes(out.segs.back());
            Segment g;
            g.vo_begin = vo;
            g.vo_limit = lim;
            g.tid_first = g.tid_last = L.tid;
            out.segs.push_back(g);
        }
        out.locus_index.push_back(L.index);
        out.locus_tid.push_back(L.tid);
        out.locus_start.push_back(L.start);
        out.locus_end.push_back(L.end);
    }
    return !out.segs.empty();
}

// ---------------------------------------------------------------- loader
SpanLoader::~SpanLoader() {
    if (fd_ >= 0) ::close(fd_);
}

bool SpanLoader::open(const std::string &path, std::string *err) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    struct stat st;
    if (fd_ < 0 || ::fstat(fd_, &st) != 0) {
        if (err) *err = "cannot open " + path;
        return false;
    }
    size_ = (uint64_t)st.st_size;
    return true;
}

static bool pread_all(int fd, uint8_t *dst, uint64_t off, uint64_t n) {
    while (n) {
        ssize_t g = ::pread(fd, dst, (size_t)std::min<uint64_t>(n, 1ull << 30), (off_t)off);
        if (g <= 0) return false;
        dst += g;
        off += (uint64_t)g;
        n -= (uint64_t)g;
    }
    return true;
}

// size of the BGZF block whose header starts at h (>= 18 readable bytes), 0 if it is not one
static uint32_t bgzf_block_size(const uint8_t *h, size_t avail, uint32_t *head_len) {
    if (avail < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
    const uint32_t xlen = le16(h + 10);
    if (12 + (size_t)xlen > avail) return 0;
    int bsize = -1;
    for (uint32_t i = 0; i + 4 <= xlen;) {
        const uint8_t *f = h + 12 + i;
        const uint32_t slen = le16(f + 2);
        if (f[0] == 'B' && f[1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = le16(f + 4);
        i += 4 + slen;
    }
    if (bsize < 0) return 0;
    *head_len = 12 + xlen;
    return (uint32_t)bsize + 1;
}

bool SpanLoader::extent(const Segment &p, uint64_t *begin, uint64_t *end, std::string *err) const {
    *begin = p.vo_begin >> 16;
    const uint64_t lb = p.vo_limit >> 16;
    if (*begin >= size_) {
        if (err) *err = "index points behind the end of the BAM file";
        return false;
    }
    if (lb >= size_) {
        *end = size_;
        return true;
    }
    if ((p.vo_limit & 0xffff) == 0) {  // the limit record opens its block: the block itself is not needed
        *end = lb;
        return true;
    }
    uint8_t h[64];
    const size_t want = (size_t)std::min<uint64_t>(sizeof h, size_ - lb);
    uint32_t head = 0;
    if (!pread_all(fd_, h, lb, want)) {
        if (err) *err = "read error in BAM file";
        return false;
    }
    const uint32_t bs = bgzf_block_size(h, want, &head);
    if (!bs) {
        if (err) *err = "index offset " + std::to_string(lb) + " is not a BGZF block";
        return false;
    }
    *end = std::min<uint64_t>(lb + bs, size_);
    return true;
}

bool SpanLoader::total_bytes(const SpanPlan &p, uint64_t *bytes, std::string *err) const {
    *bytes = 0;
    for (const Segment &g : p.segs) {
        uint64_t b, e;
        if (!extent(g, &b, &e, err)) return false;
        *bytes += e - b;
    }
    return true;
}

bool SpanLoader::load(const SpanPlan &p, BaiAnchors &anch, uint8_t *buf, int n_threads, SpanData &out, std::string *err, IoPool *pool) const {
    out.blocks.clear();
    out.anchors.clear();
    out.anchor_stop.clear();
    out.comp_bytes = 0;
    out.file_begin = p.segs.empty() ? 0 : p.segs[0].vo_begin >> 16;
    // 1. where every segment goes in buf
    struct Piece {
        uint64_t begin, end, at;
    };
    std::vector<Piece> pieces;
    for (const Segment &g : p.segs) {
        uint64_t b, e;
        if (!extent(g, &b, &e, err)) return false;
        if (!pieces.empty() && b < pieces.back().end) {
            if (err) *err = "index offsets of neighbouring loci overlap out of order";
            return false;
        }
        pieces.push_back({b, e, out.comp_bytes});
        out.comp_bytes += e - b;
    }
    // 2. parallel pread + block tables.  A job = a run of whole BGZF blocks: it starts at the piece's begin or at an index anchor inside
    // the piece (every virtual offset of the index names a block start), so the thread that copied the bytes also hops through
    // their block headers - while the tail of what it copied is still in its cache, and side by side with the other jobs.  (Round 4
    // hopped through the whole span on the loader thread behind the copy: 1 ms per 268 MB span, a quarter of a 16-reader load.)
    const auto t_read0 = std::chrono::steady_clock::now();
    struct Job {
        uint64_t off, n, at;
        size_t piece;
        std::vector<inq_bgzf_block_t> blocks;  // out_off relative to the job's first inflated byte
        std::vector<uint64_t> starts;          // file offset of every block
        std::string err;
    };
    std::vector<Job> jobs;
    {
        if (pool) n_threads = pool->threads();
        uint64_t chunk = std::max<uint64_t>(4ull << 20, out.comp_bytes / (uint64_t)std::max(n_threads, 1) / 4 + 1);
        if (const char *e = std::getenv("INQ_SPAN_JOB_BYTES")) chunk = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));  // tests: a job per anchor
        for (size_t si = 0; si < pieces.size(); ++si) {
            const Piece &pc = pieces[si];
            const Segment &g = p.segs[si];
            uint64_t at = pc.begin;
            auto emit = [&](uint64_t to) {
                Job j;
                j.off = at, j.n = to - at, j.at = pc.at + (at - pc.begin), j.piece = si;
                jobs.push_back(std::move(j));
                at = to;
            };
            for (int tid = g.tid_first; tid <= g.tid_last; ++tid) {
                const auto &A = anch.ref(tid).anchors;
                for (auto it = std::lower_bound(A.begin(), A.end(), g.vo_begin); it != A.end() && *it < g.vo_limit; ++it) {
                    const uint64_t co = *it >> 16;
                    if (co >= pc.end) break;
                    if (co >= at + chunk) emit(co);
                }
            }
            emit(pc.end);
        }
        std::atomic<bool> ok{true};
        auto run_job = [&](size_t k) {
            Job &j = jobs[k];
            if (!pread_all(fd_, buf + j.at, j.off, j.n)) {
                j.err = "read error in BAM file";
                ok = false;
                return;
            }
            uint64_t q = 0, uo_local = 0;
            j.blocks.reserve((size_t)(j.n / 16384 + 4));
            j.starts.reserve((size_t)(j.n / 16384 + 4));
            while (q < j.n) {
                uint32_t head = 0;
                const uint8_t *h = buf + j.at + q;
                const uint32_t bs = bgzf_block_size(h, (size_t)(j.n - q), &head);
                if (!bs || q + bs > j.n || bs < head + 8) {
                    j.err = "not a BGZF block at offset " + std::to_string(j.off + q);
                    ok = false;
                    return;
                }
                const uint32_t isize = le32(h + bs - 4);
                if (isize > 65536) {
                    j.err = "BGZF block with ISIZE > 64 KiB at offset " + std::to_string(j.off + q);
                    ok = false;
                    return;
                }
                inq_bgzf_block_t b;
                b.comp_off = j.at + q + head;
                b.comp_len = bs - head - 8;
                b.isize = isize;
                b.out_off = uo_local;
                j.blocks.push_back(b);
                j.starts.push_back(j.off + q);
                uo_local += isize;
                q += bs;
            }
        };
        if (pool) {
            pool->run(jobs.size(), run_job);
        } else {
            std::atomic<size_t> nextj{0};
            auto work = [&] {
                for (;;) {
                    const size_t k = nextj.fetch_add(1);
                    if (k >= jobs.size()) return;
                    run_job(k);
                }
            };
            const int nt = (int)std::min<size_t>((size_t)std::max(n_threads, 1), jobs.size());
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(work);
            work();
            for (auto &x : th) x.join();
        }
        if (!ok) {
            if (err) {
                *err = "read error in BAM file";
                for (const Job &j : jobs)
                    if (!j.err.empty()) {
                        *err = j.err;  // the first in file order
                        break;
                    }
            }
            return false;
        }
    }
    // 3. the jobs' tables joined in file order, and the anchors, segment by segment
    const auto t_read1 = std::chrono::steady_clock::now();
    out.ms_read = std::chrono::duration<double, std::milli>(t_read1 - t_read0).count();
    struct Lap {
        std::chrono::steady_clock::time_point t0;
        double *out;
        ~Lap() { *out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    } lap{t_read1, &out.ms_tables};
    {
        size_t nb = 0;
        for (const Job &j : jobs) nb += j.blocks.size();
        out.blocks.reserve(nb);
    }
    uint64_t uo = 0;
    size_t jk = 0;
    for (size_t si = 0; si < p.segs.size(); ++si) {
        const Segment &g = p.segs[si];
        const Piece &pc = pieces[si];
        const size_t first_block = out.blocks.size();
        std::vector<uint64_t> starts;  // file offset of every block of the segment
        for (; jk < jobs.size() && jobs[jk].piece == si; ++jk) {
            Job &j = jobs[jk];
            for (inq_bgzf_block_t b : j.blocks) {
                b.out_off += uo;
                out.blocks.push_back(b);
            }
            if (!j.blocks.empty()) uo = out.blocks.back().out_off + out.blocks.back().isize;
            starts.insert(starts.end(), j.starts.begin(), j.starts.end());
            std::vector<inq_bgzf_block_t>().swap(j.blocks);
            std::vector<uint64_t>().swap(j.starts);
        }
        const uint64_t seg_u_end = uo;
        // (the anchors come in ascending order, and so do the blocks: the search for an anchor's block starts where the last one
        // ended and usually ends there or a step further - a CIGAR-only file has 35 000 anchors per 100 MB, and a binary search for
        // each was 1.5 ms of the loader's 2.7 ms per span)
        size_t cursor = 0;
        auto map_vo = [&](uint64_t v, uint64_t *u) -> bool {
            const uint64_t co = v >> 16, within = v & 0xffff;
            if (co == pc.end && within == 0) {
                *u = seg_u_end;
                return true;
            }
            if (cursor >= starts.size() || starts[cursor] > co) cursor = 0;  // (an offset that goes backwards: start over)
            size_t steps = 0;
            while (cursor < starts.size() && starts[cursor] < co && steps < 8) ++cursor, ++steps;
            if (cursor < starts.size() && starts[cursor] < co) cursor = (size_t)(std::lower_bound(starts.begin() + (long)cursor, starts.end(), co) - starts.begin());
            if (cursor >= starts.size() || starts[cursor] != co) return false;
            const inq_bgzf_block_t &b = out.blocks[first_block + cursor];
            if (within > b.isize) return false;
            *u = b.out_off + within;
            return true;
        };
        const size_t first_anchor = out.anchors.size();
        uint64_t u0;
        if (!map_vo(g.vo_begin, &u0)) {
            if (err) *err = "index offset does not match the BGZF blocks of the file";
            return false;
        }
        out.anchors.push_back(u0);
        // the index offsets of every contig with records in the segment, in file order
        for (int tid = g.tid_first; tid <= g.tid_last; ++tid) {
            const auto &A = anch.ref(tid).anchors;
            auto lo = std::lower_bound(A.begin(), A.end(), g.vo_begin), hi = std::upper_bound(A.begin(), A.end(), g.vo_limit);
            for (auto it = lo; it != hi; ++it) {
                const uint64_t co = *it >> 16;
                if (co > pc.end || (co == pc.end && (*it & 0xffff) != 0)) break;
                if (co < pc.begin) continue;
                uint64_t u;
                if (!map_vo(*it, &u)) {
                    if (err) *err = "index offset does not match the BGZF blocks of the file";
                    return false;
                }
                if (u > out.anchors.back() && u < seg_u_end) out.anchors.push_back(u);  // (block, isize) == (next block, 0)
            }
        }
        for (size_t k = first_anchor; k < out.anchors.size(); ++k)
            out.anchor_stop.push_back(k + 1 < out.anchors.size() ? out.anchors[k + 1] : (seg_u_end | INQ_ANCHOR_SEGMENT_END));
    }
    return true;
}

}  // namespace inqhost
