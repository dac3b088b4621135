#include "span_planner.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <thread>

namespace inqhost {

static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ---------------------------------------------------------------- .bai views
const BaiAnchors::PerRef &BaiAnchors::ref(int tid) {
    PerRef &P = refs_[tid];
    if (P.built) return P;
    const BaiRef &R = idx_.refs[tid];
    // UCSC binning, min_shift 14, depth 5: first bin id and width (as a shift) of every level
    const int shifts[6] = {29, 26, 23, 20, 17, 14};
    const uint32_t firsts[7] = {0, 1, 9, 73, 585, 4681, 37449};
    for (const auto &kv : R.bins) {
        const uint32_t bin = kv.first;
        if (bin >= 37449 || kv.second.empty()) continue;
        int l = 0;
        while (bin >= firsts[l + 1]) ++l;
        const int64_t start = (int64_t)(bin - firsts[l]) << shifts[l];
        uint64_t beg = kv.second[0].first;
        for (const auto &c : kv.second) {
            beg = std::min(beg, c.first);
            P.anchors.push_back(c.first);  // a chunk begins at a record
        }
        P.bins.emplace_back(start, beg);
    }
    for (uint64_t v : R.ioffset)
        if (v) P.anchors.push_back(v);  // the first record overlapping a 16 kb window
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
    : bam_(bam), anch_(bam.index()), max_comp_(max_comp_bytes ? max_comp_bytes : (256ull << 20)) {
    std::vector<std::pair<int, uint32_t>> order;
    for (uint32_t i = 0; i < targets.size(); ++i) order.emplace_back(bam_.tid(targets[i].chrom), i);
    std::stable_sort(order.begin(), order.end(), [&](const auto &a, const auto &b) {
        if (a.first != b.first) return a.first < b.first;
        return targets[a.second].start < targets[b.second].start;
    });
    for (auto &o : order) {
        if (o.first < 0 || (size_t)o.first >= bam_.index().refs.size()) continue;  // no index entry: nothing to fetch
        const RepeatInterval &t = targets[o.second];
        if (groups_.empty() || groups_.back().tid != o.first) groups_.push_back({o.first, {}});
        groups_.back().loci.push_back({t.start, t.end, o.second});
    }
}

bool SpanPlanner::next(SpanPlan &out) {
    constexpr uint64_t kGap = 4ull << 20;  // compressed bytes worth reading through rather than starting a new span
    const BaiIndex &idx = bam_.index();
    for (; g_ < groups_.size(); ++g_, j_ = 0) {
        const Group &G = groups_[g_];
        while (j_ < G.loci.size()) {
            const Locus &L0 = G.loci[j_];
            // src/call.rs:285-286,335-336 (start >= 10 was checked by the driver)
            const uint64_t vo0 = idx.scan_start(G.tid, (int64_t)L0.start - 10);
            uint64_t lim = vo0 ? anch_.limit_after(G.tid, (int64_t)L0.end + 10) : 0;
            if (vo0 == 0 || lim <= vo0) {  // no record can overlap this window
                ++j_;
                continue;
            }
            out = SpanPlan();
            out.tid = G.tid;
            out.vo_begin = vo0;
            auto take = [&](const Locus &L) {
                out.locus_index.push_back(L.index);
                out.locus_start.push_back(L.start);
                out.locus_end.push_back(L.end);
            };
            take(L0);
            size_t j = j_ + 1;
            for (; j < G.loci.size(); ++j) {
                const Locus &L = G.loci[j];
                const uint64_t vj = idx.scan_start(G.tid, (int64_t)L.start - 10);
                if (vj == 0 || vj < vo0) break;
                const uint64_t lj = std::max(lim, anch_.limit_after(G.tid, (int64_t)L.end + 10));
                if ((vj >> 16) > (lim >> 16) + kGap) break;                 // jump the gap with a new span
                if ((lj >> 16) - (vo0 >> 16) > max_comp_) break;            // span full
                lim = lj;
                take(L);
            }
            out.vo_limit = lim;
            j_ = j;
            return true;
        }
    }
    return false;
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

bool SpanLoader::extent(const SpanPlan &p, uint64_t *begin, uint64_t *end, std::string *err) const {
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

bool SpanLoader::load(const SpanPlan &p, BaiAnchors &anch, uint64_t begin, uint64_t end, uint8_t *buf, int n_threads,
                      SpanData &out, std::string *err) const {
    out.blocks.clear();
    out.anchors.clear();
    out.comp_bytes = end - begin;
    out.file_begin = begin;
    const uint64_t n = end - begin;
    // parallel pread: the page cache (or the device underneath) serves several streams faster than one
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(n_threads, 1), n >> 22));
    bool ok = true;
    if (nt == 1) {
        ok = pread_all(fd_, buf, begin, n);
    } else {
        std::vector<std::thread> th;
        std::vector<char> res((size_t)nt, 1);
        for (int t = 0; t < nt; ++t) {
            const uint64_t lo = n * (uint64_t)t / (uint64_t)nt, hi = n * (uint64_t)(t + 1) / (uint64_t)nt;
            th.emplace_back([&, t, lo, hi] { res[(size_t)t] = pread_all(fd_, buf + lo, begin + lo, hi - lo) ? 1 : 0; });
        }
        for (auto &x : th) x.join();
        for (char r : res) ok = ok && r;
    }
    if (!ok) {
        if (err) *err = "read error in BAM file";
        return false;
    }
    // block table: hop from header to header
    std::vector<uint64_t> starts;  // file offset of every block
    uint64_t q = 0, uo = 0;
    while (q < n) {
        uint32_t head = 0;
        const uint32_t bs = bgzf_block_size(buf + q, (size_t)(n - q), &head);
        if (!bs || q + bs > n || bs < head + 8) {
            if (err) *err = "not a BGZF block at offset " + std::to_string(begin + q);
            return false;
        }
        const uint32_t isize = le32(buf + q + bs - 4);
        if (isize > 65536) {
            if (err) *err = "BGZF block with ISIZE > 64 KiB at offset " + std::to_string(begin + q);
            return false;
        }
        inq_bgzf_block_t b;
        b.comp_off = q + head;
        b.comp_len = bs - head - 8;
        b.isize = isize;
        b.out_off = uo;
        out.blocks.push_back(b);
        starts.push_back(begin + q);
        uo += isize;
        q += bs;
    }
    // anchors: the contig's index offsets inside [vo_begin, vo_limit], as offsets into the inflated bytes
    const auto &A = anch.ref(p.tid).anchors;
    auto lo = std::lower_bound(A.begin(), A.end(), p.vo_begin), hi = std::upper_bound(A.begin(), A.end(), p.vo_limit);
    out.anchors.reserve((size_t)(hi - lo) + 2);
    auto map_vo = [&](uint64_t v, uint64_t *u) -> bool {
        const uint64_t co = v >> 16, within = v & 0xffff;
        if (co == end && within == 0) {
            *u = uo;
            return true;
        }
        auto it = std::lower_bound(starts.begin(), starts.end(), co);
        if (it == starts.end() || *it != co) return false;
        const inq_bgzf_block_t &b = out.blocks[(size_t)(it - starts.begin())];
        if (within > b.isize) return false;
        *u = b.out_off + within;
        return true;
    };
    uint64_t u0;
    if (!map_vo(p.vo_begin, &u0)) {
        if (err) *err = "index offset does not match the BGZF blocks of the file";
        return false;
    }
    out.anchors.push_back(u0);
    for (auto it = lo; it != hi; ++it) {
        uint64_t u;
        if ((*it >> 16) >= end && !((*it >> 16) == end && (*it & 0xffff) == 0)) break;
        if (!map_vo(*it, &u)) {
            if (err) *err = "index offset does not match the BGZF blocks of the file";
            return false;
        }
        if (u > out.anchors.back()) out.anchors.push_back(u);  // (block, isize) and (next block, 0) name the same byte
    }
    return true;
}

}  // namespace inqhost
