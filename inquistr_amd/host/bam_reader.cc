#include "bam_reader.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace inqhost {

// (little-endian host: x86-64, the only host of a gfx950 box; one load each instead of byte-wise assembly - an index holds 10^5 - 10^6 of them)
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "little-endian host");
static inline uint16_t le16(const uint8_t *p) { uint16_t v; std::memcpy(&v, p, 2); return v; }
static inline uint32_t le32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
static inline uint64_t le64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// ---------------- BAI ----------------

void BinMap::seal() {
    // bin ids of a .bai are below 37 451: a counting sort; anything wider (.csi with a deep binning) goes through std::stable_sort
    uint32_t mx = 0;
    bool sorted = true;
    for (size_t i = 0; i < entries_.size(); ++i) {
        mx = std::max(mx, entries_[i].first);
        if (i && entries_[i].first < entries_[i - 1].first) sorted = false;
    }
    if (!sorted) {
        if (mx < (1u << 17)) {
            std::vector<uint32_t> start((size_t)mx + 2, 0u);
            for (const Entry &e : entries_) ++start[(size_t)e.first + 1];
            for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
            std::vector<Entry> out(entries_.size());
            for (const Entry &e : entries_) out[start[e.first]++] = e;
            entries_.swap(out);
        } else {
            std::stable_sort(entries_.begin(), entries_.end(), [](const Entry &a, const Entry &b) { return a.first < b.first; });
        }
    }
    fix();
}

void BinMap::fix() {
    for (Entry &e : entries_) e.second.b = chunks_.data() + e.at;
}

BinMap::const_iterator BinMap::lower_bound(uint32_t bin) const {
    return std::lower_bound(begin(), end(), bin, [](const Entry &e, uint32_t b) { return e.first < b; });
}

bool BaiIndex::load(const std::string &path, std::string *err) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open index " + path;
        return false;
    }
    std::vector<uint8_t> d;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) d.insert(d.end(), tmp, tmp + n);
    std::fclose(f);
    size_t p = 0;
    auto need = [&](size_t k) { return p + k <= d.size(); };
    if (!need(8) || std::memcmp(d.data(), "BAI\1", 4) != 0) {
        if (err) *err = "not a BAI file: " + path;
        return false;
    }
    p = 4;
    uint32_t n_ref = le32(&d[p]);
    p += 4;
    // every contig takes at least its n_bin and n_intv words: a count the file cannot hold is damage, and must be refused
    // BEFORE one BaiRef per claimed contig is built (a flipped n_ref of 2e8 would otherwise allocate tens of GB first)
    if ((uint64_t)n_ref > (d.size() - p) / 8) goto trunc;
    refs.assign(n_ref, BaiRef());
    for (uint32_t r = 0; r < n_ref; ++r) {
        BaiRef &R = refs[r];
        if (!need(4)) goto trunc;
        {
            uint32_t n_bin = le32(&d[p]);
            p += 4;
            bool first = true;
            for (uint32_t b = 0; b < n_bin; ++b) {
                if (!need(8)) goto trunc;
                uint32_t bin = le32(&d[p]);
                uint32_t n_chunk = le32(&d[p + 4]);
                p += 8;
                if (!need((size_t)n_chunk * 16)) goto trunc;
                if (bin == 37450 && n_chunk == 2) {  // htslib pseudo-bin: [file range], [mapped, unmapped]
                    R.has_meta = true;
                    R.n_mapped = le64(&d[p + 16]);
                    R.n_unmapped = le64(&d[p + 24]);
                } else {
                    BinMap::Chunk *v = R.bins.add(bin, n_chunk);
                    for (uint32_t c = 0; c < n_chunk; ++c) {
                        uint64_t beg = le64(&d[p + 16 * c]), end = le64(&d[p + 16 * c + 8]);
                        v[c] = BinMap::Chunk(beg, end);
                        if (first || beg < R.min_offset) R.min_offset = beg;
                        if (first || end > R.max_offset) R.max_offset = end;
                        first = false;
                    }
                }
                p += (size_t)n_chunk * 16;
            }
            R.bins.seal();
            if (!need(4)) goto trunc;
            uint32_t n_intv = le32(&d[p]);
            p += 4;
            if (!need((size_t)n_intv * 8)) goto trunc;
            R.ioffset.resize(n_intv);
            for (uint32_t i = 0; i < n_intv; ++i) R.ioffset[i] = le64(&d[p + 8 * i]);
            p += (size_t)n_intv * 8;
        }
    }
    if (need(8)) n_no_coor = le64(&d[p]);
    return true;
trunc:
    if (err) *err = "truncated BAI file: " + path;
    return false;
}

// [3P] CSI v1 (htslib csi spec): magic, min_shift, depth, l_aux + aux, n_ref, then per contig n_bin x {bin, loffset,
// n_chunk, chunks}; optional n_no_coor.  The file is BGZF-compressed: gzread() walks the concatenated gzip members.
bool BaiIndex::load_csi(const std::string &path, std::string *err) {
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) {
        if (err) *err = "cannot open index " + path;
        return false;
    }
    std::vector<uint8_t> d;
    uint8_t tmp[1 << 16];
    int n;
    while ((n = gzread(g, tmp, sizeof tmp)) > 0) d.insert(d.end(), tmp, tmp + n);
    gzclose(g);
    size_t p = 0;
    auto need = [&](size_t k) { return p + k <= d.size(); };
    if (n < 0 || !need(16) || std::memcmp(d.data(), "CSI\1", 4) != 0) {
        if (err) *err = "not a CSI file: " + path;
        return false;
    }
    min_shift = (int)le32(&d[4]);
    depth = (int)le32(&d[8]);
    const uint32_t l_aux = le32(&d[12]);
    if (min_shift < 1 || min_shift > 30 || depth < 1 || depth > 9 || min_shift + 3 * depth > 62) {
        if (err) *err = "unsupported CSI binning in " + path;
        return false;
    }
    p = 16;
    if (!need((size_t)l_aux + 4)) goto trunc;
    p += l_aux;
    {
        const uint32_t n_ref = le32(&d[p]);
        p += 4;
        if ((uint64_t)n_ref > (d.size() - p) / 4) goto trunc;  // a contig takes at least its n_bin word (see load())
        refs.assign(n_ref, BaiRef());
        csi = true;
        const uint32_t meta = meta_bin();
        for (uint32_t r = 0; r < n_ref; ++r) {
            BaiRef &R = refs[r];
            if (!need(4)) goto trunc;
            const uint32_t n_bin = le32(&d[p]);
            p += 4;
            bool first = true;
            for (uint32_t b = 0; b < n_bin; ++b) {
                if (!need(16)) goto trunc;
                const uint32_t bin = le32(&d[p]);
                const uint64_t loffset = le64(&d[p + 4]);
                const uint32_t n_chunk = le32(&d[p + 12]);
                p += 16;
                if (!need((size_t)n_chunk * 16)) goto trunc;
                if (bin == meta && n_chunk == 2) {
                    R.has_meta = true;
                    R.n_mapped = le64(&d[p + 16]);
                    R.n_unmapped = le64(&d[p + 24]);
                } else {
                    BinMap::Chunk *v = R.bins.add(bin, n_chunk);
                    R.loff[bin] = loffset;
                    if (n_chunk && bin < level_first(depth + 1)) {
                        int l = 0;
                        while (bin >= level_first(l + 1)) ++l;
                        R.csi_reach = std::max(R.csi_reach, (int64_t)(bin - level_first(l) + 1) << level_shift(l));
                    }
                    for (uint32_t c = 0; c < n_chunk; ++c) {
                        const uint64_t beg = le64(&d[p + 16 * c]), end = le64(&d[p + 16 * c + 8]);
                        v[c] = BinMap::Chunk(beg, end);
                        if (first || beg < R.min_offset) R.min_offset = beg;
                        if (first || end > R.max_offset) R.max_offset = end;
                        first = false;
                    }
                }
                p += (size_t)n_chunk * 16;
            }
            R.bins.seal();
        }
        if (need(8)) n_no_coor = le64(&d[p]);
    }
    return true;
trunc:
    if (err) *err = "truncated CSI file: " + path;
    return false;
}

uint64_t BaiIndex::csi_min_off(int tid, int64_t beg) const {
    const BaiRef &R = refs[tid];
    if (beg < 0) beg = 0;
    const int64_t max_pos = (1ll << (min_shift + 3 * depth)) - 1;
    if (beg > max_pos) beg = max_pos;
    uint32_t bin = level_first(depth) + (uint32_t)(beg >> min_shift);
    for (;;) {
        auto it = R.loff.find(bin);
        if (it != R.loff.end()) return it->second;
        if (bin == 0) return 0;
        const uint32_t parent = (bin - 1) >> 3, first_sibling = (parent << 3) + 1;
        bin = bin > first_sibling ? bin - 1 : parent;
    }
}

uint64_t BaiIndex::scan_start(int tid, int64_t beg) const {
    if (tid < 0 || (size_t)tid >= refs.size()) return 0;
    const BaiRef &R = refs[tid];
    if (R.bins.empty()) return 0;
    if (beg < 0) beg = 0;
    if (csi) {
        // nothing at or behind beg when every bin ends in front of it (a record lies inside its bin); otherwise htslib's
        // starting offset, and where that is 0 (no bin at or in front of beg's window) the contig's first record
        if (beg >= R.csi_reach) return 0;
        const uint64_t off = csi_min_off(tid, beg);
        return off ? off : R.min_offset;
    }
    size_t w = (size_t)(beg >> 14);
    if (w >= R.ioffset.size()) {
        // no record overlaps any window at or beyond the last indexed one... except through the
        // linear index being shorter than the data's reach; fall back to the bins
        return R.ioffset.empty() ? R.min_offset : 0;
    }
    // the linear index holds, per 16 kb window, the smallest offset of a record overlapping it;
    // empty windows are 0 in files written by some tools: look forward for the next filled one
    // (records overlapping an empty window do not exist, later records start later in the file)
    for (size_t i = w; i < R.ioffset.size(); ++i)
        if (R.ioffset[i]) return R.ioffset[i];
    return 0;
}

std::vector<std::pair<uint64_t, uint64_t>> BaiIndex::query(int tid, int64_t beg, int64_t end) const {
    std::vector<std::pair<uint64_t, uint64_t>> out;
    if (tid < 0 || (size_t)tid >= refs.size() || beg >= end) return out;
    const BaiRef &R = refs[tid];
    if (beg < 0) beg = 0;
    uint64_t min_off = 0;
    if (csi) min_off = csi_min_off(tid, beg);
    else if (!R.ioffset.empty()) {
        size_t w = (size_t)(beg >> 14);
        min_off = w < R.ioffset.size() ? R.ioffset[w] : R.ioffset.back();
    }
    // reg2bins over the levels of the UCSC binning scheme (.bai: min_shift 14, depth 5)
    int64_t e = end - 1;
    const int64_t max_pos = (1ll << (min_shift + 3 * depth)) - 1;
    if (beg > max_pos) beg = max_pos;
    if (e > max_pos) e = max_pos;
    for (int l = 0; l <= depth; ++l) {
        uint32_t b0 = level_first(l) + (uint32_t)(beg >> level_shift(l)), b1 = level_first(l) + (uint32_t)(e >> level_shift(l));
        for (auto it = R.bins.lower_bound(b0); it != R.bins.end() && it->first <= b1; ++it)
            for (auto &c : it->second)
                if (c.second > min_off) out.emplace_back(c.first, c.second);
    }
    std::sort(out.begin(), out.end());
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    for (auto &c : out) {
        if (!merged.empty() && (c.first >> 16) <= (merged.back().second >> 16)) {  // same or adjacent block
            if (c.second > merged.back().second) merged.back().second = c.second;
        } else
            merged.push_back(c);
    }
    return merged;
}

// ---------------- BAM ----------------

int64_t bam_ref_span(const uint32_t *cigar, uint32_t n) {
    int64_t rlen = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t op = cigar[i] & 0xf;
        if ((0x18Du >> op) & 1u) rlen += cigar[i] >> 4;  // M D N = X
    }
    return rlen;
}

bool BamFile::open(const std::string &path, std::string *err) {
    if (!bgzf_.open(path, err)) return false;
    uint8_t m[8];
    if (bgzf_.read(m, 8, err) != 8 || std::memcmp(m, "BAM\1", 4) != 0) {
        if (err && err->empty()) *err = "not a BAM file: " + path;
        return false;
    }
    uint32_t l_text = le32(m + 4);
    text_.assign(l_text, '\0');
    if (l_text && bgzf_.read(&text_[0], l_text, err) != (int64_t)l_text) return false;
    while (!text_.empty() && text_.back() == '\0') text_.pop_back();
    uint8_t nr[4];
    if (bgzf_.read(nr, 4, err) != 4) return false;
    uint32_t n_ref = le32(nr);
    refs_.clear();
    name2tid_.clear();
    for (uint32_t i = 0; i < n_ref; ++i) {
        uint8_t l[4];
        if (bgzf_.read(l, 4, err) != 4) return false;
        uint32_t l_name = le32(l);
        std::string name(l_name, '\0');
        if (l_name && bgzf_.read(&name[0], l_name, err) != (int64_t)l_name) return false;
        while (!name.empty() && name.back() == '\0') name.pop_back();
        if (bgzf_.read(l, 4, err) != 4) return false;
        refs_.push_back({name, (int64_t)le32(l)});
        name2tid_.emplace(name, (int)i);  // first wins, like a hash built front to back
    }
    first_rec_ = bgzf_.tell();
    // [3P] htslib (hts_idx_load -> idx_find_and_load: hts_idx_getfn(fn, ".csi") first, ".bai" only after it): <path>.csi, then
    // <path minus extension>.csi, then <path>.bai, then <path minus extension>.bai.  With x.bam.bai next to x.csi the .csi wins.
    // The first of the four that EXISTS is the index; failing to load it is an error (no falling through to the next).
    std::string alt = path;
    const size_t dot = alt.rfind('.');
    const size_t slash = alt.rfind('/');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) alt = alt.substr(0, dot);
    auto exists = [](const std::string &f) {
        FILE *t = std::fopen(f.c_str(), "rb");
        if (t) std::fclose(t);
        return t != nullptr;
    };
    if (exists(path + ".csi")) return bai_.load_csi(path + ".csi", err);
    if (alt != path && exists(alt + ".csi")) return bai_.load_csi(alt + ".csi", err);
    if (exists(path + ".bai")) return bai_.load(path + ".bai", err);
    if (alt != path && exists(alt + ".bai")) return bai_.load(alt + ".bai", err);
    if (err) *err = "could not load index for " + path + " (no .csi or .bai next to it)";
    return false;
}

int BamFile::tid(const std::string &name) const {
    auto it = name2tid_.find(name);
    return it == name2tid_.end() ? -1 : it->second;
}

std::map<std::string, uint64_t> BamFile::sq_lengths(std::string *err) const {
    std::map<std::string, uint64_t> out;
    size_t p = 0;
    while (p < text_.size()) {
        size_t e = text_.find('\n', p);
        if (e == std::string::npos) e = text_.size();
        std::string line = text_.substr(p, e - p);
        p = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.compare(0, 3, "@SQ") != 0) continue;
        std::string sn, ln;
        bool have_sn = false, have_ln = false;
        size_t q = 3;
        while (q < line.size()) {
            size_t t = line.find('\t', q + 1);
            if (t == std::string::npos) t = line.size();
            std::string f = line.substr(q + 1, t - q - 1);
            if (f.compare(0, 3, "SN:") == 0) sn = f.substr(3), have_sn = true;
            if (f.compare(0, 3, "LN:") == 0) ln = f.substr(3), have_ln = true;
            q = t;
        }
        if (!have_sn || !have_ln) {  // record["SN"] / record["LN"] index panics in the reference
            if (err) *err = "@SQ line without SN or LN";
            return {};
        }
        uint64_t v = 0;
        if (ln.empty()) {
            if (err) *err = "Failed to parse length of chromosome";
            return {};
        }
        for (char c : ln) {
            if (c < '0' || c > '9') {
                if (err) *err = "Failed to parse length of chromosome";
                return {};
            }
            v = v * 10 + (uint64_t)(c - '0');
        }
        out[sn] = v;
    }
    return out;
}

static size_t aux_value_size(char type, const uint8_t *p, const uint8_t *end, bool *ok) {
    *ok = true;
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': {
        const uint8_t *q = p;
        while (q < end && *q) ++q;
        if (q >= end) { *ok = false; return 0; }
        return (size_t)(q - p) + 1;
    }
    case 'B': {
        if (p + 5 > end) { *ok = false; return 0; }
        char st = (char)p[0];
        uint32_t n = le32(p + 1);
        size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
        return 5 + (size_t)n * es;
    }
    default: *ok = false; return 0;
    }
}

int BamFile::next(BamRec &rec, std::string *err) {
    uint64_t vo = bgzf_.tell();
    uint8_t l[4];
    std::string e;
    int64_t g = bgzf_.read(l, 4, &e);
    if (g == 0) return 0;
    if (g != 4) {
        if (err) *err = e.empty() ? "truncated BAM record" : e;
        return -1;
    }
    uint32_t block_size = le32(l);
    if (block_size < 32 || block_size > (1u << 30)) {
        if (err) *err = "corrupt BAM record";
        return -1;
    }
    buf_.resize(block_size + 1);
    if (bgzf_.read(buf_.data(), block_size, &e) != (int64_t)block_size) {
        if (err) *err = e.empty() ? "truncated BAM record" : e;
        return -1;
    }
    buf_[block_size] = 0;
    const uint8_t *b = buf_.data();
    rec.voffset = vo;
    rec.tid = (int32_t)le32(b);
    rec.pos = (int32_t)le32(b + 4);
    uint8_t l_read_name = b[8];
    rec.mapq = b[9];
    uint16_t n_cigar = le16(b + 12);
    rec.flag = le16(b + 14);
    uint32_t l_seq = le32(b + 16);
    size_t off_cigar = 32 + (size_t)l_read_name;
    size_t off_seq = off_cigar + (size_t)n_cigar * 4;
    size_t off_aux = off_seq + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (off_aux > block_size) {
        if (err) *err = "corrupt BAM record (field lengths)";
        return -1;
    }
    rec.n_cigar = n_cigar;
    rec.cigar = reinterpret_cast<const uint32_t *>(b + off_cigar);  // block payload starts 4-aligned in buf_
    if ((off_cigar & 3) != 0) {  // read name length not a multiple of 4: take an aligned copy
        cg_.resize(n_cigar);
        std::memcpy(cg_.data(), b + off_cigar, (size_t)n_cigar * 4);
        rec.cigar = cg_.data();
    }
    rec.hp_type = 0;
    rec.hp_value = 0;
    rec.sa_type = 0;
    rec.sa = nullptr;
    const uint8_t *cg_payload = nullptr;
    uint32_t cg_len = 0;
    bool cg_ok_type = false;
    const uint8_t *p = b + off_aux, *end = b + block_size;
    while (p + 3 <= end) {
        char t0 = (char)p[0], t1 = (char)p[1], type = (char)p[2];
        const uint8_t *v = p + 3;
        bool ok;
        size_t sz = aux_value_size(type, v, end, &ok);
        if (!ok || v + sz > end) break;  // htslib stops at a malformed aux field
        if (t0 == 'H' && t1 == 'P' && rec.hp_type == 0) {  // first match, like bam_aux_get
            rec.hp_type = type;
            switch (type) {
            case 'c': rec.hp_value = (int8_t)v[0]; break;
            case 'C': rec.hp_value = v[0]; break;
            case 's': rec.hp_value = (int16_t)le16(v); break;
            case 'S': rec.hp_value = le16(v); break;
            case 'i': rec.hp_value = (int32_t)le32(v); break;
            case 'I': rec.hp_value = le32(v); break;
            default: break;
            }
        } else if (t0 == 'S' && t1 == 'A' && rec.sa_type == 0) {
            rec.sa_type = type;
            if (type == 'Z') rec.sa = reinterpret_cast<const char *>(v);
        } else if (t0 == 'C' && t1 == 'G' && !cg_payload) {
            if (type == 'B' && (v[0] == 'I' || v[0] == 'i')) {
                cg_ok_type = true;
                cg_len = le32(v + 1);
                cg_payload = v + 5;
            } else {
                cg_payload = v;  // present but wrong type: htslib leaves the record alone
            }
        }
        p = v + sz;
    }
    // [3P] bam_tag2cigar: real CIGAR in CG:B,I when the stored one is <l_seq>S<ref_len>N
    if (cg_payload && cg_ok_type && n_cigar > 0 && rec.tid >= 0 && rec.pos >= 0) {
        uint32_t c0 = rec.cigar[0];
        if ((c0 & 0xf) == 4 && (c0 >> 4) == l_seq && cg_len >= n_cigar && cg_len < (1u << 29)) {
            cg_.resize(cg_len);
            std::memcpy(cg_.data(), cg_payload, (size_t)cg_len * 4);
            rec.cigar = cg_.data();
            rec.n_cigar = cg_len;
        }
    }
    return 1;
}

}  // namespace inqhost
