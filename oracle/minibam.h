// minibam.h — a second, independent BGZF / BAM / BAI reader.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.
//
// Used by oracle/ref_shaped_call.cc (the CPU baseline and end-to-end cross-check) so that the CPU side shares NO decoding code
// with the product (inquistr_amd/host/bgzf.cc, bam_reader.cc, the GPU inflate and record scan): an error in either decoder shows
// up as a `.inq` difference.  Written from the SAM/BAM specification (SAMv1 section 4: BGZF, BAM, the BAI index and its
// reg2bins query) on plain zlib; [3P] where it follows htslib behaviour rather than the spec:
//   * a region query reads the chunks of every bin overlapping [beg, end) whose end lies behind the linear-index offset of
//     beg's 16 kb window, in file order, and yields records with tid == target, pos < end, bam_endpos > beg;
//   * bam_endpos = pos + reference span of the CIGAR (M D N = X), 1 if unmapped or 0;
//   * a CIGAR of `<l_seq>S<n>N` with a CG:B,I tag is replaced by the tag's payload (CIGARs beyond 65 535 ops);
//   * the first HP / SA tag wins; a malformed aux field ends the scan.
// Nothing here is tuned: one block inflated at a time, every record copied.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace minibam {

inline uint32_t u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t u32(const uint8_t *p) { return u16(p) | (u16(p + 2) << 16); }
inline uint64_t u64(const uint8_t *p) { return (uint64_t)u32(p) | ((uint64_t)u32(p + 4) << 32); }

struct Record {
    int32_t tid = -1, pos = -1;
    uint8_t mapq = 0;
    uint16_t flag = 0;
    std::vector<uint32_t> cigar;
    char hp_type = 0;  // 0 = absent, else the BAM aux type character
    int64_t hp_value = 0;
    char sa_type = 0;
    std::string sa;
    int64_t endpos() const {
        int64_t span = 0;
        if (!(flag & 4))
            for (uint32_t w : cigar)
                if ((0x18Du >> (w & 15u)) & 1u) span += w >> 4;
        return (int64_t)pos + (span ? span : 1);
    }
};

class Bgzf {
public:
    ~Bgzf() {
        if (f_) std::fclose(f_);
    }
    bool open(const std::string &path) {
        f_ = std::fopen(path.c_str(), "rb");
        return f_ != nullptr;
    }
    // positions the cursor on a virtual offset (compressed offset of a block << 16 | offset inside its inflated bytes)
    bool seek(uint64_t voffset) {
        if (!load(voffset >> 16)) return false;
        pos_ = voffset & 0xffffu;
        return pos_ <= data_.size();
    }
    uint64_t tell() {
        if (pos_ == data_.size() && !eof_) {  // htslib reports the start of the next block for a cursor at a block's end
            if (!load(next_)) return ~0ull;
        }
        return (coff_ << 16) | pos_;
    }
    // false = end of file or error before n bytes were there
    bool read(void *dst, size_t n) {
        uint8_t *d = (uint8_t *)dst;
        while (n) {
            if (pos_ == data_.size()) {
                if (eof_ || !load(next_)) return false;
                if (data_.empty() && eof_) return false;
                continue;
            }
            const size_t k = std::min(n, data_.size() - pos_);
            std::memcpy(d, data_.data() + pos_, k);
            d += k;
            pos_ += k;
            n -= k;
        }
        return true;
    }
    bool failed() const { return bad_; }

private:
    bool load(uint64_t coff) {
        data_.clear();
        pos_ = 0;
        coff_ = coff;
        uint8_t h[18];
        if (fseeko(f_, (off_t)coff, SEEK_SET) != 0) return bad_ = true, false;
        const size_t got = std::fread(h, 1, 18, f_);
        if (got == 0) {
            eof_ = true;
            next_ = coff;
            return true;
        }
        // gzip member with one extra subfield "BC" holding BSIZE (SAMv1 4.1)
        if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || u16(h + 10) != 6 || h[12] != 'B' || h[13] != 'C' ||
            u16(h + 14) != 2)
            return bad_ = true, false;
        const size_t bsize = u16(h + 16) + 1;
        if (bsize < 26) return bad_ = true, false;
        std::vector<uint8_t> body(bsize - 18);
        if (std::fread(body.data(), 1, body.size(), f_) != body.size()) return bad_ = true, false;
        const uint32_t crc = u32(body.data() + body.size() - 8), isize = u32(body.data() + body.size() - 4);
        data_.resize(isize);
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) return bad_ = true, false;
        z.next_in = body.data();
        z.avail_in = (uInt)(body.size() - 8);
        z.next_out = data_.data();
        z.avail_out = isize;
        const int rc = isize ? inflate(&z, Z_FINISH) : Z_STREAM_END;
        const bool ok = (rc == Z_STREAM_END || (isize == 0 && rc == Z_OK)) && z.total_out == isize;
        inflateEnd(&z);
        if (!ok || (uint32_t)crc32(crc32(0L, Z_NULL, 0), data_.data(), isize) != crc) return bad_ = true, false;
        next_ = coff + bsize;
        eof_ = false;
        return true;
    }
    FILE *f_ = nullptr;
    std::vector<uint8_t> data_;
    size_t pos_ = 0;
    uint64_t coff_ = 0, next_ = 0;
    bool eof_ = false, bad_ = false;
};

struct Bai {
    struct Ref {
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> linear;
    };
    std::vector<Ref> refs;
    bool load(const std::string &path) {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        std::fclose(f);
        if (d.size() < 8 || std::memcmp(d.data(), "BAI\1", 4) != 0) return false;
        size_t p = 8;
        refs.assign(u32(&d[4]), Ref());
        for (Ref &R : refs) {
            if (p + 4 > d.size()) return false;
            const uint32_t n_bin = u32(&d[p]);
            p += 4;
            for (uint32_t b = 0; b < n_bin; ++b) {
                if (p + 8 > d.size()) return false;
                const uint32_t bin = u32(&d[p]), n_chunk = u32(&d[p + 4]);
                p += 8;
                if (p + 16ull * n_chunk > d.size()) return false;
                if (bin != 37450)  // the metadata pseudo-bin is not data
                    for (uint32_t c = 0; c < n_chunk; ++c) R.bins[bin].emplace_back(u64(&d[p + 16 * c]), u64(&d[p + 16 * c + 8]));
                p += 16ull * n_chunk;
            }
            if (p + 4 > d.size()) return false;
            const uint32_t n_intv = u32(&d[p]);
            p += 4;
            if (p + 8ull * n_intv > d.size()) return false;
            for (uint32_t i = 0; i < n_intv; ++i) R.linear.push_back(u64(&d[p + 8 * i]));
            p += 8ull * n_intv;
        }
        return true;
    }
    // SAMv1 5.3 reg2bins + the linear index as a lower bound
    std::vector<std::pair<uint64_t, uint64_t>> chunks(int tid, int64_t beg, int64_t end) const {
        std::vector<std::pair<uint64_t, uint64_t>> out;
        if (tid < 0 || (size_t)tid >= refs.size() || beg >= end) return out;
        const Ref &R = refs[tid];
        uint64_t min_off = 0;
        if (!R.linear.empty()) min_off = R.linear[std::min<size_t>((size_t)(beg >> 14), R.linear.size() - 1)];
        --end;
        std::vector<uint32_t> bins = {0};
        for (uint32_t k = 1 + (uint32_t)(beg >> 26); k <= 1 + (uint32_t)(end >> 26); ++k) bins.push_back(k);
        for (uint32_t k = 9 + (uint32_t)(beg >> 23); k <= 9 + (uint32_t)(end >> 23); ++k) bins.push_back(k);
        for (uint32_t k = 73 + (uint32_t)(beg >> 20); k <= 73 + (uint32_t)(end >> 20); ++k) bins.push_back(k);
        for (uint32_t k = 585 + (uint32_t)(beg >> 17); k <= 585 + (uint32_t)(end >> 17); ++k) bins.push_back(k);
        for (uint32_t k = 4681 + (uint32_t)(beg >> 14); k <= 4681 + (uint32_t)(end >> 14); ++k) bins.push_back(k);
        for (uint32_t b : bins) {
            auto it = R.bins.find(b);
            if (it == R.bins.end()) continue;
            for (auto &c : it->second)
                if (c.second > min_off) out.push_back(c);
        }
        std::sort(out.begin(), out.end());
        // overlapping or touching chunks become one, so that no record is yielded twice
        std::vector<std::pair<uint64_t, uint64_t>> merged;
        for (auto &c : out) {
            if (!merged.empty() && c.first <= merged.back().second) merged.back().second = std::max(merged.back().second, c.second);
            else merged.push_back(c);
        }
        return merged;
    }
};

class Reader {
public:
    std::vector<std::pair<std::string, int64_t>> refs;
    std::string text;
    bool open(const std::string &path) {
        if (!bgzf_.open(path)) return false;
        uint8_t m[8];
        if (!bgzf_.seek(0) || !bgzf_.read(m, 8) || std::memcmp(m, "BAM\1", 4) != 0) return false;
        text.assign(u32(m + 4), '\0');
        if (!text.empty() && !bgzf_.read(&text[0], text.size())) return false;
        uint8_t w[4];
        if (!bgzf_.read(w, 4)) return false;
        const uint32_t n_ref = u32(w);
        for (uint32_t i = 0; i < n_ref; ++i) {
            if (!bgzf_.read(w, 4)) return false;
            std::string name(u32(w), '\0');
            if (!name.empty() && !bgzf_.read(&name[0], name.size())) return false;
            while (!name.empty() && name.back() == '\0') name.pop_back();
            if (!bgzf_.read(w, 4)) return false;
            refs.emplace_back(name, (int64_t)u32(w));
        }
        std::string base = path;
        if (!bai_.load(path + ".bai")) {
            const size_t dot = base.rfind('.');
            if (dot == std::string::npos || !bai_.load(base.substr(0, dot) + ".bai")) return false;
        }
        return true;
    }
    int tid(const std::string &name) const {
        for (size_t i = 0; i < refs.size(); ++i)
            if (refs[i].first == name) return (int)i;  // the first of equal names, like a lookup table built front to back
        return -1;
    }
    // the records a region query yields, in file order; false on a read error
    bool fetch(int t, int64_t beg, int64_t end, std::vector<Record> &out) {
        out.clear();
        for (auto &c : bai_.chunks(t, beg, end)) {
            if (!bgzf_.seek(c.first)) return false;
            for (;;) {
                const uint64_t at = bgzf_.tell();
                if (at == ~0ull) return false;
                if (at >= c.second) break;
                Record r;
                const int rc = next(r);
                if (rc < 0) return false;
                if (rc == 0) break;
                if (r.tid != t || (int64_t)r.pos >= end) return true;  // coordinate-sorted: nothing further on can overlap
                if (r.endpos() > beg) out.push_back(std::move(r));
            }
        }
        return !bgzf_.failed();
    }

private:
    // 1 = record, 0 = end of file, -1 = malformed
    int next(Record &r) {
        uint8_t w[4];
        if (!bgzf_.read(w, 4)) return bgzf_.failed() ? -1 : 0;
        const uint32_t bs = u32(w);
        if (bs < 32) return -1;
        buf_.resize(bs);
        if (!bgzf_.read(buf_.data(), bs)) return -1;
        const uint8_t *b = buf_.data();
        r.tid = (int32_t)u32(b);
        r.pos = (int32_t)u32(b + 4);
        const uint32_t l_name = b[8], n_cigar = u16(b + 12), l_seq = u32(b + 16);
        r.mapq = b[9];
        r.flag = (uint16_t)u16(b + 14);
        const uint64_t off_cigar = 32 + (uint64_t)l_name, off_seq = off_cigar + 4ull * n_cigar;
        const uint64_t off_aux = off_seq + ((uint64_t)l_seq + 1) / 2 + l_seq;
        if (off_aux > bs) return -1;
        r.cigar.resize(n_cigar);
        for (uint32_t k = 0; k < n_cigar; ++k) r.cigar[k] = u32(b + off_cigar + 4 * k);
        // aux fields: TAG (2), type (1), value
        const uint8_t *p = b + off_aux, *end = b + bs;
        const uint8_t *cg = nullptr;
        uint32_t cg_n = 0;
        bool have_cg = false;
        while (p + 3 <= end) {
            const char t0 = (char)p[0], t1 = (char)p[1], type = (char)p[2];
            const uint8_t *v = p + 3;
            uint64_t size;
            switch (type) {
            case 'A': case 'c': case 'C': size = 1; break;
            case 's': case 'S': size = 2; break;
            case 'i': case 'I': case 'f': size = 4; break;
            case 'd': size = 8; break;
            case 'Z': case 'H': {
                const void *z = std::memchr(v, 0, (size_t)(end - v));
                size = z ? (uint64_t)((const uint8_t *)z - v) + 1 : 0;
                break;
            }
            case 'B': {
                if (v + 5 > end) { size = 0; break; }
                const char st = (char)v[0];
                const uint64_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                size = 5 + es * u32(v + 1);
                break;
            }
            default: size = 0;
            }
            if (size == 0 || v + size > end) break;
            if (t0 == 'H' && t1 == 'P' && !r.hp_type) {
                r.hp_type = type;
                switch (type) {
                case 'c': r.hp_value = (int8_t)v[0]; break;
                case 'C': r.hp_value = v[0]; break;
                case 's': r.hp_value = (int16_t)u16(v); break;
                case 'S': r.hp_value = u16(v); break;
                case 'i': r.hp_value = (int32_t)u32(v); break;
                case 'I': r.hp_value = u32(v); break;
                default: r.hp_value = 0;
                }
            } else if (t0 == 'S' && t1 == 'A' && !r.sa_type) {
                r.sa_type = type;
                if (type == 'Z') r.sa.assign((const char *)v, (size_t)size - 1);
            } else if (t0 == 'C' && t1 == 'G' && !have_cg) {
                have_cg = true;
                if (type == 'B' && (v[0] == 'I' || v[0] == 'i')) cg = v + 5, cg_n = u32(v + 1);
            }
            p = v + size;
        }
        if (cg && n_cigar > 0 && r.pos >= 0 && (r.cigar[0] & 15u) == 4u && (r.cigar[0] >> 4) == l_seq && cg_n >= n_cigar) {
            r.cigar.resize(cg_n);
            for (uint32_t k = 0; k < cg_n; ++k) r.cigar[k] = u32(cg + 4 * k);
        }
        return 1;
    }
    Bgzf bgzf_;
    Bai bai_;
    std::vector<uint8_t> buf_;
};

}  // namespace minibam
