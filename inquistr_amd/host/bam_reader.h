// bam_reader.h — the slice of BAM/BAI the `inquiSTR call` path touches, on top of bgzf.h.
//
// Replaces rust-htslib / htslib on the reference's path:
//   IndexedReader::from_path + header()        src/call.rs:161-163, 239, 337
//   fetch((tid, beg, end)) / rc_records()      src/call.rs:288, 294, 338, 345   (BAI lookup, record decode)
//   Record::{reference_start, mapq, cigar, aux(b"HP"), aux(b"SA"), is_reverse}   src/call.rs:297-299, 380-382, 423, 483
// SEQ / QUAL are skipped, never copied.  Long CIGARs stored in the CG:B,I tag are swapped in as
// htslib does ([3P] sam.c bam_tag2cigar).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "bgzf.h"

namespace inqhost {

struct BamRef {
    std::string name;
    int64_t len = 0;
};

// The bins of one contig: bin id -> its chunks.  What the code needs of a std::map<uint32_t, std::vector<chunk>> - iteration in bin
// order with ->first / ->second, lower_bound, the chunks as a range - without a tree node and a vector allocation per bin: a
// whole-genome index holds 10^5 bins, and building the map was 8 of the 10 ms an open took (every rank of a multi-GPU run opens).
// Entries are appended while the index is parsed (add), then sorted once (seal); the chunks of all bins lie in one array.
class BinMap {
public:
    typedef std::pair<uint64_t, uint64_t> Chunk;
    struct Chunks {  // the chunks of one bin, as a range
        const Chunk *b = nullptr;
        uint32_t n = 0;
        const Chunk *begin() const { return b; }
        const Chunk *end() const { return b + n; }
        bool empty() const { return n == 0; }
        size_t size() const { return n; }
        const Chunk &operator[](size_t i) const { return b[i]; }
    };
    struct Entry {
        uint32_t first = 0;  // bin id
        Chunks second;
        uint32_t at = 0;     // (index of the first chunk in the array, until seal() turns it into the pointer)
    };
    typedef const Entry *const_iterator;
    // parsing: room for `n` more chunks of bin `bin`; returns where to write them
    Chunk *add(uint32_t bin, uint32_t n) {
        Entry e;
        e.first = bin, e.second.n = n, e.at = (uint32_t)chunks_.size();
        entries_.push_back(e);
        chunks_.resize(chunks_.size() + n);
        return chunks_.data() + e.at;
    }
    void reserve(size_t bins, size_t chunks) { entries_.reserve(bins), chunks_.reserve(chunks); }
    void seal();  // sorts by bin id (stable: equal ids keep file order), fixes the pointers
    BinMap() = default;
    BinMap(BinMap &&) = default;  // (a moved vector keeps its storage: the pointers stay good)
    BinMap &operator=(BinMap &&) = default;
    BinMap(const BinMap &o) : entries_(o.entries_), chunks_(o.chunks_) { fix(); }
    BinMap &operator=(const BinMap &o) {
        if (this != &o) entries_ = o.entries_, chunks_ = o.chunks_, fix();
        return *this;
    }
    bool empty() const { return entries_.empty(); }
    size_t size() const { return entries_.size(); }
    const_iterator begin() const { return entries_.data(); }
    const_iterator end() const { return entries_.data() + entries_.size(); }
    const_iterator lower_bound(uint32_t bin) const;

private:
    void fix();  // entries' chunk pointers from their indices
    std::vector<Entry> entries_;
    std::vector<Chunk> chunks_;
};

struct BaiRef {
    BinMap bins;
    std::map<uint32_t, uint64_t> loff;  // .csi only: per bin, the smallest offset of a record overlapping the bin's first window
    std::vector<uint64_t> ioffset;  // .bai only: 16 kb linear index
    uint64_t n_mapped = 0, n_unmapped = 0;
    bool has_meta = false;
    uint64_t min_offset = 0, max_offset = 0;  // over real bins
    int64_t csi_reach = 0;  // .csi only: end of the right-most real bin - no record of the contig reaches this position
};

// A .bai, or a .csi ([3P] htslib: IndexedReader::from_path takes either, src/call.rs:242).  Both use the UCSC binning
// scheme; a .bai fixes min_shift = 14 and depth = 5 and carries a linear index, a .csi states both numbers and keeps one
// offset per bin instead (the linear index entry of the bin's first window).
struct BaiIndex {
    std::vector<BaiRef> refs;
    uint64_t n_no_coor = 0;
    int min_shift = 14, depth = 5;
    bool csi = false;
    bool load(const std::string &path, std::string *err);      // .bai
    bool load_csi(const std::string &path, std::string *err);  // .csi (BGZF-compressed)
    uint32_t level_first(int l) const { return (uint32_t)(((1ull << (3 * l)) - 1) / 7); }  // first bin id of level l
    int level_shift(int l) const { return min_shift + 3 * (depth - l); }                  // log2 of a level-l bin's width
    uint32_t meta_bin() const { return level_first(depth + 1) + 1; }                       // htslib's pseudo-bin
    // .csi: the offset htslib's iterator starts from for position beg - the loff of the deepest existing bin at or in
    // front of beg's window, walking left through the siblings and up through the parents ([3P] hts_itr_query)
    uint64_t csi_min_off(int tid, int64_t beg) const;
    // a virtual offset from which a forward scan sees every record of `tid` overlapping positions >= beg; 0 when the
    // index shows that the contig has no such record (.bai: no filled linear-index window at or behind beg; .csi: beg
    // lies behind the contig's right-most bin).  .bai: the smallest such offset, monotone in beg.  .csi: htslib's
    // iterator start (csi_min_off), which is NOT monotone in beg (a deeper bin left of beg can carry a larger loff
    // than its parent): the planner opens a new span when an offset steps back.
    uint64_t scan_start(int tid, int64_t beg) const;
    // [3P] htslib hts_itr_query for a BAI: the chunks a region query [beg, end) has to read, from the
    // bins overlapping the region, cut below by the linear index, sorted and merged.
    std::vector<std::pair<uint64_t, uint64_t>> query(int tid, int64_t beg, int64_t end) const;
};

// One decoded record: only the fields the path reads.
struct BamRec {
    int32_t tid = -1;
    int32_t pos = -1;
    uint8_t mapq = 0;
    uint16_t flag = 0;
    uint32_t n_cigar = 0;
    const uint32_t *cigar = nullptr;  // into the reader's record buffer (or CG tag payload)
    // aux
    char hp_type = 0;   // 0 = absent, else BAM aux type char
    int64_t hp_value = 0;
    char sa_type = 0;   // 0 = absent
    const char *sa = nullptr;  // NUL-terminated when sa_type == 'Z'
    uint64_t voffset = 0;      // virtual offset of the record (identity for de-duplication)
};

class BamFile {
public:
    explicit BamFile(int n_threads = 1) : bgzf_(n_threads) {}
    // Opens <path> and its index (<path>.csi, <path minus extension>.csi, <path>.bai, <path minus extension>.bai: htslib's
    // order). Both are required, as for IndexedReader::from_path.
    bool open(const std::string &path, std::string *err);
    const std::vector<BamRef> &refs() const { return refs_; }
    const std::string &header_text() const { return text_; }
    int tid(const std::string &name) const;  // -1 if absent (header().tid())
    // @SQ SN -> LN from the header TEXT, as get_chrom_lengths_from_bam_header (src/call.rs:161-180)
    std::map<std::string, uint64_t> sq_lengths(std::string *err) const;
    const BaiIndex &index() const { return bai_; }

    bool seek(uint64_t voffset, std::string *err) { return bgzf_.seek(voffset, err); }
    // Next record in file order. Returns 1 = record, 0 = end of file, -1 = error.
    int next(BamRec &rec, std::string *err);
    uint64_t first_record_voffset() const { return first_rec_; }

private:
    BgzfReader bgzf_;
    std::vector<BamRef> refs_;
    std::map<std::string, int> name2tid_;
    std::string text_;
    BaiIndex bai_;
    std::vector<uint8_t> buf_;
    std::vector<uint32_t> cg_;  // aligned copy of a CG tag payload
    uint64_t first_rec_ = 0;
};

// [3P] htslib bam_endpos / bam_cigar2rlen on a decoded record (host needs it for the sweep join and
// for is_accidental_2d; the device recomputes it for the filters).
int64_t bam_ref_span(const uint32_t *cigar, uint32_t n);
inline int64_t bam_endpos(const BamRec &r) {
    int64_t rlen = (r.flag & 0x4) ? 0 : bam_ref_span(r.cigar, r.n_cigar);
    if (rlen == 0) rlen = 1;
    return (int64_t)r.pos + rlen;
}

}  // namespace inqhost
