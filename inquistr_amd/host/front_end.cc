#include "front_end.h"

#include <algorithm>
#include <cstring>

#include "sa2d.h"

namespace inqhost {

void HostBatch::clear() {
    cigar.clear();
    reads.clear();
    pair_read.clear();
    locus_pair_off.clear();
    locus_start.clear();
    locus_end.clear();
    locus_index.clear();
}

void HostBatch::view(inq_batch_t *b, uint32_t minlen, uint32_t support, bool unphased) const {
    std::memset(b, 0, sizeof *b);
    b->n_reads = reads.size();
    b->n_cigar_words = cigar.size();
    b->n_pairs = pair_read.size();
    b->n_loci = locus_start.size();
    b->cigar = cigar.data();
    b->reads = reads.data();
    b->pair_read = pair_read.data();
    b->locus_pair_off = locus_pair_off.data();
    b->locus_start = locus_start.data();
    b->locus_end = locus_end.data();
    b->minlen = minlen;
    b->support = support;
    b->unphased = unphased ? 1u : 0u;
}

FrontEnd::FrontEnd(BamFile &bam, const std::vector<RepeatInterval> &targets, bool unphased)
    : bam_(bam), unphased_(unphased) {
    // group by contig (header().tid(), src/call.rs:287,337), sort each group by start
    std::vector<std::pair<int, uint32_t>> order;
    for (uint32_t i = 0; i < targets.size(); ++i) order.emplace_back(bam_.tid(targets[i].chrom), i);
    std::stable_sort(order.begin(), order.end(), [&](const auto &a, const auto &b) {
        if (a.first != b.first) return a.first < b.first;
        const RepeatInterval &x = targets[a.second], &y = targets[b.second];
        if (x.start != y.start) return x.start < y.start;
        return x.end < y.end;
    });
    for (auto &o : order) {
        const RepeatInterval &t = targets[o.second];
        if (groups_.empty() || groups_.back().tid != o.first) groups_.push_back({o.first, {}});
        // src/call.rs:285-286 / 335-336 (u32; start >= 10 is checked by the driver before)
        groups_.back().loci.push_back({t.start - 10u, t.end + 10u, t.start, t.end, o.second});
    }
}

bool FrontEnd::begin_group(std::string *err) {
    Group &G = groups_[g_];
    pairs_.assign(G.loci.size(), {});
    cig_.clear();
    reads_.clear();
    lo_ = 0;
    flushed_ = 0;
    group_eof_ = false;
    in_group_ = true;
    if (G.tid < 0 || G.loci.empty()) {
        group_eof_ = true;
        return true;
    }
    uint64_t vo = bam_.index().scan_start(G.tid, (int64_t)G.loci[0].start_ext);
    if (vo == 0) {
        group_eof_ = true;  // nothing on this contig at or after the first window
        return true;
    }
    return bam_.seek(vo, err);
}

int FrontEnd::add_read(const BamRec &r, std::string *err, bool *panic) {
    // get_phase(): any aux type but U8 ('C') / I32 ('i') panics, for every record fetch() yields in
    // phased mode (src/call.rs:349, 482-491)
    uint8_t bits = 0, phase = 0;
    if (r.flag & 0x4) bits |= INQ_READ_UNMAPPED;
    if (r.flag & 0x10) bits |= INQ_READ_REVERSE;
    if (r.hp_type) {
        if (r.hp_type == 'C' || r.hp_type == 'i') {
            bits |= INQ_READ_HAS_HP;
            phase = (uint8_t)(uint32_t)(int32_t)r.hp_value;  // `v as u8`
        } else if (!unphased_) {
            *err = std::string("Unexpected type of Aux for HP: ") + r.hp_type;
            *panic = true;
            return -1;
        }
    }
    // is_accidental_2d is only reached from a soft-clip op (src/call.rs:394)
    bool has_clip = false;
    for (uint32_t i = 0; i < r.n_cigar && !has_clip; ++i) has_clip = (r.cigar[i] & 0xf) == 4;
    if (has_clip && r.sa_type) {
        std::string pm;
        int v = is_accidental_2d(r, &pm);
        if (v < 0) {
            *err = pm;
            *panic = true;
            return -1;
        }
        if (v) bits |= INQ_READ_IS_2D;
    }
    inq_read_t rd;
    std::memset(&rd, 0, sizeof rd);
    rd.cigar_off4 = (uint32_t)(cig_.size() / 4);
    rd.n_cigar = r.n_cigar;
    rd.pos = r.pos;
    rd.mapq = r.mapq;
    rd.bits = bits;
    rd.phase = phase;
    cig_.insert(cig_.end(), r.cigar, r.cigar + r.n_cigar);
    while (cig_.size() & 3) cig_.push_back(0u);  // 0M padding to the next 16-byte boundary
    reads_.push_back(rd);
    return (int)reads_.size() - 1;
}

void FrontEnd::emit(size_t from, size_t to, HostBatch &out) {
    const Group &G = groups_[g_];
    out.clear();
    // reads referenced by loci [from, to), renumbered in first-use order
    std::vector<uint32_t> remap(reads_.size(), 0xffffffffu);
    out.locus_pair_off.push_back(0);
    for (size_t j = from; j < to; ++j) {
        for (uint32_t ri : pairs_[j]) {
            if (remap[ri] == 0xffffffffu) {
                remap[ri] = (uint32_t)out.reads.size();
                inq_read_t rd = reads_[ri];
                const uint32_t *src = cig_.data() + (size_t)rd.cigar_off4 * 4;
                rd.cigar_off4 = (uint32_t)(out.cigar.size() / 4);
                size_t n4 = ((size_t)rd.n_cigar + 3) / 4 * 4;
                out.cigar.insert(out.cigar.end(), src, src + n4);
                out.reads.push_back(rd);
            }
            out.pair_read.push_back(remap[ri]);
        }
        out.locus_pair_off.push_back(out.pair_read.size());
        out.locus_start.push_back(G.loci[j].start);
        out.locus_end.push_back(G.loci[j].end);
        out.locus_index.push_back(G.loci[j].index);
    }
}

void FrontEnd::compact(size_t keep_from) {
    // drop reads no open locus references
    std::vector<uint32_t> remap(reads_.size(), 0xffffffffu);
    std::vector<uint32_t> ncig;
    std::vector<inq_read_t> nreads;
    for (size_t j = keep_from; j < pairs_.size(); ++j) {
        for (uint32_t &ri : pairs_[j]) {
            if (remap[ri] == 0xffffffffu) {
                remap[ri] = (uint32_t)nreads.size();
                inq_read_t rd = reads_[ri];
                const uint32_t *src = cig_.data() + (size_t)rd.cigar_off4 * 4;
                rd.cigar_off4 = (uint32_t)(ncig.size() / 4);
                size_t n4 = ((size_t)rd.n_cigar + 3) / 4 * 4;
                ncig.insert(ncig.end(), src, src + n4);
                nreads.push_back(rd);
            }
            ri = remap[ri];
        }
    }
    for (size_t j = 0; j < keep_from; ++j) std::vector<uint32_t>().swap(pairs_[j]);
    cig_.swap(ncig);
    reads_.swap(nreads);
}

int FrontEnd::next(HostBatch &out, std::string *err, bool *panic) {
    *panic = false;
    for (;;) {
        if (g_ >= groups_.size()) return 0;
        if (!in_group_ && !begin_group(err)) return -1;
        Group &G = groups_[g_];
        const size_t m = G.loci.size();
        BamRec rec;
        while (!group_eof_) {
            int rc = bam_.next(rec, err);
            if (rc < 0) return -1;
            if (rc == 0 || rec.tid != G.tid) {
                if (rc != 0 && rec.tid >= 0 && rec.tid < G.tid) continue;  // still before the contig
                group_eof_ = true;
                break;
            }
            const int64_t pos = rec.pos;
            bool closed_some = false;
            while (lo_ < m && (int64_t)G.loci[lo_].end_ext <= pos) {
                ++lo_;
                closed_some = true;
            }
            if (lo_ >= m) {
                group_eof_ = true;
                break;
            }
            if (closed_some && (int64_t)G.loci[lo_].start_ext > pos + (1 << 16)) {
                // far from the next locus: let the linear index jump the gap
                uint64_t vo = bam_.index().scan_start(G.tid, (int64_t)G.loci[lo_].start_ext);
                if (vo == 0) {
                    group_eof_ = true;
                    break;
                }
                if (vo > rec.voffset) {
                    if (!bam_.seek(vo, err)) return -1;
                    continue;
                }
            }
            const int64_t endpos = bam_endpos(rec);
            int ridx = -1;
            for (size_t j = lo_; j < m && (int64_t)G.loci[j].start_ext < endpos; ++j) {
                if ((int64_t)G.loci[j].end_ext > pos) {  // [3P] htslib: pos < end && endpos > beg
                    if (ridx < 0) {
                        ridx = add_read(rec, err, panic);
                        if (ridx < 0) return -1;
                    }
                    pairs_[j].push_back((uint32_t)ridx);
                }
            }
            if (cig_.size() > max_words_ && lo_ > flushed_) {
                emit(flushed_, lo_, out);
                compact(lo_);
                flushed_ = lo_;
                return 1;
            }
        }
        // contig done: everything left is closed
        if (flushed_ < m) {
            emit(flushed_, m, out);
            flushed_ = m;
            in_group_ = false;
            ++g_;
            return 1;
        }
        in_group_ = false;
        ++g_;
    }
}

}  // namespace inqhost
