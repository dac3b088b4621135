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
    edges_.clear();
    cig_.clear();
    reads_.clear();
    lo_ = 0;
    flushed_ = 0;
    group_eof_ = false;
    in_group_ = true;
    last_pos_ = -1;
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

int FrontEnd::add_read(const BamRec &r, bool has_clip, std::string *err, bool *panic) {
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
    // is_accidental_2d is only reached from a soft-clip op (src/call.rs:394) of a read that passed the filter
    // (call_from_cigar is called behind it, :303,357): a panic in there is carried as a bit of the descriptor
    // and raised by the device for KEPT reads only
    if (has_clip && r.sa_type) {
        int v = is_accidental_2d(r, nullptr);
        if (v < 0) bits |= INQ_READ_SA_PANIC;
        else if (v) bits |= INQ_READ_IS_2D;
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
    const size_t nl = to - from;
    // stable counting sort of the edges of loci [from, to) by locus: file order inside a locus is kept
    bucket_.assign(nl + 1, 0);
    size_t n_out = 0;
    for (const Edge &e : edges_)
        if (e.locus >= from && e.locus < to) {
            ++bucket_[e.locus - from + 1];
            ++n_out;
        }
    for (size_t j = 0; j < nl; ++j) bucket_[j + 1] += bucket_[j];
    out.locus_pair_off.assign(bucket_.begin(), bucket_.end());
    out.pair_read.resize(n_out);
    const bool whole = n_out == edges_.size();  // nothing stays behind: the store becomes the batch as it is
    std::vector<uint32_t> remap;
    if (whole) {
        for (const Edge &e : edges_) out.pair_read[bucket_[e.locus - from]++] = e.read;
        out.cigar.swap(cig_);
        out.reads.swap(reads_);
        cig_.clear();
        reads_.clear();
        edges_.clear();
    } else {
        remap.assign(reads_.size(), 0xffffffffu);
        for (const Edge &e : edges_) {
            if (e.locus < from || e.locus >= to) continue;
            if (remap[e.read] == 0xffffffffu) {
                remap[e.read] = (uint32_t)out.reads.size();
                inq_read_t rd = reads_[e.read];
                const uint32_t *src = cig_.data() + (size_t)rd.cigar_off4 * 4;
                rd.cigar_off4 = (uint32_t)(out.cigar.size() / 4);
                const size_t n4 = ((size_t)rd.n_cigar + 3) / 4 * 4;
                out.cigar.insert(out.cigar.end(), src, src + n4);
                out.reads.push_back(rd);
            }
            out.pair_read[bucket_[e.locus - from]++] = remap[e.read];
        }
    }
    for (size_t j = from; j < to; ++j) {
        out.locus_start.push_back(G.loci[j].start);
        out.locus_end.push_back(G.loci[j].end);
        out.locus_index.push_back(G.loci[j].index);
    }
}

void FrontEnd::compact(size_t keep_from) {
    // drop the edges of emitted loci and the reads no open locus references
    std::vector<uint32_t> remap(reads_.size(), 0xffffffffu);
    std::vector<uint32_t> ncig;
    std::vector<inq_read_t> nreads;
    size_t w = 0;
    for (const Edge &e0 : edges_) {
        if (e0.locus < keep_from) continue;
        Edge e = e0;
        if (remap[e.read] == 0xffffffffu) {
            remap[e.read] = (uint32_t)nreads.size();
            inq_read_t rd = reads_[e.read];
            const uint32_t *src = cig_.data() + (size_t)rd.cigar_off4 * 4;
            rd.cigar_off4 = (uint32_t)(ncig.size() / 4);
            const size_t n4 = ((size_t)rd.n_cigar + 3) / 4 * 4;
            ncig.insert(ncig.end(), src, src + n4);
            nreads.push_back(rd);
        }
        e.read = remap[e.read];
        edges_[w++] = e;
    }
    edges_.resize(w);
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
            if (pos < last_pos_) {  // an index exists only for coordinate-sorted files; a sweep needs the order too
                *err = "BAM records are not coordinate-sorted (contig " + bam_.refs()[G.tid].name + ")";
                return -1;
            }
            last_pos_ = pos;
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
                    last_pos_ = -1;  // the first record behind a jump only has to overlap the target window
                    continue;
                }
            }
            // one pass over the CIGAR: reference span ([3P] bam_endpos) and "has a soft clip"
            int64_t rlen = 0;
            bool has_clip = false;
            for (uint32_t i = 0; i < rec.n_cigar; ++i) {
                const uint32_t op = rec.cigar[i] & 0xf;
                if ((0x18Du >> op) & 1u) rlen += rec.cigar[i] >> 4;
                has_clip |= op == 4u;
            }
            if ((rec.flag & 0x4) || rlen == 0) rlen = 1;
            const int64_t endpos = pos + rlen;
            int ridx = -1;
            for (size_t j = lo_; j < m && (int64_t)G.loci[j].start_ext < endpos; ++j) {
                if ((int64_t)G.loci[j].end_ext > pos) {  // [3P] htslib: pos < end && endpos > beg
                    if (ridx < 0) {
                        ridx = add_read(rec, has_clip, err, panic);
                        if (ridx < 0) return -1;
                    }
                    edges_.push_back({(uint32_t)j, (uint32_t)ridx});
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
