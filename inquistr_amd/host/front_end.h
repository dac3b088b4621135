// front_end.h — BAM + targets -> batches for the device (the fetch()/rc_records() stage).
//
// The reference index-fetches every locus separately (and, with -t > 1, re-opens the BAM and
// reloads the .bai per locus, src/call.rs:217), so a read overlapping k loci is inflated and
// decoded k times.  Here each contig is swept ONCE in file order and joined against its loci
// sorted by start; the index is only used to find where to start and to jump over gaps.  A read
// is stored once and referenced by every locus it overlaps; candidate lists stay in file order,
// which is the order rc_records() yields (the unphased tie rule depends on it).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/inquistr_hip.h"
#include "bam_reader.h"
#include "targets.h"

namespace inqhost {

struct HostBatch {
    std::vector<uint32_t> cigar;
    std::vector<inq_read_t> reads;
    std::vector<uint32_t> pair_read;
    std::vector<uint64_t> locus_pair_off;
    std::vector<uint32_t> locus_start, locus_end;
    std::vector<uint32_t> locus_index;  // index into the target list
    void clear();
    void view(inq_batch_t *b, uint32_t minlen, uint32_t support, bool unphased) const;
};

class FrontEnd {
public:
    FrontEnd(BamFile &bam, const std::vector<RepeatInterval> &targets, bool unphased);
    void set_max_batch_words(uint64_t w) { max_words_ = w ? w : max_words_; }
    // 1 = batch produced, 0 = done, -1 = error (err = message; panic = reference would panic)
    int next(HostBatch &out, std::string *err, bool *panic);

private:
    struct Locus {
        uint32_t start_ext, end_ext, start, end, index;
    };
    struct Group {
        int tid;
        std::vector<Locus> loci;
    };
    bool begin_group(std::string *err);
    int add_read(const BamRec &r, bool has_clip, std::string *err, bool *panic);
    void emit(size_t from, size_t to, HostBatch &out);
    void compact(size_t keep_from);

    BamFile &bam_;
    bool unphased_;
    uint64_t max_words_ = 48ull << 20;  // 192 MB of CIGAR per batch
    std::vector<Group> groups_;
    size_t g_ = 0;       // current group
    bool in_group_ = false, group_eof_ = false;
    size_t lo_ = 0;      // first locus of the group not yet closed
    size_t flushed_ = 0; // loci [0, flushed_) already emitted
    int64_t last_pos_ = -1;  // position of the previous record of the sweep (sortedness check)
    // contig-local store
    std::vector<uint32_t> cig_;
    std::vector<inq_read_t> reads_;
    // (locus of the group, read of the store) in file order; grouped by locus (stable) when a batch is emitted
    struct Edge {
        uint32_t locus, read;
    };
    std::vector<Edge> edges_;
    std::vector<uint64_t> bucket_;  // scratch of emit()
};

}  // namespace inqhost
