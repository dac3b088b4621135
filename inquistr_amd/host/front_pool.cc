// front_pool.cc - the host sweep's worker pool (front_pool.h): several sweep workers, each with its own reader, over contiguous
// slices of the position-sorted target list; batches flow to the caller through a bounded queue.  Replaces the reference's rayon
// par_bridge over loci (src/call.rs:115-118) on the decode side.
#include "front_pool.h"

#include <algorithm>

namespace inqhost {

ParallelFrontEnd::ParallelFrontEnd(const std::string &bam_path, BamFile &hdr, const std::vector<RepeatInterval> &targets, bool unphased,
                 int n_workers, uint64_t max_words)
    : path_(bam_path), targets_(targets), unphased_(unphased), max_words_(max_words) {
    std::vector<uint32_t> order(targets.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        int ta = hdr.tid(targets[a].chrom), tb = hdr.tid(targets[b].chrom);
        if (ta != tb) return ta < tb;
        return targets[a].start < targets[b].start;
    });
    const size_t n = order.size();
    const size_t n_slices = std::max<size_t>(1, std::min<size_t>(n, (size_t)n_workers * 3));
    for (size_t k = 0; k < n_slices; ++k) {
        size_t lo = n * k / n_slices, hi = n * (k + 1) / n_slices;
        if (hi > lo) slices_.emplace_back(order.begin() + lo, order.begin() + hi);
    }
    live_ = n_workers;
    for (int w = 0; w < n_workers; ++w) pool_.emplace_back([this] { work(); });
}

void ParallelFrontEnd::work() {
    BamFile bam(1);
    std::string e;
    bool ok = bam.open(path_, &e);
    for (;;) {
        size_t k = next_slice_.fetch_add(1);
        if (!ok || k >= slices_.size()) break;
        std::vector<RepeatInterval> sub;
        sub.reserve(slices_[k].size());
        for (uint32_t i : slices_[k]) sub.push_back(targets_[i]);
        FrontEnd fe(bam, sub, unphased_);
        if (max_words_) fe.set_max_batch_words(max_words_);
        for (;;) {
            Item it;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!free_.empty()) {
                    it = std::move(free_.back());
                    free_.pop_back();
                }
            }
            bool panic = false;
            int rc = fe.next(it.batch, &e, &panic);
            if (rc < 0) {
                std::lock_guard<std::mutex> g(mu_);
                if (!failed_) failed_ = true, err_ = e, panic_ = panic;
                ok = false;
                break;
            }
            if (rc == 0) break;
            it.index.resize(it.batch.locus_index.size());
            for (size_t j = 0; j < it.index.size(); ++j) it.index[j] = slices_[k][it.batch.locus_index[j]];
            std::unique_lock<std::mutex> g(mu_);
            cv_space_.wait(g, [&] { return q_.size() < 8 || stop_; });
            if (stop_) return;
            q_.push_back(std::move(it));
            cv_item_.notify_one();
        }
        if (!ok) break;
    }
    if (!ok && !e.empty()) {
        std::lock_guard<std::mutex> g(mu_);
        if (!failed_) failed_ = true, err_ = e, panic_ = true;
    }
    std::lock_guard<std::mutex> g(mu_);
    --live_;
    cv_item_.notify_all();
}

}  // namespace inqhost
