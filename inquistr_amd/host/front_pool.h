// front_pool.h - the host sweep's worker pool: several sweep workers, each with its own reader (front_pool.cc)
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "front_end.h"
#include "targets.h"

namespace inqhost {

// Several sweep workers, each with its own reader, over contiguous slices of the position-sorted
// target list; batches flow to the caller through a bounded queue.  Replaces the reference's
// rayon par_bridge over loci (src/call.rs:115-118) on the decode side.
class ParallelFrontEnd {
public:
    struct Item {
        HostBatch batch;
        std::vector<uint32_t> index;  // position of each batch locus in the full target list
    };
    ParallelFrontEnd(const std::string &bam_path, BamFile &hdr, const std::vector<RepeatInterval> &targets, bool unphased,
                     int n_workers, uint64_t max_words = 0);
    ~ParallelFrontEnd() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_space_.notify_all();
        for (auto &t : pool_) t.join();
    }
    // hand a consumed item back so its vectors' capacity is reused (no mmap/munmap churn while the
    // HIP runtime is pinning pages on another thread)
    void recycle(Item &&it) {
        std::lock_guard<std::mutex> g(mu_);
        if (free_.size() < 16) free_.push_back(std::move(it));
    }
    // 1 = item, 0 = done, -1 = error
    int next(Item &out, std::string *err, bool *panic) {
        std::unique_lock<std::mutex> g(mu_);
        cv_item_.wait(g, [&] { return !q_.empty() || live_ == 0 || failed_; });
        if (failed_) {
            *err = err_;
            *panic = panic_;
            return -1;
        }
        if (q_.empty()) return 0;
        out = std::move(q_.front());
        q_.pop_front();
        cv_space_.notify_one();
        return 1;
    }

private:
    void work();

    std::string path_;
    const std::vector<RepeatInterval> &targets_;
    bool unphased_;
    uint64_t max_words_ = 0;
    std::vector<std::vector<uint32_t>> slices_;
    std::atomic<size_t> next_slice_{0};
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_item_, cv_space_;
    std::deque<Item> q_;
    std::vector<Item> free_;
    int live_ = 0;
    bool stop_ = false, failed_ = false, panic_ = false;
    std::string err_;
};

}  // namespace inqhost
