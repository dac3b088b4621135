// bgzf.h — minimal BGZF (blocked gzip) reader on zlib, with a pool of inflate workers.
//
// Replaces the slice of htslib's bgzf.c the path uses under rust-htslib's IndexedReader
// (reference call sites: src/call.rs:239 from_path, :288/:338 fetch, :294/:345 rc_records).
// Only what `inquiSTR call` needs: seek to a virtual offset, then read forward.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace inqhost {

struct BgzfBlock {
    uint64_t coffset = 0;        // file offset of the block
    uint32_t csize = 0;          // compressed size (whole block)
    std::vector<uint8_t> raw;    // compressed bytes
    std::vector<uint8_t> data;   // inflated payload
    bool ok = false, done = false, eof_marker = false;
};

class BgzfReader {
public:
    // n_threads <= 1: inflate on the calling thread; otherwise n_threads-1 helper threads inflate ahead.
    explicit BgzfReader(int n_threads = 1);
    ~BgzfReader();
    bool open(const std::string &path, std::string *err);
    void close();
    // Position at a BAM virtual offset (coffset << 16 | uoffset).
    bool seek(uint64_t voffset, std::string *err);
    // Reads exactly n bytes into dst. Returns n, 0 at clean EOF before the first byte, -1 on error/truncation.
    int64_t read(void *dst, size_t n, std::string *err);
    // Virtual offset of the next byte read() would return.
    uint64_t tell() const;
    uint64_t file_size() const { return file_size_; }

private:
    bool fill(std::string *err);                 // make cur_ the next non-empty block
    std::shared_ptr<BgzfBlock> fetch_next(std::string *err);
    bool read_raw_block(BgzfBlock &b, std::string *err);
    static bool inflate_block(BgzfBlock &b);
    void worker();
    void schedule_ahead();

    FILE *fp_ = nullptr;
    uint64_t file_size_ = 0;
    uint64_t next_coffset_ = 0;                  // where the next raw block will be read from
    std::shared_ptr<BgzfBlock> cur_;
    size_t cur_pos_ = 0;
    bool at_eof_ = false;

    int n_threads_;
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<std::shared_ptr<BgzfBlock>> todo_;     // raw blocks waiting for inflate
    std::deque<std::shared_ptr<BgzfBlock>> ordered_;  // blocks in file order (being or already inflated)
    bool stop_ = false;
    size_t ahead_ = 0;
};

}  // namespace inqhost
