#include "bgzf.h"

#include <zlib.h>

#include <cstring>

namespace inqhost {

static inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

BgzfReader::BgzfReader(int n_threads) : n_threads_(n_threads < 1 ? 1 : n_threads) {
    ahead_ = n_threads_ > 1 ? (size_t)n_threads_ * 8 : 0;
    for (int i = 1; i < n_threads_; ++i) pool_.emplace_back([this] { worker(); });
}

BgzfReader::~BgzfReader() {
    close();
    {
        std::lock_guard<std::mutex> g(mu_);
        stop_ = true;
    }
    cv_work_.notify_all();
    for (auto &t : pool_) t.join();
}

bool BgzfReader::open(const std::string &path, std::string *err) {
    close();
    fp_ = std::fopen(path.c_str(), "rb");
    if (!fp_) {
        if (err) *err = "cannot open " + path;
        return false;
    }
    std::fseek(fp_, 0, SEEK_END);
    file_size_ = (uint64_t)std::ftell(fp_);
    std::fseek(fp_, 0, SEEK_SET);
    next_coffset_ = 0;
    cur_.reset();
    cur_pos_ = 0;
    at_eof_ = false;
    return true;
}

void BgzfReader::close() {
    {
        std::unique_lock<std::mutex> g(mu_);
        // wait for in-flight inflates: workers hold shared_ptrs, so dropping the queues is enough
        todo_.clear();
        ordered_.clear();
    }
    if (fp_) std::fclose(fp_);
    fp_ = nullptr;
    cur_.reset();
}

bool BgzfReader::read_raw_block(BgzfBlock &b, std::string *err) {
    // BGZF block: gzip member with an extra subfield 'B','C',len=2 holding BSIZE (block size - 1)
    uint8_t hdr[18];
    if (std::fseek(fp_, (long)next_coffset_, SEEK_SET) != 0) {
        if (err) *err = "seek failed";
        return false;
    }
    size_t got = std::fread(hdr, 1, 12, fp_);
    if (got == 0) {
        b.eof_marker = true;  // physical end of file
        b.csize = 0;
        return true;
    }
    if (got != 12 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) {
        if (err) *err = "not a BGZF block at offset " + std::to_string(next_coffset_);
        return false;
    }
    uint16_t xlen = le16(hdr + 10);
    std::vector<uint8_t> extra(xlen);
    if (std::fread(extra.data(), 1, xlen, fp_) != xlen) {
        if (err) *err = "truncated BGZF header";
        return false;
    }
    int bsize = -1;
    for (size_t i = 0; i + 4 <= extra.size();) {
        uint16_t slen = le16(&extra[i + 2]);
        if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= extra.size()) bsize = le16(&extra[i + 4]);
        i += 4 + slen;
    }
    if (bsize < 0) {
        if (err) *err = "BGZF BC subfield missing";
        return false;
    }
    b.coffset = next_coffset_;
    b.csize = (uint32_t)bsize + 1;
    size_t head = 12 + (size_t)xlen;
    if (b.csize < head + 8) {
        if (err) *err = "bad BGZF block size";
        return false;
    }
    b.raw.resize(b.csize - head);
    if (std::fread(b.raw.data(), 1, b.raw.size(), fp_) != b.raw.size()) {
        if (err) *err = "truncated BGZF block";
        return false;
    }
    next_coffset_ += b.csize;
    return true;
}

bool BgzfReader::inflate_block(BgzfBlock &b) {
    // raw = deflate stream + CRC32 + ISIZE
    if (b.raw.size() < 8) return false;
    uint32_t isize = le32(&b.raw[b.raw.size() - 4]);
    uint32_t crc = le32(&b.raw[b.raw.size() - 8]);
    b.data.resize(isize);
    if (isize) {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = b.raw.data();
        zs.avail_in = (uInt)(b.raw.size() - 8);
        zs.next_out = b.data.data();
        zs.avail_out = isize;
        int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize) return false;
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), b.data.data(), isize) != crc) return false;
    }
    std::vector<uint8_t>().swap(b.raw);
    return true;
}

void BgzfReader::worker() {
    for (;;) {
        std::shared_ptr<BgzfBlock> b;
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_work_.wait(g, [this] { return stop_ || !todo_.empty(); });
            if (stop_) return;
            b = todo_.front();
            todo_.pop_front();
        }
        bool ok = inflate_block(*b);
        {
            std::lock_guard<std::mutex> g(mu_);
            b->ok = ok;
            b->done = true;
        }
        cv_done_.notify_all();
    }
}

void BgzfReader::schedule_ahead() {
    // called with mu_ NOT held; reads raw blocks sequentially and queues them for the workers
    for (;;) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (ordered_.size() >= ahead_ || at_eof_) return;
        }
        auto b = std::make_shared<BgzfBlock>();
        std::string e;
        if (!read_raw_block(*b, &e)) {
            b->done = true;
            b->ok = false;
            std::lock_guard<std::mutex> g(mu_);
            ordered_.push_back(b);
            at_eof_ = true;
            return;
        }
        std::lock_guard<std::mutex> g(mu_);
        if (b->eof_marker) {
            b->done = b->ok = true;
            ordered_.push_back(b);
            at_eof_ = true;
            return;
        }
        ordered_.push_back(b);
        todo_.push_back(b);
        cv_work_.notify_one();
    }
}

std::shared_ptr<BgzfBlock> BgzfReader::fetch_next(std::string *err) {
    if (n_threads_ <= 1) {
        auto b = std::make_shared<BgzfBlock>();
        if (!read_raw_block(*b, err)) return nullptr;
        if (b->eof_marker) return b;
        if (!inflate_block(*b)) {
            if (err) *err = "BGZF inflate / CRC failure at offset " + std::to_string(b->coffset);
            return nullptr;
        }
        b->ok = b->done = true;
        return b;
    }
    schedule_ahead();
    std::shared_ptr<BgzfBlock> b;
    {
        std::unique_lock<std::mutex> g(mu_);
        if (ordered_.empty()) {  // physical EOF reached earlier
            auto e = std::make_shared<BgzfBlock>();
            e->eof_marker = e->ok = e->done = true;
            return e;
        }
        b = ordered_.front();
        cv_done_.wait(g, [&] { return b->done; });
        ordered_.pop_front();
    }
    if (!b->ok) {
        if (err) *err = "BGZF read / inflate failure near offset " + std::to_string(b->coffset);
        return nullptr;
    }
    return b;
}

bool BgzfReader::seek(uint64_t voffset, std::string *err) {
    {
        std::lock_guard<std::mutex> g(mu_);
        todo_.clear();
        ordered_.clear();
        at_eof_ = false;
    }
    next_coffset_ = voffset >> 16;
    cur_.reset();
    cur_pos_ = 0;
    const uint32_t uoff = (uint32_t)(voffset & 0xffff);
    auto b = fetch_next(err);
    if (!b) return false;
    cur_ = b;
    if (b->eof_marker) return uoff == 0;
    if (uoff > b->data.size()) {
        if (err) *err = "virtual offset beyond block";
        return false;
    }
    cur_pos_ = uoff;
    return true;
}

bool BgzfReader::fill(std::string *err) {
    while (!cur_ || cur_pos_ >= cur_->data.size()) {
        if (cur_ && cur_->eof_marker) return false;
        auto b = fetch_next(err);
        if (!b) return false;
        cur_ = b;
        cur_pos_ = 0;
        if (b->eof_marker) return false;
    }
    return true;
}

int64_t BgzfReader::read(void *dst, size_t n, std::string *err) {
    uint8_t *out = (uint8_t *)dst;
    size_t done = 0;
    while (done < n) {
        std::string e;
        if (!fill(&e)) {
            if (!e.empty()) {
                if (err) *err = e;
                return -1;
            }
            if (done == 0) return 0;
            if (err) *err = "truncated file";
            return -1;
        }
        size_t take = std::min(n - done, cur_->data.size() - cur_pos_);
        std::memcpy(out + done, cur_->data.data() + cur_pos_, take);
        cur_pos_ += take;
        done += take;
    }
    return (int64_t)done;
}

uint64_t BgzfReader::tell() const {
    if (!cur_) return next_coffset_ << 16;
    if (cur_pos_ >= cur_->data.size() && !cur_->eof_marker) return (cur_->coffset + cur_->csize) << 16;
    return (cur_->coffset << 16) | (uint64_t)cur_pos_;
}

}  // namespace inqhost
