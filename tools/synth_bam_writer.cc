// synth_bam_writer.cc — native twin of tools/make_synth_bam.py + tools/bamio.py for large synthetic BAMs.
//
// Measurement plumbing, never the product: turns a synthetic workload batch (inquistr_amd/synth.py, the SoA of
// inq_batch_t) into a coordinate-sorted BAM + .bai, byte for byte what the Python writer produces (same
// record layout, same 0xff00-byte BGZF blocks, zlib level / memLevel, same index), but with the record
// assembly, the deflate and the index built by native threads: 100 000 loci x 30 reads (2.5 GB of records)
// take seconds instead of minutes, which is what lets bench.py time the end-to-end (L2) leg in its default run.
//
//   g++ -O2 -std=c++17 -fPIC -shared -pthread -o libsynthbam.so synth_bam_writer.cc -lz
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t kBlock = 0xFF00;

struct Read {  // inq_read_t
    uint32_t cigar_off4, n_cigar;
    int32_t pos;
    uint8_t mapq, bits, phase, reserved;
};

uint32_t reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

template <class T>
void put(std::vector<uint8_t> &v, T x) {
    const size_t n = v.size();
    v.resize(n + sizeof(T));
    std::memcpy(v.data() + n, &x, sizeof(T));
}
template <class T>
void put_at(uint8_t *p, T x) {
    std::memcpy(p, &x, sizeof(T));
}

template <class F>
void parallel_for(uint64_t n, int threads, F f) {
    std::atomic<uint64_t> next{0};
    const uint64_t grain = std::max<uint64_t>(1, n / ((uint64_t)threads * 64));
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            for (;;) {
                const uint64_t a = next.fetch_add(grain);
                if (a >= n) return;
                const uint64_t b = std::min(n, a + grain);
                for (uint64_t i = a; i < b; ++i) f(i);
            }
        });
    for (auto &th : pool) th.join();
}


// ---- .bai (bamio._write_bai): bins with merged chunks, 16 kb linear index, htslib's metadata pseudo-bin.  u0[k] = offset of
// record k in the inflated byte string, coff[b] = file offset of block b.  Returns an error message, empty on success.
std::string write_bai(const char *bam_path, uint64_t n_reads, const uint64_t *order, const int32_t *tid, int n_contigs,
                      const std::vector<uint64_t> &u0, const std::vector<int64_t> &beg, const std::vector<int64_t> &end,
                      const std::vector<uint64_t> &coff, uint64_t n_blocks) {
    auto fail = [](const std::string &m) { return m; };
    auto vo = [&](uint64_t u) -> uint64_t {
        const uint64_t blk = u / kBlock, within = u % kBlock;
        if (blk >= n_blocks) return coff[n_blocks] << 16;
        return (coff[blk] << 16) | within;
    };
    std::vector<uint8_t> bai = {'B', 'A', 'I', 1};
    put<uint32_t>(bai, (uint32_t)n_contigs);
    uint64_t k = 0;
    for (int t = 0; t < n_contigs; ++t) {
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> lin;
        std::vector<uint8_t> lin_set;
        bool any = false;
        uint64_t m0 = 0, m1 = 0, n_mapped = 0;
        for (; k < n_reads && tid[order[k]] == t; ++k) {
            const uint64_t v0 = vo(u0[k]), v1 = vo(u0[k + 1]);
            const int64_t b0 = std::max<int64_t>(beg[k], 0), e0 = std::max<int64_t>(end[k], 1);
            auto &ch = bins[reg2bin(b0, e0)];
            if (!ch.empty() && ch.back().second == v0) ch.back().second = v1;
            else ch.emplace_back(v0, v1);
            for (int64_t w = b0 >> 14; w <= (e0 - 1) >> 14; ++w) {
                if ((uint64_t)w >= lin.size()) lin.resize(w + 1, 0), lin_set.resize(w + 1, 0);
                if (!lin_set[w]) lin[w] = v0, lin_set[w] = 1;
            }
            m0 = any ? std::min(m0, v0) : v0;
            m1 = any ? std::max(m1, v1) : v1;
            any = true;
            ++n_mapped;
        }
        put<uint32_t>(bai, (uint32_t)bins.size() + (any ? 1u : 0u));
        for (auto &kv : bins) {
            put<uint32_t>(bai, kv.first);
            put<uint32_t>(bai, (uint32_t)kv.second.size());
            for (auto &c : kv.second) put<uint64_t>(bai, c.first), put<uint64_t>(bai, c.second);
        }
        if (any) {
            put<uint32_t>(bai, 37450u);
            put<uint32_t>(bai, 2u);
            put<uint64_t>(bai, m0), put<uint64_t>(bai, m1), put<uint64_t>(bai, n_mapped), put<uint64_t>(bai, 0);
        }
        put<uint32_t>(bai, (uint32_t)lin.size());
        uint64_t nxt = 0;  // [3P] htslib fills empty windows with the NEXT filled window's offset, from the right (bamio._write_bai)
        for (size_t w = lin.size(); w-- > 0;) {
            if (lin_set[w]) nxt = lin[w];
            lin[w] = nxt;
        }
        for (size_t w = 0; w < lin.size(); ++w) put<uint64_t>(bai, lin[w]);
    }
    if (k != n_reads) return fail("order[] is not grouped by ascending tid");
    put<uint64_t>(bai, 0);  // n_no_coor
    {
        const std::string p = std::string(bam_path) + ".bai";
        FILE *f = std::fopen(p.c_str(), "wb");
        if (!f) return fail("cannot open " + p);
        std::fwrite(bai.data(), 1, bai.size(), f);
        if (std::fclose(f) != 0) return fail("close failed");
    }
    return std::string();
}

}  // namespace

extern "C" {

// reads / cigar: the batch's SoA (host).  order[k] = index of the k-th record of the file (sorted by (tid, pos),
// stable), tid[i] = contig of read i, name_id[i] = number printed into the read name "r%010d".
// Contigs are "chr1" .. "chr<n_contigs>", all contig_len long.  Returns 0, or -1 with a message in err.
int inq_synth_write_bam(const char *bam_path, uint64_t n_reads, const Read *reads, const uint32_t *cigar, const uint64_t *order,
                        const int32_t *tid, const uint64_t *name_id, int n_contigs, uint32_t contig_len, int level, int threads,
                        char *err, size_t err_cap) {
    auto fail = [&](const std::string &m) {
        std::snprintf(err, err_cap, "%s", m.c_str());
        return -1;
    };
    if (threads < 1) threads = 1;
    // ---- header (bamio.BamWriter.__init__)
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int c = 0; c < n_contigs; ++c) text += "@SQ\tSN:chr" + std::to_string(c + 1) + "\tLN:" + std::to_string(contig_len) + "\n";
    std::vector<uint8_t> head;
    head.insert(head.end(), {'B', 'A', 'M', 1});
    put<uint32_t>(head, (uint32_t)text.size());
    head.insert(head.end(), text.begin(), text.end());
    put<uint32_t>(head, (uint32_t)n_contigs);
    for (int c = 0; c < n_contigs; ++c) {
        const std::string nm = "chr" + std::to_string(c + 1);
        put<uint32_t>(head, (uint32_t)nm.size() + 1);
        head.insert(head.end(), nm.begin(), nm.end());
        head.push_back(0);
        put<uint32_t>(head, contig_len);
    }
    // ---- record extents: 4 (block_size) + 32 (core) + 12 (name) + 4 * n_cigar + 4 (HP:C:x)
    std::vector<uint64_t> u0(n_reads + 1);
    u0[0] = head.size();
    for (uint64_t k = 0; k < n_reads; ++k) u0[k + 1] = u0[k] + 52 + 4ull * reads[order[k]].n_cigar;
    const uint64_t total = u0[n_reads];
    std::vector<uint8_t> data(total);
    std::memcpy(data.data(), head.data(), head.size());
    std::vector<int64_t> beg(n_reads), end(n_reads);
    parallel_for(n_reads, threads, [&](uint64_t k) {
        const uint64_t i = order[k];
        const Read &r = reads[i];
        const uint32_t *w = cigar + (uint64_t)r.cigar_off4 * 4;
        int64_t span = 0;
        for (uint32_t c = 0; c < r.n_cigar; ++c)
            if ((0x18Du >> (w[c] & 15u)) & 1u) span += w[c] >> 4;
        beg[k] = r.pos;
        end[k] = (int64_t)r.pos + std::max<int64_t>(span, 1);
        uint8_t *p = data.data() + u0[k];
        put_at<int32_t>(p, (int32_t)(48 + 4 * r.n_cigar));
        put_at<int32_t>(p + 4, tid[i]);
        put_at<int32_t>(p + 8, r.pos);
        p[12] = 12;
        p[13] = r.mapq;
        put_at<uint16_t>(p + 14, (uint16_t)reg2bin(beg[k], end[k]));
        put_at<uint16_t>(p + 16, (uint16_t)r.n_cigar);
        put_at<uint16_t>(p + 18, 0);
        put_at<int32_t>(p + 20, 0);
        put_at<int32_t>(p + 24, -1);
        put_at<int32_t>(p + 28, -1);
        put_at<int32_t>(p + 32, 0);
        char name[16];
        std::snprintf(name, sizeof name, "r%010llu", (unsigned long long)name_id[i]);
        std::memcpy(p + 36, name, 12);  // 11 characters + NUL
        std::memcpy(p + 48, w, 4ull * r.n_cigar);
        uint8_t *a = p + 48 + 4ull * r.n_cigar;
        a[0] = 'H', a[1] = 'P', a[2] = 'C', a[3] = r.phase;
    });
    // ---- BGZF (bamio.bgzf_block): one deflate stream per 0xff00 bytes
    const uint64_t n_blocks = std::max<uint64_t>(1, (total + kBlock - 1) / kBlock);
    std::vector<std::vector<uint8_t>> comp(n_blocks);
    std::atomic<int> zfail{0};
    parallel_for(n_blocks, threads, [&](uint64_t b) {
        const uint64_t off = b * kBlock, len = std::min(kBlock, total - off);
        std::vector<uint8_t> &out = comp[b];
        out.resize(18 + compressBound((uLong)len) + 64 + 8);
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        std::memcpy(out.data(), hdr, 16);
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
            zfail = 1;
            return;
        }
        z.next_in = data.data() + off;
        z.avail_in = (uInt)len;
        z.next_out = out.data() + 18;
        z.avail_out = (uInt)(out.size() - 26);
        if (deflate(&z, Z_FINISH) != Z_STREAM_END) zfail = 1;
        const uint64_t body = z.total_out;
        deflateEnd(&z);
        put_at<uint16_t>(out.data() + 16, (uint16_t)(body + 25));
        put_at<uint32_t>(out.data() + 18 + body, (uint32_t)crc32(crc32(0L, Z_NULL, 0), data.data() + off, (uInt)len));
        put_at<uint32_t>(out.data() + 22 + body, (uint32_t)len);
        out.resize(26 + body);
        if (out.size() > 65536) zfail = 1;
    });
    if (zfail) return fail("deflate failed or a block does not fit 64 KB");
    std::vector<uint64_t> coff(n_blocks + 1, 0);
    for (uint64_t b = 0; b < n_blocks; ++b) coff[b + 1] = coff[b] + comp[b].size();
    {
        FILE *f = std::fopen(bam_path, "wb");
        if (!f) return fail(std::string("cannot open ") + bam_path);
        for (uint64_t b = 0; b < n_blocks; ++b)
            if (std::fwrite(comp[b].data(), 1, comp[b].size(), f) != comp[b].size()) {
                std::fclose(f);
                return fail("short write");
            }
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::fwrite(eof, 1, 28, f);
        if (std::fclose(f) != 0) return fail("close failed");
    }
    std::vector<std::vector<uint8_t>>().swap(comp);
    std::string e = write_bai(bam_path, n_reads, order, tid, n_contigs, u0, beg, end, coff, n_blocks);
    if (!e.empty()) return fail(e);
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Records shaped like a real long-read BAM (the native twin of make_synth_bam.records_with_seq, NOT byte-identical to it: the
// Python writer draws from numpy's generator): every record carries SEQ and QUAL of its query length, NM:i, an ML:B,C
// methylation array and its MM:Z string (one call per 25 bases), and HP:C as the LAST tag (where phasing tools append it).
//   qual_mode 0: Phred uniform in 0..50 and SEQ bytes uniform in 0..255 (what the Python writer does: barely compressible)
//   qual_mode 1: bases from ACGT only, Phred a clamped random walk (steps -2..2, occasional drops), ML skewed to confident
//                calls: the file compresses to 0.59 of its inflated size at zlib level 1 (0.57 at 6), in the range of
//                basecaller output; mode 0 gives 0.87
// The file is produced in slabs of `slab_blocks` BGZF blocks (records generated, deflated by `threads` workers, appended), so
// that tens of GB can be written with a few hundred MB of memory.  Every byte is a function of (seed, name_id) only.
namespace {

struct Rng {  // splitmix64 seeding + xoshiro256**
    uint64_t s[4];
    static uint64_t mix(uint64_t &x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) {
        for (auto &v : s) v = mix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0], s[3] ^= s[1], s[1] ^= s[2], s[0] ^= s[3], s[2] ^= t, s[3] = rotl(s[3], 45);
        return r;
    }
};

struct SeqShape {
    uint32_t l_seq, n_mod, mm_len;
    uint64_t size;  // whole record, block_size field included
};

// decimal length of v (0..39)
inline uint32_t dec_len(uint32_t v) { return v < 10 ? 1u : 2u; }

// The MM string's numbers come from their own generator (seeded from the record's) so that its length is known from a cheap pass.
inline Rng mm_rng(uint64_t seed, uint64_t id) { return Rng(seed ^ (id * 0x9e3779b97f4a7c15ull) ^ 0x4d4dull); }

SeqShape seq_shape(const Read &r, const uint32_t *w, uint64_t seed, uint64_t id) {
    uint64_t q = 0;
    for (uint32_t c = 0; c < r.n_cigar; ++c)
        if ((0x193u >> (w[c] & 15u)) & 1u) q += w[c] >> 4;  // M I S = X consume the query
    SeqShape sh;
    sh.l_seq = (uint32_t)std::min<uint64_t>(q, 0x7fffffffu);
    sh.n_mod = sh.l_seq / 25;
    Rng g = mm_rng(seed, id);
    uint32_t len = 4;  // "C+m,"
    for (uint32_t k = 0; k < sh.n_mod; ++k) len += dec_len((uint32_t)(g.next() % 40)) + (k + 1 < sh.n_mod ? 1u : 0u);
    sh.mm_len = len + 2;  // ";\0"
    sh.size = 4 + 32 + 12 + 4ull * r.n_cigar + (sh.l_seq + 1) / 2 + sh.l_seq + 7 + (8 + sh.n_mod) + (3 + sh.mm_len) + 4;
    return sh;
}

void seq_record(uint8_t *p, const Read &r, const uint32_t *w, int32_t tid, uint64_t id, uint64_t seed, const SeqShape &sh, uint32_t bin,
                int qual_mode) {
    put_at<int32_t>(p, (int32_t)(sh.size - 4));
    put_at<int32_t>(p + 4, tid);
    put_at<int32_t>(p + 8, r.pos);
    p[12] = 12;
    p[13] = r.mapq;
    put_at<uint16_t>(p + 14, (uint16_t)bin);
    put_at<uint16_t>(p + 16, (uint16_t)r.n_cigar);
    put_at<uint16_t>(p + 18, 0);
    put_at<int32_t>(p + 20, (int32_t)sh.l_seq);
    put_at<int32_t>(p + 24, -1);
    put_at<int32_t>(p + 28, -1);
    put_at<int32_t>(p + 32, 0);
    char name[16];
    std::snprintf(name, sizeof name, "r%010llu", (unsigned long long)id);
    std::memcpy(p + 36, name, 12);
    std::memcpy(p + 48, w, 4ull * r.n_cigar);
    uint8_t *q = p + 48 + 4ull * r.n_cigar;
    Rng g(seed ^ (id * 0x9e3779b97f4a7c15ull));
    const uint32_t n_seq = (sh.l_seq + 1) / 2;
    if (qual_mode == 0) {
        for (uint32_t i = 0; i < n_seq; i += 8) {
            const uint64_t v = g.next();
            std::memcpy(q + i, &v, std::min<uint32_t>(8, n_seq - i));
        }
        q += n_seq;
        for (uint32_t i = 0; i < sh.l_seq;) {
            uint64_t v = g.next();
            for (int k = 0; k < 10 && i < sh.l_seq; ++k, v >>= 6) q[i++] = (uint8_t)((v & 63u) * 51u >> 6);
        }
        q += sh.l_seq;
    } else {
        static const uint8_t two[16] = {0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88};
        for (uint32_t i = 0; i < n_seq;) {
            uint64_t v = g.next();
            for (int k = 0; k < 16 && i < n_seq; ++k, v >>= 4) q[i++] = two[v & 15u];
        }
        if (sh.l_seq & 1u) q[n_seq - 1] &= 0xf0;
        q += n_seq;
        int ph = 25;
        for (uint32_t i = 0; i < sh.l_seq;) {
            uint64_t v = g.next();
            for (int k = 0; k < 12 && i < sh.l_seq; ++k, v >>= 5) {
                const uint32_t c = (uint32_t)(v & 31u);
                // steps -2..2 around a drift to ~Q28; 1 in 32: a drop to a low quality
                if (c == 31u) ph = 3 + (int)((v >> 5) & 7u);
                else ph += (int)(c % 5u) - 2 + (ph < 28 ? (int)(c >> 4) : -(int)(c >> 4));
                ph = ph < 2 ? 2 : (ph > 50 ? 50 : ph);
                q[i++] = (uint8_t)ph;
            }
        }
        q += sh.l_seq;
    }
    q[0] = 'N', q[1] = 'M', q[2] = 'i';
    put_at<int32_t>(q + 3, 17);
    q += 7;
    q[0] = 'M', q[1] = 'L', q[2] = 'B', q[3] = 'C';
    put_at<uint32_t>(q + 4, sh.n_mod);
    q += 8;
    for (uint32_t i = 0; i < sh.n_mod; i += 8) {
        uint64_t v = g.next();
        if (qual_mode) v |= 0xc0c0c0c0c0c0c0c0ull & (v << 1);  // methylation calls are mostly confident: skewed to the high values
        std::memcpy(q + i, &v, std::min<uint32_t>(8, sh.n_mod - i));
    }
    q += sh.n_mod;
    q[0] = 'M', q[1] = 'M', q[2] = 'Z';
    q += 3;
    std::memcpy(q, "C+m,", 4);
    q += 4;
    Rng gm = mm_rng(seed, id);
    for (uint32_t k = 0; k < sh.n_mod; ++k) {
        const uint32_t v = (uint32_t)(gm.next() % 40);
        if (v >= 10) *q++ = (uint8_t)('0' + v / 10);
        *q++ = (uint8_t)('0' + v % 10);
        if (k + 1 < sh.n_mod) *q++ = ',';
    }
    *q++ = ';';
    *q++ = 0;
    q[0] = 'H', q[1] = 'P', q[2] = 'C', q[3] = r.phase;
    q += 4;
    if ((uint64_t)(q - p) != sh.size) std::abort();  // the size pass and the writer disagree: a bug in this file
}

}  // namespace

// Arguments as inq_synth_write_bam, plus qual_mode (see above), seed, slab_blocks (0 = 4096) and two optional outputs:
// the inflated size of the file and the number of BGZF blocks.
int inq_synth_write_bam_seq(const char *bam_path, uint64_t n_reads, const Read *reads, const uint32_t *cigar, const uint64_t *order,
                            const int32_t *tid, const uint64_t *name_id, int n_contigs, uint32_t contig_len, int level, int threads,
                            int qual_mode, uint64_t seed, uint64_t slab_blocks, uint64_t *inflated_bytes, uint64_t *n_blocks_out, char *err,
                            size_t err_cap) {
    auto fail = [&](const std::string &m) {
        std::snprintf(err, err_cap, "%s", m.c_str());
        return -1;
    };
    if (threads < 1) threads = 1;
    if (!slab_blocks) slab_blocks = 4096;
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int c = 0; c < n_contigs; ++c) text += "@SQ\tSN:chr" + std::to_string(c + 1) + "\tLN:" + std::to_string(contig_len) + "\n";
    std::vector<uint8_t> head;
    head.insert(head.end(), {'B', 'A', 'M', 1});
    put<uint32_t>(head, (uint32_t)text.size());
    head.insert(head.end(), text.begin(), text.end());
    put<uint32_t>(head, (uint32_t)n_contigs);
    for (int c = 0; c < n_contigs; ++c) {
        const std::string nm = "chr" + std::to_string(c + 1);
        put<uint32_t>(head, (uint32_t)nm.size() + 1);
        head.insert(head.end(), nm.begin(), nm.end());
        head.push_back(0);
        put<uint32_t>(head, contig_len);
    }
    // ---- pass 1: shapes, extents, reference spans
    std::vector<SeqShape> shape(n_reads);
    std::vector<int64_t> beg(n_reads), end(n_reads);
    parallel_for(n_reads, threads, [&](uint64_t k) {
        const uint64_t i = order[k];
        const Read &r = reads[i];
        const uint32_t *w = cigar + (uint64_t)r.cigar_off4 * 4;
        shape[k] = seq_shape(r, w, seed, name_id[i]);
        int64_t span = 0;
        for (uint32_t c = 0; c < r.n_cigar; ++c)
            if ((0x18Du >> (w[c] & 15u)) & 1u) span += w[c] >> 4;
        beg[k] = r.pos;
        end[k] = (int64_t)r.pos + std::max<int64_t>(span, 1);
    });
    std::vector<uint64_t> u0(n_reads + 1);
    u0[0] = head.size();
    for (uint64_t k = 0; k < n_reads; ++k) u0[k + 1] = u0[k] + shape[k].size;
    const uint64_t total = u0[n_reads];
    const uint64_t n_blocks = std::max<uint64_t>(1, (total + kBlock - 1) / kBlock);
    std::vector<uint64_t> coff(n_blocks + 1, 0);
    FILE *f = std::fopen(bam_path, "wb");
    if (!f) return fail(std::string("cannot open ") + bam_path);
    std::vector<uint8_t> slab;
    std::vector<std::vector<uint8_t>> comp;
    std::atomic<int> zfail{0};
    uint64_t k_lo = 0;  // first record that reaches into the slab
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += slab_blocks) {
        const uint64_t b1 = std::min(n_blocks, b0 + slab_blocks);
        const uint64_t lo = b0 * kBlock, hi = std::min(total, b1 * kBlock);
        while (k_lo < n_reads && u0[k_lo + 1] <= lo) ++k_lo;
        uint64_t k_hi = k_lo;
        while (k_hi < n_reads && u0[k_hi] < hi) ++k_hi;
        // the slab's bytes: [base, top) covers whole records; the blocks are cut from [lo, hi)
        const uint64_t base = std::min(lo, k_lo < n_reads ? u0[k_lo] : lo), top = std::max(hi, k_hi ? u0[k_hi] : hi);
        slab.resize(top - base);
        if (lo < head.size()) std::memcpy(slab.data() + (0 - base), head.data(), head.size());  // base == 0 here
        parallel_for(k_hi - k_lo, threads, [&](uint64_t j) {
            const uint64_t k = k_lo + j, i = order[k];
            const Read &r = reads[i];
            seq_record(slab.data() + (u0[k] - base), r, cigar + (uint64_t)r.cigar_off4 * 4, tid[i], name_id[i], seed, shape[k],
                       reg2bin(beg[k], end[k]), qual_mode);
        });
        comp.assign(b1 - b0, {});
        parallel_for(b1 - b0, threads, [&](uint64_t j) {
            const uint64_t off = (b0 + j) * kBlock, len = std::min(kBlock, total - off);
            const uint8_t *src = slab.data() + (off - base);
            std::vector<uint8_t> &out = comp[j];
            out.resize(18 + compressBound((uLong)len) + 64 + 8);
            static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            std::memcpy(out.data(), hdr, 16);
            z_stream z;
            std::memset(&z, 0, sizeof z);
            if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                zfail = 1;
                return;
            }
            z.next_in = const_cast<uint8_t *>(src);
            z.avail_in = (uInt)len;
            z.next_out = out.data() + 18;
            z.avail_out = (uInt)(out.size() - 26);
            if (deflate(&z, Z_FINISH) != Z_STREAM_END) zfail = 1;
            uint64_t body = z.total_out;
            deflateEnd(&z);
            if (body + 26 > 65536) {  // incompressible data grew past 64 KB: a stored block (5 bytes of framing) always fits
                const uint16_t l16 = (uint16_t)len;
                uint8_t *o = out.data() + 18;
                o[0] = 1;
                put_at<uint16_t>(o + 1, l16);
                put_at<uint16_t>(o + 3, (uint16_t)~l16);
                std::memcpy(o + 5, src, len);
                body = 5 + len;
            }
            put_at<uint16_t>(out.data() + 16, (uint16_t)(body + 25));
            put_at<uint32_t>(out.data() + 18 + body, (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)len));
            put_at<uint32_t>(out.data() + 22 + body, (uint32_t)len);
            out.resize(26 + body);
            if (out.size() > 65536) zfail = 1;
        });
        if (zfail) {
            std::fclose(f);
            return fail("deflate failed or a block does not fit 64 KB");
        }
        for (uint64_t j = 0; j < b1 - b0; ++j) {
            coff[b0 + j + 1] = coff[b0 + j] + comp[j].size();
            if (std::fwrite(comp[j].data(), 1, comp[j].size(), f) != comp[j].size()) {
                std::fclose(f);
                return fail("short write");
            }
        }
    }
    {
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::fwrite(eof, 1, 28, f);
        if (std::fclose(f) != 0) return fail("close failed");
    }
    if (inflated_bytes) *inflated_bytes = total;
    if (n_blocks_out) *n_blocks_out = n_blocks;
    std::string e = write_bai(bam_path, n_reads, order, tid, n_contigs, u0, beg, end, coff, n_blocks);
    if (!e.empty()) return fail(e);
    return 0;
}

}  // extern "C"
