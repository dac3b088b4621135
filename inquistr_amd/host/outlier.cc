// outlier.cc — `inquiSTR outlier` (src/outlier.rs:33-73, src/main.rs:75-99,202-229): reads a combined .inq,
// hands the numbers of all loci to the GPU in one matrix (inq_outlier_rows) and prints the loci with
// outlying samples.  Text in, text out; the arithmetic is not here.
#include <zlib.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <chrono>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <string>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../../include/inquistr_host.h"

namespace {

void set_err(char *buf, size_t cap, const std::string &m) {
    if (buf && cap) std::snprintf(buf, cap, "%s", m.c_str());
}

// utils::reader (src/utils.rs:7-13): niffler sniffs the compression from the first bytes; zlib's gz layer does
// the same for gzip / plain text (bzip2, xz and zstd inputs are not supported here)
struct Lines {
    gzFile gz = nullptr;
    bool open(const char *path) {
        gz = gzopen(path, "rb");
        if (gz) gzbuffer(gz, 1 << 20);
        return gz != nullptr;
    }
    bool next(std::string &line) {  // BufRead::lines(): split on '\n', a trailing '\r' goes too
        line.clear();
        char buf[1 << 16];
        bool got = false;
        for (;;) {
            if (!gzgets(gz, buf, sizeof buf)) break;
            got = true;
            const size_t n = std::strlen(buf);
            line.append(buf, n);
            if (n && buf[n - 1] == '\n') break;
        }
        if (!got) return false;
        if (!line.empty() && line.back() == '\n') line.pop_back();
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
    ~Lines() {
        if (gz) gzclose(gz);
    }
};

void split_tabs(const std::string &s, std::vector<std::pair<size_t, size_t>> &out) {
    out.clear();
    size_t p = 0;
    for (;;) {
        const size_t t = s.find('\t', p);
        if (t == std::string::npos) {
            out.emplace_back(p, s.size() - p);
            return;
        }
        out.emplace_back(p, t - p);
        p = t + 1;
    }
}

// Rust's `str::parse::<f32>`: [+-]? ( inf | infinity | nan | digits [. digits] [e [+-] digits] ), nothing around it
bool parse_f32(const char *s, size_t n, float *out) {
    // What a combined .inq holds - integers, halves, NaN (src/call.rs:57-65, 518-520) - without strtof: up to six digits with
    // nothing, ".0" or ".5" behind them are exact in f32 (< 2^23), so any correctly rounding parser returns this value.
    {
        size_t k = 0;
        const bool neg = n && s[0] == '-';
        if (neg) k = 1;
        if (n - k == 3 && s[k] == 'N' && s[k + 1] == 'a' && s[k + 2] == 'N' && !neg) {
            *out = std::numeric_limits<float>::quiet_NaN();
            return true;
        }
        uint32_t v = 0;
        size_t d = 0;
        while (k < n && d < 7 && (unsigned)(s[k] - '0') < 10u) v = v * 10u + (uint32_t)(s[k] - '0'), ++k, ++d;
        if (d >= 1 && d <= 6) {
            float f = (float)v;
            bool ok = k == n;
            if (!ok && n - k == 2 && s[k] == '.' && (s[k + 1] == '0' || s[k + 1] == '5')) {
                if (s[k + 1] == '5') f += 0.5f;
                ok = true;
            }
            if (ok) {
                *out = neg ? -f : f;
                return true;
            }
        }
    }
    size_t i = 0;
    if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
    auto word = [&](const char *w) {
        const size_t m = std::strlen(w);
        if (n - i != m) return false;
        for (size_t k = 0; k < m; ++k)
            if ((s[i + k] | 0x20) != w[k]) return false;
        return true;
    };
    bool ok = word("inf") || word("infinity") || word("nan");
    if (!ok) {
        size_t j = i, digits = 0;
        while (j < n && s[j] >= '0' && s[j] <= '9') ++j, ++digits;
        if (j < n && s[j] == '.') {
            ++j;
            while (j < n && s[j] >= '0' && s[j] <= '9') ++j, ++digits;
        }
        if (!digits) return false;
        if (j < n && (s[j] == 'e' || s[j] == 'E')) {
            ++j;
            if (j < n && (s[j] == '+' || s[j] == '-')) ++j;
            size_t ed = 0;
            while (j < n && s[j] >= '0' && s[j] <= '9') ++j, ++ed;
            if (!ed) return false;
        }
        if (j != n) return false;
    }
    char tmp[128];
    if (n >= sizeof tmp) {  // absurdly long literal: strtof on a heap copy
        std::string t(s, n);
        *out = std::strtof(t.c_str(), nullptr);
        return true;
    }
    std::memcpy(tmp, s, n);
    tmp[n] = 0;
    *out = std::strtof(tmp, nullptr);  // correctly rounded, like Rust
    return true;
}

std::string strip_hap(std::string s) {  // .replace("_H1", "").replace("_H2", ""), src/outlier.rs:108,128
    for (const char *pat : {"_H1", "_H2"}) {
        size_t p = 0;
        while ((p = s.find(pat, p)) != std::string::npos) s.erase(p, 3);
    }
    return s;
}

bool write_all(int fd, const std::string &s) {
    size_t off = 0;
    while (off < s.size()) {
        ssize_t w = ::write(fd, s.data() + off, s.size() - off);
        if (w <= 0) return false;
        off += (size_t)w;
    }
    return true;
}

int outlier_impl(const inq_outlier_args_t *a, int out_fd, char *errbuf, size_t errcap) {
    if (!a || !a->combined) {
        set_err(errbuf, errcap, "no combined file given");
        return INQ_EXIT_ERROR;
    }
    struct stat st;
    if (::stat(a->combined, &st) != 0) {  // src/main.rs:210-212
        set_err(errbuf, errcap, "Combined file does not exist!");
        return INQ_EXIT_PANIC;
    }
    if (a->sample && a->subset_file) {  // src/main.rs:214-216
        set_err(errbuf, errcap, "Cannot use both -s and -S arguments");
        return INQ_EXIT_PANIC;
    }
    if (a->method != INQ_OUTLIER_ZSCORE && a->method != INQ_OUTLIER_DBSCAN) {
        set_err(errbuf, errcap, "unknown method");
        return INQ_EXIT_ERROR;
    }
    bool have_subset = false;
    std::vector<std::string> subset;
    if (a->sample) have_subset = true, subset.emplace_back(a->sample);
    if (a->subset_file) {  // one name per line, src/main.rs:220-223
        Lines sf;
        if (!sf.open(a->subset_file)) {
            set_err(errbuf, errcap, "Problem opening file");
            return INQ_EXIT_PANIC;
        }
        have_subset = true;
        std::string l;
        while (sf.next(l)) subset.push_back(l);
    }
    // the HIP runtime starts (0.1 - 0.3 s) while the text is read and parsed
    inq_ctx_t *ctx = nullptr;
    int hrc = INQ_OK;
    std::thread ctx_thread([&] { hrc = inq_ctx_create(a->device, &ctx); });
    struct CtxGuard {
        inq_ctx_t *&c;
        std::thread &t;
        ~CtxGuard() {
            if (t.joinable()) t.join();
            const char *fast = std::getenv("INQ_FAST_EXIT");  // set by the CLI, which is about to leave the process
            if (!(fast && fast[0] == '1')) inq_ctx_destroy(c);
        }
    } ctx_guard{ctx, ctx_thread};
    // the whole text in memory, then lines parsed by several threads: the numbers are what the command spends its
    // time on (40 million of them in a 200 000 x 200 cohort), the GPU part is milliseconds.  A plain file is mapped (no copy:
    // the page cache is the buffer; round 2 pulled 139 MB through gzread, 0.1 s on one thread), a gzip file inflated.
    using clk = std::chrono::steady_clock;
    const bool timing = std::getenv("INQ_TIMING") != nullptr;
    const auto t_begin = clk::now();
    auto lap = [&](const char *what) {
        if (timing) std::fprintf(stderr, "[inq outlier] %-22s at %7.1f ms\n", what, std::chrono::duration<double, std::milli>(clk::now() - t_begin).count());
    };
    std::string text;
    const char *tdata = nullptr;
    size_t tsize = 0;
    struct Mapping {
        void *p = MAP_FAILED;
        size_t len = 0;
        ~Mapping() {
            if (p != MAP_FAILED) ::munmap(p, len);
        }
    } mapping;
    {
        int fd = ::open(a->combined, O_RDONLY);
        unsigned char magic[2] = {0, 0};
        struct stat st;
        const bool plain = fd >= 0 && ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && ::pread(fd, magic, 2, 0) == 2 &&
                           !(magic[0] == 0x1f && magic[1] == 0x8b);
        if (plain) {
            mapping.len = (size_t)st.st_size;
            mapping.p = ::mmap(nullptr, mapping.len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (mapping.p != MAP_FAILED) {
                (void)::madvise(mapping.p, mapping.len, MADV_WILLNEED);
                tdata = (const char *)mapping.p;
                tsize = mapping.len;
            }
        }
        if (fd >= 0) ::close(fd);
    }
    if (!tdata) {
        Lines in;
        if (!in.open(a->combined)) {
            set_err(errbuf, errcap, "Problem opening file");
            return INQ_EXIT_PANIC;
        }
        char buf[1 << 16];
        int got;
        while ((got = gzread(in.gz, buf, sizeof buf)) > 0) text.append(buf, (size_t)got);
        if (got < 0) {
            set_err(errbuf, errcap, "Problem reading file");
            return INQ_EXIT_PANIC;
        }
        tdata = text.data();
        tsize = text.size();
    }
    lap("text in memory");
    // BufRead::lines(): split on '\n' (a trailing '\r' goes too); no line after a final newline.  The newlines are found by
    // several threads, each over its own stretch of the text.
    std::vector<std::pair<size_t, size_t>> lines;  // (offset, length)
    {
        const unsigned hw0 = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const size_t parts = std::max<size_t>(1, std::min<size_t>(hw0, tsize / (4u << 20) + 1));
        std::vector<std::vector<size_t>> nl(parts);  // positions of '\n' per stretch
        std::vector<std::thread> th;
        auto scan = [&](size_t t) {
            const size_t lo = tsize * t / parts, hi = tsize * (t + 1) / parts;
            nl[t].reserve((hi - lo) / 256 + 16);
            for (size_t p = lo; p < hi;) {
                const char *q = (const char *)std::memchr(tdata + p, '\n', hi - p);
                if (!q) break;
                nl[t].push_back((size_t)(q - tdata));
                p = (size_t)(q - tdata) + 1;
            }
        };
        for (size_t t = 1; t < parts; ++t) th.emplace_back(scan, t);
        scan(0);
        for (auto &x : th) x.join();
        size_t total = 1;
        for (auto &v : nl) total += v.size();
        lines.reserve(total);
        size_t p = 0;
        auto push = [&](size_t e) {
            size_t len = e - p;
            if (len && tdata[p + len - 1] == '\r') --len;
            lines.emplace_back(p, len);
            p = e + 1;
        };
        for (auto &v : nl)
            for (size_t e : v) push(e);
        if (p < tsize) push(tsize);
    }
    lap("lines found");
    if (lines.empty()) {  // lines.next().unwrap(), src/outlier.rs:36
        set_err(errbuf, errcap, "called `Option::unwrap()` on a `None` value (empty combined file)");
        return INQ_EXIT_PANIC;
    }
    std::string line(tdata + lines[0].first, lines[0].second);
    std::vector<std::pair<size_t, size_t>> fld;
    split_tabs(line, fld);
    std::vector<std::string> samples;
    for (size_t k = 3; k < fld.size(); ++k) samples.push_back(line.substr(fld[k].first, fld[k].second));
    if (samples.empty()) {  // samples.len().ilog2(), :39; the header went out one line earlier (:37)
        write_all(out_fd, "chrom\tbegin\tend\toutliers\n");
        set_err(errbuf, errcap, "argument of integer logarithm must be positive (no sample columns)");
        return INQ_EXIT_PANIC;
    }
    uint32_t mincluster = 0;
    for (size_t n = samples.size(); n > 1; n >>= 1) ++mincluster;
    for (auto &s : samples) s = strip_hap(s);

    // The reference works line by line: what it printed before a panic stays printed.  The first line that makes
    // it panic (fewer than three fields, a number that does not parse) ends the input here; the lines in front
    // of it are still computed and printed.
    const size_t n_lines = lines.size() - 1;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t n_thr = std::max<size_t>(1, std::min<size_t>(hw, n_lines / 2048 + 1));
    std::vector<uint32_t> row_len(n_lines, 0);
    std::vector<size_t> bad_a(n_thr, (size_t)-1);
    auto chunk = [&](size_t t) { return std::make_pair(n_lines * t / n_thr, n_lines * (t + 1) / n_thr); };
    auto parallel = [&](const std::function<void(size_t)> &fn) {
        std::vector<std::thread> th;
        for (size_t t = 1; t < n_thr; ++t) th.emplace_back(fn, t);
        fn(0);
        for (auto &x : th) x.join();
    };
    parallel([&](size_t t) {  // pass A: fields per line
        auto [lo, hi] = chunk(t);
        for (size_t i = lo; i < hi; ++i) {
            const char *p = tdata + lines[i + 1].first;
            size_t tabs = 0;
            for (size_t k = 0; k < lines[i + 1].second; ++k) tabs += p[k] == '\t';
            if (tabs < 2) {  // splitline[2], :43
                bad_a[t] = i;
                return;
            }
            row_len[i] = (uint32_t)(tabs - 2);
        }
    });
    size_t first_bad = n_lines;
    std::string late_panic;
    for (size_t t = 0; t < n_thr; ++t)
        if (bad_a[t] < first_bad) first_bad = bad_a[t], late_panic = "index out of bounds: a line with fewer than three fields";
    size_t stride = 0;
    for (size_t i = 0; i < first_bad; ++i) stride = std::max<size_t>(stride, row_len[i]);
    // not zero-filled: a row's values beyond row_len are never read (inq_outlier_rows), and 160 MB of zeros is time
    std::unique_ptr<float[]> mat_mem(new float[std::max<size_t>(first_bad * stride, 1)]);
    float *const mat = mat_mem.get();
    std::vector<size_t> bad_b(n_thr, (size_t)-1);
    parallel([&](size_t t) {  // pass B: the numbers
        auto [lo, hi] = chunk(t);
        hi = std::min(hi, first_bad);
        for (size_t i = lo; i < hi; ++i) {
            const char *p = tdata + lines[i + 1].first, *end = p + lines[i + 1].second;
            int field = 0;
            const char *f0 = p;
            for (const char *q = p;; ++q) {
                if (q == end || *q == '\t') {
                    if (field >= 3 && !parse_f32(f0, (size_t)(q - f0), &mat[i * stride + (size_t)(field - 3)])) {  // :78
                        bad_b[t] = i;
                        return;
                    }
                    ++field;
                    f0 = q + 1;
                    if (q == end) break;
                }
            }
        }
    });
    for (size_t t = 0; t < n_thr; ++t)
        if (bad_b[t] < first_bad) first_bad = bad_b[t], late_panic = "Failed to parse number";
    row_len.resize(first_bad);
    auto coords_of = [&](size_t i) {  // chrom, begin, end as they stand in the file
        const char *p = tdata + lines[i + 1].first;
        size_t k = 0, tabs = 0;
        for (; k < lines[i + 1].second; ++k)
            if (p[k] == '\t' && ++tabs == 3) break;
        return std::string(p, k);
    };
    const size_t n_rows = first_bad;
    std::vector<uint8_t> flags(n_rows * stride, 0), keep(n_rows, 0);

    lap("numbers parsed");
    ctx_thread.join();
    lap("device context there");
    if (hrc != INQ_OK) {
        set_err(errbuf, errcap, std::string("cannot open HIP device: ") + inq_strerror(hrc));
        return INQ_EXIT_ERROR;
    }
    hrc = inq_outlier_rows(ctx, mat, row_len.data(), n_rows, (uint32_t)stride, a->method, a->minsize, a->zscore,
                           mincluster, flags.data(), keep.data());
    std::string detail = hrc == INQ_ERR_HIP ? inq_last_error(ctx) : "";
    if (hrc != INQ_OK) {
        set_err(errbuf, errcap, std::string("device call failed: ") + inq_strerror(hrc) + (detail.empty() ? "" : " [" + detail + "]"));
        return INQ_EXIT_ERROR;
    }

    lap("kernels done");
    std::string out = "chrom\tbegin\tend\toutliers\n";  // :37
    for (size_t i = 0; i < n_rows; ++i) {
        switch (keep[i]) {
        case INQ_OUTLIER_ROW_SKIP: continue;
        case INQ_OUTLIER_ROW_EMPTY:
            write_all(out_fd, out);
            set_err(errbuf, errcap, "called `Option::unwrap()` on a `None` value (a locus without values)");
            return INQ_EXIT_PANIC;
        case INQ_OUTLIER_ROW_NO_MODE:
            write_all(out_fd, out);
            set_err(errbuf, errcap, "No mode found for repeat");
            return INQ_EXIT_PANIC;
        case INQ_OUTLIER_ROW_TOO_WIDE:
            set_err(errbuf, errcap, "DBSCAN on more than 8192 values per locus is not supported");
            return INQ_EXIT_ERROR;
        default: break;
        }
        std::string names;
        bool any = false, in_subset = !have_subset;
        for (uint32_t k = 0; k < row_len[i]; ++k) {
            if (!flags[i * stride + k]) continue;
            if (k >= samples.size()) {  // samples[index], :108
                write_all(out_fd, out);
                set_err(errbuf, errcap, "index out of bounds: more values than samples in the header");
                return INQ_EXIT_PANIC;
            }
            if (any) names += ',';
            names += samples[k];
            any = true;
            if (have_subset && !in_subset)
                for (const auto &s : subset)
                    if (s == samples[k]) in_subset = true;
        }
        if (any && in_subset) {  // :47-66
            out += coords_of(i);
            out += '\t';
            out += names;
            out += '\n';
        }
        if (out.size() > (1u << 20)) {
            if (!write_all(out_fd, out)) return INQ_EXIT_PANIC;
            out.clear();
        }
    }
    if (!write_all(out_fd, out)) return INQ_EXIT_PANIC;
    if (!late_panic.empty()) {
        set_err(errbuf, errcap, late_panic);
        return INQ_EXIT_PANIC;
    }
    return INQ_EXIT_OK;
}

}  // namespace

extern "C" int inq_outlier(const inq_outlier_args_t *args, int out_fd, char *errbuf, size_t errcap) {
    try {
        return outlier_impl(args, out_fd, errbuf, errcap);
    } catch (const std::exception &e) {
        set_err(errbuf, errcap, std::string("internal error: ") + e.what());
        return INQ_EXIT_ERROR;
    } catch (...) {
        set_err(errbuf, errcap, "internal error");
        return INQ_EXIT_ERROR;
    }
}
