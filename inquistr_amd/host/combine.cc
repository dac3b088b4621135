// combine.cc - `inquiSTR combine` (src/combine.rs:27-59): the other half of the .inq format.
#include "driver_internal.h"

using namespace inqhost;

// ---- combine, src/combine.rs ----
namespace {
struct LineReader {
    gzFile gz = nullptr;
    FILE *fp = nullptr;
    bool open(const std::string &path, std::string *err) {
        const bool is_gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;  // Path::extension() == "gz"
        if (is_gz) {
            gz = gzopen(path.c_str(), "rb");
            if (gz) gzbuffer(gz, 128 * 1024);
        } else {
            fp = std::fopen(path.c_str(), "rb");
        }
        if (!gz && !fp) {
            *err = "couldn't open " + path;
            return false;
        }
        return true;
    }
    // BufRead::lines(): splits on '\n', strips a trailing "\r\n" or "\n"; false at end of input
    bool next(std::string &line) {
        line.clear();
        char buf[1 << 16];
        bool got = false;
        for (;;) {
            char *r = gz ? gzgets(gz, buf, sizeof buf) : std::fgets(buf, sizeof buf, fp);
            if (!r) break;
            got = true;
            size_t n = std::strlen(buf);
            line.append(buf, n);
            if (n && buf[n - 1] == '\n') break;
        }
        if (!got) return false;
        if (!line.empty() && line.back() == '\n') line.pop_back();
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
    ~LineReader() {
        if (gz) gzclose(gz);
        if (fp) std::fclose(fp);
    }
};
}  // namespace

static int inq_combine_impl(const char *const *files, size_t n_files, int out_fd, char *errbuf, size_t errcap) {
    if (!files || n_files == 0) {
        set_err(errbuf, errcap, "no input files");
        return INQ_EXIT_ERROR;
    }
    for (size_t i = 0; i < n_files; ++i) {  // src/combine.rs:29-33
        struct stat st;
        if (::stat(files[i], &st) != 0) {
            set_err(errbuf, errcap, std::string("File ") + files[i] + " does not exist!");
            return INQ_EXIT_PANIC;
        }
    }
    std::vector<std::unique_ptr<LineReader>> rd;
    for (size_t i = 0; i < n_files; ++i) {
        rd.emplace_back(new LineReader());
        std::string e;
        if (!rd.back()->open(files[i], &e)) {
            set_err(errbuf, errcap, e);
            return INQ_EXIT_PANIC;
        }
    }
    std::string line, other, out;
    while (rd[0]->next(line)) {  // :42
        out += line;
        for (size_t i = 1; i < n_files; ++i) {
            if (!rd[i]->next(other)) {  // :49 file2.next().unwrap() on None
                set_err(errbuf, errcap, std::string("called `Option::unwrap()` on a `None` value (") + files[i] +
                                            " has fewer lines than " + files[0] + ")");
                return INQ_EXIT_PANIC;
            }
            // :52-55 split('\t').skip(3): every field after the third
            size_t p = 0;
            int tabs = 0;
            while (tabs < 3) {
                size_t t = other.find('\t', p);
                if (t == std::string::npos) {
                    p = std::string::npos;
                    break;
                }
                p = t + 1;
                ++tabs;
            }
            if (p != std::string::npos) {
                out += '\t';
                out.append(other, p, std::string::npos);
            }
        }
        out += '\n';
        if (out.size() > (1u << 20)) {
            if (!write_all(out_fd, out)) return INQ_EXIT_PANIC;
            out.clear();
        }
    }
    if (!write_all(out_fd, out)) return INQ_EXIT_PANIC;
    return INQ_EXIT_OK;
}

extern "C" {
int inq_combine(const char *const *files, size_t n_files, int out_fd, char *errbuf, size_t errcap) {
    INQ_GUARD(inq_combine_impl(files, n_files, out_fd, errbuf, errcap), errbuf, errcap)
}
}  // extern "C"
