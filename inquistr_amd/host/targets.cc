#include "targets.h"

#include <fstream>
#include <sstream>

namespace inqhost {

// [3P] Rust str::parse::<u32>(): optional '+', one or more ASCII digits, no overflow
static bool parse_u32(const std::string &s, uint32_t *out) {
    size_t i = 0;
    if (s.empty()) return false;
    if (s[0] == '+') i = 1;
    if (i == s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (uint64_t)(s[i] - '0');
        if (v > 0xffffffffull) return false;
    }
    *out = (uint32_t)v;
    return true;
}

// serde u64 from a csv field: digits only (no sign handling beyond what u64::from_str accepts)
static bool parse_u64(const std::string &s, uint64_t *out) {
    size_t i = 0;
    if (s.empty()) return false;
    if (s[0] == '+') i = 1;
    if (i == s.size()) return false;
    unsigned __int128 v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (unsigned)(s[i] - '0');
        if (v > (unsigned __int128)0xffffffffffffffffull) return false;
    }
    *out = (uint64_t)v;
    return true;
}

static std::vector<std::string> split(const std::string &s, char d) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        size_t e = s.find(d, p);
        if (e == std::string::npos) {
            out.push_back(s.substr(p));
            break;
        }
        out.push_back(s.substr(p, e - p));
        p = e + 1;
    }
    return out;
}

std::string check_interval(const std::string &chrom, uint32_t start, uint32_t end,
                           const std::map<std::string, uint64_t> &chrom_lengths) {
    if (end < start)  // src/repeats.rs:102-104
        return "End coordinate is smaller than start coordinate for " + chrom + ":" + std::to_string(start) + "-" +
               std::to_string(end);
    auto it = chrom_lengths.find(chrom);
    if (it != chrom_lengths.end() && (uint64_t)end < it->second) return "";  // :108-110
    return "Chromosome " + chrom + " is not in the fasta file or the end coordinate is out of bounds";  // :112-114
}

TargetsResult targets_from_string(const std::string &reg, const std::map<std::string, uint64_t> &chrom_lengths) {
    TargetsResult r;
    auto fail = [&](const std::string &m) {
        r.panicked = true;
        r.message = m;
        return r;
    };
    auto c = split(reg, ':');  // :14-15
    if (c.size() < 2) return fail("index out of bounds: the len is 1 but the index is 1");
    auto iv = split(c[1], '-');
    uint32_t start, end;
    if (!parse_u32(iv[0], &start)) return fail("called `Result::unwrap()` on an `Err` value: ParseIntError");  // :16-18
    if (iv.size() < 2) return fail("index out of bounds: the len is 1 but the index is 1");                     // :19
    if (!parse_u32(iv[1], &end)) return fail("called `Result::unwrap()` on an `Err` value: ParseIntError");      // :19-21
    std::string m = check_interval(c[0], start, end, chrom_lengths);
    if (!m.empty()) return fail(m);
    r.data.push_back({c[0], start, end});
    return r;
}

// One csv record: tab separated, double quotes as in RFC 4180 (the csv crate's default quoting)
static bool parse_csv_line(const std::string &line, std::vector<std::string> &f) {
    f.clear();
    std::string cur;
    size_t i = 0;
    bool at_field_start = true;
    while (i <= line.size()) {
        if (i == line.size()) {
            f.push_back(cur);
            break;
        }
        char ch = line[i];
        if (at_field_start && ch == '"') {
            ++i;
            for (;;) {
                if (i >= line.size()) return false;  // unterminated quote inside one line
                if (line[i] == '"') {
                    if (i + 1 < line.size() && line[i + 1] == '"') {
                        cur.push_back('"');
                        i += 2;
                    } else {
                        ++i;
                        break;
                    }
                } else
                    cur.push_back(line[i++]);
            }
            at_field_start = false;
            continue;
        }
        if (ch == '\t') {
            f.push_back(cur);
            cur.clear();
            at_field_start = true;
            ++i;
            continue;
        }
        cur.push_back(ch);
        at_field_start = false;
        ++i;
    }
    return true;
}

TargetsResult targets_from_bed(const std::string &path, const std::map<std::string, uint64_t> &chrom_lengths) {
    TargetsResult r;
    auto fail = [&](const std::string &m) {
        r.panicked = true;
        r.message = m;
        r.data.clear();
        return r;
    };
    std::ifstream in(path);
    if (!in) return fail("Problem reading bed file!");  // :31
    std::string line;
    std::vector<std::string> f;
    size_t n_fields = 0;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;      // csv skips empty lines
        if (line[0] == '#') continue;    // comment(Some(b'#'))
        if (!parse_csv_line(line, f)) return fail("Error reading bed record.");
        if (n_fields == 0) n_fields = f.size();
        if (f.size() != n_fields || f.size() < 3) return fail("Error reading bed record.");  // flexible(false)
        uint64_t s64, e64;
        if (!parse_u64(f[1], &s64) || !parse_u64(f[2], &e64)) return fail("Error reading bed record.");  // :35
        if (s64 > 0xffffffffull || e64 > 0xffffffffull)  // :91-92 try_into().unwrap()
            return fail("called `Result::unwrap()` on an `Err` value: TryFromIntError(())");
        std::string m = check_interval(f[0], (uint32_t)s64, (uint32_t)e64, chrom_lengths);
        if (!m.empty()) return fail(m);
        r.data.push_back({f[0], (uint32_t)s64, (uint32_t)e64});
    }
    return r;
}

}  // namespace inqhost
