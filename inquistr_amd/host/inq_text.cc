#include "inq_text.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace inqhost {

// decimal digits of a non-negative integer, appended
static void append_u64(std::string &out, uint64_t v) {
    char buf[24];
    int n = 0;
    do {
        buf[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) out.push_back(buf[--n]);
}

// Rust's `{}` for f64 restricted to what the path produces: NaN, integers and halves ((a + b) as f64 / 2.0,
// src/call.rs:518) of at most 2^53 in magnitude take the fast path; anything else goes the general way below
void append_f64(std::string &out, double v) {
    if (std::isnan(v)) {
        out += "NaN";
        return;
    }
    const double a = std::fabs(v);
    if (a < 9007199254740992.0) {
        const uint64_t twice = (uint64_t)(a * 2.0);
        if ((double)twice == a * 2.0) {  // an integer or a half
            if (std::signbit(v)) out.push_back('-');
            append_u64(out, twice >> 1);
            if (twice & 1u) out += ".5";
            return;
        }
    }
    // anything else - never produced by the path - as Rust's Display prints it [3P core::fmt::float]: the SHORTEST decimal digits that
    // read back as v, written out positionally (no exponent; 2^60 is "1152921504606847000", not its exact digits), "inf", "-0"
    if (std::isinf(v)) {
        out += v < 0 ? "-inf" : "inf";
        return;
    }
    char e[40];
    int prec = 0;
    for (; prec <= 16; ++prec) {
        std::snprintf(e, sizeof e, "%.*e", prec, a);
        if (std::strtod(e, nullptr) == a) break;
    }
    if (prec > 16) std::snprintf(e, sizeof e, "%.16e", a);
    std::string digits;
    int exp10 = 0;
    for (const char *q = e; *q; ++q) {
        if (*q >= '0' && *q <= '9') digits.push_back(*q);
        else if (*q == 'e') {
            exp10 = std::atoi(q + 1);
            break;
        }
    }
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    if (std::signbit(v)) out.push_back('-');
    const int nd = (int)digits.size();
    if (exp10 >= nd - 1) {
        out += digits;
        out.append((size_t)(exp10 - (nd - 1)), '0');
    } else if (exp10 >= 0) {
        out.append(digits, 0, (size_t)exp10 + 1);
        out.push_back('.');
        out.append(digits, (size_t)exp10 + 1, std::string::npos);
    } else {
        out += "0.";
        out.append((size_t)(-exp10 - 1), '0');
        out += digits;
    }
}

std::string format_f64(double v) {
    std::string s;
    append_f64(s, v);
    return s;
}

void append_row(std::string &out, const std::string &chrom, uint32_t start, uint32_t end, double p1, double p2) {
    out += chrom;
    out.push_back('\t');
    append_u64(out, start);
    out.push_back('\t');
    append_u64(out, end);
    out.push_back('\t');
    append_f64(out, p1);
    out.push_back('\t');
    append_f64(out, p2);
}

// the same text through a bare pointer (the output stage formats 10^5 .. 10^6 rows): dst must hold row_capacity(chrom) bytes
static inline char *put_u64(char *p, uint64_t v) {
    char buf[24];
    int n = 0;
    do {
        buf[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = buf[--n];
    return p;
}
static inline char *put_f64(char *p, double v) {
    if (std::isnan(v)) {
        *p++ = 'N', *p++ = 'a', *p++ = 'N';
        return p;
    }
    const double a = std::fabs(v);
    if (a < 9007199254740992.0) {
        const uint64_t twice = (uint64_t)(a * 2.0);
        if ((double)twice == a * 2.0) {
            if (std::signbit(v)) *p++ = '-';
            p = put_u64(p, twice >> 1);
            if (twice & 1u) *p++ = '.', *p++ = '5';
            return p;
        }
    }
    std::string s;
    append_f64(s, v);  // the rare rest (not produced by the path): up to 1 + 309 digits + ".5"
    for (char c : s) *p++ = c;
    return p;
}
char *write_row(char *p, const std::string &chrom, uint32_t start, uint32_t end, double p1, double p2) {
    for (char c : chrom) *p++ = c;
    *p++ = '\t';
    p = put_u64(p, start);
    *p++ = '\t';
    p = put_u64(p, end);
    *p++ = '\t';
    p = put_f64(p, p1);
    *p++ = '\t';
    p = put_f64(p, p2);
    return p;
}

std::string format_row(const std::string &chrom, uint32_t start, uint32_t end, double p1, double p2) {
    std::string s;
    append_row(s, chrom, start, end, p1, p2);
    return s;
}

std::string format_header(const std::string &sample) {
    return "chromosome\tbegin\tend\t" + sample + "_H1\t" + sample + "_H2";
}

static void erase_all(std::string &s, const std::string &pat) {
    size_t p = 0;
    while ((p = s.find(pat, p)) != std::string::npos) s.erase(p, pat.size());
}

std::string sample_name_from_path(const std::string &bam_path) {
    std::string p = bam_path;
    while (p.size() > 1 && p.back() == '/') p.pop_back();
    size_t slash = p.rfind('/');
    std::string name = slash == std::string::npos ? p : p.substr(slash + 1);
    if (name != "..") {
        size_t dot = name.rfind('.');
        if (dot != std::string::npos && dot != 0) name = name.substr(0, dot);
    }
    erase_all(name, ".bam");
    erase_all(name, ".cram");
    return name;
}

int human_compare(const std::string &a, const std::string &b) {
    size_t i = 0, j = 0;
    auto digit = [](char c) { return c >= '0' && c <= '9'; };
    while (i < a.size() && j < b.size()) {
        if (digit(a[i]) && digit(b[j])) {
            unsigned __int128 x = 0, y = 0;
            while (i < a.size() && digit(a[i])) x = x * 10 + (unsigned)(a[i++] - '0');
            while (j < b.size() && digit(b[j])) y = y * 10 + (unsigned)(b[j++] - '0');
            if (x != y) return x < y ? -1 : 1;
        } else {
            unsigned char ca = (unsigned char)a[i], cb = (unsigned char)b[j];
            if (ca != cb) return ca < cb ? -1 : 1;
            ++i;
            ++j;
        }
    }
    if (i < a.size()) return 1;
    if (j < b.size()) return -1;
    return 0;
}

}  // namespace inqhost
