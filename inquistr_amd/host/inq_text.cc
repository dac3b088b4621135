#include "inq_text.h"

#include <cmath>
#include <cstdio>

namespace inqhost {

std::string format_f64(double v) {
    if (std::isnan(v)) return "NaN";
    char buf[64];
    double ip;
    double frac = std::modf(v, &ip);
    if (frac == 0.0) {
        if (v == 0.0 && std::signbit(v)) return "-0";
        std::snprintf(buf, sizeof buf, "%.0f", v);
        return buf;
    }
    // halves are the only fractions the path produces ((a + b) as f64 / 2.0, src/call.rs:518)
    std::snprintf(buf, sizeof buf, "%s%.0f.5", v < 0 ? "-" : "", std::fabs(ip));
    return buf;
}

std::string format_row(const std::string &chrom, uint32_t start, uint32_t end, double p1, double p2) {
    return chrom + "\t" + std::to_string(start) + "\t" + std::to_string(end) + "\t" + format_f64(p1) + "\t" +
           format_f64(p2);
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
