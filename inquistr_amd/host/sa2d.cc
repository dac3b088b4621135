#include "sa2d.h"

#include <cstring>

namespace inqhost {

static bool parse_i64(const char *s, size_t n, int64_t *out) {
    size_t i = 0;
    bool neg = false;
    if (n == 0) return false;
    if (s[0] == '+' || s[0] == '-') {
        neg = s[0] == '-';
        i = 1;
    }
    if (i == n) return false;
    unsigned __int128 v = 0;
    for (; i < n; ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (unsigned)(s[i] - '0');
        if (v > ((unsigned __int128)1 << 63)) return false;
    }
    if (!neg && v > (unsigned __int128)INT64_MAX) return false;
    *out = neg ? (int64_t)(0 - (uint64_t)v) : (int64_t)v;
    return true;
}

int64_t cigar_string_rlen(const char *s, size_t n, bool *ok) {
    *ok = true;
    int64_t rlen = 0;
    size_t num_start = 0, num_len = 0;
    for (size_t i = 0; i < n; ++i) {
        char c = s[i];
        if (c >= '0' && c <= '9') {
            if (num_len == 0) num_start = i;
            ++num_len;
        } else {
            int64_t v;
            if (!parse_i64(s + num_start, num_len, &v)) {  // src/call.rs:469
                *ok = false;
                return 0;
            }
            if (c == 'M' || c == '=' || c == 'X' || c == 'D' || c == 'N') rlen += v;
            num_len = 0;
        }
    }
    return rlen;
}

int is_accidental_2d(const BamRec &r, std::string *panic_msg) {
    const char read_strand = (r.flag & 0x10) ? '-' : '+';  // src/call.rs:422
    if (r.sa_type == 0) return 0;                          // :425-427
    if (r.sa_type != 'Z') {                                // :429-432
        if (panic_msg) *panic_msg = "Unexpected type of Aux";
        return -1;
    }
    // :434 entries separated by ';', empty ones dropped
    const char *s = r.sa;
    const char *first = nullptr;
    size_t first_len = 0;
    int n_entries = 0;
    for (;;) {
        const char *e = std::strchr(s, ';');
        size_t len = e ? (size_t)(e - s) : std::strlen(s);
        if (len) {
            if (!n_entries) first = s, first_len = len;
            ++n_entries;
        }
        if (!e) break;
        s = e + 1;
    }
    if (n_entries > 1) return 0;  // :436-438
    if (n_entries == 0) {
        if (panic_msg) *panic_msg = "index out of bounds: the len is 0 but the index is 0";
        return -1;
    }
    // :439 rname,POS,strand,CIGAR,mapQ,NM
    const char *fld[6];
    size_t flen[6];
    int nf = 0;
    const char *p = first, *end = first + first_len, *start = first;
    for (;; ++p) {
        if (p == end || *p == ',') {
            if (nf < 6) fld[nf] = start, flen[nf] = (size_t)(p - start);
            ++nf;
            start = p + 1;
            if (p == end) break;
        }
    }
    if (nf < 3 || flen[2] == 0) {  // :441
        if (panic_msg) *panic_msg = "malformed SA entry";
        return -1;
    }
    if (read_strand == fld[2][0]) return 0;  // :441-443
    const int64_t rs = r.pos, re = bam_endpos(r);  // :448-449
    int64_t sa_start;
    if (!parse_i64(fld[1], flen[1], &sa_start)) {  // :450
        if (panic_msg) *panic_msg = "called `Result::unwrap()` on an `Err` value: ParseIntError";
        return -1;
    }
    if (nf < 4) {
        if (panic_msg) *panic_msg = "index out of bounds: the len is 3 but the index is 3";
        return -1;
    }
    bool ok;
    const int64_t sa_end = sa_start + cigar_string_rlen(fld[3], flen[3], &ok);  // :451
    if (!ok) {
        if (panic_msg) *panic_msg = "called `Result::unwrap()` on an `Err` value: ParseIntError";
        return -1;
    }
    const int64_t lo = rs > sa_start ? rs : sa_start, hi = re < sa_end ? re : sa_end;
    return lo < hi ? 1 : 0;  // :454-458
}

}  // namespace inqhost
