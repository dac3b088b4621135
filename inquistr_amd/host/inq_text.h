// inq_text.h — the text side of `inquiSTR call`: .inq rows, header, sample name, output order.
#pragma once
#include <cstdint>
#include <string>

namespace inqhost {

// Rust `{}` of an f64 that is an integer, a half, or NaN (src/call.rs:57-65 prints phase1/phase2 so)
std::string format_f64(double v);
// Genotype Display, src/call.rs:57-65
// bare-pointer form of append_row for the output stage; returns the end.  dst needs chrom.size() + kRowBytesMax bytes.
constexpr size_t kRowBytesMax = 4 + 2 * 10 + 2 * 320;
char *write_row(char *dst, const std::string &chrom, uint32_t start, uint32_t end, double p1, double p2);
std::string format_row(const std::string &chrom, uint32_t start, uint32_t end, double p1, double p2);
// the same, appended to a growing buffer (no temporaries: the output of 10^5 .. 10^6 rows is on the critical path)
void append_row(std::string &out, const std::string &chrom, uint32_t start, uint32_t end, double p1, double p2);
void append_f64(std::string &out, double v);
// src/call.rs:101
std::string format_header(const std::string &sample);
// src/call.rs:91-100: file_stem, then every ".bam" and ".cram" removed
std::string sample_name_from_path(const std::string &bam_path);
// [3P] human_sort::compare as used by Genotype::cmp (src/call.rs:33-38): <0, 0, >0
int human_compare(const std::string &a, const std::string &b);

}  // namespace inqhost
