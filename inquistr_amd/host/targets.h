// targets.h — RepeatInterval / RepeatIntervalIterator of the reference (src/repeats.rs:4-123).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace inqhost {

struct RepeatInterval {  // src/repeats.rs:75-79
    std::string chrom;
    uint32_t start = 0, end = 0;
};

// Outcome classes of the reference: ok, or a panic (exit code 101) with its message.
struct TargetsResult {
    std::vector<RepeatInterval> data;  // RepeatIntervalIterator.data (whole BED materialised, :31-44)
    bool panicked = false;
    std::string message;
};

// RepeatIntervalIterator::from_string, src/repeats.rs:13-29
TargetsResult targets_from_string(const std::string &reg, const std::map<std::string, uint64_t> &chrom_lengths);
// RepeatIntervalIterator::from_bed, src/repeats.rs:30-45 ([3P] bio::io::bed::Reader = csv, tab
// delimited, no header row, '#' comment lines, all records the same number of fields)
TargetsResult targets_from_bed(const std::string &path, const std::map<std::string, uint64_t> &chrom_lengths);
// RepeatInterval::new_interval, src/repeats.rs:96-115: returns "" if ok, else the panic message
std::string check_interval(const std::string &chrom, uint32_t start, uint32_t end,
                           const std::map<std::string, uint64_t> &chrom_lengths);

}  // namespace inqhost
