// sa2d.h — is_accidental_2d of the reference (src/call.rs:415-477) on a decoded record.
#pragma once
#include "bam_reader.h"

namespace inqhost {
// Returns 0 / 1, or -1 where the reference would panic (SA of a non-string type, malformed entry).
int is_accidental_2d(const BamRec &r, std::string *panic_msg);
// src/call.rs:461-477; ok=false where `num.parse::<i64>().unwrap()` would panic
int64_t cigar_string_rlen(const char *s, size_t n, bool *ok);
}  // namespace inqhost
