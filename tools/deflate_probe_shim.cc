// Test shim: the host-side probe of inquistr_amd/csrc/deflate_probe.h behind two C symbols (tests/test_deflate_probe.py builds it
// with g++; the header has no HIP in it).
#include "../inquistr_amd/csrc/deflate_probe.h"
extern "C" int inq_probe_literal_mass(const uint8_t *payload, size_t len) { return inq::deflate_literal_mass(payload, len); }
extern "C" uint32_t inq_probe_wants_pairs(const uint8_t *comp, uint64_t comp_bytes, const inq_bgzf_block_t *blocks, uint64_t n_blocks) {
    return inq::inflate_wants_literal_pairs(comp, comp_bytes, blocks, n_blocks);
}
