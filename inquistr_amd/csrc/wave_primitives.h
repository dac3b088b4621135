// wave_primitives.h — wave64 cross-lane building blocks for gfx950 (CDNA4).
//
// Everything here is wave-wide (64 lanes) and uses DPP row operations directly
// (`__builtin_amdgcn_update_dpp`), not LDS and not 32-wide shuffle idioms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace inq {

constexpr int kWave = 64;

// DPP control words (GFX9 encoding)
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_SHR2 = 0x112;
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int DPP_ROW_BCAST31 = 0x143;  // lane 31 -> every lane of rows 2 and 3

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// x + (x shifted right by N lanes inside a 16-lane row, 0 shifted in)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_shr_zero(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_bcast(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}

// Inclusive prefix sum over the 64 lanes, wrapping u32 arithmetic.  6 DPP adds.
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t x) {
    x += dpp_shr_zero<DPP_ROW_SHR1>(x);
    x += dpp_shr_zero<DPP_ROW_SHR2>(x);
    x += dpp_shr_zero<DPP_ROW_SHR4>(x);
    x += dpp_shr_zero<DPP_ROW_SHR8>(x);
    x += dpp_bcast<DPP_ROW_BCAST15, 0xa>(x);  // row0 total -> row1, row2 total -> row3
    x += dpp_bcast<DPP_ROW_BCAST31, 0xc>(x);  // rows0-1 total -> rows 2,3
    return x;
}

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int l) {
    uint32_t lo = readlane_u32((uint32_t)v, l);
    uint32_t hi = readlane_u32((uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// v_writelane_b32 through the LLVM intrinsic (this clang has no __builtin_amdgcn_writelane): the value and the
// lane select are wave-uniform, register allocation and M0 are the compiler's business
extern "C" __device__ int inq_llvm_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t writelane_u32(uint32_t old, uint32_t value, int l) {
    return (uint32_t)inq_llvm_writelane_i32((int)value, l, (int)old);
}
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

}  // namespace inq
