// fem_planes.hip.h — where the reference's bit planes live in HBM (shared by fem_kernels.hip.h and fem_tail.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace femk {
// The reference's bit planes (bit q of the base code, q = 0..2; plane 3: "the uploaded character is none of ACGTN"; bit
// i of a plane's byte b <-> base 8 b + i) are kept interleaved in groups of 128 bytes = one cache line: group g holds, for
// each plane q, its bytes [16 g, 16 g + 32) at g * 128 + 32 q — sixteen bytes of its own and the next group's sixteen
// again.  So the 16-byte window [at, at + 16) of every plane, at any byte offset, is ONE unaligned load inside ONE line,
// and the windows of all planes at the same `at` share that line: a candidate's or a record's reference window costs one
// fabric request where four separate planes cost four to five (unaligned 16-byte loads straddle 64-byte sectors a quarter
// of the time) — and the kernels that read them are bound by the number of those requests, not by bytes (DESIGN.md
// §4.7).  Twice the bytes of plain planes (1 byte per base): 3 GB for a 3 Gbp reference.
constexpr uint32_t kPlaneGroup = 128;
__host__ __device__ inline uint64_t plane_bytes(uint64_t n_plane_bytes) { return ((n_plane_bytes + 15u) / 16u + 2u) * kPlaneGroup; }
__host__ __device__ __forceinline__ const uint8_t *plane_addr(const uint8_t *planes, int q, uint64_t at) {
  return planes + (at >> 4) * kPlaneGroup + (uint32_t)q * 32u + ((uint32_t)at & 15u);
}
}  // namespace femk
