// tables.h -- read-only lookup tables built once on the host and kept in HBM
// (they are small enough to stay resident in L2/Infinity Cache):
//   * FFT twiddles  tw[k] = (cos, sin)(2 pi k / kTwN), k < kTwN
//   * xorshift128 jump-ahead tables: for j < kJumpLevels the GF(2) matrix of
//     "advance the generator by 2^j randn() calls" (one randn() = 12 xorshift
//     steps, reference src/matlabfunctions.cpp:244-264), stored as 32 nibble
//     tables of 16 x 128-bit entries so a matrix-vector product is 32 lookups.
#pragma once
#include "devrt.h"

namespace world_hip {

constexpr int kTwLog2 = 14;            // largest real FFT on the path: 16384 (D4C and its LoveTrain pass above 96 kHz)
constexpr int kTwN = 1 << kTwLog2;
constexpr int kJumpLevels = 32;        // every 32-bit stream position (xs_jump takes a uint32_t)
constexpr int kJumpStride = 32 * 16;   // uint4 entries per level

// Behind the kTwN pairs the same allocation holds, for every transform size 2^lg (2 <= lg <= kTwLog2), the DENSE
// quarter-wave cosine table a kernel stages in LDS (fft.h: stage_twiddles): q_lg[r] = cos(2 pi r / 2^lg),
// r = 0 .. 2^lg / 4 -- the same doubles as tw[r << (kTwLog2 - lg)].x, contiguous, so that staging is a coalesced
// copy instead of a strided gather (a lane per 64-byte line: 2.3 thousand cycles at the head of every FFT workgroup).
constexpr __host__ __device__ int quarter_table_offset(int lg) { return ((1 << (lg - 2)) - 1) + (lg - 2); }   // doubles before q_lg
constexpr int kQuarterDoubles = quarter_table_offset(kTwLog2 + 1);
constexpr int kTwAlloc = kTwN + (kQuarterDoubles + 1) / 2;   // double2 entries of the allocation

struct Tables {
  const double2 *tw;      // [kTwAlloc]: kTwN twiddles, then the dense quarter tables
  const uint4 *jump;      // [kJumpLevels][32][16]
};

// host-side construction (tables.cpp)
void build_twiddles(double2 *out);                 // kTwAlloc entries
void build_jump_tables(uint4 *out);                // kJumpLevels * kJumpStride entries

}  // namespace world_hip
