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

constexpr int kTwLog2 = 13;            // largest real FFT on the path: 8192 (StoneMask)
constexpr int kTwN = 1 << kTwLog2;
constexpr int kJumpLevels = 32;        // every 32-bit stream position (xs_jump takes a uint32_t)
constexpr int kJumpStride = 32 * 16;   // uint4 entries per level

struct Tables {
  const double2 *tw;      // [kTwN]
  const uint4 *jump;      // [kJumpLevels][32][16]
};

// host-side construction (tables.cpp)
void build_twiddles(double2 *out);                 // kTwN entries
void build_jump_tables(uint4 *out);                // kJumpLevels * kJumpStride entries

}  // namespace world_hip
