// rng.h -- the reference's randn() stream, reproduced exactly and in parallel.
//
// randn() (reference src/matlabfunctions.cpp:237-264) is xorshift128 advanced 12
// times per call, the 12 outputs' top 28 bits summed and recentred.  CheapTrick
// and D4C add this noise to every windowed sample, so the stream position of
// every frame is part of the numerical contract.  The generator
// is linear over GF(2): jumping ahead by k calls is a 128x128 bit-matrix applied
// to the state; tables.cpp precomputes the matrices for k = 2^j as nibble tables.
#pragma once
#include "devrt.h"
#include "tables.h"

namespace world_hip {

struct Xs128 { uint32_t x, y, z, w; };

__device__ __forceinline__ Xs128 xs_seed() {           // randn_reseed(), :237-242
  Xs128 s;
  s.x = 123456789u; s.y = 362436069u; s.z = 521288629u; s.w = 88675123u;
  return s;
}

__device__ __forceinline__ uint32_t xs_step(Xs128 &s) {
  uint32_t t = s.x ^ (s.x << 11);
  s.x = s.y; s.y = s.z; s.z = s.w;
  s.w = (s.w ^ (s.w >> 19)) ^ (t ^ (t >> 8));
  return s.w;
}

// randn(), :244-264, in two halves: the integer the 12 steps add up to (< 12 * 2^28, fits 32 bits) ...
__device__ __forceinline__ uint32_t xs_randn_word(Xs128 &s) {
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k) acc += xs_step(s) >> 4;
  return acc;
}
// ... and its recentring, acc / 2^28 - 6: the scaling is exact, so the one fma rounds exactly as the
// reference's divide-then-subtract does.
__device__ __forceinline__ double randn_value(uint32_t word) { return fma(static_cast<double>(word), 0x1p-28, -6.0); }
__device__ __forceinline__ double xs_randn(Xs128 &s) { return randn_value(xs_randn_word(s)); }

// state <- M_level * state  (advance by 2^level randn() calls)
__device__ __forceinline__ Xs128 xs_jump_level(const uint4 *jump, Xs128 s, int level) {
  const uint4 *tab = jump + (size_t)level * kJumpStride;
  uint32_t in[4] = {s.x, s.y, s.z, s.w};
  uint32_t a = 0, b = 0, c = 0, d = 0;
#pragma unroll
  for (int wd = 0; wd < 4; ++wd) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      uint4 e = tab[(wd * 8 + n) * 16 + ((in[wd] >> (4 * n)) & 15u)];
      a ^= e.x; b ^= e.y; c ^= e.z; d ^= e.w;
    }
  }
  Xs128 r; r.x = a; r.y = b; r.z = c; r.w = d;
  return r;
}

// advance by an arbitrary number of randn() calls
__device__ __forceinline__ Xs128 xs_jump(const uint4 *jump, Xs128 s, uint32_t calls) {
  for (int level = 0; calls != 0; ++level, calls >>= 1)
    if (calls & 1u) s = xs_jump_level(jump, s, level);
  return s;
}

// Whole-stream generation (rng_fill.hip): thread t produces draws
// [begin + t*kFillRun, begin + (t+1)*kFillRun) of the stream with ONE jump-ahead, so the
// jump cost is amortised over kFillRun draws and every draw of the stream is produced
// exactly once, in parallel.  The table holds the draws' 32-bit integer sums (4 bytes per
// draw, not 8: it is streamed once per frame by CheapTrick and D4C and is their largest
// input); consumers recentre with randn_value() and apply their own scale (1e-12, 1e-6,
// |.|*eps).
constexpr int kFillRun = 32;
struct RngFillArgs {
  uint32_t *noise;          // randn_value(noise[k]) = draw number k of the stream
  size_t begin, end;        // positions to (re)generate
  const uint4 *jump;
};
void launch_rng_fill(const RngFillArgs &a, hipStream_t stream);

// The stream is the same for every utterance, every call and every context (the reference
// reseeds at the top of CheapTrick(), D4C() and Synthesis()), so a PROCESS keeps ONE table
// noise[k] = k-th randn() per device, shared by all contexts of that device and only ever
// replaced by a longer one.  Every table is checked IN FULL before it is published: the fill
// kernel's output is reduced to one (sum, index-weighted sum) pair per kNoiseChunk draws on
// the device and compared with the same pairs computed on the host by stepping the generator
// sequentially from the seed (no jump tables involved) -- a wrong word anywhere stops the
// call instead of shifting D4C / CheapTrick results by 1e-4.  Superseded tables stay
// allocated until the device's last context is destroyed (other contexts' kernels may still
// be reading them); with doubling growth they add up to less than the live one.
constexpr size_t kNoiseChunk = (size_t)1 << 16;
constexpr size_t kNoiseMaxDraws = 0xFFFFFFF0ull;   // stream positions are 32-bit (xs_jump)
void noise_table_retain(int device);
void noise_table_release(int device);              // last release frees the device's tables
// first `draws` words resident, verified and visible to `stream`; throws std::runtime_error
const uint32_t *noise_table_acquire(int device, size_t draws, const uint4 *d_jump, hipStream_t stream);
// re-reduce the live table on `stream` and compare with the host's sums (diagnostic; synchronises); throws on mismatch
void noise_table_verify(int device, hipStream_t stream);
size_t noise_table_bytes(int device);              // live + superseded tables
// host wall time spent building + verifying this device's tables so far (every generation), and how many were built
double noise_table_build_ms(int device, int *builds);

}  // namespace world_hip
