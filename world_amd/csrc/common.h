// common.h -- constants and small device helpers shared by the stage kernels.
#pragma once
#include "devrt.h"
#include "fft.h"
#include "rng.h"
#include "tables.h"

namespace world_hip {

// reference src/world/constantnumbers.h:8-37 (values are part of the contract)
constexpr size_t kLdsPerCu = 160 * 1024;   // bytes of LDS a gfx950 workgroup can be given
constexpr double kPi = 3.1415926535897932384;
constexpr double kTiny = 0.000000000001;           // kMySafeGuardMinimum
constexpr double kEps = 0.00000000000000022204460492503131;
constexpr double kLog2 = 0.69314718055994529;
constexpr double kDefaultF0 = 500.0;
constexpr double kFloorF0D4C = 47.0;
constexpr double kSafeGuardD4C = 0.000001;
constexpr double kMaximumValue = 100000.0;

// matlab_round(), src/matlabfunctions.cpp:206-208: truncation of x +- 0.5
__host__ __device__ __forceinline__ int mround(double x) {
  return x > 0 ? static_cast<int>(x + 0.5) : static_cast<int>(x - 0.5);
}
constexpr int const_log2(int n) { return n <= 1 ? 0 : 1 + const_log2(n / 2); }
__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// interp1Q (src/matlabfunctions.cpp:214-235) for one query point on a uniform
// grid x0 + k*dx; `n` = number of samples in y (the slope beyond y[n-1] is 0).
__device__ __forceinline__ double interp_uniform(double x0, double dx, const double *y, int n, double xi) {
  double p = (xi - x0) / dx;
  int b = static_cast<int>(p);
  double frac = p - b;
  // the slope beyond y[n-1] is 0: the clamped neighbour gives that without a divergent branch (b <= n - 1 for every caller)
  const double y0 = y[b], y1 = y[b + 1 < n ? b + 1 : n - 1];
  return y0 + (y1 - y0) * frac;
}

// Same, with the caller's 1/dx: one FP64 division per query is the dominant cost of the
// smoothing loops.  (xi - x0) * (1/dx) differs from (xi - x0) / dx by at most an ulp, i.e. the
// interpolation weight moves by ~1e-13 -- and the interpolant is continuous across knots, so
// even a flipped bin index changes the value by no more than that.
__device__ __forceinline__ double interp_uniform_rcp(double x0, double inv_dx, const double *y, int n, double xi) {
  double p = (xi - x0) * inv_dx;
  int b = static_cast<int>(p);
  double frac = p - b;
  const double y0 = y[b], y1 = y[b + 1 < n ? b + 1 : n - 1];
  return y0 + (y1 - y0) * frac;
}

// a / b to an ulp or two where the IEEE-exact quotient is not part of the contract: the hardware reciprocal estimate, two
// Newton steps and one correction of the quotient -- 8 FP64 operations in a 6-deep chain where the compiler's exact
// division is ~25 instructions and twice as deep.  b must be a normal number (no scaling step).
__device__ __forceinline__ double fast_div(double a, double b) {
#ifndef WORLD_EMU
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
#else
  return a / b;
#endif
}

// A value passed through keep() counts as used where it is computed: the compiler's sinking pass cannot move its
// arithmetic down to a later (conditional) use.
__device__ __forceinline__ double keep(double v) {
#if !defined(WORLD_EMU) && !defined(WORLD_SIMT)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

__device__ __forceinline__ uint32_t keep_word(uint32_t v) {
#if !defined(WORLD_EMU) && !defined(WORLD_SIMT)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// Nothing is scheduled across this point.  A wavefront issues in order; where a serial recurrence is followed by
// independent work on its results, the fence keeps the compiler from weaving that work (and its own dependent
// latencies) back between the recurrence's steps.
__device__ __forceinline__ void sched_fence() {
#ifndef WORLD_EMU
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// NuttallWindow(), src/common.cpp:113-121
__device__ __forceinline__ double nuttall_at(int i, int len) {
  double t = i / (len - 1.0);
  return 0.355768 - 0.487396 * cos(2.0 * kPi * t) + 0.144232 * cos(4.0 * kPi * t) -
         0.012604 * cos(6.0 * kPi * t);
}

// One batch of utterances resident in HBM.  Every per-utterance array is dense
// and padded to the batch maximum so a (frame, utterance) grid indexes it directly.
struct BatchView {
  int n_utt;
  int fs;
  int x_stride;          // samples per utterance slot (>= max x_length)
  int f_stride;          // frames per utterance slot (>= max frame count)
  const double *x;       // [n_utt][x_stride]
  const int *x_len;      // [n_utt]
  const int *n_frames;   // [n_utt]
};

}  // namespace world_hip
