// prepare.h -- where each frame's noise sits in the reference's one randn() stream: exclusive prefix sums
// of the per-frame draw counts, one workgroup per utterance.  CheapTrick's scan and D4C's first (the LoveTrain windows)
// follow from F0 alone, so one launch serves both stages of a job (spectral_prepare, d4c.hip); D4C's second pass depends on
// the LoveTrain result: d4c_lovetrain leaves each frame's draw count, and a d4c_frame workgroup sums those of the frames
// before its own (d4c.hip).
#pragma once
#include "stage_params.h"

namespace world_hip {

// cheaptrick.cpp:218.  An F0 above fs/2 (no estimator produces one; a caller-made track can) is analysed as fs/2: the frame
// kernel's smoothing segment is laid out for widths up to 2/3 of that (ct_seg_cap) and a wider one would run over the
// next LDS region -- the one place this path does not follow the reference, which has no such bound.
// A NaN F0 (undefined behaviour in the reference: it becomes a window length through a double -> int conversion,
// cheaptrick.cpp:95) is analysed like an unvoiced frame -- `!(f0 > floor)` instead of `f0 <= floor`.
__device__ __forceinline__ double ct_effective_f0(double f0, double floor_f0, int fs) {
  const double nyq = 0.5 * fs;
  return !(f0 > floor_f0) ? kDefaultF0 : (f0 > nyq ? nyq : f0);
}
// D4C's view of a caller-made F0: NaN counts as 0 (an unvoiced frame: the row is 1 - 1e-12), values above fs/2 as fs/2.
// The reference indexes its spectra with 2 + int(f0 fft_size / fs) (common.cpp:60-62: beyond the arrays from about fs/2
// on) and turns NaN into a window length (d4c.cpp:55-56): both are undefined there; here every window length, stream
// offset and LDS index stays inside what the launch allocated.  Negative and -Inf values take the reference's own
// route (its floors of 40 and 47 Hz apply: d4c.cpp:263, :300).
__device__ __forceinline__ double d4c_sane_f0(double f0, int fs) {
  const double nyq = 0.5 * fs;
  return f0 != f0 ? 0.0 : (f0 > nyq ? nyq : f0);
}

// CheapTrick: window draws, then one per bin (cheaptrick.cpp:27-43, :147-149)
__device__ __forceinline__ void ct_offsets_utt(const CtParams &p, int u, double *scratch) {
  const int nf = p.b.n_frames[u];
  const int nb = (1 << p.lg_fft) / 2 + 1;
  const double *f0 = p.f0 + (size_t)u * p.b.f_stride;
  unsigned *off_out = p.offsets + (size_t)u * p.b.f_stride;
  unsigned running = 0;
  for (int base = 0; base < nf; base += blockDim.x) {
    int f = base + threadIdx.x;
    int cnt = 0;
    if (f < nf) {
      double cf0 = ct_effective_f0(f0[f], p.f0_floor, p.b.fs);
      cnt = 2 * mround(1.5 * p.b.fs / cf0) + 1 + nb;
    }
    int total, off = block_excl_scan_int(cnt, &total, scratch);
    if (f < nf) off_out[f] = running + (unsigned)off;
    running += (unsigned)total;
  }
}

// D4C pass 1: the LoveTrain window of every voiced frame (d4c.cpp:263-279)
__device__ __forceinline__ void d4c_offsets1_utt(const D4cParams &p, int u, double *scratch) {
  const int nf = p.b.n_frames[u];
  const double *f0 = p.f0 + (size_t)u * p.b.f_stride;
  unsigned *off_out = p.offsets1 + (size_t)u * p.b.f_stride;
  unsigned running = 0;
  for (int base = 0; base < nf; base += blockDim.x) {
    int f = base + threadIdx.x, cnt = 0;
    const double f0f = f < nf ? d4c_sane_f0(f0[f], p.b.fs) : 0.0;
    if (f < nf && f0f != 0.0) {
      double cf0 = f0f > 40.0 ? f0f : 40.0;                          // d4c.cpp:263,279
      cnt = 2 * mround(3.0 * p.b.fs / cf0 / 2.0) + 1;
    }
    int total, off = block_excl_scan_int(cnt, &total, scratch);
    if (f < nf) off_out[f] = running + (unsigned)off;
    running += (unsigned)total;
  }
  if (threadIdx.x == 0) p.draws1[u] = running;
}

}  // namespace world_hip
