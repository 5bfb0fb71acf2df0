// stonemask.hip -- StoneMask F0 refinement on gfx950 (reference src/stonemask.cpp:24-218).
//
// The reference builds a Blackman window and its central difference around each
// frame, takes two zero-padded r2c FFTs of 512..8192 points (a new FFT plan per
// frame) and then reads at most 2 + 6 harmonic bins.  Here one 256-thread
// workgroup per frame keeps the window in LDS and evaluates just those bins as
// direct DFT sums (twiddles from the shared table, exact integer phase index).
#include "dio.h"

namespace world_hip {

constexpr double kFloorF0StoneMask = 40.0;

// X[idx] of (x .* main window) and (x .* diff window) for `nh` harmonics of f0:
// returns power and "numerator_i" per harmonic (stonemask.cpp:159-164) in pw/ni.
__device__ __forceinline__ void sm_harmonic_bins(const double *x, int x_len, const double *mw, const signed char *rawd, int raw0,
                                                 int blen, int lgN, double f0, int fs, int nh,
                                                 const double2 *tw, double *scratch, double *pw, double *ni) {
  const int N = 1 << lgN, tid = threadIdx.x, nt = blockDim.x;
  for (int h = 0; h < nh; ++h) {
    const int idx = imin(mround(f0 * N / fs * (h + 1)), N / 2);     // FixF0, stonemask.cpp:102-103
    double are = 0, aim = 0, dre = 0, dim = 0;
    for (int i = tid; i < blen; i += nt) {
      double dwv;                                                  // GetDiffWindow (:49-55)
      if (i == 0) dwv = -mw[1] / 2.0;
      else if (i == blen - 1) dwv = mw[blen - 2] / 2.0;
      else dwv = -(mw[i + 1] - mw[i - 1]) / 2.0;
      const double xv = x[imax(0, imin(x_len - 1, raw0 + i + rawd[i] - 1))];   // GetSpectra (:67-70)
      const double a = xv * mw[i], d = xv * dwv;
      // e^{-2 pi i idx i / N}: phase reduced exactly in integers, then the table -- or, for the
      // transforms beyond its resolution (f0 < 70 Hz above 48 kHz), sincospi of the exact fraction
      const long long ph = ((long long)idx * i) & (N - 1);
      double2 w;
      if (lgN <= kTwLog2) w = tw[(size_t)ph << (kTwLog2 - lgN)];
      else { double sn, cs; sincospi(2.0 * static_cast<double>(ph) / N, &sn, &cs); w.x = cs; w.y = sn; }
      are = fma(a, w.x, are); aim = fma(-a, w.y, aim);
      dre = fma(d, w.x, dre); dim = fma(-d, w.y, dim);
    }
    block_sum2(are, aim, scratch);
    block_sum2(dre, dim, scratch);
    pw[h] = are * are + aim * aim;
    ni[h] = are * dim - aim * dre;
  }
}

// FixF0 (stonemask.cpp:96-118)
__device__ __forceinline__ double sm_fix_f0(const double *pw, const double *ni, int lgN, int fs, double f0, int nh) {
  const int N = 1 << lgN;
  double num = 0.0, den = 0.0;
  for (int h = 0; h < nh; ++h) {
    const int idx = imin(mround(f0 * N / fs * (h + 1)), N / 2);
    double inst = pw[h] == 0.0 ? 0.0 : static_cast<double>(idx) * fs / N + ni[h] / pw[h] * fs / 2.0 / kPi;
    double amp = sqrt(pw[h]);
    num += amp * inst;
    den += amp * (h + 1);
  }
  return num / (den + kTiny);
}

__global__ void __launch_bounds__(256) sm_frame(StoneMaskParams p) {
  DYN_LDS(lds);
  const int f = blockIdx.x, u = blockIdx.y;
  if (f >= p.b.n_frames[u]) return;
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const double f0 = p.f0[fi], pos = p.tpos[fi];
  const int fs = p.b.fs, tid = threadIdx.x, nt = blockDim.x;
  if (!(f0 > kFloorF0StoneMask) || f0 > fs / 12.0) {                // stonemask.cpp:187-188 (a NaN F0 -- undefined there -- gives 0)
    if (tid == 0) p.refined[fi] = 0.0;
    return;
  }
  // LDS: the main window | scratch | the samples' indices.  Every index is rounded on its own (below) and so lands within
  // one of round(pos fs) - hw + i: a byte per sample holds that difference (as ints they were a third of the allocation
  // and capped the window -- 3 fs / 40 samples at the 40 Hz floor -- at fs = 180 kHz)
  double *mw = reinterpret_cast<double *>(lds);
  double *scratch = mw + p.win_cap;
  signed char *rawd = reinterpret_cast<signed char *>(scratch + 64);
  const double *x = p.b.x + (size_t)u * p.b.x_stride;
  const int x_len = p.b.x_len[u];
  const int hw = static_cast<int>(1.5 * fs / f0 + 1.0);
  const int blen = 2 * hw + 1;
  const double wlen_t = (2.0 * hw + 1.0) / fs;
  int lgN = 2;
  while ((2 << (lgN - 2)) <= blen) ++lgN;                           // 2^(2 + floor(log2(2hw+1)))
  // GetBaseIndex + GetMainWindow (stonemask.cpp:24-43): every sample index is rounded on its own
  const int raw0 = mround(pos * fs) - hw;
  for (int i = tid; i < blen; i += nt) {
    const double bt = static_cast<double>(-hw + i) / fs;
    const int r = mround((pos + bt) * fs);
    rawd[i] = static_cast<signed char>(r - (raw0 + i));
    const double t = (r - 1.0) / fs - pos;
    const double c1 = cospi(2.0 * t / wlen_t);
    mw[i] = 0.42 + 0.5 * c1 + 0.08 * (2.0 * c1 * c1 - 1.0);
  }
  __syncthreads();
  double pw[6], ni[6];
  // GetTentativeF0 (stonemask.cpp:123-132): 2 harmonics, then 6 around the tentative value
  sm_harmonic_bins(x, x_len, mw, rawd, raw0, blen, lgN, f0, fs, 2, p.tab.tw, scratch, pw, ni);
  double tent = sm_fix_f0(pw, ni, lgN, fs, f0, 2);
  double mean;
  if (tent <= 0.0 || tent > f0 * 2) {
    mean = 0.0;
  } else {
    sm_harmonic_bins(x, x_len, mw, rawd, raw0, blen, lgN, tent, fs, 6, p.tab.tw, scratch, pw, ni);
    mean = sm_fix_f0(pw, ni, lgN, fs, tent, 6);
  }
  if (fabs(mean - f0) > f0 * 0.2) mean = f0;                        // stonemask.cpp:203
  if (tid == 0) p.refined[fi] = mean;
}

size_t stonemask_lds_bytes(int win_cap) {
  return sizeof(double) * (size_t)win_cap + sizeof(double) * 64 + (((size_t)win_cap + 15) & ~(size_t)15);
}

void launch_stonemask(const StoneMaskParams &p, int max_frames, hipStream_t stream) {
  const size_t lds = stonemask_lds_bytes(p.win_cap);
  // Workgroup size follows the transform: a radix-8 stage of an N-point real transform has N/16 butterflies,
  // and threads beyond that idle through every stage.  Up to 24 kHz the transforms are 512-1024 points
  // (64 threads: 0.54 ms for 64 x 1001 frames at 16 kHz, 128: 0.69, 256: 1.32); at 48 kHz they are 2048-4096.
  WH_BLOCKS(sm_frame, dim3(max_frames, p.b.n_utt), p.b.fs <= 24000 ? 64 : 256, lds, stream, p);
}

}  // namespace world_hip
