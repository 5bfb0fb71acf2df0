// d4c.hip -- D4C band aperiodicity on gfx950.
//
// Replaces D4C() (reference src/d4c.cpp:342-403) and everything it calls:
// D4CLoveTrain (:227-285), GetStaticCentroid/GetCentroid (:90-143),
// GetSmoothedPowerSpectrum (:149-166), GetStaticGroupDelay (:172-188),
// GetCoarseAperiodicity (:194-225), GetAperiodicity (:330-338).
//
// The reference draws noise from ONE stream: first the LoveTrain window of every
// voiced frame in order, then three windows per frame that passed the LoveTrain
// threshold.  Frame independence is recovered by two prefix sums over the
// per-frame draw counts (d4c_prepare1 / d4c_prepare2) and GF(2) jump-ahead.
//
//   d4c_lovetrain : 256-thread workgroup per frame, one r2c FFT, two band sums.
//   d4c_groupdelay: 512-thread workgroup per selected frame, all in LDS:
//                   2 centroid transforms (each = the reference's two r2c FFTs
//                   packed into ONE complex FFT: z = w x + i (n+1) w x), 1 power
//                   spectrum, 2 DC corrections, 3 rectangular smoothings
//                   (block-parallel prefix sums) -> static group delay to HBM.
//   d4c_band      : workgroup per (3 kHz band, selected frame): windowed slice ->
//                   r2c FFT -> power -> radix select in registers replacing the
//                   reference's std::sort (only the sum of the N/2-boundary
//                   smallest powers is used) -> one coarse aperiodicity value.
//   d4c_finish    : the 3 kHz-grid interpolation, every row written once to HBM.
#include "stage_params.h"
#include "trace.h"
WH_TRACE_DEFINE(d4c)

namespace world_hip {

constexpr int kHanning = 1, kBlackman = 2;

// window value of sample i; scale = 2 / ratio / fs * f0, so that scale * (i - hw) is
// position * f0 of d4c.cpp:36,41 (the two divisions hoisted out of the per-sample loop)
__device__ __forceinline__ double d4c_window_at(int i, int hw, int kind, double scale) {
  const double c1 = cospi(scale * (i - hw));                        // cos(pi * position * f0)
  if (kind == kHanning) return 0.5 * c1 + 0.5;
  return 0.42 + 0.5 * c1 + 0.08 * (2.0 * c1 * c1 - 1.0);             // cos(2a) = 2 cos^2(a) - 1
}


// ---------------------------------------------------------------------------
__global__ void d4c_prepare1(D4cParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  int u = blockIdx.x, nf = p.b.n_frames[u];
  const double *f0 = p.f0 + (size_t)u * p.b.f_stride;
  unsigned *off_out = p.offsets1 + (size_t)u * p.b.f_stride;
  unsigned running = 0;
  for (int base = 0; base < nf; base += blockDim.x) {
    int f = base + threadIdx.x, cnt = 0;
    if (f < nf && f0[f] != 0.0) {
      double cf0 = f0[f] > 40.0 ? f0[f] : 40.0;                      // d4c.cpp:263,279
      cnt = 2 * mround(3.0 * p.b.fs / cf0 / 2.0) + 1;
    }
    int total, off = block_excl_scan_int(cnt, &total, scratch);
    if (f < nf) off_out[f] = running + (unsigned)off;
    running += (unsigned)total;
  }
  if (threadIdx.x == 0) p.draws1[u] = running;
}

__global__ void d4c_prepare2(D4cParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  int u = blockIdx.x, nf = p.b.n_frames[u];
  const double *f0 = p.f0 + (size_t)u * p.b.f_stride;
  const double *ap0 = p.ap0 + (size_t)u * p.b.f_stride;
  unsigned *off_out = p.offsets2 + (size_t)u * p.b.f_stride;
  unsigned running = p.draws1[u];        // pass 2 continues the stream where pass 1 ended
  for (int base = 0; base < nf; base += blockDim.x) {
    int f = base + threadIdx.x, cnt = 0;
    if (f < nf && !(f0[f] == 0 || ap0[f] <= p.threshold)) {          // d4c.cpp:386
      double cf0 = kFloorF0D4C > f0[f] ? kFloorF0D4C : f0[f];
      cnt = 3 * (2 * mround(4.0 * p.b.fs / cf0 / 2.0) + 1);
    }
    int total, off = block_excl_scan_int(cnt, &total, scratch);
    if (f < nf) off_out[f] = running + (unsigned)off;
    running += (unsigned)total;
  }
}

// Windowed, noise-dithered, DC-balanced segment (GetWindowedWaveform, d4c.cpp:52-84),
// written straight into a transform's input: sample i goes to the real part of
// complex element i (`packed`, the centroid transform) or to real element i of an
// r2c input.  The window shape is recomputed in the second pass (one cosine) rather
// than stored: no LDS, no long-lived registers.  Returns 2*hw+1.
__device__ __forceinline__ int d4c_windowed(const double *x, int x_len, int fs, double f0, double pos,
                                            int kind, double ratio, const uint32_t *noise,
                                            cplx *z, bool packed, double *scratch) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int hw = mround(ratio * fs / f0 / 2.0);
  const int wlen = 2 * hw + 1;
  const int origin = mround(pos * fs + 0.001);
  // Second pass needs each sample's window value again.  The packed layout has a free slot for
  // it (the imaginary half, overwritten by the caller afterwards); the r2c layout has none, so
  // the value is recomputed (one cospi, no division).
  const double scale = 2.0 / ratio / fs * f0;
  double s1 = 0.0, s2 = 0.0;
  for (int i = tid; i < wlen; i += nt) {
    const double w = d4c_window_at(i, hw, kind, scale);
    // noise[i]: the window's draws in sample order (d4c.cpp:67-69)
    double v = x[imin(x_len - 1, imax(0, origin + i - hw))] * w + randn_value(noise[i]) * kSafeGuardD4C;
    if (packed) { cplx e; e.re = v; e.im = w; z[swz(i)] = e; }
    else rfft_in(z, i) = v;
    s1 += v; s2 += w;
  }
  block_sum2(s1, s2, scratch);
  const double coef = s1 / s2;
  if (packed) {
    for (int i = tid; i < wlen; i += nt) { cplx &e = z[swz(i)]; e.re -= e.im * coef; }
  } else {
    for (int i = tid; i < wlen; i += nt) rfft_in(z, i) -= d4c_window_at(i, hw, kind, scale) * coef;
  }
  __syncthreads();
  return wlen;
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) d4c_lovetrain(D4cParams p) {
  DYN_LDS(lds);
  const int u = blockIdx.y, f = blockIdx.x;
  if (f >= p.b.n_frames[u]) return;
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const double f0 = p.f0[fi];
  if (f0 == 0.0) { if (threadIdx.x == 0) p.ap0[fi] = 0.0; return; }   // d4c.cpp:274-277
  const int lgn = p.lg_love, M = 1 << lgn, fs = p.b.fs;
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds);
  double *scratch = Zr + M;
  const TwLds tw = stage_twiddles(scratch + 64, lgn - 1, p.tab.tw);    // inner complex transform's table
  const double cf0 = f0 > 40.0 ? f0 : 40.0;
  const int wlen = d4c_windowed(p.b.x + (size_t)u * p.b.x_stride, p.b.x_len[u], fs, cf0, p.tpos[fi],
                                kBlackman, 3.0, p.noise + p.offsets1[fi], Z, false, scratch);
  for (int i = wlen + threadIdx.x; i < M; i += blockDim.x) rfft_in(Z, i) = 0.0;
  const int b0 = static_cast<int>(ceil(100.0 * M / fs));
  const int b1 = static_cast<int>(ceil(4000.0 * M / fs));
  const int b2 = static_cast<int>(ceil(7900.0 * M / fs));
  double lo = 0.0, hi = 0.0;     // cumulative power (b0, b1] and (b0, b2]   (d4c.cpp:241-249)
  block_rfft<3>(Z, lgn, tw, [&](int k, double re, double im) {
    if (k > b0 && k <= b2) {
      double pw = re * re + im * im;
      hi += pw;
      if (k <= b1) lo += pw;
    }
  });
  block_sum2(lo, hi, scratch);
  if (threadIdx.x == 0) p.ap0[fi] = lo / hi;
}

// ---------------------------------------------------------------------------
// LinearSmoothing (common.cpp:27-111) on an LDS spectrum of half+1 bins.
// seg: LDS work area of >= half + 2*bnd + 1 doubles.  in == out allowed.
__device__ __forceinline__ void d4c_smooth(const double *in, double width, int fs, int N, double *seg,
                                           double *out, double *scratch) {
  const int tid = threadIdx.x, nt = blockDim.x, half = N / 2;
  const int bnd = static_cast<int>(width * N / fs) + 1;
  const int seg_len = half + 2 * bnd + 1;
  const double inv_n = 1.0 / N;
  __syncthreads();
  for (int i = tid; i < seg_len; i += nt) {
    double m;
    if (i < bnd) m = in[bnd - i];
    else if (i < half + bnd) m = in[i - bnd];
    else m = in[half - (i - (half + bnd))];
    seg[i] = m * fs * inv_n;                           // == m * fs / N: N is a power of two
  }
  block_scan_incl_double(seg, seg_len, scratch);
  const double origin_axis = -(bnd - 0.5) * fs / N;
  const double inv_step = static_cast<double>(N) / fs, inv_width = 1.0 / width;
  for (int i = tid; i <= half; i += nt) {
    double fa = static_cast<double>(i) * inv_n * fs - width / 2.0;
    double lo = interp_uniform_rcp(origin_axis, inv_step, seg, seg_len, fa);
    fa += width;
    double hi = interp_uniform_rcp(origin_axis, inv_step, seg, seg_len, fa);
    out[i] = (hi - lo) * inv_width;
  }
  __syncthreads();
}

// DCCorrection (common.cpp:56-75) in place on an LDS spectrum; tmp >= 2 + f0*N/fs doubles.
__device__ __forceinline__ void d4c_dc_correct(double *spec, double f0, int fs, int N, double *tmp) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int upper = 2 + static_cast<int>(f0 * N / fs);
  const int nrep = upper - 1;
  const double dx = -static_cast<double>(fs) / N;
  __syncthreads();
  for (int i = tid; i < nrep; i += nt)
    tmp[i] = interp_uniform(f0, dx, spec, upper + 1, static_cast<double>(i) * fs / N);
  __syncthreads();
  for (int i = tid; i < nrep; i += nt) spec[i] = spec[i] + tmp[i];
  __syncthreads();
}

// Sum of the `m` smallest of v[0..n) (all >= 0) and the sum of all of them: a radix
// select on the IEEE bit patterns (monotone for non-negative doubles), 8 bits per pass,
// histogram in LDS.  Keys stay in registers.  Digits start at the first bit in which any
// two keys differ (found from the block min/max), so the first histogram spans exactly
// [kmin, kmax]; the walk stops as soon as the bucket holding the threshold contains a
// single key -- typically after two passes.  Barriers: two for min/max, ONE per pass (the
// first kSelHists histograms are zeroed up front), two for the final sums.
// keys one thread of a 256-thread workgroup receives from block_rfft of NMAX points
template <int NMAX> struct SelKeys {
#ifdef WORLD_EMU
  static constexpr int n = NMAX / 2 + 1;
#else
  static constexpr int n = 2 * ((NMAX / 4 + 1 + 255) / 256);
#endif
};
constexpr int kSelHists = 3;
// key[q], q < mine: bit patterns of this thread's elements (n elements block-wide).
// hist: kSelHists x 256 ints of LDS.
template <int kSelKeys>
__device__ __forceinline__ void block_smallest_sum(const unsigned long long (&key)[kSelKeys], int mine, int n, int m,
                                                   int *hist, double *scratch, double *partial, double *total,
                                                   bool trace_me = false) {
  (void)trace_me;
  const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id(), wv = wave_in_block(), nw = waves_per_block();
  unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
  for (int q = 0; q < kSelKeys; ++q)
    if (q < mine) { kmin = key[q] < kmin ? key[q] : kmin; kmax = key[q] > kmax ? key[q] : kmax; }
  for (int i = tid; i < kSelHists * 256; i += nt) hist[i] = 0;
#ifndef WORLD_EMU
  for (int s = 32; s >= 1; s >>= 1) {
    unsigned long long a = __shfl_xor(kmin, s, 64), b = __shfl_xor(kmax, s, 64);
    kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
  }
  unsigned long long *ks = reinterpret_cast<unsigned long long *>(scratch);
  __syncthreads();
  if (lane == 0) { ks[wv] = kmin; ks[32 + wv] = kmax; }
  __syncthreads();
  for (int w = 0; w < nw; ++w) { kmin = ks[w] < kmin ? ks[w] : kmin; kmax = ks[32 + w] > kmax ? ks[32 + w] : kmax; }
#else
  (void)lane; (void)wv; (void)nw;
#endif
  WH_STAMP(0, 3);
  const unsigned long long diff = kmin ^ kmax;
  int hi = diff ? 64 - __clzll((long long)diff) : 0;    // bits >= hi are common to every key
  unsigned long long prefix = hi >= 64 ? 0ull : (kmin >> hi) << hi;
  int remaining = m - 1;                           // rank (0-based, ascending) of the threshold element
  int bucket = n;                                   // keys that still match the prefix
  for (int it = 0; hi > 0 && bucket > 1; ++it) {
    const int shift = hi > 8 ? hi - 8 : 0;
    const unsigned long long mask = (1ull << (hi - shift)) - 1ull;
    int *h = hist + (it % kSelHists) * 256;
    if (it >= kSelHists) {                          // rare: recycle a histogram
      __syncthreads();
      for (int i = tid; i < 256; i += nt) h[i] = 0;
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < kSelKeys; ++q) {
      const bool in = q < mine && (it == 0 || (key[q] >> hi) == (prefix >> hi));
      if (in) atomicAdd(&h[(int)((key[q] >> shift) & mask)], 1);
    }
    __syncthreads();
    // every wave locates the digit redundantly: each lane sums its bins, a wave scan finds the rank
    const int per_lane = 256 / WAVE;
    int local = 0;
    for (int j = 0; j < per_lane; ++j) local += h[lane_id() * per_lane + j];
    int tot, before = wave_excl_scan_int(local, &tot);
    int digit = -1, below = -1, hsel = -1;
    if (before <= remaining && remaining < before + local) {
      int acc = before;
      for (int j = 0; j < per_lane; ++j) {
        int c = h[lane_id() * per_lane + j];
        if (remaining < acc + c) { digit = lane_id() * per_lane + j; below = acc; hsel = c; break; }
        acc += c;
      }
    }
    // exactly one lane found the bin: two max-reductions broadcast (digit, below) and hsel
    {
      const int got = wave_max_int(digit < 0 ? -1 : (digit << 12) | below);     // below <= 2049 < 2^12
      hsel = wave_max_int(hsel);
      digit = got >> 12; below = got & 4095;
    }
    remaining -= below;
    bucket = hsel;
    prefix |= (unsigned long long)digit << shift;
    hi = shift;
    WH_STAMP(0, 4 + (it < 3 ? it : 3));
  }
  // Bits >= hi of the threshold are known.  If hi > 0 the bucket holds exactly one key (the
  // threshold itself), every other key differs from it above bit hi, and its owner
  // contributes it through the third sum; if hi == 0 the prefix IS the threshold (possibly
  // shared by several equal keys, `remaining` of which lie below the rank).
  const unsigned long long pfx = hi >= 64 ? 0ull : prefix >> hi;
  double s_lt = 0.0, s_all = 0.0, s_thr = 0.0;
#pragma unroll
  for (int q = 0; q < kSelKeys; ++q) {
    if (q < mine) {
      const double x = __longlong_as_double((long long)key[q]);
      const unsigned long long top = hi >= 64 ? 0ull : key[q] >> hi;
      s_all += x;
      if (top < pfx) s_lt += x;
      else if (top == pfx && hi > 0) s_thr += x;
    }
  }
  (void)n;
  WH_STAMP(0, 8);
  block_sum3(s_lt, s_all, s_thr, scratch);
  const double thr = hi > 0 ? s_thr : __longlong_as_double((long long)prefix);
  *partial = s_lt + (remaining + 1) * thr;
  *total = s_all;
}

// ---------------------------------------------------------------------------
// Stage A of D4CGeneralBody: static group delay of one selected frame -> HBM.
// (GetStaticCentroid, GetSmoothedPowerSpectrum, GetStaticGroupDelay: d4c.cpp:126-188)
// Workgroup shape of d4c_groupdelay: 256 threads with radix-16 butterflies, or 512 threads with
// radix-8 butterflies (every thread busy in every FFT stage, half the registers per thread).
constexpr int kGdThreads = 512;
constexpr bool kGdRadix8 = true;
template <int NMAX>
__global__ void __launch_bounds__(kGdThreads, kGdThreads == 512 ? 4 : 1) d4c_groupdelay(D4cParams p) {
  DYN_LDS(lds);
  const int u = blockIdx.y, f = blockIdx.x;
  if (f >= p.b.n_frames[u]) return;
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double f0 = p.f0[fi];
  if (f0 == 0 || p.ap0[fi] <= p.threshold) return;                     // d4c.cpp:386
  const bool trace_me = f == 1000; (void)trace_me;
  WH_STAMP(32, 0);
  const int lgn = p.lg_d4c, N = 1 << lgn, H = N / 2, fs = p.b.fs;
  // LDS: Z (N complex + 8) | scratch (64) | twiddles.  The packed centroid transform
  // needs all of Z; afterwards Z is re-carved into the real-FFT / prefix-sum work area
  // [0, N), B = [N, N+H+1) and A = [N+H+1, N+2H+2).  During the centroid phase A lives
  // in registers (each thread always owns the same bins).
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds);
  double *B = Zr + N;
  double *A = B + (H + 1);
  double *scratch = Zr + 2 * N + 8;
  const TwLds tw = stage_twiddles(scratch + 64, lgn, p.tab.tw);
#ifdef WORLD_EMU
  constexpr int kBinsPerThread = NMAX / 2 + 1;          // one emulated thread owns every bin
#else
  constexpr int kBinsPerThread = (NMAX / 2 + 1 + kGdThreads - 1) / kGdThreads;
#endif
  double a_reg[kBinsPerThread];
  const FftPlan plan_c = kGdRadix8 ? make_plan_r8(lgn) : make_plan(lgn);            // the packed centroid transform: 2^lgn complex points
  const int top_bit = plan_c.rl(plan_c.ns - 1) - 1; // bit of a slot index that carries the top bit of its bin

  const double *x = p.b.x + (size_t)u * p.b.x_stride;
  const int x_len = p.b.x_len[u];
  const double pos = p.tpos[fi];
  const double cf0 = kFloorF0D4C > f0 ? kFloorF0D4C : f0;
  const uint32_t *noise = p.noise + p.offsets2[fi];
  const int wdraws = 2 * mround(4.0 * fs / cf0 / 2.0) + 1;

  // ---- GetStaticCentroid (d4c.cpp:126-143) ----------------------------------
  for (int c = 0; c < 2; ++c) {
    const double cpos = c == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0;
    __syncthreads();
    WH_STAMP(32, 1 + 5 * c);
    const int wlen = d4c_windowed(x, x_len, fs, cf0, cpos, kBlackman, 4.0, noise + (size_t)c * wdraws,
                                  Z, true, scratch);
    WH_STAMP(32, 2 + 5 * c);
    double pw = 0.0;
    for (int i = tid; i < wlen; i += nt) { const double v = Z[swz(i)].re; pw += v * v; }
    pw = block_sum(pw, scratch);
    const double inv_nrm = 1.0 / sqrt(pw);
    for (int i = tid; i < N; i += nt) {
      cplx &e = Z[swz(i)];
      double v = i < wlen ? e.re * inv_nrm : 0.0;
      e.re = v;
      e.im = v * (i + 1.0);                          // second transform's input (d4c.cpp:111-112)
    }
    WH_STAMP(32, 3 + 5 * c);
    block_cfft_dif<kGdRadix8 ? 3 : 4>(Z, plan_c, tw);
    WH_STAMP(32, 4 + 5 * c);
    // Bins k <= H in an order that makes consecutive lanes read consecutive physical
    // slots (conflict-free): bins below H are exactly the slots whose last-stage digit
    // has its top bit clear.  Item `it` names the same bin in both centroid passes.
#pragma unroll
    for (int slot = 0; slot < kBinsPerThread; ++slot) {
      const int it = tid + slot * nt;
      if (it > H) break;
      int k, phys;
      if (it < H) {
        const int pos = ((it >> top_bit) << (top_bit + 1)) | (it & ((1 << top_bit) - 1));
        phys = swz(pos);
        k = fft_bin_of_slot(plan_c, phys);
      } else {
        k = H;
        phys = fft_slot(plan_c, H);
      }
      cplx za = Z[phys], zb = Z[fft_slot(plan_c, (N - k) & (N - 1))];
      double x1r = 0.5 * (za.re + zb.re), x1i = 0.5 * (za.im - zb.im);
      double x2r = 0.5 * (za.im + zb.im), x2i = -0.5 * (za.re - zb.re);
      if (k == 0 || k == H) { x1i = 0.0; x2i = 0.0; }
      double cen = x2r * x1r + x1i * x2i;            // d4c.cpp:115-116
      a_reg[slot] = c == 0 ? cen : a_reg[slot] + cen;
    }
    WH_STAMP(32, 5 + 5 * c);
  }
  __syncthreads();
#pragma unroll
  for (int slot = 0; slot < kBinsPerThread; ++slot) {
    const int it = tid + slot * nt;
    if (it > H) break;
    int k = H;
    if (it < H) {
      const int pos = ((it >> top_bit) << (top_bit + 1)) | (it & ((1 << top_bit) - 1));
      k = fft_bin_of_slot(plan_c, swz(pos));
    }
    A[k] = a_reg[slot];
  }
  WH_STAMP(32, 11);
  d4c_dc_correct(A, cf0, fs, N, Zr);
  WH_STAMP(32, 12);

  // ---- GetSmoothedPowerSpectrum (d4c.cpp:149-166) ----------------------------
  {
    const int wlen = d4c_windowed(x, x_len, fs, cf0, pos, kHanning, 4.0, noise + (size_t)2 * wdraws,
                                  Z, false, scratch);
    for (int i = wlen + tid; i < N; i += nt) rfft_in(Z, i) = 0.0;
    WH_STAMP(32, 13);
    block_rfft<kGdRadix8 ? 3 : 4>(Z, lgn, tw, [&](int k, double re, double im) { B[k] = re * re + im * im; });
    WH_STAMP(32, 14);
  }
  d4c_dc_correct(B, cf0, fs, N, Zr);
  WH_STAMP(32, 15);
  d4c_smooth(B, cf0, fs, N, Zr, B, scratch);
  WH_STAMP(32, 16);

  // ---- GetStaticGroupDelay (d4c.cpp:172-188) ----------------------------------
  for (int i = tid; i <= H; i += nt) A[i] = A[i] / B[i];
  WH_STAMP(32, 17);
  d4c_smooth(A, cf0 / 2.0, fs, N, Zr, A, scratch);
  WH_STAMP(32, 18);
  d4c_smooth(A, cf0, fs, N, Zr, B, scratch);
  WH_STAMP(32, 19);
  double *gd = p.gd + fi * p.gd_stride;
  for (int i = tid; i <= H; i += nt) gd[i] = A[i] - B[i];
  WH_STAMP(32, 20);
}

// ---------------------------------------------------------------------------
// Stage B: one workgroup per (band, selected frame).  GetCoarseAperiodicity
// (d4c.cpp:194-225): Nuttall-windowed slice of the group delay -> r2c -> power ->
// share of the N/2-boundary smallest bins.  The power values never touch LDS: the
// transform's merge step hands bin tid + q*T to thread tid, which is exactly the key
// layout of the radix select.
template <int NMAX>
__global__ void __launch_bounds__(256) d4c_band(D4cParams p) {
  constexpr int kSelKeys = SelKeys<NMAX>::n;
  DYN_LDS(lds);
  const int band = blockIdx.x, f = blockIdx.y, u = blockIdx.z;
  if (f >= p.b.n_frames[u]) return;
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const double f0 = p.f0[fi];
  if (f0 == 0 || p.ap0[fi] <= p.threshold) return;
  const int tid = threadIdx.x;
  const bool trace_me = band == 2 && f == 1000; (void)trace_me;
  WH_STAMP(0, 0);
  const int lgn = p.lg_d4c, N = 1 << lgn, H = N / 2, fs = p.b.fs;
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds);
  int *hist = reinterpret_cast<int *>(Zr + N);
  double *scratch = reinterpret_cast<double *>(hist + kSelHists * 256);
  // table for the inner N/2-point complex transform only (the merge step derives its odd twiddles)
  const TwLds tw = stage_twiddles(scratch + 64, lgn - 1, p.tab.tw);
  WH_STAMP(0, 1);
  const double cf0 = kFloorF0D4C > f0 ? kFloorF0D4C : f0;
  const int bnd = mround(N * 8.0 / p.wl);
  const int hwl = p.wl / 2;
  const int center = static_cast<int>(3000.0 * (band + 1) * N / fs);
  const double *gd = p.gd + fi * p.gd_stride + (center - hwl);
  const double *nut = p.nuttall;
  const int wl = 2 * hwl + 1;
  unsigned long long key[kSelKeys];
#pragma unroll
  for (int q = 0; q < kSelKeys; ++q) key[q] = ~0ull;
  int filled = 0;
  // the windowed slice is short (wl of N samples): the first FFT stage reads it straight from HBM
  block_rfft_from(Z, lgn, tw,
    [&](int n) {
      cplx v; v.re = 0.0; v.im = 0.0;
      const int i = 2 * n;
      if (i < wl) v.re = gd[i] * nut[i];
      if (i + 1 < wl) v.im = gd[i + 1] * nut[i + 1];
      return v;
    },
    [&](int k, double re, double im) {
      (void)k;
      key[filled < kSelKeys ? filled : kSelKeys - 1] = (unsigned long long)__double_as_longlong(re * re + im * im);
      ++filled;
    });
  WH_STAMP(0, 2);
  double part, tot;
  block_smallest_sum(key, filled, H + 1, H - bnd, hist, scratch, &part, &tot, trace_me);
  WH_STAMP(0, 9);
  if (tid == 0) {
    double c = 10 * log10(part / tot);
    c = c + (cf0 - 100) / 50.0;                       // d4c.cpp:314-316
    p.coarse[fi * 16 + 1 + band] = c < 0.0 ? c : 0.0;
  }
}

// ---------------------------------------------------------------------------
// Stage C: GetAperiodicity (d4c.cpp:323-338) -- every frame's output row.
__global__ void d4c_finish(D4cParams p) {
  const int u = blockIdx.y, f = blockIdx.x;
  if (f >= p.b.n_frames[u]) return;
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const int tid = threadIdx.x, nt = blockDim.x, fs = p.b.fs;
  const int nb_out = p.fft_out / 2 + 1;
  double *row = p.aperiodicity + fi * nb_out;
  const double f0 = p.f0[fi];
  if (f0 == 0 || p.ap0[fi] <= p.threshold) {                          // d4c.cpp:323-328,386
    for (int i = tid; i < nb_out; i += nt) row[i] = 1.0 - kTiny;
    return;
  }
  const double *coarse_in = p.coarse + fi * 16;
  const int nk = p.nap + 2;
  for (int i = tid; i < nb_out; i += nt) {
    double xi = static_cast<double>(i) * fs / p.fft_out;
    int cnt = 0;                                       // knots <= xi  (histc semantics)
    for (int k = 0; k < nk; ++k) {
      double knot = k <= p.nap ? k * 3000.0 : fs / 2.0;
      if (knot <= xi) cnt++;
    }
    int k = cnt < 1 ? 1 : (cnt > nk - 1 ? nk - 1 : cnt);
    auto cval = [&](int j) { return j == 0 ? -60.0 : (j == p.nap + 1 ? -kTiny : coarse_in[j]); };   // d4c.cpp:373-375
    double x0 = (k - 1) <= p.nap ? (k - 1) * 3000.0 : fs / 2.0;
    double x1 = k <= p.nap ? k * 3000.0 : fs / 2.0;
    double s = (xi - x0) / (x1 - x0);
    double y = cval(k - 1) + s * (cval(k) - cval(k - 1));
    row[i] = pow(10.0, y / 20.0);
  }
}

// ---------------------------------------------------------------------------
size_t d4c_love_lds_bytes(int lg) { return sizeof(double) * (size_t)((1 << lg) + 64 + (1 << lg) / 8 + 2); }
size_t d4c_groupdelay_lds_bytes(int lg) {
  int N = 1 << lg;
  return sizeof(double) * (size_t)(2 * N + 8 + 64 + N / 4 + 2);
}
size_t d4c_band_lds_bytes(int lg) {
  int N = 1 << lg;
  return sizeof(double) * (size_t)(N + kSelHists * 128 + 64 + N / 8 + 2);
}

// worst case per frame: LoveTrain window at 40 Hz + 3 body windows at 47 Hz
size_t d4c_max_draws_per_frame(int fs) {
  return (size_t)(2 * mround(3.0 * fs / 40.0 / 2.0) + 1) + 3 * (size_t)(2 * mround(4.0 * fs / kFloorF0D4C / 2.0) + 1);
}

void launch_d4c(const D4cParams &p, int max_frames, hipStream_t stream) {
  WH_BLOCKS(d4c_prepare1, dim3(p.b.n_utt), 256, 64 * sizeof(double), stream, p);
  // workgroup sizes follow the transform size (threads beyond N/16 idle through every radix-8 stage): for the
  // 2048-point internal FFT of fs <= 24 kHz, 64 x 1001 frames: lovetrain 0.53 -> 0.42 ms, groupdelay 3.77 -> 2.52,
  // band 0.97 -> 0.65
  WH_BLOCKS(d4c_lovetrain, dim3(max_frames, p.b.n_utt), p.lg_love <= 11 ? 128 : 256, d4c_love_lds_bytes(p.lg_love), stream, p);
  WH_BLOCKS(d4c_prepare2, dim3(p.b.n_utt), 256, 64 * sizeof(double), stream, p);
  // per-thread register arrays are sized for the internal FFT: 4096 points up to 48 kHz, 8192 up to 96 kHz
  if (p.lg_d4c <= 12) {
    devrt::launch_blocks("d4c_groupdelay", d4c_groupdelay<4096>, dim3(max_frames, p.b.n_utt), p.lg_d4c <= 11 ? 256 : kGdThreads,
                         d4c_groupdelay_lds_bytes(p.lg_d4c), stream, p);
    devrt::launch_blocks("d4c_band", d4c_band<4096>, dim3(p.nap, max_frames, p.b.n_utt), p.lg_d4c <= 11 ? 128 : 256,
                         d4c_band_lds_bytes(p.lg_d4c), stream, p);
  } else {
    devrt::launch_blocks("d4c_groupdelay", d4c_groupdelay<8192>, dim3(max_frames, p.b.n_utt), kGdThreads,
                         d4c_groupdelay_lds_bytes(p.lg_d4c), stream, p);
    devrt::launch_blocks("d4c_band", d4c_band<8192>, dim3(p.nap, max_frames, p.b.n_utt), 256,
                         d4c_band_lds_bytes(p.lg_d4c), stream, p);
  }
  WH_BLOCKS(d4c_finish, dim3(max_frames, p.b.n_utt), 256, 0, stream, p);
}

}  // namespace world_hip
