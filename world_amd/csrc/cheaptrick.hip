// cheaptrick.hip -- CheapTrick spectral envelope on gfx950.
//
// Replaces CheapTrick() / CheapTrickGeneralBody() (reference src/cheaptrick.cpp:22-229)
// and the helpers it pulls from src/common.cpp (DCCorrection :56-75,
// LinearSmoothing :27-111).  The reference walks frames serially with one RNG
// stream; here every frame is an independent workgroup:
//
//   ct_prepare : per utterance, prefix-sum of randn() calls consumed per frame
//                -> each frame's xorshift128 state by GF(2) jump-ahead.
//   ct_spectrum: one workgroup per (frame, utterance): F0-adaptive window + noise ->
//                r2c FFT -> |X|^2 -> DC correction -> the mirrored segment whose
//                prefix sum LinearSmoothing needs, to the frame's scratch row.
//   ct_scan    : that prefix sum, SERIAL and in FP64 order (its rounding is visible
//                in low-energy bins) -- one LANE per frame, 64 chains
//                per wavefront.
//   ct_envelope: one workgroup per frame: rectangular smoothing -> +|randn|*eps ->
//                log -> r2c FFT (cepstrum) -> lifter -> c2r FFT -> exp -> HBM.
//
// HBM traffic per frame: the window's samples of x (L2-resident: adjacent
// frames overlap ~97%), the smoothing segment's round trip through its scratch
// row (~1200 doubles written, scanned in place, read: Infinity-Cache resident)
// and one row of the spectrogram written once.
#include "stage_params.h"
#include "trace.h"
WH_TRACE_DEFINE(ct)

namespace world_hip {

__device__ __forceinline__ double ct_effective_f0(double f0, double floor_f0) {
  return f0 <= floor_f0 ? kDefaultF0 : f0;              // cheaptrick.cpp:218
}

// doubles reserved for Z and the smoothing work area that overlays it (even)
__host__ __device__ __forceinline__ int ct_seg_cap(int N) {
  const int need = N / 2 + 1 + 2 * (N / 3 + 2) + 1;
  const int cap = need > N ? need : N;
  return cap + (cap & 1);
}

// ---------------------------------------------------------------------------
__global__ void ct_prepare(CtParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  int u = blockIdx.x;
  int nf = p.b.n_frames[u];
  int nb = (1 << p.lg_fft) / 2 + 1;
  const double *f0 = p.f0 + (size_t)u * p.b.f_stride;
  unsigned *off_out = p.offsets + (size_t)u * p.b.f_stride;
  unsigned running = 0;
  for (int base = 0; base < nf; base += blockDim.x) {
    int f = base + threadIdx.x;
    int cnt = 0;
    if (f < nf) {
      double cf0 = ct_effective_f0(f0[f], p.f0_floor);
      cnt = 2 * mround(1.5 * p.b.fs / cf0) + 1 + nb;     // window draws, then one per bin
    }
    int total, off = block_excl_scan_int(cnt, &total, scratch);
    if (f < nf) off_out[f] = running + (unsigned)off;
    running += (unsigned)total;
  }
}

// ---------------------------------------------------------------------------
// Largest butterfly of the frame kernels' transforms (log2): a 1024-point complex transform has only 64
// radix-16 butterflies for 256 threads; smaller butterflies keep more threads busy per stage.
constexpr int kCtMaxLr = 3;

// LinearSmoothing's geometry for one frame (common.cpp:27-50): boundary bins and segment length
struct CtSmooth { double width; int bnd, seg_len; };
__device__ __forceinline__ CtSmooth ct_smooth_shape(double cf0, int N, int fs) {
  CtSmooth s;
  s.width = cf0 * 2.0 / 3.0;
  s.bnd = static_cast<int>(s.width * N / fs) + 1;
  s.seg_len = N / 2 + 2 * s.bnd + 1;
  return s;
}

// Layout of the smoothing segments in HBM: frames in groups of WAVE, PAIRS of consecutive elements of the
// group's frames side by side -- [utterance][frame / WAVE][i / 2][frame % WAVE][i % 2] -- so that ct_scan, one
// lane per frame, moves 1 KB of contiguous memory per instruction (16 bytes per lane).  Row-per-frame, a lane's
// access is its own cache line, 64 lines per instruction: measured 74 us for the scan.  The frame kernels pay
// with 8-byte accesses 1 KB apart; they have the waves to hide it.
__device__ __forceinline__ double *ct_seg_at(const CtParams &p, int u, int f) {
  const size_t groups = (size_t)(p.b.f_stride + WAVE - 1) / WAVE;
  return p.seg + (((size_t)u * groups + f / WAVE) * p.seg_stride) * WAVE + (size_t)(f % WAVE) * 2;
}
__device__ __forceinline__ size_t ct_seg_elem(int i) { return (size_t)(i >> 1) * (2 * WAVE) + (i & 1); }   // offset of element i

// ---- stage 1: window -> r2c -> |X|^2 -> DC correction -> mirrored segment of LinearSmoothing, to HBM ----------
// PER: samples of the window a thread owns (fft_size / threads); LGN: log2(fft_size) when it is a compile-time
// constant of the instantiation (static FFT stages), 0 = taken from p.lg_fft.
// TB: the largest workgroup the instantiation is launched with (512 for the 8192-point transform: one frame then owns
// 107 KB of LDS, one workgroup per CU -- the shape exists so that f0 floors below 35 Hz at 48 kHz and sampling rates
// above 96 kHz run at all, not to be fast)
template <int PER, int LGN, int TB = 256>
__global__ void __launch_bounds__(TB, TB <= 256 ? 4 : 1) ct_spectrum(CtParams p) {
  DYN_LDS(lds);
  const int lgn = LGN > 0 ? LGN : p.lg_fft, N = 1 << lgn, half = N / 2, nb = half + 1;
  const int fs = p.b.fs;
  const int u = blockIdx.y, f = p.frame_lo + xcd_grouped(blockIdx.x, gridDim.x);
  if (f >= p.b.n_frames[u] || f >= p.frame_hi) return;
  const bool trace_me = f == 1000; (void)trace_me;
  WH_STAMP(0, 0);
  // LDS (doubles): Z: N | P: nb+1 | scratch: 64 | quarter-wave table of the inner N/2-point transform
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds);
  double *P = Zr + N;
  double *scratch = P + (nb + 1) + ((nb + 1) & 1);
  const TwLds tw = stage_twiddles(scratch + 64, lgn - 1, p.tab.tw);

  const size_t fi = (size_t)u * p.b.f_stride + f;
  const double *x = p.b.x + (size_t)u * p.b.x_stride;
  const int x_len = p.b.x_len[u];
  const double pos = p.tpos[fi];
  const double cf0 = ct_effective_f0(p.f0[fi], p.f0_floor);
  const uint32_t *noise = p.noise + p.offsets[fi];
  const int tid = threadIdx.x, nt = blockDim.x;

  WH_STAMP(0, 1);
  // ---- GetWindowedWaveform (cheaptrick.cpp:87-142) -------------------------
  const int hw = mround(1.5 * fs / cf0);
  const int wlen = 2 * hw + 1;
  const int origin = mround(pos * fs + 0.001);
  // One pass, one block reduction: with w the raw window, a = x w and n the dither, the
  // reference's normalised window is c w (c = 1/sqrt(sum w^2)), its waveform v = c a + n, and
  // the DC-balanced result v - c w (sum v / sum c w) -- all linear in four sums.  A thread keeps
  // the three values of each of its samples in registers; the frame's draws (sample order) come
  // from the noise stream.  cos(pi position f0) advances by a fixed angle from one of the thread's
  // samples to the next (they are nt apart): one sincospi pair per thread, then rotations.
  const double win_scale = 1.0 / 1.5 / fs * cf0;        // position * f0 = (i - hw) / 1.5 / fs * f0
  double wv[PER], av[PER], nv[PER];
  double s_ww = 0.0, s_a = 0.0, s_n = 0.0, s_w = 0.0;
  {
    double rc, rs, dc, ds;
    sincospi(win_scale * (tid - hw), &rs, &rc);
    sincospi(win_scale * nt, &ds, &dc);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = tid + q * nt;
      wv[q] = 0.0; av[q] = 0.0; nv[q] = 0.0;
      if (i < wlen) {
        const double w = 0.5 * rc + 0.5;                   // cos(pi * position * f0), cheaptrick.cpp:101-102
        const double a = x[imin(x_len - 1, imax(0, origin + i - hw))] * w, n = randn_value(noise[i]) * kTiny;
        wv[q] = w; av[q] = a; nv[q] = n;
        s_ww += w * w; s_a += a; s_n += n; s_w += w;
      }
      const double t = rc * dc - rs * ds;
      rs = rs * dc + rc * ds;
      rc = t;
    }
  }
  block_sum4(s_ww, s_a, s_n, s_w, scratch);
  const double c = 1.0 / sqrt(s_ww);
  const double cc = c * ((c * s_a + s_n) / (c * s_w));
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * nt;
    if (i < N) rfft_in(Z, i) = i < wlen ? av[q] * c + nv[q] - wv[q] * cc : 0.0;
  }

  WH_STAMP(0, 2);
  // ---- GetPowerSpectrum (cheaptrick.cpp:64-82): r2c, |X|^2 -----------------
  block_rfft<kCtMaxLr, LGN>(Z, lgn, tw, [&](int k, double re, double im) { P[k] = re * re + im * im; });

  WH_STAMP(0, 3);
  // DCCorrection (common.cpp:56-75); replica staged in Z (dead now), then added
  {
    const int upper = 2 + static_cast<int>(cf0 * N / fs);
    const int nrep = upper - 1;
    const double dx = -static_cast<double>(fs) / N;
    for (int i = tid; i < nrep; i += nt) {
      double axis = static_cast<double>(i) * fs / N;
      Zr[i] = interp_uniform(cf0, dx, P, upper + 1, axis);
    }
    __syncthreads();
    for (int i = tid; i < nrep; i += nt) P[i] = P[i] + Zr[i];
    __syncthreads();
  }

  WH_STAMP(0, 4);
  // ---- LinearSmoothing (common.cpp:27-111), width = 2/3 f0: the mirrored, scaled segment whose prefix sum
  // the reference walks serially goes to this frame's row in HBM (ct_scan)
  const CtSmooth sm = ct_smooth_shape(cf0, N, fs);
  const double inv_n = 1.0 / N;
  double *row = ct_seg_at(p, u, f);
  block_map<4, double>(sm.seg_len,
    [&](int i) {
      double m;
      if (i < sm.bnd) m = P[sm.bnd - i];
      else if (i < half + sm.bnd) m = P[i - sm.bnd];
      else m = P[half - (i - (half + sm.bnd))];
      return m * fs * inv_n;                           // == m * fs / N: N is a power of two
    },
    [&](int i, double v) { row[ct_seg_elem(i)] = v; });
  WH_STAMP(0, 5);
}

// ---- stage 2: the order-sensitive prefix sums, one LANE per frame ---------------------------------------------
// The cumulative sum of LinearSmoothing (common.cpp:85-86) must round exactly as the reference's left-to-right
// loop does: `hi - lo` below cancels up to 12 digits where the envelope sits at the noise floor, so any other
// summation order moves those bins by 1e-4 (tests/test_gpu_parity.py::test_hard_inputs_vs_oracle).
// A chain of dependent FP64 adds costs the same whatever the other lanes do, so the chains of 64 frames run
// side by side in one wavefront: 1200 steps for 64 frames instead of 1200 steps per frame with 255 threads of a
// workgroup waiting at a barrier (round 1: 40k of a frame's 135k cycles).  Rows are read in batches of 8 values
// per lane, the next batch in flight while the current one is added up.
constexpr int kScanBatch = 16;     // values per lane and batch (8 loads of 16 bytes)
constexpr int kScanDepth = 5;      // batches in flight per lane: enough rows requested ahead to cover a trip to HBM
                                   // (40 loads + a batch of stores stay below the 63 the vmcnt counter can tell apart)
// A frame's row holds seg_stride values although its segment is shorter: every lane walks to the longest
// segment of its wavefront, rounded up to whole rings (what lies beyond a frame's own length is scratch that
// nobody reads), so the loop has no per-element bounds -- no divergent branch, no waiting for all loads at
// every step.
__global__ void __launch_bounds__(WAVE) ct_scan(CtParams p) {
  const int u = blockIdx.y, f = p.frame_lo + (int)(blockIdx.x * WAVE) + lane_id();
  const int N = 1 << p.lg_fft;
  const bool active = f < p.b.n_frames[u] && f < p.frame_hi;
  const size_t fi = (size_t)u * p.b.f_stride + (active ? f : 0);
  const int len = active ? ct_smooth_shape(ct_effective_f0(p.f0[fi], p.f0_floor), N, p.b.fs).seg_len : 0;
  const int max_len = wave_max_int(len);
  if (!active) return;
  double2 *row = reinterpret_cast<double2 *>(ct_seg_at(p, u, f));      // pair j of this frame: row[j * WAVE]
  constexpr int kPairs = kScanBatch / 2;
  double2 buf[kScanDepth][kPairs];
#pragma unroll
  for (int d = 0; d < kScanDepth; ++d)
#pragma unroll
    for (int q = 0; q < kPairs; ++q) buf[d][q] = row[(size_t)(d * kPairs + q) * WAVE];
  double acc = 0.0;                   // 0 + seg[0] == seg[0]: the first element passes through unchanged, as in the reference
  for (int i0 = 0; i0 < max_len; i0 += kScanDepth * kScanBatch) {
#pragma unroll
    for (int d = 0; d < kScanDepth; ++d) {
      double2 *at = row + (size_t)(i0 / 2 + d * kPairs) * WAVE;
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {                               // strictly left to right
        acc = buf[d][q].x + acc; buf[d][q].x = acc;
        acc = buf[d][q].y + acc; buf[d][q].y = acc;
      }
#pragma unroll
      for (int q = 0; q < kPairs; ++q) at[(size_t)q * WAVE] = buf[d][q];
#pragma unroll
      for (int q = 0; q < kPairs; ++q) buf[d][q] = at[(size_t)(kScanDepth * kPairs + q) * WAVE];
    }
  }
}

// ---- stage 3: rectangular smoothing from the prefix sums -> log -> cepstrum -> lifter -> exp, to HBM -----------
template <int LGN, int TB = 256>
__global__ void __launch_bounds__(TB, TB <= 256 ? 4 : 1) ct_envelope(CtParams p) {
  DYN_LDS(lds);
  const int lgn = LGN > 0 ? LGN : p.lg_fft, N = 1 << lgn, half = N / 2, nb = half + 1;
  const int fs = p.b.fs;
  const int u = blockIdx.y, f = p.frame_lo + xcd_grouped(blockIdx.x, gridDim.x);
  if (f >= p.b.n_frames[u] || f >= p.frame_hi) return;
  const bool trace_me = f == 1000; (void)trace_me;
  WH_STAMP(0, 6);
  // LDS (doubles): seg / Z overlaid (the segment is dead once the transforms start): ct_seg_cap | P: nb+1 |
  // scratch: 64 | quarter-wave table of the inner N/2-point transform
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *seg = reinterpret_cast<double *>(lds);
  double *P = seg + ct_seg_cap(N);
  double *scratch = P + (nb + 1) + ((nb + 1) & 1);
  const TwLds tw = stage_twiddles(scratch + 64, lgn - 1, p.tab.tw);
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const double cf0 = ct_effective_f0(p.f0[fi], p.f0_floor);
  const int wlen = 2 * mround(1.5 * fs / cf0) + 1;
  const uint32_t *noise = p.noise + p.offsets[fi];
  const CtSmooth sm = ct_smooth_shape(cf0, N, fs);
  const double inv_n = 1.0 / N;
  const double *row = ct_seg_at(p, u, f);
  block_map<8, double>(sm.seg_len, [&](int i) { return row[ct_seg_elem(i)]; }, [&](int i, double v) { seg[i] = v; });
  __syncthreads();
  {
    const double origin_axis = -(sm.bnd - 0.5) * fs / N;
    const double step = static_cast<double>(fs) / N;
    // hi - lo cancels up to 12 digits where the envelope sits at the noise floor:
    // the interpolation weights and the final quotient keep the reference's exact operations
    // (true divisions), so that every product rounds as it does there.
    block_map<4, double>(half + 1,
      [&](int i) {
        double fa = static_cast<double>(i) * inv_n * fs - sm.width / 2.0;
        double lo = interp_uniform(origin_axis, step, seg, sm.seg_len, fa);
        fa += sm.width;
        double hi = interp_uniform(origin_axis, step, seg, sm.seg_len, fa);
        double smoothed = (hi - lo) / sm.width;
        // AddInfinitesimalNoise: the per-bin draws continue the frame's stream after the
        // window draws (cheaptrick.cpp:147-151); then the log of SmoothingWithRecovery (:39-42)
        return log(smoothed + fabs(randn_value(noise[wlen + i])) * kEps);
      },
      [&](int i, double lg) { P[i] = lg; });
  }

  WH_STAMP(0, 7);
  // ---- SmoothingWithRecovery (cheaptrick.cpp:22-57) -------------------------
  // the symmetric extension of the log spectrum is read by the first FFT stage directly from P.
  // Lifters: sin(pi f0 q) / (pi f0 q) and (1 - 2 q1) + 2 q1 cos(2 pi f0 q) at quefrency q = k / fs; with
  // a = f0 k / fs both come from ONE sinpi: cos(2 pi a) = 1 - 2 sin^2(pi a).
  const double q1 = p.q1, f0_over_fs = cf0 / fs;
  auto mirrored = [&](int i) { return i <= half ? P[i] : P[N - i]; };
  block_rfft_from<kCtMaxLr, LGN>(Z, lgn, tw, [&](int n) { cplx v; v.re = mirrored(2 * n); v.im = mirrored(2 * n + 1); return v; },
                  [&](int k, double re, double im) {
    (void)im;
    double sl, cl;
    if (k == 0) {
      sl = 1.0;
      cl = (1.0 - 2.0 * q1) + 2.0 * q1;
    } else {
      const double a = static_cast<double>(k) * f0_over_fs;
      const double sp = sinpi(a);
      sl = sp / (kPi * a);
      cl = (1.0 - 2.0 * q1) + 2.0 * q1 * (1.0 - 2.0 * sp * sp);
    }
    P[k] = re * sl * cl * inv_n;                        // == .. / N: N is a power of two
  });
  WH_STAMP(0, 8);
  block_irfft<kCtMaxLr, LGN>(Z, lgn, tw, [&](int k) { cplx c; c.re = P[k]; c.im = 0.0; return c; });
  WH_STAMP(0, 9);
  char *out_at = reinterpret_cast<char *>(p.spectrogram + (p.out_row ? (size_t)p.out_row[u] + f : fi) * p.out_stride) + p.out_col_bytes;
  if (p.out_f32) {
    float *out = reinterpret_cast<float *>(out_at);
    block_map<4, double>(half + 1, [&](int i) { return exp(rfft_in(Z, i)); }, [&](int i, double v) { out[i] = static_cast<float>(v); });
  } else {
    double *out = reinterpret_cast<double *>(out_at);
    block_map<4, double>(half + 1, [&](int i) { return exp(rfft_in(Z, i)); }, [&](int i, double v) { out[i] = v; });
  }
  WH_STAMP(0, 10);
}

// ---------------------------------------------------------------------------
size_t ct_spectrum_lds_bytes(int lg_fft) {
  int N = 1 << lg_fft, nb = N / 2 + 1;
  return sizeof(double) * (size_t)(N + nb + 1 + ((nb + 1) & 1) + 64 + N / 8 + 2);
}
size_t ct_envelope_lds_bytes(int lg_fft) {
  int N = 1 << lg_fft, nb = N / 2 + 1;
  return sizeof(double) * (size_t)(ct_seg_cap(N) + nb + 1 + ((nb + 1) & 1) + 64 + N / 8 + 2);
}
// room for the longest segment rounded up to whole rings of ct_scan, plus the ring it prefetches beyond
int ct_seg_stride(int fft_size) {
  const int ring = kScanDepth * kScanBatch;
  return ((ct_seg_cap(fft_size) + ring - 1) / ring + 1) * ring;
}

size_t ct_max_draws_per_frame(int fft_size) { return (size_t)fft_size + fft_size / 2 + 1; }   // window < fft_size, + bins

void launch_cheaptrick(const CtParams &p, int max_frames_all, hipStream_t stream) {
  if (!p.skip_prepare) WH_BLOCKS(ct_prepare, dim3(p.b.n_utt), 1024, 64 * sizeof(double), stream, p);    // 1024 threads: a 10 s utterance is two scans
  const int max_frames = imin(max_frames_all, p.frame_hi) - p.frame_lo;          // frames of the range
  if (max_frames <= 0) return;
  const dim3 grid(max_frames, p.b.n_utt);
  const size_t lds1 = ct_spectrum_lds_bytes(p.lg_fft), lds3 = ct_envelope_lds_bytes(p.lg_fft);
  const dim3 scan_grid((max_frames + WAVE - 1) / WAVE, p.b.n_utt);
#ifdef WORLD_EMU
  devrt::launch_blocks("ct_spectrum", ct_spectrum<8192, 0>, grid, 256, lds1, stream, p);
  WH_BLOCKS(ct_scan, scan_grid, WAVE, 0, stream, p);
  devrt::launch_blocks("ct_envelope", ct_envelope<0>, grid, 256, lds3, stream, p);
#else
  // Workgroup size follows the transform: fft_size 1024 (fs <= 24 kHz) runs with 128 threads (64 butterflies per
  // radix-8 stage; measured 1.33 ms for 64 x 1001 frames against 1.60 with 256 threads and 1.49 with 64), 2048 and
  // 4096 with 256.  The three sizes the sampling rates of speech lead to get compile-time plans.
  if (p.lg_fft == 10) {
    devrt::launch_blocks("ct_spectrum", ct_spectrum<8, 10>, grid, 128, lds1, stream, p);
    WH_BLOCKS(ct_scan, scan_grid, WAVE, 0, stream, p);
    devrt::launch_blocks("ct_envelope", ct_envelope<10>, grid, 128, lds3, stream, p);
  } else if (p.lg_fft == 11) {
    devrt::launch_blocks("ct_spectrum", ct_spectrum<8, 11>, grid, 256, lds1, stream, p);
    WH_BLOCKS(ct_scan, scan_grid, WAVE, 0, stream, p);
    devrt::launch_blocks("ct_envelope", ct_envelope<11>, grid, 256, lds3, stream, p);
  } else if (p.lg_fft == 12) {
    devrt::launch_blocks("ct_spectrum", ct_spectrum<16, 12>, grid, 256, lds1, stream, p);
    WH_BLOCKS(ct_scan, scan_grid, WAVE, 0, stream, p);
    devrt::launch_blocks("ct_envelope", ct_envelope<12>, grid, 256, lds3, stream, p);
  } else if (p.lg_fft == 13) {
    devrt::launch_blocks("ct_spectrum", ct_spectrum<16, 0, 512>, grid, 512, lds1, stream, p);
    WH_BLOCKS(ct_scan, scan_grid, WAVE, 0, stream, p);
    devrt::launch_blocks("ct_envelope", ct_envelope<0, 512>, grid, 512, lds3, stream, p);
  } else {
    devrt::launch_blocks("ct_spectrum", ct_spectrum<8, 0>, grid, 128, lds1, stream, p);
    WH_BLOCKS(ct_scan, scan_grid, WAVE, 0, stream, p);
    devrt::launch_blocks("ct_envelope", ct_envelope<0>, grid, 128, lds3, stream, p);
  }
#endif
}

}  // namespace world_hip
