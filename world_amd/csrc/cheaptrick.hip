// cheaptrick.hip -- CheapTrick spectral envelope on gfx950.
//
// Replaces CheapTrick() / CheapTrickGeneralBody() (reference src/cheaptrick.cpp:22-229)
// and the helpers it pulls from src/common.cpp (DCCorrection :56-75,
// LinearSmoothing :27-111).  The reference walks frames serially with one RNG
// stream; here every frame is an independent workgroup:
//
//   ct_prepare : per utterance, prefix-sum of randn() calls consumed per frame
//                -> each frame's xorshift128 state by GF(2) jump-ahead.
//   ct_frame   : one 256-thread workgroup per (frame, utterance), everything in
//                LDS: F0-adaptive window + noise -> r2c FFT -> |X|^2 -> DC
//                correction -> mirrored prefix sum (kept SERIAL and in FP64
//                order: its rounding is visible in low-energy bins, SURVEY.md
//                H2) -> rectangular smoothing -> +|randn|*eps -> log -> r2c FFT
//                (cepstrum) -> lifter -> c2r FFT -> exp -> HBM.
//
// HBM traffic per frame: the window's samples of x (L2-resident: adjacent
// frames overlap ~97%) and one row of the spectrogram written once.
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
// Largest butterfly of ct_frame's transforms (log2): a 1024-point complex transform has only 64
// radix-16 butterflies for 256 threads; smaller butterflies keep more threads busy per stage.
constexpr int kCtMaxLr = 3;
// PER: samples of the window a thread owns (fft_size / 256 on the GPU)
template <int PER>
__global__ void __launch_bounds__(256, 4) ct_frame(CtParams p) {
  DYN_LDS(lds);
  const int lgn = p.lg_fft, N = 1 << lgn, half = N / 2, nb = half + 1;
  const int fs = p.b.fs;
  const int u = blockIdx.y, f = xcd_grouped(blockIdx.x, gridDim.x);
  if (f >= p.b.n_frames[u]) return;
  const bool trace_me = f == 1000; (void)trace_me;
  WH_STAMP(0, 0);

  // LDS carve-up (doubles): Z: N | seg overflow | P: nb+1 | scratch: 64 | twiddles.  The smoothing
  // work area `seg` (up to nb + 2(N/3+2) + 1 values) starts on top of Z, which is dead whenever
  // seg is live, and runs into the small overflow strip behind it: 32 KB per frame instead of 48.
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds);
  double *seg = Zr;
  double *P = Zr + ct_seg_cap(N);
  double *scratch = P + (nb + 1) + ((nb + 1) & 1);
  const TwLds tw = stage_twiddles(scratch + 64, lgn, p.tab.tw);

  const double *x = p.b.x + (size_t)u * p.b.x_stride;
  const int x_len = p.b.x_len[u];
  const double pos = p.tpos[(size_t)u * p.b.f_stride + f];
  const double cf0 = ct_effective_f0(p.f0[(size_t)u * p.b.f_stride + f], p.f0_floor);
  const uint32_t *noise = p.noise + p.offsets[(size_t)u * p.b.f_stride + f];
  const int tid = threadIdx.x, nt = blockDim.x;

  WH_STAMP(0, 1);
  // ---- GetWindowedWaveform (cheaptrick.cpp:87-142) -------------------------
  const int hw = mround(1.5 * fs / cf0);
  const int wlen = 2 * hw + 1;
  const int origin = mround(pos * fs + 0.001);
  // One pass, one block reduction: with w the raw window, a = x w and n the dither, the
  // reference's normalised window is c w (c = 1/sqrt(sum w^2)), its waveform v = c a + n, and
  // the DC-balanced result v - c w (sum v / sum c w) -- all linear in four sums.  A thread keeps
  // the three values of each of its samples in registers (the FFT butterflies, not this phase,
  // set the register budget); the frame's draws (sample order) come from the noise stream.
  const double win_scale = 1.0 / 1.5 / fs * cf0;        // position * f0 = (i - hw) / 1.5 / fs * f0
  double wv[PER], av[PER], nv[PER];
  double s_ww = 0.0, s_a = 0.0, s_n = 0.0, s_w = 0.0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * nt;
    wv[q] = 0.0; av[q] = 0.0; nv[q] = 0.0;
    if (i < wlen) {
      const double w = 0.5 * cospi(win_scale * (i - hw)) + 0.5;   // cos(pi * position * f0), cheaptrick.cpp:101-102
      const double a = x[imin(x_len - 1, imax(0, origin + i - hw))] * w, n = randn_value(noise[i]) * kTiny;
      wv[q] = w; av[q] = a; nv[q] = n;
      s_ww += w * w; s_a += a; s_n += n; s_w += w;
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
  block_rfft<kCtMaxLr>(Z, lgn, tw, [&](int k, double re, double im) { P[k] = re * re + im * im; });

  WH_STAMP(0, 3);
  // DCCorrection (common.cpp:56-75); replica staged in seg, then added
  {
    const int upper = 2 + static_cast<int>(cf0 * N / fs);
    const int nrep = upper - 1;
    const double dx = -static_cast<double>(fs) / N;
    for (int i = tid; i < nrep; i += nt) {
      double axis = static_cast<double>(i) * fs / N;
      seg[i] = interp_uniform(cf0, dx, P, upper + 1, axis);
    }
    __syncthreads();
    for (int i = tid; i < nrep; i += nt) P[i] = P[i] + seg[i];
    __syncthreads();
  }

  WH_STAMP(0, 4);
  // ---- LinearSmoothing (common.cpp:27-111), width = 2/3 f0 -----------------
  const double width = cf0 * 2.0 / 3.0;
  const int bnd = static_cast<int>(width * N / fs) + 1;
  const int seg_len = half + 2 * bnd + 1;
  const double inv_n = 1.0 / N;
  block_map<4, double>(seg_len,
    [&](int i) {
      double m;
      if (i < bnd) m = P[bnd - i];
      else if (i < half + bnd) m = P[i - bnd];
      else m = P[half - (i - (half + bnd))];
      return m * fs * inv_n;                           // == m * fs / N: N is a power of two
    },
    [&](int i, double v) { seg[i] = v; });
  __syncthreads();
  WH_STAMP(0, 5);
  if (tid < WAVE) {                     // the order-sensitive serial prefix sum (one lane)
#ifndef WORLD_EMU
    // the rest of the workgroup waits for this dependent chain: let its wave issue ahead of the
    // other workgroups' waves that share the SIMD
    __builtin_amdgcn_s_setprio(3);
#endif
  }
  if (tid == 0) {
    // strictly left-to-right FP64 additions; only the LDS traffic is batched (16 loads in
    // flight, 16 dependent adds, 16 stores).  The chain runs at the FP64 add's dependent-issue
    // latency (~38 cycles measured in situ): ~40k cycles per frame, the largest single phase.
    constexpr int kB = 16;
    double acc = seg[0];
    int i0 = 1;
    for (; i0 + kB <= seg_len; i0 += kB) {
      double v[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) v[q] = seg[i0 + q];
#pragma unroll
      for (int q = 0; q < kB; ++q) { acc = v[q] + acc; v[q] = acc; }
#pragma unroll
      for (int q = 0; q < kB; ++q) seg[i0 + q] = v[q];
    }
    for (; i0 < seg_len; ++i0) { acc = seg[i0] + acc; seg[i0] = acc; }
  }
#ifndef WORLD_EMU
  if (tid < WAVE) __builtin_amdgcn_s_setprio(0);
#endif
  __syncthreads();
  WH_STAMP(0, 6);
  {
    const double origin_axis = -(bnd - 0.5) * fs / N;
    const double step = static_cast<double>(fs) / N;
    // hi - lo cancels up to 12 digits where the envelope sits at the noise floor:
    // the interpolation weights and the final quotient keep the reference's exact operations
    // (true divisions), so that every product rounds as it does there.
    block_map<4, double>(half + 1,
      [&](int i) {
        double fa = static_cast<double>(i) * inv_n * fs - width / 2.0;
        double lo = interp_uniform(origin_axis, step, seg, seg_len, fa);
        fa += width;
        double hi = interp_uniform(origin_axis, step, seg, seg_len, fa);
        double smoothed = (hi - lo) / width;
        // AddInfinitesimalNoise: the per-bin draws continue the frame's stream after the
        // window draws (cheaptrick.cpp:147-151); then the log of SmoothingWithRecovery (:39-42)
        return log(smoothed + fabs(randn_value(noise[wlen + i])) * kEps);
      },
      [&](int i, double lg) { P[i] = lg; });
  }

  WH_STAMP(0, 7);
  // ---- SmoothingWithRecovery (cheaptrick.cpp:22-57) -------------------------
  // the symmetric extension of the log spectrum is read by the first FFT stage directly from P
  const double q1 = p.q1, inv_fs = 1.0 / fs;
  auto mirrored = [&](int i) { return i <= half ? P[i] : P[N - i]; };
  block_rfft_from<kCtMaxLr>(Z, lgn, tw, [&](int n) { cplx v; v.re = mirrored(2 * n); v.im = mirrored(2 * n + 1); return v; },
                  [&](int k, double re, double im) {
    (void)im;
    double sl, cl;
    if (k == 0) {
      sl = 1.0;
      cl = (1.0 - 2.0 * q1) + 2.0 * q1;
    } else {
      double quef = static_cast<double>(k) * inv_fs;
      sl = sin(kPi * cf0 * quef) / (kPi * cf0 * quef);
      cl = (1.0 - 2.0 * q1) + 2.0 * q1 * cos(2.0 * kPi * quef * cf0);
    }
    P[k] = re * sl * cl * inv_n;                        // == .. / N: N is a power of two
  });
  WH_STAMP(0, 8);
  block_irfft<kCtMaxLr>(Z, lgn, tw, [&](int k) { cplx c; c.re = P[k]; c.im = 0.0; return c; });
  WH_STAMP(0, 9);
  double *out = p.spectrogram + ((size_t)u * p.b.f_stride + f) * nb;
  block_map<4, double>(half + 1, [&](int i) { return exp(rfft_in(Z, i)); }, [&](int i, double v) { out[i] = v; });
  WH_STAMP(0, 10);
}

// ---------------------------------------------------------------------------
size_t ct_frame_lds_bytes(int lg_fft) {
  int N = 1 << lg_fft, nb = N / 2 + 1;
  return sizeof(double) * (size_t)(ct_seg_cap(N) + nb + 1 + ((nb + 1) & 1) + 64 + N / 4 + 2);
}

size_t ct_max_draws_per_frame(int fft_size) { return (size_t)fft_size + fft_size / 2 + 1; }   // window < fft_size, + bins

void launch_cheaptrick(const CtParams &p, int max_frames, hipStream_t stream) {
  WH_BLOCKS(ct_prepare, dim3(p.b.n_utt), 256, 64 * sizeof(double), stream, p);
#ifdef WORLD_EMU
  devrt::launch_blocks("ct_frame", ct_frame<4096>, dim3(max_frames, p.b.n_utt), 256, ct_frame_lds_bytes(p.lg_fft), stream, p);
#else
  // fft_size 1024 (fs <= 24 kHz): 128 threads (64 butterflies per radix-8 stage; measured 1.33 ms for 64 x 1001
  // frames against 1.60 with 256 threads and 1.49 with 64); 2048 stays at 256 (128 threads: 147 us against 116)
  if (p.lg_fft <= 10) devrt::launch_blocks("ct_frame", ct_frame<8>, dim3(max_frames, p.b.n_utt), 128, ct_frame_lds_bytes(p.lg_fft), stream, p);
  else if (p.lg_fft <= 11) devrt::launch_blocks("ct_frame", ct_frame<8>, dim3(max_frames, p.b.n_utt), 256, ct_frame_lds_bytes(p.lg_fft), stream, p);
  else devrt::launch_blocks("ct_frame", ct_frame<16>, dim3(max_frames, p.b.n_utt), 256, ct_frame_lds_bytes(p.lg_fft), stream, p);
#endif
}

}  // namespace world_hip
