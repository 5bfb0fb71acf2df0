// synthesis.hip -- WORLD waveform synthesis on the GPU (reference src/synthesis.cpp).
//
//   sy_increments  : per sample, interp1 of f0 / vuv onto the sample grid and the phase
//                    increment 2 pi f0 / fs                                        (:225-318)
//   sy_phase_serial: the running phase, strictly left to right, one wavefront per utterance
//   sy_detect      : a pulse sits where the wrapped phase jumps by more than pi
//                    (GetPulseLocationsForTimeBase, :251-290)
//   sy_compact     : pulse index / fractional time shift lists in time order
//   sy_pulse       : ONE workgroup per pulse: time-varying envelope and aperiodic ratio,
//                    minimum-phase spectra (common.cpp:182-220) of the periodic and the
//                    aperiodic part, fractional delay, noise excitation from the randn
//                    stream, two inverse transforms -> the pulse's impulse response (:38-218)
//   sy_overlap_add : every output sample sums the responses that cover it IN PULSE ORDER,
//                    so the accumulation rounds exactly like the reference's loop (:376-385)
//
// The phase accumulation has to round exactly like the reference's loop: unvoiced stretches
// run at the default 500 Hz, and at fs = 16 / 32 / 48 kHz that is an exact divisor of the
// sampling rate, so every unvoiced pulse lands within an ulp of the wrap boundary and which
// sample it is detected at depends on the last bit of the running sum: one add chain per
// utterance (sy_phase_serial), 1.3 ms of dependent adds for 10 s of 48 kHz audio; everything
// else here is parallel.  The randn stream is consumed strictly in pulse order, noise_size draws per pulse, so pulse
// p starts at draw pidx[p] - pidx[0].
#include "synthesis.h"

namespace world_hip {

// interp1 (matlabfunctions.cpp:136-176) of the coarse f0 / vuv tracks at sample i.
// Knots are i * frame_period, i = 0 .. nf (nf + 1 of them); the value at knot nf is the
// linear extrapolation 2 v[nf-1] - v[nf-2] (synthesis.cpp:239-243).
__device__ __forceinline__ void coarse_tracks(const SynthParams &p, const double *f0, int nf, int i, double *f0_out,
                                              int *vuv_out) {
  const double t = i / static_cast<double>(p.fs);
  const double fp = p.frame_period;
  const int n = nf + 1;
  // count of knots <= t, from a guess corrected against the knots themselves (k * fp)
  int g = static_cast<int>(t / fp);
  if (g > nf) g = nf;
  while (g + 1 <= nf && (g + 1) * fp <= t) ++g;
  while (g > 0 && g * fp > t) --g;
  int c = g + 1;                                   // knots 0..g are <= t
  int k = c < 1 ? 1 : (c > n - 1 ? n - 1 : c);
  auto cf0 = [&](int j) {
    if (j < nf) return f0[j] < p.lowest_f0 ? 0.0 : f0[j];
    const double a = f0[nf - 1] < p.lowest_f0 ? 0.0 : f0[nf - 1], b = f0[nf - 2] < p.lowest_f0 ? 0.0 : f0[nf - 2];
    return a * 2 - b;
  };
  auto cvuv = [&](int j) {
    if (j < nf) return (f0[j] < p.lowest_f0 ? 0.0 : f0[j]) == 0.0 ? 0.0 : 1.0;
    const double a = (f0[nf - 1] < p.lowest_f0 ? 0.0 : f0[nf - 1]) == 0.0 ? 0.0 : 1.0;
    const double b = (f0[nf - 2] < p.lowest_f0 ? 0.0 : f0[nf - 2]) == 0.0 ? 0.0 : 1.0;
    return a * 2 - b;
  };
  const double x0 = (k - 1) * fp, x1 = k * fp;
  const double s = (t - x0) / (x1 - x0);
  const double f = cf0(k - 1) + s * (cf0(k) - cf0(k - 1));
  const double v = cvuv(k - 1) + s * (cvuv(k) - cvuv(k - 1));
  const int voiced = v > 0.5 ? 1 : 0;
  *vuv_out = voiced;
  *f0_out = voiced ? f : kDefaultF0;               // synthesis.cpp:307-311
}

__global__ void __launch_bounds__(kSyThreads) sy_increments(SynthParams p) {
  const int u = blockIdx.y, tid = threadIdx.x;
  const int n = p.y_len[u], nf = p.n_frames[u];
  const double *f0 = p.f0 + (size_t)u * p.f_stride;
  for (int i = blockIdx.x * kSyTile + tid; i < imin(n, (blockIdx.x + 1) * kSyTile); i += kSyThreads) {
    double f; int voiced;
    coarse_tracks(p, f0, nf, i, &f, &voiced);
    p.inc[(size_t)u * p.y_stride + i] = 2.0 * kPi * f / p.fs;        // synthesis.cpp:255,258
    p.flags[(size_t)u * p.y_stride + i] = (unsigned char)voiced;
  }
}

// total_phase[i] = total_phase[i-1] + increment[i], in place, strictly left to right: one wavefront per utterance.
// A dependent FP64 add follows its producer after ~6 cycles (tools/probe/fp64_chain.hip), so the chain itself is
// 1.3 ms for 10 s of 48 kHz audio; a lone lane fetching its own values spent four times that waiting for them.  Here
// every lane owns kPhaseRun consecutive samples of a chunk of WAVE * kPhaseRun; the lanes take turns in order -- a
// turn is the kPhaseRun adds of one lane under a one-lane mask, the running sum handed on through a scalar register
// (~60 cycles of mask, branch and v_readlane per turn: hence the long runs) -- and the next chunk's samples are
// requested before the turns begin.  Bit for bit the sum of one lane walking the array.
// (Measured and dropped: a second wavefront doing the stores out of LDS, so that the adder's memory counter only ever
// holds loads -- 2.9 ms where this takes 2.4 at kPhaseRun = 16: the chunk's store latency is not what is exposed.)
constexpr int kPhaseRun = 32;
__global__ void __launch_bounds__(WAVE) sy_phase_serial(SynthParams p) {
  const int u = blockIdx.x, lane = lane_id();
  const int n = p.y_len[u];
  if (n <= 0) return;
  double *a = p.inc + (size_t)u * p.y_stride;
  constexpr int kChunk = WAVE * kPhaseRun;
  auto fetch = [&](double (&v)[kPhaseRun], int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < kPhaseRun; ++q) v[q] = a[imin(n - 1, i0 + lane * kPhaseRun + q)];
  };
  double cur[kPhaseRun], nxt[kPhaseRun];
  fetch(cur, 0);
  double carry = -0.0;                 // -0.0 + x == x bit for bit: the first sample passes through, as in the reference
  for (int i0 = 0; i0 < n; i0 += kChunk) {
    fetch(nxt, i0 + kChunk);           // (past the end the clamped addresses repeat the last sample: unused)
    double last = 0.0;
    for (int turn = 0; turn < WAVE; ++turn) {
      if (lane == turn) {
        double acc = carry;
#pragma unroll
        for (int q = 0; q < kPhaseRun; ++q) { acc = acc + cur[q]; cur[q] = acc; }
        last = acc;
      }
      carry = wave_pick(last, turn);
    }
#pragma unroll
    for (int q = 0; q < kPhaseRun; ++q) {
      const int i = i0 + lane * kPhaseRun + q;
      if (i < n) a[i] = cur[q];
    }
#pragma unroll
    for (int q = 0; q < kPhaseRun; ++q) cur[q] = nxt[q];
  }
}

__global__ void __launch_bounds__(kSyThreads) sy_detect(SynthParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int u = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int n = p.y_len[u];
  const double *total = p.inc + (size_t)u * p.y_stride;
  const int i0 = blk * kSyTile + tid * kSyPer;
  const double two_pi = 2.0 * kPi;
  int count = 0;
  double w = i0 < n ? fmod(total[i0], two_pi) : 0.0;
#pragma unroll
  for (int q = 0; q < kSyPer; ++q) {
    const int i = i0 + q;
    if (i + 1 < n) {
      const double wn = fmod(total[i + 1], two_pi);
      unsigned char *fl = p.flags + (size_t)u * p.y_stride + i;
      if (fabs(wn - w) > kPi) { *fl = (unsigned char)(*fl | 2); ++count; }
      w = wn;
    }
  }
  int tot;
  block_excl_scan_int(count, &tot, scratch);
  if (tid == 0) p.blk_cnt[(size_t)u * p.nblk + blk] = tot;
}

__global__ void __launch_bounds__(kSyThreads) sy_compact(SynthParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  int *first = reinterpret_cast<int *>(scratch + 64);
  const int u = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int n = p.y_len[u];
  const double *total = p.inc + (size_t)u * p.y_stride;
  if (tid == 0) {
    int c0 = 0;
    for (int c = 0; c < blk; ++c) c0 += p.blk_cnt[(size_t)u * p.nblk + c];
    first[0] = c0;
    if (blk == p.nblk - 1) {                        // the last tile publishes the pulse count
      const int all = c0 + p.blk_cnt[(size_t)u * p.nblk + blk];
      p.np[u] = all < p.pulse_cap ? all : p.pulse_cap;
      if (all > p.pulse_cap) atomicMax(p.need, all);  // the caller is told (world_hip_sync / the drop-in retries)
    }
  }
  __syncthreads();
  const int i0 = blk * kSyTile + tid * kSyPer;
  const unsigned char *fl = p.flags + (size_t)u * p.y_stride;
  int count = 0;
#pragma unroll
  for (int q = 0; q < kSyPer; ++q) if (i0 + q + 1 < n && (fl[i0 + q] & 2)) ++count;
  int tot, at = first[0] + block_excl_scan_int(count, &tot, scratch);
  const double two_pi = 2.0 * kPi;
#pragma unroll
  for (int q = 0; q < kSyPer; ++q) {
    const int i = i0 + q;
    if (i + 1 < n && (fl[i] & 2)) {
      if (at < p.pulse_cap) {
        // the exact crossing between samples i and i+1 (synthesis.cpp:268-279)
        const double y1 = fmod(total[i], two_pi) - two_pi;
        const double y2 = fmod(total[i + 1], two_pi);
        const double x = -y1 / (y2 - y1);
        p.pidx[(size_t)u * p.pulse_cap + at] = i;
        p.pshift[(size_t)u * p.pulse_cap + at] = x / p.fs;
      }
      ++at;
    }
  }
}

// ---------------------------------------------------------------------------
// One pulse.  LDS: Z (N complex) | C (N/2+1 complex; its head doubles as the log spectrum
// the first FFT stage reads) | scratch | twiddles.
// (fft_size 8192 -- the default above 96 kHz: Z alone is 128 KB, and C lives in the pulse's slot of the response buffer)
int synth_tile_samples() { return kSyTile; }
size_t synth_pulse_lds_bytes(int lg_fft) {
  const size_t N = (size_t)1 << lg_fft;
  const size_t c_lds = lg_fft > 12 ? 0 : 2 * (N / 2 + 1) + 2;
  return sizeof(double) * (2 * N + 16 + c_lds + 64 + twiddle_lds_doubles(lg_fft - 1));
}

// GetMinimumPhaseSpectrum (common.cpp:182-220): LG[0..H] = log spectrum in; C[0..H] = the
// minimum-phase spectrum out.  LG aliases the head of C (see above).
__device__ __forceinline__ void minimum_phase(cplx *Z, cplx *C, const double *LG, int lgn, const TwLds &tw) {
  const int N = 1 << lgn, H = N / 2, tid = threadIdx.x, nt = blockDim.x;
  // r2c of the mirrored log spectrum -> cepstrum, folded to its causal half on the way out
  block_rfft_from<3>(Z, lgn, tw,
    [&](int n) {
      cplx v;
      const int a = 2 * n, b = 2 * n + 1;
      v.re = a <= H ? LG[a] : LG[N - a];
      v.im = b <= H ? LG[b] : LG[N - b];
      return v;
    },
    [&](int k, double re, double im) {
      // cepstrum[k] = (re, -im) scaled by 1, 2.., 1; the forward c2c plan then transforms its
      // conjugate (fft.cpp:62-71), i.e. (s re, + s im)
      const double s = (k == 0 || k == H) ? 1.0 : 2.0;
      cplx c; c.re = s * re; c.im = s * im;
      C[k] = c;
    });
  const FftPlan plan = make_plan_max(lgn, 3);
  block_cfft_dif_from<3>(Z, plan, tw, [&](int j) { cplx z; z.re = 0.0; z.im = 0.0; return j <= H ? C[j] : z; });
  const double inv_n = 1.0 / N;
  for (int k = tid; k <= H; k += nt) {
    const cplx x = Z[fft_slot(plan, k)];
    const double t = exp(x.re * inv_n);
    double sn, cs;
    sincos(x.im * inv_n, &sn, &cs);
    cplx m; m.re = t * cs; m.im = t * sn;
    C[k] = m;
  }
  __syncthreads();
}

// NMAX: the largest fft_size of the instantiation (a thread's samples of the periodic response wait in registers).
// 8192 points: the N-point complex transform of the minimum-phase step is 128 KB of LDS by itself, so the pulse's spectrum C
// (N/2 + 1 complex) lives in global memory -- in the pulse's own slot of the response buffer, two doubles longer for it
// (p.resp_stride); the response is written there last, when nobody reads C any more.
template <int NMAX>
__global__ void __launch_bounds__(kSyThreads) sy_pulse(SynthParams p) {
  DYN_LDS(lds);
  const int u = blockIdx.y, pi = blockIdx.x;
  const int np = p.np[u];
  if (pi >= np) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lgn = p.lg_fft, N = 1 << lgn, H = N / 2, nb = H + 1;
  constexpr bool c_global = NMAX > 4096;
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *const slot = p.resp + ((size_t)u * p.pulse_cap + pi) * p.resp_stride;
  cplx *C = c_global ? reinterpret_cast<cplx *>(slot) : Z + N + 8;
  double *LG = reinterpret_cast<double *>(C);
  double *scratch = c_global ? reinterpret_cast<double *>(Z + N + 8) : reinterpret_cast<double *>(C + nb + 1);
  // table of the half-size transforms; the N-point complex transform derives its odd twiddles
  const TwLds tw = stage_twiddles(scratch + 64, lgn - 1, p.tab.tw);

  const int *pidx = p.pidx + (size_t)u * p.pulse_cap;
  const int idx = pidx[pi];
  const int nxt = pi + 1 < np - 1 ? pi + 1 : np - 1;
  const int noise_size = pidx[nxt] - idx;                              // synthesis.cpp:369-370
  const double vuv = (p.flags[(size_t)u * p.y_stride + idx] & 1) ? 1.0 : 0.0;
  const double t = idx / static_cast<double>(p.fs);                    // pulse_locations = time_axis[i]
  const int nf = p.n_frames[u];
  // GetSpectralEnvelope / GetAperiodicRatio (:140-180)
  const double fp = p.frame_period;
  const int ff = imin(nf - 1, static_cast<int>(floor(t / fp))), fc = imin(nf - 1, static_cast<int>(ceil(t / fp)));
  const double wgt = t / fp - ff;
  const double *sp0 = p.sp + ((size_t)u * p.f_stride + ff) * nb, *sp1 = p.sp + ((size_t)u * p.f_stride + fc) * nb;
  const double *ap0 = p.ap + ((size_t)u * p.f_stride + ff) * nb, *ap1 = p.ap + ((size_t)u * p.f_stride + fc) * nb;
  auto envelope = [&](int i) {
    return ff == fc ? fabs(sp0[i]) : (1.0 - wgt) * fabs(sp0[i]) + wgt * fabs(sp1[i]);
  };
  auto safe = [](double x) { return fmax(0.001, fmin(0.999999999999, x)); };   // common.h:111-113
  auto ratio = [&](int i) {
    const double a = ff == fc ? safe(ap0[i]) : (1.0 - wgt) * safe(ap0[i]) + wgt * safe(ap1[i]);
    return a * a;                                                          // pow(.., 2.0)
  };

  // ---- periodic response (GetPeriodicResponse, :103-135); this thread's samples stay in registers
  constexpr int kPer = NMAX / kSyThreads;            // fft_size <= NMAX
  double per[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) per[q] = 0.0;
  const bool has_periodic = !(vuv <= 0.5 || ratio(0) > 0.999);
  if (has_periodic) {
    __syncthreads();
    for (int i = tid; i < nb; i += nt) LG[i] = log(envelope(i) * (1.0 - ratio(i)) + kTiny) / 2.0;
    minimum_phase(Z, C, LG, lgn, tw);
    // fractional delay by a linear phase (GetSpectrumWithFractionalTimeShift, :86-98; the
    // reference takes sin = sqrt(1 - cos^2), i.e. |sin|)
    const double coef = 2.0 * kPi * p.pshift[(size_t)u * p.pulse_cap + pi] * p.fs / N;
    for (int i = tid; i < nb; i += nt) {
      const cplx m = C[i];
      const double re2 = cos(coef * i), im2 = sqrt(1.0 - re2 * re2);
      cplx s; s.re = m.re * re2 + m.im * im2; s.im = m.im * re2 - m.re * im2;
      C[i] = s;
    }
    block_irfft<3>(Z, lgn, tw, [&](int k) { return C[k]; });
    __syncthreads();
    // fftshift + RemoveDCComponent in place (:72-80): the first half is REPLACED by -dc * remover
    double dc = 0.0;
    for (int n = tid; n < H; n += nt) dc += rfft_in(Z, n);               // shifted index n + H
    dc = block_sum(dc, scratch);
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int i = tid + q * nt;
      if (i < N) per[q] = i < H ? -dc * p.dc_remover[i] : rfft_in(Z, i - H) - dc * p.dc_remover[i];
    }
  }

  // ---- aperiodic response (GetAperiodicResponse, :38-66)
  __syncthreads();
  for (int i = tid; i < nb; i += nt) LG[i] = vuv != 0.0 ? log(envelope(i) * ratio(i)) / 2.0 : log(envelope(i)) / 2.0;
  minimum_phase(Z, C, LG, lgn, tw);
  // GetNoiseSpectrum (:19-33): this pulse's draws, mean removed, zero padded; its spectrum is
  // multiplied into the minimum-phase spectrum as the merge step emits it
  const uint32_t *noise = p.noise + (idx - pidx[0]);
  double avg = 0.0;
  for (int i = tid; i < noise_size; i += nt) avg += randn_value(noise[i]);
  avg = block_sum(avg, scratch) / noise_size;
  block_rfft_from<3>(Z, lgn, tw,
    [&](int n) {
      cplx v;
      const int a = 2 * n, b = 2 * n + 1;
      v.re = a < noise_size ? randn_value(noise[a]) - avg : 0.0;
      v.im = b < noise_size ? randn_value(noise[b]) - avg : 0.0;
      return v;
    },
    [&](int k, double re, double im) {
      const cplx m = C[k];
      cplx s; s.re = m.re * re - m.im * im; s.im = m.re * im + m.im * re;
      C[k] = s;
    });
  block_irfft<3>(Z, lgn, tw, [&](int k) { return C[k]; });
  __syncthreads();
  // GetOneFrameSegment (:213-218): (periodic sqrt(noise_size) + fftshift(aperiodic)) / fft_size
  const double sq = sqrt(static_cast<double>(noise_size));
  double *out = slot;                                 // (8192 points: over C, which the transform above has consumed)
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int i = tid + q * nt;
    if (i < N) out[i] = (per[q] * sq + rfft_in(Z, (i + H) & (N - 1))) / N;
  }
}

// ---------------------------------------------------------------------------
__global__ void sy_overlap_add(SynthParams p) {
  const int n = flat_thread_x(), u = blockIdx.y;
  if (n >= p.y_len[u]) return;
  const int N = p.fft_size, H = N / 2;
  const int np = p.np[u];
  const int *pidx = p.pidx + (size_t)u * p.pulse_cap;
  // pulses whose response covers sample n: offset = pidx - H + 1 <= n < offset + N
  int lo = 0, hi = np;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (pidx[mid] < n - H) lo = mid + 1; else hi = mid; }
  double acc = 0.0;
  for (int q = lo; q < np; ++q) {
    const int offset = pidx[q] - H + 1;
    if (offset > n) break;
    acc += p.resp[((size_t)u * p.pulse_cap + q) * p.resp_stride + (n - offset)];
  }
  p.y[(size_t)u * p.y_stride + n] = acc;
}

void launch_synthesis(const SynthParams &p, int max_y, hipStream_t stream) {
  const size_t small = sizeof(double) * 80;
  WH_BLOCKS(sy_increments, dim3(p.nblk, p.n_utt), kSyThreads, 0, stream, p);
  WH_BLOCKS(sy_phase_serial, dim3(p.n_utt), WAVE, 0, stream, p);
  WH_BLOCKS(sy_detect, dim3(p.nblk, p.n_utt), kSyThreads, small, stream, p);
  WH_BLOCKS(sy_compact, dim3(p.nblk, p.n_utt), kSyThreads, small, stream, p);
  if (p.lg_fft <= 12) devrt::launch_blocks("sy_pulse", sy_pulse<4096>, dim3(p.pulse_cap, p.n_utt), kSyThreads, synth_pulse_lds_bytes(p.lg_fft), stream, p);
  else devrt::launch_blocks("sy_pulse", sy_pulse<8192>, dim3(p.pulse_cap, p.n_utt), kSyThreads, synth_pulse_lds_bytes(p.lg_fft), stream, p);
  WH_THREADS(sy_overlap_add, max_y, p.n_utt, 1, stream, p);
}

}  // namespace world_hip
