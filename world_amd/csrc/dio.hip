// dio.hip -- DIO F0 estimation on gfx950 (reference src/dio.cpp:578-648 and below).
//
// Same MI355X design as Harvest's front half (bandfilter.h): the reference's
// whole-utterance FFT filtering -- spectrum * low-cut filter spectrum
// (GetSpectrumForEstimation :60-106), then per channel spectrum * Nuttall low-pass
// and an inverse FFT (GetFilteredSignal :296-343) -- is the linear convolution
// with two short FIRs, evaluated directly out of LDS; the four zero-crossing
// detectors run on the LDS tile.  Candidates/scores (:441-572) are one thread per
// (frame, channel); the contour fix (:112-289) is frame-parallel for steps 1-2 and
// one wavefront per utterance (lanes = channels) for the two tracking sweeps.
#include "bandfilter.h"
#include "decimate.h"
#include "dio.h"
#include "trace.h"
WH_TRACE_DEFINE(dio)

namespace world_hip {

// ---- front end: decimate / copy, remove DC, low-cut FIR --------------------------
__global__ void __launch_bounds__(kDecThreads) dio_decimate_fwd(DioParams p, IirCoef c) {
  DYN_LDS(lds);
  const int u = blockIdx.y;
  dec_forward_block(p.b.x + (size_t)u * p.b.x_stride, p.b.x_len[u], 0, c, blockIdx.x, p.fwd + (size_t)u * p.m_stride,
                    reinterpret_cast<double *>(lds), dec_warm(p.ratio));
}
__global__ void __launch_bounds__(kDecThreads) dio_decimate_bwd(DioParams p, IirCoef c) {
  DYN_LDS(lds);
  const int u = blockIdx.y;
  dec_backward_block(p.fwd + (size_t)u * p.m_stride, p.b.x_len[u], 0, p.ratio, c, blockIdx.x, 0, p.y_len[u],
                     p.y + (size_t)u * p.y_stride, reinterpret_cast<double *>(lds), dec_warm(p.ratio));
}
__global__ void dio_copy_signal(DioParams p) {              // dio.cpp:71-72
  int u = blockIdx.y, i = flat_thread_x();
  if (i < p.b.x_len[u]) p.y[(size_t)u * p.y_stride + i] = p.b.x[(size_t)u * p.b.x_stride + i];
}
// y <- y - mean(y)  (dio.cpp:74-79): per-slice partial sums, then every workgroup adds the partials in the
// same fixed order (deterministic) and subtracts the mean -- one workgroup per utterance walked the
// whole signal twice at memory latency (0.13 ms for 5 s of 16 kHz audio).
constexpr int kDioMeanSlice = 4096;
__global__ void dio_partial_sums(DioParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int slice = blockIdx.x, u = blockIdx.y, n = p.y_len[u];
  const double *y = p.y + (size_t)u * p.y_stride;
  const int lo = slice * kDioMeanSlice, hi = imin(n, lo + kDioMeanSlice);
  double s = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) s += y[i];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) p.fwd[(size_t)u * p.m_stride + slice] = s;     // fwd is free after decimation
}
__global__ void dio_remove_mean(DioParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int slice = blockIdx.x, u = blockIdx.y, n = p.y_len[u];
  double *y = p.y + (size_t)u * p.y_stride;
  const int nslice = (n + kDioMeanSlice - 1) / kDioMeanSlice;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < nslice; ++k) s += p.fwd[(size_t)u * p.m_stride + k];
    scratch[0] = s / n;
  }
  __syncthreads();
  const double mean = scratch[0];
  const int lo = slice * kDioMeanSlice, hi = imin(n, lo + kDioMeanSlice);
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) y[i] -= mean;
}

// z[m] = (y * lowcut)[m] for m in [-C, y_len + C), stored at z[m + C]  (dio.cpp:40-53,85-101)
// TAPS_GLOBAL: the filter stays in global memory (2 round(fs / 50) + 1 taps: above ~185 kHz at speed 1 the taps, the
// input tile and the outputs no longer fit a CU's LDS together; the reference filters any fs, dio.cpp:85-101)
template <bool TAPS_GLOBAL>
__global__ void __launch_bounds__(kBpThreads) dio_lowcut(DioParams p) {
  DYN_LDS(lds);
  const int tile = blockIdx.x, u = blockIdx.y;
  const int out_len = p.y_len[u] + 2 * p.cut;
  const int t0 = tile * kTile;
  if (t0 >= out_len) return;
  BandJob job;
  job.in = p.y + (size_t)u * p.y_stride;
  job.in_len = p.y_len[u];
  job.n = out_len;
  job.taps = p.lowcut_taps;
  job.ntap = 2 * p.cut + 1;
  job.shift = 0;
  job.max_ntap = job.ntap;
  double *yt = reinterpret_cast<double *>(lds) + (TAPS_GLOBAL ? 0 : job.max_ntap + 1);
  double *s = yt + pad8(kTile + 2 + job.max_ntap + 3 + 8) + 1;
  if constexpr (TAPS_GLOBAL) {
    fir_tile(job, job.taps, t0, yt, s);
  } else {
    double *taps = reinterpret_cast<double *>(lds);
    for (int j = threadIdx.x; j < job.ntap; j += blockDim.x) taps[j] = job.taps[j];
    fir_tile(job, taps, t0, yt, s);
  }
  double *z = p.z + (size_t)u * p.z_stride;
  for (int k = threadIdx.x; k < kTile; k += blockDim.x)
    if (t0 + k < out_len) z[t0 + k] = s[pad8(k)];
}

// ---- the reference's mirror store (bandfilter.h): the utterance's two spectrum bins, with the
// low-cut filter's spectrum applied (dio.cpp:85-101), and the per-channel constants ----------
__global__ void dio_nyquist_slices(DioParams p) {       // partial sums per slice of kDioMeanSlice samples
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int slice = blockIdx.x, u = blockIdx.y;
  double s0, s1r, s1i;
  nyquist_pair(p.y + (size_t)u * p.y_stride, p.y_len[u], 2.0 / p.ref_fft[u], scratch, &s0, &s1r, &s1i,
               slice * kDioMeanSlice, (slice + 1) * kDioMeanSlice);
  if (threadIdx.x == 0) {                                 // fwd is free again once the mean is removed
    double *o = p.fwd + (size_t)u * p.m_stride + 4 * slice;
    o[0] = s0; o[1] = s1r; o[2] = s1i;
  }
}
__global__ void dio_nyquist_bins(DioParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int u = blockIdx.x, n = p.y_len[u], N = p.ref_fft[u];
  const double w = 2.0 / N;
  double s0 = 0.0, s1r = 0.0, s1i = 0.0;                   // the slices' sums, in slice order (thread 0 uses them)
  const double *part = p.fwd + (size_t)u * p.m_stride;
  for (int k = 0; k < (n + kDioMeanSlice - 1) / kDioMeanSlice; ++k) { s0 += part[4 * k]; s1r += part[4 * k + 1]; s1i += part[4 * k + 2]; }
  // low-cut spectrum at the two bins (real: the filter is symmetric about its centre tap)
  double l0 = 0.0, l1 = 0.0;
  for (int m = 1 + threadIdx.x; m <= p.cut; m += blockDim.x) {
    const double t = p.lowcut_taps[p.cut + m], sgn = (m & 1) ? -1.0 : 1.0;
    l0 += 2.0 * t * sgn;
    l1 += 2.0 * t * sgn * cospi(m * w);
  }
  block_sum2(l0, l1, scratch);
  if (threadIdx.x == 0) {
    const double c = p.lowcut_taps[p.cut];
    double *o = p.nyq + (size_t)u * 4;
    o[0] = (c + l0) * s0;
    o[1] = (c + l1) * s1r;
    o[2] = (c + l1) * s1i;
    o[3] = w;
  }
}

__global__ void dio_band_quirk(DioParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int band = blockIdx.x, u = blockIdx.y;
  const double *h = p.band_taps + p.band_off[band];
  const int L = 4 * p.band_hal[band];
  const double *ny = p.nyq + (size_t)u * 4;
  const double w = ny[3];
  double h0, h1r, h1i;                                 // H[N/2], H[N/2-1]
  nyquist_pair(h, L, w, scratch, &h0, &h1r, &h1i);
  if (threadIdx.x == 0)
    mirror_store_constants(ny[0], ny[1], ny[2], h0, h1r, h1i, w, p.quirk + ((size_t)u * p.nb + band) * 4);
}

// ---- channels: Nuttall low-pass + zero-crossing events ----------------------------
__global__ void __launch_bounds__(kBpThreads) dio_band_events(DioParams p) {
  const int seg = blockIdx.x, band = blockIdx.y, u = blockIdx.z;
  const int hal = p.band_hal[band];
  BandJob job;
  job.in = p.z + (size_t)u * p.z_stride;
  job.in_len = p.y_len[u] + 2 * p.cut;
  job.n = p.y_len[u];
  job.taps = p.band_taps + p.band_off[band];
  job.ntap = 4 * hal;
  job.shift = 2 * hal + p.cut;                    // index_bias 2*hal (dio.cpp:335), + the storage offset of z
  job.max_ntap = p.max_ntap;
  job.nseg = p.nseg;
  job.seg_events = p.seg_events + ((size_t)(u * p.nb + band) * 4) * p.nseg * kSegCap;
  job.seg_count = p.seg_count + ((size_t)(u * p.nb + band) * 4) * p.nseg;
  job.quirk = p.quirk + ((size_t)u * p.nb + band) * 4;
  job.quirk_delay = 2 * hal;                      // the term is a function of the undelayed index
  band_events_segment(job, seg);
}
__global__ void dio_compact_events(DioParams p) {
  DYN_LDS(lds);
  const int bf = blockIdx.x, u = blockIdx.y;
  const size_t list = (size_t)u * p.nb * 4 + bf;
  compact_event_segments(p.seg_events + list * p.nseg * kSegCap, p.seg_count + list * p.nseg, p.nseg, kSegCap,
                         p.events + list * p.ev_cap, p.ev_cap, p.ev_count + list, lds);
}

// ---- candidates and scores (dio.cpp:441-572), one thread per (frame, channel, utt) ----
__global__ void dio_candidates(DioParams p) {
  const int frame = flat_thread_x(), band = blockIdx.y, u = blockIdx.z;
  if (frame >= p.b.n_frames[u]) return;
  const int *cnt = p.ev_count + (u * p.nb + band) * 4;
  const double *ev = p.events + ((size_t)(u * p.nb + band) * 4) * p.ev_cap;
  double cand = 0.0, score = kMaximumValue;
  int n_int[4];
  bool ok = true;
  for (int fam = 0; fam < 4; ++fam) {
    n_int[fam] = cnt[fam] >= 2 ? cnt[fam] - 1 : 0;
    if (n_int[fam] - 2 <= 0) ok = false;                     // dio.cpp:475-484
  }
  if (ok) {
    const double t = frame * p.frame_period / 1000.0;         // dio.cpp:609-610
    double v[4];
    for (int fam = 0; fam < 4; ++fam) v[fam] = interp_intervals(ev + (size_t)fam * p.ev_cap, n_int[fam], p.afs, t);
    double c = (v[0] + v[1] + v[2] + v[3]) / 4.0;
    double s = sqrt(((v[0] - c) * (v[0] - c) + (v[1] - c) * (v[1] - c) + (v[2] - c) * (v[2] - c) +
                     (v[3] - c) * (v[3] - c)) / 3.0);
    const double fb = p.band_f0[band];
    if (c > fb || c < fb / 2.0 || c > p.f0_ceil || c < p.f0_floor) { c = 0.0; s = kMaximumValue; }
    cand = c; score = s;
  }
  const size_t at = ((size_t)u * p.nb + band) * p.b.f_stride + frame;
  p.cand[at] = cand;
  p.score[at] = score / (cand + kTiny);                       // dio.cpp:564-565
}

// ---- GetBestF0Contour + FixStep1 + FixStep2 (dio.cpp:112-169), one thread per frame ----
__device__ __forceinline__ double dio_best_at(const DioParams &p, int u, int f) {
  const size_t row = (size_t)u * p.nb * p.b.f_stride + f;
  double best = p.cand[row], sc = p.score[row];
  for (int b = 1; b < p.nb; ++b) {
    double s = p.score[row + (size_t)b * p.b.f_stride];
    if (sc > s) { sc = s; best = p.cand[row + (size_t)b * p.b.f_stride]; }
  }
  return best;
}
__device__ __forceinline__ double dio_step1_at(const DioParams &p, int u, int f, int nf) {
  const int vrm = p.vrm;
  if (f < vrm) return 0.0;
  auto base = [&](int i) { return (i >= vrm && i < nf - vrm) ? dio_best_at(p, u, i) : 0.0; };
  double b0 = base(f), b1 = base(f - 1);
  return fabs((b0 - b1) / (kTiny + b0)) < p.allowed_range ? b0 : 0.0;
}
__global__ void dio_step1(DioParams p) {
  const int f = flat_thread_x(), u = blockIdx.y;
  const int nf = p.b.n_frames[u];
  if (f >= nf || nf <= p.vrm) return;                         // dio.cpp:266: nothing is written
  p.t1[(size_t)u * p.b.f_stride + f] = dio_step1_at(p, u, f, nf);
}
__global__ void dio_step2(DioParams p) {
  const int f = flat_thread_x(), u = blockIdx.y;
  const int nf = p.b.n_frames[u];
  if (f >= nf || nf <= p.vrm) return;
  const double *t1 = p.t1 + (size_t)u * p.b.f_stride;
  const int center = (p.vrm - 1) / 2;
  double v = t1[f];
  if (f >= center && f < nf - center)
    for (int j = -center; j <= center; ++j)
      if (t1[f + j] == 0) { v = 0.0; break; }
  p.t2[(size_t)u * p.b.f_stride + f] = v;
}

// ---- FixStep3 / FixStep4 (dio.cpp:190-253): the two tracking sweeps ------------------
// One wavefront per utterance; lanes hold the channels' candidates of the target frame.
__device__ __forceinline__ double dio_track(const DioParams &p, int u, double cur, double past, int target) {
  const double ref = (cur * 3.0 - past) / 2.0;
  // nearest candidate, FIRST minimum wins (strict <, dio.cpp:199-205)
  double best = 0.0, emin = 0.0;
  int best_i = 1 << 30;
  for (int b = lane_id(); b < p.nb; b += WAVE) {
    double c = p.cand[((size_t)u * p.nb + b) * p.b.f_stride + target];
    double e = fabs(ref - c);
    if (best_i == (1 << 30) || e < emin) { emin = e; best = c; best_i = b; }
  }
#ifndef WORLD_EMU
  for (int m = 32; m >= 1; m >>= 1) {
    double oe = __shfl_xor(emin, m, 64), ob = __shfl_xor(best, m, 64);
    int oi = __shfl_xor(best_i, m, 64);
    if (oi != (1 << 30) && (best_i == (1 << 30) || oe < emin || (oe == emin && oi < best_i))) {
      emin = oe; best = ob; best_i = oi;
    }
  }
#endif
  if (fabs(1.0 - best / ref) > p.allowed_range) return 0.0;
  return best;
}

__global__ void dio_track_sweeps(DioParams p) {
  const int u = wave_item_x();
  if (u >= p.b.n_utt) return;
  const int nf = p.b.n_frames[u];
  if (nf <= p.vrm) return;
  const int lane = lane_id();
  const double *t2 = p.t2 + (size_t)u * p.b.f_stride;
  double *s3 = p.t1 + (size_t)u * p.b.f_stride;               // step-3 result (t1 is free again)
  double *out = p.f0 + (size_t)u * p.b.f_stride;
  for (int f = lane; f < nf; f += WAVE) s3[f] = t2[f];
  wave_sync();
  // step 3: from every falling edge forwards (GetNumberOfVoicedSections, :174-184)
  {
    int i = 1;
    while (i < nf) {
      // next falling edge at or after i: t2[i] == 0 && t2[i-1] != 0
      while (i < nf && !(t2[i] == 0 && t2[i - 1] != 0)) ++i;
      if (i >= nf) break;
      const int edge = i - 1;
      // limit = the next falling edge's index, or nf - 1
      int nxt = i + 1;
      while (nxt < nf && !(t2[nxt] == 0 && t2[nxt - 1] != 0)) ++nxt;
      const int limit = nxt < nf ? nxt - 1 : nf - 1;
      for (int j = edge; j < limit; ++j) {
        double v = dio_track(p, u, s3[j], s3[j - 1], j + 1);
        if (lane == 0) s3[j + 1] = v;
        wave_sync();
        if (v == 0) break;
      }
      i = nxt;
    }
  }
  wave_sync();
  for (int f = lane; f < nf; f += WAVE) out[f] = s3[f];
  wave_sync();
  // step 4: from every rising edge backwards, edges taken from the STEP-2 contour
  {
    int i = nf - 1;
    while (i >= 1) {
      while (i >= 1 && !(t2[i - 1] == 0 && t2[i] != 0)) --i;
      if (i < 1) break;
      const int edge = i;
      int prv = i - 1;
      while (prv >= 1 && !(t2[prv - 1] == 0 && t2[prv] != 0)) --prv;
      const int limit = prv >= 1 ? prv : 1;
      for (int j = edge; j > limit; --j) {
        double v = dio_track(p, u, out[j], out[j + 1], j - 1);
        if (lane == 0) out[j - 1] = v;
        wave_sync();
        if (v == 0) break;
      }
      i = prv;
    }
  }
}

__global__ void dio_time_axis(DioParams p) {
  const int i = flat_thread_x(), u = blockIdx.y;
  if (i < p.b.n_frames[u]) p.tpos[(size_t)u * p.b.f_stride + i] = i * p.frame_period / 1000.0;
}

void launch_dio(const DioParams &p, int max_x_len, int max_y_len, int max_frames, hipStream_t stream) {
  const int B = p.b.n_utt;
  devrt::dzero(p.y, sizeof(double) * (size_t)B * p.y_stride, stream);
  if (p.ratio == 1) {
    WH_THREADS(dio_copy_signal, max_x_len, B, 1, stream, p);
  } else {
    IirCoef c = decimate_coef(p.ratio);
    const int spans = (max_x_len + 2 * kDecPad + kDecSpan - 1) / kDecSpan;
    WH_BLOCKS(dio_decimate_fwd, dim3(spans, B), kDecThreads, dec_lds_bytes(), stream, p, c);
    WH_BLOCKS(dio_decimate_bwd, dim3(spans, B), kDecThreads, dec_lds_bytes(), stream, p, c);
  }
  const int mean_slices = (max_y_len + kDioMeanSlice - 1) / kDioMeanSlice;
  WH_BLOCKS(dio_partial_sums, dim3(mean_slices, B), 256, 64 * sizeof(double), stream, p);
  WH_BLOCKS(dio_remove_mean, dim3(mean_slices, B), 256, 64 * sizeof(double), stream, p);
  const int lc_tiles = (max_y_len + 2 * p.cut + kTile - 1) / kTile;
  if (band_lds_bytes(2 * p.cut + 1) <= kLdsPerCu)
    devrt::launch_blocks("dio_lowcut", dio_lowcut<false>, dim3(lc_tiles, B), kBpThreads, band_lds_bytes(2 * p.cut + 1), stream, p);
  else
    devrt::launch_blocks("dio_lowcut", dio_lowcut<true>, dim3(lc_tiles, B), kBpThreads,
                         band_lds_bytes(2 * p.cut + 1) - sizeof(double) * (2 * p.cut + 2), stream, p);
  WH_BLOCKS(dio_nyquist_slices, dim3(mean_slices, B), 256, 64 * sizeof(double), stream, p);
  WH_BLOCKS(dio_nyquist_bins, dim3(B), 256, 64 * sizeof(double), stream, p);
  WH_BLOCKS(dio_band_quirk, dim3(p.nb, B), 64, 64 * sizeof(double), stream, p);
  WH_BLOCKS(dio_band_events, dim3(p.nseg, p.nb, B), kBpThreads, band_lds_bytes(p.max_ntap), stream, p);
  WH_BLOCKS(dio_compact_events, dim3(p.nb * 4, B), 256, compact_lds_bytes(p.nseg), stream, p);
  WH_THREADS(dio_candidates, max_frames, p.nb, B, stream, p);
  WH_THREADS(dio_time_axis, max_frames, B, 1, stream, p);
  WH_THREADS(dio_step1, max_frames, B, 1, stream, p);
  WH_THREADS(dio_step2, max_frames, B, 1, stream, p);
  WH_WAVES(dio_track_sweeps, B, 1, 1, 0, stream, p);
}

}  // namespace world_hip
