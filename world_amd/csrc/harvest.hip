// harvest.hip -- Harvest F0 estimation on gfx950, front half: from the waveform
// to refined, pruned F0 candidates at a 1 ms hop.  (The contour logic that turns
// candidates into the final F0 track is in harvest_contour.hip.)
//
// Reference: HarvestGeneralBody (src/harvest.cpp:1145-1215) and everything below
// it.  What changes on MI355X:
//
//  * The 152-band filter bank is NOT 152 x (r2c + c2r) FFTs of 65k..131k points
//    streamed through HBM (harvest.cpp:99-148).  The filters are short symmetric
//    FIRs (<= 2*246+1 taps at 8 kHz), so each (band, utterance) workgroup
//    convolves directly out of LDS -- FP64 FMA bound, the decimated signal is
//    read from L2 once per band -- and the filtered signal never exists in HBM:
//    the four zero-crossing detectors (harvest.cpp:162-238) run on the LDS tile
//    and only the compacted sub-sample crossing times are written out.
//    Power-of-two scaling (the reference's unnormalised inverse FFT) cancels
//    exactly in the crossing-time ratio.
//  * Candidate refinement (46 % of the reference's CPU time: two zero-padded r2c
//    FFTs of 128..2048 points per candidate, harvest.cpp:541-584) only ever reads
//    <= 6 harmonic bins of each spectrum, so it is done as 6-bin DFTs over the
//    un-padded window: one wavefront per frame, lanes = (harmonic, sample phase).
// devrt.h: in this unit wg_thread<NT>() hands out opaque copies of the thread index, so that what a transform stage
// derives from it (LDS addresses, butterfly numbers) is recomputed where it is used.  Hoisted out of
// hv_band_events_fft's loop over blocks those were ~35 registers; the kernel sat at its 168 with four of them in
// scratch, reloaded -- a trip to memory each -- at the top of two phases of every block.  134 registers, no scratch,
// 3 % more vector instructions; the unit's kernels together: 29.5 -> 29.15 ms per 128 utterances.  (d4c.hip keeps the
// plain index: d4c_frame's 13 % more instructions cost more than its one spilled register.)
#ifndef HV_PLAIN_TID
#define WH_FRESH_TID
#endif
#include "bandfilter.h"
#include "decimate.h"
#include "harvest.h"
#include "trace.h"
WH_TRACE_DEFINE(hv)

namespace world_hip {

// ---------------------------------------------------------------------------
// decimation to ~8 kHz (GetWaveformAndSpectrumSub, harvest.cpp:43-66)
__global__ void __launch_bounds__(kDecThreads) hv_decimate_fwd(HarvestParams p, IirCoef c) {
  DYN_LDS(lds);
  const int u = blockIdx.y;
  dec_forward_block(p.b.x + (size_t)u * p.b.x_stride, p.b.x_len[u], p.lag, c, blockIdx.x,
                    p.fwd + (size_t)u * p.m_stride, reinterpret_cast<double *>(lds), dec_warm(p.ratio));
}
__global__ void __launch_bounds__(kDecThreads) hv_decimate_bwd(HarvestParams p, IirCoef c) {
  DYN_LDS(lds);
  const int u = blockIdx.y;
  // ... and the sum of the span's decimated samples (the mean removed next: no pass of its own over y)
  dec_backward_block(p.fwd + (size_t)u * p.m_stride, p.b.x_len[u], p.lag, p.ratio, c, blockIdx.x,
                     p.lag / p.ratio, p.y_len[u], p.y + (size_t)u * p.y_stride, reinterpret_cast<double *>(lds),
                     dec_warm(p.ratio), p.mean_part + (size_t)u * p.mean_parts);
}
__global__ void hv_copy_signal(HarvestParams p) {          // ratio == 1 (harvest.cpp:45-48)
  int u = blockIdx.y, i = flat_thread_x();
  if (i < p.y_len[u]) p.y[(size_t)u * p.y_stride + i] = p.b.x[(size_t)u * p.b.x_stride + i];
}
// y <- y - mean(y)  (harvest.cpp:81-85).  The sum arrives as per-span partials (of the backward decimation sweep; at
// ratio 1, where nothing is decimated, of hv_partial_sums' slices): every workgroup adds them in the same fixed order
// (deterministic; a span beyond the utterance's end holds 0) and subtracts the mean from its slice.
constexpr int kMeanSlice = 4096;
__global__ void hv_partial_sums(HarvestParams p) {          // ratio == 1 only
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int slice = blockIdx.x, u = blockIdx.y, n = p.y_len[u];
  const double *y = p.y + (size_t)u * p.y_stride;
  const int lo = slice * kMeanSlice, hi = imin(n, lo + kMeanSlice);
  double s = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) s += y[i];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) p.mean_part[(size_t)u * p.mean_parts + slice] = s;
}
__global__ void hv_remove_mean(HarvestParams p) {
  DYN_LDS(lds);
  double *scratch = reinterpret_cast<double *>(lds);
  const int slice = blockIdx.x, u = blockIdx.y, n = p.y_len[u];
  double *y = p.y + (size_t)u * p.y_stride;
  if (wave_in_block() == 0) {
    const double *part = p.mean_part + (size_t)u * p.mean_parts;
    double s = 0.0;
    for (int k = lane_id(); k < p.mean_parts; k += WAVE) s += part[k];
    s = wave_sum(s);
    if (lane_id() == 0) scratch[0] = s / n;
  }
  if (threadIdx.x == 0 && slice == 0) p.nc[u] = 0;           // hv_detect's running maximum starts from here
  __syncthreads();
  const double mean = scratch[0];
  const int lo = slice * kMeanSlice, hi = imin(n, lo + kMeanSlice);
  // ... and, on the way, the slice's share of the utterance's two spectrum bins next to Nyquist (the per-band constants
  // of the reference's mirror-store term need them, bandfilter.h: nyquist_pair's sums over the mean-free samples --
  // each thread's own, in its own order; round 4 had a kernel of its own read y back for them)
  const double w = 2.0 / p.ref_fft[u];
  double a = 0.0, br = 0.0, bi = 0.0, sn, cs, sr, cr;
  sincospi((double)(lo + (int)threadIdx.x) * w, &sn, &cs);
  sincospi((double)blockDim.x * w, &sr, &cr);
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const double v = y[i] - mean;
    y[i] = v;
    const double x = (i & 1) ? -v : v;
    a += x; br += x * cs; bi += x * sn;
    const double c2 = cs * cr - sn * sr;
    sn = sn * cr + cs * sr;
    cs = c2;
  }
  block_sum3(a, br, bi, scratch);                    // (its leading barrier: everybody has read scratch[0])
  if (threadIdx.x == 0) { double *o = p.nyq + ((size_t)u * gridDim.x + slice) * 4; o[0] = a; o[1] = br; o[2] = bi; o[3] = w; }
}
// the per-band constants of the reference's mirror-store term (bandfilter.h) from the utterance's two spectrum bins next
// to Nyquist and the band's: one workgroup per (band, utterance) -- on the overlap-save route the tail of hv_block_spectra's
// grid, on the direct-form route a launch of its own
__device__ __forceinline__ void band_quirk_constants(const HarvestParams &p, int band, int u, double *scratch) {
  const double *part = p.nyq + (size_t)u * p.nyq_slices * 4;
  double y0 = 0.0, y1r = 0.0, y1i = 0.0;
  for (int k = 0; k < p.nyq_slices; ++k) { y0 += part[4 * k]; y1r += part[4 * k + 1]; y1i += part[4 * k + 2]; }
  const double w = part[3];
  double h0, h1r, h1i;
  nyquist_pair(p.band_taps + p.band_off[band], 2 * p.band_half[band] + 1, w, scratch, &h0, &h1r, &h1i);
  if (threadIdx.x == 0)
    mirror_store_constants(y0, y1r, y1i, h0, h1r, h1i, w, p.quirk + ((size_t)u * p.nch + band) * 4);
}
__global__ void hv_band_quirk(HarvestParams p) {
  DYN_LDS(lds);
  band_quirk_constants(p, blockIdx.x, blockIdx.y, reinterpret_cast<double *>(lds));
}

// ---------------------------------------------------------------------------
// Band-pass FIR + the four zero-crossing families (bandfilter.h).  One workgroup
// per (time segment, band, utterance).
__global__ void __launch_bounds__(kBpThreads) hv_band_events(HarvestParams p) {
  const int seg = blockIdx.x, band = blockIdx.y, ue = blockIdx.z, u = ue + p.ev_u0;   // (ue: the utterance's slot in the event lists)
  BandJob job;
  job.in = p.y + (size_t)u * p.y_stride;
  job.in_len = p.y_len[u];
  job.n = p.y_len[u];
  job.taps = p.band_taps + p.band_off[band];
  job.ntap = 2 * p.band_half[band] + 1;
  job.shift = p.band_half[band] + 1;             // delay compensation L+1 (harvest.cpp:140-142)
  job.max_ntap = 2 * p.max_half + 1;
  job.nseg = p.nseg;
  job.seg_events = p.seg_events + ((size_t)(ue * p.nch + band) * 4) * p.nseg * kSegCap;   // seg_cap == kSegCap on this path
  job.seg_count = p.seg_count + ((size_t)(ue * p.nch + band) * 4) * p.nseg;
  job.quirk = p.quirk + ((size_t)u * p.nch + band) * 4;
  job.quirk_delay = job.shift;                     // the term is a function of the undelayed index
  band_events_segment(job, seg);
}

// ---------------------------------------------------------------------------
// The same filter bank by overlap-save (fast convolution) when the filters fit: the reference itself
// filters through FFTs (harvest.cpp:99-148); the direct form above costs 2 (2L+1) flop per output, on
// average ~350 for the 152 channels of the default range (L = 18 .. 246), a 4096-point block transform
// ~40.  The signal is cut into blocks of kBandFft samples that start fft_pre samples before their
// fft_seg outputs; a block's spectrum is formed ONCE (hv_block_spectra) and serves all channels; a
// channel's spectrum (taps zero-padded, scaled by 1 / kBandFft) is formed once per band set
// (hv_band_spectra, cached with the taps).  hv_band_events_fft = product of the two, one c2r transform
// in LDS, and the zero-crossing detectors on the result: output i = t0 + k sits at sample
// k + fft_pre + shift of the block, clear of the circular wrap for every channel.
__global__ void __launch_bounds__(256) hv_band_spectra(const double *taps, const int *off, const int *half, double2 *spec,
                                                       const double2 *tw_global) {
  DYN_LDS(lds);
  const int band = blockIdx.x, ntap = 2 * half[band] + 1;
  cplx *Z = reinterpret_cast<cplx *>(lds);
  const TwLds tw = stage_twiddles(reinterpret_cast<double *>(lds) + kBandFft, kBandFftLg - 1, tw_global);
  const double *h = taps + off[band];
  for (int i = threadIdx.x; i < kBandFft; i += blockDim.x) rfft_in(Z, i) = i < ntap ? h[i] : 0.0;
  double2 *out = spec + (size_t)band * kBandFftBins;
  block_rfft<3>(Z, kBandFftLg, tw, [&](int k, double re, double im) {
    out[k] = make_double2(re * (1.0 / kBandFft), im * (1.0 / kBandFft));       // exact: a power of two
  });
}
void launch_band_spectra(const double *d_taps, const int *d_off, const int *d_half, int nch, double2 *d_spec,
                         const Tables &tab, hipStream_t stream) {
  WH_BLOCKS(hv_band_spectra, dim3(nch), 256, sizeof(double) * (kBandFft + twiddle_lds_doubles(kBandFftLg - 1)), stream, d_taps,
            d_off, d_half, d_spec, tab.tw);
}
int hv_fft_segment(int max_half) {
  const int seg = (kBandFft - 2 * max_half - 2) & ~7;
  return seg >= kBandFft / 2 ? seg : 0;          // below half a block per transform the direct form is kept
}

__global__ void __launch_bounds__(256) hv_block_spectra(HarvestParams p) {
  DYN_LDS(lds);
  const int blk = blockIdx.x, u = blockIdx.y, n = p.y_len[u];
  if (blk >= p.nblk) {                                        // the grid's tail: one workgroup per band (see above)
    band_quirk_constants(p, blk - p.nblk, u, reinterpret_cast<double *>(lds));
    return;
  }
  const int m0 = blk * p.fft_seg - p.fft_pre;                 // signal index of the block's first sample
  if (blk * p.fft_seg >= n) return;                           // no output of this utterance lies in the block
  cplx *Z = reinterpret_cast<cplx *>(lds);
  const TwLds tw = stage_twiddles(reinterpret_cast<double *>(lds) + kBandFft, kBandFftLg - 1, p.tab.tw);
  const double *y = p.y + (size_t)u * p.y_stride;
  for (int i = threadIdx.x; i < kBandFft; i += blockDim.x) {
    const int idx = m0 + i;
    rfft_in(Z, i) = (idx >= 0 && idx < n) ? y[idx] : 0.0;
  }
  double2 *out = p.blk_spec + ((size_t)u * p.nblk + blk) * kBandFftBins;
#ifdef WORLD_EMU
  block_rfft<3>(Z, kBandFftLg, tw, [&](int k, double re, double im) { out[k] = make_double2(re, im); });
#else
  {
    // bins in natural order per wavefront: the spectrum goes to global memory (see irfft_pretwiddle_items)
    constexpr FftPlan plan = make_plan_max(kBandFftLg - 1, 3);
    block_cfft_dif_static<kBandFftLg - 1, 3, 256>(Z, tw);
    rfft_merge_items<(kBandFft / 4 + 1 + 255) / 256, 256>(Z, kBandFftLg, plan, tw,
      [&](int, int k, double ar, double ai, bool paired, double br, double bi) {
        out[k] = make_double2(ar, ai);
        if (paired) out[kBandFft / 2 - k] = make_double2(br, bi);
      });
  }
#endif
}

// One workgroup = one band x one run of `chunk_blocks` consecutive blocks x one utterance.  The band's spectrum H waits in
// registers (natural-order items: 40 VGPRs) while the blocks' spectra stream past it -- consecutive workgroups are the
// bands of one utterance, so a block's spectrum is fetched from HBM once and served to its 152 readers by L2 (the
// previous launch order re-read X and H per (block, band): 8.7 x the whole pipeline's algorithmic traffic) -- the
// twiddle table, the merge twiddles and the mirror-store constants are set up once, and the crossings of block after
// block are APPENDED to the chunk's lists: with one chunk per utterance (batches) those are the final lists and
// hv_compact_events does not run at all.
#ifndef HV_FFT_MIN_WAVES
#define HV_FFT_MIN_WAVES 4      // 128 registers (two in scratch) and 35 KB of LDS: four workgroups per CU -- 3.79 -> 3.47 ms per 128 utterances against three
#endif
__global__ void __launch_bounds__(256, HV_FFT_MIN_WAVES) hv_band_events_fft(HarvestParams p) {
  DYN_LDS(lds);
  const int band = blockIdx.x, chunk = blockIdx.y, ue = blockIdx.z, u = ue + p.ev_u0, tid = threadIdx.x, nt = blockDim.x;
  const int n = p.y_len[u];
  const size_t list = ((size_t)(ue * p.nch + band) * 4);             // (ue: the utterance's slot in the event lists)
  int *cnt_out = p.seg_count + list * p.nseg + chunk;
  const int blk0 = chunk * p.chunk_blocks;
  if (blk0 * p.fft_seg >= n) {
    if (tid == 0) for (int fam = 0; fam < 4; ++fam) cnt_out[(size_t)fam * p.nseg] = 0;
    return;
  }
  const int blk1 = imin(blk0 + p.chunk_blocks, (n + p.fft_seg - 1) / p.fft_seg);
  const bool trace_me = blockIdx.x == 20 && blockIdx.y == 0 && u == WH_TRACE_UTT; (void)trace_me;
  WH_STAMP(24, 0);
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *scratch = reinterpret_cast<double *>(lds) + kBandFft;
  const TwLds tw = stage_twiddles(scratch + 64, kBandFftLg - 1, p.tab.tw);
  WH_STAMP(24, 1);
  const double2 *H = p.band_spec + (size_t)band * kBandFftBins;
  // filtered[t0 + k] = block sample k + fft_pre + shift, shift = L + 1 (delay compensation, harvest.cpp:140-142)
  const int shift = p.band_half[band] + 1;
  const int at0 = p.fft_pre + shift;
  // the reference's mirror-store term (bandfilter.h), a function of the undelayed index t0 + k + shift:
  //   (-1)^n (qc cos(pi w n) + qs sin(pi w n) + q0),  n = t0 + (s - at0) + shift for block sample s.
  const double *qc4 = p.quirk + ((size_t)u * p.nch + band) * 4;
  const double qc = qc4[0], qs = qc4[1], q0 = qc4[2], w = qc4[3];
  double *ev = p.seg_events + (list * p.nseg + chunk) * p.seg_cap;
  const size_t fam_stride = (size_t)p.nseg * p.seg_cap;
  int count[4] = {0, 0, 0, 0};
#ifdef WORLD_EMU
  for (int blk = blk0; blk < blk1; ++blk) {
    const int t0 = blk * p.fft_seg;
    const double2 *X = p.blk_spec + ((size_t)u * p.nblk + blk) * kBandFftBins;
    auto product = [&](int k) { const double2 x = X[k], h = H[k]; cplx a, b; a.re = x.x; a.im = x.y; b.re = h.x; b.im = h.y; return cmul(a, b); };
    block_irfft<3>(Z, kBandFftLg, tw, product);
    const int len = imin(p.fft_seg, n - t0);
    for (int k = tid; k < len + 2; k += nt) {
      const int ng = t0 + k + shift;
      double sn, cs;
      sincospi(ng * w, &sn, &cs);
      rfft_in(Z, k + at0) += ((ng & 1) ? -1.0 : 1.0) * (qc * cs + qs * sn + q0);
    }
    tile_events<16, false>([&](int k) { return rfft_in(Z, k + at0); }, t0, len, n, ev, fam_stride, count, p.seg_cap, scratch);
  }
#else
  // cos / sin of the term's phase at the thread's first sample of the first block; from block to block and from sample
  // to sample they advance by rotation (a sincospi per block sat in the loop with the filter spectrum's 40 registers live)
  double sr, cr, sb, cb, sn0, cs0;
  sincospi(nt * w, &sr, &cr);
  sincospi(p.fft_seg * w, &sb, &cb);
  sincospi((blk0 * p.fft_seg + tid + shift) * w, &sn0, &cs0);
  constexpr int NT = 256, kItems = (kBandFft / 4 + 1 + NT - 1) / NT;   // 256 threads: launch_harvest
  constexpr int hh = kBandFft / 2, qq = hh / 2;
  constexpr FftPlan plan = make_plan_max(kBandFftLg - 1, 3);
  cplx hk[kItems], hm[kItems];                           // the band's spectrum: natural-order items, 40 VGPRs
#pragma unroll
  for (int m = 0; m < kItems; ++m) {
    const int k = tid + m * NT;
    hk[m].re = hk[m].im = hm[m].re = hm[m].im = 0.0;
    if (k <= qq) {
      const double2 a = H[k], b = H[hh - k];
      hk[m].re = a.x; hk[m].im = a.y; hm[m].re = b.x; hm[m].im = b.y;
    }
  }
  const cplx wbf = twiddle(tw, tid, kBandFftLg, -1);     // e^{-2 pi i tid / N}; item m's inverse twiddle is conj(wbf W16^m)
  WH_STAMP(24, 2);
  for (int blk = blk0; blk < blk1; ++blk) {
    const int t0 = blk * p.fft_seg;
    const double2 *X = p.blk_spec + ((size_t)u * p.nblk + blk) * kBandFftBins;
    irfft_pretwiddle_items_w<kItems, NT>(Z, kBandFftLg, plan, [&](int m, int k, cplx &x, cplx &y, cplx &wk) {
      const double2 a = X[k], b = X[hh - k];
      cplx ca, cb2; ca.re = a.x; ca.im = a.y; cb2.re = b.x; cb2.im = b.y;
      x = cmul(ca, hk[m]); y = cmul(cb2, hm[m]);
      wk = cconj(mul_w16_fwd(wbf, m));
    });
    if (blk == blk0) WH_STAMP(24, 3);
    block_cfft_dit_static<kBandFftLg - 1, 3, NT>(Z, tw);
    if (blk == blk0) WH_STAMP(24, 4);
    const int len = imin(p.fft_seg, n - t0);
    {
      const int n0 = t0 + tid + shift;
      double sn = sn0, cs = cs0;
      { const double c2 = cs0 * cb - sn0 * sb; sn0 = sn0 * cb + cs0 * sb; cs0 = c2; }     // the next block's phase
      const double sign = (n0 & 1) ? -1.0 : 1.0;
      // (Round 4 fetched the thread's samples eight at a time before updating any -- written `Z[k] += term` the loop is a
      // read, a wait and a write per sample -- and measured it slower, 3.80 -> 3.97 ms per 128 utterances: the kernel sits
      // at its 168 registers, the batch spilled 25 more SGPRs, and what the loop waits for is not the LDS read but the
      // reload of two spilled registers from scratch.)
      // (Round 5 tried leaving alone the samples the term cannot matter for -- it is a Nyquist-rate ripple ten orders below
      // the signal and decides a crossing only in digital silence -- and advancing the phase recurrence on demand: slower,
      // 3.53 -> 3.80 ms per 128 utterances.  A per-sample magnitude test qualifies every sample near a zero crossing, the
      // local-amplitude test that would be right costs what the term does, and the loop is bound by its chain of LDS reads
      // either way.)
      for (int k = tid, j = 0; k < len + 2; k += nt, ++j) {
        const double flip = ((nt & 1) && (j & 1)) ? -sign : sign;     // (-1)^(n0 + j nt)
        rfft_in(Z, k + at0) += flip * (qc * cs + qs * sn + q0);
        const double c2 = cs * cr - sn * sr;
        sn = sn * cr + cs * sr;
        cs = c2;
      }
      __syncthreads();
    }
    if (blk == blk0) WH_STAMP(24, 5);
    tile_events<16, false>([&](int k) { return rfft_in(Z, k + at0); }, t0, len, n, ev, fam_stride, count, p.seg_cap, scratch,
                           trace_me && blk == blk0);
    if (blk == blk0) WH_STAMP(24, 6);
  }
#endif
  if (tid == 0)
    for (int fam = 0; fam < 4; ++fam) cnt_out[(size_t)fam * p.nseg] = imin(count[fam], p.seg_cap);
}

// concatenate the per-segment lists of one (family, band, utterance) in time order
__global__ void hv_compact_events(HarvestParams p) {
  DYN_LDS(lds);
  const int bf = blockIdx.x, ue = blockIdx.y;       // bf = band * 4 + family; ue: the utterance's slot in the event lists
  const size_t list = (size_t)ue * p.nch * 4 + bf;
  compact_event_segments(p.seg_events + list * p.nseg * p.seg_cap, p.seg_count + list * p.nseg, p.nseg, p.seg_cap,
                         p.events + list * p.ev_cap, p.ev_cap, p.ev_count + list, lds);
}

// ---------------------------------------------------------------------------
// interp1 of the interval F0s onto the 1 ms grid + gating (harvest.cpp:240-293);
// one workgroup per run of kRawFrames frames of one (band, utt), one thread per frame.
constexpr int kRawFrames = 256;                       // threads of a workgroup
#ifndef HV_RAW_RUN
#define HV_RAW_RUN 256
#endif
constexpr int kRawRun = HV_RAW_RUN;                    // frames of a workgroup's run (the eight end-point searches are per run)
__device__ __forceinline__ double hv_gate(const HarvestParams &p, int band, double v0, double v1, double v2, double v3) {
  double c = (v0 + v1 + v2 + v3) / 4.0;
  const double fb = p.band_f0[band];
  if (c > fb * 1.1 || c < fb * 0.9 || c > p.f0_ceil || c < p.f0_floor) c = 0.0;
  return c;
}
#ifndef HV_RAW_MIN_WG
#define HV_RAW_MIN_WG 8                                // workgroups per CU the register allocation leaves room for (bandfilter.h: kIntervalCap)
#endif
__global__ void __launch_bounds__(kRawFrames, HV_RAW_MIN_WG) hv_raw_candidates(HarvestParams p) {
  DYN_LDS(lds);
  // Bands are the fastest grid dimension, padded to a multiple of the eight XCDs: workgroups go to the XCDs round robin,
  // so every run of one band's lists is served by the same L2 and a list crosses the fabric once.  (Runs fastest, the
  // 20 runs of a list were spread over all eight L2s, each fetching the list's head and tail for the proportional
  // guess and its neighbours' boundary intervals: 3.4 x the lists' size in fetches.)
  const int band = blockIdx.x, ue = blockIdx.z, u = ue + p.ev_u0, tid = threadIdx.x, nt = blockDim.x;
  if (band >= p.nch) return;
  const int f_begin = blockIdx.y * kRawRun, f_end = imin(f_begin + kRawRun, p.nfb[u]);
  if (f_begin >= f_end) return;
  const int *cnt = p.ev_count + (ue * p.nch + band) * 4;
  const double *ev = p.events + ((size_t)(ue * p.nch + band) * 4) * p.ev_cap;
  double *out = p.raw + ((size_t)u * p.nch + band) * p.fb_stride;
  int n_int[4];
  bool ok = true;
  for (int fam = 0; fam < 4; ++fam) {
    n_int[fam] = cnt[fam] >= 2 ? cnt[fam] - 1 : 0;
    if (n_int[fam] - 2 <= 0) ok = false;                 // CheckEvent(n - 2), harvest.cpp:263-269
  }
  if (!ok) {
    for (int f = f_begin + tid; f < f_end; f += nt) out[f] = 0.0;
    return;
  }
  double *loc = reinterpret_cast<double *>(lds);          // [4][kIntervalCap]
  double *fz = loc + 4 * kIntervalCap;                    // [4][kIntervalCap]
  IntervalRange *range = reinterpret_cast<IntervalRange *>(fz + 4 * kIntervalCap);   // [4]
  // frame f sits at t = f * 1 / 1000.0 (harvest.cpp:1175, frame_period = 1)
#ifdef WORLD_EMU
  for (int job = tid; job < 8; job += nt) {
    const int fam = job >> 1;
    const bool last = job & 1;
    interval_range_ends(ev + (size_t)fam * p.ev_cap, n_int[fam], p.afs, (last ? f_end - 1 : f_begin) * 1 / 1000.0, last,
                        range + fam);
  }
#else
  {
    // the eight end-point counts, one per group of 32 lanes (kRawFrames = 8 groups), each found by its lanes together
    static_assert(kRawFrames == 8 * 32, "one 32-lane group per (family, end)");
    const int job = tid >> 5, fam = job >> 1, sub = tid & 31;
    const bool last = job & 1;
    const int c = intervals_at_or_before_group(ev + (size_t)fam * p.ev_cap, n_int[fam], p.afs,
                                               (last ? f_end - 1 : f_begin) * 1 / 1000.0, (tid >> 5) & 1, sub);
    if (sub == 0) { if (last) range[fam].c_last = c; else range[fam].c_first = c; }
  }
#endif
  __syncthreads();
  for (int fam = tid; fam < 4; fam += nt) interval_range_close(n_int[fam], range + fam);
  __syncthreads();
  const bool staged = range[0].m <= kIntervalCap && range[1].m <= kIntervalCap && range[2].m <= kIntervalCap &&
                      range[3].m <= kIntervalCap;
  if (!staged) {                                          // cannot happen at 1 ms frames; kept exact all the same
    for (int f = f_begin + tid; f < f_end; f += nt) {
      const double t = f * 1 / 1000.0;
      out[f] = hv_gate(p, band, interp_intervals(ev, n_int[0], p.afs, t), interp_intervals(ev + p.ev_cap, n_int[1], p.afs, t),
                       interp_intervals(ev + 2 * (size_t)p.ev_cap, n_int[2], p.afs, t),
                       interp_intervals(ev + 3 * (size_t)p.ev_cap, n_int[3], p.afs, t));
    }
    return;
  }
  {
    // the four families' intervals as ONE list of jobs: their loads are in flight together (four loops, one per
    // family, were four dependent trips to global memory)
    const int m0 = range[0].m, m1 = m0 + range[1].m, m2 = m1 + range[2].m, m3 = m2 + range[3].m;
    for (int job = tid; job < m3; job += nt) {
      const int fam = (job >= m0) + (job >= m1) + (job >= m2);
      const int i = job - (fam == 0 ? 0 : fam == 1 ? m0 : fam == 2 ? m1 : m2);
      const double *e = ev + (size_t)fam * p.ev_cap + range[fam].j_lo;
      loc[fam * kIntervalCap + i] = interval_loc(e, i, p.afs);
      fz[fam * kIntervalCap + i] = interval_f0(e, i, p.afs);
    }
  }
  __syncthreads();
  for (int f = f_begin + tid; f < f_end; f += nt) {
    const double t = f * 1 / 1000.0;
    double v[4];
#pragma unroll
    for (int fam = 0; fam < 4; ++fam)
      v[fam] = interp_staged(range[fam], loc + fam * kIntervalCap, fz + fam * kIntervalCap, n_int[fam], t);
    out[f] = hv_gate(p, band, v[0], v[1], v[2], v[3]);
  }
}

// ---------------------------------------------------------------------------
// DetectOfficialF0Candidates (harvest.cpp:348-412): runs of >= 10 voiced bands.
__global__ void hv_detect(HarvestParams p) {
  const int frame = flat_thread_x(), u = blockIdx.y;
  if (frame >= p.nfb[u]) return;
  const double *raw = p.raw + (size_t)u * p.nch * p.fb_stride + frame;
  double *out = p.cand_a + ((size_t)u * p.fb_stride + frame) * p.maxc;
  int cnt = 0, st = 0, prev = 0;
  double run_sum = 0.0;                 // sum of the current voiced run, in band order (:374-375)
  // Bands are fetched kBatch at a time, and the next batch is requested before the current one is walked: a batch
  // takes ~2 us to arrive from HBM, walking it a fraction of that -- eight exposed trips were most of the kernel.
  constexpr int kBatch = 19;            // 19 loads in flight per thread and buffer (152 bands = 8 batches)
  auto fetch = [&](double (&v)[kBatch], int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < kBatch; ++q) v[q] = raw[(size_t)imin(j0 + q, p.nch - 1) * p.fb_stride];
  };
  auto walk = [&](const double (&v)[kBatch], int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const int j = j0 + q;
      if (j >= p.nch || j == 0) continue;                     // vuv[0] is forced to 0 (:400)
      const int cur = (j == p.nch - 1) ? 0 : (v[q] > 0 ? 1 : 0);
      if (cur - prev == 1) { st = j; run_sum = 0.0; }
      if (cur - prev == -1 && j - st >= 10 && cnt < p.maxc) out[cnt++] = run_sum / (j - st);
      if (cur) run_sum += v[q];
      prev = cur;
    }
  };
  double va[kBatch], vb[kBatch];
  fetch(va, 0);
  for (int j0 = 0; j0 < p.nch; j0 += 2 * kBatch) {
    fetch(vb, j0 + kBatch);              // (past the last band the clamped addresses repeat it: unused)
    walk(va, j0);
    if (j0 + kBatch >= p.nch) break;
    fetch(va, j0 + 2 * kBatch);
    walk(vb, j0 + kBatch);
  }
  // Empty slots up to the most a frame can hold: voiced runs are >= 10 bands long and a band apart, bands 0 and nch - 1
  // never voiced -- at most (nch - 1) / 11 of them.  (Nobody reads a slot beyond the utterance's largest count, p.nc; the
  // rows are maxc = 7 x that wide for hv_refine's overlapped set, and zeroing all of it was 7/8 of this kernel's stores.)
  const int zcap = imin(p.maxc, (p.nch + 10) / 11 + 1);
  for (int j = cnt; j < zcap; ++j) out[j] = 0.0;
  if (cnt > 0) atomicMax(p.nc + u, cnt);
}

// ---------------------------------------------------------------------------
// OverlapF0Candidates + RefineF0Candidates (harvest.cpp:417-631), one wavefront
// per (frame, utt).  Slot s = j + nc*m takes candidate j of frame-m (m = 1..3) or
// frame+(m-3) (m = 4..6); each non-zero slot is refined by instantaneous frequency.
__device__ __forceinline__ int floor_log2_int(int v) { return 31 - __builtin_clz((unsigned)v); }   // v >= 1

__global__ void hv_refine(HarvestParams p) {
  DYN_LDS(lds);
  const int frame = wave_item_x(), u = blockIdx.y;
  if (frame >= p.nfb[u]) return;
  const int lane = lane_id();
  const int cap = p.refine_cap;
  double *yc = reinterpret_cast<double *>(lds) + (size_t)wave_in_block() * 3 * cap;
  cplx *yw = reinterpret_cast<cplx *>(yc + cap);    // signal around the frame | (y*main window, y*diff window) pairs
  // (Round 4 tried handing the tail its per-slot sums through 2.7 KB of LDS per wavefront instead of a dozen selects per
  // candidate: the extra LDS took the kernel from four workgroups per CU to three -- 6.7 -> 7.4 ms per 128 utterances.)
  const int nfb = p.nfb[u], nc = p.nc[u];
  const double *src = p.cand_a + (size_t)u * p.fb_stride * p.maxc;
  double *dst_f0 = p.cand_b + ((size_t)u * p.fb_stride + frame) * p.maxc;
  double *dst_sc = p.score_b + ((size_t)u * p.fb_stride + frame) * p.maxc;
  const double *y = p.y + (size_t)u * p.y_stride;
  const int y_len = p.y_len[u];
  const double fs = p.afs, inv_fs = 1.0 / fs;
  const double pos = frame * 1 / 1000.0;

  // Every window of this frame is centred on `pos`, so the samples any of them can touch
  // (clamped at the signal ends like GetBaseIndex's safe_index, harvest.cpp:434-441) are
  // fetched from HBM once and kept in LDS.
  const bool trace_me = frame == 5 * WH_TRACE_FRAME && u == WH_TRACE_UTT; (void)trace_me;   // (the 1 ms grid)
  WH_ACC_DECL;
  WH_ACC_BEGIN;
  const int origin = mround(pos * fs) - cap / 2;
  {
    constexpr int kB = 6;                                   // loads in flight per lane (a trip to L2 per iteration otherwise)
    for (int k0 = lane; k0 < cap; k0 += kB * WAVE) {
      double v[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) v[q] = y[imax(0, imin(y_len - 1, origin + k0 + q * WAVE))];
#pragma unroll
      for (int q = 0; q < kB; ++q) if (k0 + q * WAVE < cap) yc[k0 + q * WAVE] = v[q];
    }
  }
  wave_sync();
  WH_ACC_END(0);

  // Lane roles: harmonic h = lane / G, sample phase g = lane % G -- the phases of a harmonic sit in eight
  // neighbouring lanes, so the per-candidate reduction over phases is three DPP steps (oct_sum: no LDS), and
  // the reduction over harmonics, which crosses rows of 16 lanes, is paid once per track in the deferred tail.  Slots are visited
  // track by track (j outer, the 7 neighbouring source frames m inner): consecutive
  // candidates then differ by a fraction of a Hz, so they usually share the window length
  // -- and with it the windowed samples and often the harmonic bin indices.  Whatever is
  // identical to the previous candidate is reused.  The per-candidate tail (FixF0's
  // instantaneous-frequency arithmetic) is deferred: phase group g keeps the DFT sums of
  // slot m == g, and one vectorised tail per track finishes all seven slots at once.
  constexpr int LH = WAVE >= 8 ? 8 : 1;                           // harmonics handled side by side
  constexpr int G = WAVE / LH;                                    // sample phases
  constexpr int kIter = (6 + LH - 1) / LH;                        // harmonic groups per lane (1 on the GPU)
  constexpr int kM = (7 + G - 1) / G;                             // deferred slots per lane (1 on the GPU)
  const int hl = lane / G, g = lane % G;
  int c_hw = -1, c_first = 0;                                      // window held in LDS
  int c_idx[kIter];
  double c_are[kIter], c_aim[kIter], c_dre[kIter], c_dim[kIter];   // reduced DFT sums of the previous candidate
  for (int q = 0; q < kIter; ++q) c_idx[q] = -1;

  // What a slot's candidate fixes before any sample is touched (GetRefinedF0, harvest.cpp:589-617)
  struct Cand { double f0c, bins; int hw, first, lgN, nh; };
  auto cand_of = [&](int j, int m) {
    Cand c;
    const int sf = m == 0 ? frame : (m <= 3 ? frame - m : frame + (m - 3));
    c.f0c = (m < 7 && sf >= 0 && sf < nfb) ? src[(size_t)sf * p.maxc + j] : 0.0;
    const double f = c.f0c > 0.0 ? c.f0c : 1.0;                   // the values of an empty slot are never used
    c.hw = static_cast<int>(1.5 * fs / f + 1.0);
    c.lgN = 2 + floor_log2_int(2 * c.hw + 1);                     // fft_size = 2^(2+floor(log2(2hw+1)))
    const double base0 = static_cast<double>(-c.hw) * inv_fs;    // first = round(..) + 0.001 margin absorbs the ulp
    c.first = mround((pos + base0) * fs + 0.001);                 // GetBaseIndex, harvest.cpp:434-441
    c.nh = imin(static_cast<int>(fs / 2.0 / f), 6);
    c.bins = f * (1 << c.lgN) / fs;                               // bin of harmonic h: round(bins * (h + 1)), FixF0 :515
    return c;
  };

  for (int j = 0; j < nc; ++j) {
    double k_are[kM][kIter], k_aim[kM][kIter], k_dre[kM][kIter], k_dim[kM][kIter], k_f0[kM];
    int k_idx[kM][kIter], k_lgn[kM];
    for (int q = 0; q < kM; ++q) k_f0[q] = 0.0;
#ifndef WORLD_EMU
    // The seven slots of the track are set up side by side, slot m on lane m -- one load of the seven source
    // frames, the divisions and roundings once per track -- and handed to the wavefront as scalars (v_readlane):
    // done per slot by all 64 lanes they were a sixth of the kernel's instructions and seven dependent loads.
    const Cand mine = cand_of(j, lane < 7 ? lane : 7);
#endif
    // The slots are visited in TIME order -- frames f-3, f-2, f-1, f, f+1, f+2, f+3 = slots 3, 2, 1, 0, 4, 5, 6 -- so that
    // F0, and with it the window length, moves monotonically through the track: every distinct window is built once
    // (in slot order the walk came back from f-3 to f+1 and rebuilt windows it had already had: ~1.5 of a track's ~4.5
    // rebuilds).  What a slot computes does not depend on the order.
    for (int mi = 0; mi < 7; ++mi) {
      const int m = mi < 4 ? 3 - mi : mi;
#ifndef WORLD_EMU
      const double f0c = readlane_f64(mine.f0c, m);
      if (!(f0c > 0.0)) continue;
      const double bins = readlane_f64(mine.bins, m);
      const int hw = __builtin_amdgcn_readlane(mine.hw, m), lgN = __builtin_amdgcn_readlane(mine.lgN, m);
      const int first = __builtin_amdgcn_readlane(mine.first, m), nh = __builtin_amdgcn_readlane(mine.nh, m);
#else
      const Cand cm = cand_of(j, m);
      const double f0c = cm.f0c;
      if (!(f0c > 0.0)) continue;
      const double bins = cm.bins;
      const int hw = cm.hw, lgN = cm.lgN, first = cm.first, nh = cm.nh;
#endif
      const int blen = 2 * hw + 1;
      const int N = 1 << lgN;
      const bool same_window = hw == c_hw && first == c_first;
#ifdef HV_REFINE_ROTATE                                  // (A/B, tools/ab.py: round 4's route at every rate)
      const bool table_window = false;
#else
      const bool table_window = p.win_full != nullptr;
#endif
      if (!same_window && table_window) {
        // whole-sample frame centres (8 kHz): the window is a row of the host's table -- a load and two products per sample
        WH_ACC_BEGIN;
        const double2 *wf = p.win_full + (size_t)hw * hw;
        wave_sync();                                             // the previous candidate's reads are done
        constexpr int kB = 3;                                    // loads in flight per lane
        for (int i0 = lane; i0 < blen; i0 += kB * WAVE) {
          double2 t[kB];
#pragma unroll
          for (int q = 0; q < kB; ++q) t[q] = wf[imin(blen - 1, i0 + q * WAVE)];
#pragma unroll
          for (int q = 0; q < kB; ++q) {
            const int i = i0 + q * WAVE;
            if (i < blen) {
              const double xv = yc[imax(0, imin(cap - 1, first + i - 1 - origin))];
              cplx pr; pr.re = xv * t[q].x; pr.im = xv * t[q].y;
              yw[i] = pr;
            }
          }
        }
        wave_sync();
        c_hw = hw; c_first = first;
        WH_ACC_END(1);
      } else if (!same_window) {
        WH_ACC_BEGIN;
        // Blackman main window (harvest.cpp:446-456) and its central difference
        // (GetDiffWindow, :462-468).  w[i+1]-w[i-1] follows from the angle-addition
        // identities, so no neighbour values are exchanged.
        // sin/cos of one sample's and of WAVE samples' angle step, 2 / window length, pi d: host table indexed by hw
        const double *wt = p.win_tab + 6 * hw;
        const double sd = wt[0], cd = wt[1], sD = wt[2], cD = wt[3];
        const double s2d = 2.0 * sd * cd, c2d = 2.0 * cd * cd - 1.0;
        wave_sync();                                             // the previous candidate's reads are done
        // A lane's samples are WAVE apart: sin / cos of its first sample's angle, then a rotation by WAVE * delta per
        // further sample.  The first angle is pi ((first + lane - 1) / fs - pos) (2 / window length)
        //   = pi (lane - hw - 1) d  +  pi r d,   r = first + hw - pos fs   (|r| <= 0.501: GetBaseIndex's rounding),
        // the first term from the host's table (exact to half an ulp), the second a rotation by an angle below 0.12 rad
        // whose sine and cosine are short series (r vanishes where a millisecond is a whole number of samples: 8 kHz).
        // (round 3: a division and a sincospi per rebuild, 70 of its ~190 instructions)
        double sa, ca;
        {
          const double2 t0 = reinterpret_cast<const double2 *>(p.win_lane)[(size_t)hw * WAVE + lane];
          const double rho = ((first + hw) - pos * fs) * wt[5];
          const double x2 = rho * rho;
          const double sr = rho * fma(x2, fma(x2, fma(x2, fma(x2, 1.0 / 362880.0, -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
          const double cr = fma(x2, fma(x2, fma(x2, fma(x2, fma(x2, -1.0 / 3628800.0, 1.0 / 40320.0), -1.0 / 720.0), 1.0 / 24.0), -0.5), 1.0);
          sa = fma(t0.x, cr, t0.y * sr);
          ca = fma(t0.y, cr, -(t0.x * sr));
        }
        for (int i = lane; i < blen; i += WAVE) {
          const double c2a = 2.0 * ca * ca - 1.0, s2a = 2.0 * sa * ca;
          const double w = 0.42 + 0.5 * ca + 0.08 * c2a;
          double dwv;
          if (i == 0) dwv = -(0.42 + 0.5 * (ca * cd - sa * sd) + 0.08 * (c2a * c2d - s2a * s2d)) / 2.0;
          else if (i == blen - 1) dwv = (0.42 + 0.5 * (ca * cd + sa * sd) + 0.08 * (c2a * c2d + s2a * s2d)) / 2.0;
          else dwv = 0.5 * sa * sd + 0.08 * s2a * s2d;
          // the frame-wide cache covers every window by construction (cap = 2 hw_max + 4 around the
          // frame centre); the clamp only keeps a violated precondition from reading outside LDS
          const double xv = yc[imax(0, imin(cap - 1, first + i - 1 - origin))];
          cplx pr; pr.re = xv * w; pr.im = xv * dwv;
          yw[i] = pr;                                            // one 16-byte store, one 16-byte load per DFT step
          const double cn = ca * cD - sa * sD;
          sa = sa * cD + ca * sD;
          ca = cn;
        }
        wave_sync();
        c_hw = hw; c_first = first;
        WH_ACC_END(1);
      }
      // 6-bin DFTs of both windowed signals, one Goertzel recurrence per (harmonic, phase)
      WH_ACC_BEGIN;
      for (int hi = 0; hi < kIter; ++hi) {
        const int h = hi * LH + hl;
        double are = 0, aim = 0, dre = 0, dim = 0;
        const int idx = h < nh ? mround(bins * (h + 1)) : 0;             // FixF0, harvest.cpp:515: f0c * N / fs * (h + 1)
        const bool reuse = same_window && h < nh && idx == c_idx[hi];
        if (reuse) {
          if (g == 0) { are = c_are[hi]; aim = c_aim[hi]; dre = c_dre[hi]; dim = c_dim[hi]; }
        } else if (h < nh && g < blen) {
          // sum_n v[g+nG] e^{-i theta n}, theta = 2 pi idx G / N:  s[n] = v[n] + 2cos(theta) s[n-1] - s[n-2]
          const double2 st = p.tab.tw[(size_t)((idx * G) & (N - 1)) << (kTwLog2 - lgN)];
          const double c2 = 2.0 * st.x;
          double s1 = 0, s2 = 0, t1 = 0, t2 = 0;
          int i = g;
          // four samples per trip: the kernel is bound by instruction issue (four waves per SIMD cover the recurrence's
          // latency), and index, compare and branch are a third of a two-sample trip
#ifndef WORLD_EMU
          // ... and the trips EVERY phase can take (g + 4 G t + 3 G < blen for g = G - 1 too: blen / 4G of them) run on a
          // wave-uniform counter: no per-lane index, compare and exec-mask update -- 1 vector instruction of overhead per
          // 16 FP64 instead of 5 (round 6: 60 % of this kernel's vector instructions were not FP64).  Same operations,
          // same order per lane; the per-lane loops below take what is left (at most 7 samples).
          {
            const int n_uni = blen / (4 * G);
            const cplx *yp = yw + g;
#pragma unroll 1
            for (int t = 0; t < n_uni; ++t, yp += 4 * G) {
              const cplx p0 = yp[0], p1 = yp[G], p2 = yp[2 * G], p3 = yp[3 * G];
              const double sa = fma(c2, s1, p0.re) - s2, ta = fma(c2, t1, p0.im) - t2;
              const double sb = fma(c2, sa, p1.re) - s1, tb = fma(c2, ta, p1.im) - t1;
              s2 = fma(c2, sb, p2.re) - sa; t2 = fma(c2, tb, p2.im) - ta;
              s1 = fma(c2, s2, p3.re) - sb; t1 = fma(c2, t2, p3.im) - tb;
            }
            i += 4 * G * n_uni;
          }
#endif
          for (; i + 3 * G < blen; i += 4 * G) {
            const cplx p0 = yw[i], p1 = yw[i + G], p2 = yw[i + 2 * G], p3 = yw[i + 3 * G];
            const double sa = fma(c2, s1, p0.re) - s2, ta = fma(c2, t1, p0.im) - t2;
            const double sb = fma(c2, sa, p1.re) - s1, tb = fma(c2, ta, p1.im) - t1;
            s2 = fma(c2, sb, p2.re) - sa; t2 = fma(c2, tb, p2.im) - ta;
            s1 = fma(c2, s2, p3.re) - sb; t1 = fma(c2, t2, p3.im) - tb;
          }
          for (; i + G < blen; i += 2 * G) {
            const cplx p0 = yw[i], p1 = yw[i + G];
            const double a0 = p0.re, d0 = p0.im, a1 = p1.re, d1 = p1.im;
            const double sa = fma(c2, s1, a0) - s2, ta = fma(c2, t1, d0) - t2;
            s2 = sa; t2 = ta;
            s1 = fma(c2, sa, a1) - s1; t1 = fma(c2, ta, d1) - t1;
          }
          if (i < blen) {
            const cplx p0 = yw[i];
            const double sa = fma(c2, s1, p0.re) - s2, ta = fma(c2, t1, p0.im) - t2;
            s2 = s1; t2 = t1; s1 = sa; t1 = ta;
            i += G;
          }
          // i - G is the last sample taken; its phase e^{-2 pi i idx (i-G) / N} closes the sum
          const double2 wl = p.tab.tw[(size_t)((idx * (i - G)) & (N - 1)) << (kTwLog2 - lgN)];
          const double A = s1 - st.x * s2, B = st.y * s2, C = t1 - st.x * t2, D = st.y * t2;
          are = A * wl.x + B * wl.y; aim = B * wl.x - A * wl.y;
          dre = C * wl.x + D * wl.y; dim = D * wl.x - C * wl.y;
        }
#ifndef WORLD_EMU
        static_assert(G == 8, "oct_sum reduces over eight sample phases");
        are = oct_sum(are); aim = oct_sum(aim); dre = oct_sum(dre); dim = oct_sum(dim);
#endif
        c_idx[hi] = h < nh ? idx : -1;
        c_are[hi] = are; c_aim[hi] = aim; c_dre[hi] = dre; c_dim[hi] = dim;
        if (m % G == g) {
          const int q = m / G;
          k_are[q][hi] = are; k_aim[q][hi] = aim; k_dre[q][hi] = dre; k_dim[q][hi] = dim;
          k_idx[q][hi] = idx; k_lgn[q] = lgN; k_f0[q] = f0c;
        }
      }
      WH_ACC_END(2);
      WH_ACC_COUNT(4);
    }
    WH_ACC_BEGIN;
    // deferred tail: lane (h, g) finishes harmonic h of slot m = g (+ q*G)
    for (int q = 0; q < kM; ++q) {
      const int m = q * G + g;
      const double f0c = k_f0[q];
      double num = 0.0, den = 0.0, sc = 0.0;
      int nh = 1;
      if (f0c > 0.0) {
        nh = imin(static_cast<int>(fs / 2.0 / f0c), 6);
        const int N = 1 << k_lgn[q];
        for (int hi = 0; hi < kIter; ++hi) {
          const int h = hi * LH + hl;
          if (h >= nh) continue;
          const double are = k_are[q][hi], aim = k_aim[q][hi], dre = k_dre[q][hi], dim = k_dim[q][hi];
          const double pwv = are * are + aim * aim;                // harvest.cpp:564-569
          const double niv = are * dim - aim * dre;
          const double inst = pwv == 0.0 ? 0.0 : static_cast<double>(k_idx[q][hi]) * fs / N + niv / pwv * fs / 2.0 / kPi;
          const double amp = sqrt(pwv);
          num += amp * inst;                                        // FixF0, harvest.cpp:507-536
          den += amp * (h + 1.0);
          sc += fabs((inst / (h + 1.0) - f0c) / f0c);
        }
      }
#ifndef WORLD_EMU
      // over the harmonics: lane bits 3..5 (xor 8 stays inside a DPP row, 16 and 32 cross rows)
      num += dpp_f64<kDppRor8>(num); den += dpp_f64<kDppRor8>(den); sc += dpp_f64<kDppRor8>(sc);
      // across the four DPP rows: gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap: vector ALU) instead of
      // __shfl_xor's ds_bpermute pairs -- twelve trips through the LDS crossbar per track
      static_assert(2 * G == 16 && WAVE == 64, "lane bits 4 and 5 select the row");
      num = row_pair_sum(num); den = row_pair_sum(den); sc = row_pair_sum(sc);
      num = half_pair_sum(num); den = half_pair_sum(den); sc = half_pair_sum(sc);
#endif
      if (hl == 0 && m < 7) {
        double rf0 = 0.0, rsc = 0.0;
        if (f0c > 0.0) {
          rf0 = num / (den + kTiny);
          rsc = 1.0 / (sc / nh + kTiny);
          if (rf0 < p.f0_floor || rf0 > p.f0_ceil || rsc < 2.5) { rf0 = 0.0; rsc = 0.0; }
        }
        const int slot = j + nc * m;
        dst_f0[slot] = rf0; dst_sc[slot] = rsc;
      }
    }
    WH_ACC_END(3);
  }
  WH_ACC_FLUSH(0, lane == 0);
}

// ---------------------------------------------------------------------------
// RemoveUnreliableCandidates (harvest.cpp:636-688): keep a candidate only if a
// neighbouring frame holds one within 5 %.  Reads the refined set (b), writes (a).
__device__ __forceinline__ double nearest_error(double ref, const double *c, int nc) {
  // SelectBestF0 with allowed_range = 1 (:636-650): min(1, min_i |ref - c_i| / ref).  Division by the common
  // positive ref is monotone in the numerator, so the smallest rounded quotient is the rounded quotient of the
  // smallest distance: one FP64 division per call instead of one per candidate (84 per pruned slot).
  double dmin = 1e300;
  for (int i = 0; i < nc; ++i) {
    const double d = fabs(ref - c[i]);
    dmin = d < dmin ? d : dmin;
  }
  const double e = dmin / ref;
  return e > 1.0 ? 1.0 : e;
}
constexpr int kPruneFrames = 16;                          // frames per workgroup (+1 neighbour row each side; with the pruned scores kept
                                                          // for the base pick the LDS is what 32 frames took without them)
__global__ void hv_prune(HarvestParams p) {
  DYN_LDS(lds);
  double *rows = reinterpret_cast<double *>(lds);         // [kPruneFrames + 2][maxc] refined candidates
  const int u = blockIdx.y, f0 = blockIdx.x * kPruneFrames;
  const int nfb = p.nfb[u];
  if (f0 >= nfb) return;
  const int nslot = p.nc[u] * 7;
  const int nrows = imin(kPruneFrames + 2, nfb - f0 + 1);
  // stage rows f0-1 .. f0+kPruneFrames (coalesced, eight loads in flight per thread at clamped addresses: a load under
  // a condition, stored to LDS in the same trip, made every trip wait for memory -- 14 trips a workgroup); rows
  // outside the utterance are never compared
  // Only the nslot slots in use are staged, as rows of nslot (the other maxc - nslot of a row are never written by
  // hv_refine's tail and never compared).
  const int nt = blockDim.x;
  if (nslot == 0) {                                        // no candidate anywhere in the utterance: nothing to prune, no base
    for (int r = threadIdx.x; r < imin(kPruneFrames, nfb - f0); r += nt) p.c0[(size_t)u * p.fb_stride + f0 + r] = 0.0;
    return;
  }
  {
    constexpr int kB = 8;
    const int n = nrows * nslot;
    for (int i0 = threadIdx.x; i0 < n; i0 += kB * nt) {
      double v[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const int i = imin(i0 + q * nt, n - 1);
        const int r = i / nslot, j = i - r * nslot, f = imax(0, imin(nfb - 1, f0 - 1 + r));
        v[q] = p.cand_b[((size_t)u * p.fb_stride + f) * p.maxc + j];
      }
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const int i = i0 + q * nt;
        if (i < n) {
          const int f = f0 - 1 + i / nslot;
          rows[i] = (f >= 0 && f < nfb) ? v[q] : 0.0;
        }
      }
    }
  }
  __syncthreads();
  // the tile's pruned scores stay in LDS for the base pick below (its F0 is the staged one: a pruned slot's score is 0 and cannot win)
  double *kept_sc = rows + (size_t)(kPruneFrames + 2) * nslot;
  {
    constexpr int kB = 4;                                  // the scores of a thread's next slots are requested together
    const int n = imin(kPruneFrames, nfb - f0) * nslot;    // slots of the tile's frames inside the utterance
    for (int i0 = threadIdx.x; i0 < n; i0 += kB * nt) {
      double scv[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const int i = imin(i0 + q * nt, n - 1);
        const int r = i / nslot, j = i - r * nslot;
        scv[q] = p.score_b[((size_t)u * p.fb_stride + f0 + r) * p.maxc + j];
      }
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const int i = i0 + q * nt;
        if (i >= n) break;
        const int r = i / nslot, j = i - r * nslot, frame = f0 + r;
        const size_t at = ((size_t)u * p.fb_stride + frame) * p.maxc + j;
        double ref = rows[(r + 1) * nslot + j], sc = scv[q];
        if (frame >= 1 && frame < nfb - 1 && ref != 0) {
          double e1 = nearest_error(ref, rows + (r + 2) * nslot, nslot);
          double e2 = nearest_error(ref, rows + r * nslot, nslot);
          if ((e1 < e2 ? e1 : e2) > 0.05) { ref = 0; sc = 0; }
        }
        p.cand_a[at] = ref;
        p.score_a[at] = sc;
        kept_sc[i] = sc;
      }
    }
  }
  __syncthreads();
  // SearchF0Base (harvest.cpp:693-705) on the way: the frame's candidate with the highest score, the FIRST one among
  // equals -- one wavefront per frame, lanes over the slots -- into c0 (round 4: a launch of its own, hc_base, that read
  // both arrays back from HBM).  FixStep1 then needs three neighbouring values of it (harvest_contour.hip).
  {
    const int lane = lane_id(), nfr = imin(kPruneFrames, nfb - f0);
    for (int r = wave_in_block(); r < nfr; r += waves_per_block()) {
      const double *c = rows + (size_t)(r + 1) * nslot, *sv = kept_sc + (size_t)r * nslot;
      double best = 0.0, top = 0.0;
      int slot = 0x7FFFFFFF;                               // lowest slot holding this lane's maximum
      for (int j = lane; j < nslot; j += WAVE) {
        const double sj = sv[j];
        if (sj > top) { best = c[j]; top = sj; slot = j; }
      }
      const double wtop = wave_max(top);
      double v = 0.0;
      if (wtop > 0.0) {
        // among the lanes that hold the maximum the lowest slot wins (the serial loop keeps the first one it meets)
        const int mine = top == wtop ? slot : 0x7FFFFFFF;
        const int win = -wave_max_int(-mine);
#ifndef WORLD_EMU
        v = readlane_f64(best, __builtin_amdgcn_readfirstlane(win % WAVE));
#else
        (void)win;
        v = best;
#endif
      }
      if (lane == 0) p.c0[(size_t)u * p.fb_stride + f0 + r] = v;
    }
  }
}

// ---------------------------------------------------------------------------
size_t hv_band_lds_bytes(int max_half) { return band_lds_bytes(2 * max_half + 1); }

void launch_harvest(const HarvestParams &p, int max_x_len, int max_y_len, int max_fb, int max_frames,
                    hipStream_t stream) {
  const int B = p.b.n_utt;
  // (y needs no clearing: the decimation -- or the copy at ratio 1 -- writes every sample below y_len, and every reader
  // stops there or clamps: a memset per job was one more launch in the in-flight mode)
  if (p.ratio == 1) {
    WH_THREADS(hv_copy_signal, max_y_len, B, 1, stream, p);
  } else {
    IirCoef c = decimate_coef(p.ratio);
    const int spans = (max_x_len + 2 * p.lag + 2 * kDecPad + kDecSpan - 1) / kDecSpan;
    WH_BLOCKS(hv_decimate_fwd, dim3(spans, B), kDecThreads, dec_lds_bytes(), stream, p, c);
    WH_BLOCKS(hv_decimate_bwd, dim3(spans, B), kDecThreads, dec_lds_bytes(), stream, p, c);
  }
  // (the mean's partial sums came with the backward sweep; at ratio 1 a pass of its own forms them)
  if (p.ratio == 1) WH_BLOCKS(hv_partial_sums, dim3(p.mean_parts, B), 256, 64 * sizeof(double), stream, p);
  WH_BLOCKS(hv_remove_mean, dim3(p.nyq_slices, B), 256, 64 * sizeof(double), stream, p);
  const size_t lds = sizeof(double) * (kBandFft + 64 + twiddle_lds_doubles(kBandFftLg - 1));
  if (p.fft_seg > 0) WH_BLOCKS(hv_block_spectra, dim3(p.nblk + p.nch, B), 256, lds, stream, p);       // + the bands' mirror-store constants
  else WH_BLOCKS(hv_band_quirk, dim3(p.nch, B), 64, 64 * sizeof(double), stream, p);
  // The zero-crossing lists are the path's largest workspace -- 152 bands x 4 families x a crossing every other sample at
  // worst (digital silence really produces that): 97 MB per 5 s utterance, two thirds of round 4's 0.14 GB -- and they live
  // only from the filter bank to the interpolation onto the 1 ms grid.  A batch therefore walks its utterances in groups of
  // p.ev_group: the lists are allocated for one group and reused by the next (stream order).
  for (int u0 = 0; u0 < B; u0 += p.ev_group) {
    HarvestParams g = p;
    g.ev_u0 = u0;
    const int G = imin(p.ev_group, B - u0);
    if (p.fft_seg > 0) WH_BLOCKS(hv_band_events_fft, dim3(p.nch, p.nseg, G), 256, lds, stream, g);
    else WH_BLOCKS(hv_band_events, dim3(p.nseg, p.nch, G), kBpThreads, hv_band_lds_bytes(p.max_half), stream, g);
    // one chunk per utterance: the kernel above appended straight into the final lists (seg_events == events)
    if (!(p.fft_seg > 0 && p.nseg == 1)) WH_BLOCKS(hv_compact_events, dim3(p.nch * 4, G), 256, compact_lds_bytes(p.nseg), stream, g);
    WH_BLOCKS(hv_raw_candidates, dim3((p.nch + 7) / 8 * 8, (max_fb + kRawRun - 1) / kRawRun, G), kRawFrames,
              8 * kIntervalCap * sizeof(double) + 4 * sizeof(IntervalRange), stream, g);
  }
  WH_THREADS(hv_detect, max_fb, B, 1, stream, p);
  WH_WAVES(hv_refine, max_fb, B, 1, 3 * sizeof(double) * p.refine_cap, stream, p);
  WH_BLOCKS(hv_prune, dim3((max_fb + kPruneFrames - 1) / kPruneFrames, B), 256,
            sizeof(double) * (size_t)(2 * kPruneFrames + 2) * p.maxc, stream, p);          // + the base pick (c0)
  launch_harvest_contour(p, max_fb, max_frames, stream);
}

}  // namespace world_hip
