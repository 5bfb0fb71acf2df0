// cheaptrick.hip -- CheapTrick spectral envelope on gfx950.
//
// Replaces CheapTrick() / CheapTrickGeneralBody() (reference src/cheaptrick.cpp:22-229)
// and the helpers it pulls from src/common.cpp (DCCorrection :56-75,
// LinearSmoothing :27-111).  The reference walks frames serially with one RNG
// stream; here every frame is an independent workgroup:
//
//   ct_prepare : per utterance, prefix-sum of randn() calls consumed per frame
//                -> each frame's xorshift128 state by GF(2) jump-ahead.
//   ct_frame   : one workgroup per (frame, utterance): F0-adaptive window + noise -> r2c FFT -> |X|^2 -> DC
//                correction -> the mirrored segment of LinearSmoothing and its prefix sum, SERIAL and in FP64
//                order (its rounding is visible in low-energy bins), walked by one lane in LDS
//                while the other wavefronts prepare the per-bin noise terms -> rectangular smoothing ->
//                +|randn|*eps -> log -> r2c FFT (cepstrum) -> lifter -> c2r FFT -> exp -> HBM.
//
// HBM traffic per frame: the window's samples of x (L2-resident: adjacent frames overlap ~97%), the frame's draws
// from the noise table (4 bytes each: window length + one per bin) and one row of the spectrogram written once.
#include "stage_params.h"
#include "prepare.h"
#include "trace.h"
WH_TRACE_DEFINE(ct)

namespace world_hip {

// doubles reserved for Z and the smoothing work area that overlays it: a whole number of the prefix sum's rows of 16
__host__ __device__ __forceinline__ int ct_seg_cap(int N) {
  const int need = N / 2 + 1 + 2 * (N / 3 + 2) + 1;
  const int cap = need > N ? need : N;
  return (cap + 15) / 16 * 16;
}

// ---------------------------------------------------------------------------
__global__ void ct_prepare(CtParams p) {
  DYN_LDS(lds);
  ct_offsets_utt(p, blockIdx.x, reinterpret_cast<double *>(lds));
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

// ---- LinearSmoothing's prefix sum (common.cpp:85-86), SERIAL and in FP64 order -------------------------------------
// `hi - lo` of the smoothing cancels up to 12 digits where the envelope sits at the noise floor, so any other summation
// order moves those bins by 1e-4 (tests/test_gpu_parity.py::test_hard_inputs_vs_oracle): the chain of
// seg_len dependent additions is walked in LDS, by one wavefront of the frame's workgroup.  (Rounds 1-3 took other
// routes: the whole workgroup waiting on a lane that read, added and wrote element by element -- 40 k cycles a frame;
// then a kernel of its own, one lane per frame and 64 chains per wavefront, which needed every frame's segment written
// to HBM, read, written and read again: 7.6 GB per 128 utterances for a stage whose inputs and outputs are 2.4 GB.)
// This is the plain form -- one thread, pairs of values read a batch ahead of the additions -- that the host-compiled
// test build runs; on the GPU it measured 52 k cycles a frame (below) and the DPP row further down replaced it.
constexpr int kScanPairs = 8;         // pairs per batch: 16 chained additions between one group of reads and writes
__device__ __forceinline__ void ct_serial_prefix_sum(double *seg, int seg_len, int cap) {
  double2 *s2 = reinterpret_cast<double2 *>(seg);
  const int pairs = (seg_len + 1) >> 1, last = cap / 2 - 1;            // (an odd length drags one value of scratch along)
  double2 cur[kScanPairs], nxt[kScanPairs];
#pragma unroll
  for (int q = 0; q < kScanPairs; ++q) cur[q] = s2[imin(q, last)];
  double acc = 0.0;                   // 0 + seg[0] == seg[0]: the first element passes through unchanged, as in the reference
  for (int j0 = 0; j0 < pairs; j0 += kScanPairs) {
#pragma unroll
    for (int q = 0; q < kScanPairs; ++q) nxt[q] = s2[imin(j0 + kScanPairs + q, last)];
#pragma unroll
    for (int q = 0; q < kScanPairs; ++q) {                             // strictly left to right
      acc = cur[q].x + acc; cur[q].x = acc;
      acc = cur[q].y + acc; cur[q].y = acc;
    }
#pragma unroll
    for (int q = 0; q < kScanPairs; ++q) if (j0 + q < pairs) s2[j0 + q] = cur[q];
#pragma unroll
    for (int q = 0; q < kScanPairs; ++q) cur[q] = nxt[q];
  }
}

#ifndef WORLD_EMU
// The same chain walked by the lanes of a wavefront, immune to the latency of a loaded LDS: lane k (of the first 16)
// holds element k of a row of 16, the running sum lives in all 16, and step l adds lane l's value to it in the lanes
// >= l only -- so when the row is done lane k is left with the sum up to ITS element; one 128-byte read and one write
// per row, rows requested six ahead, no memory on the chain.  A step is ONE vector instruction: v_fmac_f64 with the
// DPP control row_newbcast:l, which hands every lane of a 16-lane row lane l's operand (the one DPP form CDNA's FP64
// ALU takes, and only in the VOP2 encoding: v_add_f64 has none) -- acc = v * 1.0 + acc, one rounding of the exact
// v + acc, i.e. the addition -- under an EXEC mask the scalar unit rewrites in its shadow.  Measured per frame, in a CU whose other fifteen
// wavefronts are busy with transforms: 52 k cycles for the single-lane walk above (every batch of 16 values waited
// ~750 cycles for its turn in the LDS pipe), 35 k for this scheme with v_readlane pairs feeding a plain v_add_f64
// (three vector instructions a step, each queueing behind the other wavefronts' FP64 work).
// One row: on entry lane 15 holds the sum so far (the last element of the previous row; 0 before the first), which a
// DPP move hands to the 16 lanes first.  (The s_nop pairs are the wait states a DPP read of a freshly written VGPR asks
// for; inside an asm block nobody inserts them for us.)
__device__ __forceinline__ double ct_row_prefix_sum(double v, double acc, double one) {
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %[saved], exec\n s_mov_b64 exec, 0xffff\n s_nop 1\n"
      "v_mov_b64_dpp %[acc], %[acc] row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xfffe\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xfffc\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xfff8\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xfff0\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xffe0\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xffc0\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xff80\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xff00\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xfe00\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xfc00\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xf800\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xf000\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xe000\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0xc000\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, 0x8000\n v_fmac_f64_dpp %[acc], %[v], %[one] row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 exec, %[saved]\n"
      : [acc] "+v"(acc), [saved] "=&s"(saved)
      : [v] "v"(v), [one] "v"(one));
  return acc;
}
constexpr int kScanRow = 16;          // the width of a DPP row
constexpr int kScanAhead = 4;         // rows in flight: 64 elements' worth of additions covers a trip through a busy LDS pipe
__device__ __forceinline__ void ct_wave_prefix_sum(double *seg, int seg_len) {
  const int lane = lane_id(), k = lane & (kScanRow - 1);         // (lanes 16.. mirror the first 16's reads and idle)
  const bool mine = lane < kScanRow;
  const int rows = (seg_len + kScanRow - 1) / kScanRow;           // rows * 16 <= ct_seg_cap: the last row's tail is scratch
  double *at = seg + k;
  double ring[kScanAhead];
  // (rows beyond the segment are read, never added: they lie in the frame's LDS -- the segment's spare room, then P)
#pragma unroll
  for (int q = 0; q < kScanAhead; ++q) ring[q] = at[q * kScanRow];
  double acc = 0.0;                   // 0 + seg[0] == seg[0]: the first element passes through unchanged, as in the reference
  const double one = 1.0;
  for (int r0 = 0; r0 < rows; r0 += kScanAhead) {
#pragma unroll
    for (int q = 0; q < kScanAhead; ++q) {             // (whole groups of rows: a branch per row cost more register moves
      const double v = ring[q];                          // than the few rows of scratch a group may add past the segment)
      ring[q] = at[(q + kScanAhead) * kScanRow];
      acc = ct_row_prefix_sum(v, acc, one);            // strictly left to right
      if (mine && r0 + q < rows) at[q * kScanRow] = acc;
    }
    at += kScanAhead * kScanRow;
  }
}
#endif

// ---- one workgroup per frame: window -> r2c -> |X|^2 -> DC correction -> LinearSmoothing -> log -> cepstrum -> lifter
// -> exp, from x to the spectrogram row without touching HBM in between ---------------------------------------------------
// PER: samples of the window a thread owns (fft_size / threads); LGN: log2(fft_size) when it is a compile-time
// constant of the instantiation (static FFT stages), 0 = taken from p.lg_fft.
// TB: the threads the instantiation is launched with (0: read from the launch -- the emulator's one thread).  512 for
// the 8192-point transform: one frame then owns 118 KB of LDS, one workgroup per CU -- the shape exists so that f0 floors
// below 35 Hz at 48 kHz and sampling rates above 96 kHz run at all, not to be fast.
// CODED: the instantiation that writes coded rows (CtParams::code_ndim > 0).  A kernel of its own: the coder's radix-16
// transform raised the DENSE kernel's registers from 95 to 128 when both lived in one body -- four wavefronts per SIMD
// instead of five, ct_frame +8 % in a batch (A/B, round 6).
template <int PER, int LGN, int TB, bool CODED = false>
// (the 4096-point frame owns 59 KB of LDS: two workgroups per CU whatever the registers, so it may have 256 of them --
// its 48 window values per thread spilled at the 128 of four workgroups per CU)
__global__ void __launch_bounds__(TB > 0 ? TB : 256, TB > 256 ? 1 : LGN == 12 ? 2 : 4) ct_frame(CtParams p) {   // (threads, waves per SIMD)
  DYN_LDS(lds);
  const int lgn = LGN > 0 ? LGN : p.lg_fft, N = 1 << lgn, half = N / 2, nb = half + 1;
  const int fs = p.b.fs;
  const int u = blockIdx.y, f = p.frame_lo + xcd_grouped(blockIdx.x, gridDim.x);
  if (f >= p.b.n_frames[u] || f >= p.frame_hi) return;
  const bool trace_me = f == WH_TRACE_FRAME && u == WH_TRACE_UTT; (void)trace_me;
  WH_STAMP(0, 0);
  // LDS (doubles): Z / the smoothing segment, overlaid (the transforms' buffer is dead while the segment lives and
  // the other way round): ct_seg_cap | P: nb+1 | scratch: 64 | quarter-wave table of the inner N/2-point transform
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds), *seg = Zr;
  const int cap = ct_seg_cap(N);
  double *P = Zr + cap;
  double *scratch = P + (nb + 1) + ((nb + 1) & 1);
  const TwLds tw = stage_twiddles(scratch + 64, lgn - 1, p.tab.tw);

  const size_t fi = (size_t)u * p.b.f_stride + f;
  const double *x = p.b.x + (size_t)u * p.b.x_stride;
  const int x_len = p.b.x_len[u];
  const double pos = p.tpos[fi];
  const double cf0 = ct_effective_f0(p.f0[fi], p.f0_floor, fs);
  const uint32_t *noise = p.noise + p.offsets[fi];
  const int tid = wg_thread<TB>(), nt = wg_size<TB>();
  // (Round 5 tried fetching the NEXT round's draws -- the frame 1280 workgroups on, same XCD -- a dword per line while this
  // frame computes, and d4c_frame its own second and third windows' ahead of time: no change in either kernel, A/B in one call.
  // The window phases are bound by instruction issue among five resident workgroups, not by the cold stream.)

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
  {
    double wv[PER], av[PER], nv[PER];
    double s_ww = 0.0, s_a = 0.0, s_n = 0.0, s_w = 0.0;
    {
      // All of the thread's samples and draws are requested before the first is used, at clamped addresses: a load inside
      // `if (i < wlen)` is waited for where the branch rejoins, so the PER items took one trip to memory EACH (the window
      // phase was 16 of the frame's 75 thousand cycles in a loaded CU -- tools/trace_batch.py; d4c_frame's balance pass
      // learnt this in round 5).  Items beyond the window contribute nothing.
      double xv[PER];
      uint32_t nz[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int i = imin(tid + q * nt, wlen - 1);
        xv[q] = x[(unsigned)imin(x_len - 1, imax(0, origin + i - hw))];
        nz[q] = noise[(unsigned)i];
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) { xv[q] = keep(xv[q]); nz[q] = keep_word(nz[q]); }   // fetched here, not down in the branches
      double rc, rs, dc, ds;
      sincospi(win_scale * (tid - hw), &rs, &rc);
      sincospi(win_scale * nt, &ds, &dc);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int i = tid + q * nt;
        wv[q] = 0.0; av[q] = 0.0; nv[q] = 0.0;
        if (i < wlen) {
          const double w = 0.5 * rc + 0.5;                   // cos(pi * position * f0), cheaptrick.cpp:101-102
          const double a = xv[q] * w, n = randn_value(nz[q]) * kTiny;
          wv[q] = w; av[q] = a; nv[q] = n;
          s_ww += w * w; s_a += a; s_n += n; s_w += w;
        }
        const double t = rc * dc - rs * ds;
        rs = rs * dc + rc * ds;
        rc = t;
      }
    }
    block_sum4<TB>(s_ww, s_a, s_n, s_w, scratch);
    const double c = 1.0 / sqrt(s_ww);
    const double cc = c * ((c * s_a + s_n) / (c * s_w));
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = tid + q * nt;
      if (i < N) rfft_in(Z, i) = i < wlen ? av[q] * c + nv[q] - wv[q] * cc : 0.0;
    }
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
  // ---- LinearSmoothing (common.cpp:27-111), width = 2/3 f0: the mirrored, scaled segment whose prefix sum the
  // reference walks serially, in LDS
  const CtSmooth sm = ct_smooth_shape(cf0, N, fs);
  const double inv_n = 1.0 / N;
  block_map<4, double>(sm.seg_len,
    [&](int i) {
      double m;
      if (i < sm.bnd) m = P[sm.bnd - i];
      else if (i < half + sm.bnd) m = P[i - sm.bnd];
      else m = P[half - (i - (half + sm.bnd))];
      return m * fs * inv_n;                           // == m * fs / N: N is a power of two
    },
    [&](int i, double v) { seg[i] = v; });
  __syncthreads();
  WH_STAMP(0, 5);
  // Wavefront 0 walks the chain; the others meanwhile prepare what the rest of the frame needs
  // and the chain does not touch: AddInfinitesimalNoise's per-bin terms -- the draws continue the frame's stream after
  // the window's (cheaptrick.cpp:147-151) -- into P, whose power spectrum is dead.  (One wavefront -- the emulator's
  // one thread -- does both in turn.)
  {
    const int waves = wg_waves<TB>();
    constexpr int scan_wave = 0;        // (rotating the chain over the wavefronts, i.e. the SIMDs, by frame: 5.26 ms against 5.06)
    const int helpers = waves > 1 ? nt - WAVE : nt;
    const bool helper = waves == 1 || wave_in_block() != scan_wave;
#ifndef WORLD_EMU
    if (wave_in_block() == scan_wave) {
      // A chain of dependent additions advances one issue slot at a time; among the wavefronts of its SIMD, all busy with
      // other frames' transforms, it got every fourth slot or so.  Issue priority for the chain's duration gives the
      // frame its latency back (5.53 -> 5.26 ms per 128 utterances when the chain was still a single lane).
      __builtin_amdgcn_s_setprio(3);
      ct_wave_prefix_sum(seg, sm.seg_len);
      __builtin_amdgcn_s_setprio(0);
    }
#else
    if (tid == 0) ct_serial_prefix_sum(seg, sm.seg_len, cap);
#endif
    if (helper) {
      const int ht = (waves > 1 && wave_in_block() > scan_wave) ? tid - WAVE : tid;
      constexpr int kB = 6;                                  // draws requested together
      for (int i0 = ht; i0 <= half; i0 += kB * helpers) {
        uint32_t d[kB];
#pragma unroll
        for (int q = 0; q < kB; ++q) d[q] = noise[wlen + imin(i0 + q * helpers, half)];
#pragma unroll
        for (int q = 0; q < kB; ++q) if (i0 + q * helpers <= half) P[i0 + q * helpers] = fabs(randn_value(d[q])) * kEps;
      }
    }
  }
  __syncthreads();
  WH_STAMP(0, 6);
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
        return log(smoothed + P[i]);                   // the log of SmoothingWithRecovery (cheaptrick.cpp:39-42)
      },
      [&](int i, double lg) { P[i] = lg; });
    __syncthreads();                                   // the transforms below overwrite the segment
  }

  WH_STAMP(0, 7);
  // ---- SmoothingWithRecovery (cheaptrick.cpp:22-57) -------------------------
  // the symmetric extension of the log spectrum is read by the first FFT stage directly from P.
  const double q1 = p.q1, f0_over_fs = cf0 / fs;
  auto mirrored = [&](int i) { return i <= half ? P[i] : P[N - i]; };
  block_rfft_from<kCtMaxLr, LGN>(Z, lgn, tw, [&](int n) { cplx v; v.re = mirrored(2 * n); v.im = mirrored(2 * n + 1); return v; },
                  [&](int k, double re, double im) {
    (void)im;
    // Lifters: sin(pi f0 q) / (pi f0 q) and (1 - 2 q1) + 2 q1 cos(2 pi f0 q) at quefrency q = k / fs; with
    // a = f0 k / fs both come from ONE sinpi: cos(2 pi a) = 1 - 2 sin^2(pi a).
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
  if constexpr (CODED) {
    // ---- CodeSpectralEnvelope of this row (codec.cpp:268-297), fused: the row never goes to HBM ---------------------------
    // codec_code_sp (codec.hip) on the dense row, operation for operation: log of the value the row would hold, interp1 onto the
    // mel axis through the coder's own tables, DCTForCodec's reordering, ONE real transform of fft_size/2 points on the same
    // plan and twiddle table (the staged table is the coder's: its lg_md is this kernel's lgn - 1), the first ndim bins.
    block_map<4, double>(half + 1, [&](int i) { return log(exp(rfft_in(Z, i))); }, [&](int i, double lg) { P[i] = lg; });
    __syncthreads();                                   // every reader of the envelope is done: Z becomes the DCT's input
    const int md = half;
    for (int i = tid; i < md; i += nt) {
      const int k = p.code_knot[i];
      const double v = P[k - 1] + p.code_frac[i] * (P[k] - P[k - 1]);
      const int dest = (i & 1) ? md / 2 + (md - 1 - i) / 2 : i / 2;
      rfft_in(Z, dest) = v;
    }
    const double norm = sqrt(static_cast<double>(md));
    double *out = reinterpret_cast<double *>(out_at);
    block_rfft(Z, lgn - 1, tw, [&](int k, double re, double im) {
      if (k < p.code_ndim) out[k] = (re * p.code_w_re[k] - im * p.code_w_im[k]) / norm;
    });
    return;
  } else if (p.out_f32) {
    float *out = reinterpret_cast<float *>(out_at);
    block_map<4, double>(half + 1, [&](int i) { return exp(rfft_in(Z, i)); }, [&](int i, double v) { out[i] = static_cast<float>(v); });
  } else {
    double *out = reinterpret_cast<double *>(out_at);
    block_map<4, double>(half + 1, [&](int i) { return exp(rfft_in(Z, i)); }, [&](int i, double v) { out[i] = v; });
  }
  WH_STAMP(0, 10);
}

// ---------------------------------------------------------------------------
size_t ct_frame_lds_bytes(int lg_fft) {
  int N = 1 << lg_fft, nb = N / 2 + 1;
  return sizeof(double) * (size_t)(ct_seg_cap(N) + nb + 1 + ((nb + 1) & 1) + 64 + N / 8 + 2);
}

size_t ct_max_draws_per_frame(int fft_size) { return (size_t)fft_size + fft_size / 2 + 1; }   // window < fft_size, + bins

void launch_cheaptrick(const CtParams &p, int max_frames_all, hipStream_t stream) {
  if (!p.skip_prepare) WH_BLOCKS(ct_prepare, dim3(p.b.n_utt), 1024, 64 * sizeof(double), stream, p);    // 1024 threads: a 10 s utterance is two scans
  const int max_frames = imin(max_frames_all, p.frame_hi) - p.frame_lo;          // frames of the range
  if (max_frames <= 0) return;
  const dim3 grid(max_frames, p.b.n_utt);
  const size_t lds = ct_frame_lds_bytes(p.lg_fft);
#ifdef WORLD_EMU
  if (p.code_ndim > 0) devrt::launch_blocks("ct_frame_coded", ct_frame<8192, 0, 0, true>, grid, 256, lds, stream, p);
  else devrt::launch_blocks("ct_frame", ct_frame<8192, 0, 0>, grid, 256, lds, stream, p);
#else
  // Workgroup size follows the transform: fft_size 1024 (fs <= 24 kHz) runs with 128 threads (64 butterflies per
  // radix-8 stage; measured 1.33 ms for 64 x 1001 frames against 1.60 with 256 threads and 1.49 with 64), 2048 and
  // 4096 with 256.  The three sizes the sampling rates of speech lead to get compile-time plans.
  if (p.code_ndim > 0) {
    if (p.lg_fft == 10) devrt::launch_blocks("ct_frame_coded", ct_frame<8, 10, 128, true>, grid, 128, lds, stream, p);
    else if (p.lg_fft == 11) devrt::launch_blocks("ct_frame_coded", ct_frame<8, 11, 256, true>, grid, 256, lds, stream, p);
    else if (p.lg_fft == 12) devrt::launch_blocks("ct_frame_coded", ct_frame<16, 12, 256, true>, grid, 256, lds, stream, p);
    else if (p.lg_fft == 13) devrt::launch_blocks("ct_frame_coded", ct_frame<16, 0, 512, true>, grid, 512, lds, stream, p);
    else devrt::launch_blocks("ct_frame_coded", ct_frame<8, 0, 128, true>, grid, 128, lds, stream, p);
  } else if (p.lg_fft == 10) devrt::launch_blocks("ct_frame", ct_frame<8, 10, 128>, grid, 128, lds, stream, p);
  else if (p.lg_fft == 11) devrt::launch_blocks("ct_frame", ct_frame<8, 11, 256>, grid, 256, lds, stream, p);
  else if (p.lg_fft == 12) devrt::launch_blocks("ct_frame", ct_frame<16, 12, 256>, grid, 256, lds, stream, p);
  else if (p.lg_fft == 13) devrt::launch_blocks("ct_frame", ct_frame<16, 0, 512>, grid, 512, lds, stream, p);
  else devrt::launch_blocks("ct_frame", ct_frame<8, 0, 128>, grid, 128, lds, stream, p);
#endif
}

}  // namespace world_hip
