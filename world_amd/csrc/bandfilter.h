// bandfilter.h -- FIR filtering out of LDS fused with the four zero-crossing
// detectors, shared by Harvest (152 band-pass channels, harvest.cpp:99-238) and
// DIO (low-pass channels, dio.cpp:296-435).
//
// The reference filters by whole-utterance FFTs (r2c, multiply, c2r of 2^16..2^19
// points per channel) and then scans the filtered signal four times.  The filters
// are short FIRs, so here a workgroup convolves one tile of one channel directly in
// LDS (FP64 FMA bound; the input tile is read once) and runs the four detectors
// on the tile while it is still in LDS: the filtered signal never exists in HBM,
// only the compacted sub-sample crossing times do.  (The unnormalised inverse FFT
// of the reference scales the signal by a power of two, which cancels exactly in
// the crossing-time ratio below.)
#pragma once
#include "trace.h"
#include "common.h"

namespace world_hip {

constexpr int kBpThreads = 256;
constexpr int kOutPer = 8;                     // consecutive outputs per thread (register sliding window)
constexpr int kTile = kBpThreads * kOutPer;    // filtered samples produced per step (+2 look-ahead)
constexpr int kSegTiles = 2;
constexpr int kSeg = kTile * kSegTiles;        // one workgroup = one segment of one channel
constexpr int kSegCap = kSeg / 2 + 2;          // a crossing needs two samples

// LDS arrays indexed with a stride of kOutPer per lane are stored with one pad slot
// every kOutPer doubles: lane stride 9 doubles -> ds_read_b64 / ds_write_b64 conflict-free.
__host__ __device__ __forceinline__ int pad8(int i) { return i + (i >> 3); }

struct BandJob {
  const double *in;       // input signal, zero outside [0, in_len)
  int in_len;
  int n;                  // number of filtered samples of interest: i in [0, n)
  const double *taps;     // FIR taps h[0..ntap)
  int ntap;
  int shift;              // filtered[i] = sum_j h[j] * in[i + shift - j]
  int max_ntap;           // LDS is carved for the longest filter of the launch
  int nseg;
  double *seg_events;     // [4][nseg][kSegCap]
  int *seg_count;         // [4][nseg]
  // Optional additive term of the reference's FFT filtering (mirror_store_constants above):
  // filtered[i] += (-1)^n (q[0] cos(2 pi n/N) + q[1] sin(2 pi n/N) + q[2]) with n = i + quirk_delay,
  // q[3] = 2/N.  nullptr = none.
  const double *quirk = nullptr;
  int quirk_delay = 0;
};

// ---- the reference's mirror store (harvest.cpp:121-132, dio.cpp:317-330) -------------------
// Both GetFilteredSignal functions multiply the spectra bin by bin and copy every product to bin
// N-i-1 -- one off the Hermitian partner.  Two of those stores land inside the half spectrum the
// inverse transform reads: at i = N/2-1 the product P = Y[N/2-1] H[N/2-1] overwrites the filter's
// Nyquist bin BEFORE it is multiplied, and at i = N/2 the product Y[N/2] P overwrites bin N/2-1.
// Both bins end up holding Y[N/2] P instead of P and Y[N/2] H[N/2], which adds
//   (-1)^n [ 2 Re((Y[N/2] - 1) P e^{-2 pi i n / N}) + Y[N/2] (Re P - H[N/2]) ]
// to the (unnormalised) filtered signal -- a Nyquist-rate ripple some ten orders of magnitude
// below speech level.  It decides results in two situations: (1) DIO's 4..8-tap channels after
// speed = 8-12 on 16-22 kHz input, where P is not small (F0 off by up to 1.6e-2 without it);
// (2) stretches where the input is exactly constant (digital silence), where the true filtered
// signal vanishes and the ripple alone produces the zero crossings, so the interval-F0
// interpolation of the neighbouring frames ends at the stretch instead of bridging it (Harvest off
// by 1.5e-2 and a voiced/unvoiced flip next to a 22 ms hole without it).  The FIR paths therefore
// add the term before the zero-crossing search.  Y = spectrum of the (mean-free, zero-padded)
// signal, N = the reference's transform length for this utterance, both per utterance.

// block-wide: s0 = sum v[i] (-1)^i, (s1r, s1i) = sum v[i] (-1)^i e^{+2 pi i i / N}, w = 2 / N
// (the sums run over i in [lo, hi): a long signal is split over several workgroups)
__device__ __forceinline__ void nyquist_pair(const double *v, int n, double w, double *scratch, double *s0, double *s1r,
                                             double *s1i, int lo = 0, int hi = -1) {
  if (hi < 0 || hi > n) hi = n;
  double a = 0.0, br = 0.0, bi = 0.0;
  // thread-strided samples: one sincospi each for the first phase and the stride, then rotations
  double sn, cs, sr, cr;
  sincospi((double)(lo + (int)threadIdx.x) * w, &sn, &cs);
  sincospi((double)blockDim.x * w, &sr, &cr);
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const double x = (i & 1) ? -v[i] : v[i];
    a += x; br += x * cs; bi += x * sn;
    const double c2 = cs * cr - sn * sr;
    sn = sn * cr + cs * sr;
    cs = c2;
  }
  block_sum3(a, br, bi, scratch);
  *s0 = a; *s1r = br; *s1i = bi;
}
// constants of the term for one channel, scaled by 1/N (the FIR paths work on signal / N):
// q = {A.re, A.im, B, w} with term(n) = (-1)^n (A.re cos(pi w n) + A.im sin(pi w n) + B)
__device__ __forceinline__ void mirror_store_constants(double y0, double y1r, double y1i, double h0, double h1r,
                                                       double h1i, double w, double *q) {
  const double pr = y1r * h1r - y1i * h1i, pi = y1r * h1i + y1i * h1r;   // P
  const double inv_n = 0.5 * w;
  q[0] = 2.0 * (y0 - 1.0) * pr * inv_n;
  q[1] = 2.0 * (y0 - 1.0) * pi * inv_n;
  q[2] = y0 * (pr - h0) * inv_n;
  q[3] = w;
}

inline size_t band_lds_bytes(int max_ntap) {
  return sizeof(double) * (size_t)((max_ntap + 1) + pad8(kTile + 2 + max_ntap + 3 + 8) + 1 + pad8(kTile + 4) + 1 + 64);
}
inline int band_segments(int n) { return (n + kSeg - 1) / kSeg; }

// sub-sample crossing time between samples e-1 and e (harvest.cpp:183-186, dio.cpp:380-382)
__device__ __forceinline__ double fine_edge(int e, double prev, double cur) { return e - prev / (cur - prev); }

// ---- one tile of one channel -------------------------------------------------------
// s[pad8(k)] = filtered[t0 + k] for k in [0, kTile + 2); taps are in LDS.
// Input index for output k, tap j: t0 + k + shift - j -> tile index k + (ntap-1) - j with
// tile[0] = in[t0 + shift - (ntap-1)].  The tile is stored `org` slots into yt, org chosen so
// that (k0 + ntap-1 + org) = 7 (mod 8) for every thread's first output k0 (a multiple of 8):
// each group of 8 taps then reads 8 inputs that lie in ONE padded block of 8, so their LDS
// addresses are a base plus compile-time offsets.  Each thread produces kOutPer consecutive
// outputs from a register window that slides one input per tap: per 8 taps 16 LDS reads
// and 64 FMAs -> FP64-FMA bound instead of LDS bound.
__device__ __forceinline__ int tile_org(int ntap) { return (8 - (ntap & 7)) & 7; }

// The tile's input is fetched into registers first (so that the NEXT tile's HBM/L2 latency
// hides behind the current tile's FMAs) and committed to LDS later.
constexpr int kTileRegs = 12;                             // >= (kTile + 2 + taps - 1) / kBpThreads for <= 1022 taps
struct TileRegs { double v[kTileRegs]; };
__device__ __forceinline__ bool tile_fits_regs(const BandJob &job) {
  return (kTile + 2 + job.ntap - 1) <= kTileRegs * (int)blockDim.x;
}
__device__ __forceinline__ void tile_fetch(const BandJob &job, int t0, TileRegs &r) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lo = t0 + job.shift - (job.ntap - 1);
  const int count = kTile + 2 + job.ntap - 1;
#pragma unroll
  for (int q = 0; q < kTileRegs; ++q) {
    const int k = tid + q * nt, idx = lo + k;
    r.v[q] = (k < count && idx >= 0 && idx < job.in_len) ? job.in[idx] : 0.0;
  }
}
__device__ __forceinline__ void tile_commit(const BandJob &job, const TileRegs &r, double *yt) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int org = tile_org(job.ntap), count = kTile + 2 + job.ntap - 1;
#pragma unroll
  for (int q = 0; q < kTileRegs; ++q) {
    const int k = tid + q * nt;
    if (k < count) yt[pad8(k + org)] = r.v[q];
  }
}

// convolution of the tile resident in yt
__device__ __forceinline__ void fir_compute(const BandJob &job, const double *taps, const double *yt, double *s) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ntap = job.ntap;
  const int org = tile_org(ntap);

  for (int k0 = tid * kOutPer; k0 < kTile; k0 += nt * kOutPer) {
    double acc[kOutPer], w[kOutPer];
    const int base = k0 + ntap - 1 + org;               // = 7 (mod 8)
    const double *top = yt + pad8(base);                // tile element base; base+q (q >= 1) is at top[1 + q]
    acc[0] = 0.0; w[0] = top[0];
#pragma unroll
    for (int q = 1; q < kOutPer; ++q) { acc[q] = 0.0; w[q] = top[1 + q]; }
    // logical window at tap j: L_j[q] = tile[base + q - j] = w[(q - j) mod kOutPer].
    // Groups of kOutPer taps: the group's kOutPer new inputs and taps are fetched
    // first (independent LDS reads in flight), then kOutPer^2 FMAs run from registers.
    const int ntap_main = ntap - (ntap % kOutPer);
    const double *blk = top;                             // tile[base - j0 - u] = blk[-u] for u < 8
    for (int j0 = 0; j0 < ntap_main; j0 += kOutPer, blk -= kOutPer + 1) {
      double fresh[kOutPer], h[kOutPer];
#pragma unroll
      for (int u = 0; u < kOutPer; ++u) {
        h[u] = taps[j0 + u];
        fresh[u] = blk[-u];                              // slot u = 0 is only needed for j0 > 0
      }
#pragma unroll
      for (int u = 0; u < kOutPer; ++u) {
        if (j0 + u > 0) w[(kOutPer - u) % kOutPer] = fresh[u];
#pragma unroll
        for (int q = 0; q < kOutPer; ++q) acc[q] = fma(h[u], w[(q + kOutPer - u) % kOutPer], acc[q]);
      }
    }
#pragma unroll
    for (int u = 0; u < kOutPer; ++u) {                 // remaining ntap % kOutPer taps
      const int j = ntap_main + u;
      if (j < ntap) {
        if (j > 0) w[(kOutPer - u) % kOutPer] = blk[-u];
        const double hh = taps[j];
#pragma unroll
        for (int q = 0; q < kOutPer; ++q) acc[q] = fma(hh, w[(q + kOutPer - u) % kOutPer], acc[q]);
      }
    }
    double *sdst = s + pad8(k0);                         // k0 = 0 (mod 8): outputs are contiguous
#pragma unroll
    for (int q = 0; q < kOutPer; ++q) sdst[q] = acc[q];
  }
  // the two look-ahead samples: one wave each, taps spread over the lanes
  for (int e = wave_in_block(); e < 2; e += waves_per_block()) {
    double acc = 0.0;
    const int base = kTile + e + (ntap - 1) + org;
    for (int j = lane_id(); j < ntap; j += WAVE) acc = fma(taps[j], yt[pad8(base - j)], acc);
    acc = wave_sum(acc);
    if (lane_id() == 0) s[pad8(kTile + e)] = acc;
  }
  __syncthreads();
}

// unpipelined variant (one-off tiles, e.g. DIO's low-cut stage)
__device__ __forceinline__ void fir_tile(const BandJob &job, const double *taps, int t0, double *yt, double *s) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int org = tile_org(job.ntap);
  const int lo = t0 + job.shift - (job.ntap - 1);
  const int count = kTile + 2 + job.ntap - 1;
  __syncthreads();
  for (int k = tid; k < count; k += nt) {
    int idx = lo + k;
    yt[pad8(k + org)] = (idx >= 0 && idx < job.in_len) ? job.in[idx] : 0.0;
  }
  __syncthreads();
  fir_compute(job, taps, yt, s);
}

// sub-sample crossing time with a reciprocal-based quotient (common.h: fast_div): the exact quotient is not part of
// the contract -- an ulp of prev / (cur - prev) moves a crossing by 1e-16 samples -- and its ~25-instruction dependent
// chain per event was half of the event phase.  Denominators outside the normal range take the exact division.
__device__ __forceinline__ double fine_edge_fast(int e, double prev, double cur) {
  const double den = cur - prev;
  const double qt = fabs(den) > 1e-290 ? fast_div(prev, den) : prev / den;
  return e - qt;
}

// The four zero-crossing families (falling, rising, peaks, dips: harvest.cpp:162-238, dio.cpp:357-435) of the `len`
// filtered samples that start at time index t0; sample(k) = filtered[t0 + k] for k in [0, len + 2).  Every thread
// inspects PER consecutive samples (time order) and records the crossings of each family as a bit mask; ONE
// block scan of the four packed counts gives the list positions; the sub-sample times are then evaluated only for
// the set bits -- the four families side by side, so that four independent quotient chains are in flight -- and
// appended to the lists (capacity `cap` each; ev + fam * fam_stride is family fam's list, count[] its fill).
// PER: consecutive samples a thread inspects per pass (the FFT path covers its whole block in one pass of 16).
// PRE = false: the caller has a barrier between this call and whoever used `scratch` last.
template <int PER = kOutPer, bool PRE = true, class Sample>
__device__ __forceinline__ void tile_events(Sample sample, int t0, int len, int n, double *ev, size_t fam_stride,
                                            int (&count)[4], int cap, double *scratch, bool trace_me = false) {
  const int tid = threadIdx.x, nt = blockDim.x;
  (void)trace_me;
  WH_ACC_DECL;
  for (int sub = 0; sub < len; sub += nt * PER) {
    WH_ACC_BEGIN;
    const int kbase = sub + tid * PER;
    unsigned mask[4] = {0u, 0u, 0u, 0u};
    {
      double a = kbase < len + 2 ? sample(kbase) : 0.0, b = kbase + 1 < len + 2 ? sample(kbase + 1) : 0.0;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const double c = kbase + q + 2 < len + 2 ? sample(kbase + q + 2) : 0.0;
        const int i = t0 + kbase + q;
        const bool live = kbase + q < len;
        const double da = b - a, db = c - b;
        const bool r01 = live && i <= n - 2, r23 = live && i <= n - 3;
        if (r01 && 0.0 < a && b <= 0.0) mask[0] |= 1u << q;
        if (r01 && a < 0.0 && 0.0 <= b) mask[1] |= 1u << q;
        if (r23 && 0.0 < da && db <= 0.0) mask[2] |= 1u << q;
        if (r23 && da < 0.0 && 0.0 <= db) mask[3] |= 1u << q;
        a = b; b = c;
      }
    }
    WH_ACC_END(0);
    WH_ACC_BEGIN;
    unsigned long long packed = 0;
#pragma unroll
    for (int fam = 0; fam < 4; ++fam) packed |= (unsigned long long)__builtin_popcount(mask[fam]) << (16 * fam);
    unsigned long long total, off = block_excl_scan_u64<PRE>(packed, &total, scratch);
    WH_ACC_END(1);
    WH_ACC_BEGIN;
    int at[4];
#pragma unroll
    for (int fam = 0; fam < 4; ++fam) at[fam] = count[fam] + (int)((off >> (16 * fam)) & 0xFFFF);
    // the samples of a crossing come from LDS again (a register array indexed by the bit position would live in scratch)
    while ((mask[0] | mask[1] | mask[2] | mask[3]) != 0u) {
#pragma unroll
      for (int fam = 0; fam < 4; ++fam) {
        if (mask[fam] != 0u) {
          const int q = __builtin_ctz(mask[fam]);
          mask[fam] &= mask[fam] - 1;
          double a = sample(kbase + q), b = sample(kbase + q + 1);
          if (fam >= 2) { const double c = sample(kbase + q + 2); a = b - a; b = c - b; }
          if (at[fam] < cap) ev[fam * fam_stride + at[fam]] = fine_edge_fast(t0 + kbase + q + 1, a, b);
          ++at[fam];
        }
      }
    }
#pragma unroll
    for (int fam = 0; fam < 4; ++fam) count[fam] += (int)((total >> (16 * fam)) & 0xFFFF);
    WH_ACC_END(2);
  }
  WH_ACC_FLUSH(32, tid == 0);
}

// Whole segment `seg` of one channel: filter tile by tile and append the crossing
// times of the four families (falling, rising, peaks, dips) to the segment's lists.
__device__ __forceinline__ void band_events_segment(const BandJob &job, int seg) {
  DYN_LDS(lds);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = job.n;
  const int seg_begin = seg * kSeg;
  int *cnt_out = job.seg_count + seg;
  if (seg_begin >= n) {
    if (tid == 0) for (int fam = 0; fam < 4; ++fam) cnt_out[(size_t)fam * job.nseg] = 0;
    return;
  }
  const int seg_end = imin(n, seg_begin + kSeg);
  const bool trace_me = blockIdx.x == 5 && blockIdx.y == 20; (void)trace_me;
  WH_ACC_DECL;
  WH_ACC_BEGIN;
  double *taps = reinterpret_cast<double *>(lds);
  double *yt = taps + (job.max_ntap + 1);
  double *s = yt + pad8(kTile + 2 + job.max_ntap + 3 + 8) + 1;
  double *scratch = s + pad8(kTile + 4) + 1;
  for (int j = tid; j < job.ntap; j += nt) taps[j] = job.taps[j];

  double *ev = job.seg_events + (size_t)seg * kSegCap;
  const size_t fam_stride = (size_t)job.nseg * kSegCap;
  int count[4] = {0, 0, 0, 0};
  const bool pipelined = tile_fits_regs(job);
  TileRegs pre;
  if (pipelined) tile_fetch(job, seg_begin, pre);
  WH_ACC_END(0);
  for (int t0 = seg_begin; t0 < seg_end; t0 += kTile) {
    WH_ACC_BEGIN;
    if (pipelined) {
      __syncthreads();                               // everyone is done with the previous tile's yt and s
      tile_commit(job, pre, yt);
      __syncthreads();
      if (t0 + kTile < seg_end) tile_fetch(job, t0 + kTile, pre);   // in flight during this tile's FMAs
      WH_ACC_END(1);
      WH_ACC_BEGIN;
      fir_compute(job, taps, yt, s);
    } else {
      fir_tile(job, taps, t0, yt, s);
    }
    if (job.quirk) {
      // a thread's samples are nt apart: one sincospi for its first sample, then a rotation by nt
      // samples' phase per further one
      const double qc = job.quirk[0], qs = job.quirk[1], q0 = job.quirk[2], w = job.quirk[3];
      const int n0 = t0 + tid + job.quirk_delay;
      double sn, cs, sr, cr;
      sincospi(n0 * w, &sn, &cs);
      sincospi(nt * w, &sr, &cr);
      const double sign = (n0 & 1) ? -1.0 : 1.0;
      for (int k = tid, j = 0; k < kTile + 2; k += nt, ++j) {
        const double flip = ((nt & 1) && (j & 1)) ? -sign : sign;     // (-1)^(n0 + j nt)
        s[pad8(k)] += flip * (qc * cs + qs * sn + q0);
        const double c2 = cs * cr - sn * sr;
        sn = sn * cr + cs * sr;
        cs = c2;
      }
      __syncthreads();
    }
    WH_ACC_END(2);
    WH_ACC_BEGIN;
    tile_events([&](int k) { return s[pad8(k)]; }, t0, kTile, n, ev, fam_stride, count, kSegCap, scratch);
    WH_ACC_END(3);
  }
  WH_ACC_SET(4, job.ntap);
  WH_ACC_FLUSH(16, tid == 0);
  if (tid == 0)
    for (int fam = 0; fam < 4; ++fam) cnt_out[(size_t)fam * job.nseg] = imin(count[fam], kSegCap);
}

// concatenate the segment lists of one (channel, family) in time order.  The counts become offsets by one block
// scan, then the copy is ONE flat pass with eight loads in flight per thread (a loop over the segments paid a
// trip to memory for every segment's count and another for its copy, one after the other).
// lds: compact_lds_bytes(nseg) bytes.
inline size_t compact_lds_bytes(int nseg) { return 64 * sizeof(double) + sizeof(int) * (size_t)(nseg + 1); }
__device__ __forceinline__ void compact_event_segments(const double *seg_events, const int *seg_count, int nseg, int seg_cap,
                                                       double *events, int ev_cap, int *ev_count, char *lds) {
  double *scratch = reinterpret_cast<double *>(lds);
  int *pre = reinterpret_cast<int *>(scratch + 64);          // pre[s] = events before segment s, pre[nseg] = all
  const int tid = threadIdx.x, nt = blockDim.x;
  int running = 0;
  for (int s0 = 0; s0 < nseg; s0 += nt) {
    const int s = s0 + tid;
    const int c = s < nseg ? seg_count[s] : 0;
    int tot;
    const int at = block_excl_scan_int(c, &tot, scratch);
    if (s < nseg) pre[s] = running + at;
    running += tot;
  }
  if (tid == 0) pre[nseg] = running;
  __syncthreads();
  const int total = imin(running, ev_cap);
  constexpr int kB = 8;
  int s = 0;
  for (int j0 = tid; j0 < total; j0 += kB * nt) {
    double v[kB];
#pragma unroll
    for (int q = 0; q < kB; ++q) {
      const int j = imin(j0 + q * nt, total - 1);
      while (j >= pre[s + 1]) ++s;                            // (a thread's elements ascend: the segment only moves forward)
      v[q] = seg_events[(size_t)s * seg_cap + (j - pre[s])];
    }
#pragma unroll
    for (int q = 0; q < kB; ++q) if (j0 + q * nt < total) events[j0 + q * nt] = v[q];
  }
  if (tid == 0) *ev_count = total;
}

// ---- interval F0s of one family and their interpolation onto frame times ----------
__device__ __forceinline__ double interval_loc(const double *e, int k, double fs) { return (e[k] + e[k + 1]) / 2.0 / fs; }
__device__ __forceinline__ double interval_f0(const double *e, int k, double fs) { return fs / (e[k + 1] - e[k]); }

// #{intervals whose location is <= t} of one family (the bin interp1 derives from it, matlabfunctions.cpp:
// 136-176, is clamp(count, 1, n-1), so both ends extrapolate linearly).  Crossings of a band-passed signal
// are nearly uniform in time, so a proportional guess lands within a few intervals of the answer: bracket
// it by doubling steps from the guess, then bisect the bracket (same result as bisecting [0, n_int),
// ~4 probes, not 10).
__device__ __forceinline__ int intervals_at_or_before(const double *e, int n_int, double fs, double t) {
  int lo = 0, hi = n_int;
  if (n_int > 8) {
    const double first = interval_loc(e, 0, fs), last = interval_loc(e, n_int - 1, fs);
    const double r = (t - first) / (last - first) * (n_int - 1);
    int g = r < 0.0 ? 0 : (r > n_int - 1.0 ? n_int - 1 : static_cast<int>(r));
    if (interval_loc(e, g, fs) <= t) {            // every index <= g is <= t
      lo = g + 1;
      for (int step = 1, q = g + 1; ; q += step, step <<= 1) {
        if (q >= n_int) break;                    // hi stays n_int
        if (interval_loc(e, q, fs) > t) { hi = q; break; }
        lo = q + 1;
      }
    } else {                                      // every index >= g is > t
      hi = g;
      for (int step = 1, q = g - 1; ; q -= step, step <<= 1) {
        if (q < 0) break;                         // lo stays 0
        if (interval_loc(e, q, fs) <= t) { lo = q + 1; break; }
        hi = q;
      }
    }
  }
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (interval_loc(e, mid, fs) <= t) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// The same count found by 32 lanes together (every lane of the group passes the same arguments and gets the result):
// the serial search above is a chain of ~9 dependent probes of global memory, and a workgroup of hv_raw_candidates
// did nothing else while its eight end-point searches ran -- most of the kernel's time.  Here a round is ONE probe
// per lane: 32 consecutive intervals around the proportional guess (which brackets the answer for a band-passed
// signal), else 32 strided probes that narrow the range 32-fold.  The predicate is monotone in the index, so the
// number of lanes that see loc <= t is the position of the last such interval.  Two or three rounds.
// `group`: which 32-lane half of its wavefront the caller is (0 / 1); `sub`: lane within the group.
__device__ __forceinline__ int intervals_at_or_before_group(const double *e, int n_int, double fs, double t, int group, int sub) {
#ifdef WORLD_EMU
  (void)group; (void)sub;
  return intervals_at_or_before(e, n_int, fs, t);
#else
  if (n_int <= 0) return 0;
  const double first = interval_loc(e, 0, fs), last = interval_loc(e, n_int - 1, fs);
  if (!(first <= t)) return 0;
  if (last <= t) return n_int;
  // invariant: loc(lo) <= t < loc(hi); the answer is (last index with loc <= t) + 1, in (lo, hi]
  int lo = 0, hi = n_int - 1;
  auto ones_at = [&](int idx, bool valid) {
    const bool pred = valid && interval_loc(e, idx, fs) <= t;
    return __popc((unsigned)(__ballot(pred) >> (32 * group)));
  };
  if (hi - lo > 32) {
    const double r = (t - first) / (last - first) * (n_int - 1);
    const int g = r < 0.0 ? 0 : (r > n_int - 1.0 ? n_int - 1 : static_cast<int>(r));
    int base = g - 15;
    base = base < lo + 1 ? lo + 1 : base;
    base = base > hi - 32 ? hi - 32 : base;                 // window [base, base + 31] inside (lo, hi)
    const int ones = ones_at(base + sub, true);
    if (ones == 0) hi = base;
    else if (ones == 32) lo = base + 31;
    else return base + ones;
  }
  while (hi - lo > 32) {
    const int stride = (hi - lo + 31) / 32;                  // probes lo + stride, lo + 2 stride, ... below hi
    const int idx = lo + (sub + 1) * stride;
    const int ones = ones_at(idx, idx < hi);
    const int nlo = lo + ones * stride, nhi = lo + (ones + 1) * stride;
    lo = nlo;
    hi = nhi < hi ? nhi : hi;
  }
  return lo + 1 + ones_at(lo + 1 + sub, lo + 1 + sub < hi);   // the <= 31 indices strictly between
#endif
}

// interp1 of the n_int interval F0s of one family at time t
__device__ __forceinline__ double interp_intervals(const double *e, int n_int, double fs, double t) {
  const int lo = intervals_at_or_before(e, n_int, fs, t);
  int k = lo < 1 ? 1 : (lo > n_int - 1 ? n_int - 1 : lo);
  double x0 = interval_loc(e, k - 1, fs), x1 = interval_loc(e, k, fs);
  double y0 = interval_f0(e, k - 1, fs), y1 = interval_f0(e, k, fs);
  double sl = (t - x0) / (x1 - x0);
  return y0 + sl * (y1 - y0);
}

// The same for a workgroup that owns a run of consecutive frame times [t_first, t_last]: the counts at the
// two ends (found once, by one thread per family and end) bound every frame's count, so the intervals any
// of its frames can touch are one short contiguous range.  Their locations and F0s -- two FP64 divisions
// each, which the per-thread search above redoes at every probe -- are computed once into LDS, and each
// frame then bisects that range in LDS.  Same comparisons, same operands, same result.
struct IntervalRange {       // per family, in LDS
  int c_first, c_last;       // counts at t_first / t_last
  int j_lo, m;               // staged intervals [j_lo, j_lo + m)
};
#ifndef HV_INTERVAL_CAP
#define HV_INTERVAL_CAP 288
#endif
// staged intervals per family: a 0.256 s run holds 225 + margins at the default ceiling's 880 Hz (a run with more takes
// the unstaged route: same result, tests/test_gpu_parity.py::test_harvest_ceiling_above_the_staged_interval_capacity).
// 18.4 KB: eight workgroups per CU -- the kernel is a chain of four trips to memory, so its time is the number of
// workgroups a CU can keep waiting at once (384 intervals, six workgroups: 1.99 ms per 128 utterances; 288 with the 64
// registers of eight: 1.76)
constexpr int kIntervalCap = HV_INTERVAL_CAP;

__device__ __forceinline__ void interval_range_ends(const double *e, int n_int, double fs, double t, bool last,
                                                    IntervalRange *r) {
  const int c = intervals_at_or_before(e, n_int, fs, t);
  if (last) r->c_last = c; else r->c_first = c;
}
__device__ __forceinline__ void interval_range_close(int n_int, IntervalRange *r) {
  const int k_first = r->c_first < 1 ? 1 : (r->c_first > n_int - 1 ? n_int - 1 : r->c_first);
  const int k_last = r->c_last < 1 ? 1 : (r->c_last > n_int - 1 ? n_int - 1 : r->c_last);
  r->j_lo = k_first - 1;
  r->m = k_last - r->j_lo + 1;
}
__device__ __forceinline__ double interp_staged(const IntervalRange &r, const double *loc, const double *f0,
                                                int n_int, double t) {
  int lo = r.c_first, hi = r.c_last;               // count(t) lies between the run's ends
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (loc[mid - r.j_lo] <= t) lo = mid + 1; else hi = mid;
  }
  const int k = (lo < 1 ? 1 : (lo > n_int - 1 ? n_int - 1 : lo)) - r.j_lo;
  const double x0 = loc[k - 1], x1 = loc[k], y0 = f0[k - 1], y1 = f0[k];
  const double sl = (t - x0) / (x1 - x0);
  return y0 + sl * (y1 - y0);
}

}  // namespace world_hip
