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
};

inline size_t band_lds_bytes(int max_ntap) {
  return sizeof(double) * (size_t)((max_ntap + 1) + pad8(kTile + 2 + max_ntap + 3) + 1 + pad8(kTile + 4) + 1 + 64);
}
inline int band_segments(int n) { return (n + kSeg - 1) / kSeg; }

// sub-sample crossing time between samples e-1 and e (harvest.cpp:183-186, dio.cpp:380-382)
__device__ __forceinline__ double fine_edge(int e, double prev, double cur) { return e - prev / (cur - prev); }

// One tile: s[pad8(k)] = filtered[t0 + k] for k in [0, kTile + 2).  taps are in LDS.
// Each thread produces kOutPer consecutive outputs from a register window that
// slides one input sample per tap: per tap one LDS read of the input, one broadcast
// read of the tap and kOutPer FMAs -> FP64-FMA bound instead of LDS bound.
__device__ __forceinline__ void fir_tile(const BandJob &job, const double *taps, int t0, double *yt, double *s) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ntap = job.ntap;
  // in index for output k, tap j: t0 + k + shift - j  ->  yt[k + (ntap-1) - j],  yt[0] = in[t0 + shift - (ntap-1)]
  const int lo = t0 + job.shift - (ntap - 1);
  const int count = kTile + 2 + ntap - 1;
  __syncthreads();
  for (int k = tid; k < count; k += nt) {
    int idx = lo + k;
    yt[pad8(k)] = (idx >= 0 && idx < job.in_len) ? job.in[idx] : 0.0;
  }
  __syncthreads();
  for (int k0 = tid * kOutPer; k0 < kTile; k0 += nt * kOutPer) {
    double acc[kOutPer], w[kOutPer];
    const int base = k0 + ntap - 1;
#pragma unroll
    for (int q = 0; q < kOutPer; ++q) { acc[q] = 0.0; w[q] = yt[pad8(base + q)]; }
    // logical window at tap j: L_j[q] = in-tile[base + q - j] = w[(q - j) mod kOutPer]
    for (int j0 = 0; j0 < ntap; j0 += kOutPer) {
#pragma unroll
      for (int u = 0; u < kOutPer; ++u) {
        const int j = j0 + u;
        if (j < ntap) {
          if (j > 0) w[(kOutPer - u) % kOutPer] = yt[pad8(base - j)];
          const double h = taps[j];
#pragma unroll
          for (int q = 0; q < kOutPer; ++q) acc[q] = fma(h, w[(q + kOutPer - u) % kOutPer], acc[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kOutPer; ++q) s[pad8(k0 + q)] = acc[q];
  }
  for (int e = tid; e < 2; e += nt) {              // the two look-ahead samples
    double acc = 0.0;
    const int base = kTile + e + (ntap - 1);
    for (int j = 0; j < ntap; ++j) acc = fma(taps[j], yt[pad8(base - j)], acc);
    s[pad8(kTile + e)] = acc;
  }
  __syncthreads();
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
  double *taps = reinterpret_cast<double *>(lds);
  double *yt = taps + (job.max_ntap + 1);
  double *s = yt + pad8(kTile + 2 + job.max_ntap + 3) + 1;
  double *scratch = s + pad8(kTile + 4) + 1;
  for (int j = tid; j < job.ntap; j += nt) taps[j] = job.taps[j];

  double *ev = job.seg_events + (size_t)seg * kSegCap;
  const size_t fam_stride = (size_t)job.nseg * kSegCap;
  int count[4] = {0, 0, 0, 0};
  for (int t0 = seg_begin; t0 < seg_end; t0 += kTile) {
    fir_tile(job, taps, t0, yt, s);
    // events: each thread inspects kPer consecutive samples, in time order
    constexpr int kPer = 4;
    for (int fam = 0; fam < 4; ++fam) {
      double *dst = ev + fam * fam_stride;
      for (int sub = 0; sub < kTile; sub += nt * kPer) {
        double found[kPer];
        int nfound = 0;
        for (int q = 0; q < kPer; ++q) {
          int k = sub + tid * kPer + q;
          int i = t0 + k;
          if (k >= kTile) break;
          double a, b;                   // the family's signal at i and i+1
          bool in_range;
          const double s0 = s[pad8(k)], s1 = s[pad8(k + 1)];
          if (fam < 2) { a = s0; b = s1; in_range = i <= n - 2; }
          else { const double s2 = s[pad8(k + 2)]; a = s1 - s0; b = s2 - s1; in_range = i <= n - 3; }
          bool hit = fam % 2 == 0 ? (0.0 < a && b <= 0.0) : (a < 0.0 && 0.0 <= b);
          if (in_range && hit) found[nfound++] = fine_edge(i + 1, a, b);
        }
        int total, off = block_excl_scan_int(nfound, &total, scratch);
        for (int q = 0; q < nfound; ++q)
          if (count[fam] + off + q < kSegCap) dst[count[fam] + off + q] = found[q];
        count[fam] += total;
      }
    }
  }
  if (tid == 0)
    for (int fam = 0; fam < 4; ++fam) cnt_out[(size_t)fam * job.nseg] = imin(count[fam], kSegCap);
}

// concatenate the segment lists of one (channel, family) in time order
__device__ __forceinline__ void compact_event_segments(const double *seg_events, const int *seg_count, int nseg,
                                                       double *events, int ev_cap, int *ev_count) {
  int base = 0;
  for (int sgm = 0; sgm < nseg; ++sgm) {
    const int c = seg_count[sgm];
    for (int i = threadIdx.x; i < c; i += blockDim.x)
      if (base + i < ev_cap) events[base + i] = seg_events[(size_t)sgm * kSegCap + i];
    base += c;
  }
  if (threadIdx.x == 0) *ev_count = imin(base, ev_cap);
}

// ---- interval F0s of one family and their interpolation onto frame times ----------
__device__ __forceinline__ double interval_loc(const double *e, int k, double fs) { return (e[k] + e[k + 1]) / 2.0 / fs; }
__device__ __forceinline__ double interval_f0(const double *e, int k, double fs) { return fs / (e[k + 1] - e[k]); }

// interp1 (matlabfunctions.cpp:136-176) of the n_int intervals of one family at time t:
// the bin is clamp(#{locations <= t}, 1, n-1), so both ends extrapolate linearly.
__device__ __forceinline__ double interp_intervals(const double *e, int n_int, double fs, double t) {
  int lo = 0, hi = n_int;                       // count of locations <= t
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (interval_loc(e, mid, fs) <= t) lo = mid + 1; else hi = mid;
  }
  int k = lo < 1 ? 1 : (lo > n_int - 1 ? n_int - 1 : lo);
  double x0 = interval_loc(e, k - 1, fs), x1 = interval_loc(e, k, fs);
  double y0 = interval_f0(e, k - 1, fs), y1 = interval_f0(e, k, fs);
  double sl = (t - x0) / (x1 - x0);
  return y0 + sl * (y1 - y0);
}

}  // namespace world_hip
