// decimate.h -- MATLAB-style decimate() (reference src/matlabfunctions.cpp:27-204)
// as two chunk-parallel IIR sweeps.
//
// The reference runs a 3rd-order IIR forwards over the (reflect-padded) signal,
// reverses, runs it again and reverses back: two strictly serial recurrences.
// The filter's poles are well inside the unit circle (|p| <= 0.889 for r = 12),
// so a sweep restarted from zero state kWarm samples before a chunk has forgotten
// its start to < 1e-26 by the time it reaches the chunk.  Every
// thread owns one chunk; the first chunk starts at sample 0 exactly like the
// reference.
#pragma once
#include "common.h"

namespace world_hip {

struct IirCoef { double a0, a1, a2, b0, b1; };

// 128 threads x 32 outputs: a workgroup stages 4096 + 384 samples = 37 KB of LDS.  (Rounds 1-4: 256 threads, 71 KB.  In the
// twelve-jobs-in-flight mode the CUs are full of d4c_frame workgroups -- three per CU, 157 of the 160 KB -- and a workgroup
// that needs more LDS than ONE of them frees waits for two to retire on the same CU before it is refilled: the sweeps took
// 53 and 77 us in flight against 21 and 18 alone, profiles/r05/inflight_overlap.txt.  Every narrow kernel of a job now
// fits the 52 KB one retiring frame workgroup leaves behind.)
constexpr int kDecThreads = 128;
constexpr int kDecChunk = 32;                         // outputs per thread
constexpr int kDecSpan = kDecThreads * kDecChunk;     // outputs per workgroup
constexpr int kDecWarm = 384;     // longest warm-up (LDS is carved for it): 0.889^384 = 2e-20 at r = 12
// warm-up actually run: the pole radius is <= 0.7985 for r <= 6, and 0.7985^160 = 2e-16
__host__ __device__ __forceinline__ int dec_warm(int r) { return r <= 6 ? 160 : kDecWarm; }
constexpr int kDecBatch = 8;      // samples fetched together ahead of the serial recurrence
constexpr int kDecPad = 9;        // kNFact
// threads read the staged span with a lane stride of kDecChunk doubles: one pad slot per
// kDecChunk entries makes that stride odd (33) -> bank-conflict free
__host__ __device__ __forceinline__ int dec_pad(int k) { return k + k / kDecChunk; }
constexpr int kDecStage = kDecSpan + kDecWarm + (kDecSpan + kDecWarm) / kDecChunk + 8;   // doubles of the staged span (dec_pad of its end)
inline size_t dec_lds_bytes() { return sizeof(double) * (size_t)(kDecStage + 64); }          // + a block collective's scratch

// filter coefficients, src/matlabfunctions.cpp:29-113
inline IirCoef decimate_coef(int r) {
  static const double t[11][5] = {
      {0.041156734567757189, -0.42599112459189636, 0.041037215479961225, 0.16797464681802227, 0.50392394045406674},
      {0.95039378983237421, -0.67429146741526791, 0.15412211621346475, 0.071221945171178636, 0.21366583551353591},
      {1.4499664446880227, -0.98943497080950582, 0.24578252340690215, 0.036710750339322612, 0.11013225101796784},
      {1.7610939654280557, -1.2554914843859768, 0.3237186507788215, 0.021334858522387423, 0.06400457556716227},
      {1.9715352749512141, -1.4686795689225347, 0.3893908434965701, 0.013469181309343825, 0.040407543928031475},
      {2.1225239019534703, -1.6395144861046302, 0.44469707800587366, 0.0090366882681608418, 0.027110064804482525},
      {2.2357462340187593, -1.7780899984041358, 0.49152555365968692, 0.0063522763407111993, 0.019056829022133598},
      {2.3236003491759578, -1.8921545617463598, 0.53148928133729068, 0.0046331164041389372, 0.013899349212416812},
      {2.3936475118069387, -1.9873904075111861, 0.5658879979027055, 0.0034818622251927556, 0.010445586675578267},
      {2.450743295230728, -2.06794904601978, 0.59574774438332101, 0.0026822508007163792, 0.0080467524021491377},
      {2.4981398605924205, -2.1368928194784025, 0.62187513816221485, 0.0021097275904709001, 0.0063291827714127002}};
  IirCoef c = {0, 0, 0, 0, 0};
  if (r >= 2 && r <= 12) { c.a0 = t[r - 2][0]; c.a1 = t[r - 2][1]; c.a2 = t[r - 2][2]; c.b0 = t[r - 2][3]; c.b1 = t[r - 2][4]; }
  return c;
}

// The signal the reference feeds to FilterForDecimate is x edge-padded by `lag` on both sides (Harvest,
// harvest.cpp:50-59; lag = 0 for DIO) -- px(i) = x[clamp(i - lag)], m = n + 2 lag samples -- and then reflect-padded
// by kNFact (matlabfunctions.cpp:183-186): padded[j] = 2 px(0) - px(9 - j) for j < 9, px(j - 9) in the middle and
// 2 px(m-1) - px(m - 2 - (j - (9 + m))) beyond.  dec_forward_block() builds it while staging.

// One step of the 3rd-order recurrence w[n] = in + a0 w[n-1] + a1 w[n-2] + a2 w[n-3] (FilterForDecimate,
// matlabfunctions.cpp:115-125).  The terms that do not involve w[n-1] are summed first, so the
// dependent chain from one step to the next is a single FMA (a dependent FP64 op issues every 36
// cycles on gfx950; left-to-right evaluation would put four of them on the chain).  The different
// association moves the filtered signal by ~1e-16 relative.
__device__ __forceinline__ void iir_advance(const IirCoef &c, double in, double &w0, double &w1, double &w2) {
  const double t = fma(c.a2, w2, fma(c.a1, w1, in));
  const double wt = fma(c.a0, w0, t);
  w2 = w1; w1 = w0; w0 = wt;
}
// the filter's four taps on the recursion's state: wt = w[n], then w[n-1], w[n-2], w[n-3]
__device__ __forceinline__ double iir_taps(const IirCoef &c, double wt, double w0, double w1, double w2) {
  return c.b0 * wt + c.b1 * w0 + c.b1 * w1 + c.b0 * w2;
}
// A wavefront issues in order, so the taps (three dependent additions) written between two steps of the recursion
// would hold the next step back by their own latency.  Both sweeps therefore run the recursion alone over the
// warm-up (whose outputs nobody needs) and over the chunk, keeping the chunk's states in registers, and evaluate the
// taps afterwards as kDecChunk independent expressions.

// forward sweep over the padded signal of length n + 2*lag + 18: one workgroup
// stages its span (+ warm-up) in LDS with coalesced loads, then every thread runs
// the recurrence over its warm-up and its chunk out of LDS.
__device__ __forceinline__ void dec_forward_block(const double *x, int n, int lag, IirCoef c, int block, double *fwd,
                                                  double *stage, int warm) {
  const int total = n + 2 * lag + 2 * kDecPad;
  const int b0 = block * kDecSpan;
  if (b0 >= total) return;
  const int lo = b0 - warm;                              // stage[k] = padded[lo + k]
  const int cnt = imin(total, b0 + kDecSpan) - lo;
  {
    // kDecBatch loads in flight per thread (one load waited for per trip made staging the longest part of the kernel);
    // the padded signal (see above) without branches: every sample is x[src] or, in the reflected edges, 2 * end - x[src]
    const int m = n + 2 * lag, nt = (int)blockDim.x;
    const double head = x[0], tail = x[n - 1];
    for (int k0 = threadIdx.x; k0 < cnt; k0 += kDecBatch * nt) {
      double v[kDecBatch];
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) {
        const int j = imax(0, imin(total - 1, lo + k0 + q * nt));
        const int i = j < kDecPad ? kDecPad - j : (j >= kDecPad + m ? m - 2 - (j - (kDecPad + m)) : j - kDecPad);
        v[q] = x[imin(n - 1, imax(0, i - lag))];
      }
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) {
        const int k = k0 + q * nt, j = lo + k;
        if (k < cnt) stage[dec_pad(k)] = j < 0 ? 0.0 : (j < kDecPad ? 2 * head - v[q] : (j >= kDecPad + m ? 2 * tail - v[q] : v[q]));
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kDecThreads; t += blockDim.x) {
    const int c0 = b0 + t * kDecChunk;
    if (c0 >= total) break;
    const int c1 = imin(total, c0 + kDecChunk);
    double w0 = 0, w1 = 0, w2 = 0;
    // c0 and the warm-up lengths are multiples of kDecBatch: the warm-up runs in whole batches
    for (int j0 = imax(0, c0 - warm); j0 < c0; j0 += kDecBatch) {
      double v[kDecBatch];
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) v[q] = stage[dec_pad(j0 + q - lo)];
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) iir_advance(c, v[q], w0, w1, w2);
    }
    double w[kDecChunk + 3];
    w[0] = w2; w[1] = w1; w[2] = w0;
#pragma unroll
    for (int q = 0; q < kDecChunk; ++q) {
      const double v = stage[dec_pad(imin(c0 + q, c1 - 1) - lo)];
      iir_advance(c, v, w0, w1, w2);
      w[q + 3] = keep(w0);           // pinned here: the taps below must not pull the recursion down to them
    }
    sched_fence();
#pragma unroll
    for (int q = 0; q < kDecChunk; ++q)
      if (c0 + q < c1) fwd[c0 + q] = iir_taps(c, w[q + 3], w[q + 2], w[q + 1], w[q]);
  }
}

// backward sweep over fwd; every r-th output starting at `first` is a decimated sample.
// out[k] = decimated[skip + k] for k < out_len   (matlabfunctions.cpp:195-200, harvest.cpp:62)
// span_sum (optional): the sum of the samples this workgroup stored goes to span_sum[block] -- the mean the caller
// removes next is then a sum of per-span partials in span order (deterministic), and no pass of its own over `out`.
__device__ __forceinline__ void dec_backward_block(const double *fwd, int n, int lag, int r, IirCoef c, int block,
                                                   int skip, int out_len, double *out, double *stage, int warm,
                                                   double *span_sum = nullptr) {
  const int total = n + 2 * lag + 2 * kDecPad;
  const int m = n + 2 * lag;
  const int nout = (m - 1) / r + 1;
  const int nbeg = r - r * nout + m;
  const int first = nbeg + kDecPad - 1;         // index (in padded coordinates) of decimated[0]
  const int b0 = block * kDecSpan;
  if (b0 >= total) {                            // (the whole workgroup: a span beyond this utterance's end)
    if (span_sum && threadIdx.x == 0) span_sum[block] = 0.0;
    return;
  }
  double acc = 0.0;
  const int hi = imin(total, b0 + kDecSpan + warm);      // stage[k] = fwd[b0 + k], k < hi - b0
  {
    const int cnt = hi - b0, nt = (int)blockDim.x;
    for (int k0 = threadIdx.x; k0 < cnt; k0 += kDecBatch * nt) {      // kDecBatch loads in flight per thread
      double v[kDecBatch];
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) v[q] = fwd[b0 + imin(cnt - 1, k0 + q * nt)];
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) if (k0 + q * nt < cnt) stage[dec_pad(k0 + q * nt)] = v[q];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kDecThreads; t += blockDim.x) {
    const int c0 = b0 + t * kDecChunk;
    if (c0 >= total) break;
    const int c1 = imin(total, c0 + kDecChunk);
    double w0 = 0, w1 = 0, w2 = 0;
    // warm-up from the far end down to c1: the signal's end cuts it to any length, so the odd steps go first
    int j = imin(total - 1, c1 - 1 + warm);
    for (int odd = (j - c1 + 1) % kDecBatch; odd > 0; --odd, --j) iir_advance(c, stage[dec_pad(j - b0)], w0, w1, w2);
    for (; j >= c1; j -= kDecBatch) {
      double v[kDecBatch];
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) v[q] = stage[dec_pad(j - q - b0)];
#pragma unroll
      for (int q = 0; q < kDecBatch; ++q) iir_advance(c, v[q], w0, w1, w2);
    }
    double w[kDecChunk + 3];                      // w[q + 3] = state after the step at sample c1 - 1 - q
    w[0] = w2; w[1] = w1; w[2] = w0;
#pragma unroll
    for (int q = 0; q < kDecChunk; ++q) {
      const double v = stage[dec_pad(imax(c1 - 1 - q, c0) - b0)];
      iir_advance(c, v, w0, w1, w2);
      w[q + 3] = keep(w0);           // pinned here: the taps below must not pull the recursion down to them
    }
    // the decimated samples among c1-1 .. c0: d = sample - first runs down from d0; its remainder and quotient by r
    // are carried instead of divided out 32 times
    sched_fence();
    const int d0 = c1 - 1 - first;
    if (d0 < 0) continue;
    int rem = d0 % r, quo = d0 / r;
#pragma unroll
    for (int q = 0; q < kDecChunk; ++q) {
      const int d = d0 - q;
      if (c1 - 1 - q >= c0 && d >= 0 && rem == 0 && nbeg + d < m + kDecPad) {   // loop bound of matlabfunctions.cpp:199
        const int k = quo - skip;
        if (k >= 0 && k < out_len) {
          const double v = iir_taps(c, w[q + 3], w[q + 2], w[q + 1], w[q]);
          out[k] = v;
          acc += v;
        }
      }
      if (rem == 0) { rem = r - 1; --quo; } else { --rem; }
    }
  }
  if (span_sum) {
    acc = block_sum(acc, stage + kDecStage);     // (scratch behind the staged span; every thread of the workgroup arrives)
    if (threadIdx.x == 0) span_sum[block] = acc;
  }
}

}  // namespace world_hip
