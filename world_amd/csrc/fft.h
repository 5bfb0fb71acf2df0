// fft.h -- block-cooperative FP64 FFTs staged in LDS (replaces the reference's
// fft_plan_dft_r2c_1d / fft_plan_dft_c2r_1d / fft_execute, src/fft.cpp:26-162,
// for every per-frame transform on the analysis path).
//
// One workgroup owns one transform held entirely in LDS as interleaved complex
// doubles.  Forward transforms are decimation-in-frequency (natural order in,
// bit-reversed order out); inverse transforms are decimation-in-time
// (bit-reversed in, natural out).  Nobody ever permutes: consumers of a forward
// transform read bin k at LDS slot brev(k), producers of an inverse transform
// write bin k to slot brev(k).  Radix-4 butterflies (one barrier per two
// levels) with a radix-2 clean-up level when log2 is odd.
//
// Real transforms of length N run as N/2-point complex transforms with the
// usual split/merge step, fused into a caller-supplied functor so |X|^2,
// products of two spectra, lifters ... are formed straight from registers.
#pragma once
#include "devrt.h"
#include "tables.h"

namespace world_hip {

struct cplx { double re, im; };

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  cplx r; r.re = a.re * b.re - a.im * b.im; r.im = a.re * b.im + a.im * b.re; return r;
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { cplx r; r.re = a.re + b.re; r.im = a.im + b.im; return r; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { cplx r; r.re = a.re - b.re; r.im = a.im - b.im; return r; }
__device__ __forceinline__ cplx cconj(cplx a) { a.im = -a.im; return a; }
// Twiddles come from a quarter-wave cosine table staged in LDS by the owning kernel:
// q[r] = cos(2 pi r / 2^lg), r = 0 .. 2^lg/4.  (A butterfly needs three twiddles; from
// the HBM table each was an L2-latency gather on the critical path between barriers.)
struct TwLds { const double *q; int lg; };

// stage the table for transforms up to 2^lg points; call before the first transform
__device__ __forceinline__ TwLds stage_twiddles(double *q, int lg, const double2 *global_tw) {
  const int quarter = 1 << (lg - 2);
  for (int i = threadIdx.x; i <= quarter; i += blockDim.x) q[i] = global_tw[(size_t)i << (kTwLog2 - lg)].x;
  __syncthreads();
  TwLds t; t.q = q; t.lg = lg; return t;
}
__device__ __forceinline__ size_t twiddle_lds_doubles(int lg) { return (size_t)(1 << (lg - 2)) + 2; }

// e^{-2 pi i k / 2^lg} (forward, sign=-1) or its conjugate (sign=+1), 0 <= k < 2^lg
__device__ __forceinline__ cplx twiddle(const TwLds &tw, int k, int lg, int sign) {
  const int K = k << (tw.lg - lg);
  const int quarter = 1 << (tw.lg - 2);
  const int quad = K >> (tw.lg - 2), r = K & (quarter - 1);
  const double a = tw.q[r], b = tw.q[quarter - r];
  double c, s;
  if (quad == 0) { c = a; s = b; }
  else if (quad == 1) { c = -b; s = a; }
  else if (quad == 2) { c = -a; s = -b; }
  else { c = b; s = -a; }
  cplx w; w.re = c; w.im = sign > 0 ? s : -s; return w;
}
__device__ __forceinline__ int brev_bits(int k, int bits) { return (int)(__brev((unsigned)k) >> (32 - bits)); }

// ---- forward: natural in -> bit-reversed out --------------------------------
__device__ __forceinline__ void block_cfft_dif(cplx *z, int lg, const TwLds &tw) {
  int n = 1 << lg;
  int lev = lg;                       // current sub-transform length = 2^lev
  while (lev >= 2) {
    int q = 1 << (lev - 2);
    __syncthreads();
    for (int b = threadIdx.x; b < n / 4; b += blockDim.x) {
      int j = b & (q - 1);
      int base = ((b >> (lev - 2)) << lev) + j;
      cplx a0 = z[base], a1 = z[base + q], a2 = z[base + 2 * q], a3 = z[base + 3 * q];
      cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
      cplx t3; t3.re = d.im; t3.im = -d.re;                 // (a1-a3) * (-i)
      cplx w1 = twiddle(tw, j, lev, -1), w2 = twiddle(tw, 2 * j, lev, -1), w3 = twiddle(tw, 3 * j, lev, -1);
      z[base] = cadd(t0, t2);
      z[base + q] = cmul(csub(t0, t2), w2);
      z[base + 2 * q] = cmul(cadd(t1, t3), w1);
      z[base + 3 * q] = cmul(csub(t1, t3), w3);
    }
    lev -= 2;
  }
  if (lev == 1) {
    __syncthreads();
    for (int b = threadIdx.x; b < n / 2; b += blockDim.x) {
      cplx a = z[2 * b], c = z[2 * b + 1];
      z[2 * b] = cadd(a, c);
      z[2 * b + 1] = csub(a, c);
    }
  }
  __syncthreads();
}

// ---- inverse (unscaled): bit-reversed in -> natural out ----------------------
__device__ __forceinline__ void block_cfft_dit(cplx *z, int lg, const TwLds &tw) {
  int n = 1 << lg;
  int lev = 0;                        // sub-transforms of length 2^lev are done
  if (lg & 1) {
    __syncthreads();
    for (int b = threadIdx.x; b < n / 2; b += blockDim.x) {
      cplx a = z[2 * b], c = z[2 * b + 1];
      z[2 * b] = cadd(a, c);
      z[2 * b + 1] = csub(a, c);
    }
    lev = 1;
  }
  while (lev < lg) {
    int q = 1 << lev;
    int L = lev + 2;                  // resulting length 2^L
    __syncthreads();
    for (int b = threadIdx.x; b < n / 4; b += blockDim.x) {
      int j = b & (q - 1);
      int base = ((b >> lev) << L) + j;
      cplx b0 = z[base], b1 = z[base + q], b2 = z[base + 2 * q], b3 = z[base + 3 * q];
      cplx w1 = twiddle(tw, j, L, +1), w2 = twiddle(tw, 2 * j, L, +1);
      cplx x1 = cmul(b1, w2), x3 = cmul(b3, w2);
      cplx p0 = cadd(b0, x1), p1 = csub(b0, x1), p2 = cadd(b2, x3), p3 = csub(b2, x3);
      cplx y2 = cmul(p2, w1), y3 = cmul(p3, w1);
      cplx iy3; iy3.re = -y3.im; iy3.im = y3.re;            // (+i) * y3
      z[base] = cadd(p0, y2);
      z[base + 2 * q] = csub(p0, y2);
      z[base + q] = cadd(p1, iy3);
      z[base + 3 * q] = csub(p1, iy3);
    }
    lev += 2;
  }
  __syncthreads();
}

// ---- real forward transform of length N = 2^lgn -------------------------------
// `z` holds the N real samples (as N/2 complex slots, z[m] = x[2m] + i x[2m+1]).
// After the call the buffer holds scrambled data; emit(k, Xre, Xim) has been
// called once for every k in [0, N/2] (same semantics as the reference's r2c:
// X[k] = sum x[n] e^{-2 pi i k n / N}, imaginary part of DC/Nyquist = 0).
template <class Emit>
__device__ __forceinline__ void block_rfft(cplx *z, int lgn, const TwLds &tw, Emit emit) {
  int lgh = lgn - 1, h = 1 << lgh;
  block_cfft_dif(z, lgh, tw);
  for (int k = threadIdx.x; k <= h; k += blockDim.x) {
    int ka = k & (h - 1), kb = (h - k) & (h - 1);
    cplx za = z[brev_bits(ka, lgh)], zb = z[brev_bits(kb, lgh)];
    cplx e, o;                                         // even / odd sub-spectra
    e.re = 0.5 * (za.re + zb.re); e.im = 0.5 * (za.im - zb.im);
    o.re = 0.5 * (za.im + zb.im); o.im = -0.5 * (za.re - zb.re);
    cplx w;
    if (k < h) w = twiddle(tw, k, lgn, -1); else { w.re = -1.0; w.im = 0.0; }
    cplx ow = cmul(o, w);
    double xr = e.re + ow.re, xi = e.im + ow.im;
    if (k == 0 || k == h) xi = 0.0;
    emit(k, xr, xi);
  }
  __syncthreads();
}

// ---- real inverse transform (unscaled: N * irfft, like the reference's c2r) ----
// spec(k) returns X[k] for k in [0, N/2] (the imaginary part of DC and Nyquist
// is ignored, src/fft.cpp:28-29).  On return the N real outputs are in `z`
// viewed as doubles (out[n] = reinterpret_cast<double*>(z)[n]).
template <class Spec>
__device__ __forceinline__ void block_irfft(cplx *z, int lgn, const TwLds &tw, Spec spec) {
  int lgh = lgn - 1, h = 1 << lgh;
  __syncthreads();
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    cplx x = spec(k), y = spec(h - k);
    if (k == 0) { x.im = 0.0; y.im = 0.0; }
    y.im = -y.im;                                       // conj(X[h-k])
    cplx s = cadd(x, y), d = csub(x, y);
    cplx w = twiddle(tw, k, lgn, +1);
    cplx t = cmul(d, w);
    cplx r; r.re = s.re - t.im; r.im = s.im + t.re;     // s + i*w*d
    z[brev_bits(k, lgh)] = r;
  }
  block_cfft_dit(z, lgh, tw);
}

}  // namespace world_hip
