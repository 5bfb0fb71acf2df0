/*
 * world_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C99 restatement of the WORLD analysis path.  Every routine cites the
 * reference lines (relative to /root/reference) whose behaviour it restates.
 * The arithmetic follows the reference's evaluation order wherever the result
 * is order-sensitive (serial prefix sums, the RNG stream, truncating rounds);
 * the FFT is an independent radix-2 implementation (the reference's in-tree
 * Ooura FFT differs from it by O(1e-16) rounding only).
 *
 * Build: see oracle/Makefile (gcc -std=c99 -O2 -ffp-contract=off).
 * Pinned by: tests/test_oracle.py (against the unmodified reference in oracle/_ref, against the
 * tests/golden/ fixtures generated from it, and against SURVEY.md 8c's checksums).
 */
#define _USE_MATH_DEFINES
#define _DEFAULT_SOURCE
#include "world_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* constantnumbers.h:8-37 */
static const double K_PI = 3.1415926535897932384;
static const double K_TINY = 0.000000000001;
static const double K_EPS = 0.00000000000000022204460492503131;
static const double K_LOG2 = 0.69314718055994529;
static const double K_BIG = 100000.0;

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }
static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }
static int *ialloc(size_t n) { return (int *)calloc(n ? n : 1, sizeof(int)); }

/* ------------------------------------------------------------------ */
/* scalar helpers                                                      */
/* ------------------------------------------------------------------ */
int wo_round(double v) { return v > 0 ? (int)(v + 0.5) : (int)(v - 0.5); }

int wo_pow2_above(int n) {
  return (int)pow(2.0, (int)(log((double)n) / K_LOG2) + 1.0);
}

void wo_randn_seed(uint32_t s[4]) {
  s[0] = 123456789u; s[1] = 362436069u; s[2] = 521288629u; s[3] = 88675123u;
}

static uint32_t xorshift_step(uint32_t s[4]) {
  uint32_t t = s[0] ^ (s[0] << 11);
  s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
  s[3] = (s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8));
  return s[3];
}

double wo_randn(uint32_t s[4]) {
  uint32_t acc = 0;
  for (int k = 0; k < 12; ++k) acc += xorshift_step(s) >> 4;
  return acc / 268435456.0 - 6.0;
}

/* ------------------------------------------------------------------ */
/* FFT: radix-2 complex core + real wrappers (semantics of fft.cpp:26-60: */
/* forward = rfft, backward = N * irfft, Im of DC/Nyquist ignored)      */
/* ------------------------------------------------------------------ */
#define WO_MAX_LOG2 22
static double *g_tw[WO_MAX_LOG2 + 1]; /* g_tw[L]: cos/sin(2*pi*k/2^L), k < 2^L/2 */

static const double *twiddles(int log2n) {
  if (!g_tw[log2n]) {
    int n = 1 << log2n, h = n / 2 > 0 ? n / 2 : 1;
    double *t = (double *)malloc(sizeof(double) * 2 * h);
    for (int k = 0; k < h; ++k) {
      double a = 2.0 * 3.14159265358979323846 * k / n;
      t[2 * k] = cos(a);
      t[2 * k + 1] = sin(a);
    }
    g_tw[log2n] = t;
  }
  return g_tw[log2n];
}

static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

/* in-place complex FFT, sign = -1 forward / +1 backward, unscaled */
static void cfft(int n, double *re, double *im, int sign) {
  int lg = ilog2(n);
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  const double *tw = twiddles(lg);
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for (int base = 0; base < n; base += len)
      for (int k = 0; k < half; ++k) {
        double wr = tw[2 * k * step], wi = sign * tw[2 * k * step + 1];
        int a = base + k, b = a + half;
        double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr; im[b] = im[a] - xi;
        re[a] += xr; im[a] += xi;
      }
  }
}

void wo_rfft(int n, const double *in, double *re, double *im) {
  if (n < 4) { /* tiny sizes never occur on the path; direct DFT */
    for (int k = 0; k <= n / 2; ++k) {
      double sr = 0, si = 0;
      for (int t = 0; t < n; ++t) {
        sr += in[t] * cos(2 * 3.14159265358979323846 * k * t / n);
        si -= in[t] * sin(2 * 3.14159265358979323846 * k * t / n);
      }
      re[k] = sr; im[k] = si;
    }
    return;
  }
  int h = n / 2;
  double *zr = (double *)malloc(sizeof(double) * 2 * h), *zi = zr + h;
  for (int k = 0; k < h; ++k) { zr[k] = in[2 * k]; zi[k] = in[2 * k + 1]; }
  cfft(h, zr, zi, -1);
  const double *tw = twiddles(ilog2(n));
  for (int k = 0; k <= h; ++k) {
    int a = k % h, b = (h - k) % h;
    double er = 0.5 * (zr[a] + zr[b]), ei = 0.5 * (zi[a] - zi[b]);
    double orr = 0.5 * (zi[a] + zi[b]), oi = -0.5 * (zr[a] - zr[b]);
    double wr, wi;
    if (k < h) { wr = tw[2 * k]; wi = -tw[2 * k + 1]; } else { wr = -1.0; wi = 0.0; }
    re[k] = er + (orr * wr - oi * wi);
    im[k] = ei + (orr * wi + oi * wr);
  }
  im[0] = 0.0; im[h] = 0.0;
  free(zr);
}

void wo_irfft_unscaled(int n, const double *re, const double *im, double *out) {
  int h = n / 2;
  double *zr = (double *)malloc(sizeof(double) * 2 * h), *zi = zr + h;
  const double *tw = twiddles(ilog2(n));
  for (int k = 0; k < h; ++k) {
    int b = h - k;
    double xr = re[k], xi = (k == 0) ? 0.0 : im[k];
    double yr = re[b], yi = (b == h) ? 0.0 : -im[b];   /* conj(X[h-k]) */
    double sr = xr + yr, si = xi + yi;
    double dr = xr - yr, di = xi - yi;
    double wr = tw[2 * k], wi = tw[2 * k + 1];          /* e^{+2 pi i k/n} */
    /* i * w * d */
    double tr = -(dr * wi + di * wr), ti = dr * wr - di * wi;
    zr[k] = sr + tr; zi[k] = si + ti;
  }
  cfft(h, zr, zi, +1);
  for (int k = 0; k < h; ++k) { out[2 * k] = zr[k]; out[2 * k + 1] = zi[k]; }
  free(zr);
}

/* ------------------------------------------------------------------ */
/* shared DSP helpers                                                  */
/* ------------------------------------------------------------------ */
void wo_nuttall(int len, double *w) {
  for (int i = 0; i < len; ++i) {
    double t = i / (len - 1.0);
    w[i] = 0.355768 - 0.487396 * cos(2.0 * K_PI * t) +
           0.144232 * cos(4.0 * K_PI * t) - 0.012604 * cos(6.0 * K_PI * t);
  }
}

/* interp1 + histc (matlabfunctions.cpp:136-176): the bin of a query is
 * clamp(#{knots <= query}, 1, n-1), so both ends extrapolate linearly. */
void wo_interp1(const double *x, const double *y, int n, const double *xi,
                int ni, double *yi) {
  int c = 0; /* knots are ascending and so are the queries on every call site */
  for (int i = 0; i < ni; ++i) {
    while (c < n && x[c] <= xi[i]) ++c;
    int k = c < 1 ? 1 : (c > n - 1 ? n - 1 : c);
    double h = x[k] - x[k - 1];
    double s = (xi[i] - x[k - 1]) / h;
    yi[i] = y[k - 1] + s * (y[k] - y[k - 1]);
  }
}

void wo_interp1q(double x0, double dx, const double *y, int n,
                 const double *xi, int ni, double *yi) {
  for (int i = 0; i < ni; ++i) {
    int b = (int)((xi[i] - x0) / dx);
    double frac = (xi[i] - x0) / dx - b;
    double dy = (b < n - 1) ? y[b + 1] - y[b] : 0.0;
    yi[i] = y[b] + dy * frac;
  }
}

/* decimation IIR coefficients, matlabfunctions.cpp:27-113 */
static void decimate_coeffs(int r, double a[3], double b[2]) {
  static const double tab[11][5] = {
      /* r = 2 .. 12 : a0 a1 a2 b0 b1 */
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
  if (r < 2 || r > 12) { a[0] = a[1] = a[2] = b[0] = b[1] = 0.0; return; }
  a[0] = tab[r - 2][0]; a[1] = tab[r - 2][1]; a[2] = tab[r - 2][2];
  b[0] = tab[r - 2][3]; b[1] = tab[r - 2][4];
}

static void decimate_iir(const double *x, int n, int r, double *y) {
  double a[3], b[2], w0 = 0, w1 = 0, w2 = 0;
  decimate_coeffs(r, a, b);
  for (int i = 0; i < n; ++i) {
    double wt = x[i] + a[0] * w0 + a[1] * w1 + a[2] * w2;
    y[i] = b[0] * wt + b[1] * w0 + b[1] * w1 + b[0] * w2;
    w2 = w1; w1 = w0; w0 = wt;
  }
}

/* matlabfunctions.cpp:178-204 */
void wo_decimate(const double *x, int n, int r, double *y) {
  const int pad = 9;
  int m = n + 2 * pad;
  double *t1 = dalloc(m), *t2 = dalloc(m);
  for (int i = 0; i < pad; ++i) t1[i] = 2 * x[0] - x[pad - i];
  for (int i = 0; i < n; ++i) t1[pad + i] = x[i];
  for (int i = 0; i < pad; ++i) t1[pad + n + i] = 2 * x[n - 1] - x[n - 2 - i];
  decimate_iir(t1, m, r, t2);
  for (int i = 0; i < m; ++i) t1[i] = t2[m - 1 - i];
  decimate_iir(t1, m, r, t2);
  for (int i = 0; i < m; ++i) t1[i] = t2[m - 1 - i];
  int nout = (n - 1) / r + 1;
  int nbeg = r - r * nout + n;
  int c = 0;
  for (int i = nbeg; i < n + pad; i += r) y[c++] = t1[i + pad - 1];
  free(t1); free(t2);
}

/* common.cpp:56-75 (in == out allowed) */
void wo_dc_correction(const double *in, double f0, int fs, int fft_size, double *out) {
  int upper = 2 + (int)(f0 * fft_size / fs);
  int nrep = upper - 1;
  double *axis = dalloc(upper), *rep = dalloc(upper);
  for (int i = 0; i < upper; ++i) axis[i] = (double)i * fs / fft_size;
  wo_interp1q(f0 - axis[0], -(double)fs / fft_size, in, upper + 1, axis, nrep, rep);
  for (int i = 0; i < nrep; ++i) out[i] = in[i] + rep[i];
  if (out != in) for (int i = nrep; i <= fft_size / 2; ++i) out[i] = in[i];
  free(axis); free(rep);
}

/* common.cpp:27-46,77-111: rectangular smoothing through a SERIAL prefix sum
 * of the mirrored spectrum (the summation order is part of the result). */
void wo_linear_smoothing(const double *in, double width, int fs, int fft_size, double *out) {
  int half = fft_size / 2;
  int bnd = (int)(width * fft_size / fs) + 1;
  int len = half + 2 * bnd + 1;
  double *seg = dalloc(len), *axis = dalloc(half + 1);
  double *lo = dalloc(half + 1), *hi = dalloc(half + 1);
  for (int i = 0; i < len; ++i) {
    double m;
    if (i < bnd) m = in[bnd - i];
    else if (i < half + bnd) m = in[i - bnd];
    else m = in[half - (i - (half + bnd))];
    seg[i] = m * fs / fft_size + (i ? seg[i - 1] : 0.0);
    if (i == 0) seg[0] = m * fs / fft_size;
  }
  for (int i = 0; i <= half; ++i) axis[i] = (double)i / fft_size * fs - width / 2.0;
  double origin = -(bnd - 0.5) * fs / fft_size;
  double step = (double)fs / fft_size;
  wo_interp1q(origin, step, seg, len, axis, half + 1, lo);
  for (int i = 0; i <= half; ++i) axis[i] += width;
  wo_interp1q(origin, step, seg, len, axis, half + 1, hi);
  for (int i = 0; i <= half; ++i) out[i] = (hi[i] - lo[i]) / width;
  free(seg); free(axis); free(lo); free(hi);
}

int wo_frame_count(int fs, int x_length, double frame_period) {
  return (int)(1000.0 * x_length / fs / frame_period) + 1;
}

/* ------------------------------------------------------------------ */
/* zero-crossing machinery shared by Harvest and DIO                    */
/* (harvest.cpp:162-238 == dio.cpp:357-435)                             */
/* ------------------------------------------------------------------ */
typedef struct { double *loc, *val; int n; } Intervals;

static void crossing_intervals(const double *s, int n, double fs, Intervals *out) {
  int *edge = ialloc(n);
  int cnt = 0;
  for (int i = 0; i < n - 1; ++i)
    if (0.0 < s[i] && s[i + 1] <= 0.0) edge[cnt++] = i + 1;
  out->n = 0;
  if (cnt >= 2) {
    double *fine = dalloc(cnt);
    for (int i = 0; i < cnt; ++i)
      fine[i] = edge[i] - s[edge[i] - 1] / (s[edge[i]] - s[edge[i] - 1]);
    for (int i = 0; i < cnt - 1; ++i) {
      out->val[i] = fs / (fine[i + 1] - fine[i]);
      out->loc[i] = (fine[i] + fine[i + 1]) / 2.0 / fs;
    }
    out->n = cnt - 1;
    free(fine);
  }
  free(edge);
}

/* four event families: falling, rising, peaks, dips (signal is clobbered) */
static void four_families(double *s, int n, double fs, Intervals fam[4]) {
  for (int f = 0; f < 4; ++f) { fam[f].loc = dalloc(n); fam[f].val = dalloc(n); }
  crossing_intervals(s, n, fs, &fam[0]);
  for (int i = 0; i < n; ++i) s[i] = -s[i];
  crossing_intervals(s, n, fs, &fam[1]);
  for (int i = 0; i < n - 1; ++i) s[i] = s[i] - s[i + 1];
  crossing_intervals(s, n - 1, fs, &fam[2]);
  for (int i = 0; i < n - 1; ++i) s[i] = -s[i];
  crossing_intervals(s, n - 1, fs, &fam[3]);
}

static void free_families(Intervals fam[4]) {
  for (int f = 0; f < 4; ++f) { free(fam[f].loc); free(fam[f].val); }
}

/* Spectral product + inverse + delay removal, including the reference's
 * mirror-write side effect on bins N/2-1 and N/2 (harvest.cpp:115-142,
 * dio.cpp:310-337): the mirror store at i = N/2-1 overwrites the filter's
 * Nyquist bin before it is multiplied, and the store at i = N/2 then
 * overwrites bin N/2-1. */
static void spectral_filter(const double *yr, const double *yi, double *hr, double *hi,
                            int fft_size, int bias, int y_length, double *sig) {
  int h = fft_size / 2;
  for (int i = 0; i <= h; ++i) {
    double tr = yr[i] * hr[i] - yi[i] * hi[i];
    double ti = yr[i] * hi[i] + yi[i] * hr[i];
    hr[i] = tr; hi[i] = ti;
    if (i >= 1) {
      int m = fft_size - i - 1;
      if (m <= h) { hr[m] = tr; hi[m] = ti; }
    }
  }
  double *full = dalloc(fft_size);
  wo_irfft_unscaled(fft_size, hr, hi, full);
  for (int i = 0; i < y_length; ++i) sig[i] = full[i + bias];
  free(full);
}

/* ------------------------------------------------------------------ */
/* Harvest                                                              */
/* ------------------------------------------------------------------ */
/* harvest.cpp:43-93 */
static void harvest_front_end(const double *x, int n, int y_len, int fft_size, int ratio,
                              double *y, double *Yr, double *Yi) {
  memset(y, 0, sizeof(double) * fft_size);
  if (ratio == 1) {
    memcpy(y, x, sizeof(double) * n);
  } else {
    int lag = (int)(ceil(140.0 / ratio) * ratio);
    int m = n + 2 * lag;
    double *px = dalloc(m), *py = dalloc(m);
    for (int i = 0; i < lag; ++i) px[i] = x[0];
    for (int i = 0; i < n; ++i) px[lag + i] = x[i];
    for (int i = lag + n; i < m; ++i) px[i] = x[n - 1];
    wo_decimate(px, m, ratio, py);
    for (int i = 0; i < y_len; ++i) y[i] = py[lag / ratio + i];
    free(px); free(py);
  }
  double mean = 0.0;
  for (int i = 0; i < y_len; ++i) mean += y[i];
  mean /= y_len;
  for (int i = 0; i < y_len; ++i) y[i] -= mean;
  for (int i = y_len; i < fft_size; ++i) y[i] = 0.0;
  wo_rfft(fft_size, y, Yr, Yi);
}

/* harvest.cpp:99-148, 240-329 : one band -> raw candidate contour */
static void harvest_band(double fb, double fs, const double *Yr, const double *Yi,
                         int y_len, int fft_size, double f0_floor, double f0_ceil,
                         const double *tpos, int nf, double *cand) {
  int L = wo_round(fs / fb * 2.0);
  double *h = dalloc(fft_size);
  wo_nuttall(2 * L + 1, h);
  for (int i = -L; i <= L; ++i) h[i + L] *= cos(2 * K_PI * fb * i / fs);
  double *hr = dalloc(fft_size), *hi = dalloc(fft_size);
  wo_rfft(fft_size, h, hr, hi);
  double *sig = dalloc(fft_size);
  spectral_filter(Yr, Yi, hr, hi, fft_size, L + 1, y_len, sig);

  Intervals fam[4];
  four_families(sig, y_len, fs, fam);
  int ok = 1;
  for (int f = 0; f < 4; ++f) if (fam[f].n - 2 <= 0) ok = 0;
  if (!ok) {
    for (int i = 0; i < nf; ++i) cand[i] = 0.0;
  } else {
    double *ip[4];
    for (int f = 0; f < 4; ++f) {
      ip[f] = dalloc(nf);
      wo_interp1(fam[f].loc, fam[f].val, fam[f].n, tpos, nf, ip[f]);
    }
    double up = fb * 1.1, lo = fb * 0.9;
    for (int i = 0; i < nf; ++i) {
      double c = (ip[0][i] + ip[1][i] + ip[2][i] + ip[3][i]) / 4.0;
      if (c > up || c < lo || c > f0_ceil || c < f0_floor) c = 0.0;
      cand[i] = c;
    }
    for (int f = 0; f < 4; ++f) free(ip[f]);
  }
  free_families(fam);
  free(h); free(hr); free(hi); free(sig);
}

/* harvest.cpp:348-412 : runs of >= 10 voiced bands -> candidates (mean) */
static int harvest_detect(double **raw, int nch, int nf, int maxc, double **cands) {
  int nc = 0;
  int *v = ialloc(nch);
  for (int i = 0; i < nf; ++i) {
    for (int j = 0; j < nch; ++j) v[j] = raw[j][i] > 0 ? 1 : 0;
    v[0] = v[nch - 1] = 0;
    int cnt = 0, st = 0;
    for (int j = 1; j < nch; ++j) {
      int d = v[j] - v[j - 1];
      if (d == 1) st = j;
      if (d == -1) {
        if (j - st >= 10) {
          double s = 0.0;
          for (int k = st; k < j; ++k) s += raw[k][i];
          cands[i][cnt++] = s / (j - st);
        }
      }
    }
    for (int j = cnt; j < maxc; ++j) cands[i][j] = 0.0;
    nc = imax(nc, cnt);
  }
  free(v);
  return nc;
}

/* harvest.cpp:417-429 */
static void harvest_overlap(int nf, int nc, double **cands) {
  for (int i = 1; i <= 3; ++i)
    for (int j = 0; j < nc; ++j) {
      for (int k = i; k < nf; ++k) cands[k][j + nc * i] = cands[k - i][j];
      for (int k = 0; k < nf - i; ++k) cands[k][j + nc * (i + 3)] = cands[k + i][j];
    }
}

/* harvest.cpp:434-617 : instantaneous-frequency refinement of one candidate */
static void harvest_refine_one(const double *y, int y_len, double fs, double pos, double f0,
                               double f0_floor, double f0_ceil, double *rf0, double *rscore) {
  if (f0 <= 0.0) { *rf0 = 0.0; *rscore = 0.0; return; }
  int hw = (int)(1.5 * fs / f0 + 1.0);
  double wlen = (2.0 * hw + 1.0) / fs;
  int blen = 2 * hw + 1;
  int N = (int)pow(2.0, 2.0 + (int)(log(hw * 2.0 + 1.0) / K_LOG2));
  double base0 = (-hw + 0) / fs;
  int first = wo_round((pos + base0) * fs + 0.001);
  double *mw = dalloc(blen), *dw = dalloc(blen), *buf = dalloc(N);
  double *ar = dalloc(N / 2 + 1), *ai = dalloc(N / 2 + 1);
  double *br = dalloc(N / 2 + 1), *bi = dalloc(N / 2 + 1);
  for (int i = 0; i < blen; ++i) {
    double t = ((first + i) - 1.0) / fs - pos;
    mw[i] = 0.42 + 0.5 * cos(2.0 * K_PI * t / wlen) + 0.08 * cos(4.0 * K_PI * t / wlen);
  }
  dw[0] = -mw[1] / 2.0;
  for (int i = 1; i < blen - 1; ++i) dw[i] = -(mw[i + 1] - mw[i - 1]) / 2.0;
  dw[blen - 1] = mw[blen - 2] / 2.0;
  for (int i = 0; i < blen; ++i) buf[i] = y[imax(0, imin(y_len - 1, first + i - 1))] * mw[i];
  wo_rfft(N, buf, ar, ai);
  for (int i = 0; i < blen; ++i) buf[i] = y[imax(0, imin(y_len - 1, first + i - 1))] * dw[i];
  wo_rfft(N, buf, br, bi);

  int nh = imin((int)(fs / 2.0 / f0), 6);
  double num = 0.0, den = 0.0, sc = 0.0;
  for (int k = 0; k < nh; ++k) {
    int idx = wo_round(f0 * N / fs * (k + 1));
    double pw = ar[idx] * ar[idx] + ai[idx] * ai[idx];
    double ni = ar[idx] * bi[idx] - ai[idx] * br[idx];
    double inst = pw == 0.0 ? 0.0 : (double)idx * fs / N + ni / pw * fs / 2.0 / K_PI;
    double amp = sqrt(pw);
    num += amp * inst;
    den += amp * (k + 1.0);
    sc += fabs((inst / (k + 1.0) - f0) / f0);
  }
  *rf0 = num / (den + K_TINY);
  *rscore = 1.0 / (sc / nh + K_TINY);
  if (*rf0 < f0_floor || *rf0 > f0_ceil || *rscore < 2.5) { *rf0 = 0.0; *rscore = 0.0; }
  free(mw); free(dw); free(buf); free(ar); free(ai); free(br); free(bi);
}

/* harvest.cpp:636-650 : nearest candidate, ties resolved towards the LAST */
static double nearest_candidate(double ref, const double *c, int nc, double allowed, double *err) {
  double best = 0.0;
  *err = allowed;
  for (int i = 0; i < nc; ++i) {
    double e = fabs(ref - c[i]) / ref;
    if (e > *err) continue;
    best = c[i];
    *err = e;
  }
  return best;
}

/* harvest.cpp:652-688 */
static void harvest_prune(int nf, int nc, double **cands, double **scores) {
  double **snap = (double **)malloc(sizeof(double *) * nf);
  for (int i = 0; i < nf; ++i) {
    snap[i] = dalloc(nc);
    memcpy(snap[i], cands[i], sizeof(double) * nc);
  }
  for (int i = 1; i < nf - 1; ++i)
    for (int j = 0; j < nc; ++j) {
      double ref = cands[i][j], e1, e2;
      if (ref == 0) continue;
      nearest_candidate(ref, snap[i + 1], nc, 1.0, &e1);
      nearest_candidate(ref, snap[i - 1], nc, 1.0, &e2);
      if (dmin(e1, e2) <= 0.05) continue;
      cands[i][j] = 0; scores[i][j] = 0;
    }
  for (int i = 0; i < nf; ++i) free(snap[i]);
  free(snap);
}

/* harvest.cpp:727-743 : [start,end] index pairs of voiced runs */
static int voiced_runs(const double *f0, int nf, int *bl) {
  int nb = 0, prev = 0;
  for (int i = 1; i < nf; ++i) {
    int cur = (i == nf - 1) ? 0 : (f0[i] > 0 ? 1 : 0);
    if (cur != prev) { bl[nb] = i - nb % 2; nb++; }
    prev = cur;
  }
  return nb;
}

/* harvest.cpp:791-820 */
static int extend_run(int origin, int last, int shift, double **cands, int nc,
                      double allowed, double *ext) {
  double cur = ext[origin];
  int moved = origin, miss = 0;
  int dist = last > origin ? last - origin : origin - last;
  for (int i = 0; i <= dist; ++i) {
    int t = origin + shift * i + shift;
    double e;
    ext[t] = nearest_candidate(cur, cands[t], nc, allowed, &e);
    if (ext[t] == 0.0) { miss++; } else { cur = ext[t]; miss = 0; moved = t; }
    if (miss == 4) break;
  }
  return moved;
}

/* harvest.cpp:901-907 */
static double best_score_of(double f0, const double *c, const double *s, int nc) {
  double r = 0.0;
  for (int i = 0; i < nc; ++i) if (f0 == c[i] && r < s[i]) r = s[i];
  return r;
}

/* harvest.cpp:968-995 with 767-963 folded in (step 3 of the contour fix) */
static void harvest_step3(const double *in, int nf, int nc, double **cands, double **scores,
                          double allowed, double *out) {
  memcpy(out, in, sizeof(double) * nf);
  int *bl = ialloc(nf);
  int nb = voiced_runs(in, nf, bl);
  int ns = nb / 2;
  double **ch = (double **)malloc(sizeof(double *) * (ns ? ns : 1));
  for (int s = 0; s < ns; ++s) {
    ch[s] = dalloc(nf);
    for (int j = bl[2 * s]; j <= bl[2 * s + 1]; ++j) ch[s][j] = in[j];
  }
  /* Extend(): forwards then backwards, at most 100 frames each (861-878) */
  for (int s = 0; s < ns; ++s) {
    int ed = extend_run(bl[2 * s + 1], imin(nf - 2, bl[2 * s + 1] + 100), 1, cands, nc, allowed, ch[s]);
    int st = extend_run(bl[2 * s], imax(1, bl[2 * s] - 100), -1, cands, nc, allowed, ch[s]);
    bl[2 * s + 1] = ed;
    bl[2 * s] = st;
  }
  /* ExtendSub(): keep runs longer than 2200/mean_f0; the running mean is NOT
   * reset between runs (840-856) */
  int kept = 0;
  double mean = 0.0;
  for (int s = 0; s < ns; ++s) {
    int st = bl[2 * s], ed = bl[2 * s + 1];
    for (int j = st; j < ed; ++j) mean += ch[s][j];
    mean /= ed - st;
    if (2200.0 / mean < ed - st) {
      double *tp = ch[kept]; ch[kept] = ch[s]; ch[s] = tp;
      int t = bl[2 * kept]; bl[2 * kept] = bl[2 * s]; bl[2 * s] = t;
      t = bl[2 * kept + 1]; bl[2 * kept + 1] = bl[2 * s + 1]; bl[2 * s + 1] = t;
      kept++;
    }
  }
  if (kept != 0) {
    /* MergeF0() (937-963), including its in-place use of bl[0], bl[1] */
    int *order = ialloc(kept);
    for (int i = 0; i < kept; ++i) order[i] = i;
    for (int i = 1; i < kept; ++i)
      for (int j = i - 1; j >= 0; --j) {
        if (bl[order[j] * 2] > bl[order[i] * 2]) {
          int t = order[i]; order[i] = order[j]; order[j] = t;
        } else {
          break;
        }
      }
    for (int i = 0; i < nf; ++i) out[i] = ch[0][i];
    for (int i = 1; i < kept; ++i) {
      int o = order[i];
      if (bl[o * 2] - bl[1] > 0) {
        for (int j = bl[o * 2]; j <= bl[o * 2 + 1]; ++j) out[j] = ch[o][j];
        bl[0] = bl[o * 2];
        bl[1] = bl[o * 2 + 1];
      } else {
        int st1 = bl[0], ed1 = bl[1], st2 = bl[o * 2], ed2 = bl[o * 2 + 1];
        const double *f2 = ch[o];
        if (st1 <= st2 && ed1 >= ed2) { bl[1] = ed1; continue; }
        double s1 = 0.0, s2 = 0.0;
        for (int k = st2; k <= ed1; ++k) {
          s1 += best_score_of(out[k], cands[k], scores[k], nc);
          s2 += best_score_of(f2[k], cands[k], scores[k], nc);
        }
        if (s1 > s2) for (int k = ed1; k <= ed2; ++k) out[k] = f2[k];
        else for (int k = st2; k <= ed2; ++k) out[k] = f2[k];
        bl[1] = ed2;
      }
    }
    free(order);
  }
  for (int s = 0; s < ns; ++s) free(ch[s]);
  free(ch); free(bl);
}

/* harvest.cpp:693-1044 */
static void harvest_fix_contour(double **cands, double **scores, int nf, int nc, double *best) {
  double *a = dalloc(nf), *b = dalloc(nf);
  int *bl = ialloc(nf);
  /* SearchF0Base: first maximum wins */
  for (int i = 0; i < nf; ++i) {
    double top = 0.0; a[i] = 0.0;
    for (int j = 0; j < nc; ++j)
      if (scores[i][j] > top) { a[i] = cands[i][j]; top = scores[i][j]; }
  }
  /* step 1 (710-722) */
  for (int i = 0; i < nf; ++i) b[i] = 0.0;
  for (int i = 2; i < nf; ++i) {
    if (a[i] == 0.0) continue;
    double ref = a[i - 1] * 2 - a[i - 2];
    b[i] = fabs((a[i] - ref) / ref) > 0.008 && fabs((a[i] - a[i - 1])) / a[i - 1] > 0.008 ? 0.0 : a[i];
  }
  /* step 2 (748-762) */
  memcpy(a, b, sizeof(double) * nf);
  int nb = voiced_runs(b, nf, bl);
  for (int s = 0; s < nb / 2; ++s) {
    if (bl[2 * s + 1] - bl[2 * s] >= 6) continue;
    for (int j = bl[2 * s]; j <= bl[2 * s + 1]; ++j) a[j] = 0.0;
  }
  /* step 3 */
  harvest_step3(a, nf, nc, cands, scores, 0.18, b);
  /* step 4 (1000-1022) */
  memcpy(best, b, sizeof(double) * nf);
  nb = voiced_runs(b, nf, bl);
  for (int s = 0; s < nb / 2 - 1; ++s) {
    int dist = bl[(s + 1) * 2] - bl[s * 2 + 1] - 1;
    if (dist >= 9) continue;
    double t0 = b[bl[s * 2 + 1]] + 1;
    double t1 = b[bl[(s + 1) * 2]] - 1;
    double coef = (t1 - t0) / (dist + 1.0);
    int c = 1;
    for (int j = bl[s * 2 + 1] + 1; j <= bl[(s + 1) * 2] - 1; ++j) best[j] = t0 + coef * c++;
  }
  free(a); free(b); free(bl);
}

/* harvest.cpp:1049-1113 */
static void harvest_smooth(const double *f0, int nf, double *out) {
  const double b[2] = {0.0078202080334971724, 0.015640416066994345};
  const double a[2] = {1.7347257688092754, -0.76600660094326412};
  const int lag = 300;
  int m = nf + 2 * lag;
  double *c = dalloc(m), *chan = dalloc(m), *tmp = dalloc(m), *res = dalloc(m);
  int *bl = ialloc(m);
  for (int i = 0; i < nf; ++i) c[lag + i] = f0[i];
  int nb = voiced_runs(c, m, bl);
  for (int s = 0; s < nb / 2; ++s) {
    int st = bl[2 * s], ed = bl[2 * s + 1];
    for (int i = 0; i < m; ++i) chan[i] = (i < st) ? c[st] : (i > ed ? c[ed] : c[i]);
    double w0 = 0.0, w1 = 0.0;
    for (int i = 0; i < m; ++i) {
      double wt = chan[i] + a[0] * w0 + a[1] * w1;
      tmp[m - i - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
      w1 = w0; w0 = wt;
    }
    w0 = w1 = 0.0;
    for (int i = 0; i < m; ++i) {
      double wt = tmp[i] + a[0] * w0 + a[1] * w1;
      res[m - i - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
      w1 = w0; w0 = wt;
    }
    for (int j = st; j <= ed; ++j) out[j - lag] = res[j];
  }
  free(c); free(chan); free(tmp); free(res); free(bl);
}

/* harvest.cpp:1145-1215 */
static void harvest_body(const double *x, int n, int fs, int frame_period, double f0_floor,
                         double f0_ceil, double ch_per_oct, int speed, double *tpos, double *f0) {
  double lo = f0_floor * 0.9, hi = f0_ceil * 1.1;
  int nch = 1 + (int)(log(hi / lo) / K_LOG2 * ch_per_oct);
  double *fb = dalloc(nch);
  for (int i = 0; i < nch; ++i) fb[i] = lo * pow(2.0, (i + 1) / ch_per_oct);
  int ratio = imax(imin(speed, 12), 1);
  int y_len = (int)ceil((double)n / ratio);
  double afs = (double)fs / ratio;
  int fft_size = wo_pow2_above(y_len + 5 + 2 * (int)(2.0 * afs / fb[0]));
  double *y = dalloc(fft_size), *Yr = dalloc(fft_size), *Yi = dalloc(fft_size);
  harvest_front_end(x, n, y_len, fft_size, ratio, y, Yr, Yi);

  int nf = wo_frame_count(fs, n, frame_period);
  for (int i = 0; i < nf; ++i) { tpos[i] = i * frame_period / 1000.0; f0[i] = 0.0; }
  int maxc = wo_round(nch / 10.0) * 7;
  double **cands = (double **)malloc(sizeof(double *) * nf);
  double **scores = (double **)malloc(sizeof(double *) * nf);
  for (int i = 0; i < nf; ++i) { cands[i] = dalloc(maxc); scores[i] = dalloc(maxc); }
  double **raw = (double **)malloc(sizeof(double *) * nch);
  for (int j = 0; j < nch; ++j) {
    raw[j] = dalloc(nf);
    harvest_band(fb[j], afs, Yr, Yi, y_len, fft_size, f0_floor, f0_ceil, tpos, nf, raw[j]);
  }
  const char *dump = getenv("WO_DUMP");              /* test infrastructure: intermediate stages for debugging */
  if (dump) { char fn[512]; snprintf(fn, sizeof fn, "%s_raw.bin", dump); FILE *f = fopen(fn, "wb");
    for (int j = 0; j < nch; ++j) fwrite(raw[j], sizeof(double), nf, f); fclose(f);
    snprintf(fn, sizeof fn, "%s_y.bin", dump); f = fopen(fn, "wb"); fwrite(y, sizeof(double), y_len, f); fclose(f); }
  int nc = harvest_detect(raw, nch, nf, maxc, cands);
  if (dump) { char fn[512]; snprintf(fn, sizeof fn, "%s_det.bin", dump); FILE *f = fopen(fn, "wb");
    for (int i = 0; i < nf; ++i) fwrite(cands[i], sizeof(double), maxc, f); fclose(f); }
  harvest_overlap(nf, nc, cands);
  nc *= 7;
  for (int i = 0; i < nf; ++i)
    for (int j = 0; j < nc; ++j)
      harvest_refine_one(y, y_len, afs, tpos[i], cands[i][j], f0_floor, f0_ceil,
                         &cands[i][j], &scores[i][j]);
  if (dump) { char fn[512]; snprintf(fn, sizeof fn, "%s_ref.bin", dump); FILE *f = fopen(fn, "wb");
    for (int i = 0; i < nf; ++i) fwrite(cands[i], sizeof(double), maxc, f);
    for (int i = 0; i < nf; ++i) fwrite(scores[i], sizeof(double), maxc, f); fclose(f); }
  harvest_prune(nf, nc, cands, scores);
  if (dump) { char fn[512]; snprintf(fn, sizeof fn, "%s_prune.bin", dump); FILE *f = fopen(fn, "wb");
    for (int i = 0; i < nf; ++i) fwrite(cands[i], sizeof(double), maxc, f); fclose(f); }
  double *best = dalloc(nf);
  harvest_fix_contour(cands, scores, nf, nc, best);
  if (dump) { char fn[512]; snprintf(fn, sizeof fn, "%s_best.bin", dump); FILE *f = fopen(fn, "wb");
    fwrite(best, sizeof(double), nf, f); fclose(f); }
  harvest_smooth(best, nf, f0);

  for (int j = 0; j < nch; ++j) free(raw[j]);
  for (int i = 0; i < nf; ++i) { free(cands[i]); free(scores[i]); }
  free(raw); free(cands); free(scores); free(best);
  free(y); free(Yr); free(Yi); free(fb);
}

/* harvest.cpp:1223-1255 */
void wo_harvest(const double *x, int x_length, int fs, double f0_floor, double f0_ceil,
                double frame_period, double *tpos, double *f0) {
  int ratio = wo_round(fs / 8000.0);
  if (frame_period == 1.0) {
    harvest_body(x, x_length, fs, 1, f0_floor, f0_ceil, 40, ratio, tpos, f0);
    return;
  }
  int bn = wo_frame_count(fs, x_length, 1);
  double *bf0 = dalloc(bn), *bt = dalloc(bn);
  harvest_body(x, x_length, fs, 1, f0_floor, f0_ceil, 40, ratio, bt, bf0);
  int nf = wo_frame_count(fs, x_length, frame_period);
  for (int i = 0; i < nf; ++i) {
    tpos[i] = i * frame_period / 1000.0;
    f0[i] = bf0[imin(bn - 1, wo_round(tpos[i] * 1000.0))];
  }
  free(bf0); free(bt);
}

/* ------------------------------------------------------------------ */
/* DIO                                                                  */
/* ------------------------------------------------------------------ */
/* dio.cpp:40-53 */
static void dio_low_cut(int N, int fft_size, double *f) {
  for (int i = 1; i <= N; ++i) f[i - 1] = 0.5 - 0.5 * cos(i * 2.0 * K_PI / (N + 1));
  for (int i = N; i < fft_size; ++i) f[i] = 0.0;
  double s = 0.0;
  for (int i = 0; i < N; ++i) s += f[i];
  for (int i = 0; i < N; ++i) f[i] = -f[i] / s;
  for (int i = 0; i < (N - 1) / 2; ++i) f[fft_size - (N - 1) / 2 + i] = f[i];
  for (int i = 0; i < N; ++i) f[i] = f[i + (N - 1) / 2];
  f[0] += 1.0;
}

/* dio.cpp:60-106 */
static void dio_front_end(const double *x, int n, int y_len, double afs, int fft_size,
                          int ratio, double *Yr, double *Yi) {
  double *y = dalloc(fft_size);
  if (ratio != 1) wo_decimate(x, n, ratio, y);
  else memcpy(y, x, sizeof(double) * n);
  double mean = 0.0;
  for (int i = 0; i < y_len; ++i) mean += y[i];
  mean /= y_len;
  for (int i = 0; i < y_len; ++i) y[i] -= mean;
  for (int i = y_len; i < fft_size; ++i) y[i] = 0.0;
  wo_rfft(fft_size, y, Yr, Yi);
  int cut = wo_round(afs / 50.0);
  dio_low_cut(cut * 2 + 1, fft_size, y);
  double *Fr = dalloc(fft_size / 2 + 1), *Fi = dalloc(fft_size / 2 + 1);
  wo_rfft(fft_size, y, Fr, Fi);
  for (int i = 0; i <= fft_size / 2; ++i) {
    double t = Yr[i] * Fr[i] - Yi[i] * Fi[i];
    Yi[i] = Yr[i] * Fi[i] + Yi[i] * Fr[i];
    Yr[i] = t;
  }
  free(y); free(Fr); free(Fi);
}

/* dio.cpp:296-343, 441-544 : one band -> candidate + score */
static void dio_band(double fb, double fs, const double *Yr, const double *Yi, int y_len,
                     int fft_size, double f0_floor, double f0_ceil, const double *tpos,
                     int nf, double *cand, double *score) {
  int hal = wo_round(fs / fb / 2.0);
  double *h = dalloc(fft_size);
  wo_nuttall(hal * 4, h);
  double *hr = dalloc(fft_size), *hi = dalloc(fft_size);
  wo_rfft(fft_size, h, hr, hi);
  double *sig = dalloc(fft_size);
  spectral_filter(Yr, Yi, hr, hi, fft_size, hal * 2, y_len, sig);
  Intervals fam[4];
  four_families(sig, y_len, fs, fam);
  int ok = 1;
  for (int f = 0; f < 4; ++f) if (fam[f].n - 2 <= 0) ok = 0;
  if (!ok) {
    for (int i = 0; i < nf; ++i) { cand[i] = 0.0; score[i] = K_BIG; }
  } else {
    double *ip[4];
    for (int f = 0; f < 4; ++f) {
      ip[f] = dalloc(nf);
      wo_interp1(fam[f].loc, fam[f].val, fam[f].n, tpos, nf, ip[f]);
    }
    for (int i = 0; i < nf; ++i) {
      double c = (ip[0][i] + ip[1][i] + ip[2][i] + ip[3][i]) / 4.0;
      double s = sqrt(((ip[0][i] - c) * (ip[0][i] - c) + (ip[1][i] - c) * (ip[1][i] - c) +
                       (ip[2][i] - c) * (ip[2][i] - c) + (ip[3][i] - c) * (ip[3][i] - c)) / 3.0);
      if (c > fb || c < fb / 2.0 || c > f0_ceil || c < f0_floor) { c = 0.0; s = K_BIG; }
      cand[i] = c; score[i] = s;
    }
    for (int f = 0; f < 4; ++f) free(ip[f]);
  }
  free_families(fam);
  free(h); free(hr); free(hi); free(sig);
}

/* dio.cpp:190-209 */
static double dio_track(double cur, double past, double **cands, int nb, int t, double allowed) {
  double ref = (cur * 3.0 - past) / 2.0;
  double emin = fabs(ref - cands[0][t]);
  double best = cands[0][t];
  for (int i = 1; i < nb; ++i) {
    double e = fabs(ref - cands[i][t]);
    if (e < emin) { emin = e; best = cands[i][t]; }
  }
  if (fabs(1.0 - best / ref) > allowed) return 0.0;
  return best;
}

/* dio.cpp:112-289 */
static void dio_fix_contour(double frame_period, int nb, double **cands, const double *best,
                            int nf, double f0_floor, double allowed, double *out) {
  int vrm = (int)(0.5 + 1000.0 / frame_period / f0_floor) * 2 + 1;
  if (nf <= vrm) return;
  double *t1 = dalloc(nf), *t2 = dalloc(nf), *base = dalloc(nf);
  /* step 1 */
  for (int i = vrm; i < nf - vrm; ++i) base[i] = best[i];
  for (int i = vrm; i < nf; ++i)
    t1[i] = fabs((base[i] - base[i - 1]) / (K_TINY + base[i])) < allowed ? base[i] : 0.0;
  /* step 2 */
  memcpy(t2, t1, sizeof(double) * nf);
  int center = (vrm - 1) / 2;
  for (int i = center; i < nf - center; ++i)
    for (int j = -center; j <= center; ++j)
      if (t1[i + j] == 0) { t2[i] = 0.0; break; }
  /* section edges */
  int *pos = ialloc(nf), *neg = ialloc(nf), np = 0, nn = 0;
  for (int i = 1; i < nf; ++i) {
    if (t2[i] == 0 && t2[i - 1] != 0) neg[nn++] = i - 1;
    else if (t2[i - 1] == 0 && t2[i] != 0) pos[np++] = i;
  }
  /* step 3: forward tracking */
  memcpy(t1, t2, sizeof(double) * nf);
  for (int i = 0; i < nn; ++i) {
    int limit = i == nn - 1 ? nf - 1 : neg[i + 1];
    for (int j = neg[i]; j < limit; ++j) {
      t1[j + 1] = dio_track(t1[j], t1[j - 1], cands, nb, j + 1, allowed);
      if (t1[j + 1] == 0) break;
    }
  }
  /* step 4: backward tracking */
  memcpy(out, t1, sizeof(double) * nf);
  for (int i = np - 1; i >= 0; --i) {
    int limit = i == 0 ? 1 : pos[i - 1];
    for (int j = pos[i]; j > limit; --j) {
      out[j - 1] = dio_track(out[j], out[j + 1], cands, nb, j - 1, allowed);
      if (out[j - 1] == 0) break;
    }
  }
  free(t1); free(t2); free(base); free(pos); free(neg);
}

/* dio.cpp:578-648 */
void wo_dio(const double *x, int x_length, int fs, double f0_floor, double f0_ceil,
            double channels_in_octave, double frame_period, int speed, double allowed_range,
            double *tpos, double *f0) {
  int nb = 1 + (int)(log(f0_ceil / f0_floor) / K_LOG2 * channels_in_octave);
  double *fb = dalloc(nb);
  for (int i = 0; i < nb; ++i) fb[i] = f0_floor * pow(2.0, (i + 1) / channels_in_octave);
  int ratio = imax(imin(speed, 12), 1);
  int y_len = 1 + (int)(x_length / ratio);
  double afs = (double)fs / ratio;
  int fft_size = wo_pow2_above(y_len + wo_round(afs / 50.0) * 2 + 1 +
                               (4 * (int)(1.0 + afs / fb[0] / 2.0)));
  double *Yr = dalloc(fft_size), *Yi = dalloc(fft_size);
  dio_front_end(x, x_length, y_len, afs, fft_size, ratio, Yr, Yi);
  int nf = wo_frame_count(fs, x_length, frame_period);
  double **cands = (double **)malloc(sizeof(double *) * nb);
  double **scores = (double **)malloc(sizeof(double *) * nb);
  for (int i = 0; i < nf; ++i) tpos[i] = i * frame_period / 1000.0;
  double *c = dalloc(nf), *s = dalloc(nf);
  for (int b = 0; b < nb; ++b) {
    cands[b] = dalloc(nf); scores[b] = dalloc(nf);
    dio_band(fb[b], afs, Yr, Yi, y_len, fft_size, f0_floor, f0_ceil, tpos, nf, c, s);
    for (int j = 0; j < nf; ++j) {
      scores[b][j] = s[j] / (c[j] + K_TINY);
      cands[b][j] = c[j];
    }
  }
  double *best = dalloc(nf);
  for (int i = 0; i < nf; ++i) {
    double t = scores[0][i];
    best[i] = cands[0][i];
    for (int b = 1; b < nb; ++b)
      if (t > scores[b][i]) { t = scores[b][i]; best[i] = cands[b][i]; }
  }
  dio_fix_contour(frame_period, nb, cands, best, nf, f0_floor, allowed_range, f0);
  for (int b = 0; b < nb; ++b) { free(cands[b]); free(scores[b]); }
  free(cands); free(scores); free(c); free(s); free(best); free(Yr); free(Yi); free(fb);
}

/* ------------------------------------------------------------------ */
/* StoneMask (stonemask.cpp:24-218)                                     */
/* ------------------------------------------------------------------ */
static double stonemask_if(const double *pw, const double *ni, int N, int fs, double f0, int nh) {
  double num = 0.0, den = 0.0;
  for (int k = 0; k < nh; ++k) {
    int idx = imin(wo_round(f0 * N / fs * (k + 1)), N / 2);
    double inst = pw[idx] == 0.0 ? 0.0 : (double)idx * fs / N + ni[idx] / pw[idx] * fs / 2.0 / K_PI;
    double amp = sqrt(pw[idx]);
    num += amp * inst;
    den += amp * (k + 1);
  }
  return num / (den + K_TINY);
}

static double stonemask_one(const double *x, int n, int fs, double pos, double f0) {
  if (f0 <= 40.0 || f0 > fs / 12.0) return 0.0;
  int hw = (int)(1.5 * fs / f0 + 1.0);
  double wlen = (2.0 * hw + 1.0) / fs;
  int blen = 2 * hw + 1;
  int N = (int)pow(2.0, 2.0 + (int)(log(hw * 2.0 + 1.0) / K_LOG2));
  int *raw = ialloc(blen);
  double *mw = dalloc(blen), *dw = dalloc(blen), *buf = dalloc(N);
  double *ar = dalloc(N / 2 + 1), *ai = dalloc(N / 2 + 1), *br = dalloc(N / 2 + 1), *bi = dalloc(N / 2 + 1);
  for (int i = 0; i < blen; ++i) {
    double bt = (double)(-hw + i) / fs;
    raw[i] = wo_round((pos + bt) * fs);
    double t = (raw[i] - 1.0) / fs - pos;
    mw[i] = 0.42 + 0.5 * cos(2.0 * K_PI * t / wlen) + 0.08 * cos(4.0 * K_PI * t / wlen);
  }
  dw[0] = -mw[1] / 2.0;
  for (int i = 1; i < blen - 1; ++i) dw[i] = -(mw[i + 1] - mw[i - 1]) / 2.0;
  dw[blen - 1] = mw[blen - 2] / 2.0;
  for (int i = 0; i < blen; ++i) buf[i] = x[imax(0, imin(n - 1, raw[i] - 1))] * mw[i];
  wo_rfft(N, buf, ar, ai);
  for (int i = 0; i < blen; ++i) buf[i] = x[imax(0, imin(n - 1, raw[i] - 1))] * dw[i];
  wo_rfft(N, buf, br, bi);
  double *pw = dalloc(N / 2 + 1), *ni = dalloc(N / 2 + 1);
  for (int j = 0; j <= N / 2; ++j) {
    ni[j] = ar[j] * bi[j] - ai[j] * br[j];
    pw[j] = ar[j] * ar[j] + ai[j] * ai[j];
  }
  double tent = stonemask_if(pw, ni, N, fs, f0, 2);
  double mean;
  if (tent <= 0.0 || tent > f0 * 2) mean = 0.0;
  else mean = stonemask_if(pw, ni, N, fs, tent, 6);
  if (fabs(mean - f0) > f0 * 0.2) mean = f0;
  free(raw); free(mw); free(dw); free(buf); free(ar); free(ai); free(br); free(bi); free(pw); free(ni);
  return mean;
}

void wo_stonemask(const double *x, int x_length, int fs, const double *tpos, const double *f0,
                  int nf, double *refined) {
  for (int i = 0; i < nf; ++i) refined[i] = stonemask_one(x, x_length, fs, tpos[i], f0[i]);
}

/* ------------------------------------------------------------------ */
/* CheapTrick (cheaptrick.cpp:22-229)                                   */
/* ------------------------------------------------------------------ */
int wo_cheaptrick_fft_size(int fs, double f0_floor) {
  return (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / f0_floor + 1) / K_LOG2));
}

void wo_cheaptrick(const double *x, int x_length, int fs, const double *tpos, const double *f0,
                   int nf, double q1, int fft_size, double *spectrogram) {
  uint32_t rng[4];
  wo_randn_seed(rng);
  int half = fft_size / 2, nb = half + 1;
  double floor_f0 = 3.0 * fs / (fft_size - 3.0);
  double *wave = dalloc(fft_size), *win = dalloc(fft_size);
  double *re = dalloc(nb), *im = dalloc(nb), *zero = dalloc(nb);
  for (int f = 0; f < nf; ++f) {
    double cf0 = f0[f] <= floor_f0 ? 500.0 : f0[f];
    /* windowed waveform (87-142) */
    int hw = wo_round(1.5 * fs / cf0);
    int origin = wo_round(tpos[f] * fs + 0.001);
    double e = 0.0;
    for (int i = 0; i <= 2 * hw; ++i) {
      double p = (i - hw) / 1.5 / fs;
      win[i] = 0.5 * cos(K_PI * p * cf0) + 0.5;
      e += win[i] * win[i];
    }
    e = sqrt(e);
    for (int i = 0; i <= 2 * hw; ++i) win[i] /= e;
    for (int i = 0; i <= 2 * hw; ++i)
      wave[i] = x[imin(x_length - 1, imax(0, origin + i - hw))] * win[i] + wo_randn(rng) * K_TINY;
    double s1 = 0, s2 = 0;
    for (int i = 0; i <= 2 * hw; ++i) { s1 += wave[i]; s2 += win[i]; }
    double coef = s1 / s2;
    for (int i = 0; i <= 2 * hw; ++i) wave[i] -= win[i] * coef;
    for (int i = 2 * hw + 1; i < fft_size; ++i) wave[i] = 0.0;
    /* power spectrum + DC correction (64-82) */
    wo_rfft(fft_size, wave, re, im);
    for (int i = 0; i <= half; ++i) wave[i] = re[i] * re[i] + im[i] * im[i];
    wo_dc_correction(wave, cf0, fs, fft_size, wave);
    wo_linear_smoothing(wave, cf0 * 2.0 / 3.0, fs, fft_size, wave);
    for (int i = 0; i <= half; ++i) wave[i] = wave[i] + fabs(wo_randn(rng)) * K_EPS;
    /* cepstral smoothing + recovery (22-57) */
    for (int i = 0; i <= half; ++i) wave[i] = log(wave[i]);
    for (int i = 1; i < half; ++i) wave[fft_size - i] = wave[i];
    wo_rfft(fft_size, wave, re, im);
    for (int i = 0; i <= half; ++i) {
      double sl, cl;
      if (i == 0) { sl = 1.0; cl = (1.0 - 2.0 * q1) + 2.0 * q1; }
      else {
        double q = (double)i / fs;
        sl = sin(K_PI * cf0 * q) / (K_PI * cf0 * q);
        cl = (1.0 - 2.0 * q1) + 2.0 * q1 * cos(2.0 * K_PI * q * cf0);
      }
      re[i] = re[i] * sl * cl / fft_size;
    }
    wo_irfft_unscaled(fft_size, re, zero, wave);
    for (int i = 0; i <= half; ++i) spectrogram[(size_t)f * nb + i] = exp(wave[i]);
  }
  free(wave); free(win); free(re); free(im); free(zero);
}

/* ------------------------------------------------------------------ */
/* D4C (d4c.cpp:21-403)                                                 */
/* ------------------------------------------------------------------ */
/* d4c.cpp:21-84 ; kind 1 = Hanning, 2 = Blackman */
static int d4c_window(const double *x, int n, int fs, double f0, double pos, int kind,
                      double ratio, double *wave, uint32_t rng[4]) {
  int hw = wo_round(ratio * fs / f0 / 2.0);
  int origin = wo_round(pos * fs + 0.001);
  double *win = dalloc(2 * hw + 1);
  for (int i = 0; i <= 2 * hw; ++i) {
    double p = (2.0 * (i - hw) / ratio) / fs;
    if (kind == 1) win[i] = 0.5 * cos(K_PI * p * f0) + 0.5;
    else win[i] = 0.42 + 0.5 * cos(K_PI * p * f0) + 0.08 * cos(K_PI * p * f0 * 2);
  }
  for (int i = 0; i <= 2 * hw; ++i)
    wave[i] = x[imin(n - 1, imax(0, origin + i - hw))] * win[i] + wo_randn(rng) * 0.000001;
  double s1 = 0, s2 = 0;
  for (int i = 0; i <= 2 * hw; ++i) { s1 += wave[i]; s2 += win[i]; }
  double coef = s1 / s2;
  for (int i = 0; i <= 2 * hw; ++i) wave[i] -= win[i] * coef;
  free(win);
  return hw;
}

/* d4c.cpp:90-120 */
static void d4c_centroid(const double *x, int n, int fs, double f0, int N, double pos,
                         double *wave, double *cen, uint32_t rng[4]) {
  memset(wave, 0, sizeof(double) * N);
  d4c_window(x, n, fs, f0, pos, 2, 4.0, wave, rng);
  int last = wo_round(2.0 * fs / f0) * 2;
  double pw = 0.0;
  for (int i = 0; i <= last; ++i) pw += wave[i] * wave[i];
  for (int i = 0; i <= last; ++i) wave[i] /= sqrt(pw);
  double *ar = dalloc(N / 2 + 1), *ai = dalloc(N / 2 + 1), *br = dalloc(N / 2 + 1), *bi = dalloc(N / 2 + 1);
  wo_rfft(N, wave, ar, ai);
  for (int i = 0; i < N; ++i) wave[i] *= i + 1.0;
  wo_rfft(N, wave, br, bi);
  for (int i = 0; i <= N / 2; ++i) cen[i] = br[i] * ar[i] + ai[i] * bi[i];
  free(ar); free(ai); free(br); free(bi);
}

void wo_d4c(const double *x, int x_length, int fs, const double *tpos, const double *f0, int nf,
            int fft_size, double threshold, double *aperiodicity) {
  uint32_t rng[4];
  wo_randn_seed(rng);
  int nb = fft_size / 2 + 1;
  for (size_t i = 0; i < (size_t)nf * nb; ++i) aperiodicity[i] = 1.0 - K_TINY;

  int N = (int)pow(2.0, 1.0 + (int)(log(4.0 * fs / 47.0 + 1) / K_LOG2));
  int nap = (int)(dmin(15000.0, fs / 2.0 - 3000.0) / 3000.0);
  int wl = (int)(3000.0 * N / fs) * 2 + 1;
  double *nut = dalloc(wl);
  wo_nuttall(wl, nut);

  /* pass 1: D4CLoveTrain (227-285) */
  double *ap0 = dalloc(nf);
  {
    int M = (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / 40.0 + 1) / K_LOG2));
    int b0 = (int)ceil(100.0 * M / fs), b1 = (int)ceil(4000.0 * M / fs), b2 = (int)ceil(7900.0 * M / fs);
    double *wave = dalloc(M), *re = dalloc(M / 2 + 1), *im = dalloc(M / 2 + 1), *ps = dalloc(M);
    for (int f = 0; f < nf; ++f) {
      if (f0[f] == 0.0) { ap0[f] = 0.0; continue; }
      double cf0 = dmax(f0[f], 40.0);
      int hw = d4c_window(x, x_length, fs, cf0, tpos[f], 2, 3.0, wave, rng);
      for (int i = 2 * hw + 1; i < M; ++i) wave[i] = 0.0;
      wo_rfft(M, wave, re, im);
      for (int i = 0; i <= b0; ++i) ps[i] = 0.0;
      for (int i = b0 + 1; i < M / 2 + 1; ++i) ps[i] = re[i] * re[i] + im[i] * im[i];
      for (int i = b0; i <= b2; ++i) ps[i] += +ps[i - 1];
      ap0[f] = ps[b1] / ps[b2];
    }
    free(wave); free(re); free(im); free(ps);
  }

  /* pass 2: D4CGeneralBody on frames that pass the threshold (293-395) */
  double *coarse = dalloc(nap + 2), *caxis = dalloc(nap + 2), *faxis = dalloc(nb);
  coarse[0] = -60.0; coarse[nap + 1] = -K_TINY;
  for (int i = 0; i <= nap; ++i) caxis[i] = i * 3000.0;
  caxis[nap + 1] = fs / 2.0;
  for (int i = 0; i < nb; ++i) faxis[i] = (double)i * fs / fft_size;
  int H = N / 2;
  double *wave = dalloc(N), *c1 = dalloc(H + 1), *c2 = dalloc(H + 1), *sc = dalloc(H + 1);
  double *sp = dalloc(H + 1), *gd = dalloc(H + 1), *sg = dalloc(H + 1);
  double *re = dalloc(H + 1), *im = dalloc(H + 1), *ps = dalloc(H + 1);
  for (int f = 0; f < nf; ++f) {
    if (f0[f] == 0 || ap0[f] <= threshold) continue;
    double cf0 = dmax(47.0, f0[f]);
    /* static centroid (126-143) */
    d4c_centroid(x, x_length, fs, cf0, N, tpos[f] - 0.25 / cf0, wave, c1, rng);
    d4c_centroid(x, x_length, fs, cf0, N, tpos[f] + 0.25 / cf0, wave, c2, rng);
    for (int i = 0; i <= H; ++i) sc[i] = c1[i] + c2[i];
    wo_dc_correction(sc, cf0, fs, N, sc);
    /* smoothed power spectrum (149-166) */
    memset(wave, 0, sizeof(double) * N);
    d4c_window(x, x_length, fs, cf0, tpos[f], 1, 4.0, wave, rng);
    wo_rfft(N, wave, re, im);
    for (int i = 0; i <= H; ++i) sp[i] = re[i] * re[i] + im[i] * im[i];
    wo_dc_correction(sp, cf0, fs, N, sp);
    wo_linear_smoothing(sp, cf0, fs, N, sp);
    /* static group delay (172-188) */
    for (int i = 0; i <= H; ++i) gd[i] = sc[i] / sp[i];
    wo_linear_smoothing(gd, cf0 / 2.0, fs, N, gd);
    wo_linear_smoothing(gd, cf0, fs, N, sg);
    for (int i = 0; i <= H; ++i) gd[i] -= sg[i];
    /* coarse aperiodicity (194-225) */
    int bnd = wo_round(N * 8.0 / wl), hwl = wl / 2;
    memset(wave, 0, sizeof(double) * N);
    for (int b = 0; b < nap; ++b) {
      int center = (int)(3000.0 * (b + 1) * N / fs);
      for (int j = 0; j <= hwl * 2; ++j) wave[j] = gd[center - hwl + j] * nut[j];
      wo_rfft(N, wave, re, im);
      for (int j = 0; j <= H; ++j) ps[j] = re[j] * re[j] + im[j] * im[j];
      /* ascending sort; any correct sort gives the same array */
      for (int gap = (H + 1) / 2; gap > 0; gap /= 2)
        for (int i = gap; i <= H; ++i) {
          double t = ps[i]; int j = i;
          for (; j >= gap && ps[j - gap] > t; j -= gap) ps[j] = ps[j - gap];
          ps[j] = t;
        }
      for (int j = 1; j <= H; ++j) ps[j] += ps[j - 1];
      coarse[1 + b] = 10 * log10(ps[H - bnd - 1] / ps[H]);
    }
    for (int b = 0; b < nap; ++b) coarse[1 + b] = dmin(0.0, coarse[1 + b] + (cf0 - 100) / 50.0);
    /* spectral representation (330-338) */
    double *row = aperiodicity + (size_t)f * nb;
    wo_interp1(caxis, coarse, nap + 2, faxis, nb, row);
    for (int i = 0; i < nb; ++i) row[i] = pow(10.0, row[i] / 20.0);
  }
  free(nut); free(ap0); free(coarse); free(caxis); free(faxis);
  free(wave); free(c1); free(c2); free(sc); free(sp); free(gd); free(sg); free(re); free(im); free(ps);
}

/* ------------------------------------------------------------------ */
/* codec.cpp: band aperiodicity / mel-cepstral envelope coders          */
/* ------------------------------------------------------------------ */
static const double K_M0 = 1127.01048, K_F0 = 700.0;   /* constantnumbers.h:45-46 */
static double mel_of(double f) { return K_M0 * log(f / K_F0 + 1.0); }        /* codec.cpp:59-61 */
static double freq_of(double m) { return K_F0 * (exp(m / K_M0) - 1.0); }      /* codec.cpp:66-68 */

int wo_number_of_aperiodicities(int fs) {
  return (int)(dmin(15000.0, fs / 2.0 - 3000.0) / 3000.0);
}

void wo_code_aperiodicity(const double *ap, int nf, int fs, int fft_size, double *coded) {
  int nap = wo_number_of_aperiodicities(fs), nb = fft_size / 2 + 1;
  double *axis = dalloc(nap), *lg = dalloc(nb);
  for (int i = 0; i < nap; ++i) axis[i] = 3000.0 * (i + 1.0);
  for (int f = 0; f < nf; ++f) {
    for (int j = 0; j < nb; ++j) lg[j] = 20 * log10(ap[(size_t)f * nb + j]);
    wo_interp1q(0, (double)fs / fft_size, lg, nb, axis, nap, coded + (size_t)f * nap);
  }
  free(axis); free(lg);
}

void wo_decode_aperiodicity(const double *coded, int nf, int fs, int fft_size, double *ap) {
  int nap = wo_number_of_aperiodicities(fs), nb = fft_size / 2 + 1;
  double *faxis = dalloc(nb), *caxis = dalloc(nap + 2), *coarse = dalloc(nap + 2);
  for (int i = 0; i < nb; ++i) faxis[i] = (double)fs / fft_size * i;
  for (int i = 0; i <= nap; ++i) caxis[i] = i * 3000.0;
  caxis[nap + 1] = fs / 2.0;
  coarse[0] = -60.0;
  coarse[nap + 1] = -K_TINY;
  for (int f = 0; f < nf; ++f) {
    double *row = ap + (size_t)f * nb, mean = 0.0;
    for (int j = 0; j < nb; ++j) row[j] = 1.0 - K_TINY;              /* codec.cpp:21-26 */
    /* CheckVUV (codec.cpp:31-41): note that it stores the frame's values BEFORE deciding,
     * so `coarse` always holds the current frame's bands */
    for (int i = 0; i < nap; ++i) { mean += coded[(size_t)f * nap + i]; coarse[i + 1] = coded[(size_t)f * nap + i]; }
    mean /= nap;
    if (mean > -0.5) continue;
    wo_interp1(caxis, coarse, nap + 2, faxis, nb, row);
    for (int j = 0; j < nb; ++j) row[j] = pow(10.0, row[j] / 20.0);
  }
  free(faxis); free(caxis); free(coarse);
}

void wo_code_spectral_envelope(const double *sp, int nf, int fs, int fft_size, int ndim, double *coded) {
  int md = fft_size / 2, nb = md + 1;
  double fm = mel_of(40.0), cm = mel_of(dmin(fs / 2.0, 20000.0));
  double *mel_axis = dalloc(md), *faxis = dalloc(nb), *wr = dalloc(md), *wi = dalloc(md);
  /* the reference's r2c leaves bins above md/2 unwritten (fft.cpp:49-60), so dimensions beyond
   * md/2+1 read stale memory there: meaningful only for ndim <= md/2+1; zero here */
  double *lg = dalloc(nb), *mel = dalloc(md), *wave = dalloc(md), *re = dalloc(md + 1), *im = dalloc(md + 1);
  for (int i = 0; i < md; ++i) {                                       /* codec.cpp:162-180 */
    mel_axis[i] = (cm - fm) * i / md + fm;
    wr[i] = 2.0 * cos(i * K_PI / fft_size) / sqrt(fft_size);
    wi[i] = 2.0 * sin(i * K_PI / fft_size) / sqrt(fft_size);
  }
  wr[0] /= sqrt(2.0);
  for (int i = 0; i <= md; ++i) faxis[i] = mel_of((double)i * fs / fft_size);
  double norm = sqrt(md);
  for (int f = 0; f < nf; ++f) {
    for (int j = 0; j < nb; ++j) lg[j] = log(sp[(size_t)f * nb + j]);
    wo_interp1(faxis, lg, nb, mel_axis, md, mel);                       /* codec.cpp:120-130 */
    for (int i = 0; i < md / 2; ++i) {                                  /* codec.cpp:73-87 */
      wave[i] = mel[i * 2];
      wave[i + md / 2] = mel[md - (i * 2) - 1];
    }
    wo_rfft(md, wave, re, im);
    for (int i = 0; i < ndim; ++i)
      coded[(size_t)f * ndim + i] = (re[i] * wr[i] - im[i] * wi[i]) / norm;
  }
  free(mel_axis); free(faxis); free(wr); free(wi); free(lg); free(mel); free(wave); free(re); free(im);
}

void wo_decode_spectral_envelope(const double *coded, int nf, int fs, int fft_size, int ndim, double *sp) {
  int md = fft_size / 2, nb = md + 1;
  double fm = mel_of(40.0), cm = mel_of(dmin(fs / 2.0, 20000.0));
  double *mel_axis = dalloc(md + 2), *faxis = dalloc(nb), *wr = dalloc(md), *wi = dalloc(md);
  double *mel = dalloc(md + 2), *zr = dalloc(md), *zi = dalloc(md);
  for (int i = 0; i < ndim; ++i) {                                     /* codec.cpp:185-207 */
    wr[i] = cos(i * K_PI / fft_size) * sqrt(fft_size);
    wi[i] = sin(i * K_PI / fft_size) * sqrt(fft_size);
  }
  wr[0] /= sqrt(2.0);
  for (int i = 0; i < md; ++i) mel_axis[i + 1] = freq_of((cm - fm) * i / md + fm);
  mel_axis[0] = 0;
  mel_axis[md + 1] = fs / 2.0;
  for (int i = 0; i < nb; ++i) faxis[i] = (double)i * fs / fft_size;
  double norm = sqrt(md);
  for (int f = 0; f < nf; ++f) {
    const double *c = coded + (size_t)f * ndim;
    for (int i = 0; i < md; ++i) {                                     /* codec.cpp:93-115 */
      zr[i] = i < ndim ? c[i] * wr[i] * norm : 0.0;
      zi[i] = i < ndim ? -c[i] * wi[i] * norm : 0.0;
    }
    /* the reference's backward c2c (fft.cpp:36-45) returns conj(sum in[j] e^{-2 pi i jk/n});
     * only its real part is read here */
    cfft(md, zr, zi, -1);
    for (int i = 0; i < md / 2; ++i) {
      mel[1 + i * 2] = zr[i];
      mel[1 + i * 2 + 1] = zr[md - i - 1];
    }
    mel[0] = mel[1];
    mel[md + 1] = mel[md];
    double *row = sp + (size_t)f * nb;
    wo_interp1(mel_axis, mel, md + 2, faxis, nb, row);                  /* codec.cpp:138-156 */
    for (int j = 0; j < nb; ++j) row[j] = exp(row[j] / md);
  }
  free(mel_axis); free(faxis); free(wr); free(wi); free(mel); free(zr); free(zi);
}

/* ------------------------------------------------------------------ */
/* synthesis.cpp                                                        */
/* ------------------------------------------------------------------ */
/* GetMinimumPhaseSpectrum (common.cpp:182-220).  lg[0..N/2] in; (mr, mi)[0..N/2] out.
 * The reference's forward c2c plan computes FFT(conj(x)) (fft.cpp:62-71). */
static void minimum_phase(const double *lg, int N, double *mr, double *mi) {
  int H = N / 2;
  double *full = dalloc(N), *cr = dalloc(H + 1), *ci = dalloc(H + 1), *zr = dalloc(N), *zi = dalloc(N);
  for (int i = 0; i <= H; ++i) full[i] = lg[i];
  for (int i = H + 1; i < N; ++i) full[i] = lg[N - i];
  wo_rfft(N, full, cr, ci);
  zr[0] = cr[0]; zi[0] = -ci[0];
  for (int i = 1; i < H; ++i) { zr[i] = cr[i] * 2.0; zi[i] = ci[i] * -2.0; }
  zr[H] = cr[H]; zi[H] = -ci[H];
  for (int i = H + 1; i < N; ++i) { zr[i] = 0.0; zi[i] = 0.0; }
  for (int i = 0; i < N; ++i) zi[i] = -zi[i];            /* conj of the input ... */
  cfft(N, zr, zi, -1);                                   /* ... forward transform */
  for (int i = 0; i <= H; ++i) {
    double t = exp(zr[i] / N);
    mr[i] = t * cos(zi[i] / N);
    mi[i] = t * sin(zi[i] / N);
  }
  free(full); free(cr); free(ci); free(zr); free(zi);
}

static void fft_shift(const double *x, int n, double *y) {          /* matlabfunctions.cpp:129-134 */
  for (int i = 0; i < n / 2; ++i) { y[i] = x[i + n / 2]; y[i + n / 2] = x[i]; }
}

static double safe_ap(double x) { return dmax(0.001, dmin(0.999999999999, x)); }   /* common.h:111-113 */

void wo_synthesis(const double *f0, int nf, const double *sp, const double *ap, int fft_size,
                  double frame_period, int fs, int y_length, double *y) {
  const int N = fft_size, H = N / 2, nb = H + 1;
  const double two_pi = 2.0 * K_PI;
  uint32_t rng[4];
  wo_randn_seed(rng);
  for (int i = 0; i < y_length; ++i) y[i] = 0.0;
  double fp = frame_period / 1000.0;
  /* ---- GetTimeBase (synthesis.cpp:225-318) ---- */
  double lowest_f0 = fs / fft_size + 1.0;                /* integer division, as in the reference's call (:361) */
  double *time_axis = dalloc(y_length), *ctime = dalloc(nf + 1), *cf0 = dalloc(nf + 1), *cvuv = dalloc(nf + 1);
  double *if0 = dalloc(y_length), *ivuv = dalloc(y_length);
  for (int i = 0; i < y_length; ++i) time_axis[i] = i / (double)fs;
  for (int i = 0; i < nf; ++i) {
    ctime[i] = i * fp;
    cf0[i] = f0[i] < lowest_f0 ? 0.0 : f0[i];
    cvuv[i] = cf0[i] == 0.0 ? 0.0 : 1.0;
  }
  ctime[nf] = nf * fp;
  cf0[nf] = cf0[nf - 1] * 2 - cf0[nf - 2];
  cvuv[nf] = cvuv[nf - 1] * 2 - cvuv[nf - 2];
  wo_interp1(ctime, cf0, nf + 1, time_axis, y_length, if0);
  wo_interp1(ctime, cvuv, nf + 1, time_axis, y_length, ivuv);
  for (int i = 0; i < y_length; ++i) {
    ivuv[i] = ivuv[i] > 0.5 ? 1.0 : 0.0;
    if0[i] = ivuv[i] == 0.0 ? 500.0 : if0[i];
  }
  double *total = dalloc(y_length), *wrap = dalloc(y_length);
  double *ploc = dalloc(y_length), *pshift = dalloc(y_length);
  int *pidx = ialloc(y_length);
  total[0] = two_pi * if0[0] / fs;
  wrap[0] = fmod(total[0], two_pi);
  for (int i = 1; i < y_length; ++i) {
    total[i] = total[i - 1] + two_pi * if0[i] / fs;
    wrap[i] = fmod(total[i], two_pi);
  }
  int np = 0;
  for (int i = 0; i < y_length - 1; ++i) {
    if (fabs(wrap[i + 1] - wrap[i]) > K_PI) {
      ploc[np] = time_axis[i];
      pidx[np] = i;
      double y1 = wrap[i] - two_pi, y2 = wrap[i + 1];
      pshift[np] = (-y1 / (y2 - y1)) / fs;
      ++np;
    }
  }
  /* ---- GetDCRemover (:320-335) ---- */
  double *rem = dalloc(N), dcsum = 0.0;
  for (int i = 0; i < H; ++i) {
    rem[i] = 0.5 - 0.5 * cos(two_pi * (i + 1.0) / (1.0 + N));
    rem[N - i - 1] = rem[i];
    dcsum += rem[i] * 2.0;
  }
  for (int i = 0; i < H; ++i) { rem[i] /= dcsum; rem[N - i - 1] = rem[i]; }

  double *env = dalloc(nb), *ratio = dalloc(nb), *lg = dalloc(nb), *mr = dalloc(nb), *mi = dalloc(nb);
  double *sr = dalloc(nb), *si = dalloc(nb), *wave = dalloc(N), *per = dalloc(N), *aper = dalloc(N), *tmp = dalloc(N);
  double *nr = dalloc(nb), *ni = dalloc(nb);
  for (int p = 0; p < np; ++p) {
    int nxt = p + 1 < np - 1 ? p + 1 : np - 1;
    int noise_size = pidx[nxt] - pidx[p];
    double vuv = ivuv[pidx[p]], t = ploc[p];
    /* GetSpectralEnvelope / GetAperiodicRatio (:140-180) */
    int ff = imin(nf - 1, (int)floor(t / fp)), fc = imin(nf - 1, (int)ceil(t / fp));
    double w = t / fp - ff;
    for (int i = 0; i < nb; ++i) {
      double a0 = fabs(sp[(size_t)ff * nb + i]), a1 = fabs(sp[(size_t)fc * nb + i]);
      double b0 = safe_ap(ap[(size_t)ff * nb + i]), b1 = safe_ap(ap[(size_t)fc * nb + i]);
      if (ff == fc) { env[i] = a0; ratio[i] = pow(b0, 2.0); }
      else { env[i] = (1.0 - w) * a0 + w * a1; ratio[i] = pow((1.0 - w) * b0 + w * b1, 2.0); }
    }
    /* GetPeriodicResponse (:103-135) */
    if (vuv <= 0.5 || ratio[0] > 0.999) {
      for (int i = 0; i < N; ++i) per[i] = 0.0;
    } else {
      for (int i = 0; i < nb; ++i) lg[i] = log(env[i] * (1.0 - ratio[i]) + K_TINY) / 2.0;
      minimum_phase(lg, N, mr, mi);
      double coef = two_pi * pshift[p] * fs / N;
      for (int i = 0; i < nb; ++i) {                     /* GetSpectrumWithFractionalTimeShift (:86-98) */
        double re2 = cos(coef * i), im2 = sqrt(1.0 - re2 * re2);
        sr[i] = mr[i] * re2 + mi[i] * im2;
        si[i] = mi[i] * re2 - mr[i] * im2;
      }
      wo_irfft_unscaled(N, sr, si, tmp);
      fft_shift(tmp, N, per);
      double dc = 0.0;                                   /* RemoveDCComponent in place (:72-80) */
      for (int i = H; i < N; ++i) dc += per[i];
      for (int i = 0; i < H; ++i) per[i] = -dc * rem[i];
      for (int i = H; i < N; ++i) per[i] -= dc * rem[i];
    }
    /* GetAperiodicResponse (:38-66) with GetNoiseSpectrum (:19-33) */
    double avg = 0.0;
    for (int i = 0; i < noise_size; ++i) { wave[i] = wo_randn(rng); avg += wave[i]; }
    avg /= noise_size;
    for (int i = 0; i < noise_size; ++i) wave[i] -= avg;
    for (int i = noise_size > 0 ? noise_size : 0; i < N; ++i) wave[i] = 0.0;
    wo_rfft(N, wave, nr, ni);
    if (vuv != 0.0) for (int i = 0; i < nb; ++i) lg[i] = log(env[i] * ratio[i]) / 2.0;
    else for (int i = 0; i < nb; ++i) lg[i] = log(env[i]) / 2.0;
    minimum_phase(lg, N, mr, mi);
    for (int i = 0; i < nb; ++i) {
      sr[i] = mr[i] * nr[i] - mi[i] * ni[i];
      si[i] = mr[i] * ni[i] + mi[i] * nr[i];
    }
    wo_irfft_unscaled(N, sr, si, tmp);
    fft_shift(tmp, N, aper);
    /* GetOneFrameSegment (:213-218) and the overlap-add (:376-385) */
    double sq = sqrt((double)noise_size);
    int offset = pidx[p] - H + 1;
    int lo = imax(0, -offset), hi = imin(N, y_length - offset);
    for (int j = lo; j < hi; ++j) y[j + offset] += (per[j] * sq + aper[j]) / N;
  }
  free(time_axis); free(ctime); free(cf0); free(cvuv); free(if0); free(ivuv); free(total); free(wrap);
  free(ploc); free(pshift); free(pidx); free(rem); free(env); free(ratio); free(lg); free(mr); free(mi);
  free(sr); free(si); free(wave); free(per); free(aper); free(tmp); free(nr); free(ni);
}
