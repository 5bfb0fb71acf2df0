// fft.h -- block-cooperative FP64 FFTs staged in LDS (replaces the reference's
// fft_plan_dft_r2c_1d / fft_plan_dft_c2r_1d / fft_execute, src/fft.cpp:26-162,
// for every per-frame transform on the analysis path).
//
// One workgroup owns one transform held entirely in LDS as interleaved complex
// doubles.  Forward transforms are decimation-in-frequency (natural order in,
// digit-reversed order out); inverse transforms are decimation-in-time
// (digit-reversed in, natural out).  Nobody ever permutes: consumers of a forward
// transform read bin k at LDS slot fft_slot(k), producers of an inverse transform
// write bin k there.  Butterflies are radix-16 in registers (radix-8/4/2 for the
// remainder), so the data crosses LDS once per 4 levels; slots are XOR-swizzled so
// that every wide LDS read is bank-conflict free.
//
// Real transforms of length N run as N/2-point complex transforms with the
// usual split/merge step, fused into a caller-supplied functor so |X|^2,
// products of two spectra, lifters ... are formed straight from registers.
#pragma once
#include "devrt.h"
#include "tables.h"

// The LDS slot swizzle (WH_SWZ, below) may differ between translation units (d4c.hip runs the searched map, the rest the
// shipped one: profiles/r05/lds_under_load_ab.txt).  On the GPU every device function is private to its unit anyway; in
// the host emulation (tests/emu: g++, one shared object) these are ordinary inline functions and templates, and the
// linker would merge the units' differing definitions into one -- so everything here lives in an inline namespace named
// after the map.
#ifndef WH_SWZ
#define WH_SWZ 0
#endif
#define WH_FFT_NS2(n) fft_swz##n
#define WH_FFT_NS(n) WH_FFT_NS2(n)
namespace world_hip {
inline namespace WH_FFT_NS(WH_SWZ) {

struct cplx { double re, im; };

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  // one product + one fused multiply-add per component (4 FP64 instructions instead of 6); the fused
  // form rounds once less than the reference's FFT, whose results the path only ever has to meet to 1e-4
  cplx r; r.re = fma(a.re, b.re, -(a.im * b.im)); r.im = fma(a.re, b.im, a.im * b.re); return r;
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { cplx r; r.re = a.re + b.re; r.im = a.im + b.im; return r; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { cplx r; r.re = a.re - b.re; r.im = a.im - b.im; return r; }
__device__ __forceinline__ cplx cconj(cplx a) { a.im = -a.im; return a; }
// Twiddles come from a quarter-wave cosine table staged in LDS by the owning kernel:
// q[r] = cos(2 pi r / 2^lg), r = 0 .. 2^lg/4.  (A butterfly needs three twiddles; from
// the HBM table each was an L2-latency gather on the critical path between barriers.)
struct TwLds {
  const double *q; int lg;
  double fine_c, fine_s;     // cos/sin(2 pi / 2^(lg+1)): one level finer than the table (see twiddle())
  double fine2_c, fine2_s;   // cos/sin(2 pi / 2^(lg+2)): two levels finer
  // Optional DIRECT tables for two levels (stage_direct_twiddles): dir_a[k] = e^{-2 pi i k / 2^dir_a_lg} as twiddle()
  // itself returns it, k < dir_a_n.  A kernel with one butterfly per thread asks every stage for the same few values
  // in every one of its transforms (d4c_frame: 13 transforms a frame, stage levels 8 and 5 -> 32 and 4 distinct
  // twiddles): one 16-byte read replaces two 8-byte reads, the quadrant selects and the sign flips.  0 = none.
  const cplx *dir_a = nullptr, *dir_b = nullptr;
  int dir_a_lg = 0, dir_b_lg = 0;
};

// stage the table for transforms up to 2^lg points; call before the first transform
template <int NT = 0>
__device__ __forceinline__ TwLds stage_twiddles(double *q, int lg, const double2 *global_tw) {
  const int quarter = 1 << (lg - 2);
  const double *dense = reinterpret_cast<const double *>(global_tw + kTwN) + quarter_table_offset(lg);   // tables.h
  // everything the workgroup needs from the tables is requested before anything is waited for: four entries per
  // thread and trip, and the two fine twiddles (one load per trip, then two more behind the barrier, were five trips
  // to L2 in a row at the start of every workgroup)
  const double2 f = global_tw[lg + 1 <= kTwLog2 ? (size_t)1 << (kTwLog2 - lg - 1) : 0];
  const double2 f2 = global_tw[lg + 2 <= kTwLog2 ? (size_t)1 << (kTwLog2 - lg - 2) : 0];
  constexpr int kB = 4;
  const int tid = wg_thread<NT>(), nt = wg_size<NT>();
  for (int i0 = tid; i0 <= quarter; i0 += kB * nt) {
    double v[kB];
#pragma unroll
    for (int k = 0; k < kB; ++k) { const int i = i0 + k * nt; v[k] = dense[i < quarter ? i : quarter]; }
#pragma unroll
    for (int k = 0; k < kB; ++k) if (i0 + k * nt <= quarter) q[i0 + k * nt] = v[k];
  }
  __syncthreads();
  TwLds t; t.q = q; t.lg = lg;
  t.fine_c = f.x; t.fine_s = f.y;
  t.fine2_c = f2.x; t.fine2_s = f2.y;
  return t;
}
__host__ __device__ __forceinline__ size_t twiddle_lds_doubles(int lg) { return (size_t)(1 << (lg - 2)) + 2; }
// e^{-2 pi i k / 2^lg} (forward, sign=-1) or its conjugate (sign=+1), 0 <= k < 2^lg
__device__ __forceinline__ cplx twiddle_table(const TwLds &tw, int k, int lg, int sign) {
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
// A real transform of 2^(tw.lg+1) points needs the table's resolution only for its inner complex
// transform; its merge step asks for lg = tw.lg + 1 and gets w^k = W^(k>>1) * (k odd ? w^1 : 1).  A kernel
// short of LDS may stage a table two levels coarser than its finest request (lg = tw.lg + 2): the two low
// bits of k then select 1, f, f^2 or f^3 with f = e^{2 pi i / 2^lg} (at most two extra products).
__device__ __forceinline__ cplx twiddle(const TwLds &tw, int k, int lg, int sign) {
  if (tw.dir_a_lg != 0 && lg == tw.dir_a_lg) { cplx w = tw.dir_a[k]; if (sign > 0) w.im = -w.im; return w; }
  if (tw.dir_b_lg != 0 && lg == tw.dir_b_lg) { cplx w = tw.dir_b[k]; if (sign > 0) w.im = -w.im; return w; }
  if (lg == tw.lg + 1) {
    cplx a = twiddle_table(tw, k >> 1, tw.lg, sign);
    if (k & 1) { cplx f; f.re = tw.fine_c; f.im = sign > 0 ? tw.fine_s : -tw.fine_s; a = cmul(a, f); }
    return a;
  }
  if (lg == tw.lg + 2) {
    cplx a = twiddle_table(tw, k >> 2, tw.lg, sign);
    if (k & 2) { cplx f; f.re = tw.fine_c; f.im = sign > 0 ? tw.fine_s : -tw.fine_s; a = cmul(a, f); }
    if (k & 1) { cplx f; f.re = tw.fine2_c; f.im = sign > 0 ? tw.fine2_s : -tw.fine2_s; a = cmul(a, f); }
    return a;
  }
  return twiddle_table(tw, k, lg, sign);
}
// Fill direct tables for levels lg_a (n_a entries at area) and lg_b (n_b entries behind them) from the staged table -- the
// values twiddle() returns, bit for bit -- and return a TwLds that serves those levels from them.  The caller's next
// barrier publishes the tables (none is added here).  area: (n_a + n_b) cplx of LDS, 16-byte aligned.
template <int NT = 0>
__device__ __forceinline__ TwLds stage_direct_twiddles(const TwLds &tw, cplx *area, int lg_a, int n_a, int lg_b, int n_b) {
  const int tid = wg_thread<NT>();
  if (tid < n_a) area[tid] = twiddle(tw, tid, lg_a, -1);
  else if (tid < n_a + n_b) area[tid] = twiddle(tw, tid - n_a, lg_b, -1);
  TwLds t = tw;
  t.dir_a = area; t.dir_a_lg = lg_a;
  t.dir_b = area + n_a; t.dir_b_lg = lg_b;
  return t;
}

// ---------------------------------------------------------------------------
// Mixed-radix plan: radix-16 stages, then one radix-8/4/2 stage for the remainder.
// A radix-R butterfly is evaluated entirely in registers, so a 2048-point
// transform crosses LDS 3 times (16 x 16 x 8) instead of 6 (radix-4) or 11.
struct FftPlan {
  int lg, ns;
  unsigned stages;                                   // log2(radix) of stage s in nibble s (no arrays: stays in registers)
  constexpr __host__ __device__ int rl(int s) const { return (int)((stages >> (4 * s)) & 15u); }
};
__host__ __device__ __forceinline__ FftPlan make_plan(int lg) {
  FftPlan p; p.lg = lg; p.ns = 0; p.stages = 0;
  int r = lg;
  while (r >= 4) { p.stages |= 4u << (4 * p.ns++); r -= 4; }
  if (r) p.stages |= (unsigned)r << (4 * p.ns++);
  return p;
}
// the same with smaller butterflies (max_lr = 3: radix 8): twice the butterflies per stage (every thread of a 256-thread
// workgroup stays busy on a 2048-point transform) and half the registers, for one more LDS pass
constexpr __host__ __device__ FftPlan make_plan_max(int lg, int max_lr) {
  FftPlan p{lg, 0, 0u};
  int r = lg;
  while (r >= max_lr) { p.stages |= (unsigned)max_lr << (4 * p.ns++); r -= max_lr; }
  if (r) p.stages |= (unsigned)r << (4 * p.ns++);
  return p;
}
__host__ __device__ __forceinline__ FftPlan make_plan_r8(int lg) { return make_plan_max(lg, 3); }
// LDS slot swizzle: XOR the 16-byte slot index with bits 4..7 of itself.  With the
// butterfly->thread mapping below every ds_read_b128 of every stage of every plan
// (256..4096 points) is conflict-free in the gfx950 bank model (MI355X_MICROARCH.md
// section LDS; brute-forced in DESIGN.md), at zero cost in LDS capacity.
// The swizzle is linear over GF(2) (shift, mask, xor), so for index fields that do not overlap
//     swz(base | field) = swz(base) ^ swz(field):
// a butterfly computes swz(base) once and reaches its R elements with one XOR each against constants that are
// uniform over the workgroup (scalar registers) -- the address arithmetic was two thirds of the FFT kernels'
// VALU instructions (profiles/r02, SQ_INSTS_VALU against the FP64 counters).
__device__ __forceinline__ int swz(int i) {
#if WH_SWZ == 1
  // candidate of tools/lds_swizzle_search.py for the radix-8 plans: low nibble ^= rotl1(bits 4..7) ^ rotl3(bits 8..11)
  return i ^ (((i >> 3) & 14) | ((i >> 7) & 1)) ^ (((i >> 9) & 7) | ((i >> 5) & 8));
#elif WH_SWZ == 2
  // low nibble ^= rotl1(bits 4..7) ^ (bits 8..11 ^ rotl2(bits 8..11))
  return i ^ (((i >> 3) & 14) | ((i >> 7) & 1)) ^ ((i >> 8) & 15) ^ (((i >> 6) & 12) | ((i >> 10) & 3));
#elif WH_SWZ == 3
  return i;                                       // (no swizzle at all: what the conflicts the others avoid would cost)
#else
  return i ^ ((i >> 4) & 15);
#endif
}
// slot that holds bin k after the forward (DIF) transform = slot the inverse (DIT)
// transform expects bin k in: the digits of k, least significant first, select
// nested blocks.
__device__ __forceinline__ int fft_slot(const FftPlan &p, int k) {
  int pos = 0, rem = p.lg;
  for (int s = 0; s < p.ns; ++s) {
    const int rl = p.rl(s);
    rem -= rl;
    pos += (k & ((1 << rl) - 1)) << rem;
    k >>= rl;
  }
  return swz(pos);
}

// inverse of fft_slot: which bin lives in physical slot `slot` after the forward transform
__device__ __forceinline__ int fft_bin_of_slot(const FftPlan &p, int slot) {
  const int pos = swz(slot);                         // the swizzle is an involution
  int k = 0, rem = p.lg, shift = 0;
  for (int s = 0; s < p.ns; ++s) {
    const int rl = p.rl(s);
    rem -= rl;
    k |= ((pos >> rem) & ((1 << rl) - 1)) << shift;
    shift += rl;
  }
  return k;
}

// ---- in-register DFTs, natural order in and out; FWD: e^{-i..}, else e^{+i..} -----
template <bool FWD> __device__ __forceinline__ cplx mul_i4(cplx a) {          // a * W4 = a * (-/+ i)
  cplx r; if (FWD) { r.re = a.im; r.im = -a.re; } else { r.re = -a.im; r.im = a.re; } return r;
}
template <bool FWD> __device__ __forceinline__ void dft4(cplx &a0, cplx &a1, cplx &a2, cplx &a3) {
  cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_i4<FWD>(csub(a1, a3));
  a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}
// a * W16^m (m = 0..15) with the trivial powers special-cased
template <bool FWD, int M> __device__ __forceinline__ cplx mul_w16(cplx a) {
  constexpr double c1 = 0.92387953251128674, s1 = 0.38268343236508977, h = 0.70710678118654752;
  constexpr int m = M & 15;
  if (m == 0) return a;
  if (m == 4) return mul_i4<FWD>(a);
  if (m == 8) { cplx r; r.re = -a.re; r.im = -a.im; return r; }
  if (m == 12) { cplx r = mul_i4<FWD>(a); r.re = -r.re; r.im = -r.im; return r; }
  constexpr double ct[16] = {1, c1, h, s1, 0, -s1, -h, -c1, -1, -c1, -h, -s1, 0, s1, h, c1};
  constexpr double st[16] = {0, s1, h, c1, 1, c1, h, s1, 0, -s1, -h, -c1, -1, -c1, -h, -s1};
  cplx w; w.re = ct[m]; w.im = FWD ? -st[m] : st[m];
  return cmul(a, w);
}
template <bool FWD> __device__ __forceinline__ void dft2(cplx *a) {
  cplx t = a[0]; a[0] = cadd(t, a[1]); a[1] = csub(t, a[1]);
}
template <bool FWD> __device__ __forceinline__ void dft4v(cplx *a) { dft4<FWD>(a[0], a[1], a[2], a[3]); }
template <bool FWD> __device__ __forceinline__ void dft8(cplx *a) {
  // 8 = 2 x 4: n = 4 n1 + n2, k = k1 + 2 k2
  cplx u0[4], u1[4];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) { u0[n2] = cadd(a[n2], a[n2 + 4]); u1[n2] = csub(a[n2], a[n2 + 4]); }
  u1[1] = mul_w16<FWD, 2>(u1[1]); u1[2] = mul_w16<FWD, 4>(u1[2]); u1[3] = mul_w16<FWD, 6>(u1[3]);   // W8^{n2}
  dft4<FWD>(u0[0], u0[1], u0[2], u0[3]);
  dft4<FWD>(u1[0], u1[1], u1[2], u1[3]);
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) { a[2 * k2] = u0[k2]; a[2 * k2 + 1] = u1[k2]; }
}
// dft8 of (a0, a1, 0, 0, 0, 0, 0, 0): X[k] = a0 + a1 W8^k.  The operations are the ones dft8 above performs on those
// inputs once its additions of zero are dropped -- ONE product (a1 W8) and eight complex additions -- so the results
// are the same bits (a zero result may differ in sign) for a fifth of the instructions.
template <bool FWD> __device__ __forceinline__ void dft8_head2(cplx a0, cplx a1, cplx *x) {
  const cplx c = mul_w16<FWD, 2>(a1);                         // a1 W8
  const cplx ia = mul_i4<FWD>(a1), ic = mul_i4<FWD>(c);       // a1 W8^2, a1 W8^3
  x[0] = cadd(a0, a1); x[1] = cadd(a0, c); x[2] = cadd(a0, ia); x[3] = cadd(a0, ic);
  x[4] = csub(a0, a1); x[5] = csub(a0, c); x[6] = csub(a0, ia); x[7] = csub(a0, ic);
}
template <bool FWD> __device__ __forceinline__ void dft16(cplx *a) {
  // 16 = 4 x 4: n = 4 n1 + n2, k = k1 + 4 k2
  cplx u[4][4];                                   // u[n2][k1]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    u[n2][0] = a[n2]; u[n2][1] = a[n2 + 4]; u[n2][2] = a[n2 + 8]; u[n2][3] = a[n2 + 12];
    dft4<FWD>(u[n2][0], u[n2][1], u[n2][2], u[n2][3]);
  }
  u[1][1] = mul_w16<FWD, 1>(u[1][1]); u[1][2] = mul_w16<FWD, 2>(u[1][2]); u[1][3] = mul_w16<FWD, 3>(u[1][3]);
  u[2][1] = mul_w16<FWD, 2>(u[2][1]); u[2][2] = mul_w16<FWD, 4>(u[2][2]); u[2][3] = mul_w16<FWD, 6>(u[2][3]);
  u[3][1] = mul_w16<FWD, 3>(u[3][1]); u[3][2] = mul_w16<FWD, 6>(u[3][2]); u[3][3] = mul_w16<FWD, 9>(u[3][3]);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    dft4<FWD>(u[0][k1], u[1][k1], u[2][k1], u[3][k1]);
    a[k1] = u[0][k1]; a[k1 + 4] = u[1][k1]; a[k1 + 8] = u[2][k1]; a[k1 + 12] = u[3][k1];
  }
}
template <bool FWD, int LR> __device__ __forceinline__ void dft_reg(cplx *a) {
  if (LR == 1) dft2<FWD>(a);
  else if (LR == 2) dft4v<FWD>(a);
  else if (LR == 3) dft8<FWD>(a);
  else dft16<FWD>(a);
}
// a[k] *= w^k for k = 1 .. R-1.  Powers are formed with product depth <= 4 and applied
// as soon as they exist, so at most R/2 twiddles are live next to the R data values.
template <int LR> __device__ __forceinline__ void mul_powers(cplx *a, cplx w1) {
  constexpr int R = 1 << LR;
  if (R == 2) { a[1] = cmul(a[1], w1); return; }
  cplx w2 = cmul(w1, w1), w3 = cmul(w2, w1);
  a[1] = cmul(a[1], w1); a[2] = cmul(a[2], w2); a[3] = cmul(a[3], w3);
  if (R == 4) return;
  cplx w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
  a[4 % R] = cmul(a[4 % R], w4); a[5 % R] = cmul(a[5 % R], w5); a[6 % R] = cmul(a[6 % R], w6); a[7 % R] = cmul(a[7 % R], w7);
  if (R == 8) return;
  const cplx w8 = cmul(w4, w4);
  a[8 % R] = cmul(a[8 % R], w8);
  a[9 % R] = cmul(a[9 % R], cmul(w8, w1)); a[10 % R] = cmul(a[10 % R], cmul(w8, w2)); a[11 % R] = cmul(a[11 % R], cmul(w8, w3));
  a[12 % R] = cmul(a[12 % R], cmul(w8, w4)); a[13 % R] = cmul(a[13 % R], cmul(w8, w5)); a[14 % R] = cmul(a[14 % R], cmul(w8, w6));
  a[15 % R] = cmul(a[15 % R], cmul(w8, w7));
}

// one decimation-in-frequency stage: sub-transforms of length 2^lev split R ways
// Butterfly -> thread maps.  Plain: b = tid, tid + T, ...  Wave-region (WR = log2 of the butterflies a thread owns):
// the 64 2^WR butterflies of wavefront w are the CONSECUTIVE ones [64 2^WR w, 64 2^WR (w + 1)), so a stage whose
// butterflies span <= 64 elements keeps every wavefront inside one contiguous slice of the transform -- the same
// slice in every such stage, which therefore need no workgroup barrier between them (DifStages / DitStages).
template <int NT, int WR> struct BflyMap {
  static __device__ __forceinline__ int first() {
    const int t = wg_thread<NT>();
    if constexpr (WR >= 0) return ((t >> 6) << (6 + WR)) + (t & 63);
    else return t;
  }
  static __device__ __forceinline__ int step() { if constexpr (WR >= 0) return 64; else return wg_size<NT>(); }
  static __device__ __forceinline__ int count(int nbf) {       // butterflies of this thread (plain: loop bound on b)
    (void)nbf;
    if constexpr (WR >= 0) return 1 << WR; else return 0;
  }
};
// butterflies of one thread whose inputs are read together (dif_stage, dit_stage): the 2^WR of a wave-region stage, a
// pair for other small radices, one for the radix-8 / 16 stages (16 .. 32 registers of operands each)
template <int LR, int WR> __host__ __device__ constexpr int fft_gather() { return LR > 2 ? 1 : (WR >= 0 ? (1 << WR) : 2); }
template <int LR, int NT = 0, int WR = -1> __device__ __forceinline__ void dif_stage(cplx *z, int lg, int lev, const TwLds &tw) {
  constexpr int R = 1 << LR;
  const int sh = lev - LR, q = 1 << sh, nbf = 1 << (lg - LR);
  int c[R];                                          // swz(r q): uniform
#pragma unroll
  for (int r = 0; r < R; ++r) c[r] = swz(r << sh);
  const int b_end = WR >= 0 ? BflyMap<NT, WR>::first() + (64 << (WR >= 0 ? WR : 0)) : nbf;
  // Small butterflies come several to a thread (the remainder stage of a plan: 2 radix-4 or 4 radix-2 per thread): their
  // inputs are ALL read before any output is stored.  Written one butterfly after the other, the second one's reads
  // wait behind the first one's stores -- the compiler cannot know they do not alias -- and a stage is two or four
  // trips through the LDS pipe instead of one: ~750 cycles each in a CU whose other workgroups run their transforms
  // (tools/trace_batch.py; tools/isa_audit.py lists such loops).
  constexpr int kGather = fft_gather<LR, WR>();
  if constexpr (kGather > 1) {
    const int step = BflyMap<NT, WR>::step();
    for (int b0 = BflyMap<NT, WR>::first(); b0 < b_end; b0 += kGather * step) {
      cplx a[kGather][R];
      int s0[kGather], j[kGather];
#pragma unroll
      for (int t = 0; t < kGather; ++t) {
        const int b = b0 + t * step < b_end ? b0 + t * step : b0;       // (a missing butterfly re-reads the first: never stored)
        j[t] = b & (q - 1);
        s0[t] = swz(((b >> sh) << lev) + j[t]);
#pragma unroll
        for (int r = 0; r < R; ++r) a[t][r] = z[s0[t] ^ c[r]];
      }
#pragma unroll
      for (int t = 0; t < kGather; ++t) {
        dft_reg<true, LR>(a[t]);
        if (q > 1) mul_powers<LR>(a[t], twiddle(tw, j[t], lev, -1));
        if (b0 + t * step < b_end) {
#pragma unroll
          for (int k = 0; k < R; ++k) z[s0[t] ^ c[k]] = a[t][k];
        }
      }
    }
    return;
  }
  for (int b = BflyMap<NT, WR>::first(); b < b_end; b += BflyMap<NT, WR>::step()) {
    const int j = b & (q - 1);
    const int s0 = swz(((b >> sh) << lev) + j);      // bits [sh, lev) of the base index are clear
    cplx a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = z[s0 ^ c[r]];
    dft_reg<true, LR>(a);
    if (q > 1) mul_powers<LR>(a, twiddle(tw, j, lev, -1));
#pragma unroll
    for (int k = 0; k < R; ++k) z[s0 ^ c[k]] = a[k];
  }
}
// the first DIF stage with its inputs taken from src(n) (element n of the transform) instead of LDS:
// a caller whose input is computed on the fly, or mostly zero, skips the staging pass and its barrier
template <int LR, class Src>
__device__ __forceinline__ void dif_first_stage(cplx *z, int lg, const TwLds &tw, Src src) {
  constexpr int R = 1 << LR;
  const int sh = lg - LR, q = 1 << sh;
  int c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) c[r] = swz(r << sh);
  for (int j = threadIdx.x; j < q; j += blockDim.x) {
    cplx a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = src(j + r * q);
    dft_reg<true, LR>(a);
    if (q > 1) mul_powers<LR>(a, twiddle(tw, j, lg, -1));
    const int s0 = swz(j);
#pragma unroll
    for (int k = 0; k < R; ++k) z[s0 ^ c[k]] = a[k];
  }
}
// one decimation-in-time stage: R finished sub-transforms of length 2^done are merged
// epi(n, v): what is stored for output element n (the plain transform stores v); a caller's last stage can fold an
// element-wise pass over the result into the transform's own stores
struct FftNoEpilogue { __device__ __forceinline__ cplx operator()(int, cplx v) const { return v; } };
template <int LR, int NT = 0, int WR = -1, class Epi = FftNoEpilogue>
__device__ __forceinline__ void dit_stage(cplx *z, int lg, int done, const TwLds &tw, Epi epi = Epi()) {
  constexpr int R = 1 << LR;
  const int q = 1 << done, L = done + LR, nbf = 1 << (lg - LR);
  int c[R];
#pragma unroll
  for (int r = 0; r < R; ++r) c[r] = swz(r << done);
  const int b_end = WR >= 0 ? BflyMap<NT, WR>::first() + (64 << (WR >= 0 ? WR : 0)) : nbf;
  constexpr int kGather = fft_gather<LR, WR>();        // dif_stage: a thread's small butterflies are read together
  if constexpr (kGather > 1) {
    const int step = BflyMap<NT, WR>::step();
    for (int b0 = BflyMap<NT, WR>::first(); b0 < b_end; b0 += kGather * step) {
      cplx a[kGather][R];
      int s0[kGather], j[kGather], base[kGather];
#pragma unroll
      for (int t = 0; t < kGather; ++t) {
        const int b = b0 + t * step < b_end ? b0 + t * step : b0;
        j[t] = b & (q - 1);
        base[t] = ((b >> done) << L) + j[t];
        s0[t] = swz(base[t]);
#pragma unroll
        for (int r = 0; r < R; ++r) a[t][r] = z[s0[t] ^ c[r]];
      }
#pragma unroll
      for (int t = 0; t < kGather; ++t) {
        if (q > 1) mul_powers<LR>(a[t], twiddle(tw, j[t], L, +1));
        dft_reg<false, LR>(a[t]);
        if (b0 + t * step < b_end) {
#pragma unroll
          for (int k = 0; k < R; ++k) z[s0[t] ^ c[k]] = epi(base[t] + (k << done), a[t][k]);
        }
      }
    }
    return;
  }
  for (int b = BflyMap<NT, WR>::first(); b < b_end; b += BflyMap<NT, WR>::step()) {
    const int j = b & (q - 1);
    const int base = ((b >> done) << L) + j;         // bits [done, L) of the base index are clear
    const int s0 = swz(base);
    cplx a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = z[s0 ^ c[r]];
    if (q > 1) mul_powers<LR>(a, twiddle(tw, j, L, +1));
    dft_reg<false, LR>(a);
#pragma unroll
    for (int k = 0; k < R; ++k) z[s0 ^ c[k]] = epi(base + (k << done), a[k]);
  }
}

// ---- forward: element n at slot swz(n) in -> bin k at slot fft_slot(plan, k) out ----
// MAXLR: the plan is known to hold no stage above radix 2^MAXLR (make_plan_max), so the larger
// butterflies -- and their register footprint -- are compiled out of the instantiation.
template <int MAXLR = 4>
__device__ __forceinline__ void block_cfft_dif(cplx *z, const FftPlan &p, const TwLds &tw) {
  int lev = p.lg;
  for (int s = 0; s < p.ns; ++s) {
    __syncthreads();
    switch (p.rl(s)) {
      case 4: if (MAXLR >= 4) dif_stage<4>(z, p.lg, lev, tw); break;
      case 3: if (MAXLR >= 3) dif_stage<3>(z, p.lg, lev, tw); break;
      case 2: dif_stage<2>(z, p.lg, lev, tw); break;
      default: dif_stage<1>(z, p.lg, lev, tw); break;
    }
    lev -= p.rl(s);
  }
  __syncthreads();
}

template <int MAXLR = 4, class Src>
__device__ __forceinline__ void block_cfft_dif_from(cplx *z, const FftPlan &p, const TwLds &tw, Src src) {
  __syncthreads();                                   // earlier readers of z are done
  switch (p.rl(0)) {
    case 4: if (MAXLR >= 4) dif_first_stage<4>(z, p.lg, tw, src); break;
    case 3: if (MAXLR >= 3) dif_first_stage<3>(z, p.lg, tw, src); break;
    case 2: dif_first_stage<2>(z, p.lg, tw, src); break;
    default: dif_first_stage<1>(z, p.lg, tw, src); break;
  }
  int lev = p.lg - p.rl(0);
  for (int s = 1; s < p.ns; ++s) {
    __syncthreads();
    switch (p.rl(s)) {
      case 4: if (MAXLR >= 4) dif_stage<4>(z, p.lg, lev, tw); break;
      case 3: if (MAXLR >= 3) dif_stage<3>(z, p.lg, lev, tw); break;
      case 2: dif_stage<2>(z, p.lg, lev, tw); break;
      default: dif_stage<1>(z, p.lg, lev, tw); break;
    }
    lev -= p.rl(s);
  }
  __syncthreads();
}

// ---- the same with the transform length known at compile time ---------------------------------------
// Kernels whose shape fixes the length (d4c_frame) instantiate the stages by recursion: no stage loop, no
// radix switch, every stride a constant -- and a constexpr plan makes the digit reversals of the merge
// steps (fft_slot / fft_bin_of_slot) straight-line bit arithmetic.
// One radix-2^MAXLR butterfly per thread (NT = 2^(LG - MAXLR), whole wavefronts): a stage whose sub-transforms are <= 64
// 2^MAXLR elements long then keeps wavefront w inside elements [64 2^MAXLR w, 64 2^MAXLR (w + 1)) -- the slice its own
// lanes wrote in the stage before, if that one was of the same kind -- and the workgroup barrier between the two becomes
// a wave-level fence.  The remainder stage (smaller radix, several butterflies per thread) uses the wave-region map.
// A 2048-point complex transform on 256 threads: 5 barriers -> 2 + the closing one.
template <int LG, int MAXLR, int NT> struct FftWaveLocal {
  static constexpr bool one_per_thread = NT >= 64 && NT % 64 == 0 && LG > MAXLR && (1 << (LG - MAXLR)) == NT;
  // DIF stage at level LEV (sub-transform length 2^LEV): the stage before it had butterflies spanning 2^LEV elements
  static constexpr bool dif_local(int lev) { return one_per_thread && lev <= 6; }
  // DIT stage that merges sub-transforms of length 2^DONE: its own butterflies span 2^DONE * R elements
  static constexpr bool dit_local(int done, int lr) { return one_per_thread && done > 0 && done + lr <= 6 + MAXLR; }
};
template <int LG, int MAXLR, int LEV, int NT = 0> struct DifStages {
  static __device__ __forceinline__ void run(cplx *z, const TwLds &tw) {
    constexpr int LR = LEV >= MAXLR ? MAXLR : LEV;
    constexpr bool local = FftWaveLocal<LG, MAXLR, NT>::dif_local(LEV);
    if constexpr (local) wave_sync(); else __syncthreads();
    // the remainder stage of a one-butterfly-per-thread plan owns 2^(MAXLR - LR) butterflies per thread: wave regions
    if constexpr (FftWaveLocal<LG, MAXLR, NT>::one_per_thread && LR < MAXLR) dif_stage<LR, NT, MAXLR - LR>(z, LG, LEV, tw);
    else dif_stage<LR, NT>(z, LG, LEV, tw);
    if constexpr (LEV - LR > 0) DifStages<LG, MAXLR, LEV - LR, NT>::run(z, tw);
  }
};
template <int LG, int MAXLR, int NT = 0>
__device__ __forceinline__ void block_cfft_dif_static(cplx *z, const TwLds &tw) {
  DifStages<LG, MAXLR, LG, NT>::run(z, tw);
  __syncthreads();
}
template <int LG, int MAXLR, class Src>
__device__ __forceinline__ void block_cfft_dif_from_static(cplx *z, const TwLds &tw, Src src) {
  constexpr int LR = LG >= MAXLR ? MAXLR : LG;
  __syncthreads();                                   // earlier readers of z are done
  dif_first_stage<LR>(z, LG, tw, src);
  if constexpr (LG - LR > 0) DifStages<LG, MAXLR, LG - LR>::run(z, tw);
  __syncthreads();
}
// inverse: the plan's stages in reverse order -- the remainder stage (if any) first, then the MAXLR ones
template <int LG, int MAXLR, int DONE, int NT = 0> struct DitStages {
  template <class Epi> static __device__ __forceinline__ void run(cplx *z, const TwLds &tw, Epi epi) {
    constexpr int LR = (DONE == 0 && LG % MAXLR != 0) ? LG % MAXLR : MAXLR;
    constexpr bool local = FftWaveLocal<LG, MAXLR, NT>::dit_local(DONE, LR);
    constexpr bool last = DONE + LR >= LG;
    if constexpr (local) wave_sync(); else __syncthreads();
    if constexpr (last) {                              // the epilogue belongs to the stage that produces the result
      if constexpr (FftWaveLocal<LG, MAXLR, NT>::one_per_thread && LR < MAXLR) dit_stage<LR, NT, MAXLR - LR, Epi>(z, LG, DONE, tw, epi);
      else dit_stage<LR, NT, -1, Epi>(z, LG, DONE, tw, epi);
    } else {
      if constexpr (FftWaveLocal<LG, MAXLR, NT>::one_per_thread && LR < MAXLR) dit_stage<LR, NT, MAXLR - LR>(z, LG, DONE, tw);
      else dit_stage<LR, NT>(z, LG, DONE, tw);
      DitStages<LG, MAXLR, DONE + LR, NT>::run(z, tw, epi);
    }
  }
};
template <int LG, int MAXLR, int NT = 0, class Epi = FftNoEpilogue>
__device__ __forceinline__ void block_cfft_dit_static(cplx *z, const TwLds &tw, Epi epi = Epi()) {
  DitStages<LG, MAXLR, 0, NT>::run(z, tw, epi);
  __syncthreads();
}

// ---- inverse (unscaled): bin k at slot fft_slot(plan, k) in -> element n at slot swz(n) out ----
template <int MAXLR = 4>
__device__ __forceinline__ void block_cfft_dit(cplx *z, const FftPlan &p, const TwLds &tw) {
  int done = 0;
  for (int s = p.ns - 1; s >= 0; --s) {
    __syncthreads();
    switch (p.rl(s)) {
      case 4: if (MAXLR >= 4) dit_stage<4>(z, p.lg, done, tw); break;
      case 3: if (MAXLR >= 3) dit_stage<3>(z, p.lg, done, tw); break;
      case 2: dit_stage<2>(z, p.lg, done, tw); break;
      default: dit_stage<1>(z, p.lg, done, tw); break;
    }
    done += p.rl(s);
  }
  __syncthreads();
}

// ---- real forward transform of length N = 2^lgn -------------------------------
// Input: real sample n stored as the re/im halves of complex slot swz(n / 2)
// (use rfft_in() to address it).  After the call the buffer holds scrambled data;
// emit(k, Xre, Xim) has been called once for every k in [0, N/2] (same semantics as
// the reference's r2c: X[k] = sum x[n] e^{-2 pi i k n / N}, imaginary part of
// DC/Nyquist = 0); every thread receives at most 2 ceil((N/4+1)/T) bins.
__device__ __forceinline__ double &rfft_in(cplx *z, int n) {
  cplx &c = z[swz(n >> 1)];
  return (n & 1) ? c.im : c.re;
}
template <class Emit>
__device__ __forceinline__ void rfft_merge(cplx *z, int lgn, const FftPlan &plan, const TwLds &tw, Emit emit);

// LGN > 0: the transform length 2^LGN is a compile-time constant of the caller (static stages, constexpr plan);
// LGN = 0: taken from `lgn` at run time.
template <int MAXLR = 4, int LGN = 0, class Emit>
__device__ __forceinline__ void block_rfft(cplx *z, int lgn, const TwLds &tw, Emit emit) {
  if constexpr (LGN > 0) {
    constexpr FftPlan plan = make_plan_max(LGN - 1, MAXLR);
    block_cfft_dif_static<LGN - 1, MAXLR>(z, tw);
    rfft_merge(z, LGN, plan, tw, emit);
  } else {
    const FftPlan plan = make_plan_max(lgn - 1, MAXLR);
    block_cfft_dif<MAXLR>(z, plan, tw);
    rfft_merge(z, lgn, plan, tw, emit);
  }
}
// same transform with the packed input supplied by src(n) = (x[2n], x[2n+1]), n < N/2: z is pure workspace
template <int MAXLR = 4, int LGN = 0, class Src, class Emit>
__device__ __forceinline__ void block_rfft_from(cplx *z, int lgn, const TwLds &tw, Src src, Emit emit) {
  if constexpr (LGN > 0) {
    constexpr FftPlan plan = make_plan_max(LGN - 1, MAXLR);
    block_cfft_dif_from_static<LGN - 1, MAXLR>(z, tw, src);
    rfft_merge(z, LGN, plan, tw, emit);
  } else {
    const FftPlan plan = make_plan_max(lgn - 1, MAXLR);
    block_cfft_dif_from<MAXLR>(z, plan, tw, src);
    rfft_merge(z, lgn, plan, tw, emit);
  }
}
template <class Emit>
__device__ __forceinline__ void rfft_merge(cplx *z, int lgn, const FftPlan &plan, const TwLds &tw, Emit emit) {
  const int lgh = lgn - 1, h = 1 << lgh, q = h >> 1;
  // Bins are produced in conjugate pairs: with e / o the even / odd sub-spectra at bin k,
  //   X[k] = e + w_k o   and   X[h-k] = conj(e - w_k o)          (w_{h-k} = -conj(w_k)),
  // so one pair of LDS reads, one twiddle and one complex product yield two bins.  Items walk
  // the PHYSICAL slots whose bin is below h/2 in lane order (conflict-free wide reads): those
  // are the slots whose last-stage digit has its top bit clear.  Item h/2 is the self-paired
  // bin k = h/2.  Every bin in [0, h] is emitted exactly once, <= 2 ceil((h/2+1)/T) per thread.
  const int top_bit = plan.rl(plan.ns - 1) - 1;
  struct Pair { int k; double ar, ai, br, bi; };      // bins k and h-k (k = 0: DC and Nyquist; k = h/2: a only)
  auto item = [&](int it) {
      Pair r;
      if (it < q) {
        const int pos = ((it >> top_bit) << (top_bit + 1)) | (it & ((1 << top_bit) - 1));
        const int slot = swz(pos);
        const int k = fft_bin_of_slot(plan, slot);
        const cplx za = z[slot];
        r.k = k;
        if (k == 0) {
          r.ar = za.re + za.im; r.ai = 0.0; r.br = za.re - za.im; r.bi = 0.0;
        } else {
          const cplx zb = z[fft_slot(plan, h - k)];
          cplx e, o;
          e.re = 0.5 * (za.re + zb.re); e.im = 0.5 * (za.im - zb.im);
          o.re = 0.5 * (za.im + zb.im); o.im = -0.5 * (za.re - zb.re);
          const cplx ow = cmul(o, twiddle(tw, k, lgn, -1));
          r.ar = e.re + ow.re; r.ai = e.im + ow.im; r.br = e.re - ow.re; r.bi = ow.im - e.im;
        }
      } else {
        const cplx za = z[fft_slot(plan, q)];          // k = h/2: w = -i, X = conj(z)
        r.k = q; r.ar = za.re; r.ai = -za.im; r.br = 0.0; r.bi = 0.0;
      }
      return r;
    };
  auto out = [&](int, Pair r) {
      emit(r.k, r.ar, r.ai);
      if (r.k != q) emit(h - r.k, r.br, r.bi);
    };
  // The q paired items divide evenly among the threads; item q, the odd one out, is read by the LAST thread before
  // anybody stores (as item q + 1 of the loop it was a trip of its own for thread 0 -- a read waiting behind the first
  // trip's stores -- with the whole workgroup at the closing barrier meanwhile).
  const bool odd_one = (int)threadIdx.x == (int)blockDim.x - 1;
  Pair last;
  if (odd_one) last = item(q);
  block_map<2, Pair>(q, item, out);
  if (odd_one) out(q, last);
  __syncthreads();
}

// The merge step for callers that keep their bins in REGISTERS.  A thread's items are it = tid + m T,
// m = 0 .. KITEMS-1 (KITEMS >= ceil((N/4 + 1) / T)); the loop over m is unrolled, so `m` is a compile-time
// constant inside emit and arrays indexed by it stay in registers.  Item it owns the conjugate pair of bins
// (k, h-k) with k = it IN NATURAL ORDER -- it = 0: (DC, Nyquist); it = h/2: the single bin h/2 -- so
// consecutive lanes own consecutive bins: whatever the caller later exchanges through LDS by bin index
// (prefix-sum segments, spectrum slices) is conflict-free, at the price of digit-reversed (2..4-way
// conflicting) reads of the transform here.  rfft_merge above makes the opposite choice.
//   emit(m, k, Xre[k], Xim[k], paired, Xre[h-k], Xim[h-k])        paired = false only for it = h/2
// a * W16^m, m known after unrolling (folds to one of the mul_w16 cases)
__device__ __forceinline__ cplx mul_w16_fwd(cplx a, int m) {
  switch (m & 15) {
    case 0: return a;
    case 1: return mul_w16<true, 1>(a);
    case 2: return mul_w16<true, 2>(a);
    case 3: return mul_w16<true, 3>(a);
    case 4: return mul_w16<true, 4>(a);
    case 5: return mul_w16<true, 5>(a);
    case 6: return mul_w16<true, 6>(a);
    case 7: return mul_w16<true, 7>(a);
    case 8: return mul_w16<true, 8>(a);
    case 9: return mul_w16<true, 9>(a);
    case 10: return mul_w16<true, 10>(a);
    case 11: return mul_w16<true, 11>(a);
    case 12: return mul_w16<true, 12>(a);
    case 13: return mul_w16<true, 13>(a);
    case 14: return mul_w16<true, 14>(a);
    default: return mul_w16<true, 15>(a);
  }
}
// wk(m, k): the merge twiddle e^{-2 pi i k / N} of item m
// TWICE: emit 2 X[k] instead of X[k] -- the four halvings of every paired item are not performed (and the three unpaired
// bins doubled instead, on the one or two threads that own them).  For callers whose result is a RATIO of sums of |X|^2
// (D4C's band aperiodicity and LoveTrain statistic): every power is exactly 4 x the plain one (powers of two commute with
// every rounding on the way), so is every sum, and the quotient is the same bits.
template <int KITEMS, int NT, bool TWICE = false, class Wk, class Emit>
__device__ __forceinline__ void rfft_merge_items_w(cplx *z, int lgn, const FftPlan &plan, Wk wk, Emit emit) {
  const int lgh = lgn - 1, h = 1 << lgh, q = h >> 1;
  const int tid = wg_thread<NT>(), nt = wg_size<NT>();
#pragma unroll
  for (int m = 0; m < KITEMS; ++m) {
    const int k = tid + m * nt;
    if (k == 0) {
      const cplx za = z[fft_slot(plan, 0)];
      const double a = za.re + za.im, b = za.re - za.im;
      if (TWICE) emit(m, 0, a + a, 0.0, true, b + b, 0.0);
      else emit(m, 0, a, 0.0, true, b, 0.0);
    } else if (k < q) {
      const cplx za = z[fft_slot(plan, k)], zb = z[fft_slot(plan, h - k)];
      cplx e, o;
      if (TWICE) {
        e.re = za.re + zb.re; e.im = za.im - zb.im;
        o.re = za.im + zb.im; o.im = zb.re - za.re;
      } else {
        e.re = 0.5 * (za.re + zb.re); e.im = 0.5 * (za.im - zb.im);
        o.re = 0.5 * (za.im + zb.im); o.im = -0.5 * (za.re - zb.re);
      }
      const cplx ow = cmul(o, wk(m, k));
      emit(m, k, e.re + ow.re, e.im + ow.im, true, e.re - ow.re, ow.im - e.im);
    } else if (k == q) {
      const cplx za = z[fft_slot(plan, q)];            // k = h/2: w = -i, X = conj(z)
      if (TWICE) emit(m, q, za.re + za.re, -(za.im + za.im), false, 0.0, 0.0);
      else emit(m, q, za.re, -za.im, false, 0.0, 0.0);
    }
#ifndef WORLD_EMU
    // the callers live at the edge of their register budget: keep the scheduler from hoisting all items'
    // LDS reads to the top (10 more registers per item in flight); two items overlap, not KITEMS
    if (m & 1) __builtin_amdgcn_sched_barrier(0);
#endif
  }
  __syncthreads();
}
template <int KITEMS, int NT = 0, class Emit>
__device__ __forceinline__ void rfft_merge_items(cplx *z, int lgn, const FftPlan &plan, const TwLds &tw, Emit emit) {
  rfft_merge_items_w<KITEMS, NT>(z, lgn, plan, [&](int, int k) { return twiddle(tw, k, lgn, -1); }, emit);
}
// The same for a workgroup of exactly NT = N / 16 threads that keeps wbase = e^{-2 pi i tid / N} in registers: item m's
// twiddle is wbase * W16^m -- a constant rotation (at most 4 FP64 operations, none for m = 0 and 4) instead of a table
// lookup with its quadrant selects and the fine-level products (a third of a merge's instructions).  Other
// workgroup sizes (the one-thread host emulation) take the table.
template <int KITEMS, int NT, bool TWICE = false, class Emit>
__device__ __forceinline__ void rfft_merge_items_rot(cplx *z, int lgn, const FftPlan &plan, const TwLds &tw, cplx wbase, Emit emit) {
  rfft_merge_items_w<KITEMS, NT, TWICE>(z, lgn, plan, [&](int m, int k) {
    if (NT * 16 == (1 << lgn)) return mul_w16_fwd(wbase, m);
    return twiddle(tw, k, lgn, -1);
  }, emit);
}

// ---- real inverse transform (unscaled: N * irfft, like the reference's c2r) ----
// spec(k) returns X[k] for k in [0, N/2] (the imaginary part of DC and Nyquist
// is ignored, src/fft.cpp:28-29).  On return real output n is rfft_in(z, n).
template <class Spec>
__device__ __forceinline__ void irfft_pretwiddle(cplx *z, int lgn, const FftPlan &plan, const TwLds &tw, Spec spec) {
  const int lgh = lgn - 1, h = 1 << lgh;
  __syncthreads();
  // Pre-twiddle in conjugate pairs, walking physical slots like rfft_merge: with s = X[k] +
  // conj(X[h-k]), t = w_k (X[k] - conj(X[h-k])):  Z[k] = s + i t  and  Z[h-k] = conj(s - i t).
  const int q = h >> 1, top_bit = plan.rl(plan.ns - 1) - 1;
  struct Pair { int slot, mslot; cplx r, rm; };
  auto item = [&](int it) {
      Pair o;
      o.mslot = -1;
      if (it < q) {
        const int pos = ((it >> top_bit) << (top_bit + 1)) | (it & ((1 << top_bit) - 1));
        o.slot = swz(pos);
        const int k = fft_bin_of_slot(plan, o.slot);
        cplx x = spec(k), y = spec(h - k);
        if (k == 0) {
          o.r.re = x.re + y.re; o.r.im = x.re - y.re;    // imaginary parts of DC / Nyquist ignored
        } else {
          y.im = -y.im;                                   // conj(X[h-k])
          const cplx s = cadd(x, y), d = csub(x, y);
          const cplx t = cmul(d, twiddle(tw, k, lgn, +1));
          o.r.re = s.re - t.im; o.r.im = s.im + t.re;
          o.rm.re = s.re + t.im; o.rm.im = t.re - s.im;
          o.mslot = fft_slot(plan, h - k);
        }
      } else {
        const cplx x = spec(q);                           // k = h/2: w = +i
        o.slot = fft_slot(plan, q);
        o.r.re = 2.0 * x.re; o.r.im = -2.0 * x.im;
      }
      return o;
    };
  auto out = [&](int, Pair o) {
      z[o.slot] = o.r;
      if (o.mslot >= 0) z[o.mslot] = o.rm;
    };
  const bool odd_one = (int)threadIdx.x == (int)blockDim.x - 1;     // item q beside the q paired ones: rfft_merge
  Pair last;
  if (odd_one) last = item(q);
  block_map<2, Pair>(q, item, out);
  if (odd_one) out(q, last);
}
// The pre-twiddle for callers whose spectrum lives in GLOBAL memory: items in natural bin order (item it = tid + m T
// owns bins it and h - it, like rfft_merge_items), so that a wavefront's spec(k) calls touch consecutive addresses.
// irfft_pretwiddle above walks physical slots -- conflict-free LDS writes, but digit-reversed bins: from HBM/L2 every
// lane of a load then sits in a cache line of its own (64 lines per instruction for 1 KB of data; the filter
// bank's product step spent 23 of its 58 thousand cycles per workgroup there).  Same arithmetic, same values.
template <int KITEMS, int NT = 0, class Spec>
__device__ __forceinline__ void irfft_pretwiddle_items(cplx *z, int lgn, const FftPlan &plan, const TwLds &tw, Spec spec) {
  const int lgh = lgn - 1, h = 1 << lgh, q = h >> 1;
  const int tid = wg_thread<NT>(), nt = wg_size<NT>();
  __syncthreads();
#pragma unroll
  for (int m = 0; m < KITEMS; ++m) {
    const int k = tid + m * nt;
    if (k == 0) {
      const cplx x = spec(0), y = spec(h);
      cplx r; r.re = x.re + y.re; r.im = x.re - y.re;        // imaginary parts of DC / Nyquist ignored
      z[fft_slot(plan, 0)] = r;
    } else if (k < q) {
      cplx x = spec(k), y = spec(h - k);
      y.im = -y.im;                                           // conj(X[h-k])
      const cplx s = cadd(x, y), d = csub(x, y);
      const cplx t = cmul(d, twiddle(tw, k, lgn, +1));
      cplx r, rm;
      r.re = s.re - t.im; r.im = s.im + t.re;
      rm.re = s.re + t.im; rm.im = t.re - s.im;
      z[fft_slot(plan, k)] = r;
      z[fft_slot(plan, h - k)] = rm;
    } else if (k == q) {
      const cplx x = spec(q);                                 // k = h/2: w = +i
      cplx r; r.re = 2.0 * x.re; r.im = -2.0 * x.im;
      z[fft_slot(plan, q)] = r;
    }
  }
}
// The same for callers that hold part of the product in registers and know their twiddles: pair(m, k, &x, &y, &w) supplies
// X[k], X[h-k] and w_k = e^{+2 pi i k / N} for item m (k = tid + m NT; k = 0: x = X[0], y = X[h]; k = h/2: x only).
template <int KITEMS, int NT, class Pair>
__device__ __forceinline__ void irfft_pretwiddle_items_w(cplx *z, int lgn, const FftPlan &plan, Pair pair) {
  const int lgh = lgn - 1, h = 1 << lgh, q = h >> 1;
  const int tid = wg_thread<NT>(), nt = wg_size<NT>();
  __syncthreads();
#pragma unroll
  for (int m = 0; m < KITEMS; ++m) {
    const int k = tid + m * nt;
    cplx x, y, w;
    if (k <= q) pair(m, k, x, y, w);
    if (k == 0) {
      cplx r; r.re = x.re + y.re; r.im = x.re - y.re;        // imaginary parts of DC / Nyquist ignored
      z[fft_slot(plan, 0)] = r;
    } else if (k < q) {
      y.im = -y.im;                                           // conj(X[h-k])
      const cplx s = cadd(x, y), d = csub(x, y);
      const cplx t = cmul(d, w);
      cplx r, rm;
      r.re = s.re - t.im; r.im = s.im + t.re;
      rm.re = s.re + t.im; rm.im = t.re - s.im;
      z[fft_slot(plan, k)] = r;
      z[fft_slot(plan, h - k)] = rm;
    } else if (k == q) {
      cplx r; r.re = 2.0 * x.re; r.im = -2.0 * x.im;         // k = h/2: w = +i
      z[fft_slot(plan, q)] = r;
    }
  }
}
template <int MAXLR = 4, int LGN = 0, class Spec>
__device__ __forceinline__ void block_irfft(cplx *z, int lgn, const TwLds &tw, Spec spec) {
  if constexpr (LGN > 0) {
    constexpr FftPlan plan = make_plan_max(LGN - 1, MAXLR);
    irfft_pretwiddle(z, LGN, plan, tw, spec);
    block_cfft_dit_static<LGN - 1, MAXLR>(z, tw);
  } else {
    const FftPlan plan = make_plan_max(lgn - 1, MAXLR);
    irfft_pretwiddle(z, lgn, plan, tw, spec);
    block_cfft_dit<MAXLR>(z, plan, tw);
  }
}

}  // inline namespace (the swizzle's)
}  // namespace world_hip
