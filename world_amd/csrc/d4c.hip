// d4c.hip -- D4C band aperiodicity on gfx950.
//
// Replaces D4C() (reference src/d4c.cpp:342-403) and everything it calls:
// D4CLoveTrain (:227-285), GetStaticCentroid/GetCentroid (:90-143),
// GetSmoothedPowerSpectrum (:149-166), GetStaticGroupDelay (:172-188),
// GetCoarseAperiodicity (:194-225), GetAperiodicity (:330-338).
//
// The reference draws noise from ONE stream: first the LoveTrain window of every
// voiced frame in order, then three windows per frame that passed the LoveTrain
// threshold.  Frame independence is recovered by two prefix sums over the
// per-frame draw counts (d4c_prepare1 / d4c_prepare2) and GF(2) jump-ahead.
//
//   d4c_lovetrain : 256-thread workgroup per frame, one r2c FFT, two band sums.
//   d4c_frame     : one workgroup per selected frame, N/16 threads, one transform buffer of N doubles in LDS: each
//                   centroid position as ONE packed complex transform (two N/2-point halves, Im(P Q)/2 per bin),
//                   1 power spectrum, 2 DC corrections, 3 rectangular smoothings (block-parallel prefix sums) ->
//                   static group delay -> per 3 kHz band: Nuttall-windowed slice -> r2c FFT -> power -> radix select
//                   in registers replacing the reference's std::sort (only the sum of the N/2-boundary smallest
//                   powers is used) -> the band's two sums.  Nothing but those leaves the CU.
//   d4c_finish    : the bands' dB values and the 3 kHz-grid interpolation; every row written once to HBM, dense or
//                   straight into packed records.
// -DD4C_FP_CONTRACT (A/B, tools/ab.py): contraction for this unit only -- D4C's arithmetic is multiply-add pairs throughout
// (window rotations, interpolation, power sums) and none of it is amplified the way CheapTrick's smoothing is
#ifdef D4C_FP_CONTRACT
#pragma clang fp contract(fast)
#endif
// Host tests (round 6): this unit has NO second spelling for the CPU suite any more.  tests/emu compiles it as the GPU
// does -- WAVE = 64, the real workgroup sizes, DPP / v_readlane / ballots -- against tests/emu/simt_host.h (every thread a
// fibre, cross-lane instructions as lock-step rendezvous: -DWORLD_SIMT), so the register-first-stage transforms, the
// lane-indexed DC correction, the radix select's wave scans and the band loop's pruned first stage that ship are the ones
// `pytest -m "not gpu"` runs against the golden fixtures (rounds 3-5: 12 -> 21 -> 23 #ifdef-ed host-emulation sites here).
// The LDS slot swizzle of this unit's transforms (fft.h: swz): the map tools/lds_swizzle_search.py found for the radix-8
// plans.  Round 3 measured it on lone kernels (d4c_frame -2 %, everything else +3 %) and left the shipped map alone; under
// load -- a 128-utterance batch, profiles/r05/lds_under_load_ab.txt -- d4c_frame gains 2.5 % (10.71 -> 10.45 ms) and every
// other FFT kernel still loses (hv_band_events_fft +12 %, ct_frame +5 %, d4c_lovetrain +4 %), so it is this unit's alone:
// nothing swizzled ever leaves a kernel, the map is private to a translation unit.
#ifndef WH_SWZ
#define WH_SWZ 1
#endif
#include "stage_params.h"
#include "prepare.h"
#include "trace.h"
WH_TRACE_DEFINE(d4c)
WH_BARTRACE_DEFINE(d4c)

namespace world_hip {

constexpr int kHanning = 1, kBlackman = 2;

// ---------------------------------------------------------------------------
__global__ void d4c_prepare1(D4cParams p) {
  DYN_LDS(lds);
  d4c_offsets1_utt(p, blockIdx.x, reinterpret_cast<double *>(lds));
}
// CheapTrick's scan and D4C's first in one launch (grid (n_utt, 2)): both follow from F0 alone (prepare.h)
__global__ void spectral_prepare(CtParams cp, D4cParams dp) {
  DYN_LDS(lds);
  if (blockIdx.y == 0) ct_offsets_utt(cp, blockIdx.x, reinterpret_cast<double *>(lds));
  else d4c_offsets1_utt(dp, blockIdx.x, reinterpret_cast<double *>(lds));
}

// Sum of the `m` smallest of v[0..n) (all >= 0) and the sum of all of them: a radix
// select on the IEEE bit patterns (monotone for non-negative doubles), 8 bits per pass,
// histogram in LDS.  Keys stay in registers.  Digits start at the first bit in which any
// two keys differ (found from the block min/max), so the first histogram spans exactly
// [kmin, kmax]; the walk stops as soon as the bucket holding the threshold contains a
// single key -- typically after two passes.  Barriers: two for min/max, ONE per pass (the
// first kSelHists histograms are zeroed up front), two for the final sums.
constexpr int kSelHists = 3;
// key[q], q < mine: bit patterns of this thread's elements (n elements block-wide).
// hist: kSelHists x 256 ints of LDS.
template <int NT = 0, int kSelKeys>
__device__ __forceinline__ void block_smallest_sum(const unsigned long long (&key)[kSelKeys], int mine, int n, int m,
                                                   int *hist, double *scratch, double *partial, double *total,
                                                   bool trace_me = false) {
  (void)trace_me;
  const int tid = wg_thread<NT>(), nt = wg_size<NT>(), lane = lane_id(), wv = wave_in_block(), nw = wg_waves<NT>();
  // Range of the keys.  The keys are non-negative doubles, so their high words are non-negative ints and order
  // like the keys: the range is taken on the high words (32-bit DPP reductions, a fifth of the 64-bit ones' cost)
  // and only when every key shares its high word -- a constant spectrum -- on the full keys.
  // The lower end need not be the smallest key.  The threshold is the (n-m+1)-th largest key, and each of the nt
  // threads holds a key >= the smallest of the threads' maxima: with nt >= n-m+1 the threshold is >= that value,
  // keys below it are below the threshold whatever their rank, and only the others (a few hundred of the 2049 of
  // a band, where the first histogram's atomics used to pile 2560 keys on a few dozen bins) are histogrammed.
  const bool lower_bound = nt >= n - m + 1;
  int hmin = 0x7fffffff, hmax = 0;
#pragma unroll
  for (int q = 0; q < kSelKeys; ++q)
    if (q < mine) { const int h = (int)(key[q] >> 32); hmin = h < hmin ? h : hmin; hmax = h > hmax ? h : hmax; }
  for (int i = tid; i < kSelHists * 256; i += nt) hist[i] = 0;
  unsigned long long kmin, kmax;
  hmin = -wave_max_int(-(lower_bound ? hmax : hmin)); hmax = wave_max_int(hmax);
  // its own scratch area (doubles 48..63): whoever read it last (this function, one band ago) is behind several barriers
  int *hs = reinterpret_cast<int *>(scratch + 48);
  if (lane == 0) { hs[wv] = hmin; hs[16 + wv] = hmax; }
  __syncthreads();                                   // also: the histograms are zero before anybody counts
  for (int w = 0; w < nw; ++w) { hmin = hs[w] < hmin ? hs[w] : hmin; hmax = hs[16 + w] > hmax ? hs[16 + w] : hmax; }
  if (hmin != hmax) {
    // bits below the high word do not matter: the first differing bit is in the high word
    kmin = (unsigned long long)(unsigned)hmin << 32; kmax = (unsigned long long)(unsigned)hmax << 32;
  } else {
    kmin = ~0ull; kmax = 0ull;
#pragma unroll
    for (int q = 0; q < kSelKeys; ++q)
      if (q < mine) { kmin = key[q] < kmin ? key[q] : kmin; kmax = key[q] > kmax ? key[q] : kmax; }
    wave_minmax_u64(kmin, kmax);
    unsigned long long *ks = reinterpret_cast<unsigned long long *>(scratch);
    __syncthreads();
    if (lane == 0) { ks[wv] = kmin; ks[32 + wv] = kmax; }
    __syncthreads();
    for (int w = 0; w < nw; ++w) { kmin = ks[w] < kmin ? ks[w] : kmin; kmax = ks[32 + w] > kmax ? ks[32 + w] : kmax; }
  }
  WH_STAMP(0, 3);
  const unsigned long long diff = kmin ^ kmax;
  int hi = diff ? 64 - __clzll((long long)diff) : 0;    // bits >= hi are common to every key
  unsigned long long prefix = hi >= 64 ? 0ull : (kmin >> hi) << hi;
  int remaining = m - 1;                           // rank (0-based, ascending) of the threshold element
  int bucket = n;                                   // keys that still match the prefix
  for (int it = 0; hi > 0 && bucket > 1; ++it) {
    const int shift = hi > 8 ? hi - 8 : 0;
    const unsigned long long mask = (1ull << (hi - shift)) - 1ull;
    int *h = hist + (it % kSelHists) * 256;
    if (it >= kSelHists) {                          // rare: recycle a histogram
      __syncthreads();
      for (int i = tid; i < 256; i += nt) h[i] = 0;
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < kSelKeys; ++q) {
      // first pass: everything from the floor's bin upwards (the floor rounded down to a bin boundary, so that a
      // key is either counted here and in every later pass it qualifies for, or in none)
      const bool in = q < mine && (it == 0 ? (key[q] >> shift) >= (kmin >> shift) : (key[q] >> hi) == (prefix >> hi));
      if (in) atomicAdd(&h[(int)((key[q] >> shift) & mask)], 1);
    }
    __syncthreads();
    // every wave locates the digit redundantly: each lane takes its four bins in one 16-byte read, a wave scan
    // finds the lane holding the rank, that lane the bin
    static_assert(256 / WAVE == 4 || WAVE == 1, "four bins per lane");
    int digit = -1, below = -1, hsel = -1;
    const int4 c4 = *reinterpret_cast<const int4 *>(h + 4 * lane);
    const int local = (c4.x + c4.y) + (c4.z + c4.w);
    int tot, before = wave_excl_scan_int(local, &tot);
    if (it == 0) remaining -= n - tot;               // the keys below the floor rank below everything counted here
    if (before <= remaining && remaining < before + local) {
      const int r = remaining - before, a1 = c4.x, a2 = a1 + c4.y, a3 = a2 + c4.z;
      const int j = (r >= a1) + (r >= a2) + (r >= a3);
      digit = 4 * lane + j;
      below = before + (j == 0 ? 0 : j == 1 ? a1 : j == 2 ? a2 : a3);
      hsel = j == 0 ? c4.x : j == 1 ? c4.y : j == 2 ? c4.z : c4.w;
    }
    // exactly one lane found the bin: two max-reductions broadcast (digit, below) and hsel
    {
      const int got = wave_max_int(digit < 0 ? -1 : (digit << 14) | below);     // below <= n <= 10 keys x 1024 threads < 2^14
      hsel = wave_max_int(hsel);
      digit = got >> 14; below = got & 16383;
    }
    remaining -= below;
    bucket = hsel;
    prefix |= (unsigned long long)digit << shift;
    hi = shift;
    WH_STAMP(0, 4 + (it < 3 ? it : 3));
  }
  // Bits >= hi of the threshold are known.  If hi > 0 the bucket holds exactly one key (the
  // threshold itself), every other key differs from it above bit hi, and its owner
  // contributes it through the third sum; if hi == 0 the prefix IS the threshold (possibly
  // shared by several equal keys, `remaining` of which lie below the rank).
  const unsigned long long pfx = hi >= 64 ? 0ull : prefix >> hi;
  double s_lt = 0.0, s_all = 0.0, s_thr = 0.0;
#pragma unroll
  for (int q = 0; q < kSelKeys; ++q) {
    if (q < mine) {
      const double x = __longlong_as_double((long long)key[q]);
      const unsigned long long top = hi >= 64 ? 0ull : key[q] >> hi;
      s_all += x;
      if (top < pfx) s_lt += x;
      else if (top == pfx && hi > 0) s_thr += x;
    }
  }
  (void)n;
  WH_STAMP(0, 8);
  block_sum3<NT, false>(s_lt, s_all, s_thr, scratch);     // doubles 0..35: last read before this band's passes
  const double thr = hi > 0 ? s_thr : __longlong_as_double((long long)prefix);
  *partial = s_lt + (remaining + 1) * thr;
  *total = s_all;
}

// ---------------------------------------------------------------------------
// The same selection for the frame kernel's shape, written for INSTRUCTION COUNT: d4c_frame is bound by VALU issue
// (three resident workgroups share every SIMD), and the general routine above spent as many VALU instructions per band
// as the band's transform (~620: 64-bit key arithmetic with a `q < mine` predicate on every key, two 256-bin passes,
// three block sums, five barriers).  Here:
//   * unowned slots hold key 0 and everything is ranked FROM THE TOP (the K = n - m largest are what is excluded), so
//     padding can never be selected and no per-key ownership test exists;
//   * passes work on the keys' HIGH WORDS (32-bit compares and shifts; the low words only matter if two of the K
//     largest keys agree to 2^-20 -- then the general routine runs);
//   * the first pass maps [floor, top] of the candidates LINEARLY onto 1024 bins (bin = (hi - floor) >> s: all of the
//     resolution lies where the candidates are, whatever power-of-two boundary they straddle) with a 64-bin coarse
//     level beside it, so locating the K-th largest is one LDS word per lane, a DPP scan, a ballot, 16 fine bins
//     scanned by DPP inside a row, a ballot;
//   * once the threshold's bin holds ONE key the answer needs no threshold VALUE: the m smallest are exactly the keys
//     below that bin's lower edge -- one 32-bit compare and one masked add per key, two block sums, one barrier.
// Typical band: 3 barriers, ~200 VALU instructions.  A bin with several keys gets one 256-bin pass over its own range;
// whatever is still ambiguous after that (or a spectrum whose thread maxima all share a high word) goes to the general
// routine -- same result, every thread of the block takes the same path.
// ---- the selection's wave collectives ----
// smallest of the threads' maxima and the largest key, over the workgroup (high words); one barrier
template <int NT> __device__ __forceinline__ void sel_floor_top(int hm, int *hs, int *fl, int *tp) {
  constexpr int nw = NT / WAVE;
  const int wmin = -wave_max_int(-hm), wmax = wave_max_int(hm);
  if (lane_id() == 0) { hs[wave_in_block()] = wmin; hs[16 + wave_in_block()] = wmax; }
  __syncthreads();                                                        // also: the bins are zero before anybody counts
  int f = hs[0], t = hs[16];
#pragma unroll
  for (int w = 1; w < nw; ++w) { f = hs[w] < f ? hs[w] : f; t = hs[16 + w] > t ? hs[16 + w] : t; }
  *fl = __builtin_amdgcn_readfirstlane(f); *tp = __builtin_amdgcn_readfirstlane(t);
}
// the fine bin (of 1024, with 64 coarse counts behind them) that holds the K-th key from the top: its index, the keys
// above it and the keys in it
__device__ __forceinline__ void sel_locate_two_level(const int *hist, int K, int *bin, int *above, int *bucket) {
  const int lane = lane_id();
  // coarse level: lane l holds coarse bin l; `above` = candidates in bins above it
  const int cc = hist[1024 + lane];
  const int inc = wave_incl_scan_int(cc);
  const int above_l = __builtin_amdgcn_readlane(inc, 63) - inc;
  const unsigned long long sel = __ballot(above_l < K && K <= above_l + cc);      // exactly one lane
  const int ls = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(sel));
  const int above_c = __builtin_amdgcn_readlane(above_l, ls);
  // fine level: the 16 bins of coarse bin ls, one per lane of a row; u = candidates in this bin and above
  int u = hist[16 * ls + (lane & 15)];
  u += __builtin_amdgcn_update_dpp(0, u, 0x101, 0xf, 0xf, true);        // row_shl 1, 2, 4, 8: suffix sums inside the row
  u += __builtin_amdgcn_update_dpp(0, u, 0x102, 0xf, 0xf, true);
  u += __builtin_amdgcn_update_dpp(0, u, 0x104, 0xf, 0xf, true);
  u += __builtin_amdgcn_update_dpp(0, u, 0x108, 0xf, 0xf, true);
  u += above_c;
  const unsigned ge = (unsigned)(__ballot(u >= K) & 0xFFFFull);        // lanes 0 .. is of the first row (u falls with the lane)
  const int is = __builtin_amdgcn_readfirstlane(31 - __builtin_clz(ge));
  const int u_is = __builtin_amdgcn_readlane(u, is);
  const int above0 = is == 15 ? above_c : __builtin_amdgcn_readlane(u, (is + 1) & 15);
  *bin = 16 * ls + is; *above = above0; *bucket = u_is - above0;
}
// the same over the second pass's 256 bins
__device__ __forceinline__ void sel_locate_256(const int *h1, int K1, int *bin, int *bucket) {
  const int lane = lane_id();
  const int4 c4 = *reinterpret_cast<const int4 *>(h1 + 4 * lane);
  const int local = (c4.x + c4.y) + (c4.z + c4.w);
  const int inc1 = wave_incl_scan_int(local);
  const int ab1 = __builtin_amdgcn_readlane(inc1, 63) - inc1;
  const unsigned long long sel1 = __ballot(ab1 < K1 && K1 <= ab1 + local);
  const int l1 = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(sel1));
  int a = __builtin_amdgcn_readlane(ab1, l1);
  const int c3 = __builtin_amdgcn_readlane(c4.w, l1), c2 = __builtin_amdgcn_readlane(c4.z, l1),
            c1 = __builtin_amdgcn_readlane(c4.y, l1), c0 = __builtin_amdgcn_readlane(c4.x, l1);
  int j;                                                            // scalar walk from the top bin of the lane's four
  if (a + c3 >= K1) { j = 3; *bucket = c3; }
  else if ((a += c3) + c2 >= K1) { j = 2; *bucket = c2; }
  else if ((a += c2) + c1 >= K1) { j = 1; *bucket = c1; }
  else { j = 0; *bucket = c0; }
  *bin = 4 * l1 + j;
}
__device__ __forceinline__ void sel_zero_bins(int *hist, int tid, int nt) {
  for (int i = tid; i < 336; i += nt) reinterpret_cast<int4 *>(hist)[i] = make_int4(0, 0, 0, 0);
}

// key[0 .. kKeys-2): owned by every thread; key[kKeys-2]: thread 0 only (the merge's unpaired bin), 0 elsewhere;
// key[kKeys-1]: 0.  hist: 1344 ints of LDS.  K: how many of the largest keys are EXCLUDED from *partial.
// Returns true when *partial / *total are the BLOCK's sums (the general routine ran), false when they are the calling
// wavefront's share of them: the caller adds the wavefronts' shares up whenever it next crosses a barrier anyway (the
// block sum here -- LDS, barrier, LDS -- was a quarter of the routine's time in a CU whose LDS pipe is busy with other
// workgroups' transforms: every dependent trip through it costs ~750 cycles there).
template <int NT, int kKeys>
__device__ __forceinline__ bool block_excluding_largest(const unsigned long long (&key)[kKeys], int K, int *hist, double *scratch,
                                                        double *partial, double *total, bool trace_me = false, int threads = NT) {
  (void)trace_me;
  constexpr int kFull = kKeys - 2;
  static_assert(NT % WAVE == 0 && NT / WAVE <= 16, "whole wavefronts");
  const int hx = (int)(key[kFull] >> 32);                                 // 0 except on thread 0
  const int tid = wg_thread<NT>();
  int hk[kFull];
#pragma unroll
  for (int q = 0; q < kFull; ++q) hk[q] = (int)(key[q] >> 32);            // non-negative doubles: the high words order like the keys
  // [0, 1024) fine bins | [1024, 1088) coarse bins (16 fine each) | [1088, 1344) the second pass's bins
  sel_zero_bins(hist, tid, NT);
  // Every thread's largest key is a candidate, and there are `threads` >= K of them: the K-th largest key is >= the smallest
  // of the threads' maxima (`floor`), so keys below it are below the threshold whatever their rank.
  int fl, tp;
  {
    int hm = hx;
#pragma unroll
    for (int q = 0; q < kFull; ++q) hm = hk[q] > hm ? hk[q] : hm;
    sel_floor_top<NT>(hm, reinterpret_cast<int *>(scratch + 48), &fl, &tp);   // doubles 48..63: nobody else's scratch
  }
  WH_STAMP(0, 3);
  bool fast = threads >= K && tp > fl;
  bool t_low_two = false; (void)t_low_two;                                // (statistics of the WH_TRACE build)
  int t_low = 0;                                                          // keys with a high word below this are the m smallest
  if (fast) {
    const unsigned range = (unsigned)(tp - fl);
    const int bits = 32 - __builtin_clz(range), s0 = bits > 10 ? bits - 10 : 0;      // (tp - fl) >> s0 < 1024
    auto count0 = [&](int h) __attribute__((always_inline)) {
      const unsigned b = (unsigned)(h - fl) >> s0;                        // below the floor: wraps to >= 2^31 >> 21 = 1024
      if (b < 1024u) { atomicAdd(&hist[b], 1); atomicAdd(&hist[1024 + (b >> 4)], 1); }
    };
#pragma unroll
    for (int q = 0; q < kFull; ++q) count0(hk[q]);
    if (tid == 0 && kFull < kKeys) count0(hx);
    __syncthreads();
    int bin0, above0, bucket;
    sel_locate_two_level(hist, K, &bin0, &above0, &bucket);
    t_low = fl + (bin0 << s0);
    WH_STAMP(0, 4);
    if (bucket != 1) {
      if (s0 == 0) {
        fast = false;                                                     // several of the largest keys share a high word
      } else {
        // second pass over the keys of that bin: its range of 2^s0 high words on 256 bins
        const int K1 = K - above0, s1 = s0 > 8 ? s0 - 8 : 0;
        int *h1 = hist + 1088;
        auto count1 = [&](int h) __attribute__((always_inline)) {
          const unsigned d = (unsigned)(h - t_low);
          if (d < (1u << s0)) atomicAdd(&h1[d >> s1], 1);
        };
#pragma unroll
        for (int q = 0; q < kFull; ++q) count1(hk[q]);
        if (tid == 0 && kFull < kKeys) count1(hx);
        __syncthreads();
        int bin1;
        sel_locate_256(h1, K1, &bin1, &bucket);
        t_low += bin1 << s1;
        t_low_two = true;
        if (bucket != 1) fast = false;
        WH_STAMP(0, 5);
      }
    }
  }
#ifdef WH_TRACE
  // development aid: how the bands of a launch were decided (tools/trace.py): [100] one pass, [101] two, [102] general routine
  if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&wh_trace[!fast ? 102 : (t_low_two ? 101 : 100)]), 1ull);
#endif
  if (!fast) {
    __syncthreads();                                                      // nobody is still reading the bins
    // the general routine: every slot counts as a key (the zeros are the smallest and add nothing)
    block_smallest_sum<NT>(key, kKeys, NT * kKeys, NT * kKeys - K, hist, scratch, partial, total, trace_me);
    return true;
  }
  double s_lt = 0.0, s_all = 0.0;
#pragma unroll
  for (int q = 0; q < kFull; ++q) {
    const double x = __longlong_as_double((long long)key[q]);
    s_all += x;
    s_lt += hk[q] < t_low ? x : 0.0;
  }
  if (tid == 0 && kFull < kKeys) {
    const double x = __longlong_as_double((long long)key[kFull]);
    s_all += x;
    s_lt += hx < t_low ? x : 0.0;
  }
  WH_STAMP(0, 8);
  *partial = wave_sum(s_lt);
  *total = wave_sum(s_all);
  return false;
}

// ---------------------------------------------------------------------------
// D4CGeneralBody for one selected frame in ONE workgroup (d4c.cpp:90-225, 291-316):
//   GetStaticCentroid -> GetSmoothedPowerSpectrum -> GetStaticGroupDelay -> GetCoarseAperiodicity.
//
// Shape: T = N/16 threads (N = fft_size_d4c: 256 threads at 48 kHz), one radix-8 butterfly per thread and
// stage, and ONE transform buffer of N doubles in LDS.  Everything that is indexed by frequency bin lives in
// REGISTERS, LDS is only the exchange medium -- the smoothing's prefix-sum array, the few bins DCCorrection
// mirrors, the 513-sample band slices.
//
// The centroid of one window position is Re(X2 conj X1) with X1 = FFT(u), X2 = FFT((n+1) u), u the balanced
// windowed waveform.  Both spectra come out of ONE N-point complex transform of z[n] = u[n] (1 + i (n+1)):
// with P = Z[k], Q = Z[N-k],  X1 = (P + conj Q)/2,  X2 = (P - conj Q)/(2i)  and the product collapses to
//     Re(X2 conj X1) = (Im P Re Q + Im Q Re P) / 2 = Im(P Q) / 2.
// The N-point transform runs as the two N/2-point halves of a radix-2 DIF split -- even bins from
// e[n] = z[n] + z[n+N/2], odd bins from o[n] = (z[n] - z[n+N/2]) W_N^n -- through the one N/2-point complex
// buffer, and bins k and N-k have the same parity, so each half yields its centroid values on its own:
// nothing waits in registers across a transform (the previous version held X1, 40 VGPRs, through the second
// transform and spilled), there is no twiddled merge step, and the window's samples are evaluated ONCE: a
// thread keeps its <= 8 samples below N/2 in registers between the balance pass and the two input passes
// (windows longer than N/2 -- F0 below 94 Hz at 48 kHz -- re-evaluate their upper samples, the rare case).
// Thread t ends up owning the bin pairs (2 (t + m T), 2 (t + m T) + 1): "pair ownership".  The power spectrum
// still comes from a real transform whose merge step hands out conjugate pairs (t + m T, N/2 - that): "natural
// ownership"; LinearSmoothing goes through LDS by bin index anyway, so it converts between the two for free.
// The static group delay never goes to HBM: the band transforms of GetCoarseAperiodicity read their
// Nuttall-windowed slices from LDS (a first stage that knows all but one input of every butterfly to be zero),
// and the power values feed the radix select from registers.
template <int NMAX, int T> struct D4cShape {
  static constexpr int kItems = (NMAX / 4 + 1 + T - 1) / T;
  static constexpr int kBins = 2 * kItems;
  static constexpr int kLo = NMAX / 2 / T;             // window samples below N/2 per thread (8 on the GPU)
};

struct D4cWin {            // GetWindowedWaveform's parameters (d4c.cpp:52-84)
  const double *x;
  const uint32_t *noise;   // the window's draws in sample order (d4c.cpp:67-69)
  int x_len, hw, wlen, origin, kind;
  double scale;
};
__device__ __forceinline__ D4cWin d4c_win(const double *x, int x_len, int fs, double f0, double pos, int kind, double ratio,
                                          const uint32_t *noise) {
  D4cWin w;
  w.x = x; w.x_len = x_len; w.noise = noise; w.kind = kind;
  w.hw = mround(ratio * fs / f0 / 2.0);
  w.wlen = 2 * w.hw + 1;
  w.origin = mround(pos * fs + 0.001);
  w.scale = 2.0 / ratio / fs * f0;
  return w;
}
// The window of a thread's samples i0, i0 + step, i0 + 2 step, ... by rotation: cos(pi scale (i - hw)) advances by
// a fixed angle per sample, so one sincospi pair per thread and pass replaces one cospi per sample (the window
// evaluations were a fifth of the frame kernel's instructions).  At most N / T = 16 rotations on the GPU.
struct D4cWinRot { double c, s, dc, ds; };
__device__ __forceinline__ D4cWinRot d4c_win_rot(const D4cWin &w, int i0, int step) {
  D4cWinRot r;
  sincospi(w.scale * (i0 - w.hw), &r.s, &r.c);
  sincospi(w.scale * step, &r.ds, &r.dc);
  return r;
}
__device__ __forceinline__ double d4c_win_next(const D4cWin &w, D4cWinRot &r) {   // value at the current sample, then advance
  const double c1 = r.c;
  const double v = w.kind == kHanning ? 0.5 * c1 + 0.5 : 0.42 + 0.5 * c1 + 0.08 * (2.0 * c1 * c1 - 1.0);
  r.c = c1 * r.dc - r.s * r.ds;
  r.s = r.s * r.dc + c1 * r.ds;
  return v;
}
struct D4cSample { double v, w; };
// sample i of the windowed, dithered waveform (d4c.cpp:61-69); rot must stand at sample i
__device__ __forceinline__ D4cSample d4c_sample(const D4cWin &w, int i, D4cWinRot &rot) {
  D4cSample s;
  s.w = d4c_win_next(w, rot);
  s.v = w.x[imin(w.x_len - 1, imax(0, w.origin + i - w.hw))] * s.w + randn_value(w.noise[i]) * kSafeGuardD4C;
  return s;
}
// windowed, dithered samples into the real-transform input; returns the DC-balance coefficient
// (sum of the waveform / sum of the window, d4c.cpp:71-80) which the callers apply
// rot0: d4c_win_rot(w, thread, workgroup size)
template <int NT = 0>
__device__ __forceinline__ double d4c_window_to_lds(const D4cWin &w, const D4cWinRot &rot0, cplx *Z, double *scratch) {
  double s1 = 0.0, s2 = 0.0;
  D4cWinRot rot = rot0;
  block_map<4, D4cSample, NT>(w.wlen, [&](int i) { return d4c_sample(w, i, rot); },
                              [&](int i, D4cSample s) { rfft_in(Z, i) = s.v; s1 += s.v; s2 += s.w; });
  block_sum2<NT>(s1, s2, scratch);
  return s1 / s2;
}

// ---------------------------------------------------------------------------
// D4CLoveTrain (d4c.cpp:227-285): is the frame voiced enough to analyse?  LGN: log2 of the transform when the
// instantiation fixes it (compile-time plan), 0 = p.lg_love.
template <int LGN>
__device__ __forceinline__ void d4c_love_frame(const D4cParams &p, char *lds) {
  const int u = blockIdx.y, f = blockIdx.x;
  if (f >= p.b.n_frames[u]) return;
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const double f0 = d4c_sane_f0(p.f0[fi], p.b.fs);
  if (f0 == 0.0) { if (threadIdx.x == 0) { p.ap0[fi] = 0.0; p.draws2[fi] = 0u; } return; }   // d4c.cpp:274-277
  const int lgn = LGN > 0 ? LGN : p.lg_love, M = 1 << lgn, fs = p.b.fs;
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds);
  double *scratch = Zr + M;
  const TwLds tw = stage_twiddles(scratch + 64, lgn - 1, p.tab.tw);    // inner complex transform's table
  const double cf0 = f0 > 40.0 ? f0 : 40.0;                            // d4c.cpp:263,279
  const D4cWin w = d4c_win(p.b.x + (size_t)u * p.b.x_stride, p.b.x_len[u], fs, cf0, p.tpos[fi], kBlackman, 3.0,
                           p.noise + p.offsets1[fi]);
  const D4cWinRot rot0 = d4c_win_rot(w, threadIdx.x, blockDim.x);
  const int b0 = static_cast<int>(ceil(100.0 * M / fs));
  const int b1 = static_cast<int>(ceil(4000.0 * M / fs));
  const int b2 = static_cast<int>(ceil(7900.0 * M / fs));
  double lo = 0.0, hi = 0.0;     // cumulative power (b0, b1] and (b0, b2]   (d4c.cpp:241-249)
  auto band_power = [&](int k, double re, double im) __attribute__((always_inline)) {
    if (k > b0 && k <= b2) {
      const double pw = re * re + im * im;
      hi += pw;
      if (k <= b1) lo += pw;
    }
  };
  if constexpr (LGN > 0 && (1 << LGN) == 16 * 256) {
    // the 48 kHz shape (4096 points on 256 threads = N / 16): the window's samples stay in registers between the two
    // passes (no LDS round trip, the window evaluated once), the transform's inner stages are wave-local, the merge
    // twiddles come by rotation, and only the items that can hold a bin <= b2 are merged at all
    constexpr int T = 256, kPer = (1 << LGN) / T, kItems = ((1 << LGN) / 4 + 1 + T - 1) / T;
    const int tid = wg_thread<T>();
    double v[kPer], ww[kPer];
    double s1 = 0.0, s2 = 0.0;
    {
      D4cWinRot rot = rot0;
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        const int i = tid + j * T;
        v[j] = 0.0; ww[j] = 0.0;
        if (i < w.wlen) { const D4cSample sm = d4c_sample(w, i, rot); v[j] = sm.v; ww[j] = sm.w; s1 += sm.v; s2 += sm.w; }
      }
    }
    block_sum2<T>(s1, s2, scratch);
    const double coef = s1 / s2;
#pragma unroll
    for (int j = 0; j < kPer; ++j) rfft_in(Z, tid + j * T) = v[j] - ww[j] * coef;      // 0 beyond the window
    constexpr FftPlan plan = make_plan_max(LGN - 1, 3);
    block_cfft_dif_static<LGN - 1, 3, T>(Z, tw);
    const cplx wb = twiddle(tw, tid, LGN, -1);
    const int h = 1 << (LGN - 1);
    // (2 X[k]: lo / hi is a ratio of power sums -- fft.h: rfft_merge_items_w<TWICE>)
    rfft_merge_items_w<kItems, T, true>(Z, LGN, plan, [&](int m, int) { return mul_w16_fwd(wb, m); },
                                  [&](int, int k, double ar, double ai, bool paired, double br, double bi) {
      band_power(k, ar, ai);
      if (paired) band_power(h - k, br, bi);
    });
  } else {
    const double coef = d4c_window_to_lds(w, rot0, Z, scratch);
    {
      D4cWinRot rot = rot0;
      for (int i = threadIdx.x; i < M; i += blockDim.x)
        rfft_in(Z, i) = i < w.wlen ? rfft_in(Z, i) - d4c_win_next(w, rot) * coef : 0.0;
    }
    block_rfft<3, LGN>(Z, lgn, tw, band_power);
  }
  block_sum2<LGN == 11 ? 128 : LGN == 12 ? 256 : 0>(lo, hi, scratch);   // (launch_d4c's workgroup sizes: the wavefronts' shares are read together)
  if (threadIdx.x == 0) {
    const double ap0 = lo / hi;
    p.ap0[fi] = ap0;
    // ... and the randn() draws the frame's three body windows will take (d4c.cpp:386, :97-101, :155) -- none if this
    // statistic keeps the frame out of the second pass.  Their positions in the stream are a prefix sum over the
    // utterance's frames, which every d4c_frame workgroup forms for itself (round 4: a 1-workgroup scan kernel per job).
    const double bf0 = kFloorF0D4C > f0 ? kFloorF0D4C : f0;
    p.draws2[fi] = ap0 <= p.threshold ? 0u : 3u * (unsigned)(2 * mround(4.0 * fs / bf0 / 2.0) + 1);
  }
}
template <int LGN>
__global__ void __launch_bounds__(256) d4c_lovetrain(D4cParams p) {
  DYN_LDS(lds);
  d4c_love_frame<LGN>(p, lds);
}

// Resident waves per SIMD the register allocation aims at: 3 (three 256-thread workgroups per CU, 168 VGPRs;
// LDS allows no more).
#ifndef D4C_MIN_WAVES
#define D4C_MIN_WAVES 3
#endif
// The kernel lives at its register cap: a scheduling fence after every item of a per-bin loop keeps the compiler from
// putting all items' LDS reads in flight at once (it did, and spilled 170 dwords in the smoothing loops alone).
#if defined(D4C_FENCES)
#define D4C_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define D4C_SCHED_FENCE() do { } while (0)
#endif
// value of the key slots a thread does not own: the selection ranks from the top and needs them to be the smallest
#define D4C_KEY_PAD 0ull
#ifndef D4C_TW_LEVEL
#define D4C_TW_LEVEL 2      // the twiddle table in LDS is this many levels coarser than the N-point merge asks for
#endif
// A per-bin loop ends with the one conditional item of thread 0 (bin N/2): the compiler's sinking pass then moves the
// arithmetic of the unconditional items below that branch, next to their first use, while the LDS loads that feed it
// stay put -- dozens of loaded values wait in registers and spill.  A value passed through keep() is "used" where
// it is computed, so its arithmetic stays in front of the branch.
// (keep() itself lives in common.h)
#if !defined(WORLD_SIMT)
#define D4C_FRESH_TID() do { asm volatile("" : "+v"(tid)); __builtin_assume(tid >= 0 && tid < T); } while (0)
#else
#define D4C_FRESH_TID() do { } while (0)
#endif
// LDS of d4c_frame (doubles): Z (N) | scratch (64) | quarter-wave table of the N/2-point complex transform | the group delay
// (N/2 + 2; N <= 8192 only) | the wavefronts' shares of the bands' sums (8 bands x 2 x N / 1024 wavefronts) | direct twiddles
// of the two inner radix-8 stages (2^(lg-7) + 2^(lg-10) complex)
__host__ __device__ constexpr size_t d4c_frame_direct_tw_offset(int lg) {
  const size_t N = (size_t)1 << lg;
  const size_t park = lg > 13 ? 0 : N / 2 + 2;          // (the 16384-point shape parks in global memory: d4c_frame)
  return sizeof(double) * (N + 64 + ((size_t)1 << (lg - D4C_TW_LEVEL - 2)) + 2 + park + 16 * (N / 1024 > 1 ? N / 1024 : 1));
}
template <int NMAX, int T>
__global__ void __launch_bounds__(T, D4C_MIN_WAVES) d4c_frame(D4cParams p) {
  constexpr int kItems = D4cShape<NMAX, T>::kItems, kBins = D4cShape<NMAX, T>::kBins, kLo = D4cShape<NMAX, T>::kLo;
  constexpr bool kRot = T * 16 == NMAX;                                // twiddles by constant rotation (fft.h)
  DYN_LDS(lds);
  const int u = blockIdx.y, f = p.frame_lo + blockIdx.x;
  const size_t fi = (size_t)u * p.b.f_stride + f;                      // (f < f_stride: launch_d4c's grid ends at the range)
  int tid = wg_thread<T>();
  constexpr int nt = T;                                                // launch_d4c launches exactly T threads
  // Everything the workgroup needs from global memory before its first window is REQUESTED here, ahead of the two early
  // exits and of the table staging: the frame's F0 and LoveTrain statistic, its position, the utterance's length and
  // first-pass draw count, and the thread's share of the earlier frames' draw counts.  Asked for where each is first used
  // they were six dependent trips to L2 (n_frames -> F0 -> statistic -> tables -> counts, two at a time -> position):
  // ~9 of the frame's 157 thousand cycles in a loaded CU (tools/trace_batch.py).
  const int nf_u = p.b.n_frames[u];
  const double f0_in = p.f0[fi], ap0_in = p.ap0[fi], pos = p.tpos[fi];
  const int x_len = p.b.x_len[u];
  const unsigned draws1_u = p.draws1[u];
  constexpr int kCntAhead = 8;                                         // counts per thread in flight: 8 T frames = 10 s at 48 kHz
  unsigned cnt_v[kCntAhead];
  {
    const unsigned *cnt = p.draws2 + (size_t)u * p.b.f_stride;
#pragma unroll
    for (int k = 0; k < kCntAhead; ++k) cnt_v[k] = cnt[imin(tid + k * nt, f)];      // (index f itself is this frame's: in range)
  }
  if (f >= nf_u || f >= p.frame_hi) return;
  // where the wavefronts' shares of a window's power cross (scratch doubles): clear of the balance pass's block sum,
  // which a faster wavefront may still be reading -- that one takes doubles [0, waves) and [32, 32 + waves), so the
  // sixteen wavefronts of the 16384-point shape push this area up into the select's (48..63: idle until the band loop)
  constexpr int kPwAt = T > 8 * WAVE ? 48 : 40;
  const double f0 = d4c_sane_f0(f0_in, p.b.fs);
  if (f0 == 0 || ap0_in <= p.threshold) return;                        // d4c.cpp:386
  const bool trace_me = f == WH_TRACE_FRAME && u == WH_TRACE_UTT; (void)trace_me;
  wh_bartrace_arm(trace_me);                                           // (-DWH_BARTRACE only: tools/barrier_skew.py)
  WH_STAMP(32, 0);
  // On the GPU the transform length is the shape's (launch_d4c picks the instantiation), so the plan, every
  // stage's radix and stride and the digit reversal of the merge steps are compile-time constants.
  constexpr int lgn = const_log2(NMAX);
  const int N = 1 << lgn, H = N / 2, q = H / 2, fs = p.b.fs;
  // LDS: Z (N doubles: the transform, or whatever is exchanged between transforms) | scratch (64) | twiddles | group delay
  cplx *Z = reinterpret_cast<cplx *>(lds);
  double *Zr = reinterpret_cast<double *>(lds);
  double *scratch = Zr + N;
  // ... and behind everything else (d4c_frame_lds_bytes) the twiddles of the transforms' two inner radix-8 stages, one
  // 16-byte entry per distinct value: every one of the frame's 13 transforms reads them (fft.h: stage_direct_twiddles;
  // the barrier in front of the first transform -- the stream-position sum's, below -- publishes them)
  const TwLds tw = stage_direct_twiddles<T>(stage_twiddles<T>(scratch + 64, lgn - D4C_TW_LEVEL, p.tab.tw),
                                            reinterpret_cast<cplx *>(lds + d4c_frame_direct_tw_offset(lgn)),
                                            lgn - 1 - 3, 1 << (lgn - 1 - 6), lgn - 1 - 6, 1 << (lgn - 1 - 9));
  constexpr FftPlan plan = make_plan_max(lgn - 1, 3);
  auto cfft = [&]() __attribute__((always_inline)) { block_cfft_dif_static<lgn - 1, 3, T>(Z, tw); };
  const cplx wb = twiddle(tw, tid, lgn, -1);                           // e^{-2 pi i tid / N}: every other twiddle of the
                                                                       // thread is this one times a constant
  // body(slot, k) for every bin this thread owns; `slot` is a compile-time constant after unrolling.
  // natural ownership (rfft_merge_items): item m = bins tid + m T and H - that
  auto for_nat = [&](auto body) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      const int it = tid + m * nt;
      if (it <= q) { body(2 * m, it); if (it < q) body(2 * m + 1, H - it); }
      D4C_SCHED_FENCE();
    }
  };
  // pair ownership (the centroid's even / odd halves): item m = bins 2 (tid + m T) and the next one
  auto for_pair = [&](auto body) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      const int it = tid + m * nt;
      if (it <= q) { body(2 * m, 2 * it); if (it < q) body(2 * m + 1, 2 * it + 1); }
      D4C_SCHED_FENCE();
    }
  };

  const double *x = p.b.x + (size_t)u * p.b.x_stride;
  const double cf0 = kFloorF0D4C > f0 ? kFloorF0D4C : f0;
  // Where the frame's draws sit in the reference's one randn() stream: behind the first pass's (draws1) and behind those
  // of every earlier frame of the utterance that LoveTrain let through -- <= 8 KB of counts summed by the workgroup itself.
  unsigned stream_at;
  {
    const unsigned *cnt = p.draws2 + (size_t)u * p.b.f_stride;
    unsigned su = 0;                                         // (sums wrap like the unsigned stream position they feed)
#pragma unroll
    for (int k = 0; k < kCntAhead; ++k) su += tid + k * nt < f ? cnt_v[k] : 0u;     // requested at the top of the kernel
    for (int g = tid + kCntAhead * nt; g < f; g += nt) su += cnt[g];                // (utterances beyond 8 T frames)
    int s_ = wave_sum_int((int)su);
    int *part = reinterpret_cast<int *>(Zr);                 // (nothing lives in LDS yet)
    if (lane_id() == 0) part[wave_in_block()] = s_;
    __syncthreads();
    int tot_ = 0;
    for (int wv = 0; wv < wg_waves<T>(); ++wv) tot_ += part[wv];
    tot_ = __builtin_amdgcn_readfirstlane(tot_);             // the same in every lane: a scalar register, like the offset a scan kernel used to leave
    stream_at = draws1_u + (unsigned)tot_;
    // (no closing barrier: the next write to this area lies behind the first window's block sum, which nobody passes
    // before everybody has arrived there -- past these reads)
  }
  const uint32_t *noise = p.noise + stream_at;
  const int wdraws = 2 * mround(4.0 * fs / cf0 / 2.0) + 1;
  // `park` (N/2 + 2 doubles: the centroid sum, then the static group delay) lives in LDS beside the transform -- except
  // for the 16384-point shape (96 kHz < fs <= 192 kHz), whose transform buffer alone takes 128 of the CU's 160 KB: there
  // it is the workgroup's slot of a global staging area (launch_d4c hands out p.park_slots of them per launch; a slot is
  // written and read by its own workgroup only, between its own barriers)
  const bool park_global = lgn > 13;
  double *const after_tw = scratch + 64 + twiddle_lds_doubles(lgn - D4C_TW_LEVEL);
  double *park = park_global ? p.park_ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(H + 2) : after_tw;
  const double inv_n = 1.0 / N;

  // DCCorrection (common.cpp:56-75) on register bins: the few low bins it mirrors go through LDS
  auto dc_correct = [&](double (&S)[kBins], auto for_own) __attribute__((always_inline)) {
    const int upper = 2 + static_cast<int>(cf0 * N / fs), nrep = upper - 1;
    const double inv_dx = -static_cast<double>(N) / fs;
    __syncthreads();
    for_own([&](int slot, int k) { if (k <= upper) Zr[k] = S[slot]; });
    __syncthreads();
    // interp1Q at k fs / N on the axis cf0 - j fs / N: position cf0 N / fs - k -- the SAME fraction for every bin, and
    // the knot index a constant minus k (the per-bin subtract / scale / convert / clamp of the general routine was a
    // quarter of this pass's instructions; the weights agree to the rounding of the position, ~1e-13)
    const double pos0 = (0.0 - cf0) * inv_dx;
    const int b0 = static_cast<int>(pos0);
    const double fr0 = pos0 - b0;
    for_own([&](int slot, int k) {
      if (k < nrep) {
        const int b = b0 - k;                                // 1 <= b <= upper - 1 for k < nrep = upper - 1 ... b0 = upper - 2
        const double y0 = Zr[b], y1 = Zr[b + 1 <= upper ? b + 1 : upper];
        S[slot] = keep(S[slot] + (y0 + (y1 - y0) * fr0));
      }
    });
    __syncthreads();
  };
  // LinearSmoothing (common.cpp:27-111): register bins -> mirrored segment in LDS -> block prefix sum -> register bins;
  // the input and the output may be owned differently
  // pre: a barrier first (whoever read Zr last is not behind one yet)
  auto smooth = [&](const double (&in)[kBins], auto for_in, double width, double (&out)[kBins], auto for_out, const bool pre = true) __attribute__((always_inline)) {
    const int bnd = static_cast<int>(width * N / fs) + 1;
    const int seg_len = H + 2 * bnd + 1;
    if (pre) __syncthreads();
    for_in([&](int slot, int k) {
      const double v = in[slot] * fs * inv_n;            // == .. * fs / N: N is a power of two
      Zr[k + bnd] = v;
      if (k >= 1 && k <= bnd) Zr[bnd - k] = v;
      if (k < H && k >= H - bnd) Zr[2 * H + bnd - k] = v;
    });
    block_scan_incl_double<T>(Zr, seg_len, scratch);
    const double origin_axis = -(bnd - 0.5) * fs / N;
    const double inv_step = static_cast<double>(N) / fs, inv_width = 1.0 / width;
    // interp1Q of the prefix sums at k fs / N -+ width / 2: on the segment's axis that is position k + c_lo and k + c_hi
    // with c_lo = bnd - 0.5 - r / 2, c_hi = c_lo + r (r = width N / fs) -- the fractions are the same for EVERY bin and
    // the knot indices are k plus a constant.  The general routine recomputed position, index, fraction and an end clamp
    // per bin and edge (24 VALU instructions per bin; 6 here); the weights agree to the rounding of the position
    // (~1e-13), and k + floor(c_hi) + 1 <= H + 1.5 r + 2 < seg_len - 1: no clamp can ever apply.
    const double c_lo = (0.0 - width / 2.0 - origin_axis) * inv_step, c_hi = (width / 2.0 - origin_axis) * inv_step;
    const int i_lo = static_cast<int>(c_lo), i_hi = static_cast<int>(c_hi);
    const double f_lo = c_lo - i_lo, f_hi = c_hi - i_hi;
    const double *z_lo = Zr + i_lo, *z_hi = Zr + i_hi;
    for_out([&](int slot, int k) {
      const double l0 = z_lo[k], l1 = z_lo[k + 1], h0 = z_hi[k], h1 = z_hi[k + 1];
      const double lo = l0 + (l1 - l0) * f_lo, hi = h0 + (h1 - h0) * f_hi;
      out[slot] = keep((hi - lo) * inv_width);
    });
    (void)seg_len;
  };

  // The frame's three windows -- two Blackman for the centroid, one Hanning for the power spectrum -- share their
  // length and their angle per sample (ratio 4, the same F0: d4c.cpp:97,101,155), so the cosine every one of them
  // is built from starts and advances identically: one sincospi pair per frame.
  const D4cWinRot rot0 = d4c_win_rot(d4c_win(x, x_len, fs, cf0, pos, kHanning, 4.0, noise), tid, nt);

  // One window (d4c.cpp:52-84): the thread's samples below N/2 -- i = tid + j T, j < kLo -- balanced and in registers
  // (0 beyond the window); returns the balance coefficient.  Samples at and above N/2 exist only for windows longer
  // than N/2; they are walked again by whoever needs them (for_hi), so the common case pays for nothing.
  auto for_hi = [&](const D4cWin &w, double coef, auto body) __attribute__((always_inline)) {
    if (w.wlen > H) {
      D4cWinRot rot = rot0;                                            // H / T steps on: sample H + tid
#pragma unroll 1
      for (int i = tid; i < H; i += nt) d4c_win_next(w, rot);
      for (int i = H + tid; i < w.wlen; i += nt) {
        const D4cSample s = d4c_sample(w, i, rot);
        body(i, s.v - s.w * coef);
      }
    }
  };
  // the same walk with the slot known at compile time: sample H + tid + k T belongs to the thread's element k
  auto hi_samples = [&](const D4cWin &w, double coef, auto body) __attribute__((always_inline)) {
    if (w.wlen > H) {
      D4cWinRot rot = rot0;                                            // H / T steps on: sample H + tid
#pragma unroll 1
      for (int i = tid; i < H; i += nt) d4c_win_next(w, rot);
#pragma unroll
      for (int k = 0; k < kLo; ++k) {
        const int i = H + tid + k * nt;
        if (i < w.wlen) {
          const D4cSample sm = d4c_sample(w, i, rot);
          body(k, i, sm.v - sm.w * coef);
        }
      }
    }
  };
  auto balanced = [&](const D4cWin &w, double (&ulo)[kLo]) __attribute__((always_inline)) {
    double wlo[kLo];
    double s1 = 0.0, s2 = 0.0;
    D4cWinRot rot = rot0;
    // All of the thread's waveform samples and draws are requested before the first is used, at clamped addresses: a
    // load inside `if (i < wlen)` is waited for where the branch rejoins, so the kLo items queued up one trip to
    // memory each (24 dependent trips per frame over the three windows).  Items beyond the window contribute 0.
    double xv[kLo];
    uint32_t nz[kLo];
#pragma unroll
    for (int j = 0; j < kLo; ++j) {
      const int i = imin(tid + j * nt, w.wlen - 1);
      xv[j] = w.x[imin(w.x_len - 1, imax(0, w.origin + i - w.hw))];
      nz[j] = w.noise[i];
    }
#pragma unroll
    for (int j = 0; j < kLo; ++j) { xv[j] = keep(xv[j]); nz[j] = keep_word(nz[j]); }   // fetched here, not down in the branches
#pragma unroll
    for (int j = 0; j < kLo; ++j) {
      const int i = tid + j * nt;
      ulo[j] = 0.0; wlo[j] = 0.0;
      if (i < w.wlen && i < H) {
        const double ww = d4c_win_next(w, rot);                        // d4c_sample() on the values already fetched
        const double v = xv[j] * ww + randn_value(nz[j]) * kSafeGuardD4C;
        ulo[j] = v; wlo[j] = ww; s1 += v; s2 += ww;
      }
    }
    // a window longer than N/2 has used all kLo steps: rot stands at sample H + tid
    for (int i = H + tid; i < w.wlen; i += nt) { const D4cSample s = d4c_sample(w, i, rot); s1 += s.v; s2 += s.w; }
    block_sum2<T, false>(s1, s2, scratch);               // doubles 0..35; the last collective used another area
    const double coef = s1 / s2;
#pragma unroll
    for (int j = 0; j < kLo; ++j) ulo[j] = ulo[j] - wlo[j] * coef;     // 0 - 0 * coef beyond the window
    return coef;
  };

  // A transform whose inputs the thread already holds: element tid + r T of the N/2-point buffer is the thread's own
  // sample r (kLo = 8 = the radix), so the FIRST stage runs from registers -- no input pass through LDS (eight 16-byte
  // stores at ~13 cycles each, the LDS's slowest instruction, eight loads and a barrier per transform; the band
  // transforms below always worked this way).
  static_assert(kLo == 8 && T * 8 * 2 == NMAX, "one radix-8 butterfly per thread");
  auto cfft_from_registers = [&](cplx (&a)[8], const cplx &w1) __attribute__((always_inline)) {      // w1: twiddle(tw, tid, lgn - 1, -1)
    constexpr int sh = lgn - 1 - 3;
    dft_reg<true, 3>(a);
    mul_powers<3>(a, w1);
    const int s0 = swz(tid);
#pragma unroll
    for (int k = 0; k < 8; ++k) Z[s0 ^ swz(k << sh)] = a[k];
    DifStages<lgn - 1, 3, lgn - 1 - 3, T>::run(Z, tw);
    __syncthreads();
  };

  // ---- GetStaticCentroid (d4c.cpp:90-143) -------------------------------------
  // The centroid of the two positions is summed in LDS (`park`, natural bin order: a thread only ever touches its
  // own pairs): nothing but the window's samples waits in registers through the transforms.
#pragma unroll 1
  for (int c = 0; c < 2; ++c) {
    const D4cWin w = d4c_win(x, x_len, fs, cf0, c == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0, kBlackman, 4.0,
                             noise + (size_t)c * wdraws);
    D4C_FRESH_TID();
    double ulo[kLo];
    const double coef = balanced(w, ulo);
    WH_STAMP(32, 1 + 4 * c);
    // even half: e[n] = z[n] + z[n + H], z[n] = u[n] (1 + i (n + 1)); the previous readers of Z are behind a barrier
    double pw = 0.0;
    {
      // the butterfly's eight inputs are the thread's own: e[tid + j T] from sample j (and, for a window longer than N/2,
      // from its upper sample j as well)
      cplx a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j].re = ulo[j]; a[j].im = ulo[j] * (tid + j * nt + 1.0);
        pw += ulo[j] * ulo[j];
      }
      hi_samples(w, coef, [&](int k, int i, double uh) { a[k].re += uh; a[k].im += uh * (i + 1.0); pw += uh * uh; });
      // |w x|^2 (d4c.cpp:104-107) is needed after the transform only: the waves' partial sums cross in scratch (doubles
      // kPwAt ..) behind the transform's own barriers
      pw = wave_sum(pw);
      if (lane_id() == 0) scratch[kPwAt + wave_in_block()] = pw;
      cfft_from_registers(a, twiddle(tw, tid, lgn - 1, -1));
    }
    pw = 0.0;
    for (int wv = 0; wv < wg_waves<T>(); ++wv) pw += scratch[kPwAt + wv];
    const double half_inv_pw = 0.5 / pw;                               // and the 1/2 of Im(P Q)/2
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      const int it = tid + m * nt;
      if (it <= q) {
        const cplx P = Z[fft_slot(plan, it)], Q = Z[fft_slot(plan, (H - it) & (H - 1))];
        const double v = (P.im * Q.re + Q.im * P.re) * half_inv_pw;    // d4c.cpp:115-116 at bin 2 it
        park[2 * it] = c == 0 ? v : park[2 * it] + v;
      }
    }
    WH_STAMP(32, 2 + 4 * c);
    __syncthreads();
    D4C_FRESH_TID();
    // odd half: o[n] = (z[n] - z[n + H]) W_N^n
    {
      cplx a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j].re = ulo[j]; a[j].im = ulo[j] * (tid + j * nt + 1.0); }
      hi_samples(w, coef, [&](int k, int i, double uh) { a[k].re -= uh; a[k].im -= uh * (i + 1.0); });
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = cmul(a[j], kRot ? mul_w16_fwd(wb, j) : twiddle(tw, tid + j * nt, lgn, -1));
      WH_STAMP(32, 3 + 4 * c);
      cfft_from_registers(a, twiddle(tw, tid, lgn - 1, -1));
    }
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      const int it = tid + m * nt;
      if (it < q) {
        const cplx P = Z[fft_slot(plan, it)], Q = Z[fft_slot(plan, H - 1 - it)];
        const double v = (P.im * Q.re + Q.im * P.re) * half_inv_pw;    // bin 2 it + 1
        park[2 * it + 1] = c == 0 ? v : park[2 * it + 1] + v;
      }
    }
    WH_STAMP(32, 4 + 4 * c);
    __syncthreads();
  }
  WH_STAMP(32, 9);
  D4C_FRESH_TID();
  lds_dead(Zr, N);                                         // (-DWH_LDS_POISON only: the centroid transforms' buffer is dead)

  // ---- GetSmoothedPowerSpectrum (d4c.cpp:149-166) ----------------------------
  double B[kBins];
  {
    const D4cWin w = d4c_win(x, x_len, fs, cf0, pos, kHanning, 4.0, noise + (size_t)2 * wdraws);
    double ulo[kLo];
    const double coef = balanced(w, ulo);
#pragma unroll
    for (int j = 0; j < kLo; ++j) {
      const int n = tid + j * nt;
      if (n < H) rfft_in(Z, n) = ulo[j];
    }
    for (int i = H + tid; i < N; i += nt) rfft_in(Z, i) = 0.0;
    for_hi(w, coef, [&](int i, double uh) { rfft_in(Z, i) = uh; });    // the thread's own slots
    WH_STAMP(32, 10);
    cfft();
    double Bn[kBins];
    rfft_merge_items_rot<kItems, T>(Z, lgn, plan, tw, wb, [&](int m, int, double ar, double ai, bool, double br, double bi) {
      Bn[2 * m] = keep(ar * ar + ai * ai); Bn[2 * m + 1] = keep(br * br + bi * bi);
    });
    WH_STAMP(32, 11);
    D4C_FRESH_TID();
    // DCCorrection without LDS: a bin k <= upper of the merge's natural order is thread k's first item, so while the
    // mirrored bins sit in lanes of the first wavefront (upper < 64: F0 below ~715 Hz at 48 kHz) the two neighbours of
    // the interpolation come by lane index -- no staging of the low bins in Z, and none of the pass's three barriers
    // (the waves that have nothing to correct walk on to the smoothing's first barrier).  Same operands, same order.
    if (2 + static_cast<int>(cf0 * N / fs) < WAVE) {
      const int upper = 2 + static_cast<int>(cf0 * N / fs), nrep = upper - 1;
      const double pos0 = (0.0 - cf0) * (-static_cast<double>(N) / fs);
      const int b0 = static_cast<int>(pos0);
      const double fr0 = pos0 - b0;
      if (wave_in_block() == 0) {
        const int b = b0 - lane_id();                         // 1 <= b <= upper - 2 for the lanes that are corrected
        const double y0 = __shfl(Bn[0], b & (WAVE - 1), WAVE), y1 = __shfl(Bn[0], (b + 1 <= upper ? b + 1 : upper) & (WAVE - 1), WAVE);
        if (lane_id() < nrep) Bn[0] = keep(Bn[0] + (y0 + (y1 - y0) * fr0));
      }
    } else
    dc_correct(Bn, for_nat);
    // (the transform's readers are behind the merge's closing barrier -- and dc_correct's own, if it ran: no barrier in front)
    smooth(Bn, for_nat, cf0, B, for_pair, false);
  }
  WH_STAMP(32, 12);
  lds_dead(Zr, N);                                         // (the smoothing segment is dead: DCCorrection below stages its own bins)

  // ---- GetStaticGroupDelay (d4c.cpp:172-188) ----------------------------------
  double A[kBins];
  // the centroid (the thread's own pairs) with its DCCorrection (d4c.cpp:141-142) on the way: the bins the correction
  // mirrors are in `park` already, in natural order -- no staging in Z, no barriers (round 4: three)
  {
    const int upper = 2 + static_cast<int>(cf0 * N / fs), nrep = upper - 1;
    const double pos0 = (0.0 - cf0) * (-static_cast<double>(N) / fs);
    const int b0 = static_cast<int>(pos0);
    const double fr0 = pos0 - b0;
    for_pair([&](int slot, int k) {
      double v = park[k];
      if (k < nrep) {
        const int b = b0 - k;
        const double y0 = park[b], y1 = park[b + 1 <= upper ? b + 1 : upper];
        v = v + (y0 + (y1 - y0) * fr0);
      }
      A[slot] = keep(v);
    });
  }
  WH_STAMP(32, 13);
  D4C_FRESH_TID();
#pragma unroll
  for (int e = 0; e < kBins; ++e) A[e] = fast_div(A[e], B[e]);        // the smoothed power spectrum is positive and normal
  D4C_FRESH_TID();
  smooth(A, for_pair, cf0 / 2.0, A, for_pair);
  WH_STAMP(32, 14);
  D4C_FRESH_TID();
  smooth(A, for_pair, cf0, B, for_pair);
  // The group delay takes the centroid sum's place in `park` (natural bin order, N/2 + 1 doubles): the band transforms
  // below read their slices from there -- no staging pass, and nothing bin-indexed in registers through the hottest loop.
  for_pair([&](int slot, int k) { park[k] = A[slot] - B[slot]; });
  WH_STAMP(32, 15);

  // ---- GetCoarseAperiodicity (d4c.cpp:194-225) per 3 kHz band ------------------
  const int bnd = mround(N * 8.0 / p.wl);
  const int hwl = p.wl / 2, wl = 2 * hwl + 1;
  const int nz = hwl + 1;                               // packed complex input elements that are not zero
  int mine = 0;
  for_nat([&](int, int) { ++mine; });
  int *hist = reinterpret_cast<int *>(Zr);
  constexpr int kWaves = (T + WAVE - 1) / WAVE;
  double *band_sums = park_global ? after_tw : park + (H + 2);   // [band][partial, total][wavefront]: d4c_frame_lds_bytes
  // the Nuttall taps of the thread's own slice element: the same for every band
  const double nut0 = 2 * tid < wl ? p.nuttall[2 * tid] : 0.0, nut1 = 2 * tid + 1 < wl ? p.nuttall[2 * tid + 1] : 0.0;
  // What every band's first stage needs and no band changes: the stage's twiddle (it used to be looked up again per band:
  // two LDS reads, the quadrant selects, the fine-level product) and the taps of the thread's SECOND slice element
  // (packed element tid + T; zero beyond the window -- at 48 kHz only thread 0 has one)
  constexpr int kNzR = (512 + T - 1) / T;               // a slice is at most 512 packed elements (setup_d4c checks): kNzR T >= nz
  const cplx w_first = twiddle(tw, tid, plan.lg, -1);
  const bool second = kNzR == 2 && tid + T < nz;
  double nut2 = 0.0, nut3 = 0.0;
  if (second) { nut2 = p.nuttall[2 * (tid + T)]; nut3 = 2 * (tid + T) + 1 < wl ? p.nuttall[2 * (tid + T) + 1] : 0.0; }
  const bool any_second = kNzR == 2 && __builtin_amdgcn_ballot_w64(second) != 0ull;      // wave-uniform
  for (int band = 0; band < p.nap; ++band) {
    const int lo_k = p.band_center[band] - hwl;
    D4C_FRESH_TID();
    __syncthreads();                                    // the previous band's histograms are done (first band: park is written)
    // First DIF stage with every input beyond element nz known to be zero: the slice comes straight from `park`.
    auto slice = [&](int n) {
      cplx v;
      const double *g = park + lo_k + 2 * n;
      if (n == tid) {
        v.re = g[0] * nut0;
        v.im = 2 * n + 1 < wl ? g[1] * nut1 : 0.0;
      } else {
        v.re = g[0] * p.nuttall[2 * n];
        v.im = 2 * n + 1 < wl ? g[1] * p.nuttall[2 * n + 1] : 0.0;
      }
      return v;
    };
    {
      constexpr int R = 8;
      const int sh = plan.lg - 3, qq = 1 << sh;         // qq == T: one butterfly per thread (launch_d4c)
      const int s0 = swz(tid);
      cplx z0; z0.re = 0.0; z0.im = 0.0;
      if constexpr (kNzR <= 2) {
        // Of the butterfly's eight inputs only the thread's own element (and, for a few threads of the first wavefronts,
        // the one T further on) can be non-zero: a wavefront without a second element multiplies ONE value by the
        // stage's twiddle powers; the others form a0 + a1 W8^k first (dft8_head2: the bits dft8 would give).  The general
        // form below asked for all eight under eight exec-mask branches, copied a0 seven times and ran the full dft8
        // for the one lane that needed it -- in wavefront 0 of every workgroup, eight times per frame.
        const cplx a0 = (kNzR == 2 || tid < nz) ? slice(tid) : z0;
        cplx a[R];
        if (any_second) {
          cplx a1 = z0;
          if (second) {
            const double *g = park + lo_k + 2 * (tid + T);
            a1.re = g[0] * nut2;
            a1.im = 2 * (tid + T) + 1 < wl ? g[1] * nut3 : 0.0;
          }
          dft8_head2<true>(a0, a1, a);
          if (!second) {
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = a0;     // (a0 + 0 W8^k: the DFT of a delta, exactly)
          }
          mul_powers<3>(a, w_first);
#pragma unroll
          for (int k = 0; k < R; ++k) Z[s0 ^ swz(k << sh)] = a[k];
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) a[r] = a0;       // DFT of a delta
          mul_powers<3>(a, w_first);
#pragma unroll
          for (int k = 0; k < R; ++k) Z[s0 ^ swz(k << sh)] = a[k];
        }
      } else {
        cplx a[R];
        const bool single = tid + qq >= nz;
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = (r < kNzR && tid + r * qq < nz) ? slice(tid + r * qq) : z0;
        if (single) {
#pragma unroll
          for (int r = 1; r < R; ++r) a[r] = a[0];       // DFT of a delta
        } else {
          dft_reg<true, 3>(a);
        }
        mul_powers<3>(a, w_first);
#pragma unroll
        for (int k = 0; k < R; ++k) Z[s0 ^ swz(k << sh)] = a[k];
      }
      DifStages<lgn - 1, 3, lgn - 1 - 3, T>::run(Z, tw);
      __syncthreads();
    }
    if (band == 0) WH_STAMP(32, 16);
    unsigned long long key[kBins];
#pragma unroll
    for (int e = 0; e < kBins; ++e) key[e] = D4C_KEY_PAD;
    // (the band's keys are 4 |X|^2: only the quotient of the band's two sums is ever used -- d4c_finish -- and that is scale-free)
    rfft_merge_items_rot<kItems, T, true>(Z, lgn, plan, tw, wb, [&](int m, int, double ar, double ai, bool paired, double br, double bi) {
      key[2 * m] = (unsigned long long)__double_as_longlong(ar * ar + ai * ai);
      if (paired) key[2 * m + 1] = (unsigned long long)__double_as_longlong(br * br + bi * bi);
    });
    if (band == 0) WH_STAMP(32, 17);
    // The select's histograms alias the head of the transform buffer and are zeroed as its first step: every wavefront
    // must have finished READING the transform before any of them starts -- the merge's own closing barrier
    // (rfft_merge_items_w) is that point.  (Round 5 had a second barrier here, back to back with that one: s_barrier,
    // s_barrier in the ISA, five times a frame.)
    lds_dead(Zr, N);                                       // (the band's transform is in registers: the select must zero what it counts in)
    double part, tot;
    (void)mine;
    // the bnd + 1 largest of the H + 1 bins excluded; each wavefront parks its share of the band's two sums
    const bool whole = block_excluding_largest<T>(key, bnd + 1, hist, scratch, &part, &tot, trace_me, N / 16);
    if (lane_id() == 0) {
      const int wv = wave_in_block();
      band_sums[(2 * band) * kWaves + wv] = whole && wv > 0 ? 0.0 : part;
      band_sums[(2 * band + 1) * kWaves + wv] = whole && wv > 0 ? 0.0 : tot;
    }
    if (band == 0) WH_STAMP(32, 18);
  }
  __syncthreads();
  for (int b = tid; b < 2 * p.nap; b += nt) {             // thread b: band b's partial sum, thread nap + b: its total
    const int band = b < p.nap ? b : b - p.nap, which = b < p.nap ? 0 : 1;
    const double *w = band_sums + (2 * band + which) * kWaves;
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) t += w[k];           // wavefront order, from zero: the block sum's own order
    p.coarse[fi * 16 + (which ? 9 : 1) + band] = t;
  }
  WH_STAMP(32, 19);
}

// ---------------------------------------------------------------------------
// Stage C: GetAperiodicity (d4c.cpp:323-338) -- every frame's output row.
__global__ void d4c_finish(D4cParams p) {
  const int u = blockIdx.y, f = p.frame_lo + blockIdx.x;
  if (f >= p.b.n_frames[u] || f >= p.frame_hi) return;
  const size_t fi = (size_t)u * p.b.f_stride + f;
  const int tid = threadIdx.x, nt = blockDim.x, fs = p.b.fs;
  const int nb_out = p.fft_out / 2 + 1;
  const size_t orow = p.out_row ? (size_t)p.out_row[u] + f : fi;
  char *row_at = reinterpret_cast<char *>(p.aperiodicity + orow * p.out_stride) + p.out_col_bytes;
  double *row = reinterpret_cast<double *>(row_at);
  float *row32 = reinterpret_cast<float *>(row_at);                    // the narrow wire format (p.out_f32)
  const double f0 = d4c_sane_f0(p.f0[fi], p.b.fs);
  if (p.rec && tid == 0) { double *r = p.rec + orow * p.out_stride; r[0] = p.tpos[fi]; r[1] = p.f0[fi]; }   // the record's head (F0 as it was given)
  // CodeAperiodicity of a row (codec.cpp:217-236; codec.hip: codec_code_ap), fused: band b's value is interp1Q of
  // 20 log10(row) at 3000 (b + 1) Hz -- two of the row's values, formed here exactly as the dense row would hold them
  auto code_bands = [&](auto ap_value) __attribute__((always_inline)) {
    for (int band = tid; band < p.code_nap; band += nt) {
      const double pos = (3000.0 * (band + 1.0) - 0) / (static_cast<double>(fs) / p.fft_out);
      const int b = static_cast<int>(pos);
      const double fr = pos - b;
      const double y0 = 20 * log10(ap_value(b));
      const double dy = b < nb_out - 1 ? 20 * log10(ap_value(b + 1)) - y0 : 0.0;
      row[band] = y0 + dy * fr;
    }
  };
  if (f0 == 0 || p.ap0[fi] <= p.threshold) {                          // d4c.cpp:323-328,386
    if (p.code_nap > 0) { code_bands([&](int) { return 1.0 - kTiny; }); return; }
    if (p.out_f32) { for (int i = tid; i < nb_out; i += nt) row32[i] = static_cast<float>(1.0 - kTiny); }
    else { for (int i = tid; i < nb_out; i += nt) row[i] = 1.0 - kTiny; }
    return;
  }
  // the bands' coarse aperiodicity in dB from the sums d4c_frame left (d4c.cpp:221-224, 314-316), once per frame
  DYN_LDS(lds);
  double *coarse_db = reinterpret_cast<double *>(lds);                 // [1 + nap]
  for (int b = tid; b < p.nap; b += nt) {
    const double cf0 = kFloorF0D4C > f0 ? kFloorF0D4C : f0;
    double c = 10 * log10(p.coarse[fi * 16 + 1 + b] / p.coarse[fi * 16 + 9 + b]);
    c = c + (cf0 - 100) / 50.0;
    coarse_db[1 + b] = c < 0.0 ? c : 0.0;
  }
  __syncthreads();
  const double *coarse_in = coarse_db;
  const int nk = p.nap + 2;
  // (the bin's knot and weight -- histc over the nap + 2 knots, a division -- are the same for every frame: api.hip builds
  // them once per (fs, fft_out) with the expressions this lambda used to evaluate per bin and frame: two thirds of the
  // kernel's vector instructions were not FP64)
  (void)nk;
  auto ap_value = [&](int i) __attribute__((always_inline)) {
    const int k = p.ap_knot[i];
    const double s = p.ap_frac[i];
    auto cval = [&](int j) { return j == 0 ? -60.0 : (j == p.nap + 1 ? -kTiny : coarse_in[j]); };   // d4c.cpp:373-375
    double y = cval(k - 1) + s * (cval(k) - cval(k - 1));
    return exp10(y / 20.0);                            // the reference: pow(10.0, y / 20.0) -- same value to an ulp or two, a third of the instructions
  };
  if (p.code_nap > 0) { code_bands(ap_value); return; }
  for (int i = tid; i < nb_out; i += nt) {
    const double v = ap_value(i);
    if (p.out_f32) row32[i] = static_cast<float>(v); else row[i] = v;
  }
}

// ---------------------------------------------------------------------------
size_t d4c_love_lds_bytes(int lg) { return sizeof(double) * (size_t)((1 << lg) + 64 + (1 << lg) / 8 + 2); }
// Z (N doubles) | scratch (64) | quarter-wave table of the N/2-point complex transform | the group delay (N/2 + 2; N <= 4096 only)
// | the wavefronts' shares of the bands' sums (8 bands x 2 x N / 1024 wavefronts)
size_t d4c_frame_lds_bytes(int lg) {
  return d4c_frame_direct_tw_offset(lg) + 2 * sizeof(double) * (((size_t)1 << (lg - 7)) + ((size_t)1 << (lg - 10)));
}
// doubles of one workgroup's slot of D4cParams::park_ws (0: the shape parks in LDS)
size_t d4c_park_slot_doubles(int lg) { return lg > 13 ? ((size_t)1 << (lg - 1)) + 2 : 0; }
int d4c_frame_threads(int lg) { return (1 << lg) / 16; }   // one radix-8 butterfly per thread and stage
// worst case per frame: LoveTrain window at 40 Hz + 3 body windows at 47 Hz
size_t d4c_max_draws_per_frame(int fs) {
  return (size_t)(2 * mround(3.0 * fs / 40.0 / 2.0) + 1) + 3 * (size_t)(2 * mround(4.0 * fs / kFloorF0D4C / 2.0) + 1);
}

void launch_spectral_prepare(const CtParams &cp, const D4cParams &dp, hipStream_t stream) {
  WH_BLOCKS(spectral_prepare, dim3(cp.b.n_utt, 2), 256, 64 * sizeof(double), stream, cp, dp);     // (four wavefronts: what finds room beside other jobs' frame kernels)
}

void launch_d4c(const D4cParams &p, int max_frames, hipStream_t stream) {
  if (!(p.skip_prepare & kD4cSkipScan)) WH_BLOCKS(d4c_prepare1, dim3(p.b.n_utt), 1024, 64 * sizeof(double), stream, p);
  if (!(p.skip_prepare & kD4cSkipLoveTrain) && max_frames > 0) {
    // workgroup sizes follow the transform size (threads beyond N/16 idle through every radix-8 stage): for the
    // 2048-point internal FFT of fs <= 24 kHz, 64 x 1001 frames: lovetrain 0.53 -> 0.42 ms, groupdelay 3.77 -> 2.52,
    // band 0.97 -> 0.65
    const dim3 love_grid(max_frames, p.b.n_utt);
    const size_t love_lds = d4c_love_lds_bytes(p.lg_love);
    if (p.lg_love == 11) devrt::launch_blocks("d4c_lovetrain", d4c_lovetrain<11>, love_grid, 128, love_lds, stream, p);
    else if (p.lg_love == 12) devrt::launch_blocks("d4c_lovetrain", d4c_lovetrain<12>, love_grid, 256, love_lds, stream, p);
    else devrt::launch_blocks("d4c_lovetrain", d4c_lovetrain<0>, love_grid, p.lg_love <= 11 ? 128 : 256, love_lds, stream, p);
  }
  // one radix-8 butterfly per thread: 128 / 256 / 512 threads for the 2048- / 4096- / 8192-point internal FFT
  // (fs <= 24 kHz / <= 48 kHz / <= 96 kHz); the register arrays are sized per shape
  const int range_frames = imin(max_frames, p.frame_hi) - p.frame_lo;     // frames of the range
  if (range_frames <= 0) return;
  const size_t lds = d4c_frame_lds_bytes(p.lg_d4c);
  // the 16384-point shape: as many frames per launch as the global staging area has slots for (stream order keeps a
  // slot's workgroups of consecutive launches apart)
  const int per_launch = p.lg_d4c > 13 ? imax(1, p.park_slots / p.b.n_utt) : range_frames;
  for (int lo = 0; lo < range_frames; lo += per_launch) {
    D4cParams q = p;
    q.frame_lo = p.frame_lo + lo;
    q.frame_hi = imin(q.frame_lo + per_launch, p.frame_lo + range_frames);
    const dim3 grid(q.frame_hi - q.frame_lo, p.b.n_utt);
    if (p.lg_d4c == 11) devrt::launch_blocks("d4c_frame", d4c_frame<2048, 128>, grid, 128, lds, stream, q);
    else if (p.lg_d4c == 12) devrt::launch_blocks("d4c_frame", d4c_frame<4096, 256>, grid, 256, lds, stream, q);
    else if (p.lg_d4c == 13) devrt::launch_blocks("d4c_frame", d4c_frame<8192, 512>, grid, 512, lds, stream, q);
    else devrt::launch_blocks("d4c_frame", d4c_frame<16384, 1024>, grid, 1024, lds, stream, q);
  }
  WH_BLOCKS(d4c_finish, dim3(range_frames, p.b.n_utt), 256, 8 * sizeof(double), stream, p);
}

}  // namespace world_hip
