// harvest_contour.hip -- Harvest, back half: candidates -> F0 contour.
//
// Reference: FixF0Contour (src/harvest.cpp:1027-1044) = SearchF0Base (:693-705),
// FixStep1 (:710-722), FixStep2 (:748-762), FixStep3 (:968-995: GetMultiChannelF0,
// Extend/ExtendF0/ExtendSub, MergeF0/MergeF0Sub), FixStep4 (:1000-1022), then
// SmoothF0Contour (:1079-1113) and the hop subsampling of Harvest() (:1246-1251).
//
// The reference is one serial pass per utterance.  Here everything that is local to a frame or to a voiced section
// runs in parallel, and the inherently ordered parts are kept short and staged in LDS.  Eight launches per batch:
//   hc_base            one wavefront per frame, lanes over the candidate slots
//   hc_step12          FixStep1 + FixStep2, a workgroup per tile of 256 frames (step 1 and its margin in LDS)
//   hc_sections        the voiced runs of step 2's contour: one workgroup per utterance, flags as bit words, one scan
//   hc_extend          a workgroup per section: two wavefronts track candidates forwards / backwards ("nearest
//                      candidate, last wins" as ONE packed (distance, slot) minimum), all eight copy and sum
//   hc_merge           a workgroup per utterance: section selection with the reference's running mean, the merge of
//                      overlapping sections as a list of copy records, the contour written once at the end
//   hc_sections_step4  sections of step 3's contour, FixStep4's gap bridging, sections of the patched contour
//   hc_smooth          one wavefront per section: zero-phase Butterworth, in place in LDS
//   hc_output          Harvest()'s hop subsampling
#include "harvest.h"

namespace world_hip {

__device__ __forceinline__ double *hc_row(double *base, const HarvestParams &p, int u) {
  return base + (size_t)u * p.fb_stride;
}

// (SearchF0Base, :693-705 -- the candidate with the highest score per frame, into c0 -- rides on hv_prune: harvest.hip.)
// ---- FixStep1 (:710-722) + FixStep2 (:748-762) in one launch -----------------------------------------
// Step 1 drops a frame whose F0 jumps from the line through its two predecessors; step 2 removes voiced runs with
// end - start < 6, which needs step 1's result six frames to either side: a workgroup evaluates step 1 for its
// kStepTile frames and that margin into LDS, then step 2 out of it (two kernels and a contour row in HBM before).
constexpr int kStepMargin = 6;
__device__ __forceinline__ double hc_step1_at(const double *base, int g, int nf) {
  double v = 0.0;
  if (g >= 2 && g < nf) {
    const double b0 = base[g], b1 = base[g - 1], b2 = base[g - 2];   // (all three at once: one trip to memory)
    if (b0 != 0.0) {
      double ref = b1 * 2 - b2;
      v = fabs((b0 - ref) / ref) > 0.008 && fabs((b0 - b1)) / b1 > 0.008 ? 0.0 : b0;
    }
  }
  return v;
}
// Steps 1 and 2 of the whole utterance by its one workgroup (the section pass that follows needs the utterance's workgroup
// anyway -- hc_step12_sections -- and a launch of tiles for 5 us of work was one more narrow kernel per job): tile after
// tile of 1024 frames, step 1 of the tile and its six-frame margins into LDS (its two divisions per frame are the pass's
// cost), step 2 out of it.  The workgroup is 256 threads -- four wavefronts of <= 128 registers: what fits beside other
// jobs' frame kernels in the in-flight mode (1024 threads waited for a drained CU: 52 us in flight against 15 alone).
constexpr int kStepTile = 4096;       // frames per tile, whatever the workgroup size (three tiles for a 10 s utterance)
constexpr int kStepRun = 16;          // step 2: consecutive frames per thread and trip
// one pad slot per kStepRun entries: the threads of a wavefront read runs that start kStepRun doubles apart
__host__ __device__ __forceinline__ int hc_step_pad(int i) { return i + i / kStepRun; }
constexpr int kStepLds = kStepTile + 2 * kStepMargin + (kStepTile + 2 * kStepMargin) / kStepRun + 1;   // doubles
__device__ __forceinline__ void hc_step12_pass(const HarvestParams &p, int u, double *s1) {
  const int nf = p.nfb[u], nt = (int)blockDim.x, tid = (int)threadIdx.x;
  const double *base = hc_row(p.c0, p, u);
  double *out = hc_row(p.c2, p, u);
  for (int f0 = 0; f0 < nf; f0 += kStepTile) {
    // s1[pad(i)] = step 1 of frame f0 - kStepMargin + i, i < kStepTile + 2 kStepMargin (thread-strided: coalesced loads)
    for (int i = tid; i < kStepTile + 2 * kStepMargin; i += nt) s1[hc_step_pad(i)] = hc_step1_at(base, f0 - kStepMargin + i, nf);
    __syncthreads();
    // Step 2 on runs of kStepRun consecutive frames per thread: the run's step-1 values and its two margins come out of
    // LDS together (independent reads), their voicing becomes the bits of a word, and "how many voiced neighbours in a
    // row, up to six either way" is a count of leading / trailing ones -- no chain of dependent LDS reads per frame
    // (a walk frame by frame, neighbour by neighbour, was most of this pass once one workgroup did the whole utterance).
    for (int r0 = tid * kStepRun; r0 < kStepTile; r0 += nt * kStepRun) {
      if (f0 + r0 >= nf) break;
      double v[kStepRun + 2 * kStepMargin];
#pragma unroll
      for (int i = 0; i < kStepRun + 2 * kStepMargin; ++i) v[i] = s1[hc_step_pad(r0 + i)];
      unsigned vm = 0;                                     // bit i <-> frame f0 + r0 - kStepMargin + i is voiced after step 1
#pragma unroll
      for (int i = 0; i < kStepRun + 2 * kStepMargin; ++i) {
        const int g = f0 + r0 - kStepMargin + i;
        vm |= (unsigned)(g > 0 && g < nf - 1 && v[i] > 0) << i;          // ends forced unvoiced (:733)
      }
#pragma unroll
      for (int t = 0; t < kStepRun; ++t) {
        const int f = f0 + r0 + t, b = t + kStepMargin;
        if (f >= nf) break;
        double val = v[b];
        if ((vm >> b) & 1u) {
          const unsigned below = ~(vm << (32 - b));       // frame f-1 at bit 31, f-2 at bit 30, ...: ones = voiced
          const unsigned above = ~(vm >> (b + 1));        // frame f+1 at bit 0, ...
          const int back = imin(6, __builtin_clz(below | 1u)), fwd = imin(6, __builtin_ctz(above | 0x80000000u));
          if (back + fwd < 6) val = 0.0;
        }
        out[f] = val;
      }
    }
    __syncthreads();                                                    // the next tile overwrites s1
  }
}

// ---- GetBoundaryList (:727-743) as a block scan --------------------------------
// sec[u][0][k] = start, sec[u][1][k] = end of the k-th voiced run of `src`;
// sec[u][4][k] = offset of the run's private slice (length end-start+1+extra),
// sec_n[u][0] = number of runs.  One workgroup per utterance.
constexpr int kSmoothTail = 300;     // samples the smoother holds a section's end values for (SmoothF0Contour, below)
struct SecArgs {
  const double *src; int force_ends; int extra;
  double *copy_to;      // != nullptr: every frame of src is also copied there (the pass reads them all anyway)
  double *zero_to;      // != nullptr: the unvoiced frames are set to 0 there
};

__device__ __forceinline__ void hc_sections_pass(const HarvestParams &p, const SecArgs &a, int u, double *scratch) {
  const int nf = p.nfb[u];
  const double *in = a.src + (size_t)u * p.fb_stride;
  int *st = p.sec + (size_t)u * 6 * p.sec_cap, *ed = st + p.sec_cap, *off = st + 4 * p.sec_cap;
  // every thread owns a run of consecutive frames: ONE block scan for the whole utterance (a scan per 1024
  // frames cost ten barrier rounds, 22 us per launch)
  const int chunk = (nf + (int)blockDim.x - 1) / (int)blockDim.x;
  const int lo = (int)threadIdx.x * chunk, hi = imin(nf, lo + chunk);
  // The run's voiced flags are fetched once, sixteen loads in flight, into the bits of a word (a walk that waited for
  // one frame per trip spent 13 us per launch on memory latency, three launches per job); starts and ends of voiced
  // runs are then bit patterns.  A word covers a piece of <= 62 frames plus a neighbour on either side; longer runs
  // (utterances beyond 63 488 base frames) take several pieces.
  constexpr int kPiece = 62, kLoads = 16;
  typedef unsigned long long u64;
  auto piece_runs = [&](int base, int n, u64 &starts, u64 &ends, bool side_effects) {   // bit i of either <-> frame base + i, i < n
    u64 m = 0;                                                            // bit i <-> frame base - 1 + i is voiced
    for (int b = 0; b < n + 2; b += kLoads) {
      double v[kLoads];
#pragma unroll
      for (int q = 0; q < kLoads; ++q) v[q] = in[imax(0, imin(nf - 1, base - 1 + b + q))];
#pragma unroll
      for (int q = 0; q < kLoads; ++q) {
        const int f = base - 1 + b + q;
        const bool ok = b + q < n + 2 && f >= 0 && f < nf && !(a.force_ends && (f == 0 || f == nf - 1)) && v[q] > 0;
        m |= (u64)ok << (b + q);
        if (side_effects && b + q >= 1 && b + q <= n) {                 // the piece's own frames, once
          if (a.copy_to) a.copy_to[(size_t)u * p.fb_stride + f] = v[q];
          if (a.zero_to && !ok) a.zero_to[(size_t)u * p.fb_stride + f] = 0.0;
        }
      }
    }
    const u64 span = (1ull << n) - 1, cur = (m >> 1) & span;
    starts = cur & ~m;                                                    // voiced, the frame before is not
    ends = cur & ~(m >> 2);                                               // voiced, the frame after is not
  };
  int mine = 0;                                            // starts in the low half-word, ends in the high one
  u64 starts0 = 0, ends0 = 0;                              // the first piece's patterns are kept for the second pass
  for (int base = lo; base < hi; base += kPiece) {
    u64 s_, e_;
    piece_runs(base, imin(kPiece, hi - base), s_, e_, true);
    if (base == lo) { starts0 = s_; ends0 = e_; }
    mine += __builtin_popcountll(s_) + (__builtin_popcountll(e_) << 16);
  }
  // the two counts share one scan as 16-bit halves while they cannot overflow them (fewer than 2^15 sections: any
  // utterance below 65 s); longer ones pay a second scan instead of corrupting the end offsets
  int n_start, at_s0, at_e0;
  if (nf < 65536) {
    int tot;
    const int at = block_excl_scan_int(mine, &tot, scratch);
    n_start = tot & 0xFFFF; at_s0 = at & 0xFFFF; at_e0 = at >> 16;
  } else {
    int tot_s, tot_e;
    at_s0 = block_excl_scan_int(mine & 0xFFFF, &tot_s, scratch);
    at_e0 = block_excl_scan_int(mine >> 16, &tot_e, scratch);
    n_start = tot_s;
  }
  {
    int at_s = at_s0, at_e = at_e0;
    for (int base = lo; base < hi; base += kPiece) {
      u64 s_ = starts0, e_ = ends0;
      if (base != lo) piece_runs(base, imin(kPiece, hi - base), s_, e_, false);
      for (; s_; s_ &= s_ - 1, ++at_s) if (at_s < p.sec_cap) st[at_s] = base + __builtin_ctzll(s_);
      for (; e_; e_ &= e_ - 1, ++at_e) if (at_e < p.sec_cap) ed[at_e] = base + __builtin_ctzll(e_);
    }
  }
  __syncthreads();
  const int ns = imin(n_start, p.sec_cap);
  if (threadIdx.x == 0) {
    p.sec_n[u * 2] = ns;
    p.sec_n[u * 2 + 1] = 0;
  }
  // offsets of the sections' private slices: exclusive scan of their lengths (a serial loop of one thread over
  // global memory here cost ~1 us per section: 20 us per launch, three launches per job)
  int running = 0;
  for (int base = 0; base < ns; base += blockDim.x) {
    const int k = base + threadIdx.x;
    const int len = k < ns ? ed[k] - st[k] + 1 + a.extra : 0;
    int tot, at = block_excl_scan_int(len, &tot, scratch);
    if (k < ns) off[k] = running + at;
    running += tot;
  }
}
// FixStep1 + FixStep2 of the utterance, then the sections of their result (c2): one launch, the utterance's workgroup
__global__ void hc_step12_sections(HarvestParams p, SecArgs a) {
  DYN_LDS(lds);
  hc_step12_pass(p, blockIdx.x, reinterpret_cast<double *>(lds) + 64);
  __syncthreads();                                         // c2 is read back by other threads of this workgroup
  hc_sections_pass(p, a, blockIdx.x, reinterpret_cast<double *>(lds));
}

// FixStep4 (:1000-1022) between two section passes, one launch: the sections of step 3's contour (c3, copied to c0 on
// the way), the short unvoiced gaps between them bridged linearly in c0, then the sections of the patched contour for
// the smoother (and basic_f0's unvoiced frames zeroed on the way).  All by the utterance's one workgroup: the patches
// depend on the first pass's lists, the second pass on the patches.
__device__ __forceinline__ void hc_sections_step4_pass(const HarvestParams &p, int u, double *scratch) {
  const SecArgs a3 = {p.c3, 1, 0, p.c0, nullptr};
  hc_sections_pass(p, a3, u, scratch);
  __syncthreads();
  {
    const int ns = p.sec_n[u * 2];
    const int *sec = p.sec + (size_t)u * 6 * p.sec_cap;
    const double *in = hc_row(p.c3, p, u);
    double *out = hc_row(p.c0, p, u);
    for (int k = threadIdx.x; k < ns - 1; k += blockDim.x) {
      const int ed = sec[p.sec_cap + k], nst = sec[k + 1];
      const int dist = nst - ed - 1;
      if (dist >= 9) continue;
      double t0 = in[ed] + 1, t1 = in[nst] - 1;
      double coef = (t1 - t0) / (dist + 1.0);
      int c = 1;
      for (int j = ed + 1; j <= nst - 1; ++j) out[j] = t0 + coef * c++;
    }
  }
  __syncthreads();
  const SecArgs a4 = {p.c0, 0, kSmoothTail, nullptr, p.basic_f0};
  hc_sections_pass(p, a4, u, scratch);
}

// ---- FixStep3, part 1: Extend() per section (:791-878) --------------------------
constexpr int kMaxSlots = 256;       // candidate slots per frame handled by the tracking lanes (maxc <= 256)
constexpr int kExtReach = 100;       // frames a section may grow in each direction (:865)
constexpr int kExtMargin = kExtReach + 1;

// One workgroup per section.  The forward and the backward extension do not depend on each other (each starts from
// its end of the section and writes its own side of the slice), so two wavefronts walk them side by side; the
// bulk work around the walks -- copying the section into its slice, summing the extended run -- is spread over all
// kExtendThreads (two wavefronts took a trip to HBM per 1024 frames: 16 us for an 8 000-frame section).
// Round 5: four wavefronts of at most 168 registers and kExtendBlocks workgroups per utterance that stride over its
// sections.  (Rounds 3-4: eight wavefronts of 197 registers and one workgroup per POSSIBLE section -- 1 432 for a 10 s
// utterance, all but ~50 of which only exit.  In the twelve-jobs-in-flight mode every one of them, the empty ones
// included, needed two wavefront slots of 197 registers per SIMD, i.e. a CU that other jobs' frame kernels had drained:
// 149 us in flight against 35 alone, the third largest kernel of the trace.  A workgroup now fits what ONE retiring
// d4c_frame workgroup frees.)
constexpr int kExtendThreads = 4 * WAVE;
constexpr int kExtendBlocks = 48;
#ifndef HC_EXT_RING
#define HC_EXT_RING 4                   // batches of eight candidate rows in flight ahead of the walk (one slot per lane)
#endif
__device__ __forceinline__ void hc_extend_section(const HarvestParams &p, int k, int u, double *scratch) {
  const int nf = p.nfb[u], nslot = p.nc[u] * 7, lane = lane_id();
  int *sec = p.sec + (size_t)u * 6 * p.sec_cap;
  const int st = sec[k], ed = sec[p.sec_cap + k];
  const int lo = st - kExtMargin;                       // frame index of slice element 0
  double *e = p.ext + (size_t)u * p.ext_cap + sec[4 * p.sec_cap + k];
  const double *in = hc_row(p.c2, p, u);
  const int len = ed - st + 1 + 2 * kExtMargin;
  {                                                      // GetMultiChannelF0 (:767-778), eight loads in flight per thread
    constexpr int kB = 8;
    const int nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < len; i0 += kB * nt) {
      double v[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) { const int f = lo + i0 + q * nt; v[q] = (i0 + q * nt < len && f >= st && f <= ed) ? in[f] : 0.0; }
#pragma unroll
      for (int q = 0; q < kB; ++q) if (i0 + q * nt < len) e[i0 + q * nt] = v[q];
    }
  }
  __syncthreads();
  const double *cands = p.cand_a + (size_t)u * p.fb_stride * p.maxc;
#ifdef WORLD_EMU
  const int dir_begin = 0, dir_end = 2;                 // the one emulated thread walks both directions
#else
  const int dir_begin = wave_in_block(), dir_end = imin(2, dir_begin + 1);      // wavefronts 2.. do not walk
#endif
  for (int dir = dir_begin; dir < dir_end; ++dir) {     // 0: forwards from the section's end, 1: backwards from its start
    const int shift = dir == 0 ? 1 : -1;
    const int origin = dir == 0 ? ed : st;
    const int last = dir == 0 ? imin(nf - 2, ed + kExtReach) : imax(1, st - kExtReach);
    const int dist = last > origin ? last - origin : origin - last;
    double cur = e[origin - lo];
    int moved = origin, miss = 0;
    // The frames to visit do not depend on the tracking result, so their candidate rows are fetched kAhead frames at
    // a time, batches ahead of the one being walked (a ring of buffers in registers: a batch takes ~2 us to arrive
    // and ~1 us to walk), and the ordered part -- nearest candidate of the previous pick -- runs from registers.
    // SPL = slots per lane: 1 while the candidates fit the wavefront (nslot <= 64, the usual case: a quarter of the
    // loads and compares), kMaxSlots / WAVE otherwise; NBUF = ring size.
    bool stop = false;
    auto track = [&](auto spl_c, auto nbuf_c) __attribute__((always_inline)) {
      constexpr int SPL = decltype(spl_c)::value, NBUF = decltype(nbuf_c)::value;
      constexpr int kAhead = 8;
      auto fetch = [&](double (&row)[kAhead][SPL], int i0) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
          // plain loads at clamped addresses: a load under a condition is waited for where its branch rejoins, which
          // put the rows of a batch in a queue (one trip to HBM each) instead of in flight together.  walk() never
          // looks at a frame beyond `dist` or a slot beyond nslot, so what the clamped loads fetch there is unused.
          const int t = imax(0, imin(nf - 1, origin + shift * imin(i0 + a, dist) + shift));
#pragma unroll
          for (int r = 0; r < SPL; ++r) row[a][r] = cands[(size_t)t * p.maxc + imin(lane + r * WAVE, p.maxc - 1)];
        }
      };
      auto walk = [&](const double (&row)[kAhead][SPL], int i0) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
          if (i0 + a > dist || stop) break;
          const int t = origin + shift * (i0 + a) + shift;
          // nearest candidate within 18 %, ties -> the LAST slot (SelectBestF0, :636-650).  The
          // reference compares err = |cur - c| / cur; cur is common to all slots, so the distances
          // are ordered first (no division on the frame-to-frame chain) and the 18 % test is made
          // once, for the winner -- by a product unless it is within 1e-12 of the boundary.
          double best_d = 1e300, best_v = 0.0;
          int best_i = -1;
#pragma unroll
          for (int r = 0; r < SPL; ++r) {
            const int sl = lane + r * WAVE;
            if (sl < nslot) {
              const double d = fabs(cur - row[a][r]);
              if (!(d > best_d)) { best_d = d; best_i = sl; best_v = row[a][r]; }
            }
          }
#ifndef WORLD_EMU
          {
            // One reduction carries distance AND slot: the slot rides in the low mantissa byte of the distance
            // (255 - slot, so that the later slot of two equal distances is the smaller key).  Distances that agree to
            // 2^-44 count as equal -- two candidates equally near the previous pick to thirteen digits.  DPP +
            // readlane, no LDS round trips on the frame-to-frame chain; the winner's exact distance is recomputed.
            static_assert(kMaxSlots <= 256, "the slot index rides in one byte");
            const long long kb = (__double_as_longlong(best_d) & ~0xFFll) | (long long)(0xFF - (best_i & 0xFF));
            const double kmin = wave_min_nonneg(__longlong_as_double(kb));
            const int win = 0xFF - (__builtin_amdgcn_readfirstlane(__double2loint(kmin)) & 0xFF);
            // (keep(): left to itself the compiler turns the selects into one load at a computed index, and an array
            // indexed at run time lives in scratch memory)
            double mine = row[a][0];
#pragma unroll
            for (int r = 1; r < SPL; ++r) mine = (win >> 6) == r ? keep(row[a][r]) : mine;
            best_v = readlane_f64(mine, win & (WAVE - 1));
            best_i = kmin < 1e299 ? win : -1;
            best_d = fabs(cur - best_v);
          }
#endif
          if (best_i >= 0) {
            const double bound = 0.18 * cur;
            bool ok = best_d <= bound;
            if (fabs(best_d - bound) <= 1e-12 * cur) ok = !(best_d / cur > 0.18);
            if (!ok) best_i = -1;
          }
          const double v = best_i < 0 ? 0.0 : best_v;
          if (lane == 0) e[t - lo] = v;
          if (v == 0.0) { miss++; } else { cur = v; miss = 0; moved = t; }
          if (miss == 4) stop = true;
        }
      };
      static_assert(NBUF == 2 || NBUF == 4, "the ring is written out for two and four buffers");
      double r0[kAhead][SPL], r1[kAhead][SPL];
      if constexpr (NBUF == 4) {
        double r2[kAhead][SPL], r3[kAhead][SPL];
        fetch(r0, 0); fetch(r1, kAhead); fetch(r2, 2 * kAhead);
        for (int i0 = 0; i0 <= dist && !stop; i0 += 4 * kAhead) {
          fetch(r3, i0 + 3 * kAhead); walk(r0, i0);
          if (stop || i0 + kAhead > dist) break;
          fetch(r0, i0 + 4 * kAhead); walk(r1, i0 + kAhead);
          if (stop || i0 + 2 * kAhead > dist) break;
          fetch(r1, i0 + 5 * kAhead); walk(r2, i0 + 2 * kAhead);
          if (stop || i0 + 3 * kAhead > dist) break;
          fetch(r2, i0 + 6 * kAhead); walk(r3, i0 + 3 * kAhead);
        }
      } else {
        fetch(r0, 0);
        for (int i0 = 0; i0 <= dist && !stop; i0 += 2 * kAhead) {
          fetch(r1, i0 + kAhead); walk(r0, i0);
          if (stop || i0 + kAhead > dist) break;
          fetch(r0, i0 + 2 * kAhead); walk(r1, i0 + kAhead);
        }
      }
    };
    constexpr int kSlotsPerLane = (kMaxSlots + WAVE - 1) / WAVE;
    if (kSlotsPerLane > 1 && nslot <= WAVE) track(std::integral_constant<int, 1>{}, std::integral_constant<int, HC_EXT_RING>{});
    else track(std::integral_constant<int, kSlotsPerLane>{}, std::integral_constant<int, 2>{});
    if (lane == 0) sec[(dir == 0 ? 3 : 2) * p.sec_cap + k] = moved;     // new end / new start
  }
  __syncthreads();
  const int new_st = sec[2 * p.sec_cap + k], new_ed = sec[3 * p.sec_cap + k];
  // sum over [new_st, new_ed) for ExtendSub's running mean (:850)
  double s = 0.0;
  const int nt = blockDim.x;
  for (int f0 = new_st + (int)threadIdx.x; f0 < new_ed; f0 += 8 * nt) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = f0 + q * nt < new_ed ? e[f0 + q * nt - lo] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) {
    sec[5 * p.sec_cap + k] = lo;
    p.sec_sum[(size_t)u * p.sec_cap + k] = s;
  }
}
__global__ void __launch_bounds__(kExtendThreads, 3) hc_extend(HarvestParams p) {
  DYN_LDS(lds);
  const int u = blockIdx.y, ns = p.sec_n[u * 2];
  for (int k = blockIdx.x; k < ns; k += gridDim.x) {
    hc_extend_section(p, k, u, reinterpret_cast<double *>(lds));
    __syncthreads();                                       // the next section reuses the scratch
  }
}

// ---- FixStep3, part 2: ExtendSub + MergeF0, one wavefront per utterance ----------
__device__ __forceinline__ double wave_best_score(double f0, const double *c, const double *s, int nslot) {
  double r = 0.0;                                        // SearchScore (:901-907)
  for (int i = lane_id(); i < nslot; i += WAVE)
    if (f0 == c[i] && r < s[i]) r = s[i];
  return wave_max(r);
}

// The reference's way -- copy a section into the contour as soon as it is decided, read the contour back in
// MergeF0Sub -- by one wavefront with the records where hc_extend left them: the route of an utterance with more
// sections than hc_merge's LDS holds (minutes of speech in one piece).
__device__ __forceinline__ void hc_merge_in_hbm(const HarvestParams &p, int u) {
  const int lane = lane_id(), nf = p.nfb[u], nslot = p.nc[u] * 7;
  const int ns = p.sec_n[u * 2];
  int *sec = p.sec + (size_t)u * 6 * p.sec_cap;
  int *order = sec;                                       // original starts are no longer needed
  int *b_st = sec + 2 * p.sec_cap, *b_ed = sec + 3 * p.sec_cap;
  int *s_off = sec + 4 * p.sec_cap, *s_lo = sec + 5 * p.sec_cap;
  const double *sums = p.sec_sum + (size_t)u * p.sec_cap;
  auto work = [&](auto order, auto b_st, auto b_ed, auto s_off, auto s_lo, auto sums) __attribute__((always_inline)) {
    const double *ext = p.ext + (size_t)u * p.ext_cap;
    const double *step2 = hc_row(p.c2, p, u);
    double *out = hc_row(p.c3, p, u);
    const double *cands = p.cand_a + (size_t)u * p.fb_stride * p.maxc;
    const double *scores = p.score_a + (size_t)u * p.fb_stride * p.maxc;

    // ExtendSub (:840-856): stable compaction of the sections longer than 2200/mean_f0;
    // mean_f0 is deliberately NOT reset between sections.
    int kept = 0;
    if (lane == 0) {
      double mean = 0.0;
      for (int s = 0; s < ns; ++s) {
        int st = b_st[s], ed = b_ed[s];
        mean += sums[s];
        mean /= ed - st;
        if (2200.0 / mean < ed - st) {
          int t;
          t = b_st[kept]; b_st[kept] = b_st[s]; b_st[s] = t;
          t = b_ed[kept]; b_ed[kept] = b_ed[s]; b_ed[s] = t;
          t = s_off[kept]; s_off[kept] = s_off[s]; s_off[s] = t;
          t = s_lo[kept]; s_lo[kept] = s_lo[s]; s_lo[s] = t;
          kept++;
        }
      }
      p.sec_n[u * 2 + 1] = kept;
      // MakeSortedOrder (:883-896), quirks included
      for (int i = 0; i < kept; ++i) order[i] = i;
      for (int i = 1; i < kept; ++i)
        for (int j = i - 1; j >= 0; --j) {
          if (b_st[order[j]] > b_st[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
          else break;
        }
    }
    wave_sync();
    kept = wave_bcast_int(kept, 0);
    // out[f] = value(f) for f in [lo, hi]: 32 loads in flight per lane (a plain loop has each iteration's load
    // wait behind the previous store: one trip to HBM per 64 frames, 80 us for the 10 001 frames of a 10 s utterance)
    auto fill = [&](int lo, int hi, auto value) __attribute__((always_inline)) {
      constexpr int kB = 32;    // the wavefront is alone on its SIMD: registers are free, trips to HBM are not
      for (int f0 = lo + lane; f0 <= hi; f0 += kB * WAVE) {
        double v[kB];
  #pragma unroll
        for (int q = 0; q < kB; ++q) v[q] = f0 + q * WAVE <= hi ? value(f0 + q * WAVE) : 0.0;
  #pragma unroll
        for (int q = 0; q < kB; ++q) if (f0 + q * WAVE <= hi) out[f0 + q * WAVE] = v[q];
      }
    };
    if (kept == 0) {
      fill(0, nf - 1, [&](int f) { return step2[f]; });
      return;
    }
    // value of channel `ch` at frame f: its slice inside its extended run, zero elsewhere
    auto chan = [&](int ch, int f) {
      return (f >= b_st[ch] && f <= b_ed[ch]) ? ext[s_off[ch] + (f - s_lo[ch])] : 0.0;
    };
    // the same for a whole range, the channel's record read once
    auto fill_chan = [&](int lo, int hi, int ch) __attribute__((always_inline)) {
      const int c_st = b_st[ch], c_ed = b_ed[ch];
      const double *src = ext + s_off[ch] - s_lo[ch];
      fill(lo, hi, [&](int f) { return (f >= c_st && f <= c_ed) ? src[f] : 0.0; });
    };
    // MergeF0 (:937-963)
    fill_chan(0, nf - 1, 0);
    wave_sync();
    int cur_st = b_st[0], cur_ed = b_ed[0];                 // the reference's boundary_list[0], [1]
    for (int i = 1; i < kept; ++i) {
      const int o = order[i];
      // the reference reads boundary_list[o*2(+1)] AFTER possibly having overwritten entry 0
      const int st2 = o == 0 ? cur_st : b_st[o];
      const int ed2 = o == 0 ? cur_ed : b_ed[o];
      if (st2 - cur_ed > 0) {
        fill_chan(st2, ed2, o);
        cur_st = st2; cur_ed = ed2;
      } else {
        // MergeF0Sub (:912-932)
        const int st1 = cur_st, ed1 = cur_ed;
        if (st1 <= st2 && ed1 >= ed2) { cur_ed = ed1; wave_sync(); continue; }
        double s1 = 0.0, s2 = 0.0;
        for (int f = st2; f <= ed1; ++f) {
          s1 += wave_best_score(out[f], cands + (size_t)f * p.maxc, scores + (size_t)f * p.maxc, nslot);
          s2 += wave_best_score(chan(o, f), cands + (size_t)f * p.maxc, scores + (size_t)f * p.maxc, nslot);
        }
        if (s1 > s2) fill_chan(ed1, ed2, o);
        else fill_chan(st2, ed2, o);
        cur_ed = ed2;
      }
      wave_sync();
    }
  };
  work(order, b_st, b_ed, s_off, s_lo, sums);
}

// One workgroup per utterance.  The ordered parts -- ExtendSub's running mean, the sort, the merge decisions -- are
// walked by the first wavefront over section records staged in LDS; the contour is written once, at the end, by all
// kMergeThreads threads.  The reference's MergeF0 copies a section into the contour as soon as it is decided and
// MergeF0Sub reads the contour back; here a decision appends a record (first frame, last frame, channel) to a list,
// "the contour at frame f" is the last record covering f, and the copy happens when the list is complete: one trip
// to HBM for the whole contour instead of one per 2 048 frames and section (20 of a lone job's 27 us).  MergeF0Sub's
// scores are looked up a frame per lane with all slots of the frame in flight, and summed in frame order -- the
// reference's order -- instead of two dependent trips to HBM per frame (4 us a frame: 1.4 ms for the slowest
// utterance of a 128-batch).
constexpr int kMergeThreads = 256;       // (round 4: 1024 -- sixteen wavefronts that waited, in the in-flight mode, for a CU other jobs had drained;
constexpr int kMergeThreadsLone = 1024;  //  round 6: both, chosen by the call -- HarvestParams::lone_job)
constexpr int kMergeLdsSections = 1024;  // section records kept in LDS (41 KB: what one retiring d4c_frame workgroup frees on a
                                         // CU -- decimate.h); an utterance with more voiced sections takes hc_merge_in_hbm
inline size_t hc_merge_lds_bytes(int sections) {      // (at least a block collective's scratch: the section passes behind the merge)
  const size_t b = (size_t)(sections + 1) * (sizeof(double) + 8 * sizeof(int)) + 16 * sizeof(int);
  return b > 64 * sizeof(double) ? b : 64 * sizeof(double);
}
__device__ __forceinline__ void hc_merge_utt(const HarvestParams &p, int cap, int u, char *lds) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id();
  const int ns = p.sec_n[u * 2];
  if (ns > cap) {
    if (wave_in_block() == 0) hc_merge_in_hbm(p, u);
    return;
  }
  const int nf = p.nfb[u], nslot = p.nc[u] * 7;
  const int *sec = p.sec + (size_t)u * 6 * p.sec_cap;
  const double *ext = p.ext + (size_t)u * p.ext_cap;
  double *out = hc_row(p.c3, p, u);
  const double *cands = p.cand_a + (size_t)u * p.fb_stride * p.maxc;
  const double *scores = p.score_a + (size_t)u * p.fb_stride * p.maxc;
  // LDS: sums[cap + 1] | order, b_st, b_ed, s_off, s_lo [cap + 1 each] | fill_lo, fill_hi, fill_ch [cap + 1 each] | misc
  const int row = cap + 1;
  LDS_PTR(double) sums = (LDS_PTR(double))reinterpret_cast<double *>(lds);
  LDS_PTR(int) order = (LDS_PTR(int))reinterpret_cast<int *>(reinterpret_cast<double *>(lds) + row);
  LDS_PTR(int) b_st = order + row;
  LDS_PTR(int) b_ed = b_st + row;
  LDS_PTR(int) s_off = b_ed + row;
  LDS_PTR(int) s_lo = s_off + row;
  LDS_PTR(int) fill_lo = s_lo + row;
  LDS_PTR(int) fill_hi = fill_lo + row;
  LDS_PTR(int) fill_ch = fill_hi + row;
  LDS_PTR(int) misc = fill_ch + row;
  for (int k = tid; k < ns; k += nt) {
    sums[k] = p.sec_sum[(size_t)u * p.sec_cap + k];
    order[k] = 0; b_st[k] = sec[2 * p.sec_cap + k]; b_ed[k] = sec[3 * p.sec_cap + k];
    s_off[k] = sec[4 * p.sec_cap + k]; s_lo[k] = sec[5 * p.sec_cap + k];
  }
  __syncthreads();
  // value of channel `ch` at frame f: its slice inside its extended run, zero elsewhere (a plain load at a clamped
  // address and a select: nothing waits in a branch)
  auto chan = [&](int ch, int f) {
    const int c_st = b_st[ch], c_ed = b_ed[ch];
    const double v = ext[s_off[ch] + (imax(c_st, imin(c_ed, f)) - s_lo[ch])];
    return (f >= c_st && f <= c_ed) ? v : 0.0;
  };
  // the channel whose copy covers frame f last
  auto covering = [&](int f, int nfill) {
    int ch = fill_ch[0];
    for (int k = nfill - 1; k > 0; --k)
      if (f >= fill_lo[k] && f <= fill_hi[k]) { ch = fill_ch[k]; break; }
    return ch;
  };
  if (wave_in_block() == 0) {
    // ExtendSub (:840-856): stable compaction of the sections longer than 2200/mean_f0;
    // mean_f0 is deliberately NOT reset between sections.
    int kept = 0;
    if (lane == 0) {
      double mean = 0.0;
      for (int s = 0; s < ns; ++s) {
        int st = b_st[s], ed = b_ed[s];
        mean += sums[s];
        mean /= ed - st;
        if (2200.0 / mean < ed - st) {
          int t;
          t = b_st[kept]; b_st[kept] = b_st[s]; b_st[s] = t;
          t = b_ed[kept]; b_ed[kept] = b_ed[s]; b_ed[s] = t;
          t = s_off[kept]; s_off[kept] = s_off[s]; s_off[s] = t;
          t = s_lo[kept]; s_lo[kept] = s_lo[s]; s_lo[s] = t;
          kept++;
        }
      }
      p.sec_n[u * 2 + 1] = kept;
      // MakeSortedOrder (:883-896), quirks included
      for (int i = 0; i < kept; ++i) order[i] = i;
      for (int i = 1; i < kept; ++i)
        for (int j = i - 1; j >= 0; --j) {
          if (b_st[order[j]] > b_st[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
          else break;
        }
    }
    wave_sync();
    kept = wave_bcast_int(kept, 0);
    int nfill = 0;
    if (kept > 0) {
      // MergeF0 (:937-963)
      if (lane == 0) { fill_lo[0] = 0; fill_hi[0] = nf - 1; fill_ch[0] = 0; }
      nfill = 1;
      wave_sync();
      int cur_st = b_st[0], cur_ed = b_ed[0];                 // the reference's boundary_list[0], [1]
      for (int i = 1; i < kept; ++i) {
        const int o = order[i];
        // the reference reads boundary_list[o*2(+1)] AFTER possibly having overwritten entry 0
        const int st2 = o == 0 ? cur_st : b_st[o];
        const int ed2 = o == 0 ? cur_ed : b_ed[o];
        int lo_f = st2;
        if (st2 - cur_ed > 0) {
          cur_st = st2; cur_ed = ed2;
        } else {
          // MergeF0Sub (:912-932)
          const int st1 = cur_st, ed1 = cur_ed;
          if (st1 <= st2 && ed1 >= ed2) { cur_ed = ed1; continue; }
          double s1 = 0.0, s2 = 0.0;
          for (int base = st2; base <= ed1; base += WAVE) {
            const int f = imin(base + lane, ed1);              // (the lanes past ed1 repeat it; their scores are not summed)
            const double v1 = chan(covering(f, nfill), f), v2 = chan(o, f);
            // SearchScore (:901-907) of both values over the frame's slots, eight slots in flight
            double r1 = 0.0, r2 = 0.0;
            const double *c = cands + (size_t)f * p.maxc, *sc = scores + (size_t)f * p.maxc;
            constexpr int kB = 8;
            for (int i0 = 0; i0 < nslot; i0 += kB) {
              double cv[kB], sv[kB];
#pragma unroll
              for (int q = 0; q < kB; ++q) { const int k = imin(i0 + q, nslot - 1); cv[q] = c[k]; sv[q] = sc[k]; }
#pragma unroll
              for (int q = 0; q < kB; ++q) {
                if (i0 + q < nslot && v1 == cv[q] && r1 < sv[q]) r1 = sv[q];
                if (i0 + q < nslot && v2 == cv[q] && r2 < sv[q]) r2 = sv[q];
              }
            }
            const int n = imin(WAVE, ed1 - base + 1);
            for (int l = 0; l < n; ++l) { s1 += wave_pick(r1, l); s2 += wave_pick(r2, l); }   // in frame order
          }
          if (s1 > s2) lo_f = ed1;
          cur_ed = ed2;
        }
        if (lane == 0) { fill_lo[nfill] = lo_f; fill_hi[nfill] = ed2; fill_ch[nfill] = o; }
        ++nfill;
        wave_sync();
      }
    }
    if (lane == 0) { misc[0] = kept; misc[1] = nfill; }
  }
  __syncthreads();
  const int kept = misc[0], nfill = misc[1];
  // the contour: c3[f] = the value of the last copy covering f (step 2's contour when no section survived).  The
  // copies are walked in order for a batch of frames at a time -- the records are the same for every thread, so
  // their LDS reads are broadcasts that do not wait on each other (a backward search per frame is a chain of them).
  constexpr int kB = 10;
  for (int f0 = tid; f0 < nf; f0 += kB * nt) {
    double v[kB];
    if (kept == 0) {
      const double *step2 = hc_row(p.c2, p, u);
#pragma unroll
      for (int q = 0; q < kB; ++q) v[q] = step2[imin(nf - 1, f0 + q * nt)];
    } else {
      int ch[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) ch[q] = fill_ch[0];
      for (int k = 1; k < nfill; ++k) {
        const int lo = fill_lo[k], hi = fill_hi[k], c = fill_ch[k];
#pragma unroll
        for (int q = 0; q < kB; ++q) { const int f = f0 + q * nt; ch[q] = (f >= lo && f <= hi) ? c : ch[q]; }
      }
#pragma unroll
      for (int q = 0; q < kB; ++q) v[q] = chan(ch[q], imin(nf - 1, f0 + q * nt));
    }
#pragma unroll
    for (int q = 0; q < kB; ++q) if (f0 + q * nt < nf) out[f0 + q * nt] = v[q];
  }
}
__global__ void __launch_bounds__(kMergeThreadsLone) hc_merge(HarvestParams p, int cap) {
  DYN_LDS(lds);
  const int u = blockIdx.x;
  hc_merge_utt(p, cap, u, lds);
  __syncthreads();            // step 3's contour (c3) is complete and visible to the whole workgroup; the LDS is free
  hc_sections_step4_pass(p, u, reinterpret_cast<double *>(lds));
}

// ---- SmoothF0Contour (:1049-1113): zero-phase 2nd-order Butterworth per section ------
// The reference pads 300 zeros, holds the section's end values over the whole padded
// signal and runs the IIR forwards and backwards from rest.  300 samples is ~40 time
// constants of this filter (pole radius 0.875), so each sweep has converged to the
// DC steady state of the held value when it reaches the section: starting AT the
// section edge from that steady state differs by < 1e-17.
// One wavefront per voiced section; every lane filters one chunk of the section,
// warmed up over the kSmoothTail samples before it (same convergence argument).
//
// The sweeps are bound by instruction issue, not by the recurrence: a dependent FP64 FMA follows its producer after
// ~6 cycles and a wave64 FP64 instruction occupies the SIMD for 4 (tools/probe/fp64_chain.hip), so a step costs what
// stands between two of them.  Hence
//   * each sweep runs the recursion alone (w[n] = v[n] + a0 w[n-1] + a1 w[n-2], the a1 term folded into the input a
//     step ahead: two FMAs per step), parks w[n], and applies the three taps (y = b0 w[n] + b1 w[n-1] + b0 w[n-2]) in
//     a second loop -- the operations and their order are those of a fused sweep, so the output is bit-identical;
//   * every lane runs the same trip counts (no bounds test per step);
//   * the staged section carries its held end values with it, so a sample is one LDS load with no index clamping:
//       buf[0 .. 300) = first, buf[300 .. 300 + len) = section, then `last` up to 300 + len + 600 + kSmoothSlack;
//     sample j of the held signal is buf[j + 300].
// The section is filtered IN PLACE in that buffer -- input, forward state, forward output, backward state.  That is
// safe because the lanes of a wavefront run in lockstep: at step s of a sweep every lane stands s samples into its own
// stretch, so a lane above (forward; below, backward) has read any sample of this lane's chunk at an earlier step
// than the one this lane overwrites it at, and within a batch all loads are issued before the first store.
// A section that does not fit the LDS the launch reserved is filtered out of HBM with clamped indices.
constexpr int kSmoothSlack = 16;        // loads run up to a batch past the last step of a sweep
constexpr int kSmoothLdsMax = 6144;     // doubles: 48 KB, i.e. voiced sections up to 5 228 base frames (5.2 s in one piece) are
                                        // filtered in LDS, longer ones out of HBM.  (Round 3-4: 152 KB -- but a one-wavefront
                                        // workgroup that asks for most of a CU's LDS waits, in the in-flight mode, until the
                                        // frame workgroups of other jobs have drained from a whole CU: 67 us against 18 alone.)
constexpr int kSmoothBlocks = 64;       // wavefronts launched per utterance; they stride over its sections
__global__ void hc_smooth(HarvestParams p, int lds_doubles) {
  DYN_LDS(lds);
  const int u = blockIdx.y;
  const int ns = p.sec_n[u * 2];
  const double b0 = 0.0078202080334971724, b1 = 0.015640416066994345;
  const double a0 = 1.7347257688092754, a1 = -0.76600660094326412;
  const double dc = 1.0 / (1.0 - a0 - a1);             // state of the recursion at rest on a constant 1
  const int *sec = p.sec + (size_t)u * 6 * p.sec_cap;
  const double *in = hc_row(p.c0, p, u);
  double *out = hc_row(p.basic_f0, p, u);
  constexpr int kBatch = 10;             // inputs fetched together ahead of the recurrence; divides kSmoothTail
  static_assert(kSmoothTail % kBatch == 0, "the warm-up runs in whole batches");
  static_assert(kSmoothSlack > kBatch, "look-ahead loads stay inside the buffer");
  // Loads run past a lane's chunk (surplus steps of the whole batches, look-ahead) by at most kBatch + kSmoothSlack
  // samples at the top and, walking down, by no more than a chunk below sample 0: inside the held ends either way.
  // One step: `t` arrives holding a1 w[n-2] + v[n]; leaves holding a1 w[n-1] + v[n+1].
  auto step = [&](double &w0, double &w1, double &t, double vnext) __attribute__((always_inline)) {
    const double wt = fma(a0, w0, t);
    t = fma(a1, w0, vnext);
    w1 = w0; w0 = wt;
  };
  for (int k = wave_item_x(); k < ns; k += (int)gridDim.x * waves_per_block()) {
    const int st = sec[k], ed = sec[p.sec_cap + k];
    double *tmp = p.ext + (size_t)u * p.ext_cap + sec[4 * p.sec_cap + k];   // ed-st+1+kSmoothTail
    const int len = ed - st + 1, total = len + kSmoothTail;
    // an odd chunk: at any step the lanes stand `chunk` doubles apart, and an odd stride spreads them over all LDS banks
    const int chunk = ((total + WAVE - 1) / WAVE) | 1;
    const int steps = (chunk + kBatch - 1) / kBatch * kBatch;          // the same trip count in every lane
    const int j0 = lane_id() * chunk, j1 = imin(total, j0 + chunk);    // (the topmost lanes may own nothing: j1 <= j0)
    const double first = in[st], last = in[ed];
    // both sweeps: xin(j) = held input, tin(j) = forward output with `last` beyond it, ts[0 .. total) = where the forward
    // state and then output go, ws[0 .. len) = where the backward state goes and wload(j) reads it back
    auto sweeps = [&](auto xin, auto tin, auto ts, auto ws, auto wload) __attribute__((always_inline)) {
      {
        int j = j0 - kSmoothTail;
        double w0 = xin(j) * dc, w1 = w0;
        double t = fma(a1, w1, xin(j));
        for (int s = 0; s < kSmoothTail; s += kBatch, j += kBatch) {        // warm-up: nothing is kept
          double v[kBatch];
#pragma unroll
          for (int q = 0; q < kBatch; ++q) v[q] = xin(j + q + 1);
#pragma unroll
          for (int q = 0; q < kBatch; ++q) step(w0, w1, t, v[q]);
        }
        double p1 = w0, p2 = w1;                                            // w[j0-1], w[j0-2]
        for (int s = 0; s < steps; s += kBatch, j += kBatch) {              // the lane's chunk: w[j] parked in ts[j]
          double v[kBatch];
#pragma unroll
          for (int q = 0; q < kBatch; ++q) v[q] = xin(j + q + 1);
#pragma unroll
          for (int q = 0; q < kBatch; ++q) {
            step(w0, w1, t, v[q]);
            if (j + q < j1) ts[j + q] = w0;
          }
        }
        for (int s = 0, i = j0; s < steps; s += kBatch, i += kBatch) {     // three taps, in place, a batch of loads at a time
          double wv[kBatch];
#pragma unroll
          for (int q = 0; q < kBatch; ++q) wv[q] = tin(i + q);
#pragma unroll
          for (int q = 0; q < kBatch; ++q) {
            const double y = b0 * wv[q] + b1 * p1 + b0 * p2;
            p2 = p1; p1 = wv[q];
            if (i + q < j1) ts[i + q] = y;
          }
        }
      }
      wave_sync();
      // backward sweep over the forward output (which has settled on `last` beyond the tail)
      {
        int j = j1 - 1 + kSmoothTail;
        double w0 = tin(j) * dc, w1 = w0;
        double t = fma(a1, w1, tin(j));
        for (int s = 0; s < kSmoothTail; s += kBatch, j -= kBatch) {
          double v[kBatch];
#pragma unroll
          for (int q = 0; q < kBatch; ++q) v[q] = tin(j - q - 1);
#pragma unroll
          for (int q = 0; q < kBatch; ++q) step(w0, w1, t, v[q]);
        }
        // frames of the chunk at and beyond len are not output, but the recursion runs through them: the taps of the
        // topmost output frame read the state it left
        const int top = imin(j1, len);
        double p1 = 0, p2 = 0;
        for (int s = 0; s < steps; s += kBatch, j -= kBatch) {              // steps below j0 are surplus and not kept
          double v[kBatch];
#pragma unroll
          for (int q = 0; q < kBatch; ++q) v[q] = tin(j - q - 1);
#pragma unroll
          for (int q = 0; q < kBatch; ++q) {
            if (j - q == top - 1) { p1 = w0; p2 = w1; }                     // w[top], w[top+1]
            step(w0, w1, t, v[q]);
            if (j - q >= j0 && j - q < top) ws[j - q] = w0;
          }
        }
        for (int s = 0, i = top - 1; s < steps; s += kBatch, i -= kBatch) {
          double wv[kBatch];
#pragma unroll
          for (int q = 0; q < kBatch; ++q) wv[q] = wload(i - q);
#pragma unroll
          for (int q = 0; q < kBatch; ++q) {
            const double y = b0 * wv[q] + b1 * p1 + b0 * p2;
            p2 = p1; p1 = wv[q];
            if (i - q >= j0) out[st + i - q] = y;
          }
        }
      }
    };
    const int need = total + 2 * kSmoothTail + kSmoothSlack;
    if (need <= lds_doubles) {
      LDS_PTR(double) buf = (LDS_PTR(double))(reinterpret_cast<double *>(lds) + (size_t)wave_in_block() * lds_doubles);
      LDS_PTR(double) held = buf + kSmoothTail;                       // held[j] = sample j
      for (int i0 = lane_id(); i0 < len; i0 += kBatch * WAVE) {     // kBatch loads in flight per lane
        double v[kBatch];
#pragma unroll
        for (int q = 0; q < kBatch; ++q) v[q] = i0 + q * WAVE < len ? in[st + i0 + q * WAVE] : 0.0;
#pragma unroll
        for (int q = 0; q < kBatch; ++q) if (i0 + q * WAVE < len) held[i0 + q * WAVE] = v[q];
      }
      for (int i = lane_id(); i < kSmoothTail; i += WAVE) buf[i] = first;
      for (int i = len + lane_id(); i < need - kSmoothTail; i += WAVE) held[i] = last;
      wave_sync();
#ifdef WORLD_EMU
      // (one lane owns the whole section there: its surplus steps -- never kept -- reach a batch below the held head)
      auto at = [&](int j) { return held[imax(j, -kSmoothTail)]; };
#else
      auto at = [&](int j) { return held[j]; };       // -kSmoothTail <= j < total + kSmoothTail + kSmoothSlack in both sweeps
#endif
      sweeps(at, at, held, held, at);
      wave_sync();                                                   // the next section reuses the buffer
    } else {
      auto xin = [&](int j) { const double v = in[st + imin(len - 1, imax(0, j))]; return j < 0 ? first : (j >= len ? last : v); };
      auto tin = [&](int j) { const double v = tmp[imin(total - 1, imax(0, j))]; return j >= total ? last : v; };
      auto wload = [&](int j) { return out[st + imin(len - 1, imax(0, j))]; };
      sweeps(xin, tin, tmp, out + st, wload);
    }
  }
}

// ---- Harvest() hop subsampling (:1246-1251) -------------------------------------------
__global__ void hc_output(HarvestParams p) {
  const int i = flat_thread_x(), u = blockIdx.y;
  if (i >= p.b.n_frames[u]) return;
  const double t = i * p.frame_period / 1000.0;
  const int nfb = p.nfb[u];
  p.tpos[(size_t)u * p.b.f_stride + i] = t;
  const int src = p.frame_period == 1.0 ? i : imin(nfb - 1, mround(t * 1000.0));
  p.f0[(size_t)u * p.b.f_stride + i] = hc_row(p.basic_f0, p, u)[src];
}

void launch_harvest_contour(const HarvestParams &p, int max_fb, int max_frames, hipStream_t stream) {
  const int B = p.b.n_utt;
  SecArgs a2 = {p.c2, 1, 2 * kExtMargin, nullptr, nullptr};
  // a job that has the device to itself: sixteen wavefronts per utterance for the two one-workgroup kernels (VERDICT r05
  // item 7: the shapes sized for the in-flight mode had cost a lone job 0.04 ms)
  WH_BLOCKS(hc_step12_sections, dim3(B), p.lone_job ? 1024 : 256, (64 + kStepLds) * sizeof(double), stream, p, a2);
  WH_BLOCKS(hc_extend, dim3(imin(p.sec_cap, kExtendBlocks), B), kExtendThreads, 64 * sizeof(double), stream, p);
  // WORLD_HIP_MERGE_LDS_SECTIONS lowers the number of section records hc_merge keeps in LDS (tests use it to send an
  // ordinary utterance down the route of one with thousands of sections)
  static const int merge_limit = [] { const char *e = getenv("WORLD_HIP_MERGE_LDS_SECTIONS"); return e ? imax(0, imin(kMergeLdsSections, atoi(e))) : kMergeLdsSections; }();
  const int merge_cap = imin(p.sec_cap, merge_limit);
  // (+ FixStep4 and the two section passes around it; then hc_smooth writes basic_f0's voiced frames)
  WH_BLOCKS(hc_merge, dim3(B), p.lone_job ? kMergeThreadsLone : kMergeThreads, hc_merge_lds_bytes(merge_cap), stream, p, merge_cap);
  // one wavefront per block: each reserves LDS for the longest section the batch can hold
  const int smooth_lds = imin(kSmoothLdsMax, max_fb + 3 * kSmoothTail + kSmoothSlack);
  WH_BLOCKS(hc_smooth, dim3(imin(p.sec_cap, kSmoothBlocks), B), WAVE, smooth_lds * sizeof(double), stream, p, smooth_lds);
  WH_THREADS(hc_output, max_frames, B, 1, stream, p);
}

}  // namespace world_hip
