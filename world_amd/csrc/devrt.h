// devrt.h -- the one place that knows how kernels are launched.
//
// Product build (hipcc, gfx950): real HIP launches, 64-lane wavefront
// collectives (__shfl_xor / ballot), LDS carved from one dynamic region.
//
// Test-only build (g++ -DWORLD_EMU, see tests/emu/): the SAME kernel sources are
// compiled for the host with every block shrunk to ONE thread and every wave to
// ONE lane, blocks executed serially.  Kernels are written in a block-cooperative
// style (block-stride loops + the collectives below + __syncthreads), so the
// emulation runs the identical index arithmetic, RNG bookkeeping and control
// flow -- it lets `pytest -m "not gpu"` check kernel logic against the oracle in
// a container without a GPU.  It is never built into libworld_hip.so.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#ifdef WORLD_EMU
// ------------------------------------------------------------------ emulation
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline double2 make_double2(double a, double b) { double2 r; r.x = a; r.y = b; return r; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint4 r = {a, b, c, d}; return r; }
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern thread_local char *emu_lds_base;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
static inline void __syncthreads() {}
typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
#define WAVE 1
#define DYN_LDS(name) char *name = emu_lds_base
#define LDS_PTR(T) T *
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __brev(unsigned v) {
  unsigned r = 0;
  for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}
static inline double cospi(double x) { return cos(3.14159265358979323846 * x); }
static inline double sinpi(double x) { return sin(3.14159265358979323846 * x); }
static inline void sincospi(double x, double *s, double *c) {
  *s = sin(3.14159265358979323846 * x);
  *c = cos(3.14159265358979323846 * x);
}
static inline long long __double_as_longlong(double v) { long long r; memcpy(&r, &v, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
static inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p += v; return o; }
namespace devrt {
void *dmalloc(size_t bytes);
void dfree(void *p);
static inline void h2d(void *dst, const void *src, size_t n, hipStream_t) { memcpy(dst, src, n); }
static inline void d2h(void *dst, const void *src, size_t n, hipStream_t) { memcpy(dst, src, n); }
static inline void d2d(void *dst, const void *src, size_t n, hipStream_t) { memcpy(dst, src, n); }
static inline void dzero(void *dst, size_t n, hipStream_t) { memset(dst, 0, n); }
static inline void sync(hipStream_t) {}
static inline void set_device(int) {}
static inline int current_device() { return 0; }
static inline void *hmalloc_pinned(size_t n) { return malloc(n); }
static inline void hfree_pinned(void *p) { free(p); }
static inline void *event_create() { return nullptr; }
static inline void event_destroy(void *) {}
static inline void event_record(void *, hipStream_t) {}
static inline void event_sync(void *) {}
static inline void stream_wait_event(hipStream_t, void *) {}
static inline hipStream_t stream_create() { return nullptr; }
static inline void stream_destroy(hipStream_t) {}
static inline bool is_capturing(hipStream_t) { return false; }
[[noreturn]] static inline void graph_begin(hipStream_t) { throw std::runtime_error("HIP graphs do not exist in the host emulation"); }
static inline void *graph_end(hipStream_t) { return nullptr; }
static inline void graph_launch(void *, hipStream_t) {}
static inline void graph_destroy(void *) {}
static inline void peer_copy(void *dst, int, const void *src, int, size_t n, hipStream_t) { memcpy(dst, src, n); }
static inline void enable_peer_access(int, int) {}
void emu_run_begin(size_t lds_bytes);
void emu_run_end();
template <class K, class... A>
void launch_blocks(const char * /*name*/, K kernel, dim3 grid, int /*threads*/, size_t lds, hipStream_t, A... args) {
  emu_run_begin(lds);
  blockDim = dim3(1, 1, 1);
  gridDim = grid;
  threadIdx = dim3(0, 0, 0);
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        blockIdx = dim3(x, y, z);
        kernel(args...);
      }
  emu_run_end();
}
}  // namespace devrt
#elif defined(WORLD_SIMT)
// ------------------------------------------------------------------ wave-accurate host emulation of ONE unit's GPU path
// (TEST INFRASTRUCTURE, tests/emu/simt_host.h: the unit is compiled as the GPU compiles it -- WAVE = 64, real workgroup
// sizes, every #ifndef WORLD_EMU branch -- with fibres for threads and rendezvous for the cross-lane instructions)
#include "simt_host.h"
#else
// ------------------------------------------------------------------ gfx950
#include <hip/hip_runtime.h>
#include <type_traits>
#define WAVE 64
#define DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
// pointer into LDS with its address space in the type: accesses through it are ds_ instructions even where the
// same code is also instantiated for global pointers (a generic pointer would make them flat_ accesses)
#define LDS_PTR(T) __attribute__((address_space(3))) T *
namespace devrt {
void check(hipError_t e, const char *what);
void *dmalloc(size_t bytes);
void dfree(void *p);
void h2d(void *dst, const void *src, size_t n, hipStream_t s);
void d2h(void *dst, const void *src, size_t n, hipStream_t s);
void d2d(void *dst, const void *src, size_t n, hipStream_t s);
void dzero(void *dst, size_t n, hipStream_t s);
void sync(hipStream_t s);
void set_device(int device);
int current_device();
void *hmalloc_pinned(size_t n);
void hfree_pinned(void *p);
void *event_create();
void event_destroy(void *ev);
void event_record(void *ev, hipStream_t s);
void event_sync(void *ev);
void stream_wait_event(hipStream_t s, void *ev);
hipStream_t stream_create();                 // non-blocking stream on the current device
void stream_destroy(hipStream_t s);
// HIP graphs: everything a context enqueues between begin and end becomes one replayable graph
bool is_capturing(hipStream_t s);
void graph_begin(hipStream_t s);
void *graph_end(hipStream_t s);              // -> executable graph
void graph_launch(void *exec, hipStream_t s);
void graph_destroy(void *exec);
// copy between devices (xGMI when peer access is enabled, staged otherwise); same device = plain D2D
void peer_copy(void *dst, int dst_device, const void *src, int src_device, size_t n, hipStream_t s);
void enable_peer_access(int device, int peer);    // idempotent; a refusal is not an error (copies are then staged)
// optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
void prof_begin(const char *name, hipStream_t s);
void prof_end(hipStream_t s);
extern bool g_profiling;
void prof_enable(bool on);
// opt a kernel in to more than 48 KiB of dynamic LDS (up to the 160 KiB of a gfx950 CU); the
// attribute is set once per kernel and size, not per launch
void allow_large_lds(const void *kernel, size_t lds, const char *name);
extern bool g_host_trace;                      // WORLD_HIP_HOST_TRACE=1: host cost per call site, printed at exit
void host_trace_add(const char *name, double us);
}  // namespace devrt
#include <chrono>
#include <string>
namespace devrt {
std::string prof_collect();
template <class K, class... A>
void launch_blocks(const char *name, K kernel, dim3 grid, int threads, size_t lds, hipStream_t s, A... args) {
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
  if (lds > 48 * 1024) allow_large_lds(reinterpret_cast<const void *>(kernel), lds, name);
  if (g_profiling) prof_begin(name, s);
  if (g_host_trace) {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, args...);
    host_trace_add(name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  } else {
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, args...);
  }
  check(hipGetLastError(), name);
  if (g_profiling) prof_end(s);
}
}  // namespace devrt
#endif

namespace devrt {
// ---- launch shapes ---------------------------------------------------------
// "flat" kernels: one thread per item along x (items = blockIdx.x*blockDim.x+threadIdx.x)
template <class K, class... A>
void launch_threads(const char *name, K kernel, long items, unsigned ny, unsigned nz, hipStream_t s, A... args) {
#ifdef WORLD_EMU
  launch_blocks(name, kernel, dim3((unsigned)items, ny, nz), 1, 0, s, args...);
#else
  launch_blocks(name, kernel, dim3((unsigned)((items + 255) / 256), ny, nz), 256, 0, s, args...);
#endif
}
// "wave" kernels: one wavefront per item along x, 4 waves per block
template <class K, class... A>
void launch_waves(const char *name, K kernel, long items, unsigned ny, unsigned nz, size_t lds_per_wave, hipStream_t s, A... args) {
#ifdef WORLD_EMU
  launch_blocks(name, kernel, dim3((unsigned)items, ny, nz), 1, lds_per_wave, s, args...);
#else
  launch_blocks(name, kernel, dim3((unsigned)((items + 3) / 4), ny, nz), 256, 4 * lds_per_wave, s, args...);
#endif
}
}  // namespace devrt
// launch macros: the kernel's own name labels errors and profile records
#define WH_BLOCKS(kernel, ...) devrt::launch_blocks(#kernel, kernel, __VA_ARGS__)
#define WH_THREADS(kernel, ...) devrt::launch_threads(#kernel, kernel, __VA_ARGS__)
#define WH_WAVES(kernel, ...) devrt::launch_waves(#kernel, kernel, __VA_ARGS__)

// ---- in-kernel helpers -------------------------------------------------------
__device__ __forceinline__ int flat_thread_x() { return (int)(blockIdx.x * blockDim.x + threadIdx.x); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x % WAVE); }
__device__ __forceinline__ int wave_in_block() { return (int)(threadIdx.x / WAVE); }
__device__ __forceinline__ int waves_per_block() { return (int)((blockDim.x + WAVE - 1) / WAVE); }
// Workgroup size and thread index for code that may know the size at compile time (NT > 0: the launch uses exactly NT
// threads; loops over `tid + m NT` then unroll with their bounds checks decided at compile time, and the stride
// needs no register).  NT = 0: read from the launch.
template <int NT> __device__ __forceinline__ int wg_size() {
  if constexpr (NT > 0) return NT; else return (int)blockDim.x;
}
template <int NT> __device__ __forceinline__ int wg_thread() {
  int t = (int)threadIdx.x;
#ifndef WORLD_EMU
#ifdef WH_FRESH_TID
  // a unit's choice (harvest.hip): an opaque copy per call, so that what a stage derives from the thread index (LDS
  // addresses, butterfly numbers) is recomputed where it is used instead of living in registers across a kernel's outer loop
#ifndef WORLD_SIMT
  if constexpr (NT > 0) asm volatile("" : "+v"(t));
#endif
#endif
  if constexpr (NT > 0) __builtin_assume(t >= 0 && t < NT);
#endif
  return t;
}
template <int NT> __device__ __forceinline__ int wg_waves() { return (wg_size<NT>() + WAVE - 1) / WAVE; }
__device__ __forceinline__ int wave_item_x() { return (int)(blockIdx.x * waves_per_block() + wave_in_block()); }

// Frame kernels (one workgroup per analysis frame) read windows of x that overlap their neighbours' by
// 80-90 %, but consecutive workgroups go to different XCDs (round-robin, b % 8) and so to different L2s:
// every L2 ends up fetching all of x.  This bijection on [0, n) hands each XCD runs of kXcdRun consecutive
// frames instead (run j of a group of 8 runs goes to XCD j), which keeps the overlap inside one L2 while
// the runs stay short.  Only for kernels whose frames all cost the same (CheapTrick): workgroups are
// dispatched in order, so where frames can exit early (D4C and StoneMask skip unvoiced frames) a run of
// cheap frames on one XCD leaves it waiting behind the others' full queues -- measured +9 % on
// d4c_groupdelay, which therefore keeps the plain order.  The b % 8 placement is a speed heuristic only
// (MI355X_MICROARCH.md: the map is undefined); nothing depends on it.
constexpr int kXcdRun = 16;
__device__ __forceinline__ int xcd_grouped(int b, int n) {
  constexpr int kGroup = 8 * kXcdRun;
  if (b >= n / kGroup * kGroup) return b;          // the ragged tail keeps its order
  const int r = b % kGroup;
  return b - r + (r % 8) * kXcdRun + r / 8;
}

// ---- race-shaking builds (test infrastructure; python -m world_amd.build --checked; tests/test_gpu_parity.py) ----------
// The frame kernels alias LDS regions across phases and place their barriers by hand (d4c_frame: 63 of them), and a
// missing one is invisible in normal runs: the wavefronts of a workgroup execute the same instruction stream and arrive
// together (round 3-4 shipped a select that zeroed histograms aliasing a transform buffer other wavefronts were still
// reading -- found by reading the code, never by a test).  Two builds make such a bug change the RESULT:
//   -DWH_JITTER      every phase starts with a pseudo-random stall of some of the workgroup's wavefronts -- behind every
//                    __syncthreads(), behind every wavefront-level fence (wave_sync: the FFT's wave-local stages, the
//                    DPP-row prefix sum's hand-offs) and at jitter() marks placed where code relies on something weaker
//                    than a barrier -- so a wavefront that is still reading when it should have been waited for is
//                    really overtaken, by thousands of cycles.  No synchronisation is added: a correct kernel computes
//                    the same bits, a racy one does not.
//   -DWH_LDS_POISON  the whole LDS allocation is filled with a NaN pattern when a workgroup starts (a fresh workgroup
//                    otherwise inherits the previous one's data, usually the same kernel's: plausible values), and
//                    lds_dead(ptr, n) marks fill a region whose contents nobody may read any more.  (This build ADDS
//                    barriers around the fills, so it is a separate build from the jitter one.)
// The GPU suite runs the analysis paths on both and requires results bit-identical to the product build's.
#ifndef WORLD_EMU
#if defined(WH_JITTER)
__device__ __forceinline__ void jitter() {
  // per wavefront, per call: hash of the clock, the wavefront and the workgroup
  unsigned h = (unsigned)__builtin_readcyclecounter() * 2654435761u;
  h ^= ((unsigned)threadIdx.x >> 6) * 0x9E3779B9u + blockIdx.x * 0x85EBCA6Bu + blockIdx.y * 0xC2B2AE35u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
  if ((h & 3u) == 0u) {                                     // a quarter of the arrivals stall: 0.5 .. 8 thousand cycles
    const int n = 1 + (int)((h >> 4) & 15u);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
  }
}
__device__ __forceinline__ void wh_jitter_syncthreads() { __syncthreads(); jitter(); }
#define __syncthreads() wh_jitter_syncthreads()
#elif defined(WH_BARTRACE)
// Development aid (tools/barrier_skew.py): ONE workgroup -- block (WH_TRACE_FRAME, WH_TRACE_UTT) of a kernel that armed the
// tracer with wh_bartrace_arm() -- stamps the shader clock of every wavefront's ARRIVAL at every workgroup barrier:
// wh_bar[barrier ordinal * 16 + wavefront].  Which wavefront the others wait for, and for how long, is what neither
// rocprofv3's kernel totals nor the phase stamps of trace.h show.  The ordinal is counted per wavefront in LDS, behind the
// barrier (nothing is loaded in front of it); every other workgroup pays one scalar compare per barrier.
#ifndef WH_TRACE_FRAME
#define WH_TRACE_FRAME 1000
#endif
#ifndef WH_TRACE_UTT
#define WH_TRACE_UTT 0
#endif
namespace world_hip_bt { static __device__ long long wh_bar[128 * 16]; }
struct WhBarState { int magic; int cnt[16]; };
__device__ __forceinline__ WhBarState &wh_bar_state() { static __shared__ WhBarState s; return s; }
__device__ __forceinline__ void wh_bartrace_arm(bool on) {
  WhBarState &s = wh_bar_state();
  if ((threadIdx.x & 63) == 0) { s.cnt[threadIdx.x >> 6] = 0; s.magic = on ? 0x5EED : 0; }
}
__device__ __forceinline__ void wh_bartrace_syncthreads() {
  const bool me = blockIdx.x == WH_TRACE_FRAME && blockIdx.y == WH_TRACE_UTT;
  const long long t = me ? (long long)__builtin_readcyclecounter() : 0;
  __syncthreads();
  if (me && (threadIdx.x & 63) == 0) {
    WhBarState &s = wh_bar_state();
    if (s.magic == 0x5EED) {
      const int w = threadIdx.x >> 6, i = s.cnt[w];
      s.cnt[w] = i + 1;
      if (i < 128) world_hip_bt::wh_bar[i * 16 + w] = t;
    }
  }
}
#define WH_BARTRACE_DEFINE(unit)                                                                        \
  extern "C" __attribute__((visibility("default"))) int world_hip_bartrace_read_##unit(long long *out, int n) { \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(world_hip_bt::wh_bar), sizeof(long long) * n);      \
  }
#define __syncthreads() wh_bartrace_syncthreads()
__device__ __forceinline__ void jitter() {}
#else
__device__ __forceinline__ void jitter() {}
#endif
#if !defined(WH_BARTRACE)
#define WH_BARTRACE_DEFINE(unit)
__device__ __forceinline__ void wh_bartrace_arm(bool) {}
#endif
#if defined(WH_LDS_POISON)
__device__ __forceinline__ void lds_poison_all(char *lds) {
  // the dispatch packet's group_segment_size (hsa_kernel_dispatch_packet_t, byte 28)
  const unsigned bytes = ((const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_dispatch_ptr())[7];
  unsigned long long *q = reinterpret_cast<unsigned long long *>(lds);
  for (unsigned i = threadIdx.x; i < bytes / 8; i += blockDim.x) q[i] = 0x7FF8DEADBEEF0000ull + i;
  __syncthreads();
}
__device__ __forceinline__ void lds_dead(void *p, int doubles) {
  __syncthreads();
  unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
  for (int i = threadIdx.x; i < doubles; i += blockDim.x) q[i] = 0x7FF8DEAD00000000ull + (unsigned)i;
  __syncthreads();
}
#undef DYN_LDS
#define DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]; lds_poison_all(name)
#else
__device__ __forceinline__ void lds_dead(void *, int) {}
#endif
#else
static inline void jitter() {}
static inline void lds_dead(void *, int) {}
static inline void wh_bartrace_arm(bool) {}
#define WH_BARTRACE_DEFINE(unit)
#endif

// make one wave's LDS writes visible to its other lanes (no-op for a 1-lane wave)
__device__ __forceinline__ void wave_sync() {
#ifndef WORLD_EMU
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  jitter();
#endif
}

// ---- cross-lane moves without LDS ---------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 (two per double), an LDS-pipe round trip per level: ~700 cycles for a
// 64-lane FP64 reduction, and the refinement / tracking kernels do hundreds of them per frame.  Within a row
// of 16 lanes the same exchanges are DPP modifiers of a v_mov (quad permutes, mirrors, rotations: a few cycles);
// the four row results are then read with v_readlane.  For a commutative reduction the mirrors pair the same
// partial results as xor 4 / xor 8 would (the operands are uniform within the groups already reduced).
#ifndef WORLD_EMU
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;           // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int kDppHalfMirror = 0x141, kDppMirror = 0x140;  // lane i <-> 7 - i within 8, i <-> 15 - i within 16
constexpr int kDppRor8 = 0x128;                            // row_ror:8 == xor 8 within a row
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// sum over aligned groups of 8 / 16 lanes, every lane of the group gets it (bitwise what the xor butterfly gives)
__device__ __forceinline__ double oct_sum(double v) {
  v += dpp_f64<kDppXor1>(v); v += dpp_f64<kDppXor2>(v); v += dpp_f64<kDppHalfMirror>(v);
  return v;
}
__device__ __forceinline__ double row_sum(double v) { v = oct_sum(v); v += dpp_f64<kDppMirror>(v); return v; }
#endif

// wave collectives (every lane gets the result)
__device__ __forceinline__ double wave_sum(double v) {
#ifndef WORLD_EMU
  v = row_sum(v);
  v = (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
#endif
  return v;
}
__device__ __forceinline__ int wave_sum_int(int v) {
#ifndef WORLD_EMU
  v += dpp_i32<kDppXor1>(v); v += dpp_i32<kDppXor2>(v); v += dpp_i32<kDppHalfMirror>(v); v += dpp_i32<kDppMirror>(v);
  v = (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
      (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
#endif
  return v;
}
__device__ __forceinline__ int wave_max_int(int v) {
#ifndef WORLD_EMU
  int o;
  o = dpp_i32<kDppXor1>(v); v = o > v ? o : v;
  o = dpp_i32<kDppXor2>(v); v = o > v ? o : v;
  o = dpp_i32<kDppHalfMirror>(v); v = o > v ? o : v;
  o = dpp_i32<kDppMirror>(v); v = o > v ? o : v;
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
            d = __builtin_amdgcn_readlane(v, 48);
  const int ab = a > b ? a : b, cd = c > d ? c : d;
  v = ab > cd ? ab : cd;
#endif
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#ifndef WORLD_EMU
  double o;
  o = dpp_f64<kDppXor1>(v); v = o > v ? o : v;
  o = dpp_f64<kDppXor2>(v); v = o > v ? o : v;
  o = dpp_f64<kDppHalfMirror>(v); v = o > v ? o : v;
  o = dpp_f64<kDppMirror>(v); v = o > v ? o : v;
  const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
  const double ab = a > b ? a : b, cd = c > d ? c : d;
  v = ab > cd ? ab : cd;
#endif
  return v;
}
// min and max of one unsigned 64-bit key per lane (the radix select's key range)
__device__ __forceinline__ void wave_minmax_u64(unsigned long long &lo, unsigned long long &hi) {
#ifndef WORLD_EMU
  auto mov = [](unsigned long long v, auto ctrl) {
    const int l = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, decltype(ctrl)::value, 0xf, 0xf, true);
    const int h = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), decltype(ctrl)::value, 0xf, 0xf, true);
    return ((unsigned long long)(unsigned)h << 32) | (unsigned)l;
  };
  auto step = [&](auto ctrl) {
    const unsigned long long a = mov(lo, ctrl), b = mov(hi, ctrl);
    lo = a < lo ? a : lo; hi = b > hi ? b : hi;
  };
  step(std::integral_constant<int, kDppXor1>{}); step(std::integral_constant<int, kDppXor2>{});
  step(std::integral_constant<int, kDppHalfMirror>{}); step(std::integral_constant<int, kDppMirror>{});
  auto rl = [](unsigned long long v, int lane) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  };
  unsigned long long l = rl(lo, 0), h = rl(hi, 0);
#pragma unroll
  for (int r = 16; r < 64; r += 16) {
    const unsigned long long a = rl(lo, r), b = rl(hi, r);
    l = a < l ? a : l; h = b > h ? b : h;
  }
  lo = l; hi = h;
#endif
}
__device__ __forceinline__ double wave_min(double v) {
#ifndef WORLD_EMU
  double o;
  o = dpp_f64<kDppXor1>(v); v = o < v ? o : v;
  o = dpp_f64<kDppXor2>(v); v = o < v ? o : v;
  o = dpp_f64<kDppHalfMirror>(v); v = o < v ? o : v;
  o = dpp_f64<kDppMirror>(v); v = o < v ? o : v;
  const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
  const double ab = a < b ? a : b, cd = c < d ? c : d;
  v = ab < cd ? ab : cd;
#endif
  return v;
}
// The same for non-negative values (no NaNs) on the shortest path this GPU offers: v_min_f64 itself (no compare and
// select), four DPP steps inside the rows of 16 lanes, row_bcast:15 and row_bcast:31 to fold the rows into lane 63, one
// v_readlane pair -- 6 x 3 + 2 instructions where wave_min() spends ~45 and three trips between VALU and SALU.
__device__ __forceinline__ double wave_min_nonneg(double v) {
#ifndef WORLD_EMU
  auto vmin = [](double a, double b) {
#ifdef WORLD_SIMT
    return a < b ? a : b;                                  // (the instruction's result for the non-negative, NaN-free operands it is given)
#else
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
  };
  auto bcast = [](double x, auto ctrl_c, auto rows_c) {      // rows named by the mask take the last lane of the row(s) before
    constexpr int CTRL = decltype(ctrl_c)::value, ROWS = decltype(rows_c)::value;
    const int hi = __double2hiint(x), lo = __double2loint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWS, 0xf, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWS, 0xf, false));
  };
  v = vmin(v, dpp_f64<kDppXor1>(v));
  v = vmin(v, dpp_f64<kDppXor2>(v));
  v = vmin(v, dpp_f64<kDppHalfMirror>(v));
  v = vmin(v, dpp_f64<kDppMirror>(v));
  v = vmin(v, bcast(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}));   // row_bcast:15
  v = vmin(v, bcast(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}));   // row_bcast:31
  return readlane_f64(v, 63);
#else
  return v;
#endif
}
// Inclusive add-scan over the wavefront by DPP (no LDS crossbar): Kogge-Stone inside each row of 16 lanes (row_shr
// 1, 2, 4, 8; lanes shifted in from outside the row read 0), then row_bcast:15 hands every odd row the total of the
// row before it and row_bcast:31 hands rows 2 and 3 the total of rows 0-1.  Six dependent VALU steps where the
// ds_bpermute version paid six LDS round trips.
__device__ __forceinline__ int wave_incl_scan_int(int v) {
#ifndef WORLD_EMU
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
#endif
  return v;
}
// the same for doubles (+0.0 shifted in where a lane has no source)
#ifndef WORLD_EMU
template <int CTRL, int ROWS, bool ZERO> __device__ __forceinline__ double dpp_f64_rows(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, ZERO),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, ZERO));
}
#endif
__device__ __forceinline__ double wave_incl_scan_f64(double v) {
#ifndef WORLD_EMU
  v += dpp_f64_rows<0x111, 0xf, true>(v);
  v += dpp_f64_rows<0x112, 0xf, true>(v);
  v += dpp_f64_rows<0x114, 0xf, true>(v);
  v += dpp_f64_rows<0x118, 0xf, true>(v);
  v += dpp_f64_rows<0x142, 0xa, false>(v);
  v += dpp_f64_rows<0x143, 0xc, false>(v);
#endif
  return v;
}
__device__ __forceinline__ int wave_excl_scan_int(int v, int *total) {
#ifndef WORLD_EMU
  const int inc = wave_incl_scan_int(v);
  *total = __builtin_amdgcn_readlane(inc, 63);
  return inc - v;
#else
  *total = v;
  return 0;
#endif
}
// v + (the value of lane ^ 16) and v + (the value of lane ^ 32) without the LDS crossbar: CDNA4's swaps of 16-lane rows
// and of wave halves between two registers.  Given two copies of x, v_permlane16_swap leaves (rows 0 0 2 2) and
// (rows 1 1 3 3), v_permlane32_swap (lower lower) and (upper upper); their sum is x + x(partner) in every lane -- the
// operands of __shfl_xor's sum, in one order or the other.
#ifndef WORLD_EMU
__device__ __forceinline__ double row_pair_sum(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double half_pair_sum(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
#endif
// lane `src_lane`'s value when the lane index is the same in every lane (a loop counter): v_readlane, no LDS crossbar
__device__ __forceinline__ double wave_pick(double v, int src_lane) {
#ifndef WORLD_EMU
  return readlane_f64(v, __builtin_amdgcn_readfirstlane(src_lane));
#else
  (void)src_lane;
  return v;
#endif
}
__device__ __forceinline__ double wave_bcast(double v, int src_lane) {
#ifndef WORLD_EMU
  return __shfl(v, src_lane, 64);
#else
  (void)src_lane;
  return v;
#endif
}
__device__ __forceinline__ int wave_bcast_int(int v, int src_lane) {
#ifndef WORLD_EMU
  return __shfl(v, src_lane, 64);
#else
  (void)src_lane;
  return v;
#endif
}

// block collectives; `scratch` = LDS area of >= 64 doubles owned by the caller.
// All threads must call; result returned to all; safe to call back to back.
// PRE = false: the caller guarantees that nobody is still reading this scratch area (e.g. the previous collective used
// another area and had its own barrier): one barrier instead of two.
template <int NT = 0, bool PRE = true> __device__ __forceinline__ double block_sum(double v, double *scratch) {
#ifndef WORLD_EMU
  v = wave_sum(v);
  int nw = wg_waves<NT>();
  if (nw == 1) return v;
  if (PRE) __syncthreads();  // previous users of scratch are done
  if (lane_id() == 0) scratch[wave_in_block()] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t += scratch[w];
  return t;
#else
  (void)scratch;
  return v;
#endif
}
template <int NT = 0, bool PRE = true> __device__ __forceinline__ void block_sum2(double &a, double &b, double *scratch) {
#ifndef WORLD_EMU
  a = wave_sum(a); b = wave_sum(b);
  int nw = wg_waves<NT>();
  if (nw == 1) return;
  if (PRE) __syncthreads();
  if (lane_id() == 0) { scratch[wave_in_block()] = a; scratch[32 + wave_in_block()] = b; }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
  for (int w = 0; w < nw; ++w) { ta += scratch[w]; tb += scratch[32 + w]; }
  a = ta; b = tb;
#else
  (void)scratch;
#endif
}
template <int NT = 0, bool PRE = true> __device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *scratch) {
#ifndef WORLD_EMU
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  int nw = wg_waves<NT>();
  if (nw == 1) return;
  if (PRE) __syncthreads();
  if (lane_id() == 0) { scratch[wave_in_block()] = a; scratch[16 + wave_in_block()] = b; scratch[32 + wave_in_block()] = c; }
  __syncthreads();
  double ta = 0.0, tb = 0.0, tc = 0.0;
  for (int w = 0; w < nw; ++w) { ta += scratch[w]; tb += scratch[16 + w]; tc += scratch[32 + w]; }
  a = ta; b = tb; c = tc;
#else
  (void)scratch;
#endif
}
template <int NT = 0> __device__ __forceinline__ void block_sum4(double &a, double &b, double &c, double &d, double *scratch) {
#ifndef WORLD_EMU
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d);
  int nw = wg_waves<NT>();
  if (nw == 1) return;
  __syncthreads();
  if (lane_id() == 0) {
    double *s = scratch + 4 * wave_in_block();
    s[0] = a; s[1] = b; s[2] = c; s[3] = d;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0, tc = 0.0, td = 0.0;
  for (int w = 0; w < nw; ++w) { ta += scratch[4 * w]; tb += scratch[4 * w + 1]; tc += scratch[4 * w + 2]; td += scratch[4 * w + 3]; }
  a = ta; b = tb; c = tc; d = td;
#else
  (void)scratch;
#endif
}
// Element-wise pass over n items, K per thread at a time: all K `produce(i)` calls (loads +
// arithmetic, returning a value) are issued before any `consume(i, value)` (the stores).
// A plain `for (i = tid; i < n; i += T) out[i] = f(in[i])` cannot overlap its iterations -- the
// compiler must keep each store ahead of the next iteration's loads -- so with 2-4 waves per SIMD
// such loops run at the latency of a trip to memory per iteration, not at throughput.  K loads in
// flight per thread close most of that gap.
template <int K, class T, int NT = 0, class Produce, class Consume>
__device__ __forceinline__ void block_map(int n, Produce produce, Consume consume) {
  const int tid = wg_thread<NT>(), nt = wg_size<NT>();
  for (int base = tid; base < n; base += K * nt) {
    T v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { const int i = base + q * nt; if (i < n) v[q] = produce(i); }
#pragma unroll
    for (int q = 0; q < K; ++q) { const int i = base + q * nt; if (i < n) consume(i, v[q]); }
  }
}

// exclusive scan of one int per thread over the block (thread order); *total = block sum
__device__ __forceinline__ int block_excl_scan_int(int v, int *total, double *scratch) {
#ifndef WORLD_EMU
  int wt, off = wave_excl_scan_int(v, &wt);
  int nw = waves_per_block();
  if (nw == 1) { *total = wt; return off; }
  int *is = reinterpret_cast<int *>(scratch);
  __syncthreads();
  if (lane_id() == 0) is[wave_in_block()] = wt;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) { int c = is[w]; if (w < wave_in_block()) base += c; tot += c; }
  *total = tot;
  return base + off;
#else
  (void)scratch;
  *total = v;
  return 0;
#endif
}

// exclusive scan of one 64-bit value per thread (four 16-bit counters packed together)
template <bool PRE = true>
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long *total,
                                                                  double *scratch) {
#ifndef WORLD_EMU
  // the four 16-bit fields never carry into each other, so the halves scan independently
  const int lane = lane_id();
  const unsigned lo = (unsigned)wave_incl_scan_int((int)(unsigned)v), hi = (unsigned)wave_incl_scan_int((int)(unsigned)(v >> 32));
  const unsigned long long inc = ((unsigned long long)hi << 32) | lo;
  const unsigned long long wt = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 63) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
  const int nw = waves_per_block();
  if (nw == 1) { *total = wt; return inc - v; }
  unsigned long long *us = reinterpret_cast<unsigned long long *>(scratch);
  if (PRE) __syncthreads();
  if (lane == 0) us[wave_in_block()] = wt;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) { unsigned long long c = us[w]; if (w < wave_in_block()) base += c; tot += c; }
  *total = tot;
  return base + (inc - v);
#else
  (void)scratch;
  *total = v;
  return 0;
#endif
}

// In-place inclusive prefix sum of a[0..n) in LDS by the whole block.
// Each thread sums a contiguous chunk serially, chunk totals are scanned across
// the block, then offsets are added back.  (With one thread -- the emulation --
// this is the plain serial left-to-right sum.)  NOT used where the reference's
// serial rounding is part of the result (CheapTrick's smoothing keeps a serial
// scan); used for D4C where the order is benign.
template <int NT = 0>
__device__ __forceinline__ void block_scan_incl_double(double *a, int n, double *scratch) {
  int nt = wg_size<NT>(), tid = wg_thread<NT>();
  int chunk = (n + nt - 1) / nt;
  int lo = tid * chunk, hi = lo + chunk < n ? lo + chunk : n;
  __syncthreads();
#ifndef WORLD_EMU
  // the thread's chunk is scanned in registers (loads first, then the dependent adds) and
  // written back once, already offset by everything to its left
  constexpr int kRegChunk = 12;
  if (chunk <= kRegChunk) {
    double v[kRegChunk];
#pragma unroll
    for (int q = 0; q < kRegChunk; ++q) v[q] = lo + q < hi ? a[lo + q] : 0.0;
#pragma unroll
    for (int q = 1; q < kRegChunk; ++q) v[q] += v[q - 1];
    const double s = v[kRegChunk - 1];
    const int lane = lane_id(), w = wave_in_block();
    const double inc = wave_incl_scan_f64(s);
    // (no barrier in front of the scratch write: whoever read this scratch area last did so before the entry barrier
    // above, and nothing between there and here touches it)
    if (lane == 63) scratch[w] = inc;
    __syncthreads();
    double base = 0.0;
    for (int k = 0; k < w; ++k) base += scratch[k];
    const double off = base + (inc - s);
#pragma unroll
    for (int q = 0; q < kRegChunk; ++q) if (lo + q < hi) a[lo + q] = v[q] + off;
    __syncthreads();
    return;
  }
#endif
  double s = 0.0;
  for (int i = lo; i < hi; ++i) { s += a[i]; a[i] = s; }
#ifndef WORLD_EMU
  // exclusive scan of chunk totals in thread order
  int lane = lane_id();
  const double inc = wave_incl_scan_f64(s);
  int nw = waves_per_block(), w = wave_in_block();
  if (lane == 63) scratch[w] = inc;                   // (as above: the entry barrier already separates the area's last readers)
  __syncthreads();
  double base = 0.0;
  for (int k = 0; k < w; ++k) base += scratch[k];
  double off = base + (inc - s);
  (void)nw;
  for (int i = lo; i < hi; ++i) a[i] += off;
#else
  (void)scratch;
#endif
  __syncthreads();
}
