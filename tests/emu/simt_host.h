// simt_host.h -- TEST INFRASTRUCTURE: a wave-accurate host emulation of ONE translation unit's GPU code path.
//
// The classic host emulation (devrt.h, -DWORLD_EMU) shrinks every workgroup to one thread and every wavefront to one lane,
// so whatever a kernel does with DPP, v_readlane, ballots or register-resident butterflies needs a second, plainer spelling
// under #ifdef WORLD_EMU -- d4c.hip had 23 such sites, and only the GPU suite ever ran the shipped ones (VERDICT r05 missing 4).
// A unit compiled with -DWORLD_SIMT (and WITHOUT -DWORLD_EMU) is compiled as the GPU compiles it -- WAVE = 64, the real
// workgroup sizes, the #ifndef WORLD_EMU branches -- against this header instead of <hip/hip_runtime.h>:
//   * every thread of a workgroup is a FIBRE (its own stack, a ten-instruction context switch); blocks run one after the
//     other; __syncthreads() parks a fibre until all live fibres of the block have arrived;
//   * a cross-lane operation (DPP, readlane, readfirstlane, ballot, __shfl) publishes the lane's operand, parks the fibre
//     until all live lanes of its wavefront have arrived at the same operation, then reads its partner's operand: the 64
//     lanes proceed in lock-step BETWEEN such operations, which is all these instructions can observe;
//   * atomics on LDS are plain read-modify-writes (fibres are cooperative); LDS is one buffer per block; global memory is
//     the host's.
// Nothing here is shipped; the unit lives in its own shared object (tests/emu/Makefile) whose inline functions stay
// local, so its WAVE = 64 spellings of devrt.h / fft.h cannot be merged with the one-lane spellings of the other units.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <type_traits>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline double2 make_double2(double a, double b) { double2 r; r.x = a; r.y = b; return r; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint4 r = {a, b, c, d}; return r; }
static inline int4 make_int4(int a, int b, int c, int d) { int4 r = {a, b, c, d}; return r; }

typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define WAVE 64

namespace simt {
struct Fibre {
  void *sp = nullptr;          // saved stack pointer while parked
  char *stack = nullptr;
  dim3 tid;
  int lane = 0, wave = 0;
  int state = 0;               // 0 runnable, 1 at the block barrier, 2 at a wavefront rendezvous, 3 done
};
struct Wave {
  int alive = 0, arrived = 0;
  unsigned long long seq = 0;                 // rendezvous counter: the operands of operation `seq` live in buf[seq & 1]
  unsigned long long buf[2][64];
  unsigned long long arrived_mask[2] = {0, 0};
  const void *site[2] = {nullptr, nullptr};    // the instruction (call site) the lanes of this rendezvous came from: all the same, or abort
};
// the running block (thread-local: host threads emulate independent launches side by side)
struct Block {
  dim3 block_idx, block_dim, grid_dim;
  char *lds = nullptr;
  Fibre *cur = nullptr;
};
extern thread_local Block g_block;
void run_grid(dim3 grid, int threads, size_t lds_bytes, const std::function<void()> &body, const char *name);
void barrier();                               // __syncthreads
// wavefront rendezvous: publish `mine`, wait for the live lanes, return the buffer of this operation and the mask of lanes in it
const unsigned long long *exchange(unsigned long long mine, unsigned long long *mask);
}  // namespace simt

#define threadIdx (simt::g_block.cur->tid)
#define blockIdx (simt::g_block.block_idx)
#define blockDim (simt::g_block.block_dim)
#define gridDim (simt::g_block.grid_dim)
#define DYN_LDS(name) char *name = simt::g_block.lds
#define LDS_PTR(T) T *
static inline void __syncthreads() { simt::barrier(); }

// ---- host runtime of devrt.h (the classic emulation's: tests/emu/emu_rt.cpp defines it in libworld_emu.so; this unit only
// launches, so it needs none of it) --------------------------------------------------------------------------------------
namespace devrt {
// (device memory is the host's: the few host-side helpers units call between launches)
static inline void *dmalloc(size_t bytes) { void *p = malloc(bytes ? bytes : 1); memset(p, 0xA5, bytes); return p; }
static inline void dfree(void *p) { free(p); }
static inline void h2d(void *dst, const void *src, size_t n, hipStream_t) { memcpy(dst, src, n); }
static inline void d2h(void *dst, const void *src, size_t n, hipStream_t) { memcpy(dst, src, n); }
static inline void d2d(void *dst, const void *src, size_t n, hipStream_t) { memcpy(dst, src, n); }
static inline void dzero(void *dst, size_t n, hipStream_t) { memset(dst, 0, n); }
static inline void sync(hipStream_t) {}
template <class K, class... A>
void launch_blocks(const char *name, K kernel, dim3 grid, int threads, size_t lds, hipStream_t, A... args) {
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
  simt::run_grid(grid, threads, lds, [&] { kernel(args...); }, name);
}
}  // namespace devrt

// ---- scalar intrinsics ---------------------------------------------------------------------------------------------------
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
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
static inline int __double2hiint(double v) { return (int)(__double_as_longlong(v) >> 32); }
static inline int __double2loint(double v) { return (int)(unsigned)(__double_as_longlong(v) & 0xffffffffll); }
static inline double __hiloint2double(int hi, int lo) {
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
static inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }
#define __builtin_assume(x) ((void)0)
#define __builtin_readcyclecounter() 0ull
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_dispatch_ptr() ((const void *)nullptr)
// (v_rcp_f64 is accurate to an ulp and fast_div refines it: the emulation's exact reciprocal gives the correctly rounded
// quotient the refinement converges to)

// ---- cross-lane operations ---------------------------------------------------------------------------------------------------
static inline void __builtin_amdgcn_wave_barrier() { unsigned long long m; (void)simt::exchange(0, &m); }
static inline int __builtin_amdgcn_readlane(int v, int lane) {
  unsigned long long m;
  const unsigned long long *b = simt::exchange((unsigned long long)(unsigned)v, &m);
  return (int)(unsigned)b[lane & 63];
}
static inline int __builtin_amdgcn_readfirstlane(int v) {
  unsigned long long m;
  const unsigned long long *b = simt::exchange((unsigned long long)(unsigned)v, &m);
  return (int)(unsigned)b[__builtin_ctzll(m)];
}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) {
  unsigned long long m, r = 0;
  const unsigned long long *b = simt::exchange(p ? 1ull : 0ull, &m);
  for (int l = 0; l < 64; ++l) if (((m >> l) & 1ull) && b[l]) r |= 1ull << l;
  return r;
}
static inline unsigned long long __ballot(int p) { return __builtin_amdgcn_ballot_w64(p != 0); }
static inline int __shfl(int v, int src, int width = 64) {
  unsigned long long m;
  const unsigned long long *b = simt::exchange((unsigned long long)(unsigned)v, &m);
  const int me = simt::g_block.cur->lane;
  return (int)(unsigned)b[(me & ~(width - 1)) | (src & (width - 1))];
}
static inline double __shfl(double v, int src, int width = 64) {
  unsigned long long m;
  const unsigned long long *b = simt::exchange((unsigned long long)__double_as_longlong(v), &m);
  const int me = simt::g_block.cur->lane;
  return __longlong_as_double((long long)b[(me & ~(width - 1)) | (src & (width - 1))]);
}
static inline int __shfl_xor(int v, int mask, int width = 64) { return __shfl(v, (simt::g_block.cur->lane ^ mask) & (width - 1), width); }
static inline double __shfl_xor(double v, int mask, int width = 64) { return __shfl(v, (simt::g_block.cur->lane ^ mask) & (width - 1), width); }
// v_mov_b32_dpp: the controls this tree uses (quad_perm, row_shl / shr / ror, row_mirror, row_half_mirror, row_bcast15 / 31)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  unsigned long long m;
  const unsigned long long *b = simt::exchange((unsigned long long)(unsigned)src, &m);
  const int i = simt::g_block.cur->lane, row = i >> 4, r = i & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> (r >> 2)) & 1)) return old;
  int from = -1;                                           // -1: no source lane (out of the row)
  if (ctrl >= 0 && ctrl <= 0xFF) from = (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; from = r + n <= 15 ? i + n : -1; }
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; from = r - n >= 0 ? i - n : -1; }
  else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; from = (i & ~15) | ((r - n) & 15); }
  else if (ctrl == 0x140) from = (i & ~15) | (15 - r);
  else if (ctrl == 0x141) from = (i & ~7) | (7 - (i & 7));
  else if (ctrl == 0x142) from = row > 0 ? 16 * row - 1 : -1;                  // row_bcast15: lane 15 of the previous row
  else if (ctrl == 0x143) from = row > 1 ? 31 : -1;                            // row_bcast31: lane 31 to rows 2 and 3
  else { fprintf(stderr, "simt_host.h: DPP control 0x%x is not emulated\n", ctrl); abort(); }
  if (from < 0 || !((m >> from) & 1ull)) return bound_ctrl ? 0 : old;
  return (int)(unsigned)b[from];
}
// v_permlane16_swap / v_permlane32_swap of two copies of one value (the only use in this tree: devrt.h row_pair_sum /
// half_pair_sum): [0] = the value of the even row / lower half at this lane's position, [1] = the odd row's / upper half's
struct SimtPair { int v[2]; int operator[](int k) const { return v[k]; } };
static inline SimtPair __builtin_amdgcn_permlane16_swap(int a, int b, bool, bool) {
  if (a != b) { fprintf(stderr, "simt_host.h: v_permlane16_swap is emulated for two copies of one value only\n"); abort(); }
  unsigned long long m;
  const unsigned long long *buf = simt::exchange((unsigned long long)(unsigned)a, &m);
  const int i = simt::g_block.cur->lane;
  SimtPair r; r.v[0] = (int)(unsigned)buf[i & ~16]; r.v[1] = (int)(unsigned)buf[i | 16];
  return r;
}
static inline SimtPair __builtin_amdgcn_permlane32_swap(int a, int b, bool, bool) {
  if (a != b) { fprintf(stderr, "simt_host.h: v_permlane32_swap is emulated for two copies of one value only\n"); abort(); }
  unsigned long long m;
  const unsigned long long *buf = simt::exchange((unsigned long long)(unsigned)a, &m);
  const int i = simt::g_block.cur->lane;
  SimtPair r; r.v[0] = (int)(unsigned)buf[i & 31]; r.v[1] = (int)(unsigned)buf[(i & 31) | 32];
  return r;
}
