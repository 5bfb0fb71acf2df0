// machine_probe.hip -- 50 ms of microbenchmarks that say what box a measurement was taken on (two of
// round 4's seven boxes ran a lone job's latency-bound kernels 10-90 % slower with identical binaries, and nothing in the
// bench line could tell such a box from a regression).  bench.py puts the numbers into the line's `environment` object
// next to rocm-smi's clocks, power cap and partition modes.  Replaces nothing in the reference; measurement only.
//   * the shader clock the chip actually holds under a chip-wide FP64 load (s_memtime ticks per s_memrealtime tick), and
//     the FMA rate of that load -- a power-capped or down-clocked box shows here;
//   * dependent-load latency of one lane chasing pointers through 2 GB (HBM + fabric + TLB), through 64 MB that all CUs
//     have just read (Infinity Cache), through 1 MB read once before (the XCD's L2) -- what the lone job's short
//     latency-bound kernels are made of;
//   * a dependent chain of LDS reads on an otherwise idle CU and on a CU whose other wavefronts stream 16-byte LDS
//     accesses and FP64 work (the frame kernels' regime: round 4 measured ~100 against ~750 cycles).
#include "machine_probe.h"
#include <algorithm>
#include <vector>

namespace world_hip {

#ifndef WORLD_EMU
// next[i] = (A i + C) mod n, A = 1 mod 4, C odd: a permutation of [0, n) that is ONE cycle (Hull-Dobell); element i lives
// at buf[i * stride] (stride in 8-byte words: 16 = one element per 128-byte line)
constexpr unsigned long long kLcgA = 6364136223846793005ull, kLcgC = 1442695040888963407ull;
__global__ void mp_fill(unsigned long long *buf, unsigned long long n, int stride) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
    buf[i * stride] = ((kLcgA * i + kLcgC) & (n - 1)) * stride;
}
__global__ void mp_touch(const unsigned long long *buf, unsigned long long words, unsigned long long *sink) {
  unsigned long long acc = 0;
  for (unsigned long long i = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 16; i < words; i += (unsigned long long)gridDim.x * blockDim.x * 16)
    acc += buf[i];
  if (acc == 0x1234567ull) *sink = acc;
}
// one lane follows `hops` pointers (after `skip` untimed ones); ticks of the 100 MHz real-time counter
__global__ void mp_chase(const unsigned long long *buf, int skip, int hops, unsigned long long *out) {
  if (threadIdx.x != 0) return;
  unsigned long long p = 0;
  for (int i = 0; i < skip; ++i) p = buf[p];
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < hops; ++i) p = buf[p];
  const unsigned long long t1 = wall_clock64() + (p == ~0ull ? 1 : 0);
  out[0] = t1 - t0; out[1] = p;
}
// every thread: `iters` rounds of four independent FMA chains; workgroup 0 reads both clocks around its loop
__global__ void __launch_bounds__(256) mp_fp64_load(int iters, double a, double b, double *sink, unsigned long long *out) {
  double x0 = a + threadIdx.x, x1 = a - threadIdx.x, x2 = b + threadIdx.x, x3 = b * 0.5;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (x0 + x1 + x2 + x3 == 12345.678) *sink = x0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
// wavefront 0 of every workgroup walks a dependent chain of LDS reads; with busy != 0 the other three stream 16-byte LDS
// reads and writes and FP64 work until it is done.  out[workgroup] = shader-clock ticks for kLdsSteps steps.
constexpr int kLdsSteps = 256, kLdsWords = 4096;
__global__ void __launch_bounds__(256) mp_lds(int busy, unsigned *out) {
  __shared__ int chain[kLdsWords];
  __shared__ double2 lane_area[1024];
  __shared__ int done;
  for (int i = threadIdx.x; i < kLdsWords; i += blockDim.x) chain[i] = (int)((1664525u * (unsigned)i + 1013904223u) & (kLdsWords - 1));
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lane_area[i] = make_double2(1.0 + i, 2.0);
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (threadIdx.x < 64) {
    int p = threadIdx.x;
    const unsigned long long c0 = clock64();
    for (int i = 0; i < kLdsSteps; ++i) p = chain[p];
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = (unsigned)(c1 - c0) + (p == -1 ? 1u : 0u); __atomic_store_n(&done, 1, __ATOMIC_RELAXED); }
  } else if (busy) {
    double2 v = lane_area[threadIdx.x];
    for (int it = 0; it < 100000 && __atomic_load_n(&done, __ATOMIC_RELAXED) == 0; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double2 t = lane_area[(threadIdx.x + 64 * k + it) & 1023];
        v.x = fma(v.x, 1.0000001, t.x); v.y = fma(v.y, 0.9999999, t.y);
        lane_area[threadIdx.x + 0 * k] = v;
      }
    }
  }
}

static double chase_ns(unsigned long long *buf, unsigned long long n, int stride, int skip, int hops, bool touch_all,
                       unsigned long long *d_out, hipStream_t stream) {
  hipLaunchKernelGGL(mp_fill, dim3(2048), dim3(256), 0, stream, buf, n, stride);
  if (touch_all) hipLaunchKernelGGL(mp_touch, dim3(4096), dim3(256), 0, stream, buf, n * stride, d_out + 8);
  hipLaunchKernelGGL(mp_chase, dim3(1), dim3(64), 0, stream, buf, skip, hops, d_out);
  unsigned long long h[2] = {0, 0};
  devrt::d2h(h, d_out, sizeof h, stream);
  devrt::sync(stream);
  return (double)h[0] * 10.0 / hops;                               // 100 MHz ticks -> ns per hop
}

void run_machine_probe(double *out, hipStream_t stream) {
  for (int i = 0; i < kMachineProbeValues; ++i) out[i] = 0.0;
  int dev = devrt::current_device(), cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  out[7] = cus;
  unsigned long long *d_out = static_cast<unsigned long long *>(devrt::dmalloc(4096 * sizeof(unsigned)+ 256));
  // ---- clock and FMA rate under load: 4 workgroups per CU, ~5 ms
  {
    const int iters = 100000, wgs = 4 * (cus > 0 ? cus : 256);
    hipLaunchKernelGGL(mp_fp64_load, dim3(wgs), dim3(256), 0, stream, 2000, 1.0000000001, 1e-9, reinterpret_cast<double *>(d_out + 8), d_out);   // warm-up (clocks ramp)
    // The rate is taken over the WHOLE launch (HIP events): workgroup 0's own loop -- what rounds 4-5 timed -- ends early,
    // because the SIMD's arbiter favours its oldest wavefronts over the three younger ones it shares the SIMD with
    // (that reading, 108-114 TFLOP/s, exceeded the 78.6 the part can do at 2.4 GHz: VERDICT r05 weak 7).  The clock ratio
    // still comes from workgroup 0's two counters.
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (timed) (void)hipEventRecord(e0, stream);
    hipLaunchKernelGGL(mp_fp64_load, dim3(wgs), dim3(256), 0, stream, iters, 1.0000000001, 1e-9, reinterpret_cast<double *>(d_out + 8), d_out);
    if (timed) (void)hipEventRecord(e1, stream);
    unsigned long long h[2] = {0, 0};
    devrt::d2h(h, d_out, sizeof h, stream);
    devrt::sync(stream);
    float ms = 0.f;
    if (timed && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = 0.f;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (h[1]) {
      out[0] = (double)h[0] / (double)h[1] * 100.0;                // shader ticks per 100 MHz tick -> MHz
      const double flop = (double)wgs * 256.0 * iters * 8.0 * 2.0;
      out[1] = ms > 0.f ? flop / (ms * 1e-3) / 1e12 : flop / ((double)h[1] * 10e-9) / 1e12;
    }
  }
  // ---- dependent-load latency
  {
    const unsigned long long n_hbm = 1ull << 24;                  // 16 M lines of 128 B = 2 GB: beyond the 256 MB Infinity Cache
    unsigned long long *buf = static_cast<unsigned long long *>(devrt::dmalloc(n_hbm * 128));
    out[2] = chase_ns(buf, n_hbm, 16, 0, 2048, false, d_out, stream);
    out[3] = chase_ns(buf, 1ull << 19, 16, 0, 2048, true, d_out, stream);       // 64 MB, just read by every CU
    out[4] = chase_ns(buf, 1ull << 13, 16, 1 << 13, 1 << 13, false, d_out, stream);   // 1 MB: one full cycle untimed, the next timed
    devrt::sync(stream);
    devrt::dfree(buf);
  }
  // ---- an LDS round trip, idle and loaded
  for (int busy = 0; busy < 2; ++busy) {
    const int wgs = busy ? 4 * (cus > 0 ? cus : 256) : 1;
    hipLaunchKernelGGL(mp_lds, dim3(wgs), dim3(256), 0, stream, busy, reinterpret_cast<unsigned *>(d_out));
    std::vector<unsigned> h(wgs);
    devrt::d2h(h.data(), d_out, sizeof(unsigned) * wgs, stream);
    devrt::sync(stream);
    std::sort(h.begin(), h.end());
    out[5 + busy] = (double)h[wgs / 2] / kLdsSteps;                // the median workgroup
  }
  devrt::dfree(d_out);
}
#else
void run_machine_probe(double *out, hipStream_t) {
  for (int i = 0; i < kMachineProbeValues; ++i) out[i] = 0.0;     // nothing to measure in the host emulation
}
#endif

}  // namespace world_hip
