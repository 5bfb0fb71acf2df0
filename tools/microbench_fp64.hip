// Development aid: dependent-issue latency and throughput of FP64 VALU ops on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/microbench_fp64.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, bool FMA>
__global__ void chain(double *out, long long *cycles, int iters, double x) {
  double a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x + c;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = FMA ? fma(a[c], x, x) : a[c] + x;
  }
  long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
template <int CHAINS, bool FMA> void run(const char *name, int blocks, int threads) {
  double *out; long long *cyc, h;
  hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8);
  const int iters = 4096;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((chain<CHAINS, FMA>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0000001);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s blocks %5d threads %4d: %.2f cycles per dependent step, %.2f cycles per op\n", name, blocks, threads,
         (double)h / iters, (double)h / iters / CHAINS);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, false>("add, 1 chain", 1, 64);
  run<2, false>("add, 2 chains", 1, 64);
  run<4, false>("add, 4 chains", 1, 64);
  run<8, false>("add, 8 chains", 1, 64);
  run<16, false>("add, 16 chains", 1, 64);
  run<1, true>("fma, 1 chain", 1, 64);
  run<8, true>("fma, 8 chains", 1, 64);
  run<1, false>("add, 1 chain, 4 waves/SIMD", 256 * 4, 256);
  run<1, false>("add, 1 chain, 8 waves/SIMD", 256 * 8, 256);
  run<4, false>("add, 4 chains, 4 waves/SIMD", 256 * 4, 256);
  return 0;
}
