// Development probe (not part of the library): how long does a dependent FP64 FMA take on this GPU?
//   hipcc --offload-arch=gfx950 -O3 -o fp64_chain fp64_chain.hip && ./fp64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, class T>
__global__ void chain(T *out, T a, T b, int steps) {
  T w[CHAINS];
  for (int c = 0; c < CHAINS; ++c) w[c] = (T)threadIdx.x + c;
  for (int s = 0; s < steps; s += 16) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) w[c] = __builtin_fma(a, w[c], b);
  }
  T r = 0;
  for (int c = 0; c < CHAINS; ++c) r += w[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int CHAINS, class T>
void run(const char *name, int threads, int blocks) {
  T *d; hipMalloc(&d, sizeof(T) * threads * blocks);
  const int steps = 1 << 20;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    chain<CHAINS, T><<<blocks, threads>>>(d, (T)0.999, (T)0.001, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s threads %4d blocks %4d: %.2f ns per step of %d chain(s) -> %.1f cycles @2.4GHz\n", name, threads, blocks,
         ms * 1e6 / steps, CHAINS, ms * 1e6 / steps * 2.4);
  hipFree(d);
}
int main() {
  run<1, double>("f64 1 chain, 1 wave", 64, 1);
  run<2, double>("f64 2 chains, 1 wave", 64, 1);
  run<4, double>("f64 4 chains, 1 wave", 64, 1);
  run<8, double>("f64 8 chains, 1 wave", 64, 1);
  run<1, double>("f64 1 chain, 4 waves/1 WG", 256, 1);
  run<1, double>("f64 1 chain, 8 waves/1 WG", 512, 1);
  run<1, double>("f64 1 chain, 16 waves/1 WG", 1024, 1);
  run<1, float>("f32 1 chain, 1 wave", 64, 1);
  run<4, float>("f32 4 chains, 1 wave", 64, 1);
  run<1, double>("f64 1 chain, 256 WG x 4 waves", 256, 256);
  return 0;
}
