// rng_fill.hip -- the reference's randn() stream (src/matlabfunctions.cpp:237-264)
// materialised in HBM by jump-ahead (see rng.h): randn_value(noise[k]) = k-th draw after reseed.
#include "rng.h"

namespace world_hip {

__global__ void rng_stream_fill(RngFillArgs a) {
  const size_t start = a.begin + (size_t)blockIdx.y * ((size_t)gridDim.x * blockDim.x) * kFillRun +
                       (size_t)flat_thread_x() * kFillRun;
  if (start >= a.end) return;
  Xs128 s = xs_jump(a.jump, xs_seed(), (uint32_t)start);
  const int n = a.end - start < (size_t)kFillRun ? (int)(a.end - start) : kFillRun;
  for (int i = 0; i < n; ++i) a.noise[start + i] = xs_randn_word(s);
}

void launch_rng_fill(const RngFillArgs &a, hipStream_t stream) {
  if (a.end <= a.begin) return;
  const size_t runs = (a.end - a.begin + kFillRun - 1) / kFillRun;
  // x carries up to 2^20 runs, y the rest (a launch dimension is limited to 2^31-1 threads)
  const size_t per_y = (size_t)1 << 20;
  const unsigned ny = (unsigned)((runs + per_y - 1) / per_y);
  WH_THREADS(rng_stream_fill, (long)(runs < per_y ? runs : per_y), ny, 1, stream, a);
}

}  // namespace world_hip
