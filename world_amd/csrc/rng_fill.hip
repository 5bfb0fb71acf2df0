// rng_fill.hip -- the reference's randn() stream (src/matlabfunctions.cpp:237-264)
// materialised in HBM, one utterance stream per row, by jump-ahead (see rng.h).
#include "rng.h"

namespace world_hip {

__global__ void rng_stream_fill(RngFillArgs a) {
  const int u = blockIdx.y;
  const size_t start = (size_t)flat_thread_x() * kFillRun;
  const unsigned cnt = a.count[u];
  if (start >= cnt) return;
  const unsigned first = (a.begin ? a.begin[u] : 0u) + (unsigned)start;
  Xs128 s = xs_jump(a.jump, xs_seed(), first);
  double *out = a.noise + (size_t)u * a.stride + start;
  const int n = cnt - start < (size_t)kFillRun ? (int)(cnt - start) : kFillRun;
  for (int i = 0; i < n; ++i) out[i] = xs_randn(s);
}

void launch_rng_fill(const RngFillArgs &a, int n_utt, size_t max_count, hipStream_t stream) {
  WH_THREADS(rng_stream_fill, (long)((max_count + kFillRun - 1) / kFillRun), n_utt, 1, stream, a);
}

}  // namespace world_hip
