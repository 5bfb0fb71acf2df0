// pcm.hip -- device-side input conditioning: 16-bit PCM as stored in a
// WAV file -> the doubles wavread() hands to the analysis (tools/audioio.cpp:236-249:
// x = q / 2^(nbit-1)).  Uploading int16 and widening on the GPU moves 4x fewer bytes
// over PCIe than uploading FP64; the division by a power of two is exact.
#include "common.h"

namespace world_hip {

__global__ void pcm16_to_double(const short *pcm, double *x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = static_cast<double>(pcm[i]) / 32768.0;
}

void launch_pcm16_to_double(const short *d_pcm, double *d_x, long n, hipStream_t stream) {
  WH_THREADS(pcm16_to_double, n, 1, 1, stream, d_pcm, d_x, n);
}

}  // namespace world_hip
