// pcm.hip -- device-side input/output conditioning: the PCM
// samples of a WAV file <-> the doubles wavread()/wavwrite() exchange with the analysis
// (tools/audioio.cpp:123-127 and :230-249).  Uploading the file's bytes and widening on
// the GPU moves 8/qb x fewer bytes over PCIe than uploading FP64; every operation here is
// exact in FP64, so the results are bit-identical to the reference's.
#include "common.h"

namespace world_hip {

__global__ void pcm16_to_double(const short *pcm, double *x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = static_cast<double>(pcm[i]) / 32768.0;
}

// wavread()'s sample decode (tools/audioio.cpp:236-249) for qb = nbit/8 in 1..4 bytes per
// sample, little endian: the top byte's high bit selects sign_bias = 2^(nbit-1) and is
// cleared, the bytes accumulate base 256, and x = (tmp - sign_bias) / 2^(nbit-1).  nbit
// need not be 8*qb (the reference takes it from the header as is), hence the two arguments.
__global__ void pcm_bytes_to_double(const unsigned char *pcm, double *x, long n, int qb, double zero_line) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char *s = pcm + i * qb;
  unsigned top = s[qb - 1];
  const double sign_bias = top >= 128 ? zero_line : 0.0;
  double tmp = static_cast<double>(top & 0x7Fu);
  for (int j = qb - 2; j >= 0; --j) tmp = tmp * 256.0 + static_cast<double>(s[j]);
  x[i] = (tmp - sign_bias) / zero_line;
}

// wavwrite()'s quantiser (tools/audioio.cpp:123-127): int16(max(-32768, min(32767, int(x * 32767)))).
// int(double) truncates toward zero; outside int range (and for NaN) the x86 conversion the
// reference compiles to yields INT_MIN, which the clamp turns into -32768 -- kept, so that a
// wildly out-of-range sample lands where the reference puts it.
__global__ void double_to_pcm16(const double *x, short *pcm, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i] * 32767.0;
  int q;
  if (!(v > -2147483649.0 && v < 2147483648.0)) q = -32768;
  else q = static_cast<int>(v);
  pcm[i] = static_cast<short>(q > 32767 ? 32767 : (q < -32768 ? -32768 : q));
}

void launch_pcm16_to_double(const short *d_pcm, double *d_x, long n, hipStream_t stream) {
  WH_THREADS(pcm16_to_double, n, 1, 1, stream, d_pcm, d_x, n);
}
void launch_pcm_bytes_to_double(const unsigned char *d_pcm, double *d_x, long n, int qb, double zero_line,
                                hipStream_t stream) {
  WH_THREADS(pcm_bytes_to_double, n, 1, 1, stream, d_pcm, d_x, n, qb, zero_line);
}
void launch_double_to_pcm16(const double *d_x, short *d_pcm, long n, hipStream_t stream) {
  WH_THREADS(double_to_pcm16, n, 1, 1, stream, d_x, d_pcm, n);
}

}  // namespace world_hip
