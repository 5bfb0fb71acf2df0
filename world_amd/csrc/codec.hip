// codec.hip -- low-dimensional coding of the analysis outputs (reference src/codec.cpp).
//
//   code_sp   : log envelope -> interp1 onto the mel axis -> DCT-II through one real FFT
//               of fft_size/2 points -> first `ndim` coefficients      (codec.cpp:73-87,120-130,268-297)
//   decode_sp : coefficients -> IDCT through one complex FFT -> interp1 back to the linear
//               axis -> exp                                            (codec.cpp:93-115,138-156,299-324)
//   code_ap   : 20 log10(ap) sampled at 3 kHz multiples (interp1Q)     (codec.cpp:217-236)
//   decode_ap : band values -> interp1 over [0, 3k.., fs/2] -> 10^(x/20), aperiodic
//               frames (mean band value > -0.5 dB) left at 1 - 1e-12   (codec.cpp:21-56,238-266)
//
// Both interp1 calls have fixed knots and fixed queries, so their bin search and weights
// are tables built once on the host (api.hip: codec_tables).  One 256-thread workgroup
// per frame for the envelope kernels, one thread per output value for the band kernels.
#include "codec.h"
#include "fft.h"

namespace world_hip {

static size_t code_sp_lds_bytes(int lg_md) {
  const size_t md = (size_t)1 << lg_md;
  return sizeof(double) * (md + 2 + md + 16 + twiddle_lds_doubles(lg_md));
}
static size_t decode_sp_lds_bytes(int lg_md) {
  const size_t md = (size_t)1 << lg_md;
  return sizeof(double) * (2 * md + 16 + md + 2 + twiddle_lds_doubles(lg_md));
}

__global__ void __launch_bounds__(256) codec_code_sp(CodecParams p) {
  DYN_LDS(lds);
  const int row = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int md = 1 << p.lg_md, nb = md + 1;
  double *lg = reinterpret_cast<double *>(lds);                       // log envelope, nb bins
  cplx *Z = reinterpret_cast<cplx *>(lg + md + 2);                    // real FFT input, md reals
  const TwLds tw = stage_twiddles(reinterpret_cast<double *>(Z) + md + 16, p.lg_md, p.tab.tw);
  const double *in = p.in + (size_t)row * (p.in_stride ? p.in_stride : (size_t)nb);
  for (int i = tid; i < nb; i += nt) lg[i] = log(in[i]);
  __syncthreads();
  // interp1 onto the mel axis, written straight into DCTForCodec's even/odd reordering
  for (int i = tid; i < md; i += nt) {
    const int k = p.knot[i];
    const double v = lg[k - 1] + p.frac[i] * (lg[k] - lg[k - 1]);
    const int dest = (i & 1) ? md / 2 + (md - 1 - i) / 2 : i / 2;
    rfft_in(Z, dest) = v;
  }
  const double norm = sqrt(static_cast<double>(md));
  double *out = p.out + (size_t)row * (p.out_stride ? p.out_stride : (size_t)p.ndim);
  block_rfft(Z, p.lg_md, tw, [&](int k, double re, double im) {
    if (k < p.ndim) out[k] = (re * p.w_re[k] - im * p.w_im[k]) / norm;
  });
}

__global__ void __launch_bounds__(256) codec_decode_sp(CodecParams p) {
  DYN_LDS(lds);
  const int row = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int md = 1 << p.lg_md, nb = md + 1;
  cplx *Z = reinterpret_cast<cplx *>(lds);                            // md complex points
  double *mel = reinterpret_cast<double *>(lds) + 2 * md + 16;        // md + 2 values
  const TwLds tw = stage_twiddles(mel + md + 2, p.lg_md, p.tab.tw);
  const double *in = p.in + (size_t)row * (p.in_stride ? p.in_stride : (size_t)p.ndim);
  const double norm = sqrt(static_cast<double>(md));
  for (int i = tid; i < md; i += nt) {
    cplx v; v.re = 0.0; v.im = 0.0;
    if (i < p.ndim) { const double c = in[i]; v.re = c * p.w_re[i] * norm; v.im = -c * p.w_im[i] * norm; }
    Z[swz(i)] = v;
  }
  // The reference's backward c2c plan returns conj(sum_j in[j] e^{-2 pi i jk/n}) (fft.cpp:36-45);
  // only the real part is read, so a forward transform is what is needed.
  const FftPlan plan = make_plan(p.lg_md);
  block_cfft_dif(Z, plan, tw);
  for (int i = tid; i < md / 2; i += nt) {
    mel[1 + 2 * i] = Z[fft_slot(plan, i)].re;
    mel[2 + 2 * i] = Z[fft_slot(plan, md - 1 - i)].re;
  }
  __syncthreads();
  if (tid == 0) { mel[0] = mel[1]; mel[md + 1] = mel[md]; }
  __syncthreads();
  double *out = p.out + (size_t)row * (p.out_stride ? p.out_stride : (size_t)nb);
  for (int j = tid; j < nb; j += nt) {
    const int k = p.knot[j];
    const double v = mel[k - 1] + p.frac[j] * (mel[k] - mel[k - 1]);
    out[j] = exp(v / md);
  }
}

__global__ void codec_code_ap(CodecParams p) {
  const int item = flat_thread_x();
  if (item >= p.rows * p.ndim) return;
  const int row = item / p.ndim, band = item - row * p.ndim;
  const int nb = p.fft_size / 2 + 1;
  const double *in = p.in + (size_t)row * (p.in_stride ? p.in_stride : (size_t)nb);
  // interp1Q(0, fs/fft_size, 20 log10(ap), nb, 3000 (band+1)) -- matlabfunctions.cpp:214-235
  const double pos = (3000.0 * (band + 1.0) - 0) / (static_cast<double>(p.fs) / p.fft_size);
  const int b = static_cast<int>(pos);
  const double fr = pos - b;
  const double y0 = 20 * log10(in[b]);
  const double dy = b < nb - 1 ? 20 * log10(in[b + 1]) - y0 : 0.0;
  p.out[(size_t)row * (p.out_stride ? p.out_stride : (size_t)p.ndim) + band] = y0 + dy * fr;
}

__global__ void codec_decode_ap(CodecParams p) {
  const int nb = p.fft_size / 2 + 1;
  const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= (long)p.rows * nb) return;
  const int row = (int)(item / nb), j = (int)(item - (long)row * nb);
  const double *in = p.in + (size_t)row * (p.in_stride ? p.in_stride : (size_t)p.ndim);
  double mean = 0.0;                                  // CheckVUV, codec.cpp:31-41
  for (int i = 0; i < p.ndim; ++i) mean += in[i];
  mean /= p.ndim;
  double v = 1.0 - kTiny;                             // InitializeAperiodicity, codec.cpp:21-26
  if (!(mean > -0.5)) {
    // coarse = [-60, bands..., -1e-12] over knots [0, 3000.., fs/2]
    const int k = p.knot[j];
    const double lo = k - 1 == 0 ? -60.0 : in[k - 2];
    const double hi = k == p.ndim + 1 ? -kTiny : in[k - 1];
    v = pow(10.0, (lo + p.frac[j] * (hi - lo)) / 20.0);
  }
  p.out[(size_t)row * (p.out_stride ? p.out_stride : (size_t)nb) + j] = v;
}

void launch_code_spectral_envelope(const CodecParams &p, hipStream_t stream) {
  WH_BLOCKS(codec_code_sp, dim3(p.rows), 256, code_sp_lds_bytes(p.lg_md), stream, p);
}
void launch_decode_spectral_envelope(const CodecParams &p, hipStream_t stream) {
  WH_BLOCKS(codec_decode_sp, dim3(p.rows), 256, decode_sp_lds_bytes(p.lg_md), stream, p);
}
void launch_code_aperiodicity(const CodecParams &p, hipStream_t stream) {
  WH_THREADS(codec_code_ap, (long)p.rows * p.ndim, 1, 1, stream, p);
}
void launch_decode_aperiodicity(const CodecParams &p, hipStream_t stream) {
  WH_THREADS(codec_decode_ap, (long)p.rows * (p.fft_size / 2 + 1), 1, 1, stream, p);
}

}  // namespace world_hip
