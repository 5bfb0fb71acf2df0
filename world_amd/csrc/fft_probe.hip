// fft_probe.hip -- fft.h in isolation: batches of real transforms straight from and to HBM, one
// workgroup per transform, exactly the block_rfft / block_irfft instantiations the stage kernels use
// (radix-8 and radix-16 plans, 256 .. 8192 points).  Replaces nothing in the reference by itself:
// it is the measurement and test hook for the reference's fft_plan_dft_r2c_1d / fft_plan_dft_c2r_1d /
// fft_execute (src/world/fft.h:22-44, src/fft.cpp:143-212) as re-implemented in fft.h --
// tests/test_gpu_fft.py checks it against numpy.fft, tools/fft_microbench.py times it per N.
#include "common.h"
#include "fft_probe.h"

namespace world_hip {

// r2c: in [batch][N] doubles -> out [batch][N/2+1] (re, im); X[k] = sum x[n] e^{-2 pi i k n / N}
template <int MAXLR, int LGN = 0>
__global__ void fft_probe_rfft(const double *in, double2 *out, int lgn, const double2 *tw_global) {
  DYN_LDS(lds);
  const int N = 1 << lgn;
  cplx *Z = reinterpret_cast<cplx *>(lds);
  const TwLds tw = stage_twiddles(reinterpret_cast<double *>(lds) + N, lgn - 1, tw_global);
  const double *x = in + (size_t)blockIdx.x * N;
  double2 *X = out + (size_t)blockIdx.x * (N / 2 + 1);
  for (int i = threadIdx.x; i < N / 2; i += blockDim.x) {          // two samples per 16-byte slot
    cplx v; v.re = x[2 * i]; v.im = x[2 * i + 1];
    Z[swz(i)] = v;
  }
  block_rfft<MAXLR, LGN>(Z, lgn, tw, [&](int k, double re, double im) { X[k] = make_double2(re, im); });
}

// c2r, unscaled (N * irfft, imaginary parts of DC / Nyquist ignored): spec [batch][N/2+1] -> out [batch][N]
template <int MAXLR, int LGN = 0>
__global__ void fft_probe_irfft(const double2 *spec, double *out, int lgn, const double2 *tw_global) {
  DYN_LDS(lds);
  const int N = 1 << lgn;
  cplx *Z = reinterpret_cast<cplx *>(lds);
  const TwLds tw = stage_twiddles(reinterpret_cast<double *>(lds) + N, lgn - 1, tw_global);
  const double2 *X = spec + (size_t)blockIdx.x * (N / 2 + 1);
  double *y = out + (size_t)blockIdx.x * N;
  block_irfft<MAXLR, LGN>(Z, lgn, tw, [&](int k) { const double2 v = X[k]; cplx c; c.re = v.x; c.im = v.y; return c; });
  for (int i = threadIdx.x; i < N / 2; i += blockDim.x) {
    const cplx v = Z[swz(i)];
    y[2 * i] = v.re; y[2 * i + 1] = v.im;
  }
}

size_t fft_probe_lds_bytes(int lgn) { return sizeof(double) * ((size_t)(1 << lgn) + twiddle_lds_doubles(lgn - 1)); }

// static_plan: the instantiations whose length is a compile-time constant (radix-8 plan, 1024 / 2048 / 4096 points:
// what ct_spectrum / ct_envelope / d4c_frame run); everything else takes the length at run time
bool fft_probe_has_static(int lgn, int max_lr) { return max_lr == 3 && lgn >= 10 && lgn <= 12; }

void launch_fft_probe(bool inverse, int lgn, int max_lr, int threads, bool static_plan, long batch, const void *d_in,
                      void *d_out, const Tables &tab, hipStream_t stream) {
  const size_t lds = fft_probe_lds_bytes(lgn);
  const dim3 grid((unsigned)batch);
  if (!inverse) {
    const double *in = static_cast<const double *>(d_in);
    double2 *out = static_cast<double2 *>(d_out);
#ifndef WORLD_EMU
    if (static_plan && lgn == 10) { WH_BLOCKS((fft_probe_rfft<3, 10>), grid, threads, lds, stream, in, out, lgn, tab.tw); return; }
    if (static_plan && lgn == 11) { WH_BLOCKS((fft_probe_rfft<3, 11>), grid, threads, lds, stream, in, out, lgn, tab.tw); return; }
    if (static_plan && lgn == 12) { WH_BLOCKS((fft_probe_rfft<3, 12>), grid, threads, lds, stream, in, out, lgn, tab.tw); return; }
#endif
    if (max_lr == 3) WH_BLOCKS(fft_probe_rfft<3>, grid, threads, lds, stream, in, out, lgn, tab.tw);
    else WH_BLOCKS(fft_probe_rfft<4>, grid, threads, lds, stream, in, out, lgn, tab.tw);
  } else {
    const double2 *in = static_cast<const double2 *>(d_in);
    double *out = static_cast<double *>(d_out);
#ifndef WORLD_EMU
    if (static_plan && lgn == 10) { WH_BLOCKS((fft_probe_irfft<3, 10>), grid, threads, lds, stream, in, out, lgn, tab.tw); return; }
    if (static_plan && lgn == 11) { WH_BLOCKS((fft_probe_irfft<3, 11>), grid, threads, lds, stream, in, out, lgn, tab.tw); return; }
    if (static_plan && lgn == 12) { WH_BLOCKS((fft_probe_irfft<3, 12>), grid, threads, lds, stream, in, out, lgn, tab.tw); return; }
#endif
    if (max_lr == 3) WH_BLOCKS(fft_probe_irfft<3>, grid, threads, lds, stream, in, out, lgn, tab.tw);
    else WH_BLOCKS(fft_probe_irfft<4>, grid, threads, lds, stream, in, out, lgn, tab.tw);
  }
}

}  // namespace world_hip
