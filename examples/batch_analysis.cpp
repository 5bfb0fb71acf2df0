// batch_analysis.cpp -- Part 2 of include/world_hip.h from plain C++ (no Python, no torch): two utterances of
// different length in one batch, inputs and outputs in HBM, one HIP stream.  Prints one line per utterance
// (frames, voiced frames, sum of F0, sum of log envelope, sum of aperiodicity) that tests/ compare with the
// same analysis made through the Python binding.
//
//   hipcc -O1 -I include examples/batch_analysis.cpp -L world_amd -lworld_hip -Wl,-rpath,$PWD/world_amd -o examples/batch_analysis
//   (python -m world_amd.build --examples does exactly this)
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "world_hip.h"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define WORLD_OK(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, world_hip_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
  const int fs = 48000, n_utt = 2;
  const int x_length[2] = {48000, 31000};
  const int x_stride = 48000;
  // a vowel-like test signal, quantised to 16 bits like a WAV file's samples
  std::vector<double> x((size_t)n_utt * x_stride, 0.0);
  for (int u = 0; u < n_utt; ++u)
    for (int i = 0; i < x_length[u]; ++i) {
      const double t = (double)i / fs, f0 = 120.0 + 60.0 * u + 20.0 * sin(2.0 * M_PI * 0.9 * t);
      double v = 0.0;
      for (int h = 1; h <= 12; ++h) v += sin(2.0 * M_PI * h * f0 * t + 0.3 * h) / h;
      x[(size_t)u * x_stride + i] = floor(0.2 * v * 32768.0 + 0.5) / 32768.0;
    }
  if (argc > 1) {                       // the samples, for whoever wants to repeat the analysis another way
    FILE *fp = fopen(argv[1], "wb");
    if (!fp || fwrite(x.data(), sizeof(double), x.size(), fp) != x.size()) { fprintf(stderr, "cannot write %s\n", argv[1]); return 4; }
    fclose(fp);
  }
  HarvestOption ho; InitializeHarvestOption(&ho);
  CheapTrickOption co; InitializeCheapTrickOption(fs, &co);
  D4COption dop; InitializeD4COption(&dop);
  int n_frames[2], f_stride = 0;
  for (int u = 0; u < n_utt; ++u) {
    n_frames[u] = GetSamplesForHarvest(fs, x_length[u], ho.frame_period);
    if (n_frames[u] > f_stride) f_stride = n_frames[u];
  }
  const int nb = co.fft_size / 2 + 1;

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  WorldHipContext *ctx = world_hip_create(0, stream);
  if (!ctx) { fprintf(stderr, "world_hip_create: %s\n", world_hip_last_error()); return 1; }
  double *d_x, *d_tpos, *d_f0, *d_sp, *d_ap;
  HIP_OK(hipMalloc((void **)&d_x, sizeof(double) * x.size()));
  HIP_OK(hipMalloc((void **)&d_tpos, sizeof(double) * n_utt * f_stride));
  HIP_OK(hipMalloc((void **)&d_f0, sizeof(double) * n_utt * f_stride));
  HIP_OK(hipMalloc((void **)&d_sp, sizeof(double) * (size_t)n_utt * f_stride * nb));
  HIP_OK(hipMalloc((void **)&d_ap, sizeof(double) * (size_t)n_utt * f_stride * nb));
  HIP_OK(hipMemcpyAsync(d_x, x.data(), sizeof(double) * x.size(), hipMemcpyHostToDevice, stream));

  WORLD_OK(world_hip_harvest_batch(ctx, n_utt, fs, d_x, x_stride, x_length, &ho, f_stride, d_tpos, d_f0));
  WORLD_OK(world_hip_cheaptrick_batch(ctx, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, &co, d_sp));
  WORLD_OK(world_hip_d4c_batch(ctx, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, co.fft_size, &dop, d_ap));

  std::vector<double> f0((size_t)n_utt * f_stride), sp((size_t)n_utt * f_stride * nb), ap(sp.size());
  HIP_OK(hipMemcpyAsync(f0.data(), d_f0, sizeof(double) * f0.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(sp.data(), d_sp, sizeof(double) * sp.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(ap.data(), d_ap, sizeof(double) * ap.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  for (int u = 0; u < n_utt; ++u) {
    int voiced = 0;
    double sum_f0 = 0.0, sum_log_sp = 0.0, sum_ap = 0.0;
    for (int f = 0; f < n_frames[u]; ++f) {
      const double v = f0[(size_t)u * f_stride + f];
      voiced += v > 0.0; sum_f0 += v;
      for (int k = 0; k < nb; ++k) {
        sum_log_sp += log(sp[((size_t)u * f_stride + f) * nb + k]);
        sum_ap += ap[((size_t)u * f_stride + f) * nb + k];
      }
    }
    printf("utterance %d: frames %d voiced %d sum_f0 %.9f sum_log_sp %.6f sum_ap %.6f\n", u, n_frames[u], voiced, sum_f0,
           sum_log_sp, sum_ap);
  }
  world_hip_destroy(ctx);
  for (double *d : {d_x, d_tpos, d_f0, d_sp, d_ap}) (void)hipFree(d);
  (void)hipStreamDestroy(stream);
  return 0;
}
