// codec.h -- parameter block of the envelope / aperiodicity coders (codec.hip).
// Reference: src/codec.cpp (CodeSpectralEnvelope, DecodeSpectralEnvelope,
// CodeAperiodicity, DecodeAperiodicity).  Rows are dense and independent, so a
// "batch" is simply a row count; every table below depends on (fs, fft_size) only.
#pragma once
#include "common.h"

namespace world_hip {

struct CodecParams {
  const double *in;       // [rows][in_cols]
  double *out;            // [rows][out_cols]
  size_t in_stride, out_stride;   // doubles between consecutive rows; 0 = dense (in_cols / out_cols).  Rows that live inside
                          // packed records (world_hip_analyze_coded: the coded wire format of the multi-GPU exchange) have the
                          // records' strides
  int rows;
  int fs, fft_size;
  int lg_md;              // log2(fft_size / 2): the DCT length ("max_dimension")
  int ndim;               // number_of_dimensions (envelope) / number_of_aperiodicities
  const int *knot;        // interp1 bin of every query (1-based upper knot), host-built
  const double *frac;     // interp1 weight of every query, host-built
  const double *w_re, *w_im;   // DCT / IDCT weights (codec.cpp:170-175, 193-197)
  Tables tab;
};

void launch_code_spectral_envelope(const CodecParams &p, hipStream_t stream);
void launch_decode_spectral_envelope(const CodecParams &p, hipStream_t stream);
void launch_code_aperiodicity(const CodecParams &p, hipStream_t stream);
void launch_decode_aperiodicity(const CodecParams &p, hipStream_t stream);

}  // namespace world_hip
