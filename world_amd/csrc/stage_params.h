// stage_params.h -- parameter blocks of the per-frame spectral stages
// (cheaptrick.hip, d4c.hip) and their host launchers.
#pragma once
#include "common.h"

namespace world_hip {

struct CtParams {
  BatchView b;
  const double *tpos;    // [n_utt][f_stride]
  const double *f0;      // [n_utt][f_stride]
  double *spectrogram;   // [n_utt][f_stride][fft/2+1], or the base of packed records (out_row / out_stride / out_col_bytes)
  const int *out_row;    // nullptr: row of frame (u, f) = u f_stride + f; else out_row[u] + f (records back to back)
  size_t out_stride;     // doubles between consecutive rows (fft/2+1 for the dense layout)
  size_t out_col_bytes;  // bytes from a row's start to this stage's first value (0 for the dense layout)
  int out_f32;           // 1: the row's values are stored as float (the narrow wire format of the multi-GPU exchange)
  unsigned *offsets;     // [n_utt][f_stride] stream position of each frame's first draw
  const uint32_t *noise; // randn_value(noise[k]) = k-th randn() of the stream (context-wide table)
  Tables tab;
  double q1;
  double f0_floor;       // GetF0FloorForCheapTrick()
  int lg_fft;            // log2(fft_size)
  // Frame range of the frame kernels: rows of frames [frame_lo, frame_hi) only (the stream positions are those of the
  // whole utterance: ct_prepare always covers every frame).  0 / INT_MAX = all.
  int frame_lo, frame_hi;
  int skip_prepare;      // 1: the offsets of an earlier call with the same shape are still in the workspace
  // Coded records (world_hip_analyze_coded; SURVEY.md 8f.1: the coders "fuse onto K6/K8 outputs"): code_ndim > 0 = the frame
  // kernel writes CodeSpectralEnvelope's first code_ndim coefficients (codec.cpp:268-297) of ITS row instead of the row --
  // the values codec_code_sp would compute from the dense row, bit for bit -- at the row's place in the record.  The tables
  // are the stand-alone coder's (api.hip: codec_tables).
  int code_ndim;
  const int *code_knot;
  const double *code_frac, *code_w_re, *code_w_im;
};

struct D4cParams {
  BatchView b;
  const double *tpos;     // [n_utt][f_stride]
  const double *f0;       // [n_utt][f_stride]
  double *aperiodicity;   // [n_utt][f_stride][fft_out/2+1], or the base of packed records (out_row / out_stride / out_col_bytes)
  const int *out_row;     // as CtParams
  size_t out_stride;
  size_t out_col_bytes;
  int out_f32;
  double *rec;            // packed records' base (column 0 = tpos, 1 = f0), written by d4c_finish; nullptr: dense output
  double *ap0;            // [n_utt][f_stride]  LoveTrain result
  double *coarse;         // [n_utt][f_stride][16] coarse aperiodicity (dB) per band, slot 1 + band
  unsigned *offsets1;     // [n_utt][f_stride]  position of the LoveTrain window within pass 1
  unsigned *draws2;       // [n_utt][f_stride]  draws of the frame's 3 body windows in pass 2 (0: LoveTrain kept the frame out)
  unsigned *draws1;       // [n_utt] total draws of pass 1 (pass 2 continues the stream there)
  const uint32_t *noise;  // randn_value(noise[k]) = k-th randn() of the stream (context-wide table)
  const double *nuttall;  // [wl] NuttallWindow(wl), built on the host
  double *park_ws;        // the 16384-point shape only (96 kHz < fs <= 192 kHz): park_slots slots of N/2 + 2 doubles, one per
  int park_slots;         //   workgroup of a d4c_frame launch (the static group delay does not fit LDS beside that transform)
  Tables tab;
  double threshold;
  int fft_out;            // caller's fft_size (rows have fft_out/2+1 bins)
  int lg_love;            // log2 of the LoveTrain FFT
  int lg_d4c;             // log2 of fft_size_d4c
  int nap;                // number_of_aperiodicities
  int wl;                 // Nuttall window length
  const double *ap_frac;  // [fft_out/2+1] d4c_finish's interpolation onto the caller's bins: the weight s and the upper knot k of
  const int *ap_knot;     //   every bin (GetAperiodicity's interp1 over [0, 3 kHz .., fs/2], d4c.cpp:330-338, :373-375) depend on
                          //   (fs, fft_out) only -- host-built with the kernel's own expressions, cached in the context
  int code_nap;           // > 0: d4c_finish writes CodeAperiodicity's band values (codec.cpp:217-236) of its row instead of the row
  int band_center[8];     // static_cast<int>(3000 (band + 1) fft_size_d4c / fs), d4c.cpp:207-208: the centre bin of band b's slice
                          // (host arithmetic, the reference's expression: in the kernel it was a 20-instruction FP64 division per band)
  // Frame range of d4c_frame / d4c_finish (LoveTrain and the first offset scan always cover every frame: the second
  // pass's stream positions depend on every earlier frame's LoveTrain result).  0 / INT_MAX = all.
  int frame_lo, frame_hi;
  int skip_prepare;       // bits: kD4cSkipScan = offsets1 are in place (an earlier call, or the launch shared with CheapTrick);
                          // kD4cSkipLoveTrain = the LoveTrain pass (ap0, draws2) of an earlier call is still in the workspace
};
constexpr int kD4cSkipScan = 1, kD4cSkipLoveTrain = 2;

void launch_cheaptrick(const CtParams &p, int max_frames, hipStream_t stream);
void launch_d4c(const D4cParams &p, int max_frames, hipStream_t stream);
void launch_spectral_prepare(const CtParams &cp, const D4cParams &dp, hipStream_t stream);   // both stages' F0-only scans
size_t ct_max_draws_per_frame(int fft_size);
size_t d4c_max_draws_per_frame(int fs);
size_t d4c_park_slot_doubles(int lg_d4c);

}  // namespace world_hip
