// dio.h -- parameter blocks of the DIO and StoneMask kernels (dio.hip, stonemask.hip).
#pragma once
#include "common.h"

namespace world_hip {

struct DioParams {
  BatchView b;
  // DioOption (reference src/world/dio.h:16-23)
  double f0_floor, f0_ceil, frame_period, allowed_range;
  // derived on the host as in DioGeneralBody (dio.cpp:582-594) / FixF0Contour (:263-264)
  int ratio;                // decimation ratio = clamp(speed, 1, 12)
  double afs;               // fs / ratio
  int nb;                   // number_of_bands
  int cut;                  // matlab_round(afs / 50): half length of the low-cut filter
  int vrm;                  // voice_range_minimum
  int max_ntap;             // longest channel filter (4 * hal of channel 0)
  int nseg, ev_cap;
  int y_stride, z_stride, m_stride;
  const int *y_len;         // [n_utt] = 1 + x_len / ratio
  const double *band_f0;    // [nb]
  const int *band_hal;      // [nb] half_average_length
  const int *band_off;      // [nb]
  const double *band_taps;  // NuttallWindow(4*hal) per channel (dio.cpp:301)
  const double *lowcut_taps;  // centred low-cut FIR, 2*cut+1 taps (dio.cpp:40-53)
  const int *ref_fft;       // [n_utt] the reference's FFT length for this utterance (dio.cpp:592-594)
  double *nyq;              // [n_utt][4]: Y[N/2], Re/Im Y[N/2-1] of the low-cut filtered spectrum
  double *quirk;            // [n_utt][nb][4] per-channel constants of the mirror-store term
  // workspace
  double *fwd;              // decimation scratch
  double *y;                // [n_utt][y_stride]
  double *z;                // [n_utt][z_stride] low-cut filtered signal, stored from time -cut
  double *seg_events; int *seg_count;
  double *events; int *ev_count;
  double *cand, *score;     // [n_utt][nb][f_stride]
  double *t1, *t2;          // [n_utt][f_stride]
  // outputs
  double *tpos, *f0;        // [n_utt][f_stride]
};

struct StoneMaskParams {
  BatchView b;
  const double *tpos, *f0;  // [n_utt][f_stride]
  double *refined;          // [n_utt][f_stride]
  Tables tab;
  int win_cap;              // LDS doubles for the longest window
};

void launch_dio(const DioParams &p, int max_x_len, int max_y_len, int max_frames, hipStream_t stream);
void launch_stonemask(const StoneMaskParams &p, int max_frames, hipStream_t stream);
size_t stonemask_lds_bytes(int win_cap);      // dynamic LDS of sm_frame for a longest window of win_cap samples

}  // namespace world_hip
