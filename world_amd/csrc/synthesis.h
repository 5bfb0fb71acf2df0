// synthesis.h -- parameter block of the waveform synthesiser (synthesis.hip).
// Reference: Synthesis(), src/synthesis.cpp:339-399.
#pragma once
#include "common.h"

namespace world_hip {

constexpr int kSyThreads = 256;                  // (round 6: one spelling -- tests/emu runs this unit's real workgroups, simt_host.h)
constexpr int kSyPer = 8;                       // consecutive samples per thread in the time-base kernels
constexpr int kSyTile = kSyThreads * kSyPer;    // samples per workgroup

struct SynthParams {
  int n_utt, fs, fft_size, lg_fft;
  double frame_period;      // seconds (the reference divides by 1000 on entry, synthesis.cpp:360,366)
  double lowest_f0;         // fs / fft_size + 1.0 with the reference's integer division (synthesis.cpp:361)
  const double *f0;         // [n_utt][f_stride]
  const double *sp, *ap;    // [n_utt][f_stride][fft_size/2+1]
  int f_stride;
  const int *n_frames;      // [n_utt] (device)
  const int *y_len;         // [n_utt] (device)
  double *y;                // [n_utt][y_stride]
  int y_stride;
  // ---- workspace ----
  double *inc;              // [n_utt][y_stride] phase increments, then (in place) the running phase
  unsigned char *flags;     // [n_utt][y_stride] bit0 = interpolated vuv, bit1 = a pulse sits at this sample
  int *blk_cnt;             // [n_utt][nblk] pulses of each tile
  int nblk;
  int *pidx;                // [n_utt][pulse_cap] pulse_locations_index
  double *pshift;           // [n_utt][pulse_cap] pulse_locations_time_shift
  int *np;                  // [n_utt] number_of_pulses (clamped to pulse_cap)
  int pulse_cap;
  int *need;                // context-wide: largest pulse count that did NOT fit pulse_cap (0 = none dropped so far)
  double *resp;             // [n_utt][pulse_cap][resp_stride] impulse response of every pulse (fft_size values)
  int resp_stride;          // fft_size; fft_size + 2 for the 8192-point shape, whose pulse keeps its spectrum there (sy_pulse)
  const double *dc_remover; // [fft_size] GetDCRemover(), host-built
  const uint32_t *noise;    // randn_value(noise[k]) = k-th randn() after reseed
  Tables tab;
};

void launch_synthesis(const SynthParams &p, int max_y, hipStream_t stream);
size_t synth_pulse_lds_bytes(int lg_fft);
int synth_tile_samples();                        // samples one workgroup of the time-base kernels covers (the unit's own constant:
                                                 // callers size their per-tile arrays by asking, not by including it)

}  // namespace world_hip
