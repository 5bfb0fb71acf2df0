// harvest.h -- parameter block shared by the Harvest kernels (harvest.hip,
// harvest_contour.hip) and the host planner (pipeline.cpp).
#pragma once
#include "common.h"

namespace world_hip {

struct HarvestParams {
  BatchView b;
  // ---- options (HarvestOption, reference src/world/harvest.h:16-20) ----
  double f0_floor, f0_ceil, frame_period;
  // ---- derived on the host exactly as HarvestGeneralBody does (harvest.cpp:1145-1165) ----
  int ratio;               // decimation ratio = matlab_round(fs / 8000)
  double afs;              // fs / ratio
  int nch;                 // number of band-pass channels
  int maxc;                // max_candidates = matlab_round(nch / 10) * 7
  int lag;                 // edge padding before decimation (harvest.cpp:50-51)
  int y_stride;            // decimated samples per utterance slot
  int fb_stride;           // 1 ms ("basic") frames per utterance slot
  int ev_cap;              // capacity of one event list
  int refine_cap;          // LDS doubles per refinement window
  int sec_cap;             // voiced sections per utterance slot
  int lone_job;            // host only: this job has the device to itself (one utterance, no WORLD_HIP_HINT_SHARED_DEVICE):
                           // the one-workgroup-per-utterance contour kernels take 1024 threads instead of 256
  int ext_cap;             // doubles of extended-section storage per utterance
  const int *y_len;        // [n_utt] decimated length = ceil(x_len / ratio)
  const int *nfb;          // [n_utt] basic frame count
  const double *band_f0;   // [nch]  boundary_f0_list
  const int *band_half;    // [nch]  filter half length L
  const int *band_off;     // [nch]  offset of the band's taps in band_taps
  const double *band_taps; // Nuttall * cos band-pass FIRs (harvest.cpp:101-108), built on the host
  // FFT path of the filter bank (harvest.hip: hv_block_spectra / hv_band_events_fft); fft_seg == 0: direct FIR
  const double2 *band_spec;// [nch][kBandFftBins] spectrum of every band's taps / kBandFft, built once per band set
  double2 *blk_spec;       // [n_utt][nblk][kBandFftBins] spectra of the signal's overlapping blocks (workspace)
  int fft_seg;             // filtered samples one block yields: kBandFft - 2 max_half - 2
  int fft_pre;             // samples a block starts before its first output: max_half - 1
  int max_half;            // max L
  const int *ref_fft;      // [n_utt] the reference's FFT length for this utterance (harvest.cpp:1164-1165)
  double *nyq;             // [n_utt][nyq_slices][4]: partial sums of Y[N/2], Re/Im Y[N/2-1] (mean-free signal), 2/N
  int nyq_slices;          // slices of 4096 samples per utterance slot
  double *mean_part;       // [n_utt][mean_parts] partial sums of the decimated signal: per span of the backward sweep (ratio 1: per slice)
  int mean_parts;
  double *quirk;           // [n_utt][nch][4] per-band constants of the mirror-store term (bandfilter.h)
  const double *win_tab;   // [hw][6] = sin/cos(pi d), sin/cos(pi WAVE d), 2 / window length, pi d; d = 2/(2hw+1): refinement window steps
  const double *win_lane;  // [hw][WAVE][2] = (sin, cos)(pi (lane - hw - 1) d): a lane's first window sample at a whole-sample frame centre
  const double2 *win_full; // [hw^2 + i] = (main window, its central difference)[i] of half length hw centred on a sample; nullptr:
                           // a millisecond is not a whole number of samples at the analysis rate -- the kernel rotates (win_tab / win_lane)
  Tables tab;
  // ---- workspace (device) ----
  double *fwd;             // [n_utt][m_stride] forward-filtered padded signal
  int m_stride;
  double *y;               // [n_utt][y_stride]
  double *seg_events;      // [ev_group][nch][4][nseg][seg_cap] per-segment crossing times (nseg == 1 on the FFT path: the final lists)
  int *seg_count;          // [ev_group][nch][4][nseg]
  int nseg;                // segment lists per (utterance, band, family): chunks of blocks (FFT path) or FIR segments
  int seg_cap;             // capacity of one segment list
  int nblk;                // FFT path: blocks per utterance slot
  int chunk_blocks;        // FFT path: consecutive blocks one workgroup filters (its events form one segment list)
  double *events;          // [ev_group][nch][4][ev_cap] fine zero-crossing positions of ONE GROUP of utterances (launch_harvest walks the groups)
  int *ev_count;           // [ev_group][nch][4]
  int ev_group;            // utterances whose lists exist at once
  int ev_u0;               // first utterance of the group a launch works on (its lists are slot u - ev_u0)
  double *raw;             // [n_utt][nch][fb_stride]
  double *cand_a, *score_a;  // [n_utt][fb_stride][maxc]
  double *cand_b, *score_b;  // [n_utt][fb_stride][maxc]
  int *nc;                 // [n_utt] candidates per frame (max over frames)
  double *c0, *c2, *c3;      // [n_utt][fb_stride(+600)] contour scratch: base pick / step 4, step 2, step 3
  int *sec;                // [n_utt][6][sec_cap]: start, end, ext start, ext end, slice offset, slice origin
  int *sec_n;              // [n_utt][2]: number of sections, number kept by ExtendSub
  double *sec_sum;         // [n_utt][sec_cap] sum of a section's extended f0
  double *ext;             // [n_utt][ext_cap] extended f0 of every section (step 3)
  double *basic_f0;        // [n_utt][fb_stride] result at 1 ms hop
  // ---- outputs ----
  double *tpos;            // [n_utt][f_stride]
  double *f0;              // [n_utt][f_stride]
};

constexpr int kBandFftLg = 12, kBandFft = 1 << kBandFftLg, kBandFftBins = kBandFft / 2 + 1;
int hv_fft_segment(int max_half);          // outputs per block of the FFT path, 0 = filters too long for it
void launch_band_spectra(const double *d_taps, const int *d_off, const int *d_half, int nch, double2 *d_spec,
                         const Tables &tab, hipStream_t stream);
void launch_harvest(const HarvestParams &p, int max_x_len, int max_y_len, int max_fb, int max_frames,
                    hipStream_t stream);
void launch_harvest_contour(const HarvestParams &p, int max_fb, int max_frames, hipStream_t stream);

}  // namespace world_hip
