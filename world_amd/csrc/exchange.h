// exchange.h -- parameter block of the result packing kernels (exchange.hip)
#pragma once
#include "devrt.h"

namespace world_hip {

struct PackArgs {
  int n_utt, f_stride, nb;
  const int *n_frames;      // [n_utt] valid frames (device)
  const int *row_offset;    // [n_utt] first record of each utterance within the block (device)
  const double *tpos, *f0;  // [n_utt][f_stride]
  const double *sp, *ap;    // [n_utt][f_stride][nb]
  double *block;            // [sum n_frames][2 + 2 nb]
};
void launch_pack_rows(const PackArgs &a, int max_frames, bool unpack, hipStream_t stream);

}  // namespace world_hip
