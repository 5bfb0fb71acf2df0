// exchange.hip -- the multi-GPU exchange step.  Utterances are sharded over GPUs and
// analysed independently; what remains is to hand every GPU's results to every other one.  A GPU's
// results are packed into ONE contiguous block of records
//     row = [ tpos, f0, sp[0 .. nb), ap[0 .. nb) ]        (2 + 2 nb doubles; 16 416 B at fft_size 2048)
// with the utterances' valid frames back to back (no padding to the longest utterance), so the
// exchange is one all-gather of one block per GPU: RCCL's (world_amd/distributed.py, one process per
// GPU) or, for ONE process driving several GPUs from C/C++, direct peer copies over xGMI
// (world_hip_allgather_blocks in api.hip: every destination pulls its n-1 remote blocks on its own
// stream, so all links of the fully connected mesh carry traffic at once).  The reference has no
// counterpart; this file is HBM-bound copying and nothing else.
#include "common.h"
#include "exchange.h"

namespace world_hip {

// one workgroup per (frame, utterance): its record is written once, coalesced
__global__ void ex_pack_rows(PackArgs a) {
  const int f = blockIdx.x, u = blockIdx.y;
  if (f >= a.n_frames[u]) return;
  const size_t fi = (size_t)u * a.f_stride + f;
  const int cols = 2 + 2 * a.nb;
  double *row = a.block + ((size_t)a.row_offset[u] + f) * cols;
  const double *sp = a.sp + fi * a.nb, *ap = a.ap + fi * a.nb;
  if (threadIdx.x == 0) { row[0] = a.tpos[fi]; row[1] = a.f0[fi]; }
  for (int i = threadIdx.x; i < a.nb; i += blockDim.x) { row[2 + i] = sp[i]; row[2 + a.nb + i] = ap[i]; }
}

// the inverse: records back into the padded [n_utt][f_stride][...] arrays of the batched API
__global__ void ex_unpack_rows(PackArgs a) {
  const int f = blockIdx.x, u = blockIdx.y;
  if (f >= a.n_frames[u]) return;
  const size_t fi = (size_t)u * a.f_stride + f;
  const int cols = 2 + 2 * a.nb;
  const double *row = a.block + ((size_t)a.row_offset[u] + f) * cols;
  double *sp = const_cast<double *>(a.sp) + fi * a.nb, *ap = const_cast<double *>(a.ap) + fi * a.nb;
  if (threadIdx.x == 0) { const_cast<double *>(a.tpos)[fi] = row[0]; const_cast<double *>(a.f0)[fi] = row[1]; }
  for (int i = threadIdx.x; i < a.nb; i += blockDim.x) { sp[i] = row[2 + i]; ap[i] = row[2 + a.nb + i]; }
}

void launch_pack_rows(const PackArgs &a, int max_frames, bool unpack, hipStream_t stream) {
  if (unpack) WH_BLOCKS(ex_unpack_rows, dim3(max_frames, a.n_utt), 256, 0, stream, a);
  else WH_BLOCKS(ex_pack_rows, dim3(max_frames, a.n_utt), 256, 0, stream, a);
}

}  // namespace world_hip
