// machine_probe.h -- see machine_probe.hip
#pragma once
#include "common.h"
namespace world_hip {
constexpr int kMachineProbeValues = 8;
// out[0] shader clock under an FP64 load (MHz)   out[1] FP64 FMA rate of that load (TFLOP/s)
// out[2] pointer chase through 2 GB (ns per hop)  out[3] through 64 MB just read by all CUs   out[4] through 1 MB just walked
// out[5] LDS round trip, idle CU (cycles)        out[6] LDS round trip, loaded CU    out[7] compute units seen
void run_machine_probe(double *out, hipStream_t stream);
}  // namespace world_hip
