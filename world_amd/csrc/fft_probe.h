// fft_probe.h -- launcher of the stand-alone transform kernels (fft_probe.hip)
#pragma once
#include "devrt.h"
#include "tables.h"

namespace world_hip {
bool fft_probe_has_static(int lgn, int max_lr);
void launch_fft_probe(bool inverse, int lgn, int max_lr, int threads, bool static_plan, long batch, const void *d_in,
                      void *d_out, const Tables &tab, hipStream_t stream);
}
