// emu_rt.cpp -- runtime half of the host emulation of world_amd/csrc/devrt.h
// (TEST INFRASTRUCTURE: lets `pytest -m "not gpu"` run the kernels' logic on the
// CPU, one thread per block; never part of libworld_hip.so).
#include "devrt.h"

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local char *emu_lds_base = nullptr;

namespace devrt {
void *dmalloc(size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  memset(p, 0xA5, bytes);   // poison: kernels must not rely on zero-initialised workspace
  return p;
}
void dfree(void *p) { free(p); }
static thread_local char *lds_block = nullptr;
void emu_run_begin(size_t lds_bytes) {
  lds_block = static_cast<char *>(malloc(lds_bytes + 64));
  memset(lds_block, 0xA5, lds_bytes + 64);
  emu_lds_base = lds_block;
}
void emu_run_end() {
  free(lds_block);
  lds_block = nullptr;
  emu_lds_base = nullptr;
}
}  // namespace devrt
