// simt_rt.cpp -- TEST INFRASTRUCTURE: the fibre scheduler behind tests/emu/simt_host.h (one workgroup at a time, every
// thread a fibre; block barriers and wavefront rendezvous as scheduler states).  Never part of libworld_hip.so.
#include "simt_host.h"

#include <sys/mman.h>
#include <vector>

// void simt_switch(void **save_sp, void *load_sp): park the running context (callee-saved registers on its own stack, the
// stack pointer in *save_sp) and resume the one whose stack pointer is load_sp.  System V x86-64.
extern "C" void simt_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size simt_switch,.-simt_switch
)");

namespace simt {
thread_local Block g_block;
namespace {
constexpr size_t kStack = size_t(256) << 10;          // per fibre; mapped lazily, touched pages only
struct Runner {
  std::vector<Fibre> fibres;
  std::vector<Wave> waves;
  std::vector<char *> stacks;                           // grow-only pool of this host thread
  void *sched_sp = nullptr;
  const std::function<void()> *body = nullptr;
  int alive = 0, at_barrier = 0;
  const char *name = "";
};
thread_local Runner R;

[[noreturn]] void fibre_main() {
  Fibre *f = g_block.cur;
  (*R.body)();
  f->state = 3;
  --R.alive;
  --R.waves[f->wave].alive;
  simt_switch(&f->sp, R.sched_sp);
  abort();                                              // a finished fibre is never resumed
}
void park(int state) {
  Fibre *f = g_block.cur;
  f->state = state;
  simt_switch(&f->sp, R.sched_sp);
}
char *stack_for(size_t i) {
  while (R.stacks.size() <= i) {
    void *p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("simt: mmap"); abort(); }
    R.stacks.push_back(static_cast<char *>(p));
  }
  return R.stacks[i];
}
void run_block(int threads) {
  const int nw = (threads + 63) / 64;
  R.fibres.assign(threads, Fibre());
  R.waves.assign(nw, Wave());
  R.alive = threads; R.at_barrier = 0;
  for (int t = 0; t < threads; ++t) {
    Fibre &f = R.fibres[t];
    f.tid = dim3((unsigned)t, 0, 0); f.lane = t & 63; f.wave = t >> 6; f.state = 0;
    f.stack = stack_for(t);
    // a fresh stack as simt_switch expects to find a parked one: six saved registers, then the address `ret` jumps to;
    // fibre_main starts with rsp = 8 mod 16, as after a call
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
    void **sp = reinterpret_cast<void **>(top);
    *--sp = nullptr;                                      // (where a caller's return address would be)
    *--sp = reinterpret_cast<void *>(&fibre_main);
    for (int k = 0; k < 6; ++k) *--sp = nullptr;
    f.sp = sp;
    ++R.waves[f.wave].alive;
  }
  for (;;) {
    bool ran = false;
    for (int t = 0; t < threads; ++t) {
      Fibre &f = R.fibres[t];
      if (f.state != 0) continue;
      ran = true;
      g_block.cur = &f;
      simt_switch(&R.sched_sp, f.sp);
      // back in the scheduler: what did it park for?
      if (f.state == 1) {
        ++R.at_barrier;
      } else if (f.state == 2) {
        Wave &w = R.waves[f.wave];
        ++w.arrived;
      }
      // a finished or parked fibre may have completed a barrier or a rendezvous
      if (R.alive > 0 && R.at_barrier == R.alive) {
        R.at_barrier = 0;
        for (Fibre &g : R.fibres) if (g.state == 1) g.state = 0;
      }
      Wave &w = R.waves[f.wave];
      if (w.alive > 0 && w.arrived == w.alive) {
        w.arrived = 0;
        ++w.seq;
        w.arrived_mask[w.seq & 1] = 0;                    // the NEXT operation's mask starts empty
        for (int l = 0; l < 64 && f.wave * 64 + l < threads; ++l) {
          Fibre &g = R.fibres[f.wave * 64 + l];
          if (g.state == 2) g.state = 0;
        }
      }
    }
    if (R.alive == 0) break;
    if (!ran) {
      int b = 0, v = 0;
      for (Fibre &g : R.fibres) { b += g.state == 1; v += g.state == 2; }
      fprintf(stderr, "simt: deadlock in %s, block (%u, %u): %d fibres alive, %d at the barrier, %d at a wavefront rendezvous "
                      "(a cross-lane operation or barrier under divergent control flow)\n", R.name, g_block.block_idx.x,
              g_block.block_idx.y, R.alive, b, v);
      abort();
    }
  }
  g_block.cur = nullptr;
}
}  // namespace

void barrier() { park(1); }

__attribute__((noinline)) const unsigned long long *exchange(unsigned long long mine, unsigned long long *mask) {
  Fibre *f = g_block.cur;
  Wave &w = R.waves[f->wave];
  const int slot = (int)(w.seq & 1);
  // Lock-step is only what the GPU does while the wavefront's lanes execute the SAME instruction: lanes that meet in one
  // rendezvous from different call sites (a cross-lane instruction under divergent control flow) would exchange operands
  // of unrelated instructions -- refuse instead of computing something (harvest.hip's 32-lane group searches do this)
  const void *site = __builtin_return_address(0);
  if (w.arrived_mask[slot] == 0) w.site[slot] = site;
  else if (w.site[slot] != site) {
    fprintf(stderr, "simt: lanes of one wavefront met at different cross-lane instructions in %s (divergent control flow is not "
                    "modelled)\n", R.name);
    abort();
  }
  w.buf[slot][f->lane] = mine;
  w.arrived_mask[slot] |= 1ull << f->lane;
  park(2);
  // resumed: every live lane has published into buf[slot]; the scheduler has moved seq on, so later operations use the other buffer
  *mask = w.arrived_mask[slot];
  return w.buf[slot];
}

void run_grid(dim3 grid, int threads, size_t lds_bytes, const std::function<void()> &body, const char *name) {
  if (threads <= 0 || threads > 1024) { fprintf(stderr, "simt: %d threads per block\n", threads); abort(); }
  std::vector<char> lds(lds_bytes + 64);
  const Block saved = g_block;                            // (a kernel never launches a kernel; be tidy all the same)
  R.body = &body; R.name = name;
  g_block.grid_dim = grid; g_block.block_dim = dim3((unsigned)threads, 1, 1);
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        memset(lds.data(), 0xA5, lds.size());             // a fresh workgroup inherits nothing it may rely on
        g_block.lds = lds.data();
        g_block.block_idx = dim3(x, y, z);
        run_block(threads);
      }
  g_block = saved;
  R.body = nullptr;
}
}  // namespace simt

// ---- self-test of the cross-lane shims (tests/test_emu.py): one workgroup of 256 fibres checks every emulated instruction
// against its definition; returns the number of mismatches -------------------------------------------------------------------
extern "C" int world_hip_simt_selftest(void) {
  static int bad;
  bad = 0;
  auto expect = [](bool ok) { if (!ok) ++bad; };
  simt::run_grid(dim3(2, 1, 1), 256, 4096, [&] {
    const int t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    int *lds = reinterpret_cast<int *>(simt::g_block.lds);
    const int v = 1000 * wv + lane;
    expect(__builtin_amdgcn_readlane(v, 17) == 1000 * wv + 17);
    expect(__builtin_amdgcn_readfirstlane(v) == 1000 * wv);
    expect(__builtin_amdgcn_ballot_w64((lane & 3) == 1) == 0x2222222222222222ull);
    expect(__shfl(v, lane ^ 5) == 1000 * wv + (lane ^ 5));
    expect(__shfl_xor(2.5 * v, 32) == 2.5 * (1000 * wv + (lane ^ 32)));
    expect(__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true) == 1000 * wv + (lane ^ 1));          // quad_perm [1,0,3,2]
    expect(__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true) == 1000 * wv + (lane ^ 2));          // quad_perm [2,3,0,1]
    expect(__builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true) == 1000 * wv + ((lane & ~7) | (7 - (lane & 7))));
    expect(__builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true) == 1000 * wv + ((lane & ~15) | (15 - (lane & 15))));
    expect(__builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true) == 1000 * wv + (lane ^ 8));         // row_ror:8
    expect(__builtin_amdgcn_update_dpp(-7, v, 0x112, 0xf, 0xf, true) == ((lane & 15) >= 2 ? v - 2 : 0));     // row_shr:2, bound_ctrl
    expect(__builtin_amdgcn_update_dpp(-7, v, 0x112, 0xf, 0xf, false) == ((lane & 15) >= 2 ? v - 2 : -7));   // ... keeps old
    expect(__builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0xf, true) == ((lane & 15) + 4 <= 15 ? v + 4 : 0)); // row_shl:4
    expect(__builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false) ==
           (((lane >> 4) == 1 || (lane >> 4) == 3) ? 1000 * wv + 16 * (lane >> 4) - 1 : -1));           // row_bcast15, rows 1 and 3
    expect(__builtin_amdgcn_update_dpp(-1, v, 0x143, 0xc, 0xf, false) == ((lane >> 4) >= 2 ? 1000 * wv + 31 : -1));
    // a block barrier really orders the LDS: everybody writes, everybody reads a neighbour's
    lds[t] = 7 * t + (int)blockIdx.x;
    __syncthreads();
    expect(lds[(t + 97) & 255] == 7 * ((t + 97) & 255) + (int)blockIdx.x);
    __syncthreads();
    // threads that leave early do not hold up the others' barriers or rendezvous
    if (lane >= 48) return;
    expect(__builtin_amdgcn_ballot_w64(true) == 0x0000ffffffffffffull);
    __syncthreads();
    expect(__builtin_amdgcn_readfirstlane(v) == 1000 * wv);
  }, "simt_selftest");
  return bad;
}
