// devrt.cpp -- HIP runtime plumbing for the product build (gfx950).
#include "devrt.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace devrt {

// ---- host-side cost of every runtime call (WORLD_HIP_HOST_TRACE=1): printed at exit ----
bool g_host_trace = getenv("WORLD_HIP_HOST_TRACE") != nullptr;
namespace {
struct HostTrace {
  std::map<std::string, std::pair<double, long>> acc;
  ~HostTrace() {
    if (!g_host_trace) return;
    double total = 0;
    for (auto &kv : acc) total += kv.second.first;
    fprintf(stderr, "[world_hip host trace] total %.1f us in %zu call sites\n", total, acc.size());
    for (auto &kv : acc)
      fprintf(stderr, "  %-28s %8ld calls %10.1f us  %7.2f us/call\n", kv.first.c_str(), kv.second.second, kv.second.first,
              kv.second.first / kv.second.second);
  }
} g_trace;
struct Timed {
  const char *name;
  std::chrono::steady_clock::time_point t0;
  explicit Timed(const char *n) : name(n) { if (g_host_trace) t0 = std::chrono::steady_clock::now(); }
  ~Timed() {
    if (g_host_trace) host_trace_add(name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
};
}  // namespace
void host_trace_add(const char *name, double us) {
  static std::mutex lock;                         // drop-in calls come from several host threads
  std::lock_guard<std::mutex> g(lock);
  auto &e = g_trace.acc[name];
  e.first += us; e.second += 1;
}

void check(hipError_t e, const char *what) {
  if (e != hipSuccess)
    throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
void *dmalloc(size_t bytes) {
  void *p = nullptr;
  check(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc");
  return p;
}
void dfree(void *p) { if (p) check(hipFree(p), "hipFree"); }
void h2d(void *dst, const void *src, size_t n, hipStream_t s) { Timed t_("api:h2d");
  if (n) check(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s), "hipMemcpyAsync H2D");
}
void d2h(void *dst, const void *src, size_t n, hipStream_t s) { Timed t_("api:d2h");
  if (n) check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s), "hipMemcpyAsync D2H");
}
void d2d(void *dst, const void *src, size_t n, hipStream_t s) {
  if (n) check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync D2D");
}
void dzero(void *dst, size_t n, hipStream_t s) {
  if (n) check(hipMemsetAsync(dst, 0, n, s), "hipMemsetAsync");
}
void sync(hipStream_t s) { Timed t_("api:stream_sync"); check(hipStreamSynchronize(s), "hipStreamSynchronize"); }
int current_device() {
  int device = 0;
  check(hipGetDevice(&device), "hipGetDevice");
  return device;
}
void set_device(int device) {
  int count = 0;
  check(hipGetDeviceCount(&count), "hipGetDeviceCount");
  if (device < 0 || device >= count) throw std::runtime_error("no such HIP device");
  check(hipSetDevice(device), "hipSetDevice");
}
void *hmalloc_pinned(size_t n) {
  void *p = nullptr;
  check(hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault), "hipHostMalloc");
  return p;
}
void hfree_pinned(void *p) { if (p) check(hipHostFree(p), "hipHostFree"); }
void *event_create() {
  hipEvent_t ev;
  check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
  return ev;
}
void event_destroy(void *ev) { if (ev) check(hipEventDestroy(static_cast<hipEvent_t>(ev)), "hipEventDestroy"); }
void event_record(void *ev, hipStream_t s) { Timed t_("api:event_record"); check(hipEventRecord(static_cast<hipEvent_t>(ev), s), "hipEventRecord"); }
void event_sync(void *ev) { Timed t_("api:event_sync"); check(hipEventSynchronize(static_cast<hipEvent_t>(ev)), "hipEventSynchronize"); }
void stream_wait_event(hipStream_t s, void *ev) { check(hipStreamWaitEvent(s, static_cast<hipEvent_t>(ev), 0), "hipStreamWaitEvent"); }
hipStream_t stream_create() {
  hipStream_t s = nullptr;
  check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  return s;
}
void stream_destroy(hipStream_t s) { if (s) check(hipStreamDestroy(s), "hipStreamDestroy"); }
bool is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
void graph_begin(hipStream_t s) { check(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture"); }
void *graph_end(hipStream_t s) {
  hipGraph_t g = nullptr;
  check(hipStreamEndCapture(s, &g), "hipStreamEndCapture");
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  check(e, "hipGraphInstantiate");
  return exec;
}
void graph_launch(void *exec, hipStream_t s) { check(hipGraphLaunch(static_cast<hipGraphExec_t>(exec), s), "hipGraphLaunch"); }
void graph_destroy(void *exec) { if (exec) check(hipGraphExecDestroy(static_cast<hipGraphExec_t>(exec)), "hipGraphExecDestroy"); }
void peer_copy(void *dst, int dst_device, const void *src, int src_device, size_t n, hipStream_t s) {
  if (!n) return;
  if (dst_device == src_device) check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync D2D");
  else check(hipMemcpyPeerAsync(dst, dst_device, src, src_device, n, s), "hipMemcpyPeerAsync");
}
void enable_peer_access(int device, int peer) {
  if (device == peer) return;
  // once per pair and process: the query, the device switch and the enable are host calls that a path advertised as
  // host-wait-free should not repeat on every exchange.  A pair is recorded only after a DEFINITIVE answer
  // (enabled, already enabled, or the hardware says no); a transient failure is retried by the next exchange instead of
  // condemning the pair to staged copies for the life of the process, and the lock is held across the enable so that a
  // second thread cannot run ahead of a pair that is still being enabled.
  static std::mutex lock;
  static std::map<std::pair<int, int>, bool> done;
  std::lock_guard<std::mutex> g(lock);
  if (done.count({device, peer})) return;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess) { (void)hipGetLastError(); return; }   // transient: ask again
  if (!can) { done[{device, peer}] = false; return; }
  int before = 0;
  check(hipGetDevice(&before), "hipGetDevice");
  check(hipSetDevice(device), "hipSetDevice");
  const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
  if (e != hipSuccess) (void)hipGetLastError();
  if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) done[{device, peer}] = true;
  check(hipSetDevice(before), "hipSetDevice");
}


void allow_large_lds(const void *kernel, size_t lds, const char *name) {
  // the attribute belongs to the function ON THE CURRENT DEVICE: a process that drives several GPUs
  // (one context each) must be granted it once per device
  static std::mutex lock;
  static std::map<std::pair<int, const void *>, size_t> granted;
  int device = 0;
  check(hipGetDevice(&device), "hipGetDevice");
  std::lock_guard<std::mutex> g(lock);
  size_t &have = granted[std::make_pair(device, kernel)];
  if (lds <= have) return;
  Timed t_("api:func_set_attribute");
  check(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), name);
  have = lds;
}

// ---- per-kernel event timing ---------------------------------------------------
bool g_profiling = false;
namespace {
struct Rec { const char *name; hipEvent_t a, b; };
std::vector<Rec> g_recs;
}
void prof_begin(const char *name, hipStream_t s) {
  Rec r;
  r.name = name;
  check(hipEventCreate(&r.a), "hipEventCreate");
  check(hipEventCreate(&r.b), "hipEventCreate");
  check(hipEventRecord(r.a, s), "hipEventRecord");
  g_recs.push_back(r);
}
void prof_end(hipStream_t s) { check(hipEventRecord(g_recs.back().b, s), "hipEventRecord"); }
void prof_enable(bool on) { g_profiling = on; }
// resolve all pending records: appends "name ms\n" lines
std::string prof_collect() {
  std::string out;
  for (Rec &r : g_recs) {
    float ms = 0.f;
    check(hipEventSynchronize(r.b), "hipEventSynchronize");
    check(hipEventElapsedTime(&ms, r.a, r.b), "hipEventElapsedTime");
    char line[160];
    snprintf(line, sizeof line, "%s %.6f\n", r.name, ms);
    out += line;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  return out;
}

}  // namespace devrt
