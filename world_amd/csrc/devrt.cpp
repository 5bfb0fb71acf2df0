// devrt.cpp -- HIP runtime plumbing for the product build (gfx950).
#include "devrt.h"

#include <stdexcept>
#include <string>
#include <vector>

namespace devrt {

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
void h2d(void *dst, const void *src, size_t n, hipStream_t s) {
  if (n) check(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s), "hipMemcpyAsync H2D");
}
void d2h(void *dst, const void *src, size_t n, hipStream_t s) {
  if (n) check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s), "hipMemcpyAsync D2H");
}
void d2d(void *dst, const void *src, size_t n, hipStream_t s) {
  if (n) check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync D2D");
}
void dzero(void *dst, size_t n, hipStream_t s) {
  if (n) check(hipMemsetAsync(dst, 0, n, s), "hipMemsetAsync");
}
void sync(hipStream_t s) { check(hipStreamSynchronize(s), "hipStreamSynchronize"); }
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
void event_record(void *ev, hipStream_t s) { check(hipEventRecord(static_cast<hipEvent_t>(ev), s), "hipEventRecord"); }
void event_sync(void *ev) { check(hipEventSynchronize(static_cast<hipEvent_t>(ev)), "hipEventSynchronize"); }


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
