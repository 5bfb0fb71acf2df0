// rng_fill.hip -- the reference's randn() stream (src/matlabfunctions.cpp:237-264)
// materialised in HBM by jump-ahead (see rng.h): randn_value(noise[k]) = k-th draw after
// reseed; plus the per-device table that all contexts of a process share, verified in full
// against a sequential host statement of the generator before anybody may read it.
#include "rng.h"

#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace world_hip {

__global__ void rng_stream_fill(RngFillArgs a) {
  const size_t start = a.begin + (size_t)blockIdx.y * ((size_t)gridDim.x * blockDim.x) * kFillRun +
                       (size_t)flat_thread_x() * kFillRun;
  if (start >= a.end) return;
  Xs128 s = xs_jump(a.jump, xs_seed(), (uint32_t)start);
  const int n = a.end - start < (size_t)kFillRun ? (int)(a.end - start) : kFillRun;
  for (int i = 0; i < n; ++i) a.noise[start + i] = xs_randn_word(s);
}

void launch_rng_fill(const RngFillArgs &a, hipStream_t stream) {
  if (a.end <= a.begin) return;
  const size_t runs = (a.end - a.begin + kFillRun - 1) / kFillRun;
  // x carries up to 2^20 runs, y the rest (a launch dimension is limited to 2^31-1 threads)
  const size_t per_y = (size_t)1 << 20;
  const unsigned ny = (unsigned)((runs + per_y - 1) / per_y);
  WH_THREADS(rng_stream_fill, (long)(runs < per_y ? runs : per_y), ny, 1, stream, a);
}

// ---------------------------------------------------------------------------
// Chunk sums: for chunk c (draws [c*kNoiseChunk, min(len, (c+1)*kNoiseChunk))):
//   sums[2c] = sum w[i],  sums[2c+1] = sum w[i] * (i_local + 1)      (mod 2^64)
// The weighted sum makes the pair sensitive to order and position, not just content.
constexpr int kSumThreads = 256;
__global__ void rng_chunk_sums(const uint32_t *noise, size_t len, unsigned long long *sums) {
  DYN_LDS(lds);
  unsigned long long *part = reinterpret_cast<unsigned long long *>(lds);   // [2][blockDim.x]
  const size_t chunk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const size_t lo = chunk * kNoiseChunk;
  if (lo >= len) return;
  const size_t n = len - lo < kNoiseChunk ? len - lo : kNoiseChunk;
  unsigned long long s1 = 0, s2 = 0;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned long long w = noise[lo + i];
    s1 += w;
    s2 += w * (unsigned long long)(i + 1);
  }
  part[threadIdx.x] = s1;
  part[blockDim.x + threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t1 = 0, t2 = 0;
    for (unsigned k = 0; k < blockDim.x; ++k) { t1 += part[k]; t2 += part[blockDim.x + k]; }
    sums[2 * chunk] = t1;
    sums[2 * chunk + 1] = t2;
  }
}

namespace {

struct HostGen {                                    // src/matlabfunctions.cpp:237-264, stepped one draw at a time
  uint32_t x = 123456789u, y = 362436069u, z = 521288629u, w = 88675123u;
  uint32_t word() {
    uint32_t acc = 0;
    for (int k = 0; k < 12; ++k) {
      const uint32_t t = x ^ (x << 11);
      x = y; y = z; z = w;
      w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
      acc += w >> 4;
    }
    return acc;
  }
};

struct DeviceNoise {
  int refs = 0;
  uint32_t *live = nullptr;
  size_t len = 0;
  std::vector<uint32_t *> superseded;
  size_t superseded_bytes = 0;
  // host statement: generator positioned at draw host_pos (always a whole number of chunks
  // until the 32-bit limit is reached), and every chunk's sums up to there
  HostGen gen;
  size_t host_pos = 0;
  std::vector<unsigned long long> host_sums;
  double build_ms = 0.0;                           // wall time spent building + verifying tables (all generations)
  int builds = 0;
};

// g_noise_lock guards the map and the reference counts only; a device's table is grown, verified and re-verified under
// ITS OWN lock, so a long host statement (12 ns per new draw, sequential) for one device never stalls the contexts of
// another.  Entries are heap objects: the map may rehash while a table is being built.
std::mutex g_noise_lock;
struct DeviceEntry { std::mutex lock; DeviceNoise d; };
std::map<int, std::unique_ptr<DeviceEntry>> g_noise;

DeviceEntry &entry_of(int device) {
  std::lock_guard<std::mutex> g(g_noise_lock);
  std::unique_ptr<DeviceEntry> &e = g_noise[device];
  if (!e) e.reset(new DeviceEntry);
  return *e;
}
DeviceEntry *find_entry(int device) {
  std::lock_guard<std::mutex> g(g_noise_lock);
  auto it = g_noise.find(device);
  return it == g_noise.end() ? nullptr : it->second.get();
}

// The host's jump tables (tables.cpp), built once per process: they only SEED the helper threads below, and every
// seed is then confirmed by sequential stepping, so the check stays independent of them.
const uint4 *host_jump_tables() {
  static const std::vector<uint4> *tab = [] {
    auto *v = new std::vector<uint4>((size_t)kJumpLevels * kJumpStride);
    build_jump_tables(v->data());
    return v;
  }();
  return tab->data();
}
HostGen host_jump(size_t calls) {                    // the generator after `calls` draws from the seed, by the tables
  const uint4 *jump = host_jump_tables();
  HostGen g;
  uint32_t in[4] = {g.x, g.y, g.z, g.w};
  for (int level = 0; calls != 0; ++level, calls >>= 1) {
    if (!(calls & 1u)) continue;
    const uint4 *tab = jump + (size_t)level * kJumpStride;
    uint32_t out[4] = {0, 0, 0, 0};
    for (int wd = 0; wd < 4; ++wd)
      for (int n = 0; n < 8; ++n) {
        const uint4 e = tab[(wd * 8 + n) * 16 + ((in[wd] >> (4 * n)) & 15u)];
        out[0] ^= e.x; out[1] ^= e.y; out[2] ^= e.z; out[3] ^= e.w;
      }
    for (int k = 0; k < 4; ++k) in[k] = out[k];
  }
  g.x = in[0]; g.y = in[1]; g.z = in[2]; g.w = in[3];
  return g;
}

// sums of the chunks that cover draws [pos, upto) (pos a multiple of kNoiseChunk), stepping `g` sequentially
void host_chunks(HostGen &g, size_t pos, size_t upto, unsigned long long *sums) {
  while (pos < upto) {
    const size_t n = upto - pos < kNoiseChunk ? upto - pos : kNoiseChunk;
    unsigned long long s1 = 0, s2 = 0;
    for (size_t i = 0; i < n; ++i) {
      const unsigned long long w = g.word();
      s1 += w;
      s2 += w * (unsigned long long)(i + 1);
    }
    *sums++ = s1;
    *sums++ = s2;
    pos += n;
  }
}

// The host statement of the stream up to draw `upto`.  Stepping is sequential by nature (12 ns per draw: 0.4 s for the
// 32 M draws of a 10 s utterance -- round 3's whole cold start, 500 x a steady-state call), so the new draws are cut into
// T contiguous ranges stepped side by side: range 0 continues the generator where the last statement stopped, range k is
// SEEDED by the jump tables -- and then every seed is confirmed, not trusted: range k - 1's generator, stepped
// sequentially to its end, must arrive at exactly the state range k started from.  By induction from the true seed the
// whole statement is the sequential generator's; the jump tables cannot make a wrong table pass.
void host_extend(DeviceNoise &d, size_t upto) {
  if (d.host_pos >= upto) return;
  const size_t first_chunk = d.host_pos / kNoiseChunk;
  const size_t chunks = (upto - d.host_pos + kNoiseChunk - 1) / kNoiseChunk;
  d.host_sums.resize(2 * (first_chunk + chunks));
  static const int max_threads = [] {
    const char *e = getenv("WORLD_HIP_TABLE_THREADS");
    const int hw = (int)std::thread::hardware_concurrency();
    return e ? std::max(1, atoi(e)) : std::max(1, std::min(16, hw > 0 ? hw : 1));
  }();
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)max_threads, chunks / 4));   // >= 4 chunks (3 ms) a thread
  if (T == 1 || d.host_pos % kNoiseChunk != 0) {
    host_chunks(d.gen, d.host_pos, upto, d.host_sums.data() + 2 * first_chunk);
    d.host_pos = upto;
    return;
  }
  std::vector<size_t> lo(T + 1);
  for (int k = 0; k <= T; ++k) lo[k] = d.host_pos + (chunks * k / T) * kNoiseChunk;
  lo[T] = upto;
  std::vector<HostGen> start(T), end(T);
  start[0] = d.gen;
  for (int k = 1; k < T; ++k) start[k] = host_jump(lo[k]);
  auto work = [&](int k) {
    HostGen g = start[k];
    host_chunks(g, lo[k], lo[k + 1], d.host_sums.data() + 2 * (lo[k] / kNoiseChunk));
    end[k] = g;
  };
  std::vector<std::thread> threads;
  int spawned = 0;
  for (int k = 1; k < T; ++k) {
    try { threads.emplace_back(work, k); ++spawned; } catch (...) { break; }
  }
  work(0);
  for (std::thread &t : threads) t.join();
  for (int k = spawned + 1; k < T; ++k) work(k);     // threads that could not be started: their ranges on this one
  for (int k = 0; k + 1 < T; ++k)
    if (end[k].x != start[k + 1].x || end[k].y != start[k + 1].y || end[k].z != start[k + 1].z || end[k].w != start[k + 1].w)
      throw std::runtime_error("randn host statement: the jump-table seed of a verification range disagrees with the sequential generator");
  d.gen = end[T - 1];
  d.host_pos = upto;
}

// reduce `table[0, len)` on the device and compare every chunk with the host's sums
void check_table(DeviceNoise &d, const uint32_t *table, size_t len, hipStream_t stream, const char *when) {
  const size_t chunks = (len + kNoiseChunk - 1) / kNoiseChunk;
  unsigned long long *d_sums = static_cast<unsigned long long *>(devrt::dmalloc(sizeof(unsigned long long) * 2 * chunks));
  std::vector<unsigned long long> got(2 * chunks);
  try {
    const unsigned gx = (unsigned)(chunks < 32768 ? chunks : 32768), gy = (unsigned)((chunks + gx - 1) / gx);
    WH_BLOCKS(rng_chunk_sums, dim3(gx, gy), kSumThreads, 2 * kSumThreads * sizeof(unsigned long long), stream, table, len,
              d_sums);
    host_extend(d, len);                              // overlaps the kernels in flight
    devrt::d2h(got.data(), d_sums, sizeof(unsigned long long) * 2 * chunks, stream);
    devrt::sync(stream);
  } catch (...) {
    devrt::dfree(d_sums);
    throw;
  }
  devrt::dfree(d_sums);
  for (size_t c = 0; c < chunks; ++c)
    if (got[2 * c] != d.host_sums[2 * c] || got[2 * c + 1] != d.host_sums[2 * c + 1]) {
      char msg[256];
      snprintf(msg, sizeof msg,
               "randn table verification failed (%s): chunk %zu of %zu (draws %zu..) sums %llx/%llx, expected %llx/%llx", when, c,
               chunks, c * kNoiseChunk, got[2 * c], got[2 * c + 1], d.host_sums[2 * c], d.host_sums[2 * c + 1]);
      throw std::runtime_error(msg);
    }
}

}  // namespace

void noise_table_retain(int device) {
  DeviceEntry &e = entry_of(device);
  std::lock_guard<std::mutex> g(e.lock);
  e.d.refs += 1;
}

void noise_table_release(int device) {
  DeviceEntry *e = find_entry(device);
  if (!e) return;
  std::lock_guard<std::mutex> g(e->lock);
  DeviceNoise &d = e->d;
  if (--d.refs > 0) return;
  // the device's last context is gone (it drained its stream before releasing): nobody reads these any more.  The entry
  // itself stays (an empty table costs nothing and a concurrent retain may already hold its address).
  if (d.live) devrt::dfree(d.live);
  for (uint32_t *p : d.superseded) devrt::dfree(p);
  d = DeviceNoise();
}

const uint32_t *noise_table_acquire(int device, size_t draws, const uint4 *d_jump, hipStream_t stream) {
  if (draws > kNoiseMaxDraws) throw std::runtime_error("utterance consumes more than 2^32 randn() draws");
  DeviceEntry &e = entry_of(device);
  std::lock_guard<std::mutex> g(e.lock);
  DeviceNoise &d = e.d;
  if (draws <= d.len) return d.live;
  // grow: at least double, whole chunks (only the final 32-bit clamp leaves a partial chunk, and nothing grows past it)
  size_t cap = draws > 2 * d.len ? draws : 2 * d.len;
  if (cap < ((size_t)1 << 22)) cap = (size_t)1 << 22;
  cap = (cap + kNoiseChunk - 1) / kNoiseChunk * kNoiseChunk;
  if (cap > kNoiseMaxDraws) cap = kNoiseMaxDraws;
  const auto t_build = std::chrono::steady_clock::now();
  uint32_t *fresh = static_cast<uint32_t *>(devrt::dmalloc(sizeof(uint32_t) * cap));
  try {
    // every word comes from the fill kernel (nothing is copied over from the shorter table) ...
    RngFillArgs fill = {fresh, 0, cap, d_jump};
    launch_rng_fill(fill, stream);
    // ... and every word is accounted for before the table is published (the host statement only steps the draws beyond
    // those it has already summed: growth costs the host the NEW draws, not the whole stream)
    check_table(d, fresh, cap, stream, "new table");
  } catch (...) {
    try { devrt::sync(stream); } catch (...) {}      // a failing sync must not leak the table
    try { devrt::dfree(fresh); } catch (...) {}
    throw;
  }
  if (d.live) {
    d.superseded.push_back(d.live);
    d.superseded_bytes += sizeof(uint32_t) * d.len;
  }
  d.live = fresh;
  d.len = cap;
  d.build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
  d.builds += 1;
  return d.live;
}

void noise_table_verify(int device, hipStream_t stream) {
  DeviceEntry *e = find_entry(device);
  if (!e) return;
  std::lock_guard<std::mutex> g(e->lock);
  if (!e->d.live) return;
  check_table(e->d, e->d.live, e->d.len, stream, "re-check of the live table");
}

double noise_table_build_ms(int device, int *builds) {
  DeviceEntry *e = find_entry(device);
  if (builds) *builds = 0;
  if (!e) return 0.0;
  std::lock_guard<std::mutex> g(e->lock);
  if (builds) *builds = e->d.builds;
  return e->d.build_ms;
}

size_t noise_table_bytes(int device) {
  DeviceEntry *e = find_entry(device);
  if (!e) return 0;
  std::lock_guard<std::mutex> g(e->lock);
  return sizeof(uint32_t) * e->d.len + e->d.superseded_bytes;
}

}  // namespace world_hip
