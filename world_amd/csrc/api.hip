// api.hip -- host planner + the C ABI of libworld_hip.so (include/world_hip.h).
//
// The planner mirrors the scalar set-up code at the top of the reference's entry
// points (Harvest()/HarvestGeneralBody src/harvest.cpp:1145-1165,1223-1244,
// CheapTrick() src/cheaptrick.cpp:191-214, D4C() src/d4c.cpp:350-379), carves a
// per-call workspace out of one grow-only HBM arena and enqueues the stage
// kernels on the context's stream.  Nothing here waits for the GPU except the
// drop-in host-pointer entry points, which must hand results back in caller memory.
#include "../../include/world_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdarg>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.h"
#include "bandfilter.h"
#include "codec.h"
#include "decimate.h"
#include "dio.h"
#include "exchange.h"
#include "fft_probe.h"
#include "machine_probe.h"
#include "harvest.h"
#include "stage_params.h"
#include "synthesis.h"

namespace world_hip {

static thread_local std::string g_last_error;

[[noreturn]] static void fail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw std::runtime_error(buf);
}

// ---------------------------------------------------------------------------
struct Arena {
  char *base = nullptr;
  size_t cap = 0, used = 0;
  void reset() { used = 0; }
  void skip_to(size_t offset) { used = (offset + 255) & ~size_t(255); }
  template <class T> T *take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    if (used + bytes > cap) fail("workspace arena overflow (%zu + %zu > %zu)", used, bytes, cap);
    T *p = reinterpret_cast<T *>(base + used);
    used += bytes;
    return p;
  }
};

struct HarvestBands {            // cached per (fs, f0_floor, f0_ceil)
  int fs = 0;
  double f0_floor = 0, f0_ceil = 0;
  int nch = 0, max_half = 0;
  double *d_band_f0 = nullptr, *d_taps = nullptr;
  int *d_half = nullptr, *d_off = nullptr;
  double *d_win_tab = nullptr;   // refinement-window angle steps per half length (hv_refine)
  int win_tab_len = 0;
  size_t win_full_entries = 0;   // (w, dw) pairs of every whole-sample-centred window behind the two tables above (0: none)
  double2 *d_spec = nullptr;     // [nch][kBandFftBins] spectra of the taps / kBandFft (FFT path of the filter bank)
  int fft_seg = 0;               // outputs per block of that path (0: filters too long, direct FIR)
  std::vector<double> band_f0;
};

}  // namespace world_hip

// The small per-call host arrays (lengths, frame counts, row offsets: a few ints per utterance) depend only on the
// call's shape, and a service sees the same shapes again and again: every distinct array is uploaded ONCE into a
// persistent slab and found again by content.  A call in steady state therefore does no host-to-device copy, touches no
// staging buffer and waits for no event -- which is also what makes the batched calls capturable into a HIP graph
// (round 2 staged them through a ring of pinned buffers guarded by events; hipErrorStreamCaptureInvalidated).
// Slabs are CHAINED, never replaced under a call: a stage uploads several arrays before it launches, and a pointer it
// already holds must stay valid whatever the later uploads do (a full 4 MB slab used to be freed and
// reallocated between two uploads of one call).  The whole set is dropped only at the START of a stage (CallScope), with
// the stream drained, once it holds more than kSmallBudget bytes -- a service fed ever-new length vectors then pays one
// synchronisation per ~16 k distinct arrays instead of growing without bound.  Lookup is a hash of the content.
struct SmallArrays {
  struct Entry { std::vector<char> bytes; char *dev; };
  struct Slab { char *base; size_t cap, used; };
  std::vector<Slab> slabs;
  std::unordered_multimap<uint64_t, Entry> entries;
  size_t held = 0;               // bytes of all slabs
  static uint64_t hash(const void *data, size_t bytes) {          // FNV-1a over the content and its length
    uint64_t h = 1469598103934665603ull ^ bytes;
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
  }
};
static size_t small_slab() {                                      // WORLD_HIP_SMALL_SLAB: test hook (bytes per chained slab)
  static const size_t v = [] { const char *e = getenv("WORLD_HIP_SMALL_SLAB"); return e ? std::max<size_t>(256, (size_t)atoll(e)) : (size_t(4) << 20); }();
  return v;
}
static size_t small_budget() {                                    // WORLD_HIP_SMALL_BUDGET: test hook (bytes)
  static const size_t v = [] { const char *e = getenv("WORLD_HIP_SMALL_BUDGET"); return e ? (size_t)atoll(e) : (size_t(64) << 20); }();
  return v;
}

struct WorldHipContext {
  int device = 0;
  hipStream_t stream = nullptr;
  world_hip::Tables tab{nullptr, nullptr};
  world_hip::Arena arena;
  world_hip::HarvestBands bands;
  double *d_nuttall = nullptr;   // D4C band window
  int nuttall_len = 0;
  double *d_ap_frac = nullptr;   // d4c_finish's per-bin interpolation weight and knot (D4cParams::ap_frac / ap_knot), for
  int *d_ap_knot = nullptr;      //   (ap_grid_fs, ap_grid_fft)
  int ap_grid_fs = 0, ap_grid_fft = 0;
  void *dio_bands = nullptr;     // world_hip::DioBands (cached DIO filter tables)
  double *d_dc_remover = nullptr; // GetDCRemover(fft_size) of the synthesiser
  int dc_remover_len = 0;
  int *d_synth_need = nullptr;   // largest pulse count a synthesis call could not hold (0 = nothing was dropped)
  int synth_pulse_cap = 0;       // caller's capacity per utterance (0 = automatic)
  void *codec_tables = nullptr;  // world_hip::CodecTableSet (cached interp1 / DCT tables of the coders)
  SmallArrays small;             // the per-call host arrays, uploaded once per distinct content
  void *xchg_ready = nullptr, *xchg_done = nullptr;   // events of world_hip_allgather_blocks
  double *d_pk = nullptr;        // dense (tpos, f0) of world_hip_analyze_packed: [2][n_utt][f_stride], grow-only
  size_t pk_cap = 0;
  int hint = 0;                  // WORLD_HIP_HINT_* bits (world_hip_set_hint)
  // world_hip_analyze_sharded: this device's exchange stream, input staging (two pinned halves) and device input, grow-only
  hipStream_t xstream = nullptr;
  double *h_xin = nullptr, *d_xin = nullptr;
  size_t xin_cap = 0;            // doubles per half
  // Bumped whenever memory a captured graph may have baked in is freed or replaced (arena, small-array slabs, d_pk, the
  // cached filter / window / codec tables): world_hip_graph_launch refuses a graph of an older generation.
  unsigned long long generation = 0;
  // What the offset scans (and D4C's LoveTrain pass) still lying in the workspace were computed FOR: a later call may
  // ask to reuse them (`reuse_offsets`, the frame-range entries) and gets them only if its own inputs say the same thing
  // and nothing has touched that part of the arena since (the flag used to be trusted blindly).
  struct PrepToken {
    bool valid = false;
    int n_utt = 0, fs = 0, f_stride = 0, x_stride = 0, fft = 0;
    const void *x = nullptr, *f0 = nullptr, *tpos = nullptr, *arena_base = nullptr;
    double opt = 0.0;              // CheapTrick: its F0 floor; D4C: the threshold
    uint64_t lengths_hash = 0;     // x_length and n_frames
    size_t lo = 0, hi = 0;         // the arena bytes the prepared arrays occupy
    unsigned long long generation = 0;
    bool same_inputs(const PrepToken &o) const {
      return n_utt == o.n_utt && fs == o.fs && f_stride == o.f_stride && x_stride == o.x_stride && fft == o.fft && x == o.x &&
             f0 == o.f0 && tpos == o.tpos && arena_base == o.arena_base && opt == o.opt && lengths_hash == o.lengths_hash &&
             lo == o.lo && hi == o.hi && generation == o.generation;
    }
  };
  PrepToken prep_ct, prep_d4c;
  std::mutex lock;               // one call at a time per context
};

namespace world_hip {

static void ensure_arena(WorldHipContext *c, size_t bytes) {
  if (bytes <= c->arena.cap) return;
  devrt::sync(c->stream);
  if (c->arena.base) { devrt::dfree(c->arena.base); ++c->generation; }
  size_t cap = bytes + bytes / 8 + (1u << 20);
  c->arena.base = static_cast<char *>(devrt::dmalloc(cap));
  c->arena.cap = cap;
}

static size_t pad256(size_t bytes) { return (bytes + 255) & ~size_t(255); }

// The randn() stream is a constant of the algorithm: its first `draws` values live in the device's
// shared, fully verified table (rng.h / rng_fill.hip), generated once per process and extended on demand.
static const uint32_t *ensure_noise(WorldHipContext *c, size_t draws) {
  return noise_table_acquire(c->device, draws, c->tab.jump, c->stream);
}

// A small host array on the device: found by content in the context's slabs, or appended to them (one H2D copy from a
// host copy that lives as long as the entry).  Nothing a call already holds a pointer to is ever freed by an upload.
static void drop_small_arrays(WorldHipContext *c) {
  SmallArrays &sa = c->small;
  for (SmallArrays::Slab &sl : sa.slabs) devrt::dfree(sl.base);
  sa.slabs.clear();
  sa.entries.clear();
  sa.held = 0;
}
// The start of a stage's upload section: the ONLY place the small arrays may be dropped -- no params struct of this
// stage holds a pointer yet, and the kernels of earlier stages are waited for first.
struct CallScope {
  explicit CallScope(WorldHipContext *c, size_t) {
    if (c->small.held > small_budget() && !devrt::is_capturing(c->stream)) {
      devrt::sync(c->stream);
      drop_small_arrays(c);
      ++c->generation;
    }
  }
};

static char *small_array(WorldHipContext *c, const void *data, size_t bytes) {
  SmallArrays &sa = c->small;
  const uint64_t h = SmallArrays::hash(data, bytes);
  auto range = sa.entries.equal_range(h);
  for (auto it = range.first; it != range.second; ++it)
    if (it->second.bytes.size() == bytes && memcmp(it->second.bytes.data(), data, bytes) == 0) return it->second.dev;
  // a miss uploads (and may allocate): neither is possible while the stream is being captured into a graph
  if (devrt::is_capturing(c->stream)) fail("a call shape that was never run before cannot be captured: run it once first");
  const size_t padded = (bytes + 255) & ~size_t(255);
  if (sa.slabs.empty() || sa.slabs.back().used + padded > sa.slabs.back().cap) {
    SmallArrays::Slab sl;                                          // chain a fresh slab; the old ones stay valid
    sl.cap = std::max(small_slab(), padded);
    sl.base = static_cast<char *>(devrt::dmalloc(sl.cap));
    sl.used = 0;
    sa.slabs.push_back(sl);
    sa.held += sl.cap;
  }
  SmallArrays::Slab &sl = sa.slabs.back();
  SmallArrays::Entry e;
  e.bytes.assign(static_cast<const char *>(data), static_cast<const char *>(data) + bytes);
  e.dev = sl.base + sl.used;
  sl.used += padded;
  auto kept = sa.entries.emplace(h, std::move(e));
  devrt::h2d(kept->second.dev, kept->second.bytes.data(), bytes, c->stream);   // the source outlives the copy (node-based map)
  return kept->second.dev;
}

template <class T> static T *upload(WorldHipContext *c, const std::vector<T> &v) {
  if (v.empty()) return reinterpret_cast<T *>(small_array(c, "", 1));
  return reinterpret_cast<T *>(small_array(c, v.data(), sizeof(T) * v.size()));
}

static int ilog2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  if ((1 << l) != n) fail("fft size %d is not a power of two", n);
  return l;
}

static int frame_count(int fs, int x_length, double frame_period) {   // harvest.cpp:1219, dio.cpp:639
  return static_cast<int>(1000.0 * x_length / fs / frame_period) + 1;
}

static void check_batch(int n_utt, int fs, const void *d_x, int x_stride, const int *x_length) {
  if (n_utt <= 0) fail("n_utt must be positive");
  if (fs <= 0) fail("fs must be positive");
  if (!d_x || !x_length) fail("null input");
  for (int u = 0; u < n_utt; ++u)
    if (x_length[u] <= 0 || x_length[u] > x_stride) fail("x_length[%d]=%d outside (0, x_stride=%d]", u, x_length[u], x_stride);
}

// Where a per-frame stage writes its rows: the dense [n_utt][f_stride][bins] array of the batched API (rows == nullptr),
// or rows of packed records -- frame f of utterance u at row rows[u] + f, `stride` doubles apart (exchange.hip's layout,
// written by the stage kernels themselves instead of a pack pass over 2 x 16 KB per frame).
struct RowLayout {
  const int *rows = nullptr;     // host, [n_utt]
  size_t stride = 0;             // 0: bins
  size_t col_bytes = 0;          // bytes from a record's start to this stage's row (the stage's output pointer = records' base)
  int f32 = 0;                   // 1: rows stored as float (narrow wire format)
  double *rec = nullptr;         // D4C only: records' base for the (tpos, f0) head of every record
  // coded records (run_analyze_coded): > 0 = the stage writes its CODED row -- CheapTrick the first code_dims mel-cepstrum
  // coefficients (with the coder's tables), D4C the band aperiodicities -- instead of the row itself
  int code_dims = 0;
  const int *code_knot = nullptr;
  const double *code_frac = nullptr, *code_w_re = nullptr, *code_w_im = nullptr;
  // frames [frame_lo, frame_hi) of every utterance only (stream positions as in a whole-utterance call); skip_prepare:
  // the offsets (and D4C's LoveTrain pass) of the previous call with the same shape are still in the workspace
  int frame_lo = 0, frame_hi = 0x7FFFFFFF;
  bool skip_prepare = false;
};
// Doubles per packed record.  wire 0: [tpos, f0, sp f64[nb], ap f64[nb]]; wire 1: [tpos, f0, sp f32[nb], ap f32[nb]] -- half
// the bytes on the xGMI links and in the D2H copy, 6e-8 relative (the contract is 1e-4): 16 + 8 nb bytes = 2 + nb doubles.
static int record_cols(int fft_size, int wire) {
  const int nb = fft_size / 2 + 1;
  return wire == 1 ? 2 + nb : 2 + 2 * nb;
}

static size_t cheaptrick_arena_bytes(int n_utt, int f_stride, int fft_size) {
  (void)fft_size;       // the frame kernel keeps everything between x and the spectrogram in LDS: no per-frame scratch
  return pad256(sizeof(unsigned) * (size_t)n_utt * f_stride) + 4 * pad256(sizeof(int) * n_utt);
}
// (d4c.cpp:350-351) fft_size_d4c, the transform of the frame kernel
static int d4c_internal_fft(int fs) { return static_cast<int>(pow(2.0, 1.0 + static_cast<int>(log(4.0 * fs / kFloorF0D4C + 1) / kLog2))); }
// workgroups per d4c_frame launch of the 16384-point shape (each parks its group delay in a slot of global memory,
// stage_params.h): eight rounds of a 256-CU chip that runs one such workgroup per CU -- 134 MB
static int d4c_park_slots(int n_utt, int f_stride) {
  return static_cast<int>(std::min<size_t>((size_t)n_utt * f_stride, (size_t)std::max(n_utt, 2048)));
}
static size_t d4c_arena_bytes(int n_utt, int f_stride, int fs) {
  const size_t fr = (size_t)n_utt * f_stride;
  int lg = 0;
  while ((1 << lg) < d4c_internal_fft(fs)) ++lg;
  return 2 * pad256(sizeof(unsigned) * fr) + pad256(sizeof(double) * fr) + 6 * pad256(sizeof(int) * n_utt) + 512 +
         pad256(sizeof(double) * fr * 16) + pad256(sizeof(double) * d4c_park_slot_doubles(lg) * d4c_park_slots(n_utt, f_stride));
}

// ---------------------------------------------------------------------------
// The spectral stages' workspace.  CheapTrick's arrays sit at the bottom of the arena, D4C's behind them (a D4C call by
// itself skips CheapTrick's part for its own shape): the two never overlap, so the offset scans of both -- and D4C's
// LoveTrain pass -- survive the other stage's calls, one prepare kernel can serve both (run_spectral_stages), and frame
// ranges of the two stages can alternate while reusing them.  Any other stage's arena_reset() voids them.
// ---------------------------------------------------------------------------
static void arena_reset(WorldHipContext *c) {
  c->arena.reset();
  c->prep_ct.valid = c->prep_d4c.valid = false;
}
static uint64_t lengths_hash(const int *x_length, const int *n_frames, int n_utt) {
  const uint64_t a = SmallArrays::hash(x_length, sizeof(int) * n_utt), b = SmallArrays::hash(n_frames, sizeof(int) * n_utt);
  return a * 0x9E3779B97F4A7C15ull ^ b;
}
// Claims [lo, hi) of the arena for `mine`; with `reuse` the claim must equal what is there (else: a clear error instead
// of kernels reading garbage offsets).  The other stage's prepared arrays are void if the claim runs into them.
static void claim_prepared(WorldHipContext *c, WorldHipContext::PrepToken &mine, WorldHipContext::PrepToken &other,
                           WorldHipContext::PrepToken want, bool reuse, const char *stage) {
  want.arena_base = c->arena.base; want.generation = c->generation;
  if (reuse) {
    if (!mine.valid)
      fail("%s: reuse_offsets, but this context holds no prepared offsets (another stage, another shape or a regrown "
           "workspace since the call that prepared them)", stage);
    if (!mine.same_inputs(want))
      fail("%s: reuse_offsets, but the prepared offsets belong to other inputs (shape, buffers or options differ from the "
           "call that prepared them)", stage);
    return;
  }
  if (other.valid && want.lo < other.hi && other.lo < want.hi) other.valid = false;
  mine = want;
  mine.valid = true;
}

// ---------------------------------------------------------------------------
// CheapTrick
// ---------------------------------------------------------------------------
static CtParams setup_cheaptrick(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                                 const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                                 const double *d_f0, const CheapTrickOption *opt, double *d_sp, const RowLayout &lay,
                                 int *max_frames_out) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  const int lg = ilog2_exact(opt->fft_size);
  if (lg < 7 || lg > 13) fail("CheapTrick fft_size %d unsupported (128..8192: one frame must fit LDS)", opt->fft_size);
  int max_frames = 0;
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 0 || n_frames[u] > f_stride) fail("n_frames[%d] outside [0, f_stride]", u);
    max_frames = std::max(max_frames, n_frames[u]);
  }
  *max_frames_out = max_frames;
  CtParams p;
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, std::vector<int>(x_length, x_length + n_utt));
  p.b.n_frames = upload(c, std::vector<int>(n_frames, n_frames + n_utt));
  p.tpos = d_tpos; p.f0 = d_f0; p.spectrogram = d_sp;
  p.out_row = lay.rows ? upload(c, std::vector<int>(lay.rows, lay.rows + n_utt)) : nullptr;
  p.out_stride = lay.stride ? lay.stride : (size_t)(opt->fft_size / 2 + 1);
  p.out_col_bytes = lay.col_bytes; p.out_f32 = lay.f32;
  p.code_ndim = lay.code_dims; p.code_knot = lay.code_knot; p.code_frac = lay.code_frac;
  p.code_w_re = lay.code_w_re; p.code_w_im = lay.code_w_im;
  if (p.code_ndim > 0 && (lay.f32 || !lay.code_knot || p.code_ndim > opt->fft_size / 4 + 1)) fail("CheapTrick: bad coded-row layout");
  p.f0_floor = 3.0 * fs / (opt->fft_size - 3.0);                       // cheaptrick.cpp:196-198
  c->arena.skip_to(0);
  p.offsets = c->arena.take<unsigned>((size_t)n_utt * f_stride);
  {
    WorldHipContext::PrepToken t;
    t.n_utt = n_utt; t.fs = fs; t.f_stride = f_stride; t.x_stride = x_stride; t.fft = opt->fft_size;
    t.x = d_x; t.f0 = d_f0; t.tpos = d_tpos; t.opt = p.f0_floor; t.lengths_hash = lengths_hash(x_length, n_frames, n_utt);
    t.lo = 0; t.hi = c->arena.used;
    claim_prepared(c, c->prep_ct, c->prep_d4c, t, lay.skip_prepare, "CheapTrick");
  }
  p.noise = ensure_noise(c, (size_t)max_frames * ct_max_draws_per_frame(opt->fft_size));
  p.tab = c->tab;
  p.q1 = opt->q1;
  p.lg_fft = lg;
  if (lay.frame_lo < 0 || lay.frame_hi < lay.frame_lo) fail("bad frame range [%d, %d)", lay.frame_lo, lay.frame_hi);
  p.frame_lo = lay.frame_lo; p.frame_hi = lay.frame_hi; p.skip_prepare = lay.skip_prepare ? 1 : 0;
  return p;
}
static void run_cheaptrick(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                           const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                           const double *d_f0, const CheapTrickOption *opt, double *d_sp,
                           const RowLayout &lay = RowLayout()) {
  ensure_arena(c, cheaptrick_arena_bytes(n_utt, f_stride, opt->fft_size));
  CallScope scope(c, 3 * sizeof(int) * n_utt + 256);
  int max_frames = 0;
  const CtParams p = setup_cheaptrick(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, opt, d_sp, lay,
                                      &max_frames);
  launch_cheaptrick(p, max_frames, c->stream);
}

// ---------------------------------------------------------------------------
// D4C
// ---------------------------------------------------------------------------
static D4cParams setup_d4c(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                           const int *n_frames, int f_stride, const double *d_tpos, const double *d_f0, int fft_size,
                           const D4COption *opt, double *d_ap, const RowLayout &lay, int *max_frames_out) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  ilog2_exact(fft_size);
  int max_frames = 0;
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 0 || n_frames[u] > f_stride) fail("n_frames[%d] outside [0, f_stride]", u);
    max_frames = std::max(max_frames, n_frames[u]);
  }
  *max_frames_out = max_frames;
  // d4c.cpp:350-363 and :264-265
  const int fft_d4c = d4c_internal_fft(fs);
  const int fft_love = static_cast<int>(pow(2.0, 1.0 + static_cast<int>(log(3.0 * fs / 40.0 + 1) / kLog2)));
  if (fft_d4c > 16384 || fft_love > 16384)
    fail("D4C: fs=%d needs an internal FFT of %d > 16384 points (LDS budget); fs <= 192 kHz supported", fs, std::max(fft_d4c, fft_love));
  if (fs < 15800) fail("D4C: fs=%d is below the 15.8 kHz the reference's LoveTrain band edges require", fs);
  const int nap = static_cast<int>(std::min(15000.0, fs / 2.0 - 3000.0) / 3000.0);
  const int wl = static_cast<int>(3000.0 * fft_d4c / fs) * 2 + 1;
  if (nap < 1 || nap > 6) fail("D4C: unsupported number of aperiodicity bands %d", nap);
  if (c->nuttall_len != wl) {                                         // NuttallWindow, common.cpp:113-121
    std::vector<double> w(wl);
    for (int i = 0; i < wl; ++i) {
      double t = i / (wl - 1.0);
      w[i] = 0.355768 - 0.487396 * cos(2.0 * kPi * t) + 0.144232 * cos(4.0 * kPi * t) - 0.012604 * cos(6.0 * kPi * t);
    }
    devrt::sync(c->stream);
    if (c->d_nuttall) { devrt::dfree(c->d_nuttall); ++c->generation; }
    c->d_nuttall = static_cast<double *>(devrt::dmalloc(sizeof(double) * wl));
    devrt::h2d(c->d_nuttall, w.data(), sizeof(double) * wl, c->stream);
    devrt::sync(c->stream);
    c->nuttall_len = wl;
  }
  if (c->ap_grid_fs != fs || c->ap_grid_fft != fft_size) {
    // GetAperiodicity's interp1 (d4c.cpp:330-338) of the nap + 2 coarse values at knots [0, 3000, .., 3000 nap, fs/2] onto the
    // caller's bins i fs / fft_size: knot (histc) and weight of every bin, with the expressions d4c_finish evaluated per
    // bin and frame until round 6 (same operands, same order: the same bits)
    const int nb_out = fft_size / 2 + 1, nk = nap + 2;
    std::vector<double> frac(nb_out);
    std::vector<int> knot(nb_out);
    for (int i = 0; i < nb_out; ++i) {
      const double xi = static_cast<double>(i) * fs / fft_size;
      int cnt = 0;
      for (int k = 0; k < nk; ++k) {
        const double kn = k <= nap ? k * 3000.0 : fs / 2.0;
        if (kn <= xi) cnt++;
      }
      const int k = cnt < 1 ? 1 : (cnt > nk - 1 ? nk - 1 : cnt);
      const double x0 = (k - 1) <= nap ? (k - 1) * 3000.0 : fs / 2.0;
      const double x1 = k <= nap ? k * 3000.0 : fs / 2.0;
      knot[i] = k;
      frac[i] = (xi - x0) / (x1 - x0);
    }
    devrt::sync(c->stream);
    if (c->d_ap_frac) { devrt::dfree(c->d_ap_frac); devrt::dfree(c->d_ap_knot); ++c->generation; }
    c->d_ap_frac = static_cast<double *>(devrt::dmalloc(sizeof(double) * nb_out));
    c->d_ap_knot = static_cast<int *>(devrt::dmalloc(sizeof(int) * nb_out));
    devrt::h2d(c->d_ap_frac, frac.data(), sizeof(double) * nb_out, c->stream);
    devrt::h2d(c->d_ap_knot, knot.data(), sizeof(int) * nb_out, c->stream);
    devrt::sync(c->stream);
    c->ap_grid_fs = fs; c->ap_grid_fft = fft_size;
  }
  size_t fr = (size_t)n_utt * f_stride;
  D4cParams p;
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, std::vector<int>(x_length, x_length + n_utt));
  p.b.n_frames = upload(c, std::vector<int>(n_frames, n_frames + n_utt));
  p.tpos = d_tpos; p.f0 = d_f0; p.aperiodicity = d_ap;
  p.out_row = lay.rows ? upload(c, std::vector<int>(lay.rows, lay.rows + n_utt)) : nullptr;
  p.out_stride = lay.stride ? lay.stride : (size_t)(fft_size / 2 + 1);
  p.out_col_bytes = lay.col_bytes; p.out_f32 = lay.f32;
  p.code_nap = lay.code_dims;
  if (p.code_nap > 0 && lay.f32) fail("D4C: bad coded-row layout");
  p.rec = lay.rec;
  c->arena.skip_to(cheaptrick_arena_bytes(n_utt, f_stride, fft_size));      // behind CheapTrick's part for this shape
  const size_t region_lo = c->arena.used;
  p.ap0 = c->arena.take<double>(fr);
  p.offsets1 = c->arena.take<unsigned>(fr);
  p.draws2 = c->arena.take<unsigned>(fr);
  p.draws1 = c->arena.take<unsigned>(n_utt);
  p.coarse = c->arena.take<double>(fr * 16);
  p.park_slots = d4c_park_slots(n_utt, f_stride);
  {
    const size_t slot = d4c_park_slot_doubles(ilog2_exact(fft_d4c));
    p.park_ws = slot ? c->arena.take<double>(slot * p.park_slots) : nullptr;
  }
  {
    WorldHipContext::PrepToken t;
    t.n_utt = n_utt; t.fs = fs; t.f_stride = f_stride; t.x_stride = x_stride; t.fft = fft_size;
    t.x = d_x; t.f0 = d_f0; t.tpos = d_tpos; t.opt = opt->threshold; t.lengths_hash = lengths_hash(x_length, n_frames, n_utt);
    t.lo = region_lo; t.hi = c->arena.used;
    claim_prepared(c, c->prep_d4c, c->prep_ct, t, lay.skip_prepare, "D4C");
  }
  p.noise = ensure_noise(c, (size_t)max_frames * d4c_max_draws_per_frame(fs));
  p.nuttall = c->d_nuttall;
  p.ap_frac = c->d_ap_frac; p.ap_knot = c->d_ap_knot;
  p.tab = c->tab;
  p.threshold = opt->threshold;
  p.fft_out = fft_size;
  p.lg_love = ilog2_exact(fft_love);
  p.lg_d4c = ilog2_exact(fft_d4c);
  p.nap = nap;
  p.wl = wl;
  // d4c_frame's band transforms rely on the slice being at most 512 packed elements (3000 N / fs < 511.1 for every fs >= 15.8 kHz)
  if (wl / 2 + 1 > 512) fail("D4C: band window of %d samples at fs=%d exceeds the 1023 the frame kernel is built for", wl, fs);
  for (int b = 0; b < 8; ++b) p.band_center[b] = static_cast<int>(3000.0 * (b + 1) * fft_d4c / fs);      // d4c.cpp:207-208
  if (lay.frame_lo < 0 || lay.frame_hi < lay.frame_lo) fail("bad frame range [%d, %d)", lay.frame_lo, lay.frame_hi);
  p.frame_lo = lay.frame_lo; p.frame_hi = lay.frame_hi;
  p.skip_prepare = lay.skip_prepare ? kD4cSkipScan | kD4cSkipLoveTrain : 0;
  return p;
}
static void run_d4c(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                    const int *n_frames, int f_stride, const double *d_tpos, const double *d_f0, int fft_size,
                    const D4COption *opt, double *d_ap, const RowLayout &lay = RowLayout()) {
  ensure_arena(c, cheaptrick_arena_bytes(n_utt, f_stride, fft_size) + d4c_arena_bytes(n_utt, f_stride, fs));
  CallScope scope(c, 3 * sizeof(int) * n_utt + 256);
  int max_frames = 0;
  const D4cParams p = setup_d4c(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, fft_size, opt, d_ap,
                                lay, &max_frames);
  launch_d4c(p, max_frames, c->stream);
}

// ---------------------------------------------------------------------------
// Harvest
// ---------------------------------------------------------------------------
static void prepare_bands(WorldHipContext *c, int fs, double f0_floor, double f0_ceil, int ratio) {
  HarvestBands &hb = c->bands;
  if (hb.fs == fs && hb.f0_floor == f0_floor && hb.f0_ceil == f0_ceil && hb.nch > 0) return;
  // harvest.cpp:1149-1157
  const double lo = f0_floor * 0.9, hi = f0_ceil * 1.1, cpo = 40;
  const int nch = 1 + static_cast<int>(log(hi / lo) / kLog2 * cpo);
  const double afs = static_cast<double>(fs) / ratio;
  std::vector<double> fb(nch), taps;
  std::vector<int> half(nch), off(nch);
  int max_half = 0;
  for (int i = 0; i < nch; ++i) {
    fb[i] = lo * pow(2.0, (i + 1) / cpo);
    // GetFilteredSignal's time-domain design, harvest.cpp:101-108
    const int L = mround(afs / fb[i] * 2.0);
    half[i] = L;
    off[i] = (int)taps.size();
    max_half = std::max(max_half, L);
    const int len = 2 * L + 1;
    for (int k = -L; k <= L; ++k) {
      double t = (k + L) / (len - 1.0);
      double w = 0.355768 - 0.487396 * cos(2.0 * kPi * t) + 0.144232 * cos(4.0 * kPi * t) - 0.012604 * cos(6.0 * kPi * t);
      taps.push_back(w * cos(2 * kPi * fb[i] * k / afs));
    }
  }
  devrt::sync(c->stream);
  ++c->generation;
  if (hb.d_band_f0) { devrt::dfree(hb.d_band_f0); devrt::dfree(hb.d_taps); devrt::dfree(hb.d_half); devrt::dfree(hb.d_off); }
  hb.d_band_f0 = static_cast<double *>(devrt::dmalloc(sizeof(double) * nch));
  hb.d_taps = static_cast<double *>(devrt::dmalloc(sizeof(double) * taps.size()));
  hb.d_half = static_cast<int *>(devrt::dmalloc(sizeof(int) * nch));
  hb.d_off = static_cast<int *>(devrt::dmalloc(sizeof(int) * nch));
  devrt::h2d(hb.d_band_f0, fb.data(), sizeof(double) * nch, c->stream);
  devrt::h2d(hb.d_taps, taps.data(), sizeof(double) * taps.size(), c->stream);
  devrt::h2d(hb.d_half, half.data(), sizeof(int) * nch, c->stream);
  devrt::h2d(hb.d_off, off.data(), sizeof(int) * nch, c->stream);
  devrt::sync(c->stream);
  hb.fs = fs; hb.f0_floor = f0_floor; hb.f0_ceil = f0_ceil; hb.nch = nch; hb.max_half = max_half;
  hb.band_f0 = fb;
  if (hb.d_spec) { devrt::dfree(hb.d_spec); hb.d_spec = nullptr; }
  hb.fft_seg = getenv("WORLD_HIP_HARVEST_FIR") ? 0 : hv_fft_segment(max_half);
  if (hb.fft_seg > 0) {
    hb.d_spec = static_cast<double2 *>(devrt::dmalloc(sizeof(double2) * (size_t)nch * kBandFftBins));
    launch_band_spectra(hb.d_taps, hb.d_off, hb.d_half, nch, hb.d_spec, c->tab, c->stream);
    devrt::sync(c->stream);
  }
  // GetMainWindow's angles for every window half length hv_refine can meet (harvest.cpp:446-456), d = 2/(2hw+1):
  //   [hw][6]        sin/cos(pi d), sin/cos(pi WAVE d), 2 / window length in seconds, pi d
  //   then [hw][WAVE] (sin, cos)(pi (lane - hw - 1) d): a lane's FIRST sample of the window when the frame centre falls on
  //                  a whole sample (it does at 8 kHz: the decimated rate of 16 / 32 / 48 / 96 kHz input); the kernel turns
  //                  it by the residual angle for other rates.  One 16-byte load replaces a division and a sincospi per
  //                  window rebuild (70 of its ~190 instructions).
  const int hw_max = static_cast<int>(1.5 * afs / f0_floor + 1.0) + 2;
  const size_t head = (size_t)(hw_max + 1) * 6;
  std::vector<double> wt(head + (size_t)(hw_max + 1) * WAVE * 2);
  const long double pi_l = 3.14159265358979323846264338327950288L;
  for (int hw = 0; hw <= hw_max; ++hw) {
    const double wlen_t = (2.0 * hw + 1.0) / afs;
    const double d = (1.0 / afs) * (2.0 / wlen_t);
    wt[6 * hw + 0] = sin(kPi * d); wt[6 * hw + 1] = cos(kPi * d);
    wt[6 * hw + 2] = sin(kPi * (WAVE * d)); wt[6 * hw + 3] = cos(kPi * (WAVE * d));
    wt[6 * hw + 4] = 2.0 / ((2.0 * hw + 1.0) * (1.0 / afs));           // the kernel's two_over_t, same operations
    wt[6 * hw + 5] = kPi * d;
    const long double dl = 2.0L / (2.0L * hw + 1.0L);
    for (int lane = 0; lane < WAVE; ++lane) {
      const long double a = pi_l * (lane - hw - 1) * dl;
      wt[head + ((size_t)hw * WAVE + lane) * 2 + 0] = (double)sinl(a);
      wt[head + ((size_t)hw * WAVE + lane) * 2 + 1] = (double)cosl(a);
    }
  }
  // Where a millisecond is a whole number of samples (8 kHz: the decimated rate of 16 / 32 / 48 / 96 kHz input) every
  // refinement window is centred ON a sample, and the window of half length hw is the same 2 hw + 1 numbers at every
  // frame: the main window (GetMainWindow, harvest.cpp:446-456) and its central difference (GetDiffWindow, :462-468) as
  // (w, dw) pairs, the window of half length hw at entry hw^2 (sum of 2 h + 1 below it).  A rebuild is then a load and two
  // products per sample instead of ~25 FP64 operations (round 4: window rebuilds were 18 of a wavefront-frame's 65
  // thousand cycles).  Other rates keep the rotation route (wt above).
  hb.win_full_entries = 0;
  if (fmod(afs, 1000.0) == 0.0) {
    hb.win_full_entries = (size_t)(hw_max + 1) * (hw_max + 1);
    const size_t base = wt.size();
    wt.resize(base + 2 * hb.win_full_entries);
    std::vector<long double> w;
    for (int hw = 0; hw <= hw_max; ++hw) {
      const int blen = 2 * hw + 1;
      w.assign(blen, 0.0L);
      const long double dl = 2.0L / (2.0L * hw + 1.0L);
      for (int i = 0; i < blen; ++i) {
        const long double a = pi_l * (i - hw - 1) * dl;
        w[i] = 0.42L + 0.5L * cosl(a) + 0.08L * cosl(2.0L * a);
      }
      double *o = wt.data() + base + 2 * (size_t)hw * hw;
      for (int i = 0; i < blen; ++i) {
        long double dw;
        if (blen == 1) dw = 0.0L;
        else if (i == 0) dw = -w[1] / 2.0L;
        else if (i == blen - 1) dw = w[blen - 2] / 2.0L;
        else dw = -(w[i + 1] - w[i - 1]) / 2.0L;
        o[2 * i] = (double)w[i]; o[2 * i + 1] = (double)dw;
      }
    }
  }
  if (hb.d_win_tab) devrt::dfree(hb.d_win_tab);
  hb.d_win_tab = static_cast<double *>(devrt::dmalloc(sizeof(double) * wt.size()));
  devrt::h2d(hb.d_win_tab, wt.data(), sizeof(double) * wt.size(), c->stream);
  devrt::sync(c->stream);
  hb.win_tab_len = hw_max + 1;
}

static void run_harvest(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                        const int *x_length, const HarvestOption *opt, int f_stride, double *d_tpos,
                        double *d_f0) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  if (!(opt->f0_floor > 0) || !(opt->f0_ceil > opt->f0_floor) || !(opt->frame_period > 0)) fail("bad HarvestOption");
  HarvestParams p;
  // Harvest(): harvest.cpp:1226-1228
  p.ratio = std::max(std::min(mround(fs / 8000.0), 12), 1);
  p.afs = static_cast<double>(fs) / p.ratio;
  prepare_bands(c, fs, opt->f0_floor, opt->f0_ceil, p.ratio);
  const HarvestBands &hb = c->bands;
  p.f0_floor = opt->f0_floor; p.f0_ceil = opt->f0_ceil; p.frame_period = opt->frame_period;
  p.nch = hb.nch;
  p.maxc = mround(hb.nch / 10.0) * 7;                                  // harvest.cpp:1179-1181
  if (p.maxc > 256) fail("Harvest: %d candidate slots per frame exceed the 256 the tracking kernel handles", p.maxc);
  p.lag = static_cast<int>(ceil(140.0 / p.ratio) * p.ratio);           // harvest.cpp:50-51
  std::vector<int> xl(x_length, x_length + n_utt), yl(n_utt), nfb(n_utt), nfr(n_utt), rfft(n_utt);
  int max_x = 0, max_y = 0, max_fb = 0, max_fr = 0;
  for (int u = 0; u < n_utt; ++u) {
    yl[u] = static_cast<int>(ceil(static_cast<double>(xl[u]) / p.ratio));   // harvest.cpp:1161-1162
    // the reference's transform length (harvest.cpp:1164-1165, GetSuitableFFTSize = common.cpp:51-54)
    const int span = yl[u] + 5 + 2 * static_cast<int>(2.0 * p.afs / hb.band_f0[0]);
    rfft[u] = static_cast<int>(pow(2.0, static_cast<int>(log(static_cast<double>(span)) / kLog2) + 1.0));
    nfb[u] = frame_count(fs, xl[u], 1);
    nfr[u] = frame_count(fs, xl[u], opt->frame_period);
    if (nfr[u] > f_stride) fail("f_stride %d too small for %d frames", f_stride, nfr[u]);
    max_x = std::max(max_x, xl[u]); max_y = std::max(max_y, yl[u]);
    max_fb = std::max(max_fb, nfb[u]); max_fr = std::max(max_fr, nfr[u]);
  }
  p.y_stride = (max_y + 7) & ~7;
  p.fb_stride = (max_fb + 7) & ~7;
  p.m_stride = (max_x + 2 * p.lag + 2 * kDecPad + 7) & ~7;
  p.ev_cap = max_y / 2 + 2;
  p.refine_cap = 2 * static_cast<int>(1.5 * p.afs / opt->f0_floor + 1.0) + 4;
  p.sec_cap = max_fb / 7 + 4;
  p.lone_job = n_utt == 1 && !(c->hint & WORLD_HIP_HINT_SHARED_DEVICE);
  p.ext_cap = max_fb + 304 * p.sec_cap + 8;
  p.max_half = hb.max_half;
  p.tab = c->tab;
  p.fft_seg = hb.fft_seg;
  p.fft_pre = hb.max_half - 1;
  p.band_spec = hb.d_spec;
  if (p.fft_seg > 0) {
    // FFT path: a workgroup filters `chunk_blocks` consecutive blocks of one (band, utterance) and appends their events to one
    // list.  Batches get one chunk per utterance (the final lists, no compaction pass); a lone utterance is cut into
    // enough chunks to fill the chip (~2048 workgroups).
    p.nblk = (max_y + p.fft_seg - 1) / p.fft_seg;
    const int want = std::max(1, std::min(p.nblk, (2048 + p.nch * n_utt - 1) / (p.nch * n_utt)));
    p.chunk_blocks = (p.nblk + want - 1) / want;
    p.nseg = (p.nblk + p.chunk_blocks - 1) / p.chunk_blocks;
    p.seg_cap = p.nseg == 1 ? p.ev_cap : std::min(p.ev_cap, p.chunk_blocks * p.fft_seg / 2 + 2);
  } else {
    p.nblk = 0; p.chunk_blocks = 0;
    p.nseg = band_segments(max_y);
    p.seg_cap = kSegCap;
  }
  const bool direct_lists = p.fft_seg > 0 && p.nseg == 1;     // the filter bank writes the final event lists itself

  const size_t B = n_utt;
  // utterances whose zero-crossing lists exist at once (harvest.hip: launch_harvest): WORLD_HIP_EVENT_GROUP, default 40 --
  // a 40-utterance launch of the filter bank is 6 080 workgroups, six times what the chip holds at once.  Measured on the
  // configs[3] share (128 utterances, two contexts in flight): groups of 128 / 64 / 32 -> 4.625 / 4.614 / 4.599 M frames/s
  // with 36.0 / 21.9 / 14.9 GB of workspace; configs[2] (256 utterances): 10.68 M with 71.9 GB ungrouped, 10.69 M with
  // 29.9 GB in groups of 64.
  static const int ev_group_env = [] { const char *e = getenv("WORLD_HIP_EVENT_GROUP"); return e ? std::max(1, atoi(e)) : 40; }();
  p.ev_group = std::min(n_utt, ev_group_env);
  p.ev_u0 = 0;
  const size_t E = p.ev_group;
  const size_t cand_elems = B * p.fb_stride * p.maxc;
  size_t need = 0;
  need += 5 * pad256(sizeof(int) * B) + pad256(sizeof(double) * B * ((max_y + 4095) / 4096) * 4) +
          pad256(sizeof(double) * B * ((max_y + 4095) / 4096 + (max_x + 2 * p.lag + 2 * kDecPad) / kDecSpan + 1)) +
          pad256(sizeof(double) * B * p.nch * 4);
  need += pad256(sizeof(double) * B * p.m_stride);
  need += pad256(sizeof(double) * B * p.y_stride);
  need += pad256(sizeof(double) * E * p.nch * 4 * p.ev_cap);
  need += pad256(sizeof(int) * E * p.nch * 4);
  if (!direct_lists) {
    need += pad256(sizeof(double) * E * p.nch * 4 * p.nseg * p.seg_cap);
    need += pad256(sizeof(int) * E * p.nch * 4 * p.nseg);
  }
  need += pad256(sizeof(double2) * B * p.nblk * kBandFftBins);
  need += pad256(sizeof(double) * B * p.nch * p.fb_stride);
  need += 4 * pad256(sizeof(double) * cand_elems);
  need += pad256(sizeof(int) * B);
  need += 4 * pad256(sizeof(double) * B * p.fb_stride);
  need += pad256(sizeof(int) * B * 6 * p.sec_cap) + pad256(sizeof(int) * B * 2) + pad256(sizeof(double) * B * p.sec_cap);
  need += pad256(sizeof(double) * B * p.ext_cap);
  ensure_arena(c, need);
  arena_reset(c);
  CallScope scope(c, 5 * sizeof(int) * n_utt + 512);

  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, xl);
  p.b.n_frames = upload(c, nfr);
  p.y_len = upload(c, yl);
  p.nfb = upload(c, nfb);
  p.ref_fft = upload(c, rfft);
  p.nyq_slices = (max_y + 4095) / 4096;                                // kMeanSlice of harvest.hip
  p.nyq = c->arena.take<double>(B * p.nyq_slices * 4);
  // launch_harvest's grids: spans of the decimation sweeps (harvest.hip), or the 4096-sample slices at ratio 1
  p.mean_parts = p.ratio == 1 ? p.nyq_slices : (max_x + 2 * p.lag + 2 * kDecPad + kDecSpan - 1) / kDecSpan;
  p.mean_part = c->arena.take<double>(B * p.mean_parts);
  p.quirk = c->arena.take<double>(B * p.nch * 4);
  p.band_f0 = hb.d_band_f0; p.band_half = hb.d_half; p.band_off = hb.d_off; p.band_taps = hb.d_taps;
  p.win_tab = hb.d_win_tab; p.win_lane = hb.d_win_tab + (size_t)hb.win_tab_len * 6;
  p.win_full = hb.win_full_entries && !getenv("WORLD_HIP_REFINE_ROTATE")
                   ? reinterpret_cast<const double2 *>(p.win_lane + (size_t)hb.win_tab_len * WAVE * 2) : nullptr;
  p.fwd = c->arena.take<double>(B * p.m_stride);
  p.y = c->arena.take<double>(B * p.y_stride);
  p.events = c->arena.take<double>(E * p.nch * 4 * p.ev_cap);
  p.ev_count = c->arena.take<int>(E * p.nch * 4);
  if (direct_lists) {
    p.seg_events = p.events; p.seg_count = p.ev_count;
  } else {
    p.seg_events = c->arena.take<double>(E * p.nch * 4 * p.nseg * p.seg_cap);
    p.seg_count = c->arena.take<int>(E * p.nch * 4 * p.nseg);
  }
  p.blk_spec = c->arena.take<double2>(B * p.nblk * kBandFftBins);
  p.raw = c->arena.take<double>(B * p.nch * p.fb_stride);
  p.cand_a = c->arena.take<double>(cand_elems); p.score_a = c->arena.take<double>(cand_elems);
  p.cand_b = c->arena.take<double>(cand_elems); p.score_b = c->arena.take<double>(cand_elems);
  p.nc = c->arena.take<int>(B);
  p.c0 = c->arena.take<double>(B * p.fb_stride);
  p.c2 = c->arena.take<double>(B * p.fb_stride); p.c3 = c->arena.take<double>(B * p.fb_stride);
  p.basic_f0 = c->arena.take<double>(B * p.fb_stride);
  p.sec = c->arena.take<int>(B * 6 * p.sec_cap);
  p.sec_n = c->arena.take<int>(B * 2);
  p.sec_sum = c->arena.take<double>(B * p.sec_cap);
  p.ext = c->arena.take<double>(B * p.ext_cap);
  p.tpos = d_tpos; p.f0 = d_f0;
  launch_harvest(p, max_x, max_y, max_fb, max_fr, c->stream);
#ifdef WORLD_EMU
  // host emulation only (synchronous): intermediate stages for debugging against the oracle's WO_DUMP
  if (const char *dump = getenv("WORLD_EMU_DUMP")) {
    auto put = [&](const char *tag, const void *ptr, size_t bytes) {
      std::string fn = std::string(dump) + "_" + tag + ".bin";
      FILE *f = fopen(fn.c_str(), "wb"); fwrite(ptr, 1, bytes, f); fclose(f);
    };
    int hdr[6] = {p.nch, p.fb_stride, p.maxc, max_fb, p.y_stride, yl[0]};
    put("hdr", hdr, sizeof hdr);
    put("y", p.y, sizeof(double) * p.y_stride);
    put("raw", p.raw, sizeof(double) * p.nch * p.fb_stride);
    put("cand_a", p.cand_a, sizeof(double) * p.fb_stride * p.maxc);
    put("cand_b", p.cand_b, sizeof(double) * p.fb_stride * p.maxc);
    put("score_a", p.score_a, sizeof(double) * p.fb_stride * p.maxc);
    put("score_b", p.score_b, sizeof(double) * p.fb_stride * p.maxc);
    put("basic", p.basic_f0, sizeof(double) * p.fb_stride);
    put("nc", p.nc, sizeof(int));
  }
#endif
}

// ---------------------------------------------------------------------------
// DIO / StoneMask
// ---------------------------------------------------------------------------
struct DioBands {                // cached per (fs, f0_floor, f0_ceil, channels, ratio)
  int fs = 0, ratio = 0;
  double f0_floor = 0, f0_ceil = 0, cpo = 0;
  int nb = 0, cut = 0, max_ntap = 0;
  double band_f0_first = 0;      // boundary_f0_list[0]
  double *d_band_f0 = nullptr, *d_taps = nullptr, *d_lowcut = nullptr;
  int *d_hal = nullptr, *d_off = nullptr;
};

static void prepare_dio_bands(WorldHipContext *c, DioBands &db, int fs, const DioOption *opt, int ratio) {
  if (db.fs == fs && db.ratio == ratio && db.f0_floor == opt->f0_floor && db.f0_ceil == opt->f0_ceil &&
      db.cpo == opt->channels_in_octave && db.nb > 0)
    return;
  const double afs = static_cast<double>(fs) / ratio;
  // dio.cpp:582-586
  const int nb = 1 + static_cast<int>(log(opt->f0_ceil / opt->f0_floor) / kLog2 * opt->channels_in_octave);
  std::vector<double> fb(nb), taps, lowcut;
  std::vector<int> hal(nb), off(nb);
  int max_ntap = 0;
  for (int i = 0; i < nb; ++i) {
    fb[i] = opt->f0_floor * pow(2.0, (i + 1) / opt->channels_in_octave);
    hal[i] = mround(afs / fb[i] / 2.0);                               // dio.cpp:532
    off[i] = (int)taps.size();
    const int len = 4 * hal[i];                                      // NuttallWindow(hal*4), dio.cpp:301
    max_ntap = std::max(max_ntap, len);
    for (int k = 0; k < len; ++k) {
      double t = k / (len - 1.0);
      taps.push_back(0.355768 - 0.487396 * cos(2.0 * kPi * t) + 0.144232 * cos(4.0 * kPi * t) -
                     0.012604 * cos(6.0 * kPi * t));
    }
  }
  // DesignLowCutFilter (dio.cpp:40-53): delta minus a normalised Hanning window, centred
  const int cut = mround(afs / 50.0);
  const int n = 2 * cut + 1;
  lowcut.resize(n);
  for (int i = 1; i <= n; ++i) lowcut[i - 1] = 0.5 - 0.5 * cos(i * 2.0 * kPi / (n + 1));
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += lowcut[i];
  for (int i = 0; i < n; ++i) lowcut[i] = -lowcut[i] / sum;
  lowcut[cut] += 1.0;
  devrt::sync(c->stream);
  ++c->generation;
  if (db.d_band_f0) {
    devrt::dfree(db.d_band_f0); devrt::dfree(db.d_taps); devrt::dfree(db.d_lowcut); devrt::dfree(db.d_hal);
    devrt::dfree(db.d_off);
  }
  db.d_band_f0 = static_cast<double *>(devrt::dmalloc(sizeof(double) * nb));
  db.d_taps = static_cast<double *>(devrt::dmalloc(sizeof(double) * taps.size()));
  db.d_lowcut = static_cast<double *>(devrt::dmalloc(sizeof(double) * n));
  db.d_hal = static_cast<int *>(devrt::dmalloc(sizeof(int) * nb));
  db.d_off = static_cast<int *>(devrt::dmalloc(sizeof(int) * nb));
  devrt::h2d(db.d_band_f0, fb.data(), sizeof(double) * nb, c->stream);
  devrt::h2d(db.d_taps, taps.data(), sizeof(double) * taps.size(), c->stream);
  devrt::h2d(db.d_lowcut, lowcut.data(), sizeof(double) * n, c->stream);
  devrt::h2d(db.d_hal, hal.data(), sizeof(int) * nb, c->stream);
  devrt::h2d(db.d_off, off.data(), sizeof(int) * nb, c->stream);
  devrt::sync(c->stream);
  db.fs = fs; db.ratio = ratio; db.f0_floor = opt->f0_floor; db.f0_ceil = opt->f0_ceil;
  db.cpo = opt->channels_in_octave; db.nb = nb; db.cut = cut; db.max_ntap = max_ntap;
  db.band_f0_first = fb[0];
}

static void run_dio(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                    const DioOption *opt, int f_stride, double *d_tpos, double *d_f0) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  if (!(opt->f0_floor > 0) || !(opt->f0_ceil > opt->f0_floor) || !(opt->frame_period > 0) ||
      !(opt->channels_in_octave > 0))
    fail("bad DioOption");
  DioParams p;
  p.ratio = std::max(std::min(opt->speed, 12), 1);                    // dio.cpp:589
  p.afs = static_cast<double>(fs) / p.ratio;
  DioBands &db = *reinterpret_cast<DioBands *>(c->dio_bands);
  prepare_dio_bands(c, db, fs, opt, p.ratio);
  p.f0_floor = opt->f0_floor; p.f0_ceil = opt->f0_ceil; p.frame_period = opt->frame_period;
  p.allowed_range = opt->allowed_range;
  p.nb = db.nb; p.cut = db.cut; p.max_ntap = db.max_ntap;
  // one tile of the longest channel filter (4 round(fs / ratio / (floor 2^(1/channels)) / 2) taps) with its inputs and outputs is
  // what a workgroup of dio_band_events holds in LDS; the low-cut filter has a route with its taps left in global memory
  if (band_lds_bytes(p.max_ntap) > kLdsPerCu || band_lds_bytes(2 * p.cut + 1) - sizeof(double) * (2 * p.cut + 2) > kLdsPerCu)
    fail("Dio: fs=%d / speed %d with f0_floor %g needs a channel filter of %d taps, more LDS than a CU has "
         "(raise DioOption.speed or f0_floor: <= %d taps)", fs, p.ratio, opt->f0_floor, p.max_ntap, 7400);
  p.vrm = static_cast<int>(0.5 + 1000.0 / opt->frame_period / opt->f0_floor) * 2 + 1;   // dio.cpp:263-264
  std::vector<int> xl(x_length, x_length + n_utt), yl(n_utt), nfr(n_utt), rfft(n_utt);
  int max_x = 0, max_y = 0, max_fr = 0;
  for (int u = 0; u < n_utt; ++u) {
    yl[u] = 1 + xl[u] / p.ratio;                                      // dio.cpp:590
    // the reference's transform length (dio.cpp:592-594, GetSuitableFFTSize = common.cpp:51-54)
    const int span = yl[u] + mround(p.afs / 50.0) * 2 + 1 + 4 * static_cast<int>(1.0 + p.afs / db.band_f0_first / 2.0);
    rfft[u] = static_cast<int>(pow(2.0, static_cast<int>(log(static_cast<double>(span)) / kLog2) + 1.0));
    nfr[u] = frame_count(fs, xl[u], opt->frame_period);
    if (nfr[u] > f_stride) fail("f_stride %d too small for %d frames", f_stride, nfr[u]);
    max_x = std::max(max_x, xl[u]); max_y = std::max(max_y, yl[u]); max_fr = std::max(max_fr, nfr[u]);
  }
  p.y_stride = (max_y + 8) & ~7;
  p.z_stride = (max_y + 2 * p.cut + 8) & ~7;
  p.m_stride = (max_x + 2 * kDecPad + 8) & ~7;
  p.nseg = band_segments(max_y);
  p.ev_cap = max_y / 2 + 2;
  const size_t B = n_utt;
  const size_t seg_list = (size_t)p.nseg * kSegCap;
  size_t need = 5 * pad256(sizeof(int) * B) + pad256(sizeof(double) * B * 4) + pad256(sizeof(double) * B * p.nb * 4);
  need += pad256(sizeof(double) * B * p.m_stride) + pad256(sizeof(double) * B * p.y_stride) +
          pad256(sizeof(double) * B * p.z_stride);
  need += pad256(sizeof(double) * B * p.nb * 4 * seg_list) + pad256(sizeof(int) * B * p.nb * 4 * p.nseg);
  need += pad256(sizeof(double) * B * p.nb * 4 * p.ev_cap) + pad256(sizeof(int) * B * p.nb * 4);
  need += 2 * pad256(sizeof(double) * B * p.nb * f_stride) + 2 * pad256(sizeof(double) * B * f_stride);
  ensure_arena(c, need);
  arena_reset(c);
  CallScope scope(c, 4 * sizeof(int) * n_utt + 512);
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, xl);
  p.b.n_frames = upload(c, nfr);
  p.y_len = upload(c, yl);
  p.ref_fft = upload(c, rfft);
  p.nyq = c->arena.take<double>(B * 4);
  p.quirk = c->arena.take<double>(B * p.nb * 4);
  p.band_f0 = db.d_band_f0; p.band_hal = db.d_hal; p.band_off = db.d_off; p.band_taps = db.d_taps;
  p.lowcut_taps = db.d_lowcut;
  p.fwd = c->arena.take<double>(B * p.m_stride);
  p.y = c->arena.take<double>(B * p.y_stride);
  p.z = c->arena.take<double>(B * p.z_stride);
  p.seg_events = c->arena.take<double>(B * p.nb * 4 * seg_list);
  p.seg_count = c->arena.take<int>(B * p.nb * 4 * p.nseg);
  p.events = c->arena.take<double>(B * p.nb * 4 * p.ev_cap);
  p.ev_count = c->arena.take<int>(B * p.nb * 4);
  p.cand = c->arena.take<double>(B * p.nb * f_stride);
  p.score = c->arena.take<double>(B * p.nb * f_stride);
  p.t1 = c->arena.take<double>(B * f_stride);
  p.t2 = c->arena.take<double>(B * f_stride);
  p.tpos = d_tpos; p.f0 = d_f0;
  launch_dio(p, max_x, max_y, max_fr, c->stream);
}

static void run_stonemask(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                          const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                          const double *d_f0, double *d_refined) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  int max_frames = 0;
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 0 || n_frames[u] > f_stride) fail("n_frames[%d] outside [0, f_stride]", u);
    max_frames = std::max(max_frames, n_frames[u]);
  }
  StoneMaskParams p;
  p.win_cap = 2 * static_cast<int>(1.5 * fs / 40.0 + 1.0) + 4;       // longest window: f0 just above 40 Hz
  if (stonemask_lds_bytes(p.win_cap) > 160 * 1024)
    fail("StoneMask: fs=%d needs a %d-sample window (%zu bytes of LDS > 160 KiB); fs <= 240 kHz supported", fs, p.win_cap,
         stonemask_lds_bytes(p.win_cap));
  ensure_arena(c, 2 * pad256(sizeof(int) * n_utt));
  arena_reset(c);
  CallScope scope(c, 2 * sizeof(int) * n_utt + 256);
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, std::vector<int>(x_length, x_length + n_utt));
  p.b.n_frames = upload(c, std::vector<int>(n_frames, n_frames + n_utt));
  p.tpos = d_tpos; p.f0 = d_f0; p.refined = d_refined;
  p.tab = c->tab;
  launch_stonemask(p, max_frames, c->stream);
}

// ---------------------------------------------------------------------------
// Codec (reference src/codec.cpp): every table depends on (fs, fft_size) only
// ---------------------------------------------------------------------------
struct CodecTables {
  int fs = 0, fft_size = 0;
  int *d_knot_code = nullptr, *d_knot_dec = nullptr, *d_knot_ap = nullptr;
  double *d_frac_code = nullptr, *d_frac_dec = nullptr, *d_frac_ap = nullptr;
  double *d_wc_re = nullptr, *d_wc_im = nullptr, *d_wd_re = nullptr, *d_wd_im = nullptr;
};
struct CodecTableSet { std::vector<CodecTables> sets; };

// interp1's histc + weights (matlabfunctions.cpp:136-176) for fixed knots x and queries xi
static void interp1_tables(const std::vector<double> &x, const std::vector<double> &xi, std::vector<int> *knot,
                           std::vector<double> *frac) {
  const int n = (int)x.size();
  knot->resize(xi.size()); frac->resize(xi.size());
  for (size_t i = 0; i < xi.size(); ++i) {
    int c = 0;
    while (c < n && x[c] <= xi[i]) ++c;
    const int k = c < 1 ? 1 : (c > n - 1 ? n - 1 : c);
    (*knot)[i] = k;
    (*frac)[i] = (xi[i] - x[k - 1]) / (x[k] - x[k - 1]);
  }
}

template <class T> static T *to_device(WorldHipContext *c, const std::vector<T> &v) {
  T *d = static_cast<T *>(devrt::dmalloc(sizeof(T) * std::max<size_t>(v.size(), 1)));
  if (!v.empty()) devrt::h2d(d, v.data(), sizeof(T) * v.size(), c->stream);
  return d;
}

static const CodecTables &codec_tables(WorldHipContext *c, int fs, int fft_size) {
  CodecTableSet *set = static_cast<CodecTableSet *>(c->codec_tables);
  for (const CodecTables &t : set->sets)
    if (t.fs == fs && t.fft_size == fft_size) return t;
  const int md = fft_size / 2, nb = md + 1;
  const double kM0 = 1127.01048, kF0 = 700.0;                          // constantnumbers.h:45-46
  auto mel_of = [&](double f) { return kM0 * log(f / kF0 + 1.0); };    // codec.cpp:59-61
  auto freq_of = [&](double m) { return kF0 * (exp(m / kM0) - 1.0); }; // codec.cpp:66-68
  const double floor_mel = mel_of(40.0), ceil_mel = mel_of(std::min(fs / 2.0, 20000.0));
  std::vector<int> knot;
  std::vector<double> frac;
  CodecTables t;
  t.fs = fs; t.fft_size = fft_size;
  {  // GetParametersForCoding, codec.cpp:162-180
    std::vector<double> mel_axis(md), faxis(nb), wr(md), wi(md);
    for (int i = 0; i < md; ++i) {
      mel_axis[i] = (ceil_mel - floor_mel) * i / md + floor_mel;
      wr[i] = 2.0 * cos(i * kPi / fft_size) / sqrt(static_cast<double>(fft_size));
      wi[i] = 2.0 * sin(i * kPi / fft_size) / sqrt(static_cast<double>(fft_size));
    }
    wr[0] /= sqrt(2.0);
    for (int i = 0; i <= md; ++i) faxis[i] = mel_of(static_cast<double>(i) * fs / fft_size);
    interp1_tables(faxis, mel_axis, &knot, &frac);
    t.d_knot_code = to_device(c, knot); t.d_frac_code = to_device(c, frac);
    t.d_wc_re = to_device(c, wr); t.d_wc_im = to_device(c, wi);
    devrt::sync(c->stream);
  }
  {  // GetParametersForDecoding, codec.cpp:185-207
    std::vector<double> mel_axis(md + 2), faxis(nb), wr(md), wi(md);
    for (int i = 0; i < md; ++i) {
      wr[i] = cos(i * kPi / fft_size) * sqrt(static_cast<double>(fft_size));
      wi[i] = sin(i * kPi / fft_size) * sqrt(static_cast<double>(fft_size));
      mel_axis[i + 1] = freq_of((ceil_mel - floor_mel) * i / md + floor_mel);
    }
    wr[0] /= sqrt(2.0);
    mel_axis[0] = 0;
    mel_axis[md + 1] = fs / 2.0;
    for (int i = 0; i < nb; ++i) faxis[i] = static_cast<double>(i) * fs / fft_size;
    interp1_tables(mel_axis, faxis, &knot, &frac);
    t.d_knot_dec = to_device(c, knot); t.d_frac_dec = to_device(c, frac);
    t.d_wd_re = to_device(c, wr); t.d_wd_im = to_device(c, wi);
    devrt::sync(c->stream);
  }
  {  // DecodeAperiodicity's axes, codec.cpp:242-250
    const int nap = static_cast<int>(std::min(15000.0, fs / 2.0 - 3000.0) / 3000.0);
    std::vector<double> caxis(std::max(nap, 0) + 2), faxis(nb);
    for (int i = 0; i <= nap; ++i) caxis[i] = i * 3000.0;
    caxis[std::max(nap, 0) + 1] = fs / 2.0;
    for (int i = 0; i < nb; ++i) faxis[i] = static_cast<double>(fs) / fft_size * i;
    interp1_tables(caxis, faxis, &knot, &frac);
    t.d_knot_ap = to_device(c, knot); t.d_frac_ap = to_device(c, frac);
    devrt::sync(c->stream);
  }
  set->sets.push_back(t);
  return set->sets.back();
}

static void free_codec_tables(WorldHipContext *c) {
  CodecTableSet *set = static_cast<CodecTableSet *>(c->codec_tables);
  if (!set) return;
  for (CodecTables &t : set->sets) {
    devrt::dfree(t.d_knot_code); devrt::dfree(t.d_knot_dec); devrt::dfree(t.d_knot_ap);
    devrt::dfree(t.d_frac_code); devrt::dfree(t.d_frac_dec); devrt::dfree(t.d_frac_ap);
    devrt::dfree(t.d_wc_re); devrt::dfree(t.d_wc_im); devrt::dfree(t.d_wd_re); devrt::dfree(t.d_wd_im);
  }
  delete set;
  c->codec_tables = nullptr;
}

static int number_of_aperiodicities(int fs) {                          // codec.cpp:212-215
  return static_cast<int>(std::min(15000.0, fs / 2.0 - 3000.0) / 3000.0);
}

void launch_pcm16_to_double(const short *d_pcm, double *d_x, long n, hipStream_t stream);   // pcm.hip
void launch_pcm_bytes_to_double(const unsigned char *d_pcm, double *d_x, long n, int qb, double zero_line,
                                hipStream_t stream);
void launch_double_to_pcm16(const double *d_x, short *d_pcm, long n, hipStream_t stream);

enum CodecOp { kCodeSp, kDecodeSp, kCodeAp, kDecodeAp };
static void run_codec(WorldHipContext *c, CodecOp op, int rows, int fs, int fft_size, int ndim, const double *d_in,
                      double *d_out, size_t in_stride = 0, size_t out_stride = 0) {
  if (rows < 0) fail("negative row count");
  if (rows == 0) return;
  if (fs <= 0) fail("fs must be positive");
  if (!d_in || !d_out) fail("null buffer");
  const int lg = ilog2_exact(fft_size);
  CodecParams p;
  p.in = d_in; p.out = d_out; p.rows = rows; p.fs = fs; p.fft_size = fft_size; p.lg_md = lg - 1; p.ndim = ndim;
  p.in_stride = in_stride; p.out_stride = out_stride;
  p.tab = c->tab;
  p.knot = nullptr; p.frac = nullptr; p.w_re = nullptr; p.w_im = nullptr;
  if (op == kCodeSp || op == kDecodeSp) {
    if (lg < 7 || lg > 13) fail("codec: fft_size %d unsupported (128..8192)", fft_size);
    // the DCT is one real FFT of fft_size/2 points: it has fft_size/4+1 bins (the reference reads
    // never-written plan memory beyond them, codec.cpp:84-86)
    if (ndim < 1 || ndim > fft_size / 4 + 1) fail("codec: number_of_dimensions %d outside [1, fft_size/4+1]", ndim);
  } else {
    p.ndim = number_of_aperiodicities(fs);
    if (p.ndim < 1) fail("codec: fs=%d has no aperiodicity band (needs fs >= 12 kHz)", fs);
    if (3000.0 * p.ndim > fs / 2.0) fail("codec: band centre beyond fs/2");
  }
  const CodecTables &t = codec_tables(c, fs, fft_size);
  switch (op) {
    case kCodeSp:
      p.knot = t.d_knot_code; p.frac = t.d_frac_code; p.w_re = t.d_wc_re; p.w_im = t.d_wc_im;
      launch_code_spectral_envelope(p, c->stream);
      break;
    case kDecodeSp:
      p.knot = t.d_knot_dec; p.frac = t.d_frac_dec; p.w_re = t.d_wd_re; p.w_im = t.d_wd_im;
      launch_decode_spectral_envelope(p, c->stream);
      break;
    case kCodeAp:
      launch_code_aperiodicity(p, c->stream);
      break;
    case kDecodeAp:
      p.knot = t.d_knot_ap; p.frac = t.d_frac_ap;
      launch_decode_aperiodicity(p, c->stream);
      break;
  }
}

// ---------------------------------------------------------------------------
// Synthesis (reference src/synthesis.cpp:339-399)
// ---------------------------------------------------------------------------
static void run_synthesis(WorldHipContext *c, int n_utt, int fs, double frame_period, int fft_size, const int *n_frames,
                          int f_stride, const double *d_f0, const double *d_sp, const double *d_ap,
                          const int *y_length, int y_stride, double *d_y) {
  if (n_utt <= 0) fail("n_utt must be positive");
  if (fs <= 0 || frame_period <= 0) fail("fs and frame_period must be positive");
  if (!d_f0 || !d_sp || !d_ap || !d_y || !n_frames || !y_length) fail("null buffer");
  const int lg = ilog2_exact(fft_size);
  if (lg < 7 || lg > 13) fail("Synthesis: fft_size %d unsupported (128..8192: one pulse's transform must fit LDS)", fft_size);
  int max_y = 0;
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 2 || n_frames[u] > f_stride) fail("n_frames[%d]=%d outside [2, f_stride]", u, n_frames[u]);
    if (y_length[u] < 1 || y_length[u] > y_stride) fail("y_length[%d]=%d outside [1, y_stride]", u, y_length[u]);
    max_y = std::max(max_y, y_length[u]);
  }
  if (c->dc_remover_len != fft_size) {                                 // GetDCRemover, synthesis.cpp:320-335
    std::vector<double> rem(fft_size);
    double dc = 0.0;
    for (int i = 0; i < fft_size / 2; ++i) {
      rem[i] = 0.5 - 0.5 * cos(2.0 * kPi * (i + 1.0) / (1.0 + fft_size));
      rem[fft_size - i - 1] = rem[i];
      dc += rem[i] * 2.0;
    }
    for (int i = 0; i < fft_size / 2; ++i) { rem[i] /= dc; rem[fft_size - i - 1] = rem[i]; }
    devrt::sync(c->stream);
    if (c->d_dc_remover) { devrt::dfree(c->d_dc_remover); ++c->generation; }
    c->d_dc_remover = static_cast<double *>(devrt::dmalloc(sizeof(double) * fft_size));
    devrt::h2d(c->d_dc_remover, rem.data(), sizeof(double) * fft_size, c->stream);
    devrt::sync(c->stream);
    c->dc_remover_len = fft_size;
  }
  SynthParams p;
  p.n_utt = n_utt; p.fs = fs; p.fft_size = fft_size; p.lg_fft = lg;
  p.frame_period = frame_period / 1000.0;
  p.lowest_f0 = fs / fft_size + 1.0;                                   // integer division (synthesis.cpp:361)
  p.f0 = d_f0; p.sp = d_sp; p.ap = d_ap; p.f_stride = f_stride; p.y = d_y; p.y_stride = y_stride;
  p.nblk = (max_y + synth_tile_samples() - 1) / synth_tile_samples();
  // Room for a mean pulse rate of 1200 Hz over the longest utterance unless the caller said otherwise (voiced
  // pulses come at f0, unvoiced ones at 500 Hz).  The pulse count is only known on the device; a call that
  // needs more records the count (need), world_hip_sync / world_hip_synthesis_pulses_dropped report it, and the
  // drop-in Synthesis() -- which waits for its result anyway -- repeats the call with the exact capacity.
  p.pulse_cap = c->synth_pulse_cap > 0 ? c->synth_pulse_cap : static_cast<int>(static_cast<double>(max_y) * 1200.0 / fs) + 16;
  if (!c->d_synth_need) {
    c->d_synth_need = static_cast<int *>(devrt::dmalloc(sizeof(int)));
    devrt::dzero(c->d_synth_need, sizeof(int), c->stream);
  }
  p.need = c->d_synth_need;
  p.resp_stride = lg > 12 ? fft_size + 2 : fft_size;                   // (sy_pulse: the 8192-point pulse keeps its spectrum in its slot)
  const size_t B = n_utt;
  size_t need = 2 * pad256(sizeof(int) * B) + pad256(sizeof(double) * B * y_stride) + pad256(B * y_stride) +
                pad256(sizeof(double) * B * p.nblk) + pad256(sizeof(int) * B * p.nblk) +
                pad256(sizeof(int) * B * p.pulse_cap) + pad256(sizeof(double) * B * p.pulse_cap) + pad256(sizeof(int) * B) +
                pad256(sizeof(double) * B * p.pulse_cap * p.resp_stride);
  ensure_arena(c, need);
  arena_reset(c);
  CallScope scope(c, 2 * sizeof(int) * n_utt + 256);
  p.n_frames = upload(c, std::vector<int>(n_frames, n_frames + n_utt));
  p.y_len = upload(c, std::vector<int>(y_length, y_length + n_utt));
  p.inc = c->arena.take<double>(B * y_stride);
  p.flags = c->arena.take<unsigned char>(B * y_stride);
  p.blk_cnt = c->arena.take<int>(B * p.nblk);
  p.pidx = c->arena.take<int>(B * p.pulse_cap);
  p.pshift = c->arena.take<double>(B * p.pulse_cap);
  p.np = c->arena.take<int>(B);
  p.resp = c->arena.take<double>(B * p.pulse_cap * p.resp_stride);
  p.dc_remover = c->d_dc_remover;
  p.noise = ensure_noise(c, (size_t)max_y + 8);
  p.tab = c->tab;
  launch_synthesis(p, max_y, c->stream);
}

// ---------------------------------------------------------------------------
// Result packing for the multi-GPU exchange (exchange.hip)
// ---------------------------------------------------------------------------
static void run_pack(WorldHipContext *c, bool unpack, int n_utt, const int *n_frames, int f_stride, int nb,
                     const double *d_tpos, const double *d_f0, const double *d_sp, const double *d_ap, long long first_row,
                     double *d_block) {
  if (n_utt <= 0) fail("n_utt must be positive");
  if (!n_frames || !d_tpos || !d_f0 || !d_sp || !d_ap || !d_block) fail("null buffer");
  if (nb < 1 || first_row < 0) fail("bad record shape");
  std::vector<int> nf(n_frames, n_frames + n_utt), off(n_utt);
  long long row = first_row;
  int max_frames = 0;
  for (int u = 0; u < n_utt; ++u) {
    if (nf[u] < 0 || nf[u] > f_stride) fail("n_frames[%d] outside [0, f_stride]", u);
    if (row + nf[u] > 0x7FFFFFFFll) fail("block exceeds 2^31 records");
    off[u] = static_cast<int>(row);
    row += nf[u];
    max_frames = std::max(max_frames, nf[u]);
  }
  if (max_frames == 0) return;
  ensure_arena(c, 2 * pad256(sizeof(int) * n_utt));
  arena_reset(c);
  CallScope scope(c, 2 * sizeof(int) * n_utt + 256);
  PackArgs a;
  a.n_utt = n_utt; a.f_stride = f_stride; a.nb = nb;
  a.n_frames = upload(c, nf);
  a.row_offset = upload(c, off);
  a.tpos = d_tpos; a.f0 = d_f0; a.sp = d_sp; a.ap = d_ap; a.block = d_block;
  launch_pack_rows(a, max_frames, unpack, c->stream);
}

// ---------------------------------------------------------------------------
// CheapTrick, then D4C, of one batch.  (Measured in round 3: CheapTrick on a second stream of the context beside D4C --
// they are independent given F0 -- gains nothing: d4c_frame fills the chip by itself, a lone job stayed at 1.11 ms, and
// twelve jobs in flight dropped from 3.67 to 3.11 M frames/s with 24 streams contending for the hardware queues.)
// ---------------------------------------------------------------------------
static void run_spectral_stages(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                                const int *n_frames, int f_stride, const double *d_tpos, const double *d_f0,
                                const CheapTrickOption *copt, const D4COption *dopt, double *d_sp, double *d_ap,
                                const RowLayout &lay_sp, const RowLayout &lay_ap) {
  ensure_arena(c, cheaptrick_arena_bytes(n_utt, f_stride, copt->fft_size) + d4c_arena_bytes(n_utt, f_stride, fs));
  CallScope scope(c, 6 * sizeof(int) * n_utt + 512);        // ONE upload section for both stages' small arrays
  int max_ct = 0, max_d4c = 0;
  CtParams cp = setup_cheaptrick(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, copt, d_sp, lay_sp,
                                 &max_ct);
  D4cParams dp = setup_d4c(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, copt->fft_size, dopt, d_ap,
                           lay_ap, &max_d4c);
  // both stages' first offset scans depend on F0 alone: one launch serves both (two 1-workgroup kernels per job were two
  // of the narrow launches the twelve-jobs-in-flight mode pays 10-25 us apiece for)
  if (!cp.skip_prepare && !dp.skip_prepare) {
    launch_spectral_prepare(cp, dp, c->stream);
    cp.skip_prepare = 1; dp.skip_prepare |= kD4cSkipScan;
  }
  launch_cheaptrick(cp, max_ct, c->stream);
  launch_d4c(dp, max_d4c, c->stream);
}

// Harvest + CheapTrick + D4C of one batch into the dense arrays of the *_batch calls (world_hip_analyze_batch)
static void run_analyze_dense(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                              const HarvestOption *hopt, const CheapTrickOption *copt, const D4COption *dopt, int f_stride,
                              double *d_tpos, double *d_f0, double *d_sp, double *d_ap) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  std::vector<int> nf(n_utt);
  for (int u = 0; u < n_utt; ++u) {
    nf[u] = frame_count(fs, x_length[u], hopt->frame_period);
    if (nf[u] > f_stride) fail("f_stride %d too small for %d frames", f_stride, nf[u]);
  }
  run_harvest(c, n_utt, fs, d_x, x_stride, x_length, hopt, f_stride, d_tpos, d_f0);
  run_spectral_stages(c, n_utt, fs, d_x, x_stride, x_length, nf.data(), f_stride, d_tpos, d_f0, copt, dopt, d_sp, d_ap,
                      RowLayout(), RowLayout());
}

// ---------------------------------------------------------------------------
// Harvest + CheapTrick + D4C of one batch straight into packed records (include/world_hip.h: world_hip_analyze_packed)
// ---------------------------------------------------------------------------
// code_ndim > 0: CODED records [tpos, f0, mel-cepstrum[code_ndim], band aperiodicity[nap]] (cols = coded_cols): the frame
// kernels code their own rows (cheaptrick.hip: ct_frame's coded epilogue; d4c.hip: d4c_finish) -- no dense row exists anywhere
static void analyze_into_records(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                                 const HarvestOption *hopt, const CheapTrickOption *copt, const D4COption *dopt,
                                 long long first_row, double *d_block, int cols, int code_ndim) {
  const int nb = copt->fft_size / 2 + 1;
  // the record width names the wire format: 2 + 2 nb doubles = f64 spectra, 2 + nb doubles = f32 spectra
  const int wire = code_ndim == 0 && cols == record_cols(copt->fft_size, 1) ? 1 : 0;
  std::vector<int> nf(n_utt), rows(n_utt);
  long long row = first_row;
  int f_stride = 1;
  for (int u = 0; u < n_utt; ++u) {
    nf[u] = frame_count(fs, x_length[u], hopt->frame_period);
    if (row + nf[u] > 0x7FFFFFFFll) fail("block exceeds 2^31 records");
    rows[u] = static_cast<int>(row);
    row += nf[u];
    f_stride = std::max(f_stride, nf[u]);
  }
  // the time axis and F0 stay dense ([n_utt][f_stride]: what CheapTrick and D4C read); they outlive the stages' arenas
  const size_t fr = (size_t)n_utt * f_stride;
  if (c->pk_cap < 2 * fr) {
    devrt::sync(c->stream);
    if (c->d_pk) { devrt::dfree(c->d_pk); ++c->generation; }
    c->pk_cap = 2 * fr + fr / 4;
    c->d_pk = static_cast<double *>(devrt::dmalloc(sizeof(double) * c->pk_cap));
  }
  double *d_tpos = c->d_pk, *d_f0 = c->d_pk + fr;
  run_harvest(c, n_utt, fs, d_x, x_stride, x_length, hopt, f_stride, d_tpos, d_f0);
  RowLayout lay_sp, lay_ap;
  lay_sp.rows = lay_ap.rows = rows.data(); lay_sp.stride = lay_ap.stride = (size_t)cols;
  lay_sp.f32 = lay_ap.f32 = wire == 1;
  const size_t elem = wire == 1 ? sizeof(float) : sizeof(double);
  lay_sp.col_bytes = 2 * sizeof(double);
  lay_ap.col_bytes = 2 * sizeof(double) + (code_ndim > 0 ? sizeof(double) * code_ndim : elem * nb);
  lay_ap.rec = d_block;
  if (code_ndim > 0) {
    const CodecTables &t = codec_tables(c, fs, copt->fft_size);
    lay_sp.code_dims = code_ndim;
    lay_sp.code_knot = t.d_knot_code; lay_sp.code_frac = t.d_frac_code; lay_sp.code_w_re = t.d_wc_re; lay_sp.code_w_im = t.d_wc_im;
    lay_ap.code_dims = number_of_aperiodicities(fs);
  }
  run_spectral_stages(c, n_utt, fs, d_x, x_stride, x_length, nf.data(), f_stride, d_tpos, d_f0, copt, dopt, d_block,
                      d_block, lay_sp, lay_ap);
}
static void run_analyze_packed(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                               const HarvestOption *hopt, const CheapTrickOption *copt, const D4COption *dopt,
                               long long first_row, double *d_block, int cols) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  const int wire = cols == record_cols(copt->fft_size, 1) ? 1 : 0;
  if (cols != record_cols(copt->fft_size, wire))
    fail("analyze_packed: %d columns, fft_size %d needs %d (f64 records) or %d (f32 spectra)", cols, copt->fft_size,
         record_cols(copt->fft_size, 0), record_cols(copt->fft_size, 1));
  if (!d_block) fail("analyze_packed: null block");
  if (first_row < 0) fail("analyze_packed: negative first_row");
  analyze_into_records(c, n_utt, fs, d_x, x_stride, x_length, hopt, copt, dopt, first_row, d_block, cols, 0);
}

// ---------------------------------------------------------------------------
// Harvest + CheapTrick + D4C of one batch into CODED records [tpos, f0, mel-cepstrum[ndim], band aperiodicity[nap]]
// (include/world_hip.h: world_hip_analyze_coded; SURVEY.md 8f.1: the coders "fuse onto K6/K8 outputs" and "shrink the
// all-gather and D2H by 10-17x" -- 31 x at 48 kHz with 60 coefficients: 536 instead of 16 416 bytes per frame).
// CodeSpectralEnvelope / CodeAperiodicity (reference src/codec.cpp:217-236, 268-297) of exactly the analysis a dense call
// returns, computed by the kernels that produce the rows: ct_frame turns its log envelope into the mel-cepstrum in the LDS
// it already holds, d4c_finish evaluates the two bins each band value interpolates.  (Round 5 wrote full 16 KB records to a
// staging block and ran the stand-alone coders over it: one extra HBM write + read per frame and two more launches.)
// ---------------------------------------------------------------------------
static int coded_cols(int fs, int ndim) { return 2 + ndim + number_of_aperiodicities(fs); }
static void run_analyze_coded(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                              const HarvestOption *hopt, const CheapTrickOption *copt, const D4COption *dopt, int ndim,
                              long long first_row, double *d_block, int cols) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  if (!d_block) fail("analyze_coded: null block");
  if (first_row < 0) fail("analyze_coded: negative first_row");
  const int nap = number_of_aperiodicities(fs);
  if (nap < 1) fail("analyze_coded: fs=%d has no aperiodicity band", fs);
  if (3000.0 * nap > fs / 2.0) fail("analyze_coded: band centre beyond fs/2");
  if (ndim < 1 || ndim > copt->fft_size / 4 + 1) fail("analyze_coded: number_of_dimensions %d outside [1, fft_size/4+1]", ndim);
  if (cols != coded_cols(fs, ndim)) fail("analyze_coded: %d columns, %d coefficients at fs=%d need %d", cols, ndim, fs, coded_cols(fs, ndim));
  analyze_into_records(c, n_utt, fs, d_x, x_stride, x_length, hopt, copt, dopt, first_row, d_block, cols, ndim);
}

// ---------------------------------------------------------------------------
// CheapTrick + D4C of the frames [frame_lo, frame_hi) of every utterance, given F0, straight into packed records
// (include/world_hip.h: world_hip_spectral_packed_range): the unit of frame-level sharding ("CheapTrick /
// D4C only, F0 broadcast") and of the drop-in calls' download that overlaps the kernels.
// ---------------------------------------------------------------------------
static void run_spectral_packed_range(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                                      const int *n_frames, int f_stride, const double *d_tpos, const double *d_f0,
                                      const CheapTrickOption *copt, const D4COption *dopt, int frame_lo, int frame_hi,
                                      long long first_row, double *d_block, int cols, bool reuse_offsets) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  if (!n_frames || !d_tpos || !d_f0 || !d_block) fail("null buffer");
  if (frame_lo < 0 || frame_hi < frame_lo || first_row < 0) fail("bad frame range [%d, %d) or first row", frame_lo, frame_hi);
  const int nb = copt->fft_size / 2 + 1;
  const int wire = cols == record_cols(copt->fft_size, 1) ? 1 : 0;
  if (cols != record_cols(copt->fft_size, wire))
    fail("spectral_packed_range: %d columns, fft_size %d needs %d (f64 records) or %d (f32 spectra)", cols, copt->fft_size,
         record_cols(copt->fft_size, 0), record_cols(copt->fft_size, 1));
  // utterance u's frame f goes to row rows[u] + f: its range's first frame lands where the previous utterance's range ended
  std::vector<int> rows(n_utt);
  long long row = first_row;
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 0 || n_frames[u] > f_stride) fail("n_frames[%d] outside [0, f_stride]", u);
    const int lo = std::min(frame_lo, n_frames[u]), hi = std::min(frame_hi, n_frames[u]);
    if (row - lo < -0x7FFFFFFFll || row + (hi - lo) > 0x7FFFFFFFll) fail("block exceeds 2^31 records");
    rows[u] = static_cast<int>(row - lo);
    row += hi - lo;
  }
  RowLayout lay_sp, lay_ap;
  lay_sp.rows = lay_ap.rows = rows.data(); lay_sp.stride = lay_ap.stride = (size_t)cols;
  lay_sp.f32 = lay_ap.f32 = wire == 1;
  const size_t elem = wire == 1 ? sizeof(float) : sizeof(double);
  lay_sp.col_bytes = 2 * sizeof(double); lay_ap.col_bytes = 2 * sizeof(double) + elem * nb;
  lay_ap.rec = d_block;
  lay_sp.frame_lo = lay_ap.frame_lo = frame_lo; lay_sp.frame_hi = lay_ap.frame_hi = frame_hi;
  lay_sp.skip_prepare = lay_ap.skip_prepare = reuse_offsets;
  run_spectral_stages(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, copt, dopt, d_block, d_block,
                      lay_sp, lay_ap);
}

// ---------------------------------------------------------------------------
// Shape limits of the GPU path (the reference has none: it allocates whatever fs asks for).  One function states them,
// the stages and world_hip_check_shape ask it, and the drop-in entries ask it BEFORE any upload or launch.
// what: bit 0 StoneMask, bit 1 CheapTrick (needs fft_size), bit 2 D4C.  Returns nullptr or the reason.
// ---------------------------------------------------------------------------
static std::string shape_limit(int what, int fs, int fft_size) {
  char msg[256];
  if (fs <= 0) return "fs must be positive";
  if (what & 1) {
    const int win_cap = 2 * static_cast<int>(1.5 * fs / 40.0 + 1.0) + 4;
    if (stonemask_lds_bytes(win_cap) > 160 * 1024) {
      snprintf(msg, sizeof msg, "StoneMask: fs=%d needs a %d-sample window, more LDS than a CU has; fs <= 240 kHz supported", fs, win_cap);
      return msg;
    }
  }
  if (what & 2) {
    int lg = 0;
    while ((1 << lg) < fft_size) ++lg;
    if ((1 << lg) != fft_size || lg < 7 || lg > 13) {
      snprintf(msg, sizeof msg, "CheapTrick: fft_size %d unsupported (a power of two, 128..8192: one frame must fit LDS; fs <= 192 kHz at the default f0 floor)", fft_size);
      return msg;
    }
  }
  if (what & 4) {
    const int fft_d4c = std::max(d4c_internal_fft(fs), static_cast<int>(pow(2.0, 1.0 + static_cast<int>(log(3.0 * fs / 40.0 + 1) / kLog2))));
    if (fft_d4c > 16384) {
      snprintf(msg, sizeof msg, "D4C: fs=%d needs an internal FFT of %d > 16384 points (LDS budget); fs <= 192 kHz supported", fs, fft_d4c);
      return msg;
    }
    if (fs < 15800) {
      snprintf(msg, sizeof msg, "D4C: fs=%d is below the 15.8 kHz the reference's LoveTrain band edges require (the reference reads out of bounds there)", fs);
      return msg;
    }
  }
  return std::string();
}

// ---------------------------------------------------------------------------
// error plumbing for the C ABI
// ---------------------------------------------------------------------------
// A context's allocations and launches belong to ITS device: a thread that drives several GPUs calls in
// with whatever device it last selected, so select the context's for the duration of the call and put the
// caller's back afterwards.
struct DeviceScope {
  int before, wanted;
  explicit DeviceScope(int device) : before(devrt::current_device()), wanted(device) {
    if (before != wanted) devrt::set_device(wanted);
  }
  ~DeviceScope() {
    if (before != wanted) {
      try { devrt::set_device(before); } catch (...) {}
    }
  }
};
template <class F> static int guarded(WorldHipContext *c, F f) {
  if (!c) { g_last_error = "null context"; return 2; }
  std::lock_guard<std::mutex> g(c->lock);
  try {
    DeviceScope on_device(c->device);
    f();
    return 0;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return 1;
  }
}

}  // namespace world_hip

using namespace world_hip;

// ===========================================================================
// Part 2 of include/world_hip.h
// ===========================================================================
extern "C" {

WorldHipContext *world_hip_create(int device, void *stream) {
  try {
    DeviceScope on_device(device);       // the caller's current device is left as it was
    WorldHipContext *c = new WorldHipContext;
    c->device = device;
    c->stream = static_cast<hipStream_t>(stream);
    std::vector<double2> tw(kTwAlloc);
    build_twiddles(tw.data());
    std::vector<uint4> jump((size_t)kJumpLevels * kJumpStride);
    build_jump_tables(jump.data());
    double2 *d_tw = static_cast<double2 *>(devrt::dmalloc(sizeof(double2) * tw.size()));
    uint4 *d_jump = static_cast<uint4 *>(devrt::dmalloc(sizeof(uint4) * jump.size()));
    devrt::h2d(d_tw, tw.data(), sizeof(double2) * tw.size(), c->stream);
    devrt::h2d(d_jump, jump.data(), sizeof(uint4) * jump.size(), c->stream);
    devrt::sync(c->stream);
    c->tab.tw = d_tw;
    c->tab.jump = d_jump;
    noise_table_retain(device);
    c->dio_bands = new DioBands;
    c->codec_tables = new CodecTableSet;
    return c;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return nullptr;
  }
}

void world_hip_destroy(WorldHipContext *c) {
  if (!c) return;
  // the device's shared randn table is reference counted: give the reference back whatever happens below
  try { DeviceScope on_device(c->device); devrt::sync(c->stream); } catch (...) {}
  try { DeviceScope on_device(c->device); noise_table_release(c->device); } catch (...) {}
  try {
    DeviceScope on_device(c->device);    // the context's allocations belong to ITS device, not to the caller's current one
    devrt::sync(c->stream);
    devrt::dfree(const_cast<double2 *>(c->tab.tw));
    devrt::dfree(const_cast<uint4 *>(c->tab.jump));
    if (c->arena.base) devrt::dfree(c->arena.base);
    if (c->d_nuttall) devrt::dfree(c->d_nuttall);
    if (c->d_ap_frac) { devrt::dfree(c->d_ap_frac); devrt::dfree(c->d_ap_knot); }
    if (c->d_pk) devrt::dfree(c->d_pk);
    if (c->d_xin) devrt::dfree(c->d_xin);
    if (c->h_xin) devrt::hfree_pinned(c->h_xin);
    if (c->xstream) { devrt::sync(c->xstream); devrt::stream_destroy(c->xstream); }
    if (c->dio_bands) {
      DioBands *db = static_cast<DioBands *>(c->dio_bands);
      if (db->d_band_f0) {
        devrt::dfree(db->d_band_f0); devrt::dfree(db->d_taps); devrt::dfree(db->d_lowcut); devrt::dfree(db->d_hal);
        devrt::dfree(db->d_off);
      }
      delete db;
    }
    free_codec_tables(c);
    if (c->d_dc_remover) devrt::dfree(c->d_dc_remover);
    if (c->d_synth_need) devrt::dfree(c->d_synth_need);
    drop_small_arrays(c);
    HarvestBands &hb = c->bands;
    if (hb.d_band_f0) { devrt::dfree(hb.d_band_f0); devrt::dfree(hb.d_taps); devrt::dfree(hb.d_half); devrt::dfree(hb.d_off); }
    if (hb.d_win_tab) devrt::dfree(hb.d_win_tab);
    if (hb.d_spec) devrt::dfree(hb.d_spec);
    if (c->xchg_ready) devrt::event_destroy(c->xchg_ready);
    if (c->xchg_done) devrt::event_destroy(c->xchg_done);
  } catch (...) {
  }
  delete c;
}

int world_hip_set_hint(WorldHipContext *ctx, int hint) {
  if (!ctx) return -1;
  ctx->hint = hint;
  return 0;
}
int world_hip_abi_version(void) { return WORLD_HIP_ABI_VERSION; }
const char *world_hip_last_error(void) { return g_last_error.c_str(); }

// pulses a synthesis call since the last check had no room for (0 = none); synchronises, clears the record
static int synthesis_need(WorldHipContext *c) {
  if (!c->d_synth_need) return 0;
  int need = 0;
  devrt::d2h(&need, c->d_synth_need, sizeof(int), c->stream);
  devrt::sync(c->stream);
  if (need > 0) devrt::dzero(c->d_synth_need, sizeof(int), c->stream);
  return need;
}

int world_hip_sync(WorldHipContext *c) {
  return guarded(c, [&] {
    devrt::sync(c->stream);
    const int need = synthesis_need(c);
    if (need > 0)
      fail("Synthesis dropped pulses: an utterance has %d, more than the capacity in force; call "
           "world_hip_set_synthesis_pulse_capacity(ctx, %d) and repeat the call", need, need);
  });
}

int world_hip_set_synthesis_pulse_capacity(WorldHipContext *c, int pulses_per_utterance) {
  return guarded(c, [&] {
    if (pulses_per_utterance < 0) fail("negative capacity");
    c->synth_pulse_cap = pulses_per_utterance;
  });
}

int world_hip_synthesis_pulses_dropped(WorldHipContext *c, int *needed) {
  return guarded(c, [&] {
    const int need = synthesis_need(c);
    if (needed) *needed = need;
  });
}

unsigned long long world_hip_workspace_bytes(WorldHipContext *c) {
  return c ? c->arena.cap : 0;
}

// the device's shared randn table (live + superseded generations), see rng.h
unsigned long long world_hip_noise_table_bytes(WorldHipContext *c) { return c ? noise_table_bytes(c->device) : 0; }

// wall-clock milliseconds this process has spent building + verifying the device's randn tables (cold-start cost)
double world_hip_noise_table_build_ms(WorldHipContext *c, int *builds) {
  if (!c) { if (builds) *builds = 0; return 0.0; }
  return noise_table_build_ms(c->device, builds);
}

// re-reduce the live randn table on the device and compare it with the host's sums (diagnostic; synchronises)
int world_hip_verify_tables(WorldHipContext *c) {
  return guarded(c, [&] { noise_table_verify(c->device, c->stream); });
}

// per-kernel HIP-event timing (used by bench.py for the roofline figure)
void world_hip_profile_enable(int on) {
#ifndef WORLD_EMU
  devrt::prof_enable(on != 0);
#else
  (void)on;
#endif
}
// Waits for the recorded kernels and writes "kernel_name milliseconds\n" lines into buf
// (truncated to cap-1 bytes).  Returns the untruncated length.
int world_hip_profile_collect(char *buf, int cap) {
#ifndef WORLD_EMU
  try {
    std::string s = devrt::prof_collect();
    if (buf && cap > 0) {
      size_t n = std::min((size_t)cap - 1, s.size());
      memcpy(buf, s.data(), n);
      buf[n] = 0;
    }
    return (int)s.size();
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return -1;
  }
#else
  if (buf && cap > 0) buf[0] = 0;
  return 0;
#endif
}

int world_hip_harvest_batch(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                            const int *x_length, const HarvestOption *option, int f_stride, double *d_tpos,
                            double *d_f0) {
  return guarded(c, [&] { run_harvest(c, n_utt, fs, d_x, x_stride, x_length, option, f_stride, d_tpos, d_f0); });
}

int world_hip_dio_batch(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                        const int *x_length, const DioOption *option, int f_stride, double *d_tpos, double *d_f0) {
  return guarded(c, [&] { run_dio(c, n_utt, fs, d_x, x_stride, x_length, option, f_stride, d_tpos, d_f0); });
}

int world_hip_stonemask_batch(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                              const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                              const double *d_f0, double *d_refined) {
  return guarded(c, [&] {
    run_stonemask(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, d_refined);
  });
}

int world_hip_cheaptrick_batch(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                               const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                               const double *d_f0, const CheapTrickOption *option, double *d_sp) {
  return guarded(c, [&] {
    run_cheaptrick(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, option, d_sp);
  });
}

int world_hip_d4c_batch(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                        const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                        const double *d_f0, int fft_size, const D4COption *option, double *d_ap) {
  return guarded(c, [&] {
    run_d4c(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, fft_size, option, d_ap);
  });
}

int world_hip_analyze_batch(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                            const HarvestOption *harvest_option, const CheapTrickOption *cheaptrick_option,
                            const D4COption *d4c_option, int f_stride, double *d_tpos, double *d_f0, double *d_spectrogram,
                            double *d_aperiodicity) {
  return guarded(c, [&] {
    run_analyze_dense(c, n_utt, fs, d_x, x_stride, x_length, harvest_option, cheaptrick_option, d4c_option, f_stride, d_tpos,
                      d_f0, d_spectrogram, d_aperiodicity);
  });
}

int world_hip_record_columns(int fft_size, int wire) {
  if (fft_size < 2 || (wire != 0 && wire != 1)) return -1;
  return record_cols(fft_size, wire);
}

int world_hip_analyze_packed(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                             const HarvestOption *harvest_option, const CheapTrickOption *cheaptrick_option,
                             const D4COption *d4c_option, long long first_row, double *d_block, int cols) {
  return guarded(c, [&] {
    run_analyze_packed(c, n_utt, fs, d_x, x_stride, x_length, harvest_option, cheaptrick_option, d4c_option, first_row,
                       d_block, cols);
  });
}

int world_hip_coded_columns(int fs, int number_of_dimensions) {
  if (fs <= 0 || number_of_dimensions < 1 || world_hip::number_of_aperiodicities(fs) < 1) return -1;
  return coded_cols(fs, number_of_dimensions);
}

int world_hip_analyze_coded(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                            const HarvestOption *harvest_option, const CheapTrickOption *cheaptrick_option,
                            const D4COption *d4c_option, int number_of_dimensions, long long first_row, double *d_block,
                            int cols) {
  return guarded(c, [&] {
    run_analyze_coded(c, n_utt, fs, d_x, x_stride, x_length, harvest_option, cheaptrick_option, d4c_option,
                      number_of_dimensions, first_row, d_block, cols);
  });
}

int world_hip_spectral_packed_range(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                                     const int *n_frames, int f_stride, const double *d_tpos, const double *d_f0,
                                     const CheapTrickOption *cheaptrick_option, const D4COption *d4c_option, int frame_lo,
                                     int frame_hi, int reuse_offsets, long long first_row, double *d_block, int cols) {
  return guarded(c, [&] {
    run_spectral_packed_range(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, cheaptrick_option,
                              d4c_option, frame_lo, frame_hi, first_row, d_block, cols, reuse_offsets != 0);
  });
}

// one stage, one frame range, dense rows (the rows of frames outside the range are not touched)
int world_hip_cheaptrick_batch_range(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                                     const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                                     const double *d_f0, const CheapTrickOption *option, int frame_lo, int frame_hi,
                                     int reuse_offsets, double *d_sp) {
  return guarded(c, [&] {
    RowLayout lay;
    lay.frame_lo = frame_lo; lay.frame_hi = frame_hi; lay.skip_prepare = reuse_offsets != 0;
    run_cheaptrick(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, option, d_sp, lay);
  });
}
int world_hip_d4c_batch_range(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                              const int *n_frames, int f_stride, const double *d_tpos, const double *d_f0, int fft_size,
                              const D4COption *option, int frame_lo, int frame_hi, int reuse_offsets, double *d_ap) {
  return guarded(c, [&] {
    RowLayout lay;
    lay.frame_lo = frame_lo; lay.frame_hi = frame_hi; lay.skip_prepare = reuse_offsets != 0;
    run_d4c(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, fft_size, option, d_ap, lay);
  });
}

// HIP graphs: every batched call enqueued on the context between begin and end -- a whole
// Harvest + CheapTrick + D4C job is ~45 launches -- becomes ONE replayable graph bound to the buffers it was captured with.
// The shapes must have run once before (workspace, tables and the small per-call arrays are then resident and a call
// neither allocates nor copies nor waits); a replay costs the host one launch.
// A graph bakes in raw device pointers to memory the context owns (arena, small-array slabs, d_pk, cached tables).  The
// handle remembers the context's generation at capture; a later eager call that had to free or regrow any of that memory
// bumps the generation, and a replay of the stale graph is REFUSED instead of reading freed memory.
struct GraphHandle {
  void *exec;
  WorldHipContext *ctx;
  unsigned long long generation;
};
int world_hip_graph_begin(WorldHipContext *c) {
  return guarded(c, [&] { devrt::graph_begin(c->stream); });
}
int world_hip_graph_end(WorldHipContext *c, void **graph) {
  return guarded(c, [&] {
    if (!graph) fail("null graph handle");
    *graph = nullptr;
    void *exec = devrt::graph_end(c->stream);
    *graph = new GraphHandle{exec, c, c->generation};
  });
}
int world_hip_graph_launch(WorldHipContext *c, void *graph) {
  return guarded(c, [&] {
    if (!graph) fail("null graph");
    const GraphHandle *g = static_cast<const GraphHandle *>(graph);
    if (g->ctx != c) fail("graph_launch: the graph was captured on another context");
    if (g->generation != c->generation)
      fail("graph_launch: stale graph -- the context reallocated memory the graph points into since it was captured "
           "(a larger batch, a new option set or shape ran eagerly); capture it again");
    devrt::graph_launch(g->exec, c->stream);
  });
}
int world_hip_graph_destroy(void *graph) {
  if (!graph) return 0;
  GraphHandle *g = static_cast<GraphHandle *>(graph);
  try { devrt::graph_destroy(g->exec); delete g; return 0; } catch (const std::exception &e) { delete g; g_last_error = e.what(); return 1; }
}

// 0: every stage of the analysis path supports (fs, cheaptrick_fft_size); 1: it does not, and why (<= cap bytes) says which
// stage and limit.  Pure host arithmetic: callers (and the drop-in symbols) ask before any GPU work.
int world_hip_check_shape(int fs, int cheaptrick_fft_size, char *why, int cap) {
  const std::string r = shape_limit(7, fs, cheaptrick_fft_size);
  if (why && cap > 0) { snprintf(why, (size_t)cap, "%s", r.c_str()); }
  return r.empty() ? 0 : 1;
}

int world_hip_synthesis_batch(WorldHipContext *c, int n_utt, int fs, double frame_period, int fft_size,
                              const int *n_frames, int f_stride, const double *d_f0, const double *d_spectrogram,
                              const double *d_aperiodicity, const int *y_length, int y_stride, double *d_y) {
  return guarded(c, [&] {
    run_synthesis(c, n_utt, fs, frame_period, fft_size, n_frames, f_stride, d_f0, d_spectrogram, d_aperiodicity,
                  y_length, y_stride, d_y);
  });
}

// what box is this? (machine_probe.hip; synchronous, ~50 ms, allocates and frees 2 GB)
int world_hip_probe_machine(WorldHipContext *c, double *values, int n_values) {
  return guarded(c, [&] {
    if (!values || n_values < world_hip::kMachineProbeValues) fail("probe_machine: room for %d values needed", world_hip::kMachineProbeValues);
    devrt::sync(c->stream);
    world_hip::run_machine_probe(values, c->stream);
  });
}

// fft.h in isolation (fft_probe.hip): `batch` real transforms of 2^lg_n points, one workgroup each
static void run_fft_probe(WorldHipContext *c, bool inverse, int lg_n, int max_lr, int threads, int static_plan,
                          long long batch, const void *d_in, void *d_out) {
  if (lg_n < 8 || lg_n > kTwLog2) fail("probe: 2^%d points unsupported (256 .. %d)", lg_n, kTwN);   // (16384: 147 KB of LDS)
  if (max_lr != 3 && max_lr != 4) fail("probe: max_lr must be 3 (radix-8 plan) or 4 (radix-16 plan)");
  if (threads == 0) threads = std::max(64, (1 << lg_n) >> (max_lr + 1));      // one butterfly per thread and stage
  if (threads < 64 || threads > 1024 || threads % 64) fail("probe: bad workgroup size %d", threads);
  if (batch < 0 || batch > 0x7FFFFFFFll) fail("probe: bad batch");
  if (batch == 0) return;
  if (!d_in || !d_out) fail("null buffer");
  if (static_plan && !fft_probe_has_static(lg_n, max_lr)) fail("probe: no compile-time plan for 2^%d points, max_lr %d", lg_n, max_lr);
  launch_fft_probe(inverse, lg_n, max_lr, threads, static_plan != 0, (long)batch, d_in, d_out, c->tab, c->stream);
}
int world_hip_probe_rfft(WorldHipContext *c, int lg_n, int max_lr, int threads, int static_plan, long long batch,
                         const double *d_in, double *d_spectrum) {
  return guarded(c, [&] { run_fft_probe(c, false, lg_n, max_lr, threads, static_plan, batch, d_in, d_spectrum); });
}
int world_hip_probe_irfft(WorldHipContext *c, int lg_n, int max_lr, int threads, int static_plan, long long batch,
                          const double *d_spectrum, double *d_out) {
  return guarded(c, [&] { run_fft_probe(c, true, lg_n, max_lr, threads, static_plan, batch, d_spectrum, d_out); });
}

int world_hip_pack_results(WorldHipContext *c, int n_utt, const int *n_frames, int f_stride, int bins,
                           const double *d_tpos, const double *d_f0, const double *d_spectrogram,
                           const double *d_aperiodicity, long long first_row, double *d_block) {
  return guarded(c, [&] {
    run_pack(c, false, n_utt, n_frames, f_stride, bins, d_tpos, d_f0, d_spectrogram, d_aperiodicity, first_row, d_block);
  });
}

int world_hip_unpack_results(WorldHipContext *c, int n_utt, const int *n_frames, int f_stride, int bins,
                             const double *d_block, long long first_row, double *d_tpos, double *d_f0,
                             double *d_spectrogram, double *d_aperiodicity) {
  return guarded(c, [&] {
    run_pack(c, true, n_utt, n_frames, f_stride, bins, d_tpos, d_f0, d_spectrogram, d_aperiodicity, first_row,
             const_cast<double *>(d_block));
  });
}

// One process, n_dev contexts (normally one per GPU): every context's packed block to every context.
// Fully stream-ordered: destination d waits (on its own stream) for source s's block to be complete, pulls it
// with a peer copy, and source s's stream in turn waits until every destination has pulled -- so the caller may
// reuse the source blocks in stream order without ever synchronising the host.
int world_hip_allgather_blocks(int n_dev, WorldHipContext *const *ctxs, const double *const *d_src, const long long *rows,
                               int cols, double *const *d_dst) {
  try {
    if (n_dev <= 0 || !ctxs || !d_src || !rows || !d_dst || cols <= 0) fail("bad arguments");
    std::vector<long long> first(n_dev + 1, 0);
    for (int d = 0; d < n_dev; ++d) {
      if (!ctxs[d] || rows[d] < 0 || (rows[d] > 0 && !d_src[d]) || !d_dst[d]) fail("bad block %d", d);
      for (int e = 0; e < d; ++e)
        if (ctxs[e] == ctxs[d]) fail("context %d listed twice", d);
      first[d + 1] = first[d] + rows[d];
    }
    std::vector<std::unique_lock<std::mutex>> locks;
    {                                                       // a fixed global order keeps concurrent callers deadlock-free
      std::vector<WorldHipContext *> order(ctxs, ctxs + n_dev);
      std::sort(order.begin(), order.end());
      for (WorldHipContext *c : order) locks.emplace_back(c->lock);
    }
    for (int s = 0; s < n_dev; ++s) {
      WorldHipContext *c = ctxs[s];
      DeviceScope on_device(c->device);
      if (!c->xchg_ready) { c->xchg_ready = devrt::event_create(); c->xchg_done = devrt::event_create(); }
      for (int d = 0; d < n_dev; ++d) devrt::enable_peer_access(c->device, ctxs[d]->device);
      devrt::event_record(c->xchg_ready, c->stream);
    }
    for (int d = 0; d < n_dev; ++d) {
      WorldHipContext *c = ctxs[d];
      DeviceScope on_device(c->device);
      // remote blocks first, starting with the neighbour: destinations then pull from different sources at any moment
      for (int k = 1; k <= n_dev; ++k) {
        const int s = (d + k) % n_dev;
        if (s != d) devrt::stream_wait_event(c->stream, ctxs[s]->xchg_ready);
        devrt::peer_copy(d_dst[d] + first[s] * cols, c->device, d_src[s], ctxs[s]->device,
                         sizeof(double) * (size_t)rows[s] * cols, c->stream);
      }
      devrt::event_record(c->xchg_done, c->stream);
    }
    for (int s = 0; s < n_dev; ++s) {
      DeviceScope on_device(ctxs[s]->device);
      for (int d = 0; d < n_dev; ++d)
        if (d != s) devrt::stream_wait_event(ctxs[s]->stream, ctxs[d]->xchg_done);
    }
    return 0;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return 1;
  }
}

// Sub-batch sizes of a device's share of n utterances (world_amd/distributed.py: chunk_sizes states the same schedule):
// full sub-batches of sb, and the LAST one tapered into halves (sb/2, sb/4, sb/4) -- sub-batch k's exchange runs under the
// analysis of k + 1, so only the last exchange is exposed, and tapering shrinks it from 1/4 of a 128-utterance share to
// 1/16.  Tails of fewer than 8 utterances stay whole (smaller batches stop filling the chip).
static std::vector<int> chunk_sizes(int n, int sb) {
  std::vector<int> out;
  sb = std::max(1, sb);
  while (n > sb) { out.push_back(sb); n -= sb; }
  if (n >= 8 && !getenv("WORLD_HIP_NO_TAPER")) {
    const int a = (n + 1) / 2, b = (n - a + 1) / 2, c = n - a - b;
    out.push_back(a);
    if (b > 0) out.push_back(b);
    if (c > 0) out.push_back(c);
  } else if (n > 0) {
    out.push_back(n);
  }
  return out;
}

// One PROCESS, n_dev GPUs ("one host thread + stream per GPU"): the C/C++ counterpart of
// world_amd/distributed.py.  Utterances (host memory) are partitioned longest-first over the contexts' devices; a host
// thread per device uploads its share in sub-batches (pinned double buffer), analyses each straight into packed records
// (world_hip_analyze_packed) at the sub-batch's rows of ITS copy of the job's one record block, and as soon as a
// sub-batch is enqueued every other device pulls those rows over xGMI on its own exchange stream (peer copies by the
// copy engines) -- the exchange of sub-batch k runs under the analysis of sub-batch k + 1.  Every device ends up with
// ALL records at identical offsets.  where[3 i .. 3 i + 2] = {device index, first row, n_frames} of utterance i.
// Returns when every device's block is complete (the streams are synchronised: the inputs are host memory anyway).
int world_hip_analyze_sharded(int n_dev, WorldHipContext *const *ctxs, int n_utt, int fs, const double *const *x,
                              const int *x_length, const HarvestOption *hopt, const CheapTrickOption *copt,
                              const D4COption *dopt, int sub_batch, double *const *d_blocks, long long rows_capacity, int cols,
                              long long *where) {
  try {
    if (n_dev <= 0 || !ctxs || n_utt < 0 || !x_length || !hopt || !copt || !dopt || !d_blocks || !where) fail("bad arguments");
    if (n_utt > 0 && !x) fail("null input");
    if (cols != record_cols(copt->fft_size, 0) && cols != record_cols(copt->fft_size, 1))
      fail("analyze_sharded: %d columns do not fit fft_size %d (%d for f64 records, %d for f32 spectra)", cols, copt->fft_size,
           record_cols(copt->fft_size, 0), record_cols(copt->fft_size, 1));
    for (int d = 0; d < n_dev; ++d) {
      if (!ctxs[d] || !d_blocks[d]) fail("bad device entry %d", d);
      for (int e = 0; e < d; ++e) {
        if (ctxs[e] == ctxs[d]) fail("context %d listed twice", d);
        if (d_blocks[e] == d_blocks[d]) fail("devices %d and %d share a block", e, d);
      }
    }
    const int sb = std::max(1, sub_batch);
    // greedy longest-first partition (world_amd/distributed.py: partition), utterances of a device in index order
    std::vector<int> order(n_utt);
    for (int i = 0; i < n_utt; ++i) {
      if (x_length[i] <= 0 || !x[i]) fail("utterance %d is empty", i);
      order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return x_length[a] > x_length[b]; });
    std::vector<std::vector<int>> part(n_dev);
    std::vector<long long> load(n_dev, 0);
    for (int i : order) {
      int r = 0;
      for (int d = 1; d < n_dev; ++d) if (load[d] < load[r]) r = d;
      part[r].push_back(i);
      load[r] += x_length[i];
    }
    // rows: device after device, sub-batch after sub-batch, utterance after utterance
    struct Chunk { int dev, lo, hi; long long first, rows; int max_len; };
    std::vector<Chunk> chunks;
    long long row = 0;
    for (int r = 0; r < n_dev; ++r) {
      std::sort(part[r].begin(), part[r].end());
      size_t lo = 0;
      for (int take : chunk_sizes((int)part[r].size(), sb)) {
        Chunk c{r, (int)lo, (int)(lo + take), row, 0, 0};
        lo += take;
        for (int j = c.lo; j < c.hi; ++j) {
          const int i = part[r][j], nf = frame_count(fs, x_length[i], hopt->frame_period);
          where[3 * i] = r; where[3 * i + 1] = row; where[3 * i + 2] = nf;
          row += nf;
          c.max_len = std::max(c.max_len, x_length[i]);
        }
        c.rows = row - c.first;
        chunks.push_back(c);
      }
    }
    if (row > rows_capacity) fail("analyze_sharded: the job has %lld records, the blocks hold %lld", row, rows_capacity);
    if (n_utt == 0) return 0;
    std::vector<std::unique_lock<std::mutex>> locks;
    {                                                       // a fixed global order keeps concurrent callers deadlock-free
      std::vector<WorldHipContext *> ord(ctxs, ctxs + n_dev);
      std::sort(ord.begin(), ord.end());
      for (WorldHipContext *c : ord) locks.emplace_back(c->lock);
    }
    // per-device set-up on the calling thread: exchange stream, peer access, one event per sub-batch
    std::vector<void *> done(chunks.size(), nullptr);
    for (int d = 0; d < n_dev; ++d) {
      WorldHipContext *c = ctxs[d];
      DeviceScope on_device(c->device);
      if (!c->xstream) c->xstream = devrt::stream_create();
      for (int e = 0; e < n_dev; ++e) devrt::enable_peer_access(c->device, ctxs[e]->device);
    }
    for (size_t k = 0; k < chunks.size(); ++k) {
      DeviceScope on_device(ctxs[chunks[k].dev]->device);
      done[k] = devrt::event_create();
    }
    std::vector<std::string> errors(n_dev);
    std::mutex xlock;                                       // enqueues on another device's exchange stream, one thread at a time
    auto worker = [&](int r) {
      WorldHipContext *c = ctxs[r];
      try {
        devrt::set_device(c->device);
        size_t need = 0;
        for (const Chunk &ch : chunks) if (ch.dev == r) need = std::max(need, (size_t)(ch.hi - ch.lo) * ch.max_len);
        if (need > c->xin_cap) {
          devrt::sync(c->stream);
          if (c->d_xin) devrt::dfree(c->d_xin);
          if (c->h_xin) devrt::hfree_pinned(c->h_xin);
          c->xin_cap = need + need / 8;
          c->d_xin = static_cast<double *>(devrt::dmalloc(sizeof(double) * 2 * c->xin_cap));
          c->h_xin = static_cast<double *>(devrt::hmalloc_pinned(sizeof(double) * 2 * c->xin_cap));
        }
        void *staged[2] = {devrt::event_create(), devrt::event_create()};   // half h of the staging buffers is free again
        int turn = 0;
        for (size_t k = 0; k < chunks.size(); ++k) {
          const Chunk &ch = chunks[k];
          if (ch.dev != r) continue;
          const int b = ch.hi - ch.lo, half = turn & 1;
          if (turn >= 2) devrt::event_sync(staged[half]);  // the analysis that read this half two sub-batches ago is done
          double *h = c->h_xin + (size_t)half * c->xin_cap, *dx = c->d_xin + (size_t)half * c->xin_cap;
          std::vector<int> xl(b);
          for (int j = 0; j < b; ++j) {
            const int i = part[r][ch.lo + j];
            xl[j] = x_length[i];
            memcpy(h + (size_t)j * ch.max_len, x[i], sizeof(double) * x_length[i]);
            if (x_length[i] < ch.max_len) memset(h + (size_t)j * ch.max_len + x_length[i], 0, sizeof(double) * (ch.max_len - x_length[i]));
          }
          devrt::h2d(dx, h, sizeof(double) * (size_t)b * ch.max_len, c->stream);
          run_analyze_packed(c, b, fs, dx, ch.max_len, xl.data(), hopt, copt, dopt, ch.first, d_blocks[r], cols);
          devrt::event_record(staged[half], c->stream);
          devrt::event_record(done[k], c->stream);
          ++turn;
          // every other device pulls this sub-batch's rows on its exchange stream, the neighbour first
          std::lock_guard<std::mutex> g(xlock);
          for (int s = 1; s < n_dev; ++s) {
            const int d = (r + s) % n_dev;
            DeviceScope on_dst(ctxs[d]->device);
            devrt::stream_wait_event(ctxs[d]->xstream, done[k]);
            devrt::peer_copy(d_blocks[d] + ch.first * cols, ctxs[d]->device, d_blocks[r] + ch.first * cols, c->device,
                             sizeof(double) * (size_t)ch.rows * cols, ctxs[d]->xstream);
          }
        }
        devrt::sync(c->stream);
        devrt::event_destroy(staged[0]); devrt::event_destroy(staged[1]);
      } catch (const std::exception &e) {
        errors[r] = e.what();
      }
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < n_dev; ++r) threads.emplace_back(worker, r);
    const int before = devrt::current_device();
    worker(0);                                              // the calling thread drives device 0
    for (std::thread &t : threads) t.join();
    for (int d = 0; d < n_dev; ++d) {
      try { devrt::set_device(ctxs[d]->device); devrt::sync(ctxs[d]->xstream); } catch (const std::exception &e) { if (errors[d].empty()) errors[d] = e.what(); }
    }
    for (size_t k = 0; k < chunks.size(); ++k) {
      try { devrt::set_device(ctxs[chunks[k].dev]->device); devrt::event_destroy(done[k]); } catch (...) {}
    }
    try { devrt::set_device(before); } catch (...) {}
    for (int d = 0; d < n_dev; ++d)
      if (!errors[d].empty()) fail("device %d: %s", d, errors[d].c_str());
    return 0;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return 1;
  }
}

int world_hip_pcm16_to_double(WorldHipContext *c, long long n, const short *d_pcm, double *d_x) {
  return guarded(c, [&] {
    if (n < 0) fail("negative sample count");
    if (n > 0 && (!d_pcm || !d_x)) fail("null buffer");
    if (n > 0) launch_pcm16_to_double(d_pcm, d_x, (long)n, c->stream);
  });
}

int world_hip_code_spectral_envelope(WorldHipContext *c, int rows, int fs, int fft_size, int number_of_dimensions,
                                     const double *d_spectrogram, double *d_coded) {
  return guarded(c, [&] { run_codec(c, kCodeSp, rows, fs, fft_size, number_of_dimensions, d_spectrogram, d_coded); });
}
int world_hip_decode_spectral_envelope(WorldHipContext *c, int rows, int fs, int fft_size, int number_of_dimensions,
                                       const double *d_coded, double *d_spectrogram) {
  return guarded(c, [&] { run_codec(c, kDecodeSp, rows, fs, fft_size, number_of_dimensions, d_coded, d_spectrogram); });
}
int world_hip_code_aperiodicity(WorldHipContext *c, int rows, int fs, int fft_size, const double *d_aperiodicity,
                                double *d_coded) {
  return guarded(c, [&] { run_codec(c, kCodeAp, rows, fs, fft_size, 0, d_aperiodicity, d_coded); });
}
int world_hip_decode_aperiodicity(WorldHipContext *c, int rows, int fs, int fft_size, const double *d_coded,
                                  double *d_aperiodicity) {
  return guarded(c, [&] { run_codec(c, kDecodeAp, rows, fs, fft_size, 0, d_coded, d_aperiodicity); });
}

}  // extern "C"

#include "dropin.inc"
#include "fileio.inc"
