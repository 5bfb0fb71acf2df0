// api.hip -- host planner + the C ABI of libworld_hip.so (include/world_hip.h).
//
// The planner mirrors the scalar set-up code at the top of the reference's entry
// points (Harvest()/HarvestGeneralBody src/harvest.cpp:1145-1165,1223-1244,
// CheapTrick() src/cheaptrick.cpp:191-214, D4C() src/d4c.cpp:350-379), carves a
// per-call workspace out of one grow-only HBM arena and enqueues the stage
// kernels on the context's stream.  Nothing here waits for the GPU except the
// drop-in host-pointer entry points, which must hand results back in caller memory.
#include "../../include/world_hip.h"

#include <cmath>
#include <cstdarg>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.h"
#include "bandfilter.h"
#include "decimate.h"
#include "dio.h"
#include "harvest.h"
#include "stage_params.h"

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
  std::vector<double> band_f0;
};

}  // namespace world_hip

struct WorldHipContext {
  int device = 0;
  hipStream_t stream = nullptr;
  world_hip::Tables tab{nullptr, nullptr};
  world_hip::Arena arena;
  world_hip::HarvestBands bands;
  double *d_nuttall = nullptr;   // D4C band window
  int nuttall_len = 0;
  void *dio_bands = nullptr;     // world_hip::DioBands (cached DIO filter tables)
  double *d_noise = nullptr;     // noise[k] = k-th randn() after reseed (grow-only constant table)
  size_t noise_len = 0;
  // pinned, double-buffered staging for the small per-call host arrays
  char *stage[2] = {nullptr, nullptr};
  void *stage_ev[2] = {nullptr, nullptr};
  size_t stage_cap = 0, stage_used = 0;
  int stage_cur = 0;
  std::mutex lock;               // one call at a time per context
};

namespace world_hip {

static void ensure_arena(WorldHipContext *c, size_t bytes) {
  if (bytes <= c->arena.cap) return;
  devrt::sync(c->stream);
  if (c->arena.base) devrt::dfree(c->arena.base);
  size_t cap = bytes + bytes / 8 + (1u << 20);
  c->arena.base = static_cast<char *>(devrt::dmalloc(cap));
  c->arena.cap = cap;
}

static size_t pad256(size_t bytes) { return (bytes + 255) & ~size_t(255); }

// The randn() stream is a constant of the algorithm: make sure its first `draws`
// values are resident (generated once per context by jump-ahead, extended on demand).
static const double *ensure_noise(WorldHipContext *c, size_t draws) {
  if (draws > 0xFFFFFFF0ull) fail("utterance consumes more than 2^32 randn() draws");
  if (draws > c->noise_len) {
    size_t cap = draws + draws / 4;
    if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
    devrt::sync(c->stream);
    double *fresh = static_cast<double *>(devrt::dmalloc(sizeof(double) * cap));
    if (c->d_noise) {
      devrt::d2d(fresh, c->d_noise, sizeof(double) * c->noise_len, c->stream);
      devrt::sync(c->stream);
      devrt::dfree(c->d_noise);
    }
    RngFillArgs fill = {fresh, c->noise_len, cap, c->tab.jump};
    launch_rng_fill(fill, c->stream);
    c->d_noise = fresh;
    c->noise_len = cap;
  }
  return c->d_noise;
}

// Small host arrays travel through pinned staging so the async copy never reads
// memory the caller (or a destroyed std::vector) owns.  Two buffers alternate per
// call; a buffer is reused only after the event recorded behind its copies fired.
struct CallScope {
  WorldHipContext *c;
  explicit CallScope(WorldHipContext *ctx, size_t staging_bytes) : c(ctx) {
    if (staging_bytes > c->stage_cap) {
      devrt::sync(c->stream);
      for (int k = 0; k < 2; ++k) {
        if (c->stage[k]) devrt::hfree_pinned(c->stage[k]);
        c->stage[k] = static_cast<char *>(devrt::hmalloc_pinned(staging_bytes * 2));
        if (!c->stage_ev[k]) c->stage_ev[k] = devrt::event_create();
      }
      c->stage_cap = staging_bytes * 2;
    }
    c->stage_cur ^= 1;
    devrt::event_sync(c->stage_ev[c->stage_cur]);
    c->stage_used = 0;
  }
  ~CallScope() { devrt::event_record(c->stage_ev[c->stage_cur], c->stream); }
};

template <class T> static T *upload(WorldHipContext *c, const std::vector<T> &v) {
  T *d = c->arena.take<T>(v.size() ? v.size() : 1);
  if (!v.empty()) {
    size_t bytes = sizeof(T) * v.size();
    if (c->stage_used + bytes > c->stage_cap) fail("staging overflow");
    char *h = c->stage[c->stage_cur] + c->stage_used;
    memcpy(h, v.data(), bytes);
    c->stage_used += (bytes + 63) & ~size_t(63);
    devrt::h2d(d, h, bytes, c->stream);
  }
  return d;
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

// ---------------------------------------------------------------------------
// CheapTrick
// ---------------------------------------------------------------------------
static void run_cheaptrick(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                           const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                           const double *d_f0, const CheapTrickOption *opt, double *d_sp, bool own_arena) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  const int lg = ilog2_exact(opt->fft_size);
  if (lg < 7 || lg > 12) fail("CheapTrick fft_size %d unsupported (128..4096: one frame must fit LDS)", opt->fft_size);
  int max_frames = 0;
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 0 || n_frames[u] > f_stride) fail("n_frames[%d] outside [0, f_stride]", u);
    max_frames = std::max(max_frames, n_frames[u]);
  }
  size_t need = pad256(sizeof(unsigned) * (size_t)n_utt * f_stride) + 3 * pad256(sizeof(int) * n_utt);
  if (own_arena) { ensure_arena(c, need); c->arena.reset(); }
  CallScope scope(c, 2 * sizeof(int) * n_utt + 256);
  CtParams p;
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, std::vector<int>(x_length, x_length + n_utt));
  p.b.n_frames = upload(c, std::vector<int>(n_frames, n_frames + n_utt));
  p.tpos = d_tpos; p.f0 = d_f0; p.spectrogram = d_sp;
  p.offsets = c->arena.take<unsigned>((size_t)n_utt * f_stride);
  p.noise = ensure_noise(c, (size_t)max_frames * ct_max_draws_per_frame(opt->fft_size));
  p.tab = c->tab;
  p.q1 = opt->q1;
  p.f0_floor = 3.0 * fs / (opt->fft_size - 3.0);                       // cheaptrick.cpp:196-198
  p.lg_fft = lg;
  launch_cheaptrick(p, max_frames, c->stream);
}

// ---------------------------------------------------------------------------
// D4C
// ---------------------------------------------------------------------------
static void run_d4c(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride, const int *x_length,
                    const int *n_frames, int f_stride, const double *d_tpos, const double *d_f0, int fft_size,
                    const D4COption *opt, double *d_ap, bool own_arena) {
  check_batch(n_utt, fs, d_x, x_stride, x_length);
  ilog2_exact(fft_size);
  int max_frames = 0;
  for (int u = 0; u < n_utt; ++u) {
    if (n_frames[u] < 0 || n_frames[u] > f_stride) fail("n_frames[%d] outside [0, f_stride]", u);
    max_frames = std::max(max_frames, n_frames[u]);
  }
  // d4c.cpp:350-363 and :264-265
  const int fft_d4c = static_cast<int>(pow(2.0, 1.0 + static_cast<int>(log(4.0 * fs / kFloorF0D4C + 1) / kLog2)));
  const int fft_love = static_cast<int>(pow(2.0, 1.0 + static_cast<int>(log(3.0 * fs / 40.0 + 1) / kLog2)));
  if (fft_d4c > 4096) fail("D4C: fs=%d needs an internal FFT of %d > 4096 points (LDS budget); fs <= 48 kHz supported", fs, fft_d4c);
  if (fs < 15800) fail("D4C: fs=%d is below the 15.8 kHz the reference's LoveTrain band edges require", fs);
  const int nap = static_cast<int>(std::min(15000.0, fs / 2.0 - 3000.0) / 3000.0);
  const int wl = static_cast<int>(3000.0 * fft_d4c / fs) * 2 + 1;
  if (nap < 1 || nap > 12) fail("D4C: unsupported number of aperiodicity bands %d", nap);
  if (c->nuttall_len != wl) {                                         // NuttallWindow, common.cpp:113-121
    std::vector<double> w(wl);
    for (int i = 0; i < wl; ++i) {
      double t = i / (wl - 1.0);
      w[i] = 0.355768 - 0.487396 * cos(2.0 * kPi * t) + 0.144232 * cos(4.0 * kPi * t) - 0.012604 * cos(6.0 * kPi * t);
    }
    devrt::sync(c->stream);
    if (c->d_nuttall) devrt::dfree(c->d_nuttall);
    c->d_nuttall = static_cast<double *>(devrt::dmalloc(sizeof(double) * wl));
    devrt::h2d(c->d_nuttall, w.data(), sizeof(double) * wl, c->stream);
    devrt::sync(c->stream);
    c->nuttall_len = wl;
  }
  size_t fr = (size_t)n_utt * f_stride;
  const int gd_stride = fft_d4c / 2 + 2;
  size_t need = 2 * pad256(sizeof(unsigned) * fr) + pad256(sizeof(double) * fr) + 4 * pad256(sizeof(int) * n_utt) +
                pad256(sizeof(double) * fr * gd_stride) + pad256(sizeof(double) * fr * 16);
  if (own_arena) { ensure_arena(c, need); c->arena.reset(); }
  CallScope scope(c, 2 * sizeof(int) * n_utt + 256);
  D4cParams p;
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, std::vector<int>(x_length, x_length + n_utt));
  p.b.n_frames = upload(c, std::vector<int>(n_frames, n_frames + n_utt));
  p.tpos = d_tpos; p.f0 = d_f0; p.aperiodicity = d_ap;
  p.ap0 = c->arena.take<double>(fr);
  p.offsets1 = c->arena.take<unsigned>(fr);
  p.offsets2 = c->arena.take<unsigned>(fr);
  p.draws1 = c->arena.take<unsigned>(n_utt);
  p.gd = c->arena.take<double>(fr * gd_stride);
  p.gd_stride = gd_stride;
  p.coarse = c->arena.take<double>(fr * 16);
  p.noise = ensure_noise(c, (size_t)max_frames * d4c_max_draws_per_frame(fs));
  p.nuttall = c->d_nuttall;
  p.tab = c->tab;
  p.threshold = opt->threshold;
  p.fft_out = fft_size;
  p.lg_love = ilog2_exact(fft_love);
  p.lg_d4c = ilog2_exact(fft_d4c);
  p.nap = nap;
  p.wl = wl;
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
  std::vector<int> xl(x_length, x_length + n_utt), yl(n_utt), nfb(n_utt), nfr(n_utt);
  int max_x = 0, max_y = 0, max_fb = 0, max_fr = 0;
  for (int u = 0; u < n_utt; ++u) {
    yl[u] = static_cast<int>(ceil(static_cast<double>(xl[u]) / p.ratio));   // harvest.cpp:1161-1162
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
  p.ext_cap = max_fb + 304 * p.sec_cap + 8;
  p.max_half = hb.max_half;
  p.tab = c->tab;
  p.nseg = hv_segments(max_y);

  const size_t B = n_utt;
  const size_t cand_elems = B * p.fb_stride * p.maxc;
  size_t need = 0;
  need += 4 * pad256(sizeof(int) * B);
  need += pad256(sizeof(double) * B * p.m_stride);
  need += pad256(sizeof(double) * B * p.y_stride);
  need += pad256(sizeof(double) * B * p.nch * 4 * p.ev_cap);
  need += pad256(sizeof(int) * B * p.nch * 4);
  need += pad256(sizeof(double) * B * p.nch * 4 * hv_segment_list_doubles(p.nseg));
  need += pad256(sizeof(int) * B * p.nch * 4 * p.nseg);
  need += pad256(sizeof(double) * B * p.nch * p.fb_stride);
  need += 4 * pad256(sizeof(double) * cand_elems);
  need += pad256(sizeof(int) * B);
  need += 5 * pad256(sizeof(double) * B * p.fb_stride);
  need += pad256(sizeof(int) * B * 6 * p.sec_cap) + pad256(sizeof(int) * B * 2) + pad256(sizeof(double) * B * p.sec_cap);
  need += pad256(sizeof(double) * B * p.ext_cap);
  ensure_arena(c, need);
  c->arena.reset();
  CallScope scope(c, 4 * sizeof(int) * n_utt + 512);

  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, xl);
  p.b.n_frames = upload(c, nfr);
  p.y_len = upload(c, yl);
  p.nfb = upload(c, nfb);
  p.band_f0 = hb.d_band_f0; p.band_half = hb.d_half; p.band_off = hb.d_off; p.band_taps = hb.d_taps;
  p.fwd = c->arena.take<double>(B * p.m_stride);
  p.y = c->arena.take<double>(B * p.y_stride);
  p.events = c->arena.take<double>(B * p.nch * 4 * p.ev_cap);
  p.ev_count = c->arena.take<int>(B * p.nch * 4);
  p.seg_events = c->arena.take<double>(B * p.nch * 4 * hv_segment_list_doubles(p.nseg));
  p.seg_count = c->arena.take<int>(B * p.nch * 4 * p.nseg);
  p.raw = c->arena.take<double>(B * p.nch * p.fb_stride);
  p.cand_a = c->arena.take<double>(cand_elems); p.score_a = c->arena.take<double>(cand_elems);
  p.cand_b = c->arena.take<double>(cand_elems); p.score_b = c->arena.take<double>(cand_elems);
  p.nc = c->arena.take<int>(B);
  p.c0 = c->arena.take<double>(B * p.fb_stride); p.c1 = c->arena.take<double>(B * p.fb_stride);
  p.c2 = c->arena.take<double>(B * p.fb_stride); p.c3 = c->arena.take<double>(B * p.fb_stride);
  p.basic_f0 = c->arena.take<double>(B * p.fb_stride);
  p.sec = c->arena.take<int>(B * 6 * p.sec_cap);
  p.sec_n = c->arena.take<int>(B * 2);
  p.sec_sum = c->arena.take<double>(B * p.sec_cap);
  p.ext = c->arena.take<double>(B * p.ext_cap);
  p.tpos = d_tpos; p.f0 = d_f0;
  launch_harvest(p, max_x, max_y, max_fb, max_fr, c->stream);
}

// ---------------------------------------------------------------------------
// DIO / StoneMask
// ---------------------------------------------------------------------------
struct DioBands {                // cached per (fs, f0_floor, f0_ceil, channels, ratio)
  int fs = 0, ratio = 0;
  double f0_floor = 0, f0_ceil = 0, cpo = 0;
  int nb = 0, cut = 0, max_ntap = 0;
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
  p.vrm = static_cast<int>(0.5 + 1000.0 / opt->frame_period / opt->f0_floor) * 2 + 1;   // dio.cpp:263-264
  std::vector<int> xl(x_length, x_length + n_utt), yl(n_utt), nfr(n_utt);
  int max_x = 0, max_y = 0, max_fr = 0;
  for (int u = 0; u < n_utt; ++u) {
    yl[u] = 1 + xl[u] / p.ratio;                                      // dio.cpp:590
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
  size_t need = 4 * pad256(sizeof(int) * B);
  need += pad256(sizeof(double) * B * p.m_stride) + pad256(sizeof(double) * B * p.y_stride) +
          pad256(sizeof(double) * B * p.z_stride);
  need += pad256(sizeof(double) * B * p.nb * 4 * seg_list) + pad256(sizeof(int) * B * p.nb * 4 * p.nseg);
  need += pad256(sizeof(double) * B * p.nb * 4 * p.ev_cap) + pad256(sizeof(int) * B * p.nb * 4);
  need += 2 * pad256(sizeof(double) * B * p.nb * f_stride) + 2 * pad256(sizeof(double) * B * f_stride);
  ensure_arena(c, need);
  c->arena.reset();
  CallScope scope(c, 3 * sizeof(int) * n_utt + 512);
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, xl);
  p.b.n_frames = upload(c, nfr);
  p.y_len = upload(c, yl);
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
  if (2 * p.win_cap > kTwN * 2) fail("StoneMask: fs=%d needs an FFT beyond %d points", fs, kTwN);
  ensure_arena(c, 2 * pad256(sizeof(int) * n_utt));
  c->arena.reset();
  CallScope scope(c, 2 * sizeof(int) * n_utt + 256);
  p.b.n_utt = n_utt; p.b.fs = fs; p.b.x_stride = x_stride; p.b.f_stride = f_stride; p.b.x = d_x;
  p.b.x_len = upload(c, std::vector<int>(x_length, x_length + n_utt));
  p.b.n_frames = upload(c, std::vector<int>(n_frames, n_frames + n_utt));
  p.tpos = d_tpos; p.f0 = d_f0; p.refined = d_refined;
  p.tab = c->tab;
  launch_stonemask(p, max_frames, c->stream);
}

// ---------------------------------------------------------------------------
// error plumbing for the C ABI
// ---------------------------------------------------------------------------
template <class F> static int guarded(WorldHipContext *c, F f) {
  if (!c) { g_last_error = "null context"; return 2; }
  std::lock_guard<std::mutex> g(c->lock);
  try {
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
    devrt::set_device(device);
    WorldHipContext *c = new WorldHipContext;
    c->device = device;
    c->stream = static_cast<hipStream_t>(stream);
    std::vector<double2> tw(kTwN);
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
    c->dio_bands = new DioBands;
    return c;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return nullptr;
  }
}

void world_hip_destroy(WorldHipContext *c) {
  if (!c) return;
  try {
    devrt::sync(c->stream);
    devrt::dfree(const_cast<double2 *>(c->tab.tw));
    devrt::dfree(const_cast<uint4 *>(c->tab.jump));
    if (c->arena.base) devrt::dfree(c->arena.base);
    if (c->d_nuttall) devrt::dfree(c->d_nuttall);
    if (c->d_noise) devrt::dfree(c->d_noise);
    if (c->dio_bands) {
      DioBands *db = static_cast<DioBands *>(c->dio_bands);
      if (db->d_band_f0) {
        devrt::dfree(db->d_band_f0); devrt::dfree(db->d_taps); devrt::dfree(db->d_lowcut); devrt::dfree(db->d_hal);
        devrt::dfree(db->d_off);
      }
      delete db;
    }
    for (int k = 0; k < 2; ++k) {
      if (c->stage[k]) devrt::hfree_pinned(c->stage[k]);
      if (c->stage_ev[k]) devrt::event_destroy(c->stage_ev[k]);
    }
    HarvestBands &hb = c->bands;
    if (hb.d_band_f0) { devrt::dfree(hb.d_band_f0); devrt::dfree(hb.d_taps); devrt::dfree(hb.d_half); devrt::dfree(hb.d_off); }
  } catch (...) {
  }
  delete c;
}

const char *world_hip_last_error(void) { return g_last_error.c_str(); }

int world_hip_sync(WorldHipContext *c) {
  return guarded(c, [&] { devrt::sync(c->stream); });
}

unsigned long long world_hip_workspace_bytes(WorldHipContext *c) {
  return c ? c->arena.cap + sizeof(double) * c->noise_len : 0;
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
    run_cheaptrick(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, option, d_sp, true);
  });
}

int world_hip_d4c_batch(WorldHipContext *c, int n_utt, int fs, const double *d_x, int x_stride,
                        const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                        const double *d_f0, int fft_size, const D4COption *option, double *d_ap) {
  return guarded(c, [&] {
    run_d4c(c, n_utt, fs, d_x, x_stride, x_length, n_frames, f_stride, d_tpos, d_f0, fft_size, option, d_ap, true);
  });
}

}  // extern "C"

#include "dropin.inc"
