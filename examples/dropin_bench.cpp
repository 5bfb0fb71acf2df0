// dropin_bench.cpp -- the drop-in boundary timed the way the reference's own callers use it (test/test.cpp:111-219):
// a C++ program, host pointers, `new double[]` rows, Harvest() -> CheapTrick() -> D4C() of one utterance at a time.
// It links against libworld_hip.so alone and knows nothing of HIP.  bench.py runs it as a subprocess (`host_to_host`);
// the Python binding adds milliseconds of its own per call and is reported beside it.
//
//   dropin_bench <samples.f64> <fs> [repetitions] [threads]
//
// prints one JSON object: milliseconds per utterance with (a) separately allocated rows, as test.cpp allocates them,
// (b) rows cut out of one dense allocation, (c) `threads` host threads each analysing its own copy of the utterance
// (the library is re-entrant: one slot per caller), plus the first call (context, tables, workspace: cold) apart.
//
//   g++ -O2 -std=c++17 -pthread -I include examples/dropin_bench.cpp -L world_amd -lworld_hip -Wl,-rpath,$PWD/world_amd -o examples/dropin_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "world/cheaptrick.h"
#include "world/d4c.h"
#include "world/harvest.h"

namespace {
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Job {
  std::vector<double> x;
  int fs = 0, nf = 0, fft = 0, nb = 0;
  HarvestOption ho;
  CheapTrickOption co;
  D4COption dop;
  std::vector<double> tp, f0;
  std::vector<double *> sp, ap;          // row pointers
  std::vector<double> dense;             // backing store when the rows are one allocation
  bool separate = true;

  void setup(const std::vector<double> &samples, int rate, bool separate_rows) {
    x = samples; fs = rate; separate = separate_rows;
    InitializeHarvestOption(&ho);
    InitializeCheapTrickOption(fs, &co);
    InitializeD4COption(&dop);
    nf = GetSamplesForHarvest(fs, (int)x.size(), ho.frame_period);
    fft = co.fft_size; nb = fft / 2 + 1;
    tp.assign(nf, 0.0); f0.assign(nf, 0.0);
    sp.resize(nf); ap.resize(nf);
    if (separate) {
      for (int i = 0; i < nf; ++i) { sp[i] = new double[nb]; ap[i] = new double[nb]; memset(sp[i], 0, sizeof(double) * nb); memset(ap[i], 0, sizeof(double) * nb); }
    } else {
      dense.assign((size_t)2 * nf * nb, 0.0);
      for (int i = 0; i < nf; ++i) { sp[i] = dense.data() + (size_t)i * nb; ap[i] = dense.data() + (size_t)(nf + i) * nb; }
    }
  }
  void run(double *stage_ms = nullptr) {
    auto t0 = Clock::now();
    Harvest(x.data(), (int)x.size(), fs, &ho, tp.data(), f0.data());
    auto t1 = Clock::now();
    CheapTrick(x.data(), (int)x.size(), fs, tp.data(), f0.data(), nf, &co, sp.data());
    auto t2 = Clock::now();
    D4C(x.data(), (int)x.size(), fs, tp.data(), f0.data(), nf, fft, &dop, ap.data());
    if (stage_ms) {
      stage_ms[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
      stage_ms[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
      stage_ms[2] += ms_since(t2);
    }
  }
  double checksum() const {
    double s = 0.0;
    for (int i = 0; i < nf; ++i) s += f0[i] + sp[i][i % nb] + ap[i][(7 * i) % nb];
    return s;
  }
  ~Job() { if (separate) for (int i = 0; i < nf; ++i) { delete[] sp[i]; delete[] ap[i]; } }
};
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <samples.f64> <fs> [repetitions] [threads]\n", argv[0]); return 2; }
  const int fs = atoi(argv[2]), reps = argc > 3 ? atoi(argv[3]) : 10, nthreads = argc > 4 ? atoi(argv[4]) : 4;
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  fseek(fp, 0, SEEK_END);
  const long bytes = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  std::vector<double> x(bytes / sizeof(double));
  if (fread(x.data(), sizeof(double), x.size(), fp) != x.size()) { fprintf(stderr, "short read\n"); return 2; }
  fclose(fp);

  Job sep, den;
  sep.setup(x, fs, true);
  den.setup(x, fs, false);
  auto t0 = Clock::now();
  sep.run();                                       // cold: HIP start-up, code objects, context, tables, workspace
  const double first_ms = ms_since(t0);
  sep.run(); den.run();
  double st_sep[3] = {0, 0, 0}, st_den[3] = {0, 0, 0};
  t0 = Clock::now();
  for (int r = 0; r < reps; ++r) sep.run(st_sep);
  const double sep_ms = ms_since(t0) / reps;
  t0 = Clock::now();
  for (int r = 0; r < reps; ++r) den.run(st_den);
  const double den_ms = ms_since(t0) / reps;
  // rows the caller has never touched (fresh pages: the first write to each page faults, in the library's copy loop
  // here and in the reference's own loops there)
  double fresh_ms = 0.0;
  for (int r = 0; r < 3; ++r) {
    Job fresh;
    fresh.setup(x, fs, true);
    for (int i = 0; i < fresh.nf; ++i) { delete[] fresh.sp[i]; delete[] fresh.ap[i]; fresh.sp[i] = new double[fresh.nb]; fresh.ap[i] = new double[fresh.nb]; }
    t0 = Clock::now();
    fresh.run();
    fresh_ms += ms_since(t0) / 3;
  }
  // several host threads, each with its own copy of the signal and its own buffers
  std::vector<Job> jobs(nthreads);
  for (Job &j : jobs) j.setup(x, fs, true);
  auto sweep = [&](int r) {
    std::vector<std::thread> th;
    for (Job &j : jobs) th.emplace_back([&j, r] { for (int k = 0; k < r; ++k) j.run(); });
    for (std::thread &t : th) t.join();
  };
  sweep(2);
  t0 = Clock::now();
  sweep(reps);
  const double thr_ms = ms_since(t0) / (reps * (double)nthreads);
  bool same = sep.checksum() == den.checksum();
  for (Job &j : jobs) same = same && j.checksum() == sep.checksum();
  printf("{\"frames\": %d, \"fft_size\": %d, \"repetitions\": %d, \"first_call_ms\": %.3f, \"separate_rows_ms\": %.4f, "
         "\"separate_rows_stages_ms\": [%.4f, %.4f, %.4f], \"dense_rows_ms\": %.4f, \"dense_rows_stages_ms\": [%.4f, %.4f, %.4f], "
         "\"fresh_rows_ms\": %.4f, \"threads\": %d, \"threads_ms_per_utterance\": %.4f, \"all_results_identical\": %s}\n",
         sep.nf, sep.fft, reps, first_ms, sep_ms, st_sep[0] / reps, st_sep[1] / reps, st_sep[2] / reps, den_ms, st_den[0] / reps,
         st_den[1] / reps, st_den[2] / reps, fresh_ms, nthreads, thr_ms, same ? "true" : "false");
  return same ? 0 : 1;
}
