/*
 * world_hip.h -- C ABI of libworld_hip.so, the MI355X (gfx950) implementation of
 * the WORLD analysis path.
 *
 * Part 1 is the drop-in boundary: the 13 `extern "C"` analysis entry points and 4
 * option structs of mmorise/World with identical names, layouts and argument
 * meaning (host pointers, caller-owned buffers, `double **` row pointers for the
 * spectrogram / aperiodicity).  Each declaration cites the reference header it
 * replaces.  A program written against libworld.a links against libworld_hip.so
 * unchanged for these symbols.
 *
 * Part 2 is the batched, device-resident API the drop-in calls are built on:
 * many utterances per call, inputs and outputs in HBM (plain device pointers, no
 * framework types), one HIP stream, no host synchronisation inside a call.
 */
#ifndef WORLD_HIP_H_
#define WORLD_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define WORLD_HIP_API __attribute__((visibility("default")))
#else
#define WORLD_HIP_API
#endif

/* ------------------------------------------------------------------------- */
/* Part 1: drop-in replacements for the reference's analysis API              */
/* ------------------------------------------------------------------------- */

/* reference src/world/dio.h:16-23 */
typedef struct {
  double f0_floor;
  double f0_ceil;
  double channels_in_octave;
  double frame_period; /* msec */
  int speed;           /* 1, 2, ..., 12 */
  double allowed_range;
} DioOption;

/* reference src/world/harvest.h:16-20 */
typedef struct {
  double f0_floor;
  double f0_ceil;
  double frame_period;
} HarvestOption;

/* reference src/world/cheaptrick.h:16-20 */
typedef struct {
  double q1;
  double f0_floor;
  int fft_size;
} CheapTrickOption;

/* reference src/world/d4c.h:16-18 */
typedef struct {
  double threshold;
} D4COption;

/* reference src/world/dio.h:38,48,61 (src/dio.cpp:639-666) */
WORLD_HIP_API void Dio(const double *x, int x_length, int fs, const DioOption *option,
                       double *temporal_positions, double *f0);
WORLD_HIP_API void InitializeDioOption(DioOption *option);
WORLD_HIP_API int GetSamplesForDIO(int fs, int x_length, double frame_period);

/* reference src/world/harvest.h:35,45,59 (src/harvest.cpp:1219-1262) */
WORLD_HIP_API void Harvest(const double *x, int x_length, int fs, const HarvestOption *option,
                           double *temporal_positions, double *f0);
WORLD_HIP_API void InitializeHarvestOption(HarvestOption *option);
WORLD_HIP_API int GetSamplesForHarvest(int fs, int x_length, double frame_period);

/* reference src/world/stonemask.h:27 (src/stonemask.cpp:212-218) */
WORLD_HIP_API void StoneMask(const double *x, int x_length, int fs, const double *temporal_positions,
                             const double *f0, int f0_length, double *refined_f0);

/* reference src/world/cheaptrick.h:38,52,65,80 (src/cheaptrick.cpp:191-240) */
WORLD_HIP_API void CheapTrick(const double *x, int x_length, int fs, const double *temporal_positions,
                              const double *f0, int f0_length, const CheapTrickOption *option,
                              double **spectrogram);
WORLD_HIP_API void InitializeCheapTrickOption(int fs, CheapTrickOption *option);
WORLD_HIP_API int GetFFTSizeForCheapTrick(int fs, const CheapTrickOption *option);
WORLD_HIP_API double GetF0FloorForCheapTrick(int fs, int fft_size);

/* reference src/world/d4c.h:35,46 (src/d4c.cpp:342-407) */
WORLD_HIP_API void D4C(const double *x, int x_length, int fs, const double *temporal_positions,
                       const double *f0, int f0_length, int fft_size, const D4COption *option,
                       double **aperiodicity);
WORLD_HIP_API void InitializeD4COption(D4COption *option);

/* reference src/world/codec.h:23,38,53,69,86 (src/codec.cpp:212-324) -- SURVEY.md 8f.1.
 * All five public symbols of codec.o are defined so that the object is never pulled from
 * a reference archive linked behind this library. */
WORLD_HIP_API int GetNumberOfAperiodicities(int fs);
WORLD_HIP_API void CodeAperiodicity(const double *const *aperiodicity, int f0_length, int fs, int fft_size,
                                    double **coded_aperiodicity);
WORLD_HIP_API void DecodeAperiodicity(const double *const *coded_aperiodicity, int f0_length, int fs,
                                      int fft_size, double **aperiodicity);
WORLD_HIP_API void CodeSpectralEnvelope(const double *const *spectrogram, int f0_length, int fs, int fft_size,
                                        int number_of_dimensions, double **coded_spectral_envelope);
WORLD_HIP_API void DecodeSpectralEnvelope(const double *const *coded_spectral_envelope, int f0_length, int fs,
                                          int fft_size, int number_of_dimensions, double **spectrogram);

/* reference src/world/synthesis.h:30 (src/synthesis.cpp:339-399) -- SURVEY.md 8f.3.  The only public
 * symbol of synthesis.o; SynthesisRealtime (synthesisrealtime.o) is not provided. */
WORLD_HIP_API void Synthesis(const double *f0, int f0_length, const double *const *spectrogram,
                             const double *const *aperiodicity, int fft_size, double frame_period, int fs,
                             int y_length, double *y);

/* ---- audio and parameter files: the reference's tools/ library ------------
 * Same names, arguments and on-disk bytes as tools/audioio.h:25-47 and tools/parameterio.h:24-114,
 * so examples/parameter_io/{f0,sp,ap}analysis.cpp, readandsynthesis.cpp and test/test.cpp link
 * against this library ALONE.  Headers are parsed and written on the host; the per-sample
 * PCM <-> double conversion of wavread()/wavwrite() runs on the GPU (bit-identical results).
 * Failures print the reference's messages to stdout; there is no other error channel. */
/* tools/audioio.h:25 (tools/audioio.cpp:84-130): 16-bit mono, q = int16(clamp(int(x*32767))); nbit is ignored */
WORLD_HIP_API void wavwrite(const double *x, int x_length, int fs, int nbit, const char *filename);
/* tools/audioio.h:35 (tools/audioio.cpp:132-173): samples in the file; 0 = cannot open, -1 = not accepted */
WORLD_HIP_API int GetAudioLength(const char *filename);
/* tools/audioio.h:47 (tools/audioio.cpp:175-252): mono PCM of 8..32 bits, x = q / 2^(nbit-1) */
WORLD_HIP_API void wavread(const char *filename, int *fs, int *nbit, double *x);
/* tools/parameterio.h:24 (tools/parameterio.cpp:58-88): "F0  " NOF FP + f64 values, or "%.5f %.5f\r\n" text */
WORLD_HIP_API void WriteF0(const char *filename, int f0_length, double frame_period,
                           const double *temporal_positions, const double *f0, int text_flag);
/* tools/parameterio.h:39 (tools/parameterio.cpp:90-117): returns 1 on success; positions = i / 1000 * FP */
WORLD_HIP_API int ReadF0(const char *filename, double *temporal_positions, double *f0);
/* tools/parameterio.h:56 (tools/parameterio.cpp:119-143): "NOF ", "FP  ", "FFT ", "NOD ", "FS  "; 0 if absent */
WORLD_HIP_API double GetHeaderInformation(const char *filename, const char *parameter);
/* tools/parameterio.h:70,85 (tools/parameterio.cpp:145-191): "SPEC" NOF FP FFT NOD FS + rows of f64 */
WORLD_HIP_API void WriteSpectralEnvelope(const char *filename, int fs, int f0_length, double frame_period,
                                         int fft_size, int number_of_dimensions, const double *const *spectrogram);
WORLD_HIP_API int ReadSpectralEnvelope(const char *filename, double **spectrogram);
/* tools/parameterio.h:99,114 (tools/parameterio.cpp:193-243): "AP  ", same layout */
WORLD_HIP_API void WriteAperiodicity(const char *filename, int fs, int f0_length, double frame_period,
                                     int fft_size, int number_of_dimensions, const double *const *aperiodicity);
WORLD_HIP_API int ReadAperiodicity(const char *filename, double **aperiodicity);


/* ---- behaviour of the drop-in symbols that the reference does not have to state -----------------
 * Re-entrancy: like the reference (all state on the stack: src/cheaptrick.cpp:205-206, src/d4c.cpp:345-346) the
 * symbols above may be called from several host threads at once.  Each call runs on one SLOT of a small pool (a
 * library context on its own stream + its device / pinned buffers); WORLD_HIP_DROPIN_SLOTS (default 4) calls run side
 * by side on device WORLD_HIP_DEVICE (default 0), further callers wait for a slot.
 * Resident input: a slot keeps the last signal `x` it uploaded; StoneMask / CheapTrick / D4C (and a repeated Harvest /
 * Dio) on the same pointer, length AND content (a 64-bit hash of every sample) skip the upload.
 * WORLD_HIP_DROPIN_CACHE_X=0 turns that off.  Matrices move through pinned staging in chunks, copied to / from the
 * caller's rows by the calling thread and WORLD_HIP_DROPIN_COPY_THREADS (default 3) helper threads.
 * Errors: the reference API has no error channel and never fails; this library can (no GPU, out of device memory, a
 * shape beyond world_hip_check_shape()).  There is NO CPU fallback.  A failing drop-in call reports through the handler
 * installed here: `function` is the symbol's name, `message` the reason (also world_hip_last_error()).  If the handler
 * returns, the drop-in call returns to its caller; it may also longjmp or throw.  The caller's output buffers are untouched
 * when the call was refused up front (a shape limit, a missing GPU) and UNSPECIFIED after a failure in mid-transfer (the
 * matrices are downloaded chunk by chunk into the caller's rows: a device error between two chunks leaves earlier rows written).
 * With no handler (the default, handler = NULL) the reason is printed to stderr and the process aborts.
 * Hostile values (tests/test_hostile.py): samples may be NaN, +-Inf, 1e308 or denormal -- every call returns, the results
 * are as meaningless as the reference's, temporal_positions never depend on the samples, and the next call is unaffected.
 * Caller-made F0 tracks: the reference turns F0 into window lengths and array indices unchecked (src/cheaptrick.cpp:95,
 * src/d4c.cpp:55-56, src/stonemask.cpp:129, src/common.cpp:60-62) -- NaN and values from about fs/2 up are undefined
 * behaviour there.  Here, and ONLY for such values, the analysis differs from it by rule:
 *   F0 is NaN   CheapTrick: the frame is analysed as unvoiced (the default 500 Hz); D4C: unvoiced (the row is 1 - 1e-12);
 *               StoneMask: the refined value is 0
 *   F0 > fs/2   CheapTrick and D4C analyse the frame as F0 = fs/2 (+Inf included).  The frame then draws fewer randn()
 *               values than the reference would, so LATER frames of that utterance meet other noise samples than the
 *               reference's (differences at the 1e-12 safeguard level); utterances without such a frame are unaffected
 *   F0 <= floor, negative, -Inf: the reference's own floors apply (src/cheaptrick.cpp:218, src/d4c.cpp:263,300), as there. */
typedef void (*WorldHipErrorHandler)(const char *function, const char *message, void *user);
WORLD_HIP_API void world_hip_set_error_handler(WorldHipErrorHandler handler, void *user);
/* Releases what the drop-in layer holds: its helper threads are joined and every slot's context, streams, events, device
 * and pinned buffers are freed.  For host processes that want the GPU path gone without exiting; the next drop-in call
 * starts over (a cold call).  Returns 0, or -1 -- nothing released -- while a drop-in call is running.
 * Environment of the drop-in layer, read once: WORLD_HIP_DROPIN_SLOTS (4), WORLD_HIP_DROPIN_COPY_THREADS (3; 0 = the calling
 * thread copies alone), WORLD_HIP_DROPIN_SPIN_US (0: an idle copy helper sleeps; N > 0: it first polls N microseconds for the next chunk of a
 * running transfer -- worth under 1 %), WORLD_HIP_DROPIN_WIRE (f32: the spectrogram / aperiodicity rows cross
 * PCIe as float -- rounded once on the device, 6e-8 relative -- and are widened into the caller's double rows on the host:
 * half the bytes of the path's PCIe-bound stages; default: double, bit-identical to the device-resident analysis). */
WORLD_HIP_API int world_hip_shutdown(void);
/* diagnostic counters of the drop-in layer: slots created, calls that found their signal resident / had to upload it */
WORLD_HIP_API void world_hip_dropin_stats(unsigned long long *slots, unsigned long long *x_hits,
                                          unsigned long long *x_misses);

/* ------------------------------------------------------------------------- */
/* Part 2: batched device-resident API                                        */
/* ------------------------------------------------------------------------- */
/*
 * Layout.  A batch is n_utt utterances with one sampling rate.  Every array is
 * dense and padded to a per-batch stride:
 *   x            [n_utt][x_stride]            samples            (device)
 *   x_length     [n_utt]                      valid samples      (HOST)
 *   tpos, f0     [n_utt][f_stride]            per-frame scalars  (device)
 *   n_frames     [n_utt]                      valid frames       (HOST)
 *   sp, ap       [n_utt][f_stride][fft/2+1]   dense rows         (device)
 * Frame counts follow GetSamplesForHarvest/GetSamplesForDIO.  Rows/frames beyond
 * an utterance's own count are left untouched.  All functions enqueue work on the
 * context's stream and return without synchronising; 0 = success, non-zero =
 * failure with the reason available from world_hip_last_error().
 */
typedef struct WorldHipContext WorldHipContext;

/* stream = a hipStream_t to enqueue on (NULL = the device's default stream) */
WORLD_HIP_API WorldHipContext *world_hip_create(int device, void *stream);
WORLD_HIP_API void world_hip_destroy(WorldHipContext *ctx);
WORLD_HIP_API const char *world_hip_last_error(void);
/* Version of the batched C ABI below (the reference's own 13 symbols never change).  Bumped whenever a prototype in this
 * header changes incompatibly; a binding built against another major value must refuse to bind (world_amd/api.py does).
 *   5  round 5: world_hip_spectral_packed_range / _cheaptrick_batch_range / _d4c_batch_range take `reuse_offsets`
 *   6  round 6: + world_hip_abi_version itself; no prototype changed
 * Libraries older than 6 lack the symbol. */
#define WORLD_HIP_ABI_VERSION 6
/* Launch-geometry hints of a context (bits; default 0).  Results never depend on them.
 *   WORLD_HIP_HINT_SHARED_DEVICE  other jobs run on this device at the same time (several contexts on their own streams, as in
 *       bench.py's headline mode): single-utterance calls then keep the narrow launch shapes that fit beside other jobs'
 *       frame kernels.  Without it a single-utterance call assumes the device to itself and gives Harvest's one-workgroup-
 *       per-utterance contour kernels (FixStep1-2 + sections, MergeF0) 1024 threads instead of 256: 0.089 -> 0.047 ms of a
 *       lone 10 s job, at the price of sixteen wavefronts waiting for one CU when the device is busy. */
#define WORLD_HIP_HINT_SHARED_DEVICE 1
WORLD_HIP_API int world_hip_set_hint(WorldHipContext *ctx, int hint);
WORLD_HIP_API int world_hip_abi_version(void);
WORLD_HIP_API int world_hip_sync(WorldHipContext *ctx);
/* bytes of device workspace currently held by the context (its arena) */
WORLD_HIP_API unsigned long long world_hip_workspace_bytes(WorldHipContext *ctx);
/* The reference's randn() stream (src/matlabfunctions.cpp:237-264) is a constant of the algorithm: one
 * table per device, shared by every context of the process, checked in full against a sequential host
 * statement of the generator whenever it is (re)built.  bytes() = what it holds (live + superseded
 * generations); verify() reduces the live table again and compares (0 = intact; synchronises). */
WORLD_HIP_API unsigned long long world_hip_noise_table_bytes(WorldHipContext *ctx);
WORLD_HIP_API int world_hip_verify_tables(WorldHipContext *ctx);
/* cold-start accounting: host wall-clock milliseconds this process has spent building + verifying the device's tables
 * (all generations), and how many generations were built.  The host statement is stepped by up to
 * WORLD_HIP_TABLE_THREADS (default min(16, cores)) threads, each range's jump-table seed confirmed sequentially. */
WORLD_HIP_API double world_hip_noise_table_build_ms(WorldHipContext *ctx, int *builds);

/* Per-kernel timing with HIP events on the launch stream (process-wide switch).
 * collect() waits for the recorded kernels and returns "kernel_name ms\n" lines. */
WORLD_HIP_API void world_hip_profile_enable(int on);
WORLD_HIP_API int world_hip_profile_collect(char *buf, int cap);

WORLD_HIP_API int world_hip_harvest_batch(WorldHipContext *ctx, int n_utt, int fs, const double *d_x,
                                          int x_stride, const int *x_length, const HarvestOption *option,
                                          int f_stride, double *d_tpos, double *d_f0);
WORLD_HIP_API int world_hip_dio_batch(WorldHipContext *ctx, int n_utt, int fs, const double *d_x,
                                      int x_stride, const int *x_length, const DioOption *option,
                                      int f_stride, double *d_tpos, double *d_f0);
WORLD_HIP_API int world_hip_stonemask_batch(WorldHipContext *ctx, int n_utt, int fs, const double *d_x,
                                            int x_stride, const int *x_length, const int *n_frames,
                                            int f_stride, const double *d_tpos, const double *d_f0,
                                            double *d_refined_f0);
WORLD_HIP_API int world_hip_cheaptrick_batch(WorldHipContext *ctx, int n_utt, int fs, const double *d_x,
                                             int x_stride, const int *x_length, const int *n_frames,
                                             int f_stride, const double *d_tpos, const double *d_f0,
                                             const CheapTrickOption *option, double *d_spectrogram);
WORLD_HIP_API int world_hip_d4c_batch(WorldHipContext *ctx, int n_utt, int fs, const double *d_x,
                                      int x_stride, const int *x_length, const int *n_frames,
                                      int f_stride, const double *d_tpos, const double *d_f0,
                                      int fft_size, const D4COption *option, double *d_aperiodicity);

/* Waveform synthesis from analysis parameters (reference src/synthesis.cpp:339-399):
 *   f0 [n_utt][f_stride], spectrogram / aperiodicity [n_utt][f_stride][fft_size/2+1] (device),
 *   n_frames, y_length [n_utt] (HOST), y [n_utt][y_stride] (device).  frame_period in ms.
 * The pulse count is data dependent and only known on the device: the workspace holds a mean pulse rate of
 * 1200 Hz over the longest utterance unless world_hip_set_synthesis_pulse_capacity() said otherwise (pulses
 * per utterance; 0 = automatic).  A call that needs more does NOT fail silently: the device records the count,
 * world_hip_sync() then fails with it in world_hip_last_error(), world_hip_synthesis_pulses_dropped() returns it
 * (0 = every pulse was rendered; synchronises; clears the record) -- set the capacity and repeat the call.
 * The drop-in Synthesis() does exactly that by itself. */
WORLD_HIP_API int world_hip_synthesis_batch(WorldHipContext *ctx, int n_utt, int fs, double frame_period,
                                            int fft_size, const int *n_frames, int f_stride, const double *d_f0,
                                            const double *d_spectrogram, const double *d_aperiodicity,
                                            const int *y_length, int y_stride, double *d_y);

/* What box is this?  ~50 ms of microbenchmarks on the context's device (synchronous; allocates and frees 2 GB):
 * values[0] shader clock held under a chip-wide FP64 load (MHz), [1] that load's FMA rate over the whole launch (TFLOP/s; HIP events), [2] / [3] / [4]
 * dependent-load latency of one lane chasing pointers through 2 GB / 64 MB every CU has just read / 1 MB it has just walked (ns per hop), [5] / [6] a dependent LDS read on
 * an idle / a loaded CU (shader cycles), [7] compute units.  n_values >= 8.  bench.py records them in the line's
 * `environment` object: identical binaries ran a lone job 10-90 % slower on some boxes (profiles/r04/README.txt). */
WORLD_HIP_API int world_hip_probe_machine(WorldHipContext *ctx, double *values, int n_values);
/* The per-frame real FFT of the path in isolation (the reference's fft_plan_dft_r2c_1d / _c2r_1d +
 * fft_execute, src/world/fft.h:22-44, as re-implemented in csrc/fft.h): `batch` transforms of 2^lg_n
 * points (256 .. 16384), one workgroup each, straight from and to HBM -- the test and microbenchmark hook.
 *   rfft : d_in [batch][N] -> d_spectrum [batch][N/2+1][2] (re, im), X[k] = sum x[n] e^{-2 pi i k n / N}
 *   irfft: d_spectrum -> d_out [batch][N] = N * irfft (unscaled like the reference's c2r; Im of DC / Nyquist ignored)
 * max_lr = 3 (radix-8 plan) or 4 (radix-16 plan); threads = workgroup size, 0 = one butterfly per thread;
 * static_plan != 0 selects the instantiation whose length is a compile-time constant (what the frame kernels
 * run: radix-8 plan, 1024 / 2048 / 4096 points), 0 the one that takes it at run time. */
WORLD_HIP_API int world_hip_probe_rfft(WorldHipContext *ctx, int lg_n, int max_lr, int threads, int static_plan,
                                       long long batch, const double *d_in, double *d_spectrum);
WORLD_HIP_API int world_hip_probe_irfft(WorldHipContext *ctx, int lg_n, int max_lr, int threads, int static_plan,
                                        long long batch, const double *d_spectrum, double *d_out);

/* Multi-GPU exchange (the reference has no counterpart).  Utterances are sharded over GPUs
 * and analysed independently; a GPU's results are then packed into ONE contiguous block of records
 *     row = [ tpos, f0, spectrogram[0 .. bins), aperiodicity[0 .. bins) ]      (2 + 2 bins doubles)
 * holding the utterances' valid frames back to back, utterance u starting at record first_row + sum of
 * n_frames[0 .. u).  pack / unpack convert between the batched arrays above and such a block (device side,
 * on the context's stream, no synchronisation).
 * world_hip_allgather_blocks is the exchange for ONE process that drives n_dev contexts (one per GPU, one
 * stream each): afterwards d_dst[d] (on context d's device, room for sum(rows) records of `cols` doubles)
 * holds d_src[0], d_src[1], ... back to back.  Every destination pulls its remote blocks with peer copies
 * on its own stream, so all xGMI links of the mesh are busy at once; nothing waits on the host (the
 * destination is valid, and the sources reusable, in the respective context's stream order).
 * With one process PER GPU (torch.distributed / RCCL) the same blocks go through one all-gather:
 * world_amd/distributed.py. */
/* Harvest -> CheapTrick + D4C of one batch in ONE call, into the dense arrays of the *_batch calls (tpos, f0:
 * [n_utt][f_stride]; spectrogram, aperiodicity: [n_utt][f_stride][fft_size/2+1], fft_size = cheaptrick_option->fft_size).
 * Same results, bit for bit, as the three calls in sequence at a third of their host cost (one lock, one set of
 * small-array look-ups). */
WORLD_HIP_API int world_hip_analyze_batch(WorldHipContext *ctx, int n_utt, int fs, const double *d_x, int x_stride,
                                          const int *x_length, const HarvestOption *harvest_option,
                                          const CheapTrickOption *cheaptrick_option, const D4COption *d4c_option,
                                          int f_stride, double *d_tpos, double *d_f0, double *d_spectrogram,
                                          double *d_aperiodicity);
/* Harvest + CheapTrick + D4C of one batch written STRAIGHT into packed records: utterance u's frames occupy rows
 * first_row + sum_{v<u} n_frames[v] ... of d_block ([rows][cols] doubles); n_frames[u] = GetSamplesForHarvest(fs,
 * x_length[u], frame_period).  `cols` names the record format (world_hip_record_columns):
 *   wire 0, cols = 2 + 2 nb : [tpos, f0, sp f64[nb], ap f64[nb]]                       nb = fft_size/2 + 1
 *   wire 1, cols = 2 + nb   : [tpos, f0, sp f32[nb], ap f32[nb]]  -- half the bytes for the multi-GPU exchange and the
 *                             D2H copy; the values are the f64 results rounded once to float (6e-8 relative, the
 *                             contract is 1e-4); tpos and f0 stay f64.
 * The stage kernels store their rows at the records' stride, so no pack pass runs (world_hip_pack_results is for results
 * that already exist in the dense layout).  Same stream semantics as the *_batch calls. */
WORLD_HIP_API int world_hip_record_columns(int fft_size, int wire);
WORLD_HIP_API int world_hip_analyze_packed(WorldHipContext *ctx, int n_utt, int fs, const double *d_x, int x_stride,
                                           const int *x_length, const HarvestOption *harvest_option,
                                           const CheapTrickOption *cheaptrick_option, const D4COption *d4c_option,
                                           long long first_row, double *d_block, int cols);
/* The CODED wire format (the coders exist to "shrink the all-gather and D2H by 10-17x"): the same analysis,
 * its records coded before anything leaves the device --
 *   [tpos, f0, mel-cepstrum[number_of_dimensions], band aperiodicity[GetNumberOfAperiodicities(fs)]]   (doubles)
 * = CodeSpectralEnvelope() / CodeAperiodicity() (reference src/codec.cpp:268-297, :217-236) of exactly the spectrogram and
 * aperiodicity a dense call returns: 67 doubles = 536 bytes per frame at 48 kHz with 60 coefficients against 16 416
 * (31 x fewer bytes on the xGMI links and in the D2H copy).  cols = world_hip_coded_columns(fs, number_of_dimensions).  The
 * full records of the batch live in a staging block the context owns (device memory only) and are read once by the coders.
 * Lossy by design -- what the reference's own coder loses -- and therefore opt-in (world_amd.distributed: wire="coded"). */
WORLD_HIP_API int world_hip_coded_columns(int fs, int number_of_dimensions);
WORLD_HIP_API int world_hip_analyze_coded(WorldHipContext *ctx, int n_utt, int fs, const double *d_x, int x_stride,
                                          const int *x_length, const HarvestOption *harvest_option,
                                          const CheapTrickOption *cheaptrick_option, const D4COption *d4c_option,
                                          int number_of_dimensions, long long first_row, double *d_block, int cols);
/* Frame ranges (frame-level sharding of ONE long utterance -- CheapTrick / D4C only, F0 broadcast; the
 * reference's frames are independent given F0: src/cheaptrick.cpp:207-216, src/d4c.cpp:378-400).  The stages' rows of
 * frames [frame_lo, frame_hi) of every utterance of the batch; the positions in the reference's randn() stream are those of
 * a whole-utterance call (the offset scans, and D4C's LoveTrain pass on which its second scan depends, always cover every
 * frame), so a range's rows are BIT-IDENTICAL to the same rows of the full call.
 *   _spectral_packed_range: CheapTrick + D4C given tpos / f0 ([n_utt][f_stride], device) straight into packed records (same
 *       formats as world_hip_analyze_packed): utterance u's range starts at row first_row + sum over v < u of v's frames in range;
 *   _cheaptrick_batch_range / _d4c_batch_range: one stage into the dense arrays of the *_batch calls (rows outside the range
 *       untouched).
 * reuse_offsets != 0 (all three): an earlier call of the stage on this context had the same shape, buffers and options and
 * only the range differs -- its offsets / LoveTrain results are still in the workspace and are not recomputed (LoveTrain
 * over all frames is a seventh of a whole analysis: a rank that walks its frames in S sub-ranges would repeat it S times).
 * CheapTrick's and D4C's prepared arrays occupy disjoint parts of the workspace, so ranges of the two stages may
 * alternate.  The context remembers what the arrays were prepared FOR (shape, the x / tpos / f0 pointers, lengths, options,
 * the workspace's identity); a call that asks for reuse without matching -- another stage (Harvest, DIO, StoneMask, a coder,
 * Synthesis) ran in between, a different shape, a regrown workspace -- fails with a message instead of reading whatever
 * lies there (the CONTENT of the caller's f0 / x buffers is the caller's promise: the library compares pointers). */
WORLD_HIP_API int world_hip_spectral_packed_range(WorldHipContext *ctx, int n_utt, int fs, const double *d_x, int x_stride,
                                                  const int *x_length, const int *n_frames, int f_stride,
                                                  const double *d_tpos, const double *d_f0,
                                                  const CheapTrickOption *cheaptrick_option, const D4COption *d4c_option,
                                                  int frame_lo, int frame_hi, int reuse_offsets, long long first_row,
                                                  double *d_block, int cols);
WORLD_HIP_API int world_hip_cheaptrick_batch_range(WorldHipContext *ctx, int n_utt, int fs, const double *d_x, int x_stride,
                                                   const int *x_length, const int *n_frames, int f_stride,
                                                   const double *d_tpos, const double *d_f0, const CheapTrickOption *option,
                                                   int frame_lo, int frame_hi, int reuse_offsets, double *d_spectrogram);
WORLD_HIP_API int world_hip_d4c_batch_range(WorldHipContext *ctx, int n_utt, int fs, const double *d_x, int x_stride,
                                            const int *x_length, const int *n_frames, int f_stride, const double *d_tpos,
                                            const double *d_f0, int fft_size, const D4COption *option, int frame_lo,
                                            int frame_hi, int reuse_offsets, double *d_aperiodicity);
WORLD_HIP_API int world_hip_pack_results(WorldHipContext *ctx, int n_utt, const int *n_frames, int f_stride,
                                         int bins, const double *d_tpos, const double *d_f0,
                                         const double *d_spectrogram, const double *d_aperiodicity,
                                         long long first_row, double *d_block);
WORLD_HIP_API int world_hip_unpack_results(WorldHipContext *ctx, int n_utt, const int *n_frames, int f_stride,
                                           int bins, const double *d_block, long long first_row, double *d_tpos,
                                           double *d_f0, double *d_spectrogram, double *d_aperiodicity);
WORLD_HIP_API int world_hip_allgather_blocks(int n_dev, WorldHipContext *const *ctxs, const double *const *d_src,
                                             const long long *rows, int cols, double *const *d_dst);

/* ONE process driving n_dev GPUs (one context each; host side in C/C++, no torch, no RCCL): Harvest + CheapTrick + D4C of
 * a whole job of utterances in HOST memory, sharded longest-first over the devices.  One host thread per device uploads
 * its share in sub-batches of `sub_batch` utterances and analyses each straight into packed records; every other device
 * pulls a finished sub-batch's rows over xGMI (peer copies on its own exchange stream) while the next one is analysed.
 * d_blocks[d]: device d's buffer of rows_capacity x cols doubles, cols = 2 + 2 (fft_size/2 + 1); on return every device
 * holds ALL records at the same rows, where[3 i .. 3 i + 2] = {device index that analysed utterance i, its first row,
 * its frame count}.  Blocking (the inputs are host memory); 0 on success, else world_hip_last_error(). */
WORLD_HIP_API int world_hip_analyze_sharded(int n_dev, WorldHipContext *const *ctxs, int n_utt, int fs,
                                            const double *const *x, const int *x_length,
                                            const HarvestOption *harvest_option, const CheapTrickOption *cheaptrick_option,
                                            const D4COption *d4c_option, int sub_batch, double *const *d_blocks,
                                            long long rows_capacity, int cols, long long *where);

/* Shape limits of the GPU path (the reference has none): 0 = StoneMask, CheapTrick(cheaptrick_fft_size) and D4C all run at
 * this fs; 1 = one of them does not, `why` names the stage and the limit (fs <= 192 kHz for D4C, CheapTrick fft_size <=
 * 8192 -- its default up to fs = 192 kHz --, fs >= 15.8 kHz for D4C, fs <= 240 kHz for StoneMask).  Pure host
 * arithmetic.  The drop-in symbols make the same check before any GPU work and report through the error handler
 * (world_hip_set_error_handler; by default: message + abort -- the reference API has no error channel). */
WORLD_HIP_API int world_hip_check_shape(int fs, int cheaptrick_fft_size, char *why, int why_capacity);

/* HIP graphs: the batched calls enqueued on ctx between _begin and _end are captured into ONE executable graph (bound to
 * the device buffers they were given) instead of being run; _launch replays it on the context's stream at the cost of one
 * host launch (a Harvest + CheapTrick + D4C job is ~45 kernel launches otherwise).  Every call shape must have run once
 * before capture (a captured call may not allocate, copy from the host or wait); an error inside a capture invalidates it.
 * A graph holds raw pointers into memory the CONTEXT owns (workspace arena, small per-call arrays, cached tables).  If a
 * later eager call on the same context has to reallocate any of it (a larger batch, another option set or sampling
 * rate), the graph is stale: world_hip_graph_launch then FAILS (world_hip_last_error: "stale graph") instead of
 * replaying -- capture the job again.  Smaller or equal shapes never invalidate a graph. */
WORLD_HIP_API int world_hip_graph_begin(WorldHipContext *ctx);
WORLD_HIP_API int world_hip_graph_end(WorldHipContext *ctx, void **graph);
WORLD_HIP_API int world_hip_graph_launch(WorldHipContext *ctx, void *graph);
WORLD_HIP_API int world_hip_graph_destroy(void *graph);

WORLD_HIP_API int world_hip_set_synthesis_pulse_capacity(WorldHipContext *ctx, int pulses_per_utterance);
WORLD_HIP_API int world_hip_synthesis_pulses_dropped(WorldHipContext *ctx, int *needed);

/* 16-bit PCM (as stored in a WAV file) -> the doubles the reference's wavread() produces,
 * x = q / 32768 (tools/audioio.cpp:236-249), on the device: upload int16, not FP64. */
WORLD_HIP_API int world_hip_pcm16_to_double(WorldHipContext *ctx, long long n, const short *d_pcm, double *d_x);

/* The general form of the above for a WAV file's data bytes (nbit/8 = 1..4 bytes per sample, little
 * endian, decoded exactly as wavread() does), its inverse for 16-bit output as wavwrite() quantises,
 * and the host-side header parse a batch tool needs to find those bytes:
 *   world_hip_wav_layout() returns 1 and fills fs / nbit / x_length / data_offset (bytes from the
 *   start of the file), 0 if the file cannot be opened, -1 if wavread() would reject it. */
WORLD_HIP_API int world_hip_pcm_to_double(WorldHipContext *ctx, long long n, int nbit, const void *d_pcm, double *d_x);
WORLD_HIP_API int world_hip_double_to_pcm16(WorldHipContext *ctx, long long n, const double *d_x, short *d_pcm);
WORLD_HIP_API int world_hip_wav_layout(const char *filename, int *fs, int *nbit, int *x_length,
                                       long long *data_offset);
/* Host-side writer for samples already quantised (e.g. by world_hip_double_to_pcm16 and one D2H of
 * int16): wavwrite()'s 44-byte header + the samples.  Returns 1 on success, 0 if the file cannot be written. */
WORLD_HIP_API int world_hip_wav_write_pcm16(const char *filename, int fs, long long n, const short *pcm);

/* Coders on dense device rows (reference src/codec.cpp:217-324).  Rows are independent:
 *   spectrogram / aperiodicity  [rows][fft_size/2+1]
 *   coded spectral envelope     [rows][number_of_dimensions]
 *   coded aperiodicity          [rows][GetNumberOfAperiodicities(fs)]
 * A batch's [n_utt][f_stride][...] output of the calls above is one such array with
 * rows = n_utt * f_stride (padding rows must then hold finite positive values). */
WORLD_HIP_API int world_hip_code_spectral_envelope(WorldHipContext *ctx, int rows, int fs, int fft_size,
                                                   int number_of_dimensions, const double *d_spectrogram,
                                                   double *d_coded);
WORLD_HIP_API int world_hip_decode_spectral_envelope(WorldHipContext *ctx, int rows, int fs, int fft_size,
                                                     int number_of_dimensions, const double *d_coded,
                                                     double *d_spectrogram);
WORLD_HIP_API int world_hip_code_aperiodicity(WorldHipContext *ctx, int rows, int fs, int fft_size,
                                              const double *d_aperiodicity, double *d_coded);
WORLD_HIP_API int world_hip_decode_aperiodicity(WorldHipContext *ctx, int rows, int fs, int fft_size,
                                                const double *d_coded, double *d_aperiodicity);

#ifdef __cplusplus
}
#endif
#endif /* WORLD_HIP_H_ */
