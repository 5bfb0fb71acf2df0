/*
 * world_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the analysis path of mmorise/World
 * (Dio / Harvest / StoneMask / CheapTrick / D4C).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the shipped HIP path (world_amd/csrc) never links or calls it.
 *
 * Parity pinning: the reference ships no golden vectors, so this
 * restatement is pinned against the reference itself: oracle/_ref/libworld_ref.so
 * (unmodified reference sources compiled in place by oracle/Makefile) and the
 * fixtures under tests/golden/ generated from it (tests/golden/make_golden.py).
 *
 * All 2-D outputs are dense row-major [frames][fft_size/2+1].
 */
#ifndef WORLD_ORACLE_H_
#define WORLD_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- primitives (exposed so unit tests can pin them one by one) ---- */
void   wo_randn_seed(uint32_t s[4]);                 /* matlabfunctions.cpp:237-242 */
double wo_randn(uint32_t s[4]);                      /* matlabfunctions.cpp:244-264 */
int    wo_round(double v);                           /* matlabfunctions.cpp:206-208 */
int    wo_pow2_above(int n);                         /* common.cpp:51-54  */
void   wo_rfft(int n, const double *in, double *re, double *im);      /* fft.cpp:49-60  */
void   wo_irfft_unscaled(int n, const double *re, const double *im,
                         double *out);                                 /* fft.cpp:26-35  */
void   wo_nuttall(int len, double *w);               /* common.cpp:113-121 */
void   wo_interp1(const double *x, const double *y, int n,
                  const double *xi, int ni, double *yi);               /* matlabfunctions.cpp:136-176 */
void   wo_interp1q(double x0, double dx, const double *y, int n,
                   const double *xi, int ni, double *yi);              /* matlabfunctions.cpp:214-235 */
void   wo_decimate(const double *x, int n, int r, double *y);         /* matlabfunctions.cpp:27-204 */
void   wo_dc_correction(const double *in, double f0, int fs, int fft_size,
                        double *out);                                  /* common.cpp:56-75 */
void   wo_linear_smoothing(const double *in, double width, int fs,
                           int fft_size, double *out);                 /* common.cpp:27-111 */

/* ---- the analysis path ---- */
int  wo_frame_count(int fs, int x_length, double frame_period);       /* harvest.cpp:1219, dio.cpp:639 */
void wo_harvest(const double *x, int x_length, int fs, double f0_floor,
                double f0_ceil, double frame_period, double *tpos, double *f0);
void wo_dio(const double *x, int x_length, int fs, double f0_floor,
            double f0_ceil, double channels_in_octave, double frame_period,
            int speed, double allowed_range, double *tpos, double *f0);
void wo_stonemask(const double *x, int x_length, int fs, const double *tpos,
                  const double *f0, int nf, double *refined);
int  wo_cheaptrick_fft_size(int fs, double f0_floor);                 /* cheaptrick.cpp:191-194 */
void wo_cheaptrick(const double *x, int x_length, int fs, const double *tpos,
                   const double *f0, int nf, double q1, int fft_size,
                   double *spectrogram);
void wo_d4c(const double *x, int x_length, int fs, const double *tpos,
            const double *f0, int nf, int fft_size, double threshold,
            double *aperiodicity);


/* codec: contiguous row-major [nf][*] arrays in and out */
int  wo_number_of_aperiodicities(int fs);                              /* codec.cpp:212-215 */
void wo_code_aperiodicity(const double *ap, int nf, int fs, int fft_size,
                          double *coded);                              /* codec.cpp:217-236 */
void wo_decode_aperiodicity(const double *coded, int nf, int fs, int fft_size,
                            double *ap);                               /* codec.cpp:238-266 */
void wo_code_spectral_envelope(const double *sp, int nf, int fs, int fft_size,
                               int ndim, double *coded);               /* codec.cpp:268-297 */
void wo_decode_spectral_envelope(const double *coded, int nf, int fs, int fft_size,
                                 int ndim, double *sp);                /* codec.cpp:299-324 */

/* synthesis: dense [nf][fft_size/2+1] spectrogram / aperiodicity */
void wo_synthesis(const double *f0, int nf, const double *sp, const double *ap, int fft_size,
                  double frame_period, int fs, int y_length, double *y);   /* synthesis.cpp:339-399 */

#ifdef __cplusplus
}
#endif
#endif  /* WORLD_ORACLE_H_ */
