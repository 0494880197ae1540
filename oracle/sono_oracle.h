/*
 * sono_oracle.h -- CPU restatement of SonoRadar's spectrogram hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker for the HIP implementation: it may
 * be called from tests/, from __graft_entry__.smoke() and from bench.py's cpu_baseline leg,
 * and from nowhere else.  The shipped library (libsonospec.so) neither links nor loads it.
 *
 * Every function restates one piece of the reference's JavaScript in plain C with the same
 * arithmetic contract: IEEE double arithmetic, no fused multiply-add (build with
 * -ffp-contract=off), a float32 rounding wherever the JavaScript stores into a Float32Array,
 * and ECMAScript ToUint8 / ToUint8Clamp semantics wherever it stores into a Uint8Array /
 * Uint8ClampedArray.  Citations are file:line in the reference tree.
 *
 * Parity pin: tests/test_oracle_golden.py checks every function against the vectors in
 * tests/golden/, which tests/golden/gen_golden.js produced by executing the reference's own
 * JavaScript under Node v12.22.9.  Transcendentals come from glibc libm here and from V8's
 * fdlibm port there; double cos/sin differ by at most 1 ulp in about 3 % of the worker's
 * twiddle factors (checked), every float32 table and every uint8/RGBA output in the fixtures
 * is bit-identical.
 */
#ifndef SONO_ORACLE_H
#define SONO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* window ids, alphabetical like the reference's switch (modules/spectrogram.esm.js:105-151) */
enum {
  ORC_WIN_BARTLETT = 0,
  ORC_WIN_BARTLETT_HANN = 1,
  ORC_WIN_BLACKMAN = 2,
  ORC_WIN_COSINE = 3,
  ORC_WIN_GAUSS = 4,
  ORC_WIN_HAMMING = 5,
  ORC_WIN_HANN = 6,
  ORC_WIN_LANCZOZ = 7,
  ORC_WIN_RECTANGULAR = 8,
  ORC_WIN_TRIANGULAR = 9,
  ORC_WIN_COUNT = 10
};

/* ---- plugin: modules/spectrogram.esm.js ------------------------------------------- */

/* Window table, :105-151.  alpha <= 0 or NaN selects the reference default (.16 / .25).
 * Returns 0, or -1 for an unknown id (the reference throws "No such window function"). */
int orc_plugin_window(int window_id, int n, double alpha, float *out);

/* Twiddle seeds and bit-reverse table, :152-160.  sin_t[0]/cos_t[0] are NaN as in JS. */
void orc_plugin_tables(int n, float *sin_t, float *cos_t, uint32_t *rev);

/* calculateSpectrum, :166-200.  frame[n] -> mag[n/2] (float32 of 2/N*|X_k|).  peak and peak_band
 * carry the running strict-> maximum across calls, as the FFT object does.
 * Returns 0, -1 if n is not a power of two. */
int orc_plugin_spectrum(const float *frame, int n, const float *window, const float *sin_t,
                        const float *cos_t, const uint32_t *rev, float *mag, double *peak,
                        int *peak_band);

/* Hop/overlap rule of getFrequencies, :502-506: noverlap <= 0 means "auto" from canvas_width. */
int orc_plugin_noverlap(int n, int noverlap, size_t length, int canvas_width);

/* Number of frames getFrequencies emits, :526,537 (strict a + N < L). */
size_t orc_plugin_frame_count(size_t length, int n, int hop);

/* getFrequencies for one channel with scale 'linear', :492-542.
 * out:      frames x n/2 uint8 (bin 0 = DC), may be NULL
 * mag_out:  frames x n/2 float32 magnitudes, may be NULL
 * Returns the number of frames, or (size_t)-1 on bad arguments (hop <= 0, bad n/window). */
size_t orc_plugin_get_frequencies(const float *x, size_t length, int n, int noverlap,
                                  int canvas_width, int window_id, double alpha, double gain_db,
                                  double range_db, uint8_t *out, float *mag_out, double *peak,
                                  int *peak_band);

/* resample, :595-640.  in: in_frames x bins -> out: out_width x bins. */
void orc_plugin_resample(const uint8_t *in, size_t in_frames, int bins, size_t out_width,
                         uint8_t *out);

/* Colour LUT of the constructor, :262-271: 256 x RGBA doubles -> 1024 bytes (ToUint8Clamp of
 * Math.round(255*c)). */
void orc_plugin_lut(const double *colormap_rgba, uint8_t *lut);

/* Crop window of drawSpectrogram for scale 'linear', :353-355,377-380.
 * fmax_hz == 0 means Nyquist (:496).  Writes sy (first source row) and sh (row count). */
void orc_plugin_crop(int bins, double sample_rate, double fmin_hz, double fmax_hz, long *sy,
                     long *sh);

/* Non-linear frequency scales, spectrogram.esm.js:412-478.  scale: 0 linear, 1 mel, 2 logarithmic,
 * 3 bark, 4 erb. */
double orc_hz_to_scale(int scale, double hz);
double orc_scale_to_hz(int scale, double v);
/* createFilterBank, :389-411: row e has the taps (bin[e], 1 - c) and (bin[e] + 1, c). */
void orc_plugin_filterbank(int scale, int n, double sample_rate, int filters, int32_t *bin, double *w0,
                           double *w1);
/* applyFilterBank, :479-488, on one frame of n/2 magnitudes: a float32 accumulator over every bin. */
void orc_plugin_apply_filterbank(const float *mag, int bins, const int32_t *bin, const double *w0,
                                 const double *w1, int filters, float *out);
/* getFrequencies with `scale` (:509-530); scale 0 equals orc_plugin_get_frequencies. */
size_t orc_plugin_get_frequencies_scaled(const float *x, size_t length, int n, int noverlap,
                                         int canvas_width, int window_id, double alpha, double gain_db,
                                         double range_db, int scale, double sample_rate, uint8_t *out);
/* crop window of drawSpectrogram under a scale, :377-380. */
void orc_plugin_crop_scaled(int bins, double sample_rate, double fmin_hz, double fmax_hz, int scale,
                            long *sy, long *sh);

/* LUT + vertical flip of drawSpectrogram, :366-376.  frames: width x bins uint8 ->
 * rgba: bins rows x width columns x 4, row 0 = highest bin. */
void orc_plugin_draw(const uint8_t *frames, size_t width, int bins, const uint8_t *lut,
                     uint8_t *rgba);

/* Placement of the cropped rows on the canvas, :377-381 (channel 0), and whether :357-361 fills first. */
void orc_plugin_draw_rect(double sample_rate, double fmin_hz, double fmax_hz, int scale, int height, double *dy,
                          double *dh, int *fill_background);
/* The fill colour of :358-360 as the canvas stores it. */
void orc_plugin_background(const double *colormap_last_rgba, uint8_t *rgba);
/* The drawImage stretch of :380-381 with the engine's own filter (bilinear along y, pixel centres, clamped
 * edges, float32, round half up).  PARITY UNPINNED: the browser's filter is implementation-defined and the
 * reference holds no fixture for it; this restates the engine's definition.  image: rows x width RGBA;
 * canvas: height x width RGBA; background NULL = transparent black. */
void orc_canvas_draw(const uint8_t *image, size_t width, int rows, long sy, long sh, double dy, double dh,
                     int height, const uint8_t *background, uint8_t *canvas);

/* ---- worker: spectrogramWorker.js ---------------------------------------------------- */

/* hannWindow, :195-203. */
void orc_worker_hann(int n, float *out);

/* Twiddle factors, :110-125, flattened: the stage with half size n1 occupies [n1-1, 2*n1-1). */
void orc_worker_twiddles(int n, double *cos_t, double *sin_t);

/* fftOptimized, :206-253, in place on re/im (float32 storage, double arithmetic). */
void orc_worker_fft(float *re, float *im, int n, const double *cos_t, const double *sin_t);

/* hop and image geometry of renderSpectrogram, :39-41,59. */
int orc_worker_hop(int n, double overlap_pct);
size_t orc_worker_width(size_t length, int n, int hop);
size_t orc_worker_frames(size_t length, int n, int hop);

/* Worker colour LUT, setColorMapUint :128-143 (NaN/0 channel -> 0, missing alpha -> 1).
 * colormap_rgba uses NaN for "absent". */
void orc_worker_lut(const double *colormap_rgba, uint8_t *lut);

/* renderSpectrogram, :38-99.  rgba: (n/2) rows x width columns x 4, zero-initialised by the
 * callee like createImageData.  lut == NULL selects the grey fallback (:88-93).  Frames with
 * x >= width land where the reference's flat index puts them (column 0, one row down; the
 * write past the end of the image is dropped).  cos_t/sin_t may be NULL (computed here).
 * idx_out, if not NULL, receives the uint8 colour index per frame x bin for all frames. */
void orc_worker_render(const float *x, size_t length, int n, double overlap_pct,
                       const uint8_t *lut, const double *cos_t, const double *sin_t,
                       uint8_t *rgba, uint8_t *idx_out);

/* Same arithmetic, spread over `threads` POSIX threads by column ranges (cpu_baseline leg of
 * the benchmark).  Produces exactly orc_worker_render's image. */
void orc_worker_render_mt(const float *x, size_t length, int n, double overlap_pct,
                          const uint8_t *lut, uint8_t *rgba, int threads);

/* Worker magnitudes: float32(sqrt(re^2+im^2)) of every frame x bin (spectrogramWorker.js:71-72,
 * rounded to float32 for the feature definition below), and the colour index beside it. */
void orc_worker_magnitudes(const float *x, size_t length, int n, int hop, size_t frames, float *mag_out,
                           uint8_t *idx_out);

/* ---- per-frame features (build-defined; strict-> first-wins rule of :196) ------------ */

/* For every frame of a frames x bins float32 magnitude matrix: the first bin in [bin_lo, bin_hi)
 * whose magnitude is strictly greater than all earlier ones (no reference counterpart: the
 * reference computes no per-frame feature, so this restates the engine's own definition). */
void orc_frame_peaks_f32(const float *mag, size_t frames, int bins, int bin_lo, int bin_hi, int32_t *peak_bin);

/* For every frame of a frames x bins uint8 index matrix: the first bin in [bin_lo, bin_hi)
 * holding the maximum index, and that maximum. */
void orc_frame_peaks(const uint8_t *idx, size_t frames, int bins, int bin_lo, int bin_hi,
                     int32_t *peak_bin, uint8_t *peak_idx);

#ifdef __cplusplus
}
#endif
#endif
