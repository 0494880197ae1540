/*
 * sonospec.h -- C ABI of libsonospec.so, the MI355X (gfx950) spectrogram engine.
 *
 * The library replaces, for batch processing, the two JavaScript spectrogram paths of the
 * SonoRadar viewer.  Each entry point names the reference code it stands in for (file:line in
 * the reference tree); a binding (N-API, ctypes, ...) needs nothing but this header.
 *
 *   worker path   spectrogramWorker.js:38-99,206-253     Hann STFT -> log10(|X|)/5 -> RGBA image
 *   plugin path   modules/spectrogram.esm.js:95-201      windowed FFT with recurrence twiddles
 *                 modules/spectrogram.esm.js:492-542     getFrequencies: dB/gain/range -> uint8
 *                 modules/spectrogram.esm.js:595-640     resample to pixel columns
 *                 modules/spectrogram.esm.js:345-388     colour LUT, vertical flip, band crop
 *
 * Conventions: every function returns SONO_OK (0) or a negative sono_status; the text of the
 * last failure on a context is available from sono_last_error().  The caller owns every buffer
 * it passes.  A context belongs to one thread and one GPU; contexts are independent.  No C++
 * types, exceptions or framework objects cross this boundary.
 */
#ifndef SONOSPEC_H
#define SONOSPEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONOSPEC_ABI_VERSION 1

typedef struct sono_ctx sono_ctx;

typedef enum sono_status {
  SONO_OK = 0,
  SONO_ERR_ARG = -1,       /* bad argument (size, enum, NULL)                         */
  SONO_ERR_WINDOW = -2,    /* "No such window function" (spectrogram.esm.js:149-150)   */
  SONO_ERR_FFT_SIZE = -3,  /* "Invalid buffer size, must be a power of 2." (:168-169)  */
  SONO_ERR_COLORMAP = -4,  /* colour map validation (:213-217)                         */
  SONO_ERR_CAPACITY = -5,  /* output buffer too small                                  */
  SONO_ERR_DEVICE = -6,    /* HIP runtime error, text in sono_last_error               */
  SONO_ERR_NO_DEVICE = -7, /* no usable GPU                                            */
  SONO_ERR_FORMAT = -8,    /* malformed RIFF/WAVE data (where the reference's DataView read throws) */
  SONO_ERR_BUSY = -9       /* sono_stft_static: the GPU's static index plane is still in use by another stream */
} sono_status;

/* Window functions in the order of the plugin's switch, spectrogram.esm.js:105-151. */
typedef enum sono_window {
  SONO_WIN_BARTLETT = 0,
  SONO_WIN_BARTLETT_HANN = 1,
  SONO_WIN_BLACKMAN = 2,
  SONO_WIN_COSINE = 3,
  SONO_WIN_GAUSS = 4,
  SONO_WIN_HAMMING = 5,
  SONO_WIN_HANN = 6,
  SONO_WIN_LANCZOZ = 7,
  SONO_WIN_RECTANGULAR = 8,
  SONO_WIN_TRIANGULAR = 9,
  SONO_WIN_WORKER_HANN = 10 /* the worker's own Hann table, spectrogramWorker.js:195-203 */
} sono_window;

/* Which quantiser turns |X| into a colour index. */
typedef enum sono_scale_mode {
  SONO_SCALE_WORKER_LOG5 = 0, /* floor(255*clamp(log10|X|/5,0,1)), spectrogramWorker.js:71-78  */
  SONO_SCALE_PLUGIN_DB = 1    /* 20*log10(2/N*|X|) with gainDB/rangeDB, spectrogram.esm.js:531-535 */
} sono_scale_mode;

/* Which frames exist for a signal of L samples. */
typedef enum sono_frame_rule {
  SONO_FRAMES_PLUGIN_STRICT = 0, /* a + N <  L : ceil((L-N)/hop) frames, spectrogram.esm.js:526 */
  SONO_FRAMES_WORKER_COMPAT = 1  /* a + N <= L, image ceil((L-N)/hop) wide; a frame past the
                                    image lands in column 0 one row down, spectrogramWorker.js:40,59,79 */
} sono_frame_rule;

/* Where the FFT's twiddle factors come from. */
typedef enum sono_twiddle_mode {
  SONO_TW_ACCURATE = 0,          /* cos/sin(-2*pi*j/n2) per stage, spectrogramWorker.js:110-125 */
  SONO_TW_PLUGIN_RECURRENCE = 1, /* running product seeded with float32 cos/sin(-pi/T),
                                    spectrogram.esm.js:158-160,176-190                          */
  SONO_TW_HOST_TABLE = 2         /* supplied through sono_set_tables                            */
} sono_twiddle_mode;

typedef enum sono_arith {
  SONO_ARITH_FAST_F32 = 0, /* float32 kernels; colour indices within 1 LSB of the reference.  Holds while
                              the plugin quantiser's floor -gain_db-range_db is no lower than -100 dB (the
                              reference's defaults 20 / 80 are); a context created with a lower floor
                              computes colour indices and images in the exact arithmetic instead.
                              Under the plugin's recurrence twiddles (not conjugate-symmetric) two frames that
                              share a transform separate only up to the table's Hermitian defect; the library
                              bounds it per context (exact operator norm for the table and window) and, where a
                              loud frame could move an index near the floor by a whole step, the kernels
                              transform the loud frames once more, one frame per transform, inside the same
                              launch (sono_kernel_route names such a kernel <...,redo>), so the 1 LSB holds
                              for every input.
                              Samples are taken to be finite and within full scale (|x| <= 1), as decoded PCM
                              is.  Float32 resolves about 2e-7 of a transform's largest value: a frame whose
                              samples are far beyond full scale buries its own bins near the display floor,
                              and those of the frame it shares a transform with, under that rounding (measured:
                              the 1 LSB still holds at 10 times full scale; a tone at 30 times full scale
                              over a floor near -90 dB moves a few plugin indices by 2).  Likewise a NaN or an infinite sample also takes the frame that
                              shares a transform with its own (the reference loses only the frames that contain
                              the sample).  SONO_ARITH_EXACT_F64 transforms every frame alone in the
                              reference's own arithmetic and reproduces it for such input too.
                              SONO_OUT_F32_DB: within 0.01 dB of the reference's own arithmetic for every
                              bin above that floor that lies within 80 dB of its frame's largest bin, and
                              within 0.04 dB for the bins 80..95 dB below it (measured worst case 0.026 dB
                              at 512..4096 points under a full-scale tone: float32 rounding is about
                              2e-7 of a transform's largest value, which is 0.3 % of a bin 94 dB down --
                              a tenth of one colour index step).  Callers that need 0.01 dB all the way
                              down take SONO_ARITH_EXACT_F64, which is bit-identical                    */
  SONO_ARITH_EXACT_F64 = 1 /* the reference's radix-2 dataflow in double with a float32 rounding
                              at every store, no FMA: magnitudes bit-identical to the JavaScript */
} sono_arith;

typedef enum sono_out_kind {
  SONO_OUT_U8_INDEX = 0, /* [frame][bin] uint8, bin 0 = DC (getFrequencies layout)               */
  SONO_OUT_RGBA8 = 1,    /* [row][column] RGBA8, row 0 = highest bin, column = frame
                            (ImageData layout of the worker and of drawSpectrogram); the image must
                            start on a 4-byte boundary (SONO_ERR_ARG otherwise)                   */
  SONO_OUT_F32_DB = 2,   /* [frame][bin] float32 20*log10(m), m = |X| (worker) or 2/N*|X| (plugin) */
  SONO_OUT_F32_MAG = 3,  /* [frame][bin] float32 m                                               */
  SONO_OUT_FEATURES = 4  /* [frame] sono_frame_feature: spectral peak of each frame over the kept band */
} sono_out_kind;

/* Per-frame spectral feature (SONO_OUT_FEATURES).  peak_bin is the first bin (absolute index, DC = 0)
 * of [bin_lo, bin_hi) whose float32 magnitude is strictly greater than all earlier ones -- the
 * strict-> running-maximum rule of the plugin's FFT, spectrogram.esm.js:196, applied per frame;
 * peak_idx is that bin's colour index under the context's quantiser.  The reference derives no
 * spectral feature itself (its auto-ID panel reads user-placed markers, autoIdPanel.js:1388-1468);
 * these records are what a batch host turns into the kHz / ms fields autoIdHK expects. */
typedef struct sono_frame_feature {
  int32_t peak_bin;
  uint8_t peak_idx;
  uint8_t reserved[3];
} sono_frame_feature;

/* One detected call: a maximal run of frames whose peak_idx >= threshold (sono_segment_calls). */
typedef struct sono_call {
  uint64_t start_frame; /* first frame of the run                                  */
  uint64_t end_frame;   /* last frame of the run (inclusive)                       */
  uint64_t peak_frame;  /* frame with the largest peak_idx (first one wins)        */
  int32_t start_bin;    /* peak_bin at start_frame                                 */
  int32_t end_bin;      /* peak_bin at end_frame                                   */
  int32_t high_bin;     /* largest peak_bin in the run                             */
  int32_t low_bin;      /* smallest peak_bin in the run                            */
  int32_t peak_bin;     /* peak_bin at peak_frame                                  */
  uint32_t peak_idx;    /* peak_idx at peak_frame                                  */
} sono_call;

typedef enum sono_in_kind {
  SONO_IN_F32 = 0, /* float32 samples in [-1,1]                       */
  SONO_IN_I16 = 1  /* int16 PCM; sample value = s / 32768 (exact)      */
} sono_in_kind;

typedef enum sono_mem { SONO_MEM_HOST = 0, SONO_MEM_DEVICE = 1 } sono_mem;

typedef struct sono_config {
  uint32_t fft_n;        /* fftSamples / fftSize: power of two, 64 ... 8192                     */
  uint32_t hop;          /* fft_n - noverlap (plugin) or the worker's hop; >= 1                 */
  uint32_t window_id;    /* sono_window                                                         */
  uint32_t scale_mode;   /* sono_scale_mode                                                     */
  uint32_t frame_rule;   /* sono_frame_rule                                                     */
  uint32_t twiddle_mode; /* sono_twiddle_mode                                                   */
  uint32_t arith;        /* sono_arith                                                          */
  uint32_t out_kind;     /* sono_out_kind                                                       */
  uint32_t in_kind;      /* sono_in_kind                                                        */
  uint32_t bin_lo;       /* first bin kept (band crop); 0 = DC                                  */
  uint32_t bin_hi;       /* one past the last bin kept; 0 means fft_n/2                         */
  uint32_t reserved0;    /* must be zero                                                        */
  double alpha;          /* blackman / gauss parameter; <= 0 or NaN selects the reference default */
  double gain_db;        /* plugin quantiser gainDB (reference default 20)                      */
  double range_db;       /* plugin quantiser rangeDB (reference default 80); must be > 0        */
  uint32_t reserved[4];  /* must be zero                                                        */
} sono_config;

typedef struct sono_geometry {
  uint64_t frames; /* frames the transform computes                                              */
  uint64_t width;  /* image columns (differs from frames only under SONO_FRAMES_WORKER_COMPAT)   */
  uint32_t bins;   /* bin_hi - bin_lo: uint8 row length / image height                           */
  uint32_t hop;
} sono_geometry;

/* ---- library / device ----------------------------------------------------------------- */

int sono_abi_version(void);
/* Number of GPUs the process can use; 0 when there is none (never an error). */
int sono_device_count(void);
/* NUMA node of the host memory closest to GPU `device` (its PCI function's numa_node in sysfs), or -1 when the
 * platform does not say.  No counterpart in the reference (a browser page has no placement to make): hosts of the
 * per-file batch bind a rank's threads to that node before they allocate page-locked buffers, so that the copies
 * of the batch replacing modules/fileLoader.js:110-157 do not cross sockets. */
int sono_device_numa_node(int device);

/* ---- host-side rules (no GPU needed) ---------------------------------------------------- */

/* Worker hop: max(1, floor(N*(1-pct/100))), spectrogramWorker.js:39. */
uint32_t sono_worker_hop(uint32_t fft_n, double overlap_pct);
/* Plugin overlap: noverlap if > 0, else max(0, round(N - L/canvas_width)), spectrogram.esm.js:502-506. */
uint32_t sono_plugin_noverlap(uint32_t fft_n, uint32_t noverlap, uint64_t n_samples, uint32_t canvas_width);
/* wsManager's percent rule: floor(N * pct / 100), modules/wsManager.js:87-89. */
uint32_t sono_overlap_from_percent(uint32_t fft_n, double overlap_pct);
/* Frame count / image width for a signal length under a frame rule. */
int sono_geometry_for(uint32_t fft_n, uint32_t hop, uint32_t frame_rule, uint32_t bin_lo, uint32_t bin_hi,
                      uint64_t n_samples, sono_geometry* out);
/* Source-row window of drawSpectrogram for scale 'linear' (spectrogram.esm.js:353-355,377-380):
 * rows [sy, sy+sh) of the flipped image, i.e. bins [bins-sy-sh, bins-sy).  fmax_hz == 0 = Nyquist. */
int sono_plugin_crop(uint32_t bins, double sample_rate, double fmin_hz, double fmax_hz, int64_t* sy, int64_t* sh);

/* The plugin's non-linear frequency scales (option `scale`), spectrogram.esm.js:412-478. */
typedef enum sono_freq_scale {
  SONO_FSCALE_LINEAR = 0,
  SONO_FSCALE_MEL = 1,
  SONO_FSCALE_LOGARITHMIC = 2,
  SONO_FSCALE_BARK = 3,
  SONO_FSCALE_ERB = 4
} sono_freq_scale;
double sono_hz_to_scale(uint32_t scale, double hz); /* hzToScale, :453-465 */
double sono_scale_to_hz(uint32_t scale, double v);  /* scaleToHz, :466-478 */
/* createFilterBank, :389-411: filter e has the two taps (bin[e], w0[e] = 1 - c) and (bin[e] + 1, w1[e] = c).
 * The plugin uses filters == fft_n / 2 (:250-253). */
int sono_filterbank_table(uint32_t scale, uint32_t fft_n, double sample_rate, uint32_t filters, int32_t* bin, double* w0,
                          double* w1);
/* drawSpectrogram's source-row window under a scale, :377-380 (scale 0 == sono_plugin_crop). */
int sono_plugin_crop_scaled(uint32_t bins, double sample_rate, double fmin_hz, double fmax_hz, uint32_t scale, int64_t* sy,
                            int64_t* sh);
/* Where drawSpectrogram puts the cropped rows on its canvas (:377-381, one channel): they cover canvas rows
 * [dy, dy + dh) of `height`; fill_background = 1 when the canvas is first filled with the background colour
 * (frequencyMax above Nyquist, :357-361). */
int sono_plugin_draw_rect(double sample_rate, double fmin_hz, double fmax_hz, uint32_t scale, uint32_t height, double* dy,
                          double* dh, int* fill_background);
/* That background colour: the last colour-map entry's four numbers written into a CSS "rgba(r, g, b, a)" string as
 * they are (:358-360), so r/g/b in 0..1 read as channel values 0..255. */
int sono_plugin_background(const double* colormap_last_rgba, uint8_t* rgba);
/* Colour LUTs: 256 RGBA doubles -> 1024 bytes.  Plugin rule spectrogram.esm.js:262-271, worker
 * rule spectrogramWorker.js:128-143 (NaN channel = absent). */
int sono_plugin_lut(const double* colormap_rgba, uint8_t* lut_rgba);
int sono_worker_lut(const double* colormap_rgba, uint8_t* lut_rgba);
/* Call segmentation over per-frame features (host, O(frames)): maximal runs of frames with
 * peak_idx >= threshold that are at least min_frames long.  `first_frame` is added to the frame
 * numbers (for chunks of a longer recording).  Returns the number of calls found; at most
 * `capacity` are written.  A run touching the last frame is reported like any other: a host that
 * splits one recording into chunks merges it with a run starting at frame 0 of the next chunk. */
int64_t sono_segment_calls(const sono_frame_feature* features, uint64_t frames, uint32_t threshold, uint32_t min_frames,
                           uint64_t first_frame, sono_call* out, uint64_t capacity);
/* ---- from calls to species: the auto-ID panel's fields and the matcher (host, no GPU) ----------------------- */

/* The fourteen numeric fields autoIdHK tests, in its order (modules/autoid_HK.js:216-221).  kHz and ms. */
typedef enum sono_autoid_field {
  SONO_F_HIGHEST = 0, SONO_F_LOWEST = 1, SONO_F_KNEE = 2, SONO_F_HEEL = 3, SONO_F_START = 4, SONO_F_END = 5,
  SONO_F_CF_START = 6, SONO_F_CF_END = 7, SONO_F_DURATION = 8, SONO_F_BANDWIDTH = 9, SONO_F_KNEE_LOW_TIME = 10,
  SONO_F_KNEE_LOW_BW = 11, SONO_F_HEEL_LOW_BW = 12, SONO_F_KNEE_HEEL_BW = 13
} sono_autoid_field;
#define SONO_AUTOID_FIELDS 14
#define SONO_AUTOID_ABSENT INT32_MIN

/* The `data` object autoIdHK receives (autoIdPanel.js:1442-1462).  A NaN field is an absent one (null, undefined
 * or NaN in JavaScript: all three fail every test, autoid_HK.js:201,239). */
typedef struct sono_autoid_data {
  double field[SONO_AUTOID_FIELDS];
  int32_t call_type; /* index of data.callType in the host's list of call types (< 64); < 0 = absent            */
  int32_t harmonic;  /* data.harmonic; SONO_AUTOID_ABSENT = absent                                              */
} sono_autoid_data;

/* One test of a rule on one field. */
typedef enum sono_cond_kind {
  SONO_COND_RANGES = 0,   /* [min, max] or [[min, max], ...]: `count` intervals from interval `first` (inRange, :200-213) */
  SONO_COND_RELATION = 1, /* "op otherField" (:232-256): `first` = sono_relation, `count` = the other field          */
  SONO_COND_NEVER = 2     /* a one-string condition the reference's pattern does not parse: it never holds        */
} sono_cond_kind;
typedef enum sono_relation { SONO_REL_EQ = 0, SONO_REL_GT = 1, SONO_REL_LT = 2, SONO_REL_GE = 3, SONO_REL_LE = 4 } sono_relation;
typedef struct sono_autoid_cond {
  uint32_t field, kind, first, count;
} sono_autoid_cond;

/* One entry of a species' `rules` array (autoid_HK.js:3-198 is the reference's table; it stays with the host, which
 * flattens it into these records). */
typedef struct sono_autoid_rule {
  uint64_t call_type_mask;  /* bit i: call type i is listed in rule.callType; 0 = the rule has no callType         */
  uint32_t harmonic_first;  /* rule.harmonic: `harmonic_count` values from harmonics[harmonic_first]; 0 = none     */
  uint32_t harmonic_count;
  uint32_t cond_first;      /* the rule's field tests                                                             */
  uint32_t cond_count;
  uint32_t species;         /* which species the rule belongs to                                                  */
  uint32_t reserved;
} sono_autoid_rule;

/* autoIdHK (modules/autoid_HK.js:215-264): matched[s] = 1 when any rule of species s accepts `data`.  The host joins
 * the names of the matched species with " / " or reports "No species matched" (:261-263). */
int sono_autoid_match(const sono_autoid_rule* rules, uint32_t n_rules, const sono_autoid_cond* conds, const double* intervals,
                      const int32_t* harmonics, const sono_autoid_data* data, uint8_t* matched, uint32_t n_species);

/* Where a call turns, from its per-frame peak track.  The reference has no counterpart: its panel reads the knee,
 * heel and CF markers a user drags (autoIdPanel.js:1397-1404), so these definitions are this library's own (parity
 * unpinned).  Frames are absolute; -1 = the call has no such point.
 *   high / low     first frame holding the highest / lowest peak bin
 *   cf_start / end the longest run of frames whose peak stays within one bin of the peak at the call's loudest frame,
 *                  if it spans at least 3 frames and a quarter of the call
 *   knee           the frame where the sweep flattens most: largest |slope before| - |slope after|, slopes taken
 *                  over max(1, frames / 8) frames; calls of fewer than 5 frames have none
 *   heel           the frame at or after the knee where it steepens most */
typedef struct sono_call_shape_t {
  int64_t high_frame, low_frame, knee_frame, heel_frame, cf_start_frame, cf_end_frame;
  int32_t knee_bin, heel_bin, cf_start_bin, cf_end_bin;
} sono_call_shape_t;
int sono_call_shape(const sono_frame_feature* features, uint64_t frames, uint64_t first_frame, const sono_call* call,
                    sono_call_shape_t* out);

/* The fields of runPulseId (autoIdPanel.js:1388-1462) from a call and its shape (NULL: the eight shape-derived
 * fields stay absent): kHz = bin * sample_rate / fft_n / 1000, ms = frames * hop / sample_rate * 1000; under time
 * expansion frequencies are multiplied and times divided (:1439-1462).  bandwidth is cfStart - endFreq when
 * cf_call_type is set (the panel's rule for "FM-CF-FM" and "CF-FM", :1416-1421), else highest - lowest.
 * call_type and harmonic are left absent for the host to fill. */
int sono_autoid_data_of_call(const sono_call* call, const sono_call_shape_t* shape, uint32_t fft_n, uint32_t hop, double sample_rate,
                             double time_expansion, int cf_call_type, sono_autoid_data* out);

/* Window table exactly as the reference fills it (float32).  SONO_ERR_WINDOW for an unknown id. */
int sono_window_table(uint32_t window_id, uint32_t fft_n, double alpha, float* out);
/* Twiddle table in the flat stage layout: the stage with half size T occupies [T-1, 2T-1). */
int sono_twiddle_table(uint32_t twiddle_mode, uint32_t fft_n, double* cos_out, double* sin_out);

/* ---- context ------------------------------------------------------------------------------ */

/* Replaces `new FFT(...)` + the per-render set-up of both paths (spectrogram.esm.js:507,
 * spectrogramWorker.js:46-57).  Builds window and twiddle tables on the host and uploads them.
 * Returns NULL on failure; sono_create_error() then holds the reason. */
sono_ctx* sono_create(int device, const sono_config* cfg);
const char* sono_create_error(void);
void sono_destroy(sono_ctx* ctx);
const char* sono_last_error(const sono_ctx* ctx);
int sono_get_config(const sono_ctx* ctx, sono_config* out);

/* Launch on a caller-owned HIP stream (hipStream_t passed as void*); NULL restores the
 * context's own stream.  The new stream is made to wait (hipStreamWaitEvent, no host wait) for what the
 * context has already enqueued on the previous one: all launches of a context share its scratch.  A caller
 * that destroys its stream restores the context's own (NULL) or sets another one before the next call. */
int sono_set_stream(sono_ctx* ctx, void* hip_stream);

/* Colour LUT used by SONO_OUT_RGBA8 and sono_colorize (what _colorMapUint / colorMapUint hold,
 * spectrogram.esm.js:262-271, spectrogramWorker.js:128-143).  NULL selects the worker's grey
 * fallback (col,col,col,255), spectrogramWorker.js:88-93. */
int sono_set_colormap(sono_ctx* ctx, const uint8_t* lut_rgba /* 1024 bytes or NULL */);

/* Host-computed tables (a JavaScript host passes the ones its own Math.cos/sin produced so the
 * exact mode sees V8's values).  Either pointer may be NULL to keep the current table. */
int sono_set_tables(sono_ctx* ctx, const float* window /* fft_n */, const double* tw_cos, const double* tw_sin /* fft_n-1 each */);

/* ---- the transform ------------------------------------------------------------------------ */

/* Installs (or, with bin == NULL, removes) the filter bank that getFrequencies applies between the magnitudes
 * and the quantiser for a non-linear `scale` (applyFilterBank, spectrogram.esm.js:479-488,509-530): fft_n/2
 * filters of two taps each, from sono_filterbank_table or from the host's own arithmetic.  Plugin scale mode
 * only.  In the float32 arithmetic the transform kernels gather the two taps themselves (colour indices and images);
 * in the exact arithmetic, and for float32 / feature outputs, the transform runs in the general kernel. */
int sono_set_filterbank(sono_ctx* ctx, const int32_t* bin, const double* w0, const double* w1);
int sono_geometry_of(const sono_ctx* ctx, uint64_t n_samples, sono_geometry* out);
/* Bytes sono_stft writes for n_samples under the context's out_kind. */
uint64_t sono_out_bytes(const sono_ctx* ctx, uint64_t n_samples);

/* One recording (or one hop-aligned chunk of one): PCM -> out_kind.  Replaces
 * renderSpectrogram (spectrogramWorker.js:38-99) and getFrequencies (spectrogram.esm.js:492-542).
 * `mem` tells whether pcm/out are host or device pointers (sono_mem); host buffers are staged
 * through context-owned device memory.  Synchronous: the result is complete on return. */
int sono_stft(sono_ctx* ctx, const void* pcm, uint64_t n_samples, void* out, uint64_t out_capacity_bytes, int mem);

/* Same, device pointers only, enqueued on the context's stream without waiting.  Allocates nothing and never
 * synchronises: every device buffer it needs besides the caller's (tables, the scratch of the RGBA routes) was
 * sized by sono_create / sono_set_tables / sono_set_filterbank; an RGBA route whose colour indices do not fit the
 * context's scratch works through it in chunks of frames. */
int sono_stft_async(sono_ctx* ctx, const void* pcm_dev, uint64_t n_samples, void* out_dev, uint64_t out_capacity_bytes);
int sono_wait(sono_ctx* ctx);

/* Kernels that the most recent transform call on this context enqueued, e.g. "stft1024_u8_kernel<roots>:rgba" or
 * "stft_mw_kernel<4096>:u8 + colorize_kernel" ("" before the first call).  A context whose size, arithmetic,
 * tables or band no specialised kernel serves reads "stft_general_kernel<...>", which is about ten times slower;
 * hosts print this so that the fall-back is visible.  No counterpart in the reference. */
const char* sono_kernel_route(const sono_ctx* ctx);

/* SONO_OUT_FEATURES in the float32 arithmetic: number of records of the most recent launch that the float32 kernel
 * could not decide (two top bins too close, or the peak too close to a step of the quantiser) and that were
 * recomputed in the exact arithmetic, which is why feature records are the reference's in both arithmetics.  Waits
 * for the context's stream.  0 for other contexts.  No counterpart in the reference. */
int64_t sono_features_recomputed(sono_ctx* ctx);

/* Sizes the context's staging buffers for recordings of up to n_samples, so that later sono_stft_host_async calls
 * of that size or less do not allocate (an allocation waits for the whole device).  May allocate and wait itself.
 * No counterpart in the reference. */
int sono_reserve(sono_ctx* ctx, uint64_t n_samples);

/* sono_stft_async for callers that cannot allocate device memory or let a library do so (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc); it has no counterpart in the reference.
 * SONO_OUT_RGBA8 contexts only.  The call's colour indices ([frame][bin] uint8, the band rounded outwards to
 * groups of four bins) go through a device array of SONO_STATIC_SCRATCH_BYTES that is part of the library's code
 * object: every GPU the library runs on holds one copy until the process ends.  A call whose frames x bins exceed
 * that size returns SONO_ERR_CAPACITY (32 MiB = 65 536 frames of 512 bins, i.e. 43 s at 384 kHz, FFT 1024,
 * hop 256; the reference loads files of at most 20 s, fileLoader.js:181).  Calls on the same GPU share the
 * array, whichever context or stream they use: two such calls must not overlap in time.  Calls on one stream follow
 * each other by construction; a call that arrives on another stream while the kernels of the array's last user are
 * still running returns SONO_ERR_BUSY and enqueues nothing (checked with an event, not a wait).  Same result as
 * sono_stft_async, enqueued on the context's stream without waiting, no hipMalloc / hipFree inside. */
#define SONO_STATIC_SCRATCH_BYTES (32u << 20)
int sono_stft_static(sono_ctx* ctx, const void* pcm_dev, uint64_t n_samples, void* out_dev, uint64_t out_capacity_bytes);

/* Host pointers, enqueued without waiting: copy in, transform, copy out on the context's stream, through
 * the context's staging buffers (so: one transform in flight per context; a batch host cycles over a
 * few contexts to overlap the copies of one file with the transform of another).  The copies only run
 * asynchronously from page-locked memory (sono_host_malloc).  The staging buffers grow -- which waits for
 * the device -- only when a call is larger than every earlier one and than sono_reserve's size.  No
 * counterpart in the reference (its worker gets a transferred ArrayBuffer, main.js:1417). */
int sono_stft_host_async(sono_ctx* ctx, const void* pcm_host, uint64_t n_samples, void* out_host, uint64_t out_capacity_bytes);
void* sono_host_malloc(uint64_t bytes);
int sono_host_free(void* p);

/* resample (spectrogram.esm.js:595-640): [in_frames][bins] uint8 -> [out_width][bins] uint8. */
int sono_resample(sono_ctx* ctx, const uint8_t* in, uint64_t in_frames, uint32_t bins, uint64_t out_width,
                  uint8_t* out, int mem);
/* Columns [col0, col0 + ncols) of that result only (a tile of a zoomed view): out is [ncols][bins]. */
int sono_resample_cols(sono_ctx* ctx, const uint8_t* in, uint64_t in_frames, uint32_t bins, uint64_t out_width, uint64_t col0,
                       uint64_t ncols, uint8_t* out, int mem);

/* LUT + vertical flip + row window of drawSpectrogram (spectrogram.esm.js:366-380):
 * [width][bins] uint8 -> rows [row0, row0+rows) of the [bins][width] RGBA image. */
int sono_colorize(sono_ctx* ctx, const uint8_t* frames, uint64_t width, uint32_t bins, uint32_t row0,
                  uint32_t rows, uint8_t* rgba, int mem);

/* The canvas step of drawSpectrogram (spectrogram.esm.js:357-361,377-381): LUT + flip of [width][bins] uint8
 * columns, rows [sy, sy + sh) of that image stretched over canvas rows [dy, dy + dh) of a [height][width] RGBA
 * canvas; the rest is `background_rgba` (4 bytes; NULL = transparent black, a cleared canvas).  The browser's
 * drawImage filter is implementation-defined, so this step has no pinned counterpart: the filter here is
 * bilinear along y with clamped edges, sampled at pixel centres, float32 with round-half-up; with dh == sh and
 * an integer dy it copies the rows. */
int sono_canvas_draw(sono_ctx* ctx, const uint8_t* frames, uint64_t width, uint32_t bins, int64_t sy, int64_t sh, double dy,
                     double dh, uint32_t height, const uint8_t* background_rgba, uint8_t* canvas, int mem);

/* Per-frame peak over the kept band of a [frames][bins] uint8 index matrix: first bin holding
 * the maximum (the strict-> rule of spectrogram.esm.js:196) and that maximum. */
int sono_frame_peaks(sono_ctx* ctx, const uint8_t* idx, uint64_t frames, uint32_t bins, int32_t* peak_bin,
                     uint8_t* peak_idx, int mem);

/* ---- device memory for hosts without an allocator of their own (N-API addon, C++ bench) ---- */

void* sono_device_malloc(sono_ctx* ctx, uint64_t bytes);
int sono_device_free(sono_ctx* ctx, void* p);
int sono_copy_to_device(sono_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int sono_copy_to_host(sono_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);

/* ---- file intake: RIFF/WAVE, GUANO, crop by time (host side) ------------------------------- */

/* What the reference's three chunk walks find in a .wav file (chunks from offset 12, odd sizes padded). */
typedef struct sono_wav_info {
  uint32_t sample_rate;        /* getWavSampleRate, fileLoader.js:7-27: rate of the FIRST fmt chunk, 256000 if none */
  uint32_t has_fmt;            /* a fmt chunk was seen before the data chunk (getWavDuration / cropWavBlob rule)    */
  uint32_t has_data;
  uint32_t fmt_sample_rate;    /* rate of the last fmt chunk before the data chunk                                  */
  uint16_t format_tag;         /* 1 PCM, 3 IEEE float; WAVE_FORMAT_EXTENSIBLE is resolved to its sub-format         */
  uint16_t channels;
  uint16_t bits_per_sample;
  uint16_t reserved0;
  uint32_t block_align;        /* channels * bits_per_sample / 8                                                    */
  uint32_t fmt_bytes;
  uint64_t fmt_offset;         /* payload offset of that fmt chunk                                                  */
  uint64_t data_offset;        /* payload offset of the data chunk                                                  */
  uint64_t data_bytes;         /* its declared size                                                                 */
  uint64_t data_bytes_present; /* ... clipped to the end of the file                                                */
  uint64_t frames;             /* sample frames present                                                             */
  double duration_s;           /* getWavDuration, fileLoader.js:29-63 (0 when fmt or data is missing)                */
} sono_wav_info;
int sono_wav_parse(const void* file, uint64_t bytes, sono_wav_info* out);

/* extractGuanoMetadata, guanoReader.js:3-34: payload of the first chunk (any id) whose bytes contain
 * "GUANO|Version:"; *length == 0 when there is none (the reference then returns "(No GUANO metadata
 * found in file)").  SONO_ERR_FORMAT where the reference's reads throw (truncated chunk). */
int sono_wav_guano(const void* file, uint64_t bytes, uint64_t* offset, uint64_t* length);

/* parseGuanoMetadata, guanoReader.js:36-80.  present == 0 stands for the reference's empty object (empty
 * text or the "(No GUANO" placeholder).  Strings longer than the fields are truncated. */
typedef struct sono_guano_fields_t {
  int32_t present;
  char date[64];      /* Timestamp's date part with '-' -> '/'                                   */
  char time[16];      /* first five characters of the time part without its first ':' ("2015")   */
  char latitude[64];  /* first token of "Loc Position"                                           */
  char longitude[64]; /* second token through parseFloat / Number.toString, -113..-115 made positive */
} sono_guano_fields_t;
int sono_guano_fields(const char* text, uint64_t bytes, sono_guano_fields_t* out);

/* cropWavBlob, cropAudio.js:1-59: the file's bytes up to the data payload followed by the samples of
 * [floor(start*rate), floor(end*rate)), RIFF and data sizes patched; chunks after the data are dropped.
 * Returns the number of bytes written (with out == NULL: needed), 0 where the reference returns null,
 * or a negative sono_status. */
int64_t sono_wav_crop(const void* file, uint64_t bytes, double start_s, double end_s, void* out, uint64_t capacity);

/* The load filters of fileLoader.js:179-181: 0 accept, 1 skipped as small (< 200 KiB), 2 skipped as long
 * (> 20 s unless time-expansion mode). */
int sono_wav_accept(uint64_t file_bytes, double duration_s, int time_expansion);

/* Interleaved PCM -> one channel of float32 on the device, without a host-side float copy.
 * format_tag 1: 8-bit unsigned ((v-128)/128), 16 (v/32768), 24 (v/8388608), 32 (v/2147483648);
 * format_tag 3: float32 as is.  The browser's decodeAudioData is not reproducible offline, so these
 * scalings are this library's convention (16-bit matches SONO_IN_I16).  `src` is host or device memory
 * (`mem`), `dst_dev` device memory for `frames` floats.  Asynchronous on the context's stream for
 * device sources. */
int sono_pcm_to_f32(sono_ctx* ctx, const void* src, int mem, uint32_t format_tag, uint32_t bits_per_sample, uint32_t channels,
                    uint32_t channel, uint64_t frames, float* dst_dev);

/* ---- timing (HIP events on the context's stream) ------------------------------------------ */

/* What this GPU's vector ALU issues at the transform kernels' occupancy (sixteen wavefronts per CU of independent
 * v_fma_f32), in 10^9 wavefront-instructions per second; < 0 on error.  A benchmark host prices the kernels'
 * instruction counts against it beside the HBM roofline.  Takes a few milliseconds; allocates and waits. */
double sono_measure_valu_peak(sono_ctx* ctx);

int sono_timer_start(sono_ctx* ctx);
/* Records the stop event, waits for it and returns the elapsed milliseconds (< 0 on error). */
double sono_timer_stop_ms(sono_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SONOSPEC_H */
