// Internal declarations shared by the translation units of libsonospec.so.
#ifndef SONO_INTERNAL_H
#define SONO_INTERNAL_H

#include <stdint.h>

#include <vector>

#include "sonospec.h"

namespace sono {

bool is_pow2(uint32_t n);
uint32_t ilog2(uint32_t n);
int fill_window(uint32_t id, uint32_t n, double alpha, float* w);
int fill_twiddles(uint32_t mode, uint32_t n, double* c, double* s);
int compute_geometry(uint32_t n, uint32_t hop, uint32_t rule, uint32_t bin_lo, uint32_t bin_hi, uint64_t L,
                     sono_geometry* g);
void pair_defect_gains(uint32_t n, const double* c, const double* s, const float* window, double* g_raw, double* g_win);

// One launch of an STFT kernel: frames [frame0, frame0 + nframes) of one signal.
struct StftParams {
  const void* pcm;        // device: float32 or int16 samples
  uint64_t n_samples;
  uint32_t n, log2n, hop;
  uint64_t frame0;        // first frame of this launch
  uint64_t nframes;       // number of frames in this launch
  uint64_t col0;          // RGBA: image column of frame0
  uint64_t width;         // RGBA: image width in pixels
  uint32_t bin_lo, bin_hi;  // bins written: [bin_lo, bin_hi)
  int32_t row_of_bin0;    // RGBA: image row of bin y is row_of_bin0 - y
  uint32_t out_bins;      // [frame][bin] outputs: row length
  uint32_t out_bin0;      // [frame][bin] outputs: bin stored at column 0
  const float* window;    // device, n floats
  const void* twiddles;   // device, n-1 float2 (fast) or double2 (exact), flat stage layout
  const uint32_t* lut;    // device, 256 packed RGBA (little endian r | g<<8 | b<<16 | a<<24), or nullptr = grey
  void* out;              // device
  const void* fast_table; // device: per-size twiddle matrix of the fast kernels, or nullptr
  uint32_t fast_mode;     // 0: no fast kernel; 1: exact roots of unity (fast_table); 2: the per-stage table `twiddles`
  uint32_t mw_ok;         // the twiddle table starts every stage with (1, 0): the 2048/4096-point kernel may run
  uint32_t pair_exact;    // the table is conjugate-symmetric to float32 accuracy (w[T-j] == -conj(w[j]) in every stage), so
                          // transforming two real frames as one complex frame separates them to float32 accuracy; the
                          // plugin's recurrence factors are not (1e-5 off at 1024 points, 3e-5 at 4096), which is harmless
                          // for colour indices and peaks but not for float32 outputs held to 0.01 dB
  double pair_asym;       // that largest deviation from conjugate symmetry itself (0 for exact roots): a bin of one frame of a
                          // pair carries this fraction of the other frame's value at the same bin
  uint32_t pair_idx_ok;   // colour indices and images may come from a kernel that transforms two frames at once: either the
                          // table's Hermitian defect cannot move an index by a step even for full-scale samples, or the
                          // kernel transforms loud units once more, one frame per transform (loud_thr_raw below)
  const int32_t* fb_bin;  // device: non-linear frequency scale, first tap of output bin i (n/2 entries), or nullptr = linear
  const double* fb_w0;    // device: weight of bin fb_bin[i]
  const double* fb_w1;    // device: weight of bin fb_bin[i] + 1
  const void* fb_packed;  // device, set only for the 1024-point float32 index kernel: int32 tap[n/2], float w0[n/2], float w1[n/2]
  void* tile_scratch;     // device: fast_rgba_scratch_bytes() of scratch for the 2048/4096-point image writers, or nullptr
  uint32_t scale_mode, out_kind, in_kind;
  uint32_t feat_lo, feat_hi;  // SONO_OUT_FEATURES from a float32 kernel: the band the records are for.  The kernels search whole
                          // groups of four bins ([bin_lo, bin_hi) is that band rounded outwards) and mark a record whose
                          // peak fell outside [feat_lo, feat_hi) for the exact-arithmetic pass, which searches the band itself
  uint32_t* feat_queue;   // SONO_OUT_FEATURES from a float32 kernel that lists the records it marks itself (1024 points): the
                          // fix-up queue (a header of 4 words, word 0 = number of marked records, then feat_queue_cap frame
                          // numbers within the launch), or nullptr: the marks are collected from the records afterwards
  uint32_t feat_queue_cap;
  uint32_t fixup;         // != 0: recompute only the feature records a float32 kernel marked (reserved[0] != 0); for
                          // launch_features_fixup the capacity of the queue's list
  double gain_db, range_db;
  // Two frames per transform under a twiddle table that is not conjugate-symmetric (the plugin's recurrence factors): a
  // frame's spectrum picks up a trace of its partner's and of its own mirror image, bounded by pair_defect_gains() times
  // the largest sample.  Where that bound could move a colour index by a whole step, the kernels that pair frames (or
  // split one frame into its even and odd samples) transform a loud unit once more, one real frame per transform, whose
  // spectrum is the reference's own transform of that frame.  0: nothing to redo (symmetric table, or the bound is harmless).
  float loud_thr_raw;     // a lane's largest |sample| above this makes its wavefront's unit a loud one
  uint32_t split_ok;      // 4096 points: a frame may be transformed as its even and odd samples in one complex frame
                          // whatever its level (symmetric table, or a display floor its defect cannot reach)
  void* redo_scratch;     // device, 4096 points under a table whose split is not safe at every level: 16 KiB per wavefront of
                          // the launch (redo_scratch_bytes()), where the kernel parks the even samples' transform of a loud
                          // frame while it transforms the odd samples alone
};

// Every launch_* function reports the kernel it enqueued (a static string), for sono_kernel_route.
void note_kernel(const char* name);

// kernels_general.hip: table-driven radix-2 dataflow, any power-of-two size, both arithmetics.
int launch_stft_general(const StftParams& p, bool exact_f64, void* hip_stream);
// kernels_fast.hip: float32 kernels with accurate twiddles (returns 1 if this size/mode has no fast kernel).
int launch_stft_fast(const StftParams& p, void* hip_stream);
void build_fast_table_1024(std::vector<float>& t);
// kernels_fast_rgba.hip: the 1024-point kernel that writes the RGBA image itself, colour-index exchange in LDS.
int launch_stft_fast_rgba(const StftParams& p, void* hip_stream);
// kernels_fast512.hip: float32 kernel for 512-point frames with exact roots of unity (fast_mode 1, fast_table from
// build_fast_table_512); colour indices and per-frame peaks.  Returns 1 if it does not serve this launch.
int launch_stft_fast512(const StftParams& p, void* hip_stream);
bool fast2048_rgba_available(const StftParams& p);
bool fast512_rgba_available(const StftParams& p);  // the kernel can write the RGBA image itself
void build_fast_table_512(std::vector<float>& t);
// kernels_fast2048.hip: the same for 2048-point frames (one complex transform per wavefront), exact roots only.
int launch_stft_fast2048(const StftParams& p, void* hip_stream);
// kernels_fast4096.hip: float32 kernel for 4096-point frames with table twiddles (fast_mode 2): colour indices, or the
// RGBA image itself (given tile_scratch).  Returns 1 if it does not serve this launch.
int launch_stft_fast4096(const StftParams& p, void* hip_stream);
bool fast4096_available(const StftParams& p);
uint64_t redo_scratch_bytes();
bool fast4096_rgba_available(const StftParams& p);
void build_fast_table_2048(std::vector<float>& t);
// kernels_mw.hip: float32 kernel for 2048- and 4096-point frames with the context's twiddle table
// (returns 1 if it does not serve this size / layout).
int launch_stft_mw(const StftParams& p, void* hip_stream);
bool mw_u8_available(const StftParams& p);
// kernels_exact.hip: recomputes, in the exact arithmetic, the records of a SONO_OUT_FEATURES output that a fast kernel
// marked as too close to call (reserved[0] != 0) and clears the mark.  `p.twiddles` must be the double2 table.
// `p.tile_scratch` is the pass's queue of features_fixup_queue_bytes(p.fixup) bytes, p.fixup the capacity of its list.
int launch_features_fixup(const StftParams& p, void* hip_stream, bool listed = false);
bool features_fixup_available(const StftParams& p);
uint64_t features_fixup_queue_bytes(uint32_t capacity);
// kernels_exact.hip: the exact arithmetic for 512 / 1024 / 2048 / 4096-point frames, two in-register passes (returns 1 if it does not
// serve this launch; results are bit-identical to launch_stft_general(p, true, ...)).
int launch_stft_exact(const StftParams& p, void* hip_stream);
bool exact_available(const StftParams& p);
// kernels_post.hip
int launch_resample(const uint8_t* in, uint64_t in_frames, uint32_t bins, uint64_t out_width, uint64_t col0, uint64_t ncols,
                    uint8_t* out, void* hip_stream);
int launch_colorize(const uint8_t* frames, uint64_t ncols, uint32_t bins, uint32_t row0, uint32_t rows,
                    const uint32_t* lut, uint8_t* rgba, uint64_t img_width, uint64_t col0, void* hip_stream);
int launch_canvas_draw(const uint8_t* frames, uint64_t width, uint32_t bins, int64_t sy, int64_t sh, double dy, double dh,
                       uint32_t height, uint32_t background, const uint32_t* lut, uint8_t* canvas, void* hip_stream);
// True when launch_stft_fast has a kernel for this size / layout with uint8 output.
bool fast_u8_available(const StftParams& p);
// True when launch_stft_fast can write the RGBA image itself.
bool fast_rgba_available(const StftParams& p);
bool fast_features_available(const StftParams& p);
uint64_t fast_rgba_scratch_bytes();
int launch_pcm_to_f32(const void* src, uint32_t format_tag, uint32_t bits, uint32_t channels, uint32_t channel, uint64_t frames,
                      float* dst, void* hip_stream);
uint64_t launch_valu_peak(float* sink, uint32_t iterations, void* hip_stream);
// kernels_post.hip: the filter bank of a non-linear frequency scale (applyFilterBank, spectrogram.esm.js:479-488) over a
// [frame][half] plane of float32 magnitudes, then the plugin quantiser: colour indices of the bins [bin_lo, bin_hi) into
// out[(col0 + frame) * out_bins + bin - out_bin0].
struct FastScale;
int launch_filterbank_index(const float* mags, uint64_t nframes, uint32_t half, const int32_t* fb_bin, const double* w0, const double* w1,
                            uint32_t bin_lo, uint32_t bin_hi, uint32_t out_bins, uint32_t out_bin0, uint64_t col0, const FastScale& fs,
                            uint32_t n, uint8_t* out, void* hip_stream);
int launch_frame_peaks(const uint8_t* idx, uint64_t frames, uint32_t bins, int32_t* peak_bin, uint8_t* peak_idx,
                       void* hip_stream);

}  // namespace sono
#endif
