// Post-processing kernels of the plugin path: time resample, colour LUT + flip + row window,
// and the per-frame peak reduction.  Compile with -ffp-contract=off (resample accumulates in
// double in the reference's order).
#include <hip/hip_runtime.h>

#include "sono_device.h"
#include "sono_internal.h"
#include "sono_launch.h"

namespace sono {

// ------------------------------------------------------------------------------------------
// resample, spectrogram.esm.js:595-640.  One workgroup per output column a; threads over bins.
// The reference tests every input frame n against the column; frames far from the column have
// no overlap, so only [n0, n1) (two frames of margin) are tested, with the reference's own
// double expressions deciding membership and weight, in ascending n like its contrib list.
__global__ __launch_bounds__(256) void resample_kernel(const uint8_t* __restrict__ in, uint64_t F, uint32_t bins,
                                                      uint64_t W, uint64_t col0, double invIn, double invOut,
                                                      uint8_t* __restrict__ out) {
  const uint64_t a = col0 + blockIdx.x;  // column of the W-wide result; row blockIdx.x of `out`
  const double o = (double)a * invOut, l = o + invOut;
  double lo_f = floor((double)a * (double)F / (double)W) - 2;
  double hi_f = ceil((double)(a + 1) * (double)F / (double)W) + 2;
  const uint64_t n0 = lo_f < 0 ? 0 : (uint64_t)lo_f;
  const uint64_t n1 = hi_f > (double)F ? F : (uint64_t)hi_f;
  for (uint32_t u = threadIdx.x; u < bins; u += blockDim.x) {
    double accum = 0;
    bool any = false;
    for (uint64_t n = n0; n < n1; ++n) {
      const double s = (double)n * invIn, h = s + invIn;
      const double mn = h < l ? h : l, mx = s > o ? s : o;
      const double c = mn - mx;
      if (!(c > 0)) continue;
      const double weight = c / invOut;
      accum += weight * (double)in[n * bins + u];
      any = true;
    }
    uint32_t v = 0;
    if (any && fabs(accum) < 1e15) v = (uint32_t)((long long)trunc(accum) & 255);  // Uint8Array store
    out[(uint64_t)blockIdx.x * bins + u] = (uint8_t)v;
  }
}

int launch_resample(const uint8_t* in, uint64_t F, uint32_t bins, uint64_t W, uint64_t col0, uint64_t ncols, uint8_t* out,
                    void* stream_v) {
  if (W == 0 || F == 0 || ncols == 0) return SONO_OK;
  if (ncols > 0x7fffffffull || col0 > W || ncols > W - col0) return SONO_ERR_ARG;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  hipLaunchKernelGGL(resample_kernel, dim3((uint32_t)ncols), dim3(256), 0, stream, in, F, bins, W, col0, 1.0 / (double)F,
                     1.0 / (double)W, out);
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------------
// LUT + vertical flip (+ row window), spectrogram.esm.js:366-380.  frames is [width][bins];
// image row r holds bin bins-1-r.  A tile of 256 columns (frames) x 64 image rows goes through LDS so
// that both the byte reads (along bins) and the pixel writes (along columns) are contiguous: every
// image row gets a 1-KiB run of pixels per tile (one wavefront-wide 16-B-per-lane store), which keeps
// DRAM pages open on the write side -- the image rows are tens of megabytes apart.
// The runs are shifted row by row so that each starts on a 64-byte boundary of the image: the row pitch
// is rarely a multiple of 64 bytes, and a run that starts mid-sector leaves two partial sectors, which
// cost the memory a read-modify-write each.  Row r of tile t covers columns [t0 + s_r, t0 + 256 + s_r),
// s_r < 16 being the distance to the next boundary; the first tile also takes the s_r leading columns
// and the last tile stops at the image edge.  The tile therefore holds 16 extra columns.
constexpr int kTileCols = 256, kTileRows = 64, kTileStride = kTileRows + 4, kTileExtra = 16;

// frames: [ncols][bins] uint8.  Image: `rows` rows of `img_width` pixels; this call fills columns
// [col0, col0 + ncols).  Image row r holds bin bins-1-(row0+r).
__global__ __launch_bounds__(256) void colorize_kernel(const uint8_t* __restrict__ frames, uint64_t ncols, uint32_t bins,
                                                      uint32_t row0, uint32_t rows, const uint32_t* __restrict__ lut,
                                                      uint32_t* __restrict__ rgba, uint64_t img_width, uint64_t col0,
                                                      int aligned4) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[(kTileCols + kTileExtra) * kTileStride];  // [column][bin - e_lo]
  const uint64_t t0 = (uint64_t)blockIdx.x * kTileCols;
  const uint32_t r0 = blockIdx.y * kTileRows;               // first image row of the tile (relative to row0)
  const uint32_t nrows = min((uint32_t)kTileRows, rows - r0);
  const uint32_t ncol = (uint32_t)min((uint64_t)(kTileCols + kTileExtra), ncols - t0);  // columns held by the tile
  // image rows r0 .. r0+nrows-1 are bins e_hi-1 .. e_lo (descending); the tile stores bin e at [e - e_lo]
  const uint32_t e_hi = bins - (row0 + r0);
  const uint32_t e_lo = e_hi - nrows;
  const uint32_t tid = threadIdx.x;
  if (aligned4 && nrows == kTileRows) {
    // 16 threads read one frame's 64 bins as dwords
    for (uint32_t c = tid >> 4; c < ncol; c += 16) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(frames + (t0 + c) * bins + e_lo + 4 * (tid & 15));
      *reinterpret_cast<uint32_t*>(tile + c * kTileStride + 4 * (tid & 15)) = v;
    }
  } else {
    for (uint32_t i = tid; i < ncol * nrows; i += 256) {
      const uint32_t c = i / nrows, e = i - c * nrows;
      tile[c * kTileStride + e] = frames[(t0 + c) * bins + e_lo + e];
    }
  }
  __syncthreads();
  // one wavefront per image row, 4 consecutive columns per lane
  const uint32_t q = tid & 63;
  for (uint32_t rr = tid >> 6; rr < nrows; rr += 4) {
    const uint32_t e = nrows - 1 - rr;  // bin offset in the tile
    uint32_t* row = rgba + (uint64_t)(r0 + rr) * img_width + col0 + t0;  // pixel of the tile's column 0
    // pixels to the next 64-byte boundary (the image base is at least 4-byte aligned)
    const uint32_t s = (16u - (uint32_t)((reinterpret_cast<uintptr_t>(row) >> 2) & 15u)) & 15u;
    const uint32_t hi = min(ncol, (uint32_t)kTileCols + s);  // this row's run is the tile's columns [s, hi)
    const uint32_t c = s + 4 * q;
    if (c < hi) {
      uint32_t px[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t idx = c + j < hi ? tile[(c + j) * kTileStride + e] : 0u;
        px[j] = lut ? lut[idx] : (idx | (idx << 8) | (idx << 16) | 0xff000000u);
      }
      uint32_t* dst = row + c;
      if (c + 3 < hi && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(px[0], px[1], px[2], px[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < hi) dst[j] = px[j];
      }
    }
    // the first tile also owns the columns before its first boundary
    if (t0 == 0 && q < s && q < hi) {
      const uint32_t idx = tile[q * kTileStride + e];
      row[q] = lut ? lut[idx] : (idx | (idx << 8) | (idx << 16) | 0xff000000u);
    }
  }
}

int launch_colorize(const uint8_t* frames, uint64_t ncols, uint32_t bins, uint32_t row0, uint32_t rows,
                    const uint32_t* lut, uint8_t* rgba, uint64_t img_width, uint64_t col0, void* stream_v) {
  if (ncols == 0 || rows == 0) return SONO_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  const uint64_t gx = (ncols + kTileCols - 1) / kTileCols;
  const uint32_t gy = (rows + kTileRows - 1) / kTileRows;
  if (gx > 0x7fffffffull || gy > 65535u) return SONO_ERR_ARG;
  // dword reads need every tile's first bin and the frame stride to be multiples of four
  const int aligned4 = ((bins & 3) == 0 && ((bins - row0) & 3) == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0) ? 1 : 0;
  hipLaunchKernelGGL(colorize_kernel, dim3((uint32_t)gx, gy), dim3(256), 0, stream, frames, ncols, bins, row0, rows,
                     lut, reinterpret_cast<uint32_t*>(rgba), img_width, col0, aligned4);
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------------
// The canvas step of drawSpectrogram, spectrogram.esm.js:357-361,377-381: the cropped rows
// [sy, sy + sh) of the flipped LUT image are stretched over canvas rows [dy, dy + dh) of a
// `height`-row canvas; everything else keeps the background.  The browser's drawImage filter is
// implementation-defined; this one is bilinear along y with clamped edges, sampled at pixel centres:
//   v = (y + 0.5 - dy) * sh / dh - 0.5 clamped to [0, sh - 1], i0 = floor(v), t = float(v - i0),
//   channel = floor(a + (b - a) * t + 0.5) in float32 (no fused multiply-add),
// canvas rows whose centre lies outside [dy, dy + dh) are not covered.  Source rows outside the
// image are transparent black (createImageBitmap's rule for a crop rectangle beyond the source).
// A pixel with alpha < 255 is composited over the background (source-over, straight alpha, float32).
struct CanvasRow {
  int64_t i0, i1;  // rows of the crop window, or i0 < 0: not covered
  float t;
};
__host__ __device__ inline CanvasRow canvas_row(uint32_t y, int64_t sh, double dy, double dh) {
  CanvasRow r = {-1, -1, 0.0f};
  const double yc = (double)y + 0.5;
  if (sh <= 0 || !(dh > 0) || !(yc >= dy) || !(yc < dy + dh)) return r;
  double v = (yc - dy) * ((double)sh / dh) - 0.5;
  if (v < 0) v = 0;
  if (v > (double)(sh - 1)) v = (double)(sh - 1);
  const double f = floor(v);
  r.i0 = (int64_t)f;
  r.i1 = r.i0 + 1 < sh ? r.i0 + 1 : sh - 1;
  r.t = (float)(v - f);
  return r;
}

__device__ inline uint32_t canvas_mix(uint32_t a, uint32_t b, float t, uint32_t bg) {
  uint32_t px = 0;
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const float fa = (float)((a >> (8 * ch)) & 255u), fb = (float)((b >> (8 * ch)) & 255u);
    const float d = (fb - fa) * t;
    px |= (uint32_t)(fa + d + 0.5f) << (8 * ch);
  }
  const uint32_t sa = px >> 24;
  if (sa == 255u || bg == 0u) return px;
  // source-over on the background colour
  const float fsa = (float)sa * (1.0f / 255.0f), fda = (float)(bg >> 24) * (1.0f / 255.0f);
  const float k = fda * (1.0f - fsa), oa = fsa + k;
  if (!(oa > 0.0f)) return 0u;
  uint32_t o = (uint32_t)(oa * 255.0f + 0.5f) << 24;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float sc = (float)((px >> (8 * ch)) & 255u), dc = (float)((bg >> (8 * ch)) & 255u);
    const float num = sc * fsa + dc * k;
    o |= (uint32_t)(num / oa + 0.5f) << (8 * ch);
  }
  return o;
}

constexpr int kCanvasBand = 32;  // canvas rows per workgroup

// frames: [width][bins] uint8 (after resample).  One workgroup: `cols` image columns x kCanvasBand
// canvas rows; the source bytes it needs (`span` crop rows at most, host-computed) go through LDS as
// [column][row - lo], read along bins and written along columns.
__global__ __launch_bounds__(256) void canvas_draw_kernel(const uint8_t* __restrict__ frames, uint64_t width, uint32_t bins,
                                                         int64_t sy, int64_t sh, double dy, double dh, uint32_t height,
                                                         uint32_t background, const uint32_t* __restrict__ lut,
                                                         uint32_t cols, uint32_t span, uint32_t* __restrict__ canvas) {
  extern __shared__ __attribute__((aligned(16))) uint8_t src[];  // [cols][span]
  const uint64_t x0 = (uint64_t)blockIdx.x * cols;
  const uint32_t ncol = (uint32_t)min((uint64_t)cols, width - x0);
  const uint32_t y0 = blockIdx.y * kCanvasBand, y1 = min(height, y0 + kCanvasBand);
  // crop rows this band touches: the row map is monotonic in y
  int64_t lo = -1, hi = -1;
  for (uint32_t y = y0; y < y1; ++y) {
    const CanvasRow r = canvas_row(y, sh, dy, dh);
    if (r.i0 < 0) continue;
    if (lo < 0) lo = r.i0;
    hi = r.i1;
  }
  const uint32_t nrow = lo < 0 ? 0u : (uint32_t)(hi - lo + 1);  // <= span by the host's bound
  for (uint32_t i = threadIdx.x; i < ncol * nrow; i += 256) {
    const uint32_t c = i / nrow, k = i - c * nrow;
    const int64_t row = sy + lo + k;  // row of the flipped image: bin bins-1-row
    src[c * span + k] = (row >= 0 && row < (int64_t)bins) ? frames[(x0 + c) * bins + (bins - 1 - (uint32_t)row)] : 0;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ncol * (y1 - y0); i += 256) {
    const uint32_t yy = i / ncol, c = i - yy * ncol, y = y0 + yy;
    const CanvasRow r = canvas_row(y, sh, dy, dh);
    uint32_t px = background;
    if (r.i0 >= 0) {
      const int64_t ra = sy + r.i0, rb = sy + r.i1;
      const uint32_t ia = src[c * span + (uint32_t)(r.i0 - lo)], ib = src[c * span + (uint32_t)(r.i1 - lo)];
      const uint32_t a = (ra >= 0 && ra < (int64_t)bins) ? (lut ? lut[ia] : (ia | (ia << 8) | (ia << 16) | 0xff000000u)) : 0u;
      const uint32_t b = (rb >= 0 && rb < (int64_t)bins) ? (lut ? lut[ib] : (ib | (ib << 8) | (ib << 16) | 0xff000000u)) : 0u;
      px = canvas_mix(a, b, r.t, background);
    }
    canvas[(uint64_t)y * width + x0 + c] = px;
  }
}

int launch_canvas_draw(const uint8_t* frames, uint64_t width, uint32_t bins, int64_t sy, int64_t sh, double dy, double dh,
                       uint32_t height, uint32_t background, const uint32_t* lut, uint8_t* canvas, void* stream_v) {
  if (width == 0 || height == 0) return SONO_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  // widest run of crop rows any band of kCanvasBand canvas rows touches
  uint32_t span = 1;
  for (uint32_t y0 = 0; y0 < height; y0 += kCanvasBand) {
    int64_t lo = -1, hi = -1;
    for (uint32_t y = y0; y < y0 + kCanvasBand && y < height; ++y) {
      const CanvasRow r = canvas_row(y, sh, dy, dh);
      if (r.i0 < 0) continue;
      if (lo < 0) lo = r.i0;
      hi = r.i1;
    }
    if (lo >= 0 && (uint64_t)(hi - lo + 1) > span) span = (uint32_t)(hi - lo + 1);
  }
  span = (span + 3u) & ~3u;
  uint32_t cols = 64;
  while (cols > 1 && (uint64_t)cols * span > 48u * 1024u) cols >>= 1;
  if ((uint64_t)cols * span > 64u * 1024u) return SONO_ERR_ARG;
  const uint64_t gx = (width + cols - 1) / cols;
  const uint32_t gy = (height + kCanvasBand - 1) / kCanvasBand;
  if (gx > 0x7fffffffull || gy > 65535u) return SONO_ERR_ARG;
  hipLaunchKernelGGL(canvas_draw_kernel, dim3((uint32_t)gx, gy), dim3(256), cols * span, stream, frames, width, bins, sy, sh,
                     dy, dh, height, background, lut, cols, span, reinterpret_cast<uint32_t*>(canvas));
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------------
// Per-frame peak: first bin holding the maximum index (strict > in ascending bin order, the
// rule of spectrogram.esm.js:196).  One wavefront per frame.
__global__ __launch_bounds__(256) void frame_peaks_kernel(const uint8_t* __restrict__ idx, uint64_t frames, uint32_t bins,
                                                         int32_t* __restrict__ peak_bin, uint8_t* __restrict__ peak_idx) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t f = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= frames) return;
  const uint8_t* row = idx + f * bins;
  // key = value << 16 | (0xffff - bin): the largest key is the largest value at the smallest bin
  uint32_t best = 0;
  for (uint32_t k = lane; k < bins; k += 64) {
    const uint32_t key = ((uint32_t)row[k] << 16) | (0xffffu - k);
    best = key > best ? key : best;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t other = __shfl_xor(best, off);
    best = other > best ? other : best;
  }
  if (lane == 0) {
    peak_bin[f] = (int32_t)(0xffffu - (best & 0xffffu));
    peak_idx[f] = (uint8_t)(best >> 16);
  }
}

int launch_frame_peaks(const uint8_t* idx, uint64_t frames, uint32_t bins, int32_t* peak_bin, uint8_t* peak_idx,
                       void* stream_v) {
  if (frames == 0) return SONO_OK;
  if (bins == 0 || bins > 0xffffu) return SONO_ERR_ARG;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  const uint64_t blocks = (frames + 3) / 4;
  if (blocks > 0x7fffffffull) return SONO_ERR_ARG;
  hipLaunchKernelGGL(frame_peaks_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, idx, frames, bins, peak_bin,
                     peak_idx);
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

// One channel of interleaved PCM -> float32.  KIND: 0 u8, 1 i16, 2 i24, 3 i32, 4 f32.
template <int KIND>
__global__ __launch_bounds__(256) void pcm_to_f32_kernel(const uint8_t* __restrict__ src, uint32_t channels, uint32_t channel,
                                                         uint64_t frames, float* __restrict__ dst) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= frames) return;
  constexpr int kBytes = KIND == 0 ? 1 : KIND == 1 ? 2 : KIND == 2 ? 3 : 4;
  const uint8_t* p = src + (i * channels + channel) * kBytes;
  float v;
  if constexpr (KIND == 0) {
    v = ((float)p[0] - 128.0f) * (1.0f / 128.0f);
  } else if constexpr (KIND == 1) {
    v = (float)(int16_t)(p[0] | (p[1] << 8)) * (1.0f / 32768.0f);
  } else if constexpr (KIND == 2) {
    const int32_t w = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8;
    v = (float)w * (1.0f / 8388608.0f);
  } else if constexpr (KIND == 3) {
    const int32_t w = (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24);
    v = (float)((double)w * (1.0 / 2147483648.0));
  } else {
    const uint32_t w = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    v = __uint_as_float(w);
  }
  dst[i] = v;
}

int launch_pcm_to_f32(const void* src, uint32_t format_tag, uint32_t bits, uint32_t channels, uint32_t channel, uint64_t frames,
                      float* dst, void* stream_v) {
  if (!frames) return SONO_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  const uint8_t* s = reinterpret_cast<const uint8_t*>(src);
  const dim3 grid((unsigned)((frames + 255) / 256)), block(256);
  if (format_tag == 3 && bits == 32) hipLaunchKernelGGL(pcm_to_f32_kernel<4>, grid, block, 0, stream, s, channels, channel, frames, dst);
  else if (format_tag == 1 && bits == 8) hipLaunchKernelGGL(pcm_to_f32_kernel<0>, grid, block, 0, stream, s, channels, channel, frames, dst);
  else if (format_tag == 1 && bits == 16) hipLaunchKernelGGL(pcm_to_f32_kernel<1>, grid, block, 0, stream, s, channels, channel, frames, dst);
  else if (format_tag == 1 && bits == 24) hipLaunchKernelGGL(pcm_to_f32_kernel<2>, grid, block, 0, stream, s, channels, channel, frames, dst);
  else if (format_tag == 1 && bits == 32) hipLaunchKernelGGL(pcm_to_f32_kernel<3>, grid, block, 0, stream, s, channels, channel, frames, dst);
  else return SONO_ERR_ARG;
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------------
// What the vector ALU of this GPU issues when nothing else is in the way: sixteen wavefronts per CU (the
// transform kernels' occupancy), each a long run of independent v_fma_f32.  The STFT kernels are priced
// against this number next to the HBM roofline, because at hop = N/4 their arithmetic, not their bytes,
// is what fills the time.
constexpr int kPeakChains = 16, kPeakUnroll = 8;
__global__ __launch_bounds__(1024) void valu_peak_kernel(float* __restrict__ sink, uint32_t iterations, float a, float b) {
  float v[kPeakChains];
#pragma unroll
  for (int i = 0; i < kPeakChains; ++i) v[i] = (float)(threadIdx.x + i);
  for (uint32_t it = 0; it < iterations; ++it) {
#pragma unroll
    for (int u = 0; u < kPeakUnroll; ++u)
#pragma unroll
      for (int i = 0; i < kPeakChains; ++i) v[i] = __builtin_fmaf(v[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < kPeakChains; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;  // keeps the chains alive; never true for the a, b the host passes
}

// Enqueues the probe; returns the number of wavefront-level v_fma_f32 it issues.
uint64_t launch_valu_peak(float* sink, uint32_t iterations, void* stream_v) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  const int cus = device_cu_count();
  hipLaunchKernelGGL(valu_peak_kernel, dim3(cus), dim3(1024), 0, stream, sink, iterations, 0.999f, 0.001f);
  return (uint64_t)cus * 16 * iterations * kPeakUnroll * kPeakChains;
}

// Non-linear frequency scales in the float32 arithmetic: out[y] = m[o] * w0[y] + m[o + 1] * w1[y] with o = fb_bin[y], over a
// [frame][half] plane of float32 magnitudes, then the plugin quantiser.  One NaN or infinite magnitude turns every bin of
// its frame NaN: the reference's accumulator adds m[s] * bank[y][s] for EVERY bin s, and NaN * 0 is NaN
// (spectrogram.esm.js:479-488).  One wavefront per frame; tap tables and the frame's magnitudes in LDS.
__global__ __launch_bounds__(256) void filterbank_index_kernel(const float* __restrict__ mags, uint64_t nframes, uint32_t half,
                                                               const int32_t* __restrict__ fb_bin, const double* __restrict__ w0,
                                                               const double* __restrict__ w1, uint32_t bin_lo, uint32_t bin_hi,
                                                               uint32_t out_bins, uint32_t out_bin0, uint64_t col0, FastScale fs,
                                                               float unscale, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t* tap = reinterpret_cast<int32_t*>(smem);
  float* wa = reinterpret_cast<float*>(tap + half);
  float* wb = wa + half;
  float* frames = wb + half;  // [4][half]
  for (uint32_t y = threadIdx.x; y < half; y += 256) {
    tap[y] = fb_bin[y];
    wa[y] = (float)w0[y];
    wb[y] = (float)w1[y];
  }
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* m = frames + wave * half;
  for (uint64_t f = (uint64_t)blockIdx.x * 4 + wave; f < nframes; f += (uint64_t)gridDim.x * 4) {
    const float* row = mags + f * half;
    int bad = 0;
    for (uint32_t y = lane; y < half; y += 64) {
      const float v = row[y];
      m[y] = v;
      bad |= !isfinite(v);
    }
    bad = __any(bad);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint8_t* orow = out + (col0 + f) * out_bins - out_bin0;
    for (uint32_t y = bin_lo + lane; y < bin_hi; y += 64) {
      const int32_t o = tap[y];
      float acc = 0.0f;
      if (o >= 0 && o < (int32_t)half) acc = m[o] * wa[y];
      if (o + 1 >= 0 && o + 1 < (int32_t)half) acc += m[o + 1] * wb[y];
      if (bad) acc = __builtin_nanf("");
      const float x = acc * unscale;  // fast_index takes |X|^2 before the 2/N scaling
      // (a negative sum -- negative weights -- has no logarithm: NaN in the reference, index 0)
      orow[y] = (uint8_t)fast_index(x < 0.0f ? __builtin_nanf("") : x * x, fs);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

int launch_filterbank_index(const float* mags, uint64_t nframes, uint32_t half, const int32_t* fb_bin, const double* w0, const double* w1,
                            uint32_t bin_lo, uint32_t bin_hi, uint32_t out_bins, uint32_t out_bin0, uint64_t col0, const FastScale& fs,
                            uint32_t n, uint8_t* out, void* hip_stream) {
  if (!mags || !fb_bin || !w0 || !w1 || !out || bin_lo >= bin_hi || bin_hi > half) return SONO_ERR_ARG;
  const size_t lds = (size_t)half * (12 + 16);
  if (lds > 64 * 1024) return SONO_ERR_ARG;
  const uint64_t want = (nframes + 3) / 4;
  const uint32_t grid = (uint32_t)(want < (uint64_t)device_cu_count() * 8 ? want : (uint64_t)device_cu_count() * 8);
  hipLaunchKernelGGL(filterbank_index_kernel, dim3(grid ? grid : 1), dim3(256), lds, reinterpret_cast<hipStream_t>(hip_stream), mags,
                     nframes, half, fb_bin, w0, w1, bin_lo, bin_hi, out_bins, out_bin0, col0, fs, (float)((double)n / 2.0), out);
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
