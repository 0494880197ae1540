// General STFT kernel: the reference's radix-2 decimation-in-time dataflow with per-stage
// twiddle tables, for any power-of-two size, in two arithmetics.
//
//   float  : float32 butterflies (dB within 0.01 dB of the JavaScript, indices within 1 LSB)
//   double : double butterflies, float32 rounding at every store, no FMA contraction -- the exact
//            operation sequence of spectrogramWorker.js:233-252 and spectrogram.esm.js:175-193
//            (JavaScript doubles over Float32Array storage), so magnitudes are bit-identical.
//
// Because the twiddles come from a table, the same kernel serves the worker's accurate factors
// and the plugin's recurrence factors (which are not exact roots of unity, so no algebraic
// shortcut between stages is allowed).
//
// One 256-thread workgroup transforms FT consecutive frames, one after the other, in LDS.  For
// [frame][bin] outputs each frame's bins are stored directly; for the RGBA image the colour
// indices of the FT frames are kept in an LDS tile and written as runs of FT pixels per image
// row, so that global stores are contiguous instead of one pixel per row.
//
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "sono_device.h"
#include "sono_internal.h"
#include "sono_launch.h"

namespace sono {

constexpr int kThreads = 256;

template <typename T>
struct Tw2;
template <>
struct Tw2<float> {
  using type = float2;
};
template <>
struct Tw2<double> {
  using type = double2;
};

template <typename In>
__device__ __forceinline__ float load_sample(const In* x, uint64_t i);
template <>
__device__ __forceinline__ float load_sample<float>(const float* x, uint64_t i) {
  return x[i];
}
template <>
__device__ __forceinline__ float load_sample<int16_t>(const int16_t* x, uint64_t i) {
  return (float)x[i] * (1.0f / 32768.0f);
}

// colour index of a filter-bank output m (already the reference's float32 magnitude)
template <typename A>
__device__ __forceinline__ uint32_t fb_index(float m, const StftParams& p, const FastScale& fs, double two_over_n) {
  if constexpr (sizeof(A) == 8) {
    return plugin_index_exact(m, p.gain_db, p.range_db);
  } else {
    const float x = m * (float)(1.0 / two_over_n);  // fast_index takes |X|^2 before the 2/N scaling
    return fast_index(x * x, fs);
  }
}

template <typename A /* float | double */, typename In>
__global__ __launch_bounds__(kThreads) void stft_general_kernel(StftParams p, FastScale fs, uint32_t frames_per_block,
                                                               uint32_t tile_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* re = reinterpret_cast<float*>(smem);
  float* im = re + p.n;
  unsigned char* tile = reinterpret_cast<unsigned char*>(im + p.n);  // [frames_per_block][tile_stride]
  float* fb_mag = reinterpret_cast<float*>(tile + (size_t)frames_per_block * tile_stride);  // [n/2], filter-bank mode only
  const bool fb = p.fb_bin != nullptr;  // non-linear frequency scale: re[y] holds the filtered magnitude after the FFT

  using TW = typename Tw2<A>::type;
  const TW* __restrict__ tw = reinterpret_cast<const TW*>(p.twiddles);
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  const uint32_t n = p.n, half = n >> 1;
  const uint32_t tid = threadIdx.x;
  const uint64_t first = (uint64_t)blockIdx.x * frames_per_block;
  const uint32_t count = (uint32_t)min((uint64_t)frames_per_block, p.nframes - first);
  const uint32_t nb = p.bin_hi - p.bin_lo;
  const double two_over_n = 2.0 / (double)n;
  const ExactPred pred = make_exact_pred(p.scale_mode, n, p.gain_db, p.range_db);  // (exact flavour: see sono_device.h)

  // fix-up pass over a SONO_OUT_FEATURES output of a float32 kernel (frames_per_block == kThreads): only the
  // records it marked (reserved[0] != 0) are recomputed
  __shared__ uint32_t s_todo[kThreads];
  __shared__ uint32_t s_ntodo;
  const bool fixup = p.fixup != 0;
  if (fixup) {
    if (tid == 0) s_ntodo = 0;
    __syncthreads();
    if (tid < count && reinterpret_cast<const sono_frame_feature*>(p.out)[p.col0 + first + tid].reserved[0] != 0)
      s_todo[atomicAdd(&s_ntodo, 1u)] = tid;
    __syncthreads();
    if (tid == 0 && s_ntodo && p.tile_scratch) atomicAdd(reinterpret_cast<uint32_t*>(p.tile_scratch) + 1, s_ntodo);  // statistics only
  }
  const uint32_t rounds = fixup ? s_ntodo : count;
  for (uint32_t it = 0; it < rounds; ++it) {
    const uint32_t fl = fixup ? s_todo[it] : it;
    const uint64_t frame = p.frame0 + first + fl;
    const uint64_t a = frame * p.hop;
    // window, spectrogramWorker.js:61-64 / spectrogram.esm.js:172-174: float32(x*w) at the
    // bit-reversed position (a float32 product equals the float32 rounding of the double product)
    for (uint32_t j = tid; j < n; j += kThreads) {
      const uint32_t k = __brev(j) >> (32 - p.log2n);
      re[k] = load_sample<In>(x, a + j) * p.window[j];
      im[k] = 0.0f;
    }
    for (uint32_t T = 1; T < n; T <<= 1) {
      __syncthreads();
      for (uint32_t b = tid; b < half; b += kThreads) {
        const uint32_t z = b & (T - 1);
        const uint32_t k = ((b - z) << 1) + z;
        const uint32_t M = k + T;
        const TW w = tw[T - 1 + z];
        const A g = w.x, s = w.y;
        const A oM = re[M], lM = im[M], ok = re[k], lk = im[k];
        const A m = g * oM - s * lM;
        const A y = g * lM + s * oM;
        re[M] = (float)(ok - m);
        im[M] = (float)(lk - y);
        re[k] = (float)(ok + m);
        im[k] = (float)(lk + y);
      }
    }
    __syncthreads();
    if (fb) {
      // applyFilterBank, spectrogram.esm.js:479-488: out[i] = sum over EVERY bin s of mag[s] * bank[i][s] in a
      // Float32Array accumulator.  A row is zero outside its two taps (fb_bin[i], fb_bin[i] + 1), so the sum is
      // float32(float32(mag[o] * w0) + mag[o+1] * w1) -- unless some magnitude of the frame is NaN or infinite:
      // then mag[s] * 0 is NaN for that s and every output of the frame is NaN.
      int bad = 0;
      for (uint32_t y = tid; y < half; y += kThreads) {
        const float r = re[y], i = im[y];
        float m;
        if constexpr (sizeof(A) == 8) m = plugin_mag_exact(r, i, two_over_n);
        else m = sqrtf(r * r + i * i) * (float)two_over_n;
        fb_mag[y] = m;
        bad |= !isfinite(m);
      }
      bad = __syncthreads_or(bad);
      for (uint32_t y = tid; y < half; y += kThreads) {
        const int32_t o = p.fb_bin[y];
        float acc = 0.0f;
        if constexpr (sizeof(A) == 8) {
          if (o >= 0 && o < (int32_t)half) acc = (float)((double)acc + (double)fb_mag[o] * p.fb_w0[y]);
          if (o + 1 >= 0 && o + 1 < (int32_t)half) acc = (float)((double)acc + (double)fb_mag[o + 1] * p.fb_w1[y]);
        } else {
          if (o >= 0 && o < (int32_t)half) acc = fb_mag[o] * (float)p.fb_w0[y];
          if (o + 1 >= 0 && o + 1 < (int32_t)half) acc += fb_mag[o + 1] * (float)p.fb_w1[y];
        }
        re[y] = bad ? __builtin_nanf("") : acc;
      }
      __syncthreads();
    }
    if (p.out_kind == SONO_OUT_FEATURES) {
      // per-frame peak: first bin whose float32 magnitude exceeds all earlier ones (strict >)
      float best_m = -1.0f;
      uint32_t best_k = p.bin_lo;
      for (uint32_t y = p.bin_lo + tid; y < p.bin_hi; y += kThreads) {
        const float r = re[y], i = im[y];
        float m;
        if (fb) {
          m = r;
        } else if constexpr (sizeof(A) == 8) {
          const double e = r, s = i;
          const double mag = sqrt(e * e + s * s);
          m = (p.scale_mode == SONO_SCALE_WORKER_LOG5) ? (float)mag : (float)(two_over_n * mag);
        } else {
          m = sqrtf(r * r + i * i);
        }
        if (m > best_m) {
          best_m = m;
          best_k = y;
        }
      }
      __shared__ float red_m[kThreads];
      __shared__ uint32_t red_k[kThreads];
      red_m[tid] = best_m;
      red_k[tid] = best_k;
      __syncthreads();
      for (uint32_t s = kThreads / 2; s > 0; s >>= 1) {
        if (tid < s) {
          const float om = red_m[tid + s];
          const uint32_t ok = red_k[tid + s];
          if (om > red_m[tid] || (om == red_m[tid] && ok < red_k[tid])) {
            red_m[tid] = om;
            red_k[tid] = ok;
          }
        }
        __syncthreads();
      }
      if (tid == 0) {
        const uint32_t k = red_k[0];
        const float r = re[k], i = im[k];
        uint32_t idx;
        if (fb) {
          idx = fb_index<A>(r, p, fs, two_over_n);
        } else if constexpr (sizeof(A) == 8) {
          idx = (p.scale_mode == SONO_SCALE_WORKER_LOG5)
                    ? worker_index_exact(r, i)
                    : plugin_index_exact(plugin_mag_exact(r, i, two_over_n), p.gain_db, p.range_db);
        } else {
          idx = fast_index(r * r + i * i, fs);
        }
        sono_frame_feature ft;
        ft.peak_bin = (int32_t)k;
        ft.peak_idx = (uint8_t)idx;
        ft.reserved[0] = ft.reserved[1] = ft.reserved[2] = 0;
        reinterpret_cast<sono_frame_feature*>(p.out)[p.col0 + first + fl] = ft;
      }
      __syncthreads();
      continue;
    }
    for (uint32_t y = p.bin_lo + tid; y < p.bin_hi; y += kThreads) {
      const float r = re[y], i = im[y];
      if (p.out_kind == SONO_OUT_U8_INDEX || p.out_kind == SONO_OUT_RGBA8) {
        uint32_t idx;
        if (fb) {
          idx = fb_index<A>(r, p, fs, two_over_n);
        } else if constexpr (sizeof(A) == 8) {
          idx = (p.scale_mode == SONO_SCALE_WORKER_LOG5)
                    ? worker_index_exact_p(r, i, pred)
                    : plugin_index_exact_p(r, i, two_over_n, p.gain_db, p.range_db, pred);
        } else {
          idx = fast_index(r * r + i * i, fs);
        }
        if (p.out_kind == SONO_OUT_U8_INDEX)
          reinterpret_cast<uint8_t*>(p.out)[(p.col0 + first + fl) * p.out_bins + (y - p.out_bin0)] = (uint8_t)idx;
        else
          tile[fl * tile_stride + (y - p.bin_lo)] = (unsigned char)idx;
      } else {
        float v;
        if (fb) {
          if (p.out_kind == SONO_OUT_F32_MAG) {
            v = r;
          } else {
            const double sv = ((double)r > 1e-12) ? (double)r : 1e-12;
            v = (float)(20 * log10(sv));
          }
        } else if constexpr (sizeof(A) == 8) {
          const double e = r, s = i;
          const double mag = sqrt(e * e + s * s);
          const float m = (p.scale_mode == SONO_SCALE_WORKER_LOG5) ? (float)mag : (float)(two_over_n * mag);
          if (p.out_kind == SONO_OUT_F32_MAG) {
            v = m;
          } else {
            const double sv = ((double)m > 1e-12) ? (double)m : 1e-12;
            v = (float)(20 * log10(sv));
          }
        } else {
          const float msq = r * r + i * i;
          if (p.out_kind == SONO_OUT_F32_MAG)
            v = sqrtf(msq) * ((p.scale_mode == SONO_SCALE_WORKER_LOG5) ? 1.0f : (float)two_over_n);
          else
            v = fast_db(msq, fs);
        }
        reinterpret_cast<float*>(p.out)[(p.col0 + first + fl) * p.out_bins + (y - p.out_bin0)] = v;
      }
    }
    __syncthreads();  // re/im are rewritten by the next frame
  }

  if (p.out_kind == SONO_OUT_RGBA8) {
    // image row of bin y is row_of_bin0 - y; columns first.. are contiguous in memory
    uint32_t* img = reinterpret_cast<uint32_t*>(p.out);
    const uint32_t total = nb * count;
    for (uint32_t i = tid; i < total; i += kThreads) {
      const uint32_t by = i / count, c = i - by * count;
      const uint32_t y = p.bin_lo + by;
      const uint32_t idx = tile[c * tile_stride + by];
      const uint32_t px = p.lut ? p.lut[idx] : (idx | (idx << 8) | (idx << 16) | 0xff000000u);
      const int64_t row = (int64_t)p.row_of_bin0 - (int64_t)y;
      img[(uint64_t)row * p.width + p.col0 + first + c] = px;
    }
  }
}

int launch_stft_general(const StftParams& p, bool exact_f64, void* stream_v) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  if (p.nframes == 0) return SONO_OK;
  const uint32_t nb = p.bin_hi - p.bin_lo;
  // RGBA: keep FT frames of indices in LDS; the row stride in dwords is odd so that the
  // column-major read-out does not hit one bank.
  uint32_t ft = 1, stride = 0;
  size_t lds = (size_t)p.n * 2 * sizeof(float);
  if (p.out_kind == SONO_OUT_RGBA8) {
    stride = ((nb + 3) / 4) * 4;
    if (((stride / 4) & 1) == 0) stride += 4;
    const size_t budget = 64 * 1024;
    ft = (uint32_t)(budget / stride);
    if (ft > 64) ft = 64;
    if (ft < 1) ft = 1;
    lds += (size_t)ft * stride;
  } else {
    ft = p.fixup ? kThreads : 8;
  }
  if (p.fb_bin) {
    if (p.scale_mode != SONO_SCALE_PLUGIN_DB || !p.fb_w0 || !p.fb_w1) return SONO_ERR_ARG;
    lds = (size_t)p.n * 2 * sizeof(float) + (size_t)ft * stride + (size_t)(p.n / 2) * sizeof(float);
  }
  const uint64_t blocks = (p.nframes + ft - 1) / ft;
  if (blocks > 0x7fffffffull) return SONO_ERR_ARG;
  const FastScale fs = make_fast_scale(p.scale_mode, p.n, p.gain_db, p.range_db);
  dim3 grid((uint32_t)blocks), block(kThreads);
#define SONO_LAUNCH(A, In)                                                                                  \
  do {                                                                                                      \
    SONO_GRANT_LDS((&stft_general_kernel<A, In>), lds);                                                     \
    hipLaunchKernelGGL((stft_general_kernel<A, In>), grid, block, lds, stream, p, fs, ft, stride);          \
  } while (0)
  if (exact_f64) {
    if (p.in_kind == SONO_IN_I16) SONO_LAUNCH(double, int16_t); else SONO_LAUNCH(double, float);
  } else {
    if (p.in_kind == SONO_IN_I16) SONO_LAUNCH(float, int16_t); else SONO_LAUNCH(float, float);
  }
#undef SONO_LAUNCH
  note_kernel(p.fixup ? "stft_general_kernel<double,fixup>" : exact_f64 ? "stft_general_kernel<double>" : "stft_general_kernel<float>");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
