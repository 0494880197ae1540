// Exact-arithmetic STFT for 512-, 1024-, 2048- and 4096-point frames, in registers.
//
// Same operation sequence as the general kernel's double flavour -- the radix-2 decimation-in-time
// network of spectrogramWorker.js:206-253 / spectrogram.esm.js:166-193 with per-stage twiddle tables,
// double butterflies over float32 storage, no FMA contraction -- so the results are bit-identical to it
// and to the JavaScript.  Only the schedule differs: a butterfly's inputs and factor are fixed by the
// network, not by who computes it, so the stages are run as two in-register passes with one transpose
// through LDS in between, instead of one LDS round trip with a workgroup barrier per stage.
//
// G lanes own a frame and hold R = N / G values each (N = 512: 32 x 16, 1024: 32 x 32, 2048: 64 x 32,
// 4096: 64 x 64).  The working array A[0..N-1] (bit-reversed input order):
//   pass 1, stages T = 1..R/2   : lane l holds A[R l + r] in register r; the factor of a butterfly depends
//                                 on r only (tw[T-1 + (r mod T)]), wave-uniform, read as scalars;
//   stage T = R (512 and 2048)  : the partners A[k], A[k + R] sit in the same register of lanes 2m and 2m + 1; each
//                                 lane fetches the other's value (DPP), both evaluate the reference's product
//                                 t = w A[k + R] in the same double operations, the even lane keeps A[k] + t and
//                                 the odd one A[k] - t;
//   transpose                   : lane l holds A[G r + l] in register r;
//   pass 2, stages T = G..N/2   : partner registers are T/G apart; the factor tw[T-1 + G (r mod T/G) + l]
//                                 differs per lane and is read from an [R-1][G] double2 copy in LDS.
// A[k] is then X[k] in natural order: lane l, register r < R/2 holds bin G r + l.
//
// Colour indices come from the exact quantisers behind their float32 predictor (sono_device.h): the double
// sqrt / log10 run only for the bins whose quantiser argument is within rounding of an integer.
//
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "sono_device.h"
#include "sono_internal.h"
#include "sono_launch.h"

namespace sono {
namespace exactreg {

__host__ __device__ constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
__host__ __device__ constexpr int brevc(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

template <int N>
struct Geo {
  static constexpr int G = N >= 2048 ? 64 : 32;   // lanes per frame
  static constexpr int R = N / G;                 // values per lane
  static constexpr int kLogG = ilog2c(G), kLogR = ilog2c(R);
  static constexpr bool kCross = G > R;           // one stage (T = R) pairs neighbouring lanes
  static constexpr int kPadShift = R > 32 ? 6 : 5;  // plane cell of A[i]: i + (i >> kPadShift), conflict-free both ways
  static constexpr int kPlane = N + (N >> kPadShift);
  static constexpr int kWaves = N == 4096 ? 4 : 8;  // wavefronts per workgroup
  static constexpr int kFramesPerWave = 64 / G;
  static constexpr int kFramesPerWg = kWaves * kFramesPerWave;
  static constexpr int kTwRows = R - 1;
  static constexpr int kTwBytes = kTwRows * G * 16;
  static constexpr int kWinBytes = N * 4;
  static constexpr int kLdsBytes = kTwBytes + kWinBytes + kFramesPerWg * kPlane * 4;
  static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
  __host__ __device__ static constexpr int cell(int i) { return i + (i >> kPadShift); }
};

// spectrogramWorker.js:233-252 / spectrogram.esm.js:181-191: t = w * A[M]; A[M] = A[k] - t; A[k] += t,
// every product and sum a double, every store a float32.
__device__ __forceinline__ void butterfly(float& rk, float& ik, float& rM, float& iM, double g, double s) {
  const double oM = rM, lM = iM, ok = rk, lk = ik;
  const double m = g * oM - s * lM;
  const double y = g * lM + s * oM;
  rM = (float)(ok - m);
  iM = (float)(lk - y);
  rk = (float)(ok + m);
  ik = (float)(lk + y);
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename In>
__device__ __forceinline__ float ld(const In* p, uint64_t i);
template <>
__device__ __forceinline__ float ld<float>(const float* p, uint64_t i) {
  return p[i];
}
template <>
__device__ __forceinline__ float ld<int16_t>(const int16_t* p, uint64_t i) {
  return (float)p[i] * (1.0f / 32768.0f);
}

// Queue of the fix-up pass (p.tile_scratch): a header of kQueueHeader words, then the list of marked frames.
constexpr int kQueueCount = 0;    // number of marked records found (may exceed the list's capacity)
constexpr int kQueueHeader = 4;
constexpr int kCompactThreads = 256, kCompactPerThread = 16;

// Lists the records that a float32 feature kernel marked (reserved[0] != 0): frame numbers within the launch, in
// no particular order.  One workgroup per 4096 records collects its marks in LDS and reserves its piece of the
// list with a single atomic (one atomic per marked record on one word would take longer than the scan).
__global__ __launch_bounds__(kCompactThreads) void feature_marks_compact_kernel(const sono_frame_feature* __restrict__ marks,
                                                                                uint64_t nframes, uint32_t* __restrict__ q,
                                                                                uint32_t capacity) {
  __shared__ uint32_t s_list[kCompactThreads * kCompactPerThread];
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * (kCompactThreads * kCompactPerThread);
  uint32_t word[kCompactPerThread];
#pragma unroll
  for (int j = 0; j < kCompactPerThread; ++j) {
    const uint64_t f = base + (uint64_t)j * kCompactThreads + threadIdx.x;
    word[j] = f < nframes ? reinterpret_cast<const uint32_t*>(marks + f)[1] : 0u;   // peak_idx, reserved[0..2]
  }
#pragma unroll
  for (int j = 0; j < kCompactPerThread; ++j)
    if (word[j] & 0xff00u) s_list[atomicAdd(&s_n, 1u)] = (uint32_t)(base + (uint64_t)j * kCompactThreads + threadIdx.x);
  __syncthreads();
  const uint32_t n = s_n;
  if (n == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(q + kQueueCount, n);
  __syncthreads();
  const uint32_t at = s_base;
  for (uint32_t i = threadIdx.x; i < n; i += kCompactThreads)
    if (at + i < capacity) q[kQueueHeader + at + i] = s_list[i];
}

// FIXUP: `p.out` is a SONO_OUT_FEATURES output of a float32 kernel; only the records it marked (reserved[0] != 0) are
// recomputed: the wavefronts take the list of marked frames (or, past its capacity, blocks of 64 records) as many frames
// at a time as they transform (two at 512 / 1024 points, one at 2048 / 4096).
template <int N, typename In, bool FIXUP>
__global__ __launch_bounds__(Geo<N>::kWaves * 64) void stft_exact_kernel(StftParams p) {
  using X = Geo<N>;
  constexpr int G = X::G, R = X::R, kWaves = X::kWaves;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* tw_lds = reinterpret_cast<double2*>(smem);                    // [R-1][G]: row t-1+j of stage T = G t
  float* win_lds = reinterpret_cast<float*>(smem + X::kTwBytes);          // window[j]
  const double2* __restrict__ tw = reinterpret_cast<const double2*>(p.twiddles);
  for (int i = threadIdx.x; i < X::kTwRows * G; i += kWaves * 64) {
    const int row = i / G, lane_p = i % G;
    const int t = 1 << (31 - __builtin_clz(row + 1)), j = row + 1 - t;
    tw_lds[i] = tw[G * t - 1 + G * j + lane_p];
  }
  for (int i = threadIdx.x; i < N; i += kWaves * 64) win_lds[i] = p.window[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, l = lane & (G - 1), h = lane / G;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* plane = reinterpret_cast<float*>(smem + X::kTwBytes + X::kWinBytes) + (wave * X::kFramesPerWave + h) * X::kPlane;
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  const int lb = (int)(__builtin_bitreverse32((uint32_t)l) >> (32 - X::kLogG));  // the lane's bits reversed
  const double two_over_n = 2.0 / (double)N;
  const bool worker = p.scale_mode == SONO_SCALE_WORKER_LOG5;
  const ExactPred pred = make_exact_pred(p.scale_mode, N, p.gain_db, p.range_db);

  // one frame per group of G lanes: frame `fi` of the launch; a spare group (live == false) stores nothing
  auto process = [&](const uint64_t fi, const bool live) {
    const uint64_t frame = p.frame0 + (live ? fi : p.nframes - 1);  // a spare group repeats the last frame and stores nothing
    const uint64_t a = frame * p.hop;

    // A[R l + r] = float32(x[j] * w[j]) with j = bitreverse(R l + r) = G * brev(r) + brev(l)
    float re[R], im[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int j = G * q + lb;
      re[brevc(q, X::kLogR)] = ld<In>(x, a + j) * win_lds[j];
      im[brevc(q, X::kLogR)] = 0.0f;
    }
    // stages T = 1 .. R/2
#pragma unroll
    for (int T = 1; T < R; T <<= 1) {
#pragma unroll
      for (int z = 0; z < T; ++z) {
        const double2 w = tw[T - 1 + z];  // wave-uniform
#pragma unroll
        for (int g = 0; g < R; g += 2 * T) butterfly(re[g + z], im[g + z], re[g + z + T], im[g + z + T], w.x, w.y);
      }
    }
    if constexpr (X::kCross) {
      // stage T = R: A[k] in the even lane, A[k + R] in the odd one, factor tw[R - 1 + r]
      const bool odd = (l & 1) != 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double2 w = tw[R - 1 + r];  // wave-uniform
        const float pr = dpp_f<kDppXor1>(re[r]), pi = dpp_f<kDppXor1>(im[r]);
        const double oM = odd ? re[r] : pr, lM = odd ? im[r] : pi;
        const double ok = odd ? pr : re[r], lk = odd ? pi : im[r];
        const double m = w.x * oM - w.y * lM;
        const double y = w.x * lM + w.y * oM;
        const float dr = (float)(ok - m), di = (float)(lk - y);
        const float sr = (float)(ok + m), si = (float)(lk + y);
        re[r] = odd ? dr : sr;
        im[r] = odd ? di : si;
      }
    }
    // transpose: lane l, register r (A[R l + r])  ->  lane l, register r (A[G r + l]); one plane after the other
#pragma unroll
    for (int r = 0; r < R; ++r) plane[(R * l + r) + ((R * l + r) >> X::kPadShift)] = re[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) re[r] = plane[(G * r + l) + ((G * r + l) >> X::kPadShift)];
    wave_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) plane[(R * l + r) + ((R * l + r) >> X::kPadShift)] = im[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < R; ++r) im[r] = plane[(G * r + l) + ((G * r + l) >> X::kPadShift)];
    wave_sync();
    // stages T = G t, t = 1 .. R/2
#pragma unroll
    for (int t = 1; t < R; t <<= 1) {
#pragma unroll
      for (int j = 0; j < t; ++j) {
        const double2 w = tw_lds[(t - 1 + j) * G + l];
#pragma unroll
        for (int g = 0; g < R; g += 2 * t) butterfly(re[g + j], im[g + j], re[g + j + t], im[g + j + t], w.x, w.y);
      }
    }

    // bins G r + l, r < R/2.  The exact quantisers are long where the predictor does not decide, so the bins are
    // parked in the transpose buffer and taken one at a time by a rolled loop instead of R/2 inlined copies.
    auto cell_re = [&](int r) { return (G * r + l) + ((G * r + l) >> X::kPadShift); };
    auto cell_im = [&](int r) { return (G * (R / 2 + r) + l) + ((G * (R / 2 + r) + l) >> X::kPadShift); };
#pragma unroll
    for (int r = 0; r < R / 2; ++r) {
      plane[cell_re(r)] = re[r];
      plane[cell_im(r)] = im[r];
    }
    wave_sync();
    const uint64_t orow = (p.col0 + fi) * p.out_bins;
    if (FIXUP || p.out_kind == SONO_OUT_FEATURES) {
      // first bin of the band whose float32 magnitude exceeds all earlier ones (strict >)
      float best_m = -1.0f, best_r = 0.0f, best_i = 0.0f;
      int best_k = (int)p.bin_lo;
#pragma unroll 1
      for (int r = 0; r < R / 2; ++r) {
        const int y = G * r + l;
        const float fr = plane[cell_re(r)], fi2 = plane[cell_im(r)];
        const double e = fr, s = fi2;
        const double mag = sqrt(e * e + s * s);
        const float m = worker ? (float)mag : (float)(two_over_n * mag);
        if (y >= (int)p.bin_lo && y < (int)p.bin_hi && m > best_m) {
          best_m = m; best_k = y; best_r = fr; best_i = fi2;
        }
      }
#pragma unroll
      for (int msk = 1; msk < G; msk <<= 1) {
        const float om = __shfl_xor(best_m, msk, 64), orr = __shfl_xor(best_r, msk, 64), oi = __shfl_xor(best_i, msk, 64);
        const int ok = __shfl_xor(best_k, msk, 64);
        if (om > best_m || (om == best_m && ok < best_k)) { best_m = om; best_k = ok; best_r = orr; best_i = oi; }
      }
      if (l == 0 && live) {
        sono_frame_feature ft;
        ft.peak_bin = best_k;
        ft.peak_idx = (uint8_t)(worker ? worker_index_exact(best_r, best_i)
                                       : plugin_index_exact(plugin_mag_exact(best_r, best_i, two_over_n), p.gain_db, p.range_db));
        ft.reserved[0] = ft.reserved[1] = ft.reserved[2] = 0;
        reinterpret_cast<sono_frame_feature*>(p.out)[p.col0 + fi] = ft;
      }
    } else if (live) {
#pragma unroll 4
      for (int r = 0; r < R / 2; ++r) {
        const int y = G * r + l;
        if (y < (int)p.bin_lo || y >= (int)p.bin_hi) continue;
        const float fr = plane[cell_re(r)], fi2 = plane[cell_im(r)];
        const uint64_t o = orow + (uint32_t)(y - (int)p.out_bin0);
        if (p.out_kind == SONO_OUT_U8_INDEX) {
          const uint32_t idx = worker ? worker_index_exact_p(fr, fi2, pred)
                                      : plugin_index_exact_p(fr, fi2, two_over_n, p.gain_db, p.range_db, pred);
          reinterpret_cast<uint8_t*>(p.out)[o] = (uint8_t)idx;
        } else {
          const double e = fr, s = fi2;
          const double mag = sqrt(e * e + s * s);
          const float m = worker ? (float)mag : (float)(two_over_n * mag);
          float v = m;
          if (p.out_kind == SONO_OUT_F32_DB) {
            const double sv = ((double)m > 1e-12) ? (double)m : 1e-12;
            v = (float)(20 * log10(sv));
          }
          reinterpret_cast<float*>(p.out)[o] = v;
        }
      }
    }
    wave_sync();  // the buffer is rewritten by the next frame's transpose
  };

  if constexpr (FIXUP) {
    // Marked records are rare, come in runs (the frames around the start of a call) and cost tens of microseconds each
    // in this arithmetic; feature_marks_compact_kernel has listed them, and the wavefronts take the list two
    // entries at a time, interleaved, so that a run of neighbouring frames is dealt out to as many wavefronts.
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p.tile_scratch);
    const uint32_t listed = q[kQueueCount] < p.fixup ? q[kQueueCount] : p.fixup;   // p.fixup: capacity of the list
    const uint32_t* list = q + kQueueHeader;
    constexpr int kPer = X::kFramesPerWave;  // list entries a wavefront takes at a time (its group h takes entry e + h)
    for (uint64_t e = ((uint64_t)blockIdx.x * kWaves + wave) * kPer; e < listed; e += (uint64_t)gridDim.x * kWaves * kPer) {
      const bool mine = e + h < listed;
      process(list[mine ? e + h : e], mine);
    }
    if (q[kQueueCount] > p.fixup) {
      // more marked records than the list holds: the rest are found by scanning the records themselves, one block of
      // 64 records per wavefront at a time.  Nothing orders this scan behind the list pass of the OTHER workgroups, so it
      // may still see the mark on a listed record that one of them is recomputing and recompute it as well: the same
      // arithmetic on the same samples, i.e. the same 8 bytes stored twice -- duplicated work in the overflow case
      // only, never a different result
      const sono_frame_feature* marks = reinterpret_cast<const sono_frame_feature*>(p.out) + p.col0;
      for (uint64_t blk = (uint64_t)blockIdx.x * kWaves + wave; blk * 64 < p.nframes; blk += (uint64_t)gridDim.x * kWaves) {
        const uint64_t fl = blk * 64 + lane;
        // (sc1: another workgroup of this launch may have cleaned the record; its stores are not in this CU's L1)
        const uint32_t word = fl < p.nframes ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(marks + fl) + 1) : 0u;
        uint64_t todo = __ballot((word & 0xff00u) != 0);
        while (todo) {  // wave-uniform
          const int i0 = __builtin_ctzll(todo);
          todo &= todo - 1;
          if constexpr (X::kFramesPerWave == 2) {
            const bool two = todo != 0;
            const int i1 = two ? __builtin_ctzll(todo) : i0;
            if (two) todo &= todo - 1;
            process(blk * 64 + (h ? i1 : i0), h == 0 || two);
          } else {
            process(blk * 64 + i0, true);
          }
        }
      }
    }
  } else {
    for (uint64_t base = (uint64_t)blockIdx.x * X::kFramesPerWg; base < p.nframes; base += (uint64_t)gridDim.x * X::kFramesPerWg) {
      const uint64_t fi = base + wave * X::kFramesPerWave + h;
      process(fi, fi < p.nframes);
    }
  }
}

}  // namespace exactreg

bool exact_available(const StftParams& p) {
  return (p.n == 512 || p.n == 1024 || p.n == 2048 || p.n == 4096) && p.twiddles && !p.fb_bin && p.nframes >= 16 &&
         (p.out_kind == SONO_OUT_U8_INDEX || p.out_kind == SONO_OUT_F32_DB || p.out_kind == SONO_OUT_F32_MAG ||
          p.out_kind == SONO_OUT_FEATURES);
}

namespace exactreg {
template <int N>
static int launch_n(const StftParams& p, hipStream_t stream) {
  using X = Geo<N>;
  const int cus = device_cu_count();
  const uint64_t need = (p.nframes + X::kFramesPerWg - 1) / X::kFramesPerWg;
  const uint64_t cap = (uint64_t)cus * (N == 512 ? 2 : 1);   // (512 points: two workgroups fit a CU)
  const uint32_t grid = (uint32_t)(need < cap ? need : cap);
  if (p.in_kind == SONO_IN_I16) {
    auto kern = stft_exact_kernel<N, int16_t, false>;
    SONO_GRANT_LDS(kern, X::kLdsBytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(X::kWaves * 64), X::kLdsBytes, stream, p);
  } else {
    auto kern = stft_exact_kernel<N, float, false>;
    SONO_GRANT_LDS(kern, X::kLdsBytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(X::kWaves * 64), X::kLdsBytes, stream, p);
  }
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}
}  // namespace exactreg

// Returns 1 if this launch is not served here (the caller falls back to the general kernel).
int launch_stft_exact(const StftParams& p, void* stream_v) {
  using namespace exactreg;
  if (!exact_available(p)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  switch (p.n) {
    case 512: note_kernel("stft512_exact_kernel"); return launch_n<512>(p, stream);
    case 1024: note_kernel("stft1024_exact_kernel"); return launch_n<1024>(p, stream);
    case 2048: note_kernel("stft2048_exact_kernel"); return launch_n<2048>(p, stream);
    default: note_kernel("stft4096_exact_kernel"); return launch_n<4096>(p, stream);
  }
}

// p.tile_scratch: the queue (kQueueHeader words + p.fixup list entries, p.fixup >= 1).
bool features_fixup_available(const StftParams& p) {
  return (p.n == 512 || p.n == 1024 || p.n == 2048 || p.n == 4096) && p.twiddles && !p.fb_bin && p.out_kind == SONO_OUT_FEATURES && p.nframes > 0 && p.tile_scratch &&
         p.fixup > 0 && p.nframes < ((uint64_t)1 << 32);   // frame numbers are 32-bit list entries
}

uint64_t features_fixup_queue_bytes(uint32_t capacity) { return ((uint64_t)exactreg::kQueueHeader + capacity) * 4; }

namespace exactreg {
template <int N>
static int launch_fixup_n(const StftParams& p, hipStream_t stream) {
  using X = Geo<N>;
  // the list's length is only known on the device: a full grid, whose wavefronts find their entries or none
  const uint64_t need = (p.nframes + X::kFramesPerWg - 1) / X::kFramesPerWg;
  const uint64_t cap = (uint64_t)device_cu_count() * (N == 512 ? 2 : 1);
  const uint32_t grid = (uint32_t)(need < cap ? need : cap);
  if (p.in_kind == SONO_IN_I16) {
    auto kern = stft_exact_kernel<N, int16_t, true>;
    SONO_GRANT_LDS(kern, X::kLdsBytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(X::kWaves * 64), X::kLdsBytes, stream, p);
  } else {
    auto kern = stft_exact_kernel<N, float, true>;
    SONO_GRANT_LDS(kern, X::kLdsBytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(X::kWaves * 64), X::kLdsBytes, stream, p);
  }
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}
}  // namespace exactreg

// `listed`: the float32 kernel has put the frames it marked into the queue itself (StftParams::feat_queue).
int launch_features_fixup(const StftParams& p, void* stream_v, bool listed) {
  using namespace exactreg;
  if (!features_fixup_available(p)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  if (!listed) {
    if (hipMemsetAsync(p.tile_scratch, 0, kQueueHeader * 4, stream) != hipSuccess) return SONO_ERR_DEVICE;
    const uint32_t per = kCompactThreads * kCompactPerThread;
    hipLaunchKernelGGL(feature_marks_compact_kernel, dim3((uint32_t)((p.nframes + per - 1) / per)), dim3(kCompactThreads), 0, stream,
                       reinterpret_cast<const sono_frame_feature*>(p.out) + p.col0, p.nframes,
                       reinterpret_cast<uint32_t*>(p.tile_scratch), p.fixup);
  }
  switch (p.n) {
    case 512: note_kernel("stft512_exact_kernel<fixup>"); return launch_fixup_n<512>(p, stream);
    case 1024: note_kernel("stft1024_exact_kernel<fixup>"); return launch_fixup_n<1024>(p, stream);
    case 2048: note_kernel("stft2048_exact_kernel<fixup>"); return launch_fixup_n<2048>(p, stream);
    default: note_kernel("stft4096_exact_kernel<fixup>"); return launch_fixup_n<4096>(p, stream);
  }
}

}  // namespace sono
