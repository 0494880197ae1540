// Exact-arithmetic STFT for 1024-point frames, in registers.
//
// Same operation sequence as the general kernel's double flavour -- the radix-2 decimation-in-time
// network of spectrogramWorker.js:206-253 / spectrogram.esm.js:166-193 with per-stage twiddle tables,
// double butterflies over float32 storage, no FMA contraction -- so the results are bit-identical to it
// and to the JavaScript.  Only the schedule differs: a butterfly's inputs and factor are fixed by the
// network, not by who computes it, so the ten stages are run as two in-register passes of five stages
// with one transpose through LDS in between, instead of ten LDS round trips with workgroup barriers.
//
// One half-wavefront (32 lanes) per frame.  The working array A[0..1023] (bit-reversed input order):
//   pass 1, stages T = 1..16 : lane l holds A[32l + r] in register r; the factor of a butterfly depends
//                               on r only (tw[T-1 + (r mod T)]), wave-uniform, read as scalars;
//   transpose                 : lane l holds A[32r + l] in register r;
//   pass 2, stages T = 32..512: partner registers are T/32 apart; the factor tw[T-1 + 32*(r mod T/32) + l]
//                               differs per lane and is read from a [31][32] double2 copy in LDS.
// A[k] is then X[k] in natural order: lane l, register r < 16 holds bin 32r + l.
//
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "sono_device.h"
#include "sono_internal.h"
#include "sono_launch.h"

namespace sono {
namespace exact1024 {

constexpr int kWaves = 8;           // wavefronts per workgroup: 16 frames per iteration
constexpr int kPlane = 32 * 33;     // floats of one transpose buffer (row stride 33: conflict-free both ways)
constexpr int kTwBytes = 31 * 32 * 16;
constexpr int kWinBytes = 1024 * 4;
constexpr int kLdsBytes = kTwBytes + kWinBytes + kWaves * 2 * kPlane * 4;

__host__ __device__ constexpr int brev5(int v) {
  return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

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

// FIXUP: `p.out` is a SONO_OUT_FEATURES output of a float32 kernel; only the records it marked (reserved[0] != 0)
// are recomputed, each wavefront scanning blocks of 64 records and taking the marked frames two at a time.
template <typename In, bool FIXUP>
__global__ __launch_bounds__(kWaves * 64) void stft1024_exact_kernel(StftParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* tw_lds = reinterpret_cast<double2*>(smem);                 // [31][32]: row t-1+j of stage T = 32t
  float* win_lds = reinterpret_cast<float*>(smem + kTwBytes);          // window[j]
  const double2* __restrict__ tw = reinterpret_cast<const double2*>(p.twiddles);
  for (int i = threadIdx.x; i < 31 * 32; i += kWaves * 64) {
    const int row = i >> 5, lane_p = i & 31;
    const int t = 1 << (31 - __builtin_clz(row + 1)), j = row + 1 - t;
    tw_lds[i] = tw[32 * t - 1 + 32 * j + lane_p];
  }
  for (int i = threadIdx.x; i < 1024; i += kWaves * 64) win_lds[i] = p.window[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* plane = reinterpret_cast<float*>(smem + kTwBytes + kWinBytes) + (wave * 2 + h) * kPlane;
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  const int lb = brev5(l);
  const double two_over_n = 2.0 / 1024.0;
  const bool worker = p.scale_mode == SONO_SCALE_WORKER_LOG5;

  // one frame per half-wavefront: frame `fi` of the launch; a spare half (live == false) stores nothing
  auto process = [&](const uint64_t fi, const bool live) {
    const uint64_t frame = p.frame0 + (live ? fi : p.nframes - 1);  // a spare half repeats the last frame and stores nothing
    const uint64_t a = frame * p.hop;

    // A[32l + r] = float32(x[j] * w[j]) with j = bitreverse10(32l + r) = 32*brev5(r) + brev5(l)
    float re[32], im[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int j = 32 * q + lb;
      re[brev5(q)] = ld<In>(x, a + j) * win_lds[j];
      im[brev5(q)] = 0.0f;
    }
    // stages T = 1, 2, 4, 8, 16
#pragma unroll
    for (int T = 1; T < 32; T <<= 1) {
#pragma unroll
      for (int z = 0; z < T; ++z) {
        const double2 w = tw[T - 1 + z];  // wave-uniform
#pragma unroll
        for (int g = 0; g < 32; g += 2 * T) butterfly(re[g + z], im[g + z], re[g + z + T], im[g + z + T], w.x, w.y);
      }
    }
    // transpose: lane l, register r  ->  lane r, register l (one plane after the other through one buffer)
#pragma unroll
    for (int r = 0; r < 32; ++r) plane[l * 33 + r] = re[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 32; ++r) re[r] = plane[r * 33 + l];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 32; ++r) plane[l * 33 + r] = im[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 32; ++r) im[r] = plane[r * 33 + l];
    wave_sync();
    // stages T = 32t, t = 1, 2, 4, 8, 16
#pragma unroll
    for (int t = 1; t < 32; t <<= 1) {
#pragma unroll
      for (int j = 0; j < t; ++j) {
        const double2 w = tw_lds[(t - 1 + j) * 32 + l];
#pragma unroll
        for (int g = 0; g < 32; g += 2 * t) butterfly(re[g + j], im[g + j], re[g + j + t], im[g + j + t], w.x, w.y);
      }
    }

    // bins 32r + l, r < 16.  The quantisers are long (double sqrt and log10), so the bins are parked in
    // the transpose buffer and taken one at a time by a rolled loop instead of 16 inlined copies.
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      plane[r * 33 + l] = re[r];
      plane[(16 + r) * 33 + l] = im[r];
    }
    wave_sync();
    const uint64_t orow = (p.col0 + fi) * p.out_bins;
    if (FIXUP || p.out_kind == SONO_OUT_FEATURES) {
      // first bin of the band whose float32 magnitude exceeds all earlier ones (strict >)
      float best_m = -1.0f, best_r = 0.0f, best_i = 0.0f;
      int best_k = (int)p.bin_lo;
#pragma unroll 1
      for (int r = 0; r < 16; ++r) {
        const int y = 32 * r + l;
        const float fr = plane[r * 33 + l], fi2 = plane[(16 + r) * 33 + l];
        const double e = fr, s = fi2;
        const double mag = sqrt(e * e + s * s);
        const float m = worker ? (float)mag : (float)(two_over_n * mag);
        if (y >= (int)p.bin_lo && y < (int)p.bin_hi && m > best_m) {
          best_m = m; best_k = y; best_r = fr; best_i = fi2;
        }
      }
#pragma unroll
      for (int msk = 1; msk < 32; msk <<= 1) {
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
      for (int r = 0; r < 16; ++r) {
        const int y = 32 * r + l;
        if (y < (int)p.bin_lo || y >= (int)p.bin_hi) continue;
        const float fr = plane[r * 33 + l], fi2 = plane[(16 + r) * 33 + l];
        const uint64_t o = orow + (uint32_t)(y - (int)p.out_bin0);
        if (p.out_kind == SONO_OUT_U8_INDEX) {
          const uint32_t idx = worker ? worker_index_exact(fr, fi2)
                                      : plugin_index_exact(plugin_mag_exact(fr, fi2, two_over_n), p.gain_db, p.range_db);
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
    for (uint64_t e = ((uint64_t)blockIdx.x * kWaves + wave) * 2; e < listed; e += (uint64_t)gridDim.x * kWaves * 2) {
      const bool two = e + 1 < listed;
      process(list[e + ((h && two) ? 1 : 0)], h == 0 || two);
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
          const bool two = todo != 0;
          const int i1 = two ? __builtin_ctzll(todo) : i0;
          if (two) todo &= todo - 1;
          process(blk * 64 + (h ? i1 : i0), h == 0 || two);
        }
      }
    }
  } else {
    for (uint64_t base = (uint64_t)blockIdx.x * (kWaves * 2); base < p.nframes; base += (uint64_t)gridDim.x * (kWaves * 2)) {
      const uint64_t fi = base + wave * 2 + h;
      process(fi, fi < p.nframes);
    }
  }
}

}  // namespace exact1024

bool exact1024_available(const StftParams& p) {
  return p.n == 1024 && p.twiddles && !p.fb_bin && p.nframes >= 16 &&
         (p.out_kind == SONO_OUT_U8_INDEX || p.out_kind == SONO_OUT_F32_DB || p.out_kind == SONO_OUT_F32_MAG ||
          p.out_kind == SONO_OUT_FEATURES);
}

// Returns 1 if this launch is not served here (the caller falls back to the general kernel).
int launch_stft_exact(const StftParams& p, void* stream_v) {
  using namespace exact1024;
  if (!exact1024_available(p)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  const int cus = device_cu_count();
  const uint64_t need = (p.nframes + kWaves * 2 - 1) / (kWaves * 2);
  const uint32_t grid = (uint32_t)(need < (uint64_t)cus ? need : (uint64_t)cus);
#define SONO_LAUNCH_X(In)                                                                                        \
  do {                                                                                                           \
    auto kern = stft1024_exact_kernel<In, false>;                                                                \
    SONO_GRANT_LDS(kern, kLdsBytes);                                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), kLdsBytes, stream, p);                               \
  } while (0)
  if (p.in_kind == SONO_IN_I16) SONO_LAUNCH_X(int16_t); else SONO_LAUNCH_X(float);
#undef SONO_LAUNCH_X
  note_kernel("stft1024_exact_kernel");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

// p.tile_scratch: the queue (kQueueHeader words + p.fixup list entries, p.fixup >= 1).
bool features_fixup_available(const StftParams& p) {
  return p.n == 1024 && p.twiddles && !p.fb_bin && p.out_kind == SONO_OUT_FEATURES && p.nframes > 0 && p.tile_scratch &&
         p.fixup > 0 && p.nframes < ((uint64_t)1 << 32);   // frame numbers are 32-bit list entries
}

uint64_t features_fixup_queue_bytes(uint32_t capacity) { return ((uint64_t)exact1024::kQueueHeader + capacity) * 4; }

// `listed`: the float32 kernel has put the frames it marked into the queue itself (StftParams::feat_queue).
int launch_features_fixup(const StftParams& p, void* stream_v, bool listed) {
  using namespace exact1024;
  if (!features_fixup_available(p)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  if (!listed) {
    if (hipMemsetAsync(p.tile_scratch, 0, kQueueHeader * 4, stream) != hipSuccess) return SONO_ERR_DEVICE;
    const uint32_t per = kCompactThreads * kCompactPerThread;
    hipLaunchKernelGGL(feature_marks_compact_kernel, dim3((uint32_t)((p.nframes + per - 1) / per)), dim3(kCompactThreads), 0, stream,
                       reinterpret_cast<const sono_frame_feature*>(p.out) + p.col0, p.nframes,
                       reinterpret_cast<uint32_t*>(p.tile_scratch), p.fixup);
  }
  // the list's length is only known on the device: a full grid, whose wavefronts find their entries or none
  const uint64_t need = (p.nframes + kWaves * 2 - 1) / (kWaves * 2);
  const uint64_t cap = (uint64_t)device_cu_count();   // 192 registers: one workgroup (two wavefronts per SIMD) per CU
  const uint32_t grid = (uint32_t)(need < cap ? need : cap);
#define SONO_LAUNCH_F(In)                                                                                        \
  do {                                                                                                           \
    auto kern = stft1024_exact_kernel<In, true>;                                                                 \
    SONO_GRANT_LDS(kern, kLdsBytes);                                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), kLdsBytes, stream, p);                               \
  } while (0)
  if (p.in_kind == SONO_IN_I16) SONO_LAUNCH_F(int16_t); else SONO_LAUNCH_F(float);
#undef SONO_LAUNCH_F
  note_kernel("stft1024_exact_kernel<fixup>");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
