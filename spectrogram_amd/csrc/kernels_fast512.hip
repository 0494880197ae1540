// Float32 STFT kernel for 512-point frames, the page's default FFT size (main.js:886-889): the worker path with exact
// roots of unity (spectrogramWorker.js:38-99,206-253) and the plugin path with its twiddle table (spectrogram.esm.js:166-200).  Compile with -ffp-contract=fast.
//
// Same machine mapping as the 1024-point kernel, with two transforms per half-wavefront:
//   * a half-wavefront (32 lanes) transforms FOUR real frames as two complex 512-point frames A and B
//     (z = x0 + i*x1); lane l holds samples 32a + l, a = 0..15, of A in registers 0..15 and of B in 16..31;
//   * 512 = 16 x 32: the first pass is a 16-point transform in each register half (the first four stages of the
//     32-point radix-2 network), then the twiddle W_512^(l*k1), then the same 32 x 32 transpose through LDS as the
//     1024-point kernel: lane 16g + k1 now holds Y_l[k1] of transform g for l = 0..31, and the second pass is one
//     32-point transform per lane: register k2 = Z_g[k1 + 16*k2];
//   * epilogue in two halves of 128 bins, both transforms at once: registers 8P..8P+7 are Z[k], 128P <= k < 128P+128,
//     registers 24-8P..31-8P their partners Z[512 - k]; both go through LDS in ascending k so that a lane reads
//     four consecutive bins of both as 16-byte pieces, separates the two frames (X0 = (Z[k] + conj Z[N-k])/2,
//     X1 = (Z[k] - conj Z[N-k])/(2i)) and fuses |X|^2 -> log2 -> quantiser -> packed uint8.
//   * HOP32 > 0: hop == 32*HOP32, so the four frames of a half-wavefront are one run of 16 + 3*HOP32 loads.
// Outputs: [frame][bin] colour indices, or per-frame spectral peaks.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cmath>
#include <vector>

#include "sono_device.h"
#include "sono_internal.h"
#include "sono_launch.h"
#include "sono_loud.h"
#include "sono_fft32.h"
#include "sono_image_ring.h"

namespace sono {
namespace fast512 {

constexpr int kWaves = 16;                          // wavefronts per workgroup (one workgroup per CU): 128 frames per iteration
#ifndef SONO_SYNC_ITERS_512
#define SONO_SYNC_ITERS_512 4
#endif
constexpr int kSyncIters = SONO_SYNC_ITERS_512;     // index kernel: iterations between two barriers of its wavefronts
constexpr int kPlaneStride = 36;                    // floats per transpose row (rows 16-B aligned, b128 reads conflict-free)
constexpr int kFftBufBytes = 32 * kPlaneStride * 4; // 4608 per half-wavefront
// epilogue half P, float2 cells: Z_g[128P + j] at g*128 + j (j < 128), Z_g[512 - 128P - j] at 256 + g*130 + j (j <= 128)
constexpr int kMirrorBase = 256, kMirrorStride = 130;
static_assert((kMirrorBase + 2 * kMirrorStride) * 8 <= kFftBufBytes, "epilogue must fit the transpose buffer");
constexpr int kTwBytes = 16 * 32 * 8;               // W_512^(l*k1) as [k1][l] float2
constexpr int kWinStride = 20;                      // window as [l][kWinStride]: lane l reads w[32a + l], a = 0..15
constexpr int kWinBytes = 32 * kWinStride * 4;
constexpr int kLdsBytes = kTwBytes + kWinBytes + kWaves * 2 * kFftBufBytes;

__device__ constexpr float kC32[16] = {1.0f,          0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                       0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
                                       0.0f,          -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
                                       -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
__device__ constexpr float kS32[16] = {0.0f,          0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                                       0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f,
                                       1.0f,          0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                       0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};

__host__ __device__ constexpr int brev5(int v) {
  return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}
__host__ __device__ constexpr int brev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }

// The first STAGES stages of the in-register radix-2 decimation-in-time network on 32 values (input in bit-reversed
// order within each group of 2^STAGES registers): STAGES = 5 is a 32-point transform, STAGES = 4 two independent
// 16-point transforms on registers 0..15 and 16..31.
template <int STAGES>
__device__ __forceinline__ void dit32(float (&re)[32], float (&im)[32]) {
#pragma unroll
  for (int s = 0; s < STAGES; ++s) {
    const int half = 1 << s;
#pragma unroll
    for (int g = 0; g < 32; g += 2 * half) {
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const int t = j * (16 >> s);  // twiddle W_32^t = (C, -S)
        const int i0 = g + j, i1 = i0 + half;
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        if (t == 0) {
          re[i0] = ar + br; im[i0] = ai + bi;
          re[i1] = ar - br; im[i1] = ai - bi;
        } else if (t == 8) {  // w = -i
          re[i0] = ar + bi; im[i0] = ai - br;
          re[i1] = ar - bi; im[i1] = ai + br;
        } else {
          const float C = kC32[t], S = kS32[t];
          const float o1r = __builtin_fmaf(S, bi, __builtin_fmaf(C, br, ar));
          const float o1i = __builtin_fmaf(-S, br, __builtin_fmaf(C, bi, ai));
          re[i0] = o1r; im[i0] = o1i;
          re[i1] = __builtin_fmaf(2.0f, ar, -o1r);
          im[i1] = __builtin_fmaf(2.0f, ai, -o1i);
        }
      }
    }
  }
}

// The same network with factors from a table (the plugin's recurrence factors or a host table, which are not exact
// roots of unity: every butterfly must use the very factor the reference uses, spectrogram.esm.js:176-190).
// tw(half, j): factor (cos, sin) of the butterfly at position j of a stage with half size `half`.  FIRST: these are
// the first stages of the transform, where j == 0 is the factor (1, 0) exactly in every table the kernel accepts.
template <int STAGES, bool FIRST, class TwFn>
__device__ __forceinline__ void dit32_table(float (&re)[32], float (&im)[32], TwFn tw) {
#pragma unroll
  for (int s = 0; s < STAGES; ++s) {
    const int half = 1 << s;
#pragma unroll
    for (int j = 0; j < half; ++j) {
      float2 w = make_float2(1.0f, 0.0f);
      if (!FIRST || j != 0) w = tw(half, j);
#pragma unroll
      for (int g = 0; g < 32; g += 2 * half) {
        const int i0 = g + j, i1 = i0 + half;
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        if (FIRST && j == 0) {
          re[i0] = ar + br; im[i0] = ai + bi;
          re[i1] = ar - br; im[i1] = ai - bi;
        } else {  // o1 = a + w*b, o2 = a - w*b = 2a - o1
          const float o1r = __builtin_fmaf(-w.y, bi, __builtin_fmaf(w.x, br, ar));
          const float o1i = __builtin_fmaf(w.y, br, __builtin_fmaf(w.x, bi, ai));
          re[i0] = o1r; im[i0] = o1i;
          re[i1] = __builtin_fmaf(2.0f, ar, -o1r);
          im[i1] = __builtin_fmaf(2.0f, ai, -o1i);
        }
      }
    }
  }
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename In>
__device__ __forceinline__ float ld(const In* p, int i);
template <>
__device__ __forceinline__ float ld<float>(const float* p, int i) {
  return p[i];
}
template <>
__device__ __forceinline__ float ld<int16_t>(const int16_t* p, int i) {
  return (float)p[i] * (1.0f / 32768.0f);
}

template <bool PLUGIN, int BYTE>
__device__ __forceinline__ uint32_t put_index(float msq, const FastScale& fs, uint32_t acc) {
  if constexpr (PLUGIN) {
    return fast_plugin_byte<BYTE>(msq, fs, acc);
  } else {
    return fast_worker_byte<BYTE>(msq, fs, acc);
  }
}

// Windowed samples (frames A0, A1 as re/im of registers 0..15, B0, B1 of 16..31; lane l = samples 32a + l) to the
// two spectra: register k2 = Z_g[k1 + 16*k2] in lane 16g + k1.
// BLK: the transpose goes through the lane's 16 x 16 block (`plane`; sono_fft32.h) instead of a 32 x 32 plane.
template <bool TABLE, bool BLK = false>
__device__ __forceinline__ void transform512(float (&re)[32], float (&im)[32], const float2* tw_lds,
                                             const float2* __restrict__ tw_flat, float* plane, int l) {
  if constexpr (TABLE) {
    // stages T = 1..8: wave-uniform factors by scalar loads (the table is not written while the kernel runs; the opaque
    // copy keeps the scalars from being held in registers round the frame loop)
    typedef const float __attribute__((address_space(4))) * const_table;
    const_table twc = (const_table)(uintptr_t)tw_flat;
    asm volatile("" : "+s"(twc));
    dit32_table<4, true>(re, im, [&](int half, int j) { return make_float2(twc[2 * (half - 1 + j)], twc[2 * (half - 1 + j) + 1]); });
  } else {
    dit32<4>(re, im);  // Y_l[k1] of A in registers k1, of B in 16 + k1
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if ((r & 15) == 0) continue;
      const float2 t = tw_lds[(r & 15) * 32 + l];
      const float ur = __builtin_fmaf(re[r], t.x, -im[r] * t.y);
      const float ui = __builtin_fmaf(re[r], t.y, im[r] * t.x);
      re[r] = ur;
      im[r] = ui;
    }
  }
  // transpose lane l, register r -> lane r, register l; the planes go through one buffer one after the other
  if constexpr (BLK) {
    fast1024::transpose32_plane(re, plane, l & 15);
    fast1024::transpose32_plane(im, plane, l & 15);
  } else {
#pragma unroll
  for (int r = 0; r < 32; ++r) plane[r * kPlaneStride + l] = re[r];
  wave_sync();
  {
    const float4* row = reinterpret_cast<const float4*>(plane + l * kPlaneStride);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = row[j];
      re[brev5(4 * j)] = v.x; re[brev5(4 * j + 1)] = v.y; re[brev5(4 * j + 2)] = v.z; re[brev5(4 * j + 3)] = v.w;
    }
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < 32; ++r) plane[r * kPlaneStride + l] = im[r];
  wave_sync();
  {
    const float4* row = reinterpret_cast<const float4*>(plane + l * kPlaneStride);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = row[j];
      im[brev5(4 * j)] = v.x; im[brev5(4 * j + 1)] = v.y; im[brev5(4 * j + 2)] = v.z; im[brev5(4 * j + 3)] = v.w;
    }
  }
  wave_sync();
  }
  if constexpr (TABLE) {
    int kt = l & 15;
    asm volatile("" : "+v"(kt));  // (keeps the 31 factor addresses out of registers across the loop)
    dit32_table<5, false>(re, im, [&](int half, int j) { return tw_lds[(half - 1 + j) * 16 + kt]; });  // stages T = 16..256
  } else {
    dit32<5>(re, im);  // register k2 = Z_g[k1 + 16*k2] in lane 16g + k1
  }
}

// Half P of the epilogue: lane 16g + k1 lays its Z_g[k], 128P <= k < 128P + 128 (registers 8P..8P+7), out at cell
// g*128 + (k - 128P) and the partners Z_g[512 - j], j = 128P..128P+128, at kMirrorBase + g*kMirrorStride + (j - 128P):
// they sit in registers 24-8P..31-8P (cell 128 - k1 - 16r for register 24-8P+r); the ends of that range are one
// register off: j = 0 is Z[0] (register 0 of lane k1 = 0) and j = 128 of half 1 is Z[384] (register 24 of lane k1 = 0).
// (P folds to a constant where the caller's loop over P is unrolled)
__device__ __forceinline__ void epilogue_cells(int P, const float (&re)[32], const float (&im)[32], float2* cell, int le) {
  const int ge = le >> 4, ke = le & 15;
  float2* lz = cell + ge * 128;
  float2* mz = cell + kMirrorBase + ge * kMirrorStride;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    lz[ke + 16 * r] = make_float2(re[8 * P + r], im[8 * P + r]);
    mz[128 - ke - 16 * r] = make_float2(re[24 - 8 * P + r], im[24 - 8 * P + r]);
  }
  if (ke == 0) mz[0] = P ? make_float2(re[24], im[24]) : make_float2(re[0], im[0]);
}

// TABLE: the context's per-stage twiddle table (flat layout, the stage with half size T at [T-1, 2T-1)) instead of
// exact roots.  Lane l's registers are the 16 consecutive bit-reversed positions 16*brev5(l) + brev4(a), so the
// first pass is the reference's stages T = 1..8 (factors wave-uniform) and, after the transpose, the second pass its
// stages T = 16..256 with the factor 16*(g mod t) + k1 of stage T = 16t, read from a [31][16] copy in LDS; there is
// no multiplication between the passes.
template <bool PLUGIN, typename In, bool FEAT, int HOP32, bool TABLE>
__global__ __launch_bounds__(kWaves * 64) void stft512_kernel(StftParams p, FastScale fs, const float2* __restrict__ tw_table) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw_lds = reinterpret_cast<float2*>(smem);
  float* win_lds = reinterpret_cast<float*>(smem + kTwBytes);
  const float2* __restrict__ tw_flat = reinterpret_cast<const float2*>(p.twiddles);
  for (int i = threadIdx.x; i < 512; i += kWaves * 64) {
    if constexpr (TABLE) {
      const int r = i >> 4, kk = i & 15;  // row r = t - 1 + j of the second pass: stage T = 16t, factor 16j + kk
      if (r < 31) {
        const int t = 1 << (31 - __builtin_clz(r + 1)), j = r + 1 - t;
        tw_lds[i] = tw_flat[16 * t - 1 + 16 * j + kk];
      }  // (the 15 factors of the first pass are read by scalar loads)
    } else {
      tw_lds[i] = tw_table[i];
    }
    win_lds[(i & 31) * kWinStride + (i >> 5)] = p.window[i];  // sample n = 32a + l at [l][a]
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* fbuf = smem + kTwBytes + kWinBytes + (wave * 2 + h) * kFftBufBytes;
  float* plane = reinterpret_cast<float*>(fbuf);
  float2* cell = reinterpret_cast<float2*>(fbuf);  // epilogue view
  const float4* win_row = reinterpret_cast<const float4*>(win_lds + l * kWinStride);
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(p.out);
  // blocks with equal blockIdx % 8 share an XCD (and its L2): give them neighbouring frame ranges
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const uint64_t nunits = (p.nframes + 7) >> 3;  // a wavefront takes 8 consecutive frames, a half-wavefront 4 of them
  // a workgroup takes a contiguous range of frames (`iters` iterations of its wavefronts): the samples neighbouring
  // blocks share are re-read from its own cache lines, not fetched again by another CU
  const uint64_t iters = ((nunits + kWaves - 1) / kWaves + gridDim.x - 1) / gridDim.x;
  const uint64_t unit_end = ((uint64_t)vb + 1) * iters * kWaves < nunits ? ((uint64_t)vb + 1) * iters * kWaves : nunits;

  // REDO: under the context's table a unit too loud for the pair separation (StftParams::loud_thr_raw) is transformed one
  // frame per transform: the loop body runs twice more, mode 1 = the frames fb, fb + 2 alone, mode 2 = fb + 1, fb + 3 alone
  // (through the same code as the pairs).  Wave-uniform: the eight frames of a wavefront go together.
  constexpr bool REDO = TABLE && !FEAT;
  int mode = 0;
  // every wavefront goes through all `iters` iterations, with or without frames of its own, and they meet at a barrier
  // every kSyncIters of them: the samples neighbouring units share are then still in the caches when the next wavefront
  // asks for them (left alone the wavefronts drift apart and fetch them from memory again)
  for (uint64_t itn = 0; itn < iters;) {
    const uint64_t unit = ((uint64_t)vb * iters + itn) * kWaves + wave;
    if (mode == 0 && itn != 0 && itn % kSyncIters == 0) __syncthreads();
    if (unit >= unit_end) {
      ++itn;
      continue;
    }
    const uint64_t f0 = unit * 8 + 4 * h;  // this half-wavefront's frames are f0 .. f0+3
    // The four frames transformed are fb .. fb+3: A = (fb, fb+1), B = (fb+2, fb+3).  At the end of the recording the
    // block is shifted back so that every load stays in bounds; frames before f0 (another half-wavefront's) and
    // past the end are not stored.
    const uint64_t fb = f0 + 4 <= p.nframes ? f0 : p.nframes - 4;
    float re[32], im[32];
    {
      float win[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = win_row[j];
        win[4 * j] = v.x; win[4 * j + 1] = v.y; win[4 * j + 2] = v.z; win[4 * j + 3] = v.w;
      }
      // mode 2: the block starts one frame later and ends with frame fb + 3 (its samples come from the cache; nothing
      // behind them is read, the frame after it may lie beyond the signal)
      const bool second = REDO && mode == 2;
      const In* x0 = x + (p.frame0 + fb + (second ? 1 : 0)) * p.hop + l;
      if constexpr (REDO) asm volatile("" : "+v"(x0));  // (opaque: or samples already loaded are held in registers instead)
      float loudest = 0.0f;  // TABLE: this lane's largest |sample|
      if constexpr (HOP32 > 0) {
        float raw[16 + 3 * HOP32];
#pragma unroll
        for (int a = 0; a < 16 + 2 * HOP32; ++a) raw[a] = ld<In>(x0, 32 * a);
        if (second) {
#pragma unroll
          for (int a = 16 + 2 * HOP32; a < 16 + 3 * HOP32; ++a) raw[a] = 0.0f;
        } else {
#pragma unroll
          for (int a = 16 + 2 * HOP32; a < 16 + 3 * HOP32; ++a) raw[a] = ld<In>(x0, 32 * a);
        }
        if constexpr (TABLE) loudest = max_abs(raw);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          re[brev4(a)] = raw[a] * win[a];
          im[brev4(a)] = raw[a + HOP32] * win[a];
          re[16 + brev4(a)] = raw[a + 2 * HOP32] * win[a];
          im[16 + brev4(a)] = raw[a + 3 * HOP32] * win[a];
        }
      } else {
        const In *x1 = x0 + p.hop, *x2 = x1 + p.hop, *x3 = x2 + p.hop;
        float v3s[16];
        if (second) {
#pragma unroll
          for (int a = 0; a < 16; ++a) v3s[a] = 0.0f;
        } else {
#pragma unroll
          for (int a = 0; a < 16; ++a) v3s[a] = ld<In>(x3, 32 * a);
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          const float v0 = ld<In>(x0, 32 * a), v1 = ld<In>(x1, 32 * a), v2 = ld<In>(x2, 32 * a), v3 = v3s[a];
          if constexpr (TABLE)
            loudest = max3_abs(v2, v3, max3_abs(v0, v1, loudest));
          re[brev4(a)] = v0 * win[a];
          im[brev4(a)] = v1 * win[a];
          re[16 + brev4(a)] = v2 * win[a];
          im[16 + brev4(a)] = v3 * win[a];
        }
      }
      if constexpr (REDO) {
        // (a NaN sample does not compare as loud and needs no second look: it poisons every bin either way)
        if (mode == 0 && p.loud_thr_raw > 0.0f && __builtin_amdgcn_ballot_w64(loudest > p.loud_thr_raw) != 0) mode = 1;
        if (mode != 0) {
#pragma unroll
          for (int a = 0; a < 32; ++a) im[a] = 0.0f;
        }
      }
    }
    transform512<TABLE>(re, im, tw_lds, tw_flat, plane, l);

    // Epilogue in two halves of 128 bins.  After the transpose lane 16g + k1 holds Z_g[k1 + 16*k2] in register k2: for half P the
    // registers 8P..8P+7 go to cell g*128 + (k - 128P), and the partners Z_g[512 - j], j = 128P..128P+128, sit in
    // registers 24-8P..31-8P (j = 256 - k1 - 16*(k2 - 16)); the two ends of that range are one register off:
    // j = 0 is Z[0] (register 0 of lane k1 = 0) and j = 128 of half 1 is Z[384] (register 24 of lane k1 = 0).
    // (everything derived from the lane below is invariant in the frame loop; the empty asm keeps the compiler from
    // hoisting those addresses into registers that the transforms need -- they would be spilled)
    int le = l;
    asm volatile("" : "+v"(le));
    float best[4] = {-1.0f, -1.0f, -1.0f, -1.0f};  // FEAT: largest |X|^2 so far of frame 2q + {0, 1}, q = transform read
    int kbest[4] = {(int)p.bin_lo, (int)p.bin_lo, (int)p.bin_lo, (int)p.bin_lo};
    float next[4] = {-1.0f, -1.0f, -1.0f, -1.0f};  // FEAT: the runner-up (decides whether float32 can tell)
    // FEAT, band narrower than the spectrum: the largest |X|^2 of any bin of each frame (what the transform's rounding
    // noise and the pair's cross-talk scale with; see feature_uncertain)
    const bool banded = p.bin_lo != 0 || p.bin_hi != 256;  // wave-uniform
    float seen[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // ALONE (REDO, mode 1 / 2): every transform held one frame; its spectrum is Z itself, taken twice so that the
    // quantiser sees the same factor 2 as the pair separation leaves; it is frame fb + 2q (mode 1) or fb + 2q + 1 (mode 2).
    auto epilogue = [&](auto alone_c) __attribute__((always_inline)) {
    constexpr bool ALONE = decltype(alone_c)::value;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      epilogue_cells(P, re, im, cell, le);
      wave_sync();
#pragma unroll
      for (int q = 0; q < 2; ++q) {  // transform q, bins 128P + 4l .. +3
        const int k = 128 * P + 4 * le;
        const float2* lr = cell + q * 128 + 4 * le;
        const float2* mr = cell + kMirrorBase + q * kMirrorStride + 4 * le;
        const float4 z01 = *reinterpret_cast<const float4*>(lr);
        const float4 z23 = *reinterpret_cast<const float4*>(lr + 2);
        const float4 q01 = ALONE ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(mr);
        const float4 q23 = ALONE ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(mr + 2);
        uint32_t wa = 0, wb = 0;
        float pa[4] = {0, 0, 0, 0}, pb[4] = {0, 0, 0, 0};
        const bool in_band = k >= (int)p.bin_lo && k < (int)p.bin_hi;
#define SONO_BIN(B, ZR, ZI, PR, PI)                                                              \
  {                                                                                              \
    const float ar = ALONE ? (ZR) + (ZR) : (ZR) + (PR), ai = ALONE ? (ZI) + (ZI) : (ZI) - (PI);  \
    const float br = (ZI) + (PI), bi = (PR) - (ZR);                                              \
    const float ma = __builtin_fmaf(ar, ar, ai * ai);                                            \
    const float mb = __builtin_fmaf(br, br, bi * bi);                                            \
    if constexpr (FEAT) { /* a lane visits the bins of a frame in ascending order */             \
      pa[B] = ma;                                                                                \
      pb[B] = mb;                                                                                \
    } else {                                                                                     \
      wa = put_index<PLUGIN, B>(ma, fs, wa);                                                     \
      wb = put_index<PLUGIN, B>(mb, fs, wb);                                                     \
    }                                                                                            \
  }
        SONO_BIN(0, z01.x, z01.y, q01.x, q01.y)
        SONO_BIN(1, z01.z, z01.w, q01.z, q01.w)
        SONO_BIN(2, z23.x, z23.y, q23.x, q23.y)
        SONO_BIN(3, z23.z, z23.w, q23.z, q23.w)
#undef SONO_BIN
        if constexpr (FEAT) {
          if (banded) {
            seen[2 * q] = fmaxf(fmaxf(fmaxf(seen[2 * q], pa[0]), pa[1]), fmaxf(pa[2], pa[3]));
            seen[2 * q + 1] = fmaxf(fmaxf(fmaxf(seen[2 * q + 1], pb[0]), pb[1]), fmaxf(pb[2], pb[3]));
          }
          if (in_band) {  // (one branch per bin quad; selects inside)
#pragma unroll
            for (int B = 0; B < 4; ++B) {
              next[2 * q] = __builtin_amdgcn_fmed3f(pa[B], best[2 * q], next[2 * q]);
              next[2 * q + 1] = __builtin_amdgcn_fmed3f(pb[B], best[2 * q + 1], next[2 * q + 1]);
              const bool ga = pa[B] > best[2 * q], gb = pb[B] > best[2 * q + 1];
              best[2 * q] = ga ? pa[B] : best[2 * q]; kbest[2 * q] = ga ? k + B : kbest[2 * q];
              best[2 * q + 1] = gb ? pb[B] : best[2 * q + 1]; kbest[2 * q + 1] = gb ? k + B : kbest[2 * q + 1];
            }
          }
        }
        if (!FEAT && in_band) {
          const uint64_t fa = fb + 2 * q;
          const uint64_t row_a = (p.col0 + fa) * p.out_bins, row_b = row_a + p.out_bins;
          const uint32_t col = k - p.out_bin0;
          if constexpr (ALONE) {
            const uint64_t fr = mode == 2 ? fa + 1 : fa;
            if (fr >= f0 && fr < p.nframes) *reinterpret_cast<uint32_t*>(out + (mode == 2 ? row_b : row_a) + col) = wa;
          } else {
            if (fa >= f0 && fa < p.nframes) *reinterpret_cast<uint32_t*>(out + row_a + col) = wa;
            if (fa + 1 >= f0 && fa + 1 < p.nframes) *reinterpret_cast<uint32_t*>(out + row_b + col) = wb;
          }
        }
      }
      wave_sync();  // the buffer is rewritten by the next half / the next iteration's transpose
    }
    };  // epilogue
    if (REDO && mode != 0) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if constexpr (FEAT) {
      // larger |X|^2 wins, equal values go to the lower bin
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        reduce_peak<32>(best[i], kbest[i], next[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) seen[i] = fmaxf(best[i], banded ? reduce_max<32>(seen[i]) : 0.0f);  // the frame's largest |X|^2 anywhere
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint64_t fr = fb + i;
        if (l == 0 && fr >= f0 && fr < p.nframes) {
          // reserved[0] = 1: float32 cannot tell; the exact-arithmetic pass that follows recomputes the record
          sono_frame_feature r{};
          r.peak_bin = kbest[i];
          r.peak_idx = (uint8_t)(PLUGIN ? fast_plugin_peak_index(best[i], fs) : put_index<false, 0>(best[i], fs, 0));
          r.reserved[0] = (feature_uncertain(best[i], next[i], seen[i ^ 1], seen[i],
                                             __builtin_fmaf(__log2f(best[i]), fs.a, fs.b), fs) ||
                           peak_outside(kbest[i], p.feat_lo, p.feat_hi)) ? 1 : 0;
          reinterpret_cast<sono_frame_feature*>(p.out)[p.col0 + fr] = r;
        }
      }
    }
    if constexpr (REDO) {
      if (mode == 1) {
        mode = 2;
        continue;
      }
      mode = 0;
    }
    ++itn;
  }
}

// RGBA: the image [row = row_of_bin0 - bin][column = col0 + frame] of LUT colours instead of [frame][bin] indices,
// with the workgroup's exchange of colour indices in LDS and whole-sector stores (sono_image_ring.h), arranged like the
// 1024-point image writer (kernels_fast_rgba.hip):
//   * a round is 256 consecutive frames = two iterations (a wavefront takes 8 frames per iteration, a half-wavefront 4 of
//     them: frame group 32*iteration + 2*wave + h); a workgroup owns a contiguous range of rounds;
//   * the transposes go through 16 x 16 blocks (4.25 KiB per wavefront); in the epilogue only the partners Z[512 - k]
//     travel through a transform's block (two rounds of eight registers), Z[k] stays in its lane, and
//     v_permlane16_swap trades between the two transforms of a half-wavefront, so that a lane holds one dword (four
//     frames) of each of the bins k1 + 16*(2m + g), m = 0..7: what a ring row takes;
//   * one ring slot (256 bin rows x 272 frames), two barriers per round, both between iterations: iteration 0 writes out
//     the parts 0, 1 of the round before (wavefront w owns the 16 rows of bins 16w .. 16w+15; a part is 4 rows; a lane
//     stores 16 bytes of a row's 1-KiB run), transforms, keeps its epilogue's indices in registers and reads the parts
//     2, 3; iteration 1: barrier, writes out the parts 2, 3, transforms, stores both epilogues' indices into the slot,
//     barrier, reads the parts 0, 1 of the new round.
// At the end of the recording a block of four frames is shifted back (loads stay in bounds) and the lane moves the
// frames it owns to the front of its cell.
constexpr int kRF = 256;            // frames per workgroup round
constexpr int kRIters = kRF / 128;  // iterations per round
constexpr int kWaveBytes = 4 * fast1024::kBlkFloats * 4;  // 4352
constexpr int kBlkBytes = fast1024::kBlkFloats * 4;       // 1088: a 16-lane row's block; epilogue round: 8 rows of 16 partners
static_assert(8 * 16 * 8 <= kBlkBytes, "an epilogue round must fit the row's block");
constexpr int kLutBytes = 1024;
constexpr int kOffWave = kTwBytes + kWinBytes + kLutBytes;
constexpr int kOffSlot = kOffWave + kWaves * kWaveBytes;
constexpr int kRgbaLdsBytes = kOffSlot + image_ring::Ring<kRF>::bytes(256);
static_assert(kRgbaLdsBytes <= 160 * 1024, "LDS of one CU");

template <bool PLUGIN, typename In, int HOP32, bool TABLE>
__global__ __launch_bounds__(kWaves * 64) void stft512_rgba_kernel(StftParams p, FastScale fs, const float2* __restrict__ tw_table) {
  using namespace image_ring;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw_lds = reinterpret_cast<float2*>(smem);
  float* win_lds = reinterpret_cast<float*>(smem + kTwBytes);
  uint32_t* lut_lds = reinterpret_cast<uint32_t*>(smem + kTwBytes + kWinBytes);
  uint32_t* slot = reinterpret_cast<uint32_t*>(smem + kOffSlot);
  const float2* __restrict__ tw_flat = reinterpret_cast<const float2*>(p.twiddles);
  for (int i = threadIdx.x; i < 512; i += kWaves * 64) {
    if constexpr (TABLE) {
      const int r = i >> 4, kk = i & 15;
      if (r < 31) {
        const int t = 1 << (31 - __builtin_clz(r + 1)), j = r + 1 - t;
        tw_lds[i] = tw_flat[16 * t - 1 + 16 * j + kk];
      }  // (the 15 factors of the first pass are read by scalar loads)
    } else {
      tw_lds[i] = tw_table[i];
    }
    win_lds[(i & 31) * kWinStride + (i >> 5)] = p.window[i];
  }
  for (int i = threadIdx.x; i < 256; i += kWaves * 64) lut_lds[i] = p.lut[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* wbuf = smem + kOffWave + wave * kWaveBytes;
  float* blk = reinterpret_cast<float*>(wbuf + (lane >> 4) * kBlkBytes);  // this 16-lane row's block: transposes, then epilogue
  const float4* win_row = reinterpret_cast<const float4*>(win_lds + l * kWinStride);
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  // every workgroup runs the same number of rounds and every wavefront goes through every round (barriers); a workgroup's
  // rounds are consecutive frames
  const uint64_t nrounds = (p.nframes + kRF - 1) / kRF;
  const uint64_t rounds = (nrounds + gridDim.x - 1) / gridDim.x;
  const Range rg = make_range<kRF>(p, vb, rounds);

  // first frame of this half-wavefront in iteration `it` (>= nframes: nothing of its own)
  auto first_frame = [&](uint64_t it) -> uint64_t {
    if (it >= rounds * kRIters) return p.nframes;
    return rg.f_lo + 128 * it + 8 * wave + 4 * h;
  };
  constexpr int kRaw = HOP32 > 0 ? 16 + 3 * HOP32 : 64;
  float raw[kRaw];
  auto fetch = [&](uint64_t f0) {  // the block fb .. fb+3 that holds this half's frames, shifted back at the end
    const uint64_t fb = f0 + 4 <= p.nframes ? f0 : p.nframes - 4;
    const In* x0 = x + (p.frame0 + fb) * p.hop + l;
    if constexpr (HOP32 > 0) {
#pragma unroll
      for (int a = 0; a < kRaw; ++a) raw[a] = ld<In>(x0, 32 * a);
    } else {
      const In *x1 = x0 + p.hop, *x2 = x1 + p.hop, *x3 = x2 + p.hop;
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        raw[a] = ld<In>(x0, 32 * a);
        raw[16 + a] = ld<In>(x1, 32 * a);
        raw[32 + a] = ld<In>(x2, 32 * a);
        raw[48 + a] = ld<In>(x3, 32 * a);
      }
    }
  };
  // this lane's four frames of each of the four rows of the parts `first_part`, +1 of round k: wavefront w owns the rows of
  // bins 16w .. 16w+15, lane j the frames 256k - e + 4j .. +3
  auto read_rows = [&](uint32_t (&fq)[2][4], int first_part, uint64_t k) {
    int w = wave, j = lane;
    asm volatile("" : "+s"(w));
    asm volatile("" : "+v"(j));
#pragma unroll
    for (int q = 0; q < 2; ++q) read4<kRF>(slot, p, rg, fq[q], 16 * w + 4 * (first_part + q), j, k);
  };
  auto flush_rows = [&](const uint32_t (&d)[4], uint64_t fround, int part) {
    int w = wave, j = lane;
    asm volatile("" : "+s"(w));   // (loop-invariant addresses computed in place: hoisted, they would hold registers the
    asm volatile("" : "+v"(j));   //  transforms need)
    const int bin0 = 16 * w + 4 * part;
    if (bin0 + 4 <= (int)p.bin_lo || bin0 >= (int)p.bin_hi) return;
    flush4<kRF>(p, rg, d, lut_lds, fround, bin0, j, bin0 >= (int)p.bin_lo && bin0 + 4 <= (int)p.bin_hi);
  };

  fetch(first_frame(0));
  uint32_t fq[2][4] = {};    // indices of the two parts the next iteration writes out
  uint32_t pend[8] = {};     // dwords of iteration 0's epilogue, stored into the slot in iteration 1
  bool pend_active = false;
  // Frame group fg (four frames) of round k of this workgroup: one dword into each of the rows of the bins
  // bin0 + 32*m, m = 0..7 (whole groups of 32 rows outside the band are skipped; the rest of the slot is never read).
  auto park8 = [&](const uint32_t (&r)[8], uint64_t k, int fg, int bin0) {
    constexpr int kD = Ring<kRF>::kDwords, kP = Ring<kRF>::kPitch;
    int d = (int)(((uint64_t)(kRF / 4) * k) % kD) + fg;
    d -= d >= kD ? kD : 0;
    uint32_t* row = slot + bin0 * kP + d;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      if (32 * m + 31 >= (int)p.bin_lo && 32 * m < (int)p.bin_hi) row[m * 32 * kP] = r[m];
  };
  auto park8_or = [&](const uint32_t (&r)[8], uint64_t k, int fg, int bin0) {  // ... OR-ed onto the dwords there
    constexpr int kD = Ring<kRF>::kDwords, kP = Ring<kRF>::kPitch;
    int d = (int)(((uint64_t)(kRF / 4) * k) % kD) + fg;
    d -= d >= kD ? kD : 0;
    uint32_t* row = slot + bin0 * kP + d;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      if (32 * m + 31 >= (int)p.bin_lo && 32 * m < (int)p.bin_hi) row[m * 32 * kP] |= r[m];  // (this lane's own dwords)
  };
  // Under the context's table a unit too loud for the pair separation (StftParams::loud_thr_raw) is transformed one frame
  // per transform: the iteration's transforms run twice, mode 1 = the first frames of the pairs alone, then mode 2 = the
  // second frames alone (whose bytes are OR-ed onto the first ones').  Wave-uniform: the eight frames of a wavefront go
  // together.  A NaN sample does not compare as loud and needs no second look (it poisons every bin either way).
  int mode = 0;
  for (uint64_t it = 0; it < (rounds + 2) * kRIters;) {
    const uint64_t round = it / kRIters;
    const int sub = (int)(it % kRIters);
    const uint64_t f0 = first_frame(it);
    if (!(TABLE && mode == 2)) {
      // iteration 1: every wavefront has read its parts 2, 3 of the round before (end of its iteration 0), so the slot
      // may be overwritten from here on
      if (sub == 1) __syncthreads();
      // (the empty asm is a use of the loaded registers: the wait for them lands here, before the image stores)
#pragma unroll
      for (int a = 0; a < kRaw; ++a) asm volatile("" : "+v"(raw[a]));
      if (round > 0 && rg.f_lo < rg.f_hi) {
        const uint64_t fround = rg.f_lo + (round - 1) * kRF;
        flush_rows(fq[0], fround, 2 * sub);
        flush_rows(fq[1], fround, 2 * sub + 1);
      }
    }
    // wave-uniform: does this wavefront have a frame in this iteration?
    const bool active = it < rounds * kRIters && rg.f_lo + 128 * it + 8 * wave < p.nframes;
    if (active) {
      if constexpr (TABLE) {
        if (mode == 0 && p.loud_thr_raw > 0.0f && __builtin_amdgcn_ballot_w64(max_abs(raw) > p.loud_thr_raw) != 0) mode = 1;
      }
      float re[32], im[32];
      {
        float win[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = win_row[j];
          win[4 * j] = v.x; win[4 * j + 1] = v.y; win[4 * j + 2] = v.z; win[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          re[brev4(a)] = raw[a] * win[a];
          im[brev4(a)] = raw[a + (HOP32 > 0 ? HOP32 : 16)] * win[a];
          re[16 + brev4(a)] = raw[a + (HOP32 > 0 ? 2 * HOP32 : 32)] * win[a];
          im[16 + brev4(a)] = raw[a + (HOP32 > 0 ? 3 * HOP32 : 48)] * win[a];
        }
        if (TABLE && mode != 0) {
#pragma unroll
          for (int a = 0; a < 32; ++a) im[a] = 0.0f;
        }
      }
      transform512<TABLE, true>(re, im, tw_lds, tw_flat, blk, l);  // register k2 = Z_g[k1 + 16*k2] in lane 16g + k1

      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));  // (keeps loop-invariant addresses out of registers the transforms need)
      const int ke = lane_e & 15, ge = (lane_e >> 4) & 1, he = lane_e >> 5;
      // frames of this lane's cell: f0 .. f0+3; the block computed is fb = f0 - shift .. (shift > 0 at the end)
      const uint64_t fb = f0 + 4 <= p.nframes ? f0 : p.nframes - 4;
      const int shift = f0 >= p.nframes ? 4 : (int)(f0 - fb);
      const int fg = 32 * sub + 2 * wave + he;
      // Epilogue.  Bin k = k1 + 16*k2 (k2 < 16) stays in its lane; its partner Z[512 - k] sits in lane 16 - k1 of the same
      // transform, register 31 - k2 (lane 0: its own register 32 - k2).  Only the partners travel: two rounds of eight
      // registers (16..23, then 24..31) through the transform's 1-KiB block, register r0 + j of lane k1 at [j][k1].  Lane 0
      // stores one row down and puts register 24 (round 0) / 0 (round 1) into row 7, so that every lane reads
      // [j][(16 - k1) mod 16].  Byte 0 / 1 of w[k2]: the colour indices of the transform's frames A / B at that bin.
      unsigned char* cb = reinterpret_cast<unsigned char*>(blk);
      const bool lane0 = ke == 0;
      unsigned char* wr = cb + 8 * ke - (lane0 ? 128 : 0);
      unsigned char* wr0 = lane0 ? cb + 7 * 128 : cb + 8 * ke;
      const unsigned char* rd = cb + 8 * ((16 - ke) & 15);
      uint32_t w[16];
      if (TABLE && mode != 0) {
        // a frame alone: its spectrum is Z itself (no partner; the same factor 2 as the pair separation leaves); the
        // other frame's byte stays 0 in this pass
        if (mode == 1) {
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) {
            const float ar = re[k2] + re[k2], ai = im[k2] + im[k2];
            w[k2] = put_index<PLUGIN, 0>(__builtin_fmaf(ar, ar, ai * ai), fs, 0u);
          }
        } else {
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) {
            const float ar = re[k2] + re[k2], ai = im[k2] + im[k2];
            w[k2] = put_index<PLUGIN, 1>(__builtin_fmaf(ar, ar, ai * ai), fs, 0u);
          }
        }
      } else {
#pragma unroll
      for (int R = 0; R < 2; ++R) {
        const int r0 = 16 + 8 * R, rx = (24 + 8 * R) & 31;
        *reinterpret_cast<float2*>(wr0) = lane0 ? make_float2(re[rx], im[rx]) : make_float2(re[r0], im[r0]);
#pragma unroll
        for (int j = 1; j < 8; ++j) *reinterpret_cast<float2*>(wr + 128 * j) = make_float2(re[r0 + j], im[r0 + j]);
        wave_sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k2 = 15 - 8 * R - j;
          const float2 q = *reinterpret_cast<const float2*>(rd + 128 * j);
          const float zr = re[k2], zi = im[k2];
          const float ar = zr + q.x, ai = zi - q.y;
          const float br = zi + q.y, bi = q.x - zr;
          const uint32_t t = put_index<PLUGIN, 0>(__builtin_fmaf(ar, ar, ai * ai), fs, 0u);
          w[k2] = put_index<PLUGIN, 1>(__builtin_fmaf(br, br, bi * bi), fs, t);
        }
        wave_sync();  // the block is rewritten by the next round / the next iteration's transpose
      }
      }
      // the lanes of transform g hold the frames (fb + 2g, +1); after the swap lane (g, k1) holds all four frames of the
      // bins k1 + 16*(2m + g), m = 0..7, as one dword per bin: what a ring row takes.  At the end of the recording the
      // frames this lane owns start at byte `shift` of the dword.
      uint32_t r8[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const auto sw = __builtin_amdgcn_permlane16_swap(w[2 * m], w[2 * m + 1], false, false);
        r8[m] = (sw[0] | (sw[1] << 16)) >> (8 * (shift & 3));
      }
      if (sub == 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) pend[m] = (TABLE && mode == 2) ? pend[m] | r8[m] : r8[m];
      } else if (TABLE && mode == 2) {
        park8_or(r8, round, fg, ke + 16 * ge);  // (onto what the pass for the first frames left there)
      } else {
        park8(r8, round, fg, ke + 16 * ge);
      }
    }
    if constexpr (TABLE) {
      if (mode == 1) {
        // the first frames alone are done: the same unit once more for the second frames alone -- the block starts one
        // frame later and ends with its fourth frame (samples from the cache; nothing behind them is read, the frame
        // after it may lie beyond the signal)
        const uint64_t fb = f0 + 4 <= p.nframes ? f0 : p.nframes - 4;
        const In* x0 = x + (p.frame0 + fb + 1) * p.hop + l;
        asm volatile("" : "+v"(x0));  // (opaque: or the samples already loaded for the pairs are kept in registers instead)
        if constexpr (HOP32 > 0) {
#pragma unroll
          for (int a = 0; a < 16 + 2 * HOP32; ++a) raw[a] = ld<In>(x0, 32 * a);
#pragma unroll
          for (int a = 16 + 2 * HOP32; a < kRaw; ++a) raw[a] = 0.0f;
        } else {
          const In* x2 = x0 + 2 * (uint64_t)p.hop;
#pragma unroll
          for (int a = 0; a < 16; ++a) {
            raw[a] = ld<In>(x0, 32 * a);
            raw[16 + a] = 0.0f;
            raw[32 + a] = ld<In>(x2, 32 * a);
            raw[48 + a] = 0.0f;
          }
        }
        mode = 2;
        continue;
      }
      mode = 0;
    }
    if (sub == 0) {
      pend_active = active;
    } else if (pend_active) {  // frame group 2*wave + h of the round: the dwords iteration 0 kept
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      park8(pend, round, 2 * wave + (lane_e >> 5), lane_e & 31);
    }
    fetch(first_frame(it + 1));  // unconditional: a half with nothing of its own reads the last block of the recording
    if (sub == 0) {
      // the slot still holds the round before: take the last two parts of this wavefront's 16 rows
      if (round > 0) read_rows(fq, 2, round - 1);
    } else {
      // the round's slot is complete once every wavefront is here; then take the first two parts (after the last round:
      // of the empty round that only has the frames the last one left behind)
      __syncthreads();
      if (round <= rounds) read_rows(fq, 0, round);
    }
    ++it;
  }
}

}  // namespace fast512

// W_512^(l*k1) as [k1][l] float2 (cos, -sin), k1 < 16, l < 32, computed in double.
void build_fast_table_512(std::vector<float>& t) {
  t.resize(2 * 512);
  const double pi = 3.141592653589793;
  for (int k1 = 0; k1 < 16; ++k1)
    for (int l = 0; l < 32; ++l) {
      const double ang = -2.0 * pi * (double)(k1 * l) / 512.0;
      t[2 * (k1 * 32 + l)] = (float)std::cos(ang);
      t[2 * (k1 * 32 + l) + 1] = (float)std::sin(ang);
    }
}

bool fast512_available(const StftParams& p) {
  if (p.n != 512 || p.nframes < 8) return false;
  if (!((p.fast_mode == 1 && p.fast_table) || (p.fast_mode == 2 && p.twiddles))) return false;
  if ((p.bin_lo & 3) || (p.bin_hi & 3) || p.bin_lo >= p.bin_hi) return false;
  if (p.out_kind == SONO_OUT_FEATURES) return true;
  if (p.out_kind != SONO_OUT_U8_INDEX || !p.pair_idx_ok) return false;
  return !(p.out_bins & 3) && !(p.out_bin0 & 3) && !(reinterpret_cast<uintptr_t>(p.out) & 3);
}

// The image output needs the flipped full-height row mapping and a LUT (any band; no scratch buffer).
bool fast512_rgba_available(const StftParams& p) {
  if (p.n != 512 || p.nframes < 8 || !p.pair_idx_ok) return false;
  if (!((p.fast_mode == 1 && p.fast_table) || (p.fast_mode == 2 && p.twiddles))) return false;
  if (p.bin_lo >= p.bin_hi) return false;
  if (p.row_of_bin0 != (int32_t)p.bin_hi - 1) return false;
  return p.width != 0 && p.width <= ((uint64_t)1 << 24) && p.lut;
}

// Returns 1 if this launch is not served here.
int launch_stft_fast512(const StftParams& p, void* stream_v) {
  if ((p.out_kind == SONO_OUT_RGBA8 || p.out_kind == SONO_OUT_U8_INDEX) && !p.pair_idx_ok) return 1;  // two frames per transform: see StftParams::pair_idx_ok
  using namespace fast512;
  const bool rgba = p.out_kind == SONO_OUT_RGBA8;
  if (rgba ? !fast512_rgba_available(p) || (reinterpret_cast<uintptr_t>(p.out) & 3) : !fast512_available(p)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  FastScale fs = make_fast_scale(p.scale_mode, p.n, p.gain_db, p.range_db);
  fs.b -= 2.0f * fs.a;  // the pair separation leaves 2*X: |2X|^2 = 4|X|^2
  fs.xtalk = (float)p.pair_asym;
  const int cus = device_cu_count();
  const uint64_t nunits = (p.nframes + 7) / 8;
  const uint64_t need = (nunits + kWaves - 1) / kWaves;
  const uint32_t grid = (uint32_t)(need < (uint64_t)cus ? need : (uint64_t)cus);  // persistent: one workgroup per CU
  const float2* table = reinterpret_cast<const float2*>(p.fast_table);
  const bool plugin = p.scale_mode == SONO_SCALE_PLUGIN_DB, i16 = p.in_kind == SONO_IN_I16, feat = p.out_kind == SONO_OUT_FEATURES;
  const bool tab = p.fast_mode == 2;
  if (rgba) {
    const uint64_t nr = (p.nframes + kRF - 1) / kRF;
    const uint32_t rgrid = (uint32_t)(nr < (uint64_t)cus ? nr : (uint64_t)cus);
    const int lds = kRgbaLdsBytes;
#define SONO_R512(PL, In, H, T)                                                                                 \
  do {                                                                                                          \
    auto kern = stft512_rgba_kernel<PL, In, H, T>;                                                              \
    SONO_GRANT_LDS(kern, lds);                                                                                  \
    hipLaunchKernelGGL(kern, dim3(rgrid), dim3(kWaves * 64), lds, stream, p, fs, table);                        \
  } while (0)
#define SONO_R512_H(PL, In)                                                                                    \
  do {                                                                                                         \
    if (tab) { if (p.hop == 128) SONO_R512(PL, In, 4, true); else SONO_R512(PL, In, 0, true); }                 \
    else { if (p.hop == 128) SONO_R512(PL, In, 4, false); else SONO_R512(PL, In, 0, false); }                   \
  } while (0)
    if (plugin) { if (i16) SONO_R512_H(true, int16_t); else SONO_R512_H(true, float); }
    else { if (i16) SONO_R512_H(false, int16_t); else SONO_R512_H(false, float); }
#undef SONO_R512_H
#undef SONO_R512
    // (redo: the kernel transforms its loud units once more, one frame per transform)
    note_kernel(tab ? (p.loud_thr_raw > 0.0f ? "stft512_rgba_kernel<table,redo>" : "stft512_rgba_kernel<table>") : "stft512_rgba_kernel<roots>");
    return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
  }
#define SONO_L512(PL, In, FE, H, T)                                                                              \
  do {                                                                                                          \
    auto kern = stft512_kernel<PL, In, FE, H, T>;                                                                \
    SONO_GRANT_LDS(kern, kLdsBytes);                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), kLdsBytes, stream, p, fs, table);                   \
  } while (0)
#define SONO_L512_F(PL, In)                                                                                   \
  do {                                                                                                        \
    if (tab) {                                                                                                \
      if (p.hop == 128) { if (feat) SONO_L512(PL, In, true, 4, true); else SONO_L512(PL, In, false, 4, true); } \
      else { if (feat) SONO_L512(PL, In, true, 0, true); else SONO_L512(PL, In, false, 0, true); }             \
    } else {                                                                                                  \
      if (p.hop == 128) { if (feat) SONO_L512(PL, In, true, 4, false); else SONO_L512(PL, In, false, 4, false); } \
      else { if (feat) SONO_L512(PL, In, true, 0, false); else SONO_L512(PL, In, false, 0, false); }           \
    }                                                                                                         \
  } while (0)
  if (plugin) { if (i16) SONO_L512_F(true, int16_t); else SONO_L512_F(true, float); }
  else { if (i16) SONO_L512_F(false, int16_t); else SONO_L512_F(false, float); }
#undef SONO_L512_F
#undef SONO_L512
  note_kernel(tab ? (p.out_kind == SONO_OUT_U8_INDEX && p.loud_thr_raw > 0.0f ? "stft512_kernel<table,redo>" : "stft512_kernel<table>")
                  : "stft512_kernel<roots>");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
