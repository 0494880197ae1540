// Float32 STFT kernel for 2048-point frames with exact roots of unity (the worker path at the page's largest FFT
// size, main.js:886-889; spectrogramWorker.js:38-99,206-253).  Compile with -ffp-contract=fast.
//
// One wavefront (64 lanes x 32 registers) transforms TWO real frames as one complex 2048-point frame z = x0 + i*x1:
//   * lane l holds samples 64a + l, a = 0..31; 2048 = 32 x 64: the first pass is a 32-point transform per lane
//     (Y_l[k1]), then the twiddle W_2048^(l*k1);
//   * a 32 x 64 transpose through LDS: lane k1 + 32*hb reads y[m + 32*hb], m = 0..31, of row k1;
//   * the 64-point transform over l is decimated in frequency: one radix-2 stage ACROSS the two lanes of a row --
//     two rounds of v_permlane32_swap leave the sums u_m = y[m] + y[m+32] (all 32 of them) in the upper lane of the
//     pair and the differences d_m = y[m] - y[m+32] in the lower lane, which multiplies them by W_64^m -- and then one
//     32-point transform per lane: the lower lane ends with Z[k1 + 32*(2*k2 + 1)], the upper with Z[k1 + 32*(2*k2)];
//   * epilogue in two halves of 512 bins: registers 8P..8P+7 of all 64 lanes are Z[k], 512P <= k < 512P+512, registers
//     24-8P..31-8P their partners Z[2048 - k]; both go through LDS in ascending k so that a lane reads four
//     consecutive bins of both as 16-byte pieces, separates the two frames and fuses |X|^2 -> log2 -> quantiser.
//   * HOP64 > 0: hop == 64*HOP64, so the two frames are one run of 32 + HOP64 loads.
// Outputs: [frame][bin] colour indices, or per-frame spectral peaks.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cmath>
#include <vector>

#include "sono_device.h"
#include "sono_internal.h"
#include "sono_launch.h"
#include "sono_loud.h"
#include "sono_rgba_flush.h"

namespace sono {
namespace fast2048 {

#ifndef SONO_SYNC_ITERS_2048
#define SONO_SYNC_ITERS_2048 1
#endif
constexpr int kSyncIters = SONO_SYNC_ITERS_2048;  // index kernel: iterations between two barriers of its wavefronts

constexpr int kWaves = 16;                          // wavefronts per workgroup (one workgroup per CU): 32 frames per iteration
constexpr int kPlaneStride = 68;                    // floats per transpose row (64 + 4: rows stay 16-B aligned)
constexpr int kBufBytes = 32 * kPlaneStride * 4;    // 8704 per wavefront
constexpr int kMirrorBase = 512;                    // epilogue, float2 cells: Z[512P + j] at j, Z[2048 - 512P - j] at 512 + j (j <= 512)
static_assert((kMirrorBase + 513) * 8 <= kBufBytes, "epilogue must fit the transpose buffer");
constexpr int kTwBytes = 32 * 64 * 8;               // W_2048^(l*k1) as [k1][l] float2
constexpr int kWinBytes = 2048 * 4;                 // window as stored: lane l reads w[64a + l]
constexpr int kLdsBytes = kTwBytes + kWinBytes + kWaves * kBufBytes;
static_assert(kLdsBytes <= 160 * 1024, "LDS of one CU");

__device__ constexpr float kC32[16] = {1.0f,          0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                       0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
                                       0.0f,          -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
                                       -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
__device__ constexpr float kS32[16] = {0.0f,          0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                                       0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f,
                                       1.0f,          0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                       0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
// cos/sin(2*pi*m/64), m = 0..31
__device__ constexpr float kC64[32] = {
    1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f, 0.88192126434835502971f,
    0.83146961230254523708f, 0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f, 0.55557023301960222474f,
    0.47139673682599764856f, 0.38268343236508977173f, 0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f,
    0.0f, -0.09801714032956060199f, -0.19509032201612826785f, -0.29028467725446236764f, -0.38268343236508977173f,
    -0.47139673682599764856f, -0.55557023301960222474f, -0.63439328416364549822f, -0.70710678118654752440f, -0.77301045336273696081f,
    -0.83146961230254523708f, -0.88192126434835502971f, -0.92387953251128675613f, -0.95694033573220886494f, -0.98078528040323044913f,
    -0.99518472667219688624f};
__device__ constexpr float kS64[32] = {
    0.0f, 0.09801714032956060199f, 0.19509032201612826785f, 0.29028467725446236764f, 0.38268343236508977173f, 0.47139673682599764856f,
    0.55557023301960222474f, 0.63439328416364549822f, 0.70710678118654752440f, 0.77301045336273696081f, 0.83146961230254523708f,
    0.88192126434835502971f, 0.92387953251128675613f, 0.95694033573220886494f, 0.98078528040323044913f, 0.99518472667219688624f,
    1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f, 0.92387953251128675613f, 0.88192126434835502971f,
    0.83146961230254523708f, 0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f, 0.55557023301960222474f,
    0.47139673682599764856f, 0.38268343236508977173f, 0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f};

__host__ __device__ constexpr int brev5(int v) {
  return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

// In-register 32-point forward DFT, radix-2 decimation in time: input in bit-reversed register order, output natural.
__device__ __forceinline__ void fft32(float (&re)[32], float (&im)[32]) {
#pragma unroll
  for (int s = 0; s < 5; ++s) {
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

__host__ __device__ constexpr int brev6(int v) { return (brev5(v & 31) << 1) | (v >> 5); }

// The same 32-point network with twiddles from a table (the plugin's recurrence factors or a host table: not exact
// roots of unity, so every butterfly must use the very factor the reference uses, spectrogram.esm.js:176-190).
// Registers in position order.  tw(half, j): factor (cos, sin) of the butterfly at position j of the stage with half
// size `half`.  FIRST: the first five stages of the transform, where j == 0 is the factor (1, 0) exactly in every table
// this kernel accepts (checked on the host).
template <bool FIRST, class TwFn>
__device__ __forceinline__ void fft32_table(float (&re)[32], float (&im)[32], TwFn tw) {
#pragma unroll
  for (int s = 0; s < 5; ++s) {
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

// v_permlane32_swap: lanes 32..63 of a trade places with lanes 0..31 of b.
__device__ __forceinline__ void swap32(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
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

// The complex 2048-point transform of one wavefront: in, lane l register brev5(a) = z[64a + l].
// Exact roots (TABLE false): out, register k2 = Z[k1 + 32*(2*k2 + par)] with k1 = lane & 31 and par = 1 in lanes 0..31
// (differences), 0 in lanes 32..63.
// TABLE: the reference's radix-2 network itself, factor by factor.  Lane l's registers are the positions
// 32*brev6(l) + r of the bit-reversed order, so the first pass is the stages T = 1..16 with wave-uniform factors
// (tw_flat, the context's table: stage with half size T at [T-1, 2T-1)); the transpose hands lane k1 + 32*hb the
// positions 32*(32*hb + m) + k1, m = 0..31; stages T = 32t, t = 1..16, pair registers m, m + t with factor
// 32*(m mod t) + k1 (rows t-1+j of tw_lds, [63][32]); the last stage T = 1024 pairs the two lanes of a row with factor
// 32*m + k1 (rows 31 + m): registers m and m + 16 trade halves, each lane does 16 butterflies, and they trade back.
// Out: register m = Z[k1 + 32*m + 1024*hb].
template <bool TABLE>
__device__ __forceinline__ void transform2048(float (&re)[32], float (&im)[32], const float2* tw_lds, const float2* tw_flat,
                                              float* plane, int lane) {
  const int k1r = lane & 31, hb = lane >> 5;
  if constexpr (TABLE) {
    // stages T = 1..16: wave-uniform factors by scalar loads (the table is not written while the kernel runs; the opaque
    // copy keeps the scalars from being held in registers round the frame loop)
    typedef const float __attribute__((address_space(4))) * const_table;
    const_table twc = (const_table)(uintptr_t)tw_flat;
    asm volatile("" : "+s"(twc));
    fft32_table<true>(re, im, [&](int half, int j) { return make_float2(twc[2 * (half - 1 + j)], twc[2 * (half - 1 + j) + 1]); });
  } else {
    fft32(re, im);  // Y_l[k1]
#pragma unroll
    for (int k1 = 1; k1 < 32; ++k1) {
      const float2 t = tw_lds[k1 * 64 + lane];
      const float ur = __builtin_fmaf(re[k1], t.x, -im[k1] * t.y);
      const float ui = __builtin_fmaf(re[k1], t.y, im[k1] * t.x);
      re[k1] = ur;
      im[k1] = ui;
    }
  }
  // transpose: lane l, register k1 -> lane k1 + 32*hb, register m = y[m + 32*hb] of row k1 (natural order in m);
  // TABLE: lane l's column is its position block brev6(l)
  const int col = TABLE ? brev6(lane) : lane;
#pragma unroll
  for (int k1 = 0; k1 < 32; ++k1) plane[k1 * kPlaneStride + col] = re[k1];
  wave_sync();
  {
    const float4* row = reinterpret_cast<const float4*>(plane + k1r * kPlaneStride + 32 * hb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = row[j];
      re[4 * j] = v.x; re[4 * j + 1] = v.y; re[4 * j + 2] = v.z; re[4 * j + 3] = v.w;
    }
  }
  wave_sync();
#pragma unroll
  for (int k1 = 0; k1 < 32; ++k1) plane[k1 * kPlaneStride + col] = im[k1];
  wave_sync();
  {
    const float4* row = reinterpret_cast<const float4*>(plane + k1r * kPlaneStride + 32 * hb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = row[j];
      im[4 * j] = v.x; im[4 * j + 1] = v.y; im[4 * j + 2] = v.z; im[4 * j + 3] = v.w;
    }
  }
  wave_sync();
  if constexpr (TABLE) {
    fft32_table<false>(re, im, [&](int half, int j) { return tw_lds[(half - 1 + j) * 32 + k1r]; });  // stages T = 32..512
    const float2* wrow = tw_lds + (31 + 16 * hb) * 32 + k1r;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      swap32(re[m], re[m + 16]);
      swap32(im[m], im[m + 16]);
      // both lanes: a = position mu (register m), b = position mu + 32 (register m + 16), mu = m + 16*hb
      const float2 w = wrow[m * 32];
      const float ar = re[m], ai = im[m], br = re[m + 16], bi = im[m + 16];
      const float o1r = __builtin_fmaf(-w.y, bi, __builtin_fmaf(w.x, br, ar));
      const float o1i = __builtin_fmaf(w.y, br, __builtin_fmaf(w.x, bi, ai));
      re[m] = o1r; im[m] = o1i;
      re[m + 16] = __builtin_fmaf(2.0f, ar, -o1r);
      im[m + 16] = __builtin_fmaf(2.0f, ai, -o1i);
      swap32(re[m], re[m + 16]);
      swap32(im[m], im[m + 16]);
      // lower lane: sums mu = m (own) and m + 16 (the upper lane's); upper lane: differences mu = m and m + 16
    }
    return;
  }
  // Radix-2 stage across the lane pair (k1, k1 + 32).  First swap: registers m and m + 16 trade halves, so the lower
  // lane holds y[m] (in re[m]) and y[m + 32] (in re[m + 16]) for m < 16 and the upper lane the same for
  // m' = m + 16; sums and differences are taken in place.  Second swap: the sum register m and the difference register
  // m + 16 trade halves again: the lower lane ends with all 32 differences, the upper lane with all 32 sums,
  // each as [values m < 16 | values m >= 16] = natural order.
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    swap32(re[m], re[m + 16]);
    swap32(im[m], im[m + 16]);
    const float sr = re[m] + re[m + 16], si = im[m] + im[m + 16];
    const float dr = re[m] - re[m + 16], di = im[m] - im[m + 16];
    re[m] = dr; im[m] = di;             // differences where the lower lanes will keep them
    re[m + 16] = sr; im[m + 16] = si;
    swap32(re[m], re[m + 16]);
    swap32(im[m], im[m + 16]);
  }
  if (hb == 0) {  // the differences take the factor W_64^m before their 32-point transform
#pragma unroll
    for (int m = 1; m < 32; ++m) {
      const float C = kC64[m], S = kS64[m];  // W_64^m = (C, -S)
      const float ur = __builtin_fmaf(re[m], C, im[m] * S);
      const float ui = __builtin_fmaf(im[m], C, -re[m] * S);
      re[m] = ur;
      im[m] = ui;
    }
  }
  {  // the 32-point network wants its input in bit-reversed register order
    float tr[32], ti[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) { tr[brev5(m)] = re[m]; ti[brev5(m)] = im[m]; }
#pragma unroll
    for (int m = 0; m < 32; ++m) { re[m] = tr[m]; im[m] = ti[m]; }
  }
  fft32(re, im);
}

// Epilogue half P: Z[512P + j] to cell j and the partners Z[2048 - 512P - j] to cell kMirrorBase + j (j <= 512).  With
// kappa = k1 + 32*par a lane's register k2 is Z[kappa + 64*k2]: registers 8P..8P+7 are the bins, registers 24-8P..31-8P
// the partners (j = 512 - kappa - 64*r for register 24-8P+r); j = 0 is one register off: Z[0] (register 0, kappa = 0)
// in half 0, Z[1536] (register 24, kappa = 0) in half 1.
__device__ __forceinline__ void epilogue_cells(int P, const float (&re)[32], const float (&im)[32], float2* cell, int kappa) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    cell[kappa + 64 * r] = make_float2(re[8 * P + r], im[8 * P + r]);
    cell[kMirrorBase + 512 - kappa - 64 * r] = make_float2(re[24 - 8 * P + r], im[24 - 8 * P + r]);
  }
  if (kappa == 0) cell[kMirrorBase] = P ? make_float2(re[24], im[24]) : make_float2(re[0], im[0]);
}

// The same for the TABLE transform's register order (register m = Z[k1 + 32*m + 1024*hb]): the lower lane of a row holds
// the bins (registers 16P..16P+15 -> cells 32r + k1), the upper lane the partners (register 16(1-P) + r is
// Z[2048 - 512P - j] with j = 512 - 32r - k1); j = 0 is Z[0] (lower lane, k1 = 0, register 0) in half 0 and Z[1536]
// (upper lane, k1 = 0, register 16) in half 1.
__device__ __forceinline__ void epilogue_cells_table(int P, const float (&re)[32], const float (&im)[32], float2* cell, int le) {
  const int k1 = le & 31;
  const bool up = le >= 32;
  float2* c0 = cell + (up ? kMirrorBase + 512 - k1 : k1);
  const int step = up ? -32 : 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lo = 16 * P + r, hi = 16 * (1 - P) + r;
    // (the select as the instruction itself: from `up ? re[hi] : re[lo]` the compiler forms an indexed register read)
    auto pick = [](float lower, float upper) {
      float o;
      asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(lower), "v"(upper), "s"(0xffffffff00000000ull));
      return o;
    };
    c0[r * step] = make_float2(pick(re[lo], re[hi]), pick(im[lo], im[hi]));
  }
  if (k1 == 0 && up == (P == 1)) cell[kMirrorBase] = make_float2(re[16 * P], im[16 * P]);
}

template <bool PLUGIN, typename In, bool FEAT, int HOP64, bool TABLE>
__global__ __launch_bounds__(kWaves * 64) void stft2048_kernel(StftParams p, FastScale fs, const float2* __restrict__ tw_table) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw_lds = reinterpret_cast<float2*>(smem);
  float* win_lds = reinterpret_cast<float*>(smem + kTwBytes);
  const float2* __restrict__ tw_flat = reinterpret_cast<const float2*>(p.twiddles);
  for (int i = threadIdx.x; i < 2048; i += kWaves * 64) {
    if constexpr (TABLE) {
      const int r = i >> 5, k = i & 31;  // rows 0..30: stage T = 32t, factor 32j + k (r = t-1+j); rows 31..62: T = 1024, 32m + k
      if (r < 31) {
        const int t = 1 << (31 - __builtin_clz(r + 1)), j = r + 1 - t;
        tw_lds[i] = tw_flat[32 * t - 1 + 32 * j + k];
      } else if (r < 63) {
        tw_lds[i] = tw_flat[1023 + 32 * (r - 31) + k];
      }  // (the 31 factors of the first pass are read by scalar loads)
    } else {
      tw_lds[i] = tw_table[i];
    }
    win_lds[i] = p.window[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* fbuf = smem + kTwBytes + kWinBytes + wave * kBufBytes;
  float* plane = reinterpret_cast<float*>(fbuf);
  float2* cell = reinterpret_cast<float2*>(fbuf);  // epilogue view
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(p.out);
  // blocks with equal blockIdx % 8 share an XCD (and its L2): give them neighbouring frame ranges
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const uint64_t nunits = (p.nframes + 1) >> 1;  // a wavefront takes two consecutive frames
  // a workgroup takes a contiguous range of frames (`iters` iterations of its wavefronts): the samples neighbouring
  // blocks share are re-read from its own cache lines, not fetched again by another CU
  const uint64_t iters = ((nunits + kWaves - 1) / kWaves + gridDim.x - 1) / gridDim.x;
  const uint64_t unit_end = ((uint64_t)vb + 1) * iters * kWaves < nunits ? ((uint64_t)vb + 1) * iters * kWaves : nunits;

  // REDO: under the context's table a pair too loud for the pair separation (StftParams::loud_thr_raw) is transformed one
  // frame at a time: the loop body runs twice more, mode 1 = frame fb alone, mode 2 = frame fb + 1 alone.
  constexpr bool REDO = TABLE && !FEAT;
  int mode = 0;
  // every wavefront goes through all `iters` iterations, with or without a pair of its own, and they meet at a barrier
  // every kSyncIters of them: the samples neighbouring pairs share are then still in the caches when the next wavefront
  // asks for them (left alone the wavefronts drift apart and fetch them from memory again)
  for (uint64_t itn = 0; itn < iters;) {
    const uint64_t unit = ((uint64_t)vb * iters + itn) * kWaves + wave;
    if (mode == 0 && itn != 0 && itn % kSyncIters == 0) __syncthreads();
    if (unit >= unit_end) {
      ++itn;
      continue;
    }
    const uint64_t f0 = unit * 2;
    // the frames transformed are fb, fb + 1; an odd last frame is the second of the pair before it (frames before f0
    // are another wavefront's and are not stored)
    const uint64_t fb = f0 + 2 <= p.nframes ? f0 : p.nframes - 2;
    float re[32], im[32];
    {
      // mode 2: frame fb + 1 alone, through the same code as a pair (its samples come from the cache; nothing behind
      // them is read, the frame after it may lie beyond the signal)
      const bool second = REDO && mode == 2;
      const In* x0 = x + (p.frame0 + fb + (second ? 1 : 0)) * p.hop + lane;
      if constexpr (REDO) asm volatile("" : "+v"(x0));  // (opaque: or samples already loaded are held in registers instead)
      float loudest = 0.0f;  // TABLE: this lane's largest |sample|
      if constexpr (HOP64 > 0) {
        float raw[32 + HOP64];
#pragma unroll
        for (int a = 0; a < 32; ++a) raw[a] = ld<In>(x0, 64 * a);
        if (second) {
#pragma unroll
          for (int a = 32; a < 32 + HOP64; ++a) raw[a] = 0.0f;
        } else {
#pragma unroll
          for (int a = 32; a < 32 + HOP64; ++a) raw[a] = ld<In>(x0, 64 * a);
        }
        if constexpr (TABLE) loudest = max_abs_plain(raw);
#pragma unroll
        for (int a = 0; a < 32; ++a) {
          const float w = win_lds[64 * a + lane];
          re[brev5(a)] = raw[a] * w;
          im[brev5(a)] = raw[a + HOP64] * w;
        }
      } else {
        const In* x1 = x0 + p.hop;
        float v1[32];
        if (second) {
#pragma unroll
          for (int a = 0; a < 32; ++a) v1[a] = 0.0f;
        } else {
#pragma unroll
          for (int a = 0; a < 32; ++a) v1[a] = ld<In>(x1, 64 * a);
        }
#pragma unroll
        for (int a = 0; a < 32; ++a) {
          const float w = win_lds[64 * a + lane];
          const float v0 = ld<In>(x0, 64 * a);
          if constexpr (TABLE) loudest = __builtin_fmaxf(loudest, __builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1[a])));
          re[brev5(a)] = v0 * w;
          im[brev5(a)] = v1[a] * w;
        }
      }
      if constexpr (REDO) {
        // (a NaN sample does not compare as loud and needs no second look: it poisons every bin either way)
        if (mode == 0 && p.loud_thr_raw > 0.0f && __builtin_amdgcn_ballot_w64(loudest > p.loud_thr_raw) != 0) mode = 1;  // frame fb alone
        if (mode != 0) {
#pragma unroll
          for (int a = 0; a < 32; ++a) im[a] = 0.0f;
        }
      }
    }
    int lt = lane;
    // (TABLE: an opaque copy keeps the transform's loop-invariant LDS addresses out of registers it needs itself;
    // hoisted, they were spilled and every reload waited on the memory counter)
    if constexpr (TABLE) asm volatile("" : "+v"(lt));
    transform2048<TABLE>(re, im, tw_lds, tw_flat, plane, lt);

    // Epilogue in two halves of 512 bins.
    int le = lane;
    asm volatile("" : "+v"(le));  // (keeps loop-invariant epilogue addresses out of registers the transforms need)
    const int kappa = (le & 31) + ((le >> 5) ? 0 : 32);
    float best_a = -1.0f, best_b = -1.0f;  // FEAT: largest |X|^2 so far and its bin
    float next_a = -1.0f, next_b = -1.0f;  // FEAT: the runner-up (decides whether float32 can tell)
    int kbest_a = (int)p.bin_lo, kbest_b = (int)p.bin_lo;
    const uint64_t row_a = (p.col0 + fb) * p.out_bins, row_b = row_a + p.out_bins;
    const bool st_a = fb >= f0, st_b = true;  // fb + 1 < nframes always; frame fb belongs to this wavefront unless shifted
    // FEAT, band narrower than the spectrum: the largest |X|^2 of any bin of either frame (what the transform's
    // rounding noise and the pair's cross-talk scale with; see feature_uncertain)
    const bool banded = p.bin_lo != 0 || p.bin_hi != 1024;  // wave-uniform
    float seen_a = 0.0f, seen_b = 0.0f;
    // ALONE (REDO, mode 1 / 2): the transform held one frame; its spectrum is Z itself, taken twice so that the quantiser
    // sees the same factor 2 as the pair separation leaves; it goes to row_a (mode 1) or row_b (mode 2).
    auto epilogue = [&](auto alone_c) __attribute__((always_inline)) {
    constexpr bool ALONE = decltype(alone_c)::value;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      if constexpr (TABLE) epilogue_cells_table(P, re, im, cell, le); else epilogue_cells(P, re, im, cell, kappa);
      wave_sync();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j4 = 4 * (le + 64 * q);
        const int k = 512 * P + j4;
        const float4 z01 = *reinterpret_cast<const float4*>(cell + j4);
        const float4 z23 = *reinterpret_cast<const float4*>(cell + j4 + 2);
        const float4 q01 = ALONE ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(cell + kMirrorBase + j4);
        const float4 q23 = ALONE ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(cell + kMirrorBase + j4 + 2);
        uint32_t wa = 0, wb = 0;
        float pa[4] = {0, 0, 0, 0}, pb[4] = {0, 0, 0, 0};
        const bool in_band = k >= (int)p.bin_lo && k < (int)p.bin_hi;
#define SONO_BIN(B, ZR, ZI, PR, PI)                                          \
  {                                                                          \
    const float ar = ALONE ? (ZR) + (ZR) : (ZR) + (PR), ai = ALONE ? (ZI) + (ZI) : (ZI) - (PI); \
    const float br = (ZI) + (PI), bi = (PR) - (ZR);                          \
    const float ma = __builtin_fmaf(ar, ar, ai * ai);                        \
    const float mb = __builtin_fmaf(br, br, bi * bi);                        \
    if constexpr (FEAT) { /* a lane visits its bins in ascending order */    \
      pa[B] = ma;                                                            \
      pb[B] = mb;                                                            \
    } else {                                                                 \
      wa = put_index<PLUGIN, B>(ma, fs, wa);                                 \
      wb = put_index<PLUGIN, B>(mb, fs, wb);                                 \
    }                                                                        \
  }
        SONO_BIN(0, z01.x, z01.y, q01.x, q01.y)
        SONO_BIN(1, z01.z, z01.w, q01.z, q01.w)
        SONO_BIN(2, z23.x, z23.y, q23.x, q23.y)
        SONO_BIN(3, z23.z, z23.w, q23.z, q23.w)
#undef SONO_BIN
        if constexpr (FEAT) {
          if (banded) {
            seen_a = fmaxf(fmaxf(fmaxf(seen_a, pa[0]), pa[1]), fmaxf(pa[2], pa[3]));
            seen_b = fmaxf(fmaxf(fmaxf(seen_b, pb[0]), pb[1]), fmaxf(pb[2], pb[3]));
          }
          if (in_band) {  // (one branch per bin quad; selects inside)
#pragma unroll
            for (int B = 0; B < 4; ++B) {
              next_a = __builtin_amdgcn_fmed3f(pa[B], best_a, next_a);
              next_b = __builtin_amdgcn_fmed3f(pb[B], best_b, next_b);
              const bool ga = pa[B] > best_a, gb = pb[B] > best_b;
              best_a = ga ? pa[B] : best_a; kbest_a = ga ? k + B : kbest_a;
              best_b = gb ? pb[B] : best_b; kbest_b = gb ? k + B : kbest_b;
            }
          }
        }
        if (!FEAT && in_band) {
          const uint32_t col = k - p.out_bin0;
          if constexpr (ALONE) {
            if (mode == 1 ? st_a : st_b) *reinterpret_cast<uint32_t*>(out + (mode == 1 ? row_a : row_b) + col) = wa;
          } else {
            if (st_a) *reinterpret_cast<uint32_t*>(out + row_a + col) = wa;
            if (st_b) *reinterpret_cast<uint32_t*>(out + row_b + col) = wb;
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
      reduce_peak<64>(best_a, kbest_a, next_a);
      reduce_peak<64>(best_b, kbest_b, next_b);
      if (banded) {
        seen_a = reduce_max<64>(seen_a);
        seen_b = reduce_max<64>(seen_b);
      }
      const float top_a = fmaxf(best_a, seen_a), top_b = fmaxf(best_b, seen_b);  // the frames' largest |X|^2 anywhere
      if (lane == 0) {
        // reserved[0] = 1: float32 cannot tell; the exact-arithmetic pass that follows recomputes the record
        sono_frame_feature* ft = reinterpret_cast<sono_frame_feature*>(p.out) + (p.col0 + fb);
        const float ta = __builtin_fmaf(__log2f(best_a), fs.a, fs.b), tb = __builtin_fmaf(__log2f(best_b), fs.a, fs.b);
        if (st_a) {
          sono_frame_feature r{};
          r.peak_bin = kbest_a;
          r.peak_idx = (uint8_t)(PLUGIN ? fast_plugin_peak_index(best_a, fs) : put_index<false, 0>(best_a, fs, 0));
          r.reserved[0] = (feature_uncertain(best_a, next_a, top_b, top_a, ta, fs) || peak_outside(kbest_a, p.feat_lo, p.feat_hi)) ? 1 : 0;
          ft[0] = r;
        }
        sono_frame_feature r{};
        r.peak_bin = kbest_b;
        r.peak_idx = (uint8_t)(PLUGIN ? fast_plugin_peak_index(best_b, fs) : put_index<false, 0>(best_b, fs, 0));
        r.reserved[0] = (feature_uncertain(best_b, next_b, top_a, top_b, tb, fs) || peak_outside(kbest_b, p.feat_lo, p.feat_hi)) ? 1 : 0;
        ft[1] = r;
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

// RGBA: the image [row = row_of_bin0 - bin][column = col0 + frame] of LUT colours, with the slot mechanism of the
// 1024-point kernel (sono_rgba_flush.h).  A workgroup round is kRF = 128 consecutive frames = four iterations of 32
// frames (wavefront w: frames 32i + 2w, +1 of the round); a slot holds 1024 bins x 128 frames = 128 KiB as 32 reader
// units of 32 bins; two wavefronts fill one 16-byte cell (four bins x four frames) with 8 bytes each.  During the next
// round wavefront w writes out, in iteration i, quarter i of the reader units w and w + 16.  The samples of an iteration
// and the slot reads of its quarter are issued at the end of the iteration before.
#ifndef SONO_RF2048
#define SONO_RF2048 128
#endif
constexpr int kRF = SONO_RF2048;
constexpr int kRIters = kRF / 32;       // iterations per round
constexpr int kFPI = 8 / kRIters;       // quarters a wavefront writes out per iteration (32 units x 4 quarters per round)
constexpr int kSlotBytes = 1024 * kRF;

template <bool PLUGIN, typename In, int HOP64, bool TABLE>
__global__ __launch_bounds__(kWaves * 64) void stft2048_rgba_kernel(StftParams p, FastScale fs, const float2* __restrict__ tw_table,
                                                                    uint8_t* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw_lds = reinterpret_cast<float2*>(smem);
  float* win_lds = reinterpret_cast<float*>(smem + kTwBytes);
  const float2* __restrict__ tw_flat = reinterpret_cast<const float2*>(p.twiddles);
  for (int i = threadIdx.x; i < 2048; i += kWaves * 64) {
    if constexpr (TABLE) {
      const int r = i >> 5, k = i & 31;  // rows 0..30: stage T = 32t, factor 32j + k (r = t-1+j); rows 31..62: T = 1024, 32m + k
      if (r < 31) {
        const int t = 1 << (31 - __builtin_clz(r + 1)), j = r + 1 - t;
        tw_lds[i] = tw_flat[32 * t - 1 + 32 * j + k];
      } else if (r < 63) {
        tw_lds[i] = tw_flat[1023 + 32 * (r - 31) + k];
      }  // (the 31 factors of the first pass are read by scalar loads)
    } else {
      tw_lds[i] = tw_table[i];
    }
    win_lds[i] = p.window[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* fbuf = smem + kTwBytes + kWinBytes + wave * kBufBytes;
  float* plane = reinterpret_cast<float*>(fbuf);
  float2* cell = reinterpret_cast<float2*>(fbuf);
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  uint32_t* slots = reinterpret_cast<uint32_t*>(tiles + (size_t)blockIdx.x * 2 * kSlotBytes);
  // every workgroup runs the same number of rounds and every wavefront goes through every round (barriers)
  const uint64_t nrounds = (p.nframes + kRF - 1) / kRF;
  const uint64_t rounds = (nrounds + gridDim.x - 1) / gridDim.x;

  auto first_frame = [&](uint64_t it) -> uint64_t {  // of this wavefront in iteration `it` (>= nframes: none)
    const uint64_t round = it / kRIters;
    if (round >= rounds) return p.nframes;
    return (round * gridDim.x + vb) * kRF + 32 * (it % kRIters) + 2 * wave;
  };
  constexpr int kRaw = HOP64 > 0 ? 32 + HOP64 : 64;
  float raw[kRaw];
  auto fetch = [&](uint64_t f0) {  // frames fb, fb + 1 (shifted back by one at an odd end, and to the end when there is nothing to do)
    const uint64_t fb = f0 + 2 <= p.nframes ? f0 : p.nframes - 2;
    const In* x0 = x + (p.frame0 + fb) * p.hop + lane;
    if constexpr (HOP64 > 0) {
#pragma unroll
      for (int a = 0; a < kRaw; ++a) raw[a] = ld<In>(x0, 64 * a);
    } else {
      const In* x1 = x0 + p.hop;
#pragma unroll
      for (int a = 0; a < 32; ++a) {
        raw[a] = ld<In>(x0, 64 * a);
        raw[32 + a] = ld<In>(x1, 64 * a);
      }
    }
  };
  fetch(first_frame(0));
  // The colour table: the transforms use all of the CU's LDS, but a wavefront's own transpose buffer is idle while it
  // writes out its image quarter, so the table (1 KiB = 16 bytes per lane, kept in registers) is put there first and
  // looked up through LDS instead of through the vector memory path.
  const uint4 lut16 = reinterpret_cast<const uint4*>(p.lut)[lane];
  uint4 fq[kFPI][2] = {};
  // quarter j of iteration i: part and reader unit
  auto part_of = [](int i, int j) { return kFPI == 1 ? (i & 3) : i; };
  auto unit_of = [&](int i, int j) { return kFPI == 1 ? wave + 16 * (i >> 2) : wave + 16 * j; };
  // Under the context's table a pair too loud for the pair separation (StftParams::loud_thr_raw) is transformed one frame
  // at a time: the iteration's transform runs twice, mode 1 = the first frame alone, mode 2 = the second frame alone.
  int mode = 0;
  for (uint64_t it = 0; it < (rounds + 1) * kRIters;) {
    const uint64_t round = it / kRIters;
    const int i = (int)(it % kRIters);
    const uint64_t f0 = first_frame(it);
    uint32_t* slot = slots + (round & 1) * (kSlotBytes / 4);
    if (!(TABLE && mode == 2)) {
      // (the empty asm is a use of the loaded registers: the wait for them lands here, before the image stores)
#pragma unroll
      for (int a = 0; a < kRaw; ++a) asm volatile("" : "+v"(raw[a]));
      if (round > 0) {
        const uint64_t frame0 = ((round - 1) * gridDim.x + vb) * kRF;
        if (frame0 < p.nframes) {
          reinterpret_cast<uint4*>(fbuf)[lane] = lut16;
          wave_sync();
#pragma unroll
          for (int j = 0; j < kFPI; ++j)
            rgba_flush::flush_quarter_store<kRF>(p, fq[j], reinterpret_cast<const uint32_t*>(fbuf), frame0, part_of(i, j), unit_of(i, j), lane);
          wave_sync();  // (the transform rewrites the buffer)
        }
      }
    }
    if (f0 < p.nframes) {  // wave-uniform
      if constexpr (TABLE) {
        // (a NaN sample does not compare as loud and needs no second look: it poisons every bin either way)
        if (mode == 0 && p.loud_thr_raw > 0.0f && __builtin_amdgcn_ballot_w64(max_abs_plain(raw) > p.loud_thr_raw) != 0) mode = 1;
      }
      float re[32], im[32];
#pragma unroll
      for (int a = 0; a < 32; ++a) {
        const float w = win_lds[64 * a + lane];
        re[brev5(a)] = raw[a] * w;
        im[brev5(a)] = raw[a + (HOP64 > 0 ? HOP64 : 32)] * w;
      }
      if (TABLE && mode != 0) {
#pragma unroll
        for (int a = 0; a < 32; ++a) im[a] = 0.0f;
      }
      int lt = lane;
      if constexpr (TABLE) asm volatile("" : "+v"(lt));  // (as in the index kernel)
      transform2048<TABLE>(re, im, tw_lds, tw_flat, plane, lt);

      int le = lane;
      asm volatile("" : "+v"(le));  // (keeps loop-invariant epilogue addresses out of registers the transforms need)
      const int kappa = (le & 31) + ((le >> 5) ? 0 : 32);
      const bool shifted = f0 + 2 > p.nframes;  // odd end: frame f0 is the second of the pair transformed
      const int fg = 8 * i + (wave >> 1);
      // ALONE (mode 1 / 2): the transform held one frame; its spectrum is Z itself, taken twice so that the quantiser
      // sees the same factor 2 as the pair separation leaves; it goes to the first (mode 1) or second (mode 2) word.
      auto epilogue = [&](auto alone_c) __attribute__((always_inline)) {
      constexpr bool ALONE = decltype(alone_c)::value;
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        if constexpr (TABLE) epilogue_cells_table(P, re, im, cell, le); else epilogue_cells(P, re, im, cell, kappa);
        wave_sync();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int j4 = 4 * (le + 64 * q);
          const float4 z01 = *reinterpret_cast<const float4*>(cell + j4);
          const float4 z23 = *reinterpret_cast<const float4*>(cell + j4 + 2);
          const float4 q01 = ALONE ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(cell + kMirrorBase + j4);
          const float4 q23 = ALONE ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(cell + kMirrorBase + j4 + 2);
          uint32_t wa = 0, wb = 0;
#define SONO_BIN(B, ZR, ZI, PR, PI)                                          \
  {                                                                          \
    const float ar = ALONE ? (ZR) + (ZR) : (ZR) + (PR), ai = ALONE ? (ZI) + (ZI) : (ZI) - (PI); \
    const float br = (ZI) + (PI), bi = (PR) - (ZR);                          \
    wa = put_index<PLUGIN, B>(__builtin_fmaf(ar, ar, ai * ai), fs, wa);      \
    if constexpr (!ALONE) wb = put_index<PLUGIN, B>(__builtin_fmaf(br, br, bi * bi), fs, wb); \
  }
          SONO_BIN(0, z01.x, z01.y, q01.x, q01.y)
          SONO_BIN(1, z01.z, z01.w, q01.z, q01.w)
          SONO_BIN(2, z23.x, z23.y, q23.x, q23.y)
          SONO_BIN(3, z23.z, z23.w, q23.z, q23.w)
#undef SONO_BIN
          const int kq = 128 * P + le + 64 * q;
          if (4 * kq + 3 >= (int)p.bin_lo && 4 * kq < (int)p.bin_hi) {
            uint32_t* words = slot + (rgba_flush::slot_cell<kRF>(kq, fg) * 2 + (wave & 1)) * 2;
            if constexpr (ALONE) {
              if (mode == 2) words[1] = wa;
              if ((mode == 1) != shifted) words[0] = wa;  // (odd end: the second frame of the pair is the one that shows)
            } else {
              *reinterpret_cast<uint2*>(words) = make_uint2(shifted ? wb : wa, wb);
            }
          }
        }
        wave_sync();  // the buffer is rewritten by the next half / the next iteration's transpose
      }
      };  // epilogue
      if (TABLE && mode != 0) epilogue(std::true_type{});
      else epilogue(std::false_type{});
    }
    if constexpr (TABLE) {
      if (mode == 1) {
        // the first frame alone is done: the same pair once more for the second frame alone -- its samples (from the
        // cache) and nothing behind them, the frame after it may lie beyond the signal
        const uint64_t fb = f0 + 2 <= p.nframes ? f0 : p.nframes - 2;
        const In* x1 = x + (p.frame0 + fb + 1) * p.hop + lane;
        asm volatile("" : "+v"(x1));  // (opaque: or the samples already loaded for the pair are kept in registers instead)
#pragma unroll
        for (int a = 0; a < 32; ++a) raw[a] = ld<In>(x1, 64 * a);
#pragma unroll
        for (int a = 32; a < kRaw; ++a) raw[a] = 0.0f;
        mode = 2;
        continue;
      }
      mode = 0;
    }
    fetch(first_frame(it + 1));
    // this round's slot is complete and everyone is done reading the other one.  (The flavour with exact roots meets after
    // every iteration: measured 5 % faster, its wavefronts share more of their samples through the caches that way; under a
    // table, where a loud pair makes its wavefront late, 1 % slower.)
    if (i == kRIters - 1 || !TABLE) __syncthreads();
    const uint64_t nround = (it + 1) / kRIters;
    if (nround > 0 && nround <= rounds) {
      const int ni = (int)((it + 1) % kRIters);
#pragma unroll
      for (int j = 0; j < kFPI; ++j)
        rgba_flush::flush_quarter_load<kRF>(fq[j], slots + ((nround - 1) & 1) * (kSlotBytes / 4), part_of(ni, j), unit_of(ni, j), lane);
    }
    ++it;
  }
}

}  // namespace fast2048

// W_2048^(l*k1) as [k1][l] float2 (cos, -sin), k1 < 32, l < 64, computed in double.
void build_fast_table_2048(std::vector<float>& t) {
  t.resize(2 * 2048);
  const double pi = 3.141592653589793;
  for (int k1 = 0; k1 < 32; ++k1)
    for (int l = 0; l < 64; ++l) {
      const double ang = -2.0 * pi * (double)(k1 * l) / 2048.0;
      t[2 * (k1 * 64 + l)] = (float)std::cos(ang);
      t[2 * (k1 * 64 + l) + 1] = (float)std::sin(ang);
    }
}

bool fast2048_available(const StftParams& p) {
  if (p.n != 2048 || p.nframes < 2) return false;
  if (!((p.fast_mode == 1 && p.fast_table) || (p.fast_mode == 2 && p.twiddles))) return false;
  if ((p.bin_lo & 3) || (p.bin_hi & 3) || p.bin_lo >= p.bin_hi) return false;
  if (p.out_kind == SONO_OUT_FEATURES) return true;
  if (p.out_kind != SONO_OUT_U8_INDEX || !p.pair_idx_ok) return false;
  return !(p.out_bins & 3) && !(p.out_bin0 & 3) && !(reinterpret_cast<uintptr_t>(p.out) & 3);
}

// The image output needs the flipped full-height row mapping, a colour table and the scratch
// buffer (fast_rgba_scratch_bytes()).
bool fast2048_rgba_available(const StftParams& p) {
  if (p.n != 2048 || p.nframes < 2 || !p.pair_idx_ok) return false;
  if (!((p.fast_mode == 1 && p.fast_table) || (p.fast_mode == 2 && p.twiddles))) return false;
  if (p.bin_lo >= p.bin_hi) return false;
  if (p.row_of_bin0 != (int32_t)p.bin_hi - 1) return false;
  return p.width != 0 && p.width <= ((uint64_t)1 << 24) && p.lut;
}

// Returns 1 if this launch is not served here.
int launch_stft_fast2048(const StftParams& p, void* stream_v) {
  if ((p.out_kind == SONO_OUT_RGBA8 || p.out_kind == SONO_OUT_U8_INDEX) && !p.pair_idx_ok) return 1;  // two frames per transform: see StftParams::pair_idx_ok
  using namespace fast2048;
  const bool rgba = p.out_kind == SONO_OUT_RGBA8;
  if (rgba ? !(fast2048_rgba_available(p) && p.tile_scratch) : !fast2048_available(p)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  FastScale fs = make_fast_scale(p.scale_mode, p.n, p.gain_db, p.range_db);
  fs.b -= 2.0f * fs.a;  // the pair separation leaves 2*X: |2X|^2 = 4|X|^2
  fs.xtalk = (float)p.pair_asym;
  const int cus = device_cu_count();
  const uint64_t nunits = (p.nframes + 1) / 2;
  const uint64_t need = (nunits + kWaves - 1) / kWaves;
  const uint32_t grid = (uint32_t)(need < (uint64_t)cus ? need : (uint64_t)cus);  // persistent: one workgroup per CU
  const float2* table = reinterpret_cast<const float2*>(p.fast_table);
  const bool plugin = p.scale_mode == SONO_SCALE_PLUGIN_DB, i16 = p.in_kind == SONO_IN_I16, feat = p.out_kind == SONO_OUT_FEATURES;
  const bool tab = p.fast_mode == 2;
  if (rgba) {
    const uint64_t nr = (p.nframes + kRF - 1) / kRF;
    const uint32_t rgrid = (uint32_t)(nr < (uint64_t)cus ? nr : (uint64_t)cus);
    uint8_t* tiles = reinterpret_cast<uint8_t*>(p.tile_scratch);
#define SONO_R2048(PL, In, H)                                                                                   \
  do {                                                                                                          \
    if (tab) SONO_R2048_T(PL, In, H, true); else SONO_R2048_T(PL, In, H, false);                                \
  } while (0)
#define SONO_R2048_T(PL, In, H, T)                                                                              \
  do {                                                                                                          \
    auto kern = stft2048_rgba_kernel<PL, In, H, T>;                                                             \
    SONO_GRANT_LDS(kern, kLdsBytes);                                                                            \
    hipLaunchKernelGGL(kern, dim3(rgrid), dim3(kWaves * 64), kLdsBytes, stream, p, fs, table, tiles);           \
  } while (0)
#define SONO_R2048_H(PL, In)                                                \
  do {                                                                      \
    if (p.hop == 512) SONO_R2048(PL, In, 8); else SONO_R2048(PL, In, 0);    \
  } while (0)
    if (plugin) { if (i16) SONO_R2048_H(true, int16_t); else SONO_R2048_H(true, float); }
    else { if (i16) SONO_R2048_H(false, int16_t); else SONO_R2048_H(false, float); }
#undef SONO_R2048_H
#undef SONO_R2048_T
#undef SONO_R2048
    // (redo: the kernel transforms its loud pairs once more, one frame at a time)
    note_kernel(tab ? (p.loud_thr_raw > 0.0f ? "stft2048_rgba_kernel<table,redo>" : "stft2048_rgba_kernel<table>") : "stft2048_rgba_kernel<roots>");
    return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
  }
#define SONO_L2048(PL, In, FE, H)                                                                               \
  do {                                                                                                          \
    if (tab) SONO_L2048_T(PL, In, FE, H, true); else SONO_L2048_T(PL, In, FE, H, false);                        \
  } while (0)
#define SONO_L2048_T(PL, In, FE, H, T)                                                                          \
  do {                                                                                                          \
    auto kern = stft2048_kernel<PL, In, FE, H, T>;                                                               \
    SONO_GRANT_LDS(kern, kLdsBytes);                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), kLdsBytes, stream, p, fs, table);                   \
  } while (0)
#define SONO_L2048_F(PL, In)                                                                                   \
  do {                                                                                                         \
    if (p.hop == 512) { if (feat) SONO_L2048(PL, In, true, 8); else SONO_L2048(PL, In, false, 8); }             \
    else { if (feat) SONO_L2048(PL, In, true, 0); else SONO_L2048(PL, In, false, 0); }                          \
  } while (0)
  if (plugin) { if (i16) SONO_L2048_F(true, int16_t); else SONO_L2048_F(true, float); }
  else { if (i16) SONO_L2048_F(false, int16_t); else SONO_L2048_F(false, float); }
#undef SONO_L2048_F
#undef SONO_L2048_T
#undef SONO_L2048
  note_kernel(tab ? (p.out_kind == SONO_OUT_U8_INDEX && p.loud_thr_raw > 0.0f ? "stft2048_kernel<table,redo>" : "stft2048_kernel<table>")
                  : "stft2048_kernel<roots>");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
