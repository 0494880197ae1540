// In-register 32-point transforms and small device helpers shared by the 1024-point float32 kernels.
#ifndef SONO_FFT32_H
#define SONO_FFT32_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sono_device.h"

namespace sono {
namespace fast1024 {

// cos/sin(2*pi*j/32), j = 0..15
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

// In-register 32-point forward DFT, radix-2 decimation in time.  Input in bit-reversed register
// order, output in natural order.  All indices fold to constants after unrolling.
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

// The same 32-point network with twiddles from a table (the plugin's recurrence factors or a host
// table: not exact roots of unity, so every butterfly must use the very factor the reference uses).
// tw(half, j) returns the factor (cos, sin) of the butterfly at position j of a stage with half size
// `half` (sin < 0 for the forward transform, spectrogram.esm.js:176-190).  FIRST: these are the first
// five stages of the 1024-point transform, where j == 0 is the factor (1, 0) exactly in every table
// this kernel accepts (checked on the host); in the last five stages every butterfly has a factor.
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

// Lane-wise select between two registers by a wave-uniform 64-bit lane mask: `if_set` where the lane's bit is 1.  (Written
// as the instruction itself: given `cond ? a[i] : a[j]` over an unrolled loop the compiler forms an indexed register
// read instead, and copies made opaque to prevent that cost a move each.)
__device__ __forceinline__ float lane_select(float if_clear, float if_set, uint64_t mask) {
  float o;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(if_clear), "v"(if_set), "s"(mask));
  return o;
}

// Orders this wavefront's LDS traffic for the compiler.  The hardware executes one wavefront's LDS
// instructions in issue order, so data written by one lane is visible to a later read by another
// lane of the same wavefront without a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Transposing 32 registers x 32 lanes without a 32 x 32 LDS plane: a block of 16 x 16 floats per 16-lane row.
constexpr int kBlkFloats = 16 * 16 + 16;      // a block and 16 floats of bank offset to the next; a wavefront needs four (4.25 KiB)

// Transpose of one plane across the 32 lanes of a transform: lane l, register k1 -> lane k1, register l, the result
// in the bit-reversed register order the second pass wants.  `blk` is the lane's 16-lane row block, `ll` = lane & 15.
// Element (register r, lane ll) of a block sits at float 16 r + (ll ^ 4 (r >> 2)): the 16 lanes of a store cover 16
// consecutive banks and the neighbouring block (16 floats further on) the other 16, and the four 16-byte chunks a lane
// reads from its row ll are the chunks c ^ (ll >> 2), which differ between the rows that share banks.
__device__ __forceinline__ void transpose32_plane(float (&v)[32], float* __restrict__ blk, int ll) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {  // register bit 4 <-> lane bit 4: rows 1 / 3 of v[r] trade places with rows 0 / 2 of v[r + 16]
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r]), __float_as_uint(v[r + 16]), false, false);
    v[r] = __uint_as_float(s[0]);
    v[r + 16] = __uint_as_float(s[1]);
  }
  float t[32];
  asm volatile("" : "+v"(ll));  // (the swizzled addresses are computed here, not kept in registers round the frame loop)
#pragma unroll
  for (int half = 0; half < 2; ++half) {  // 16 x 16 within the 16-lane row, one half of the registers at a time
#pragma unroll
    for (int r = 0; r < 16; ++r) blk[r * 16 + (ll ^ (4 * (r >> 2)))] = v[16 * half + r];
    wave_sync();
    const float4* row = reinterpret_cast<const float4*>(blk + ll * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 x = row[j ^ (ll >> 2)];
      t[16 * half + 4 * j] = x.x; t[16 * half + 4 * j + 1] = x.y; t[16 * half + 4 * j + 2] = x.z; t[16 * half + 4 * j + 3] = x.w;
    }
    wave_sync();
  }
#pragma unroll
  for (int m = 0; m < 32; ++m) v[brev5(m)] = t[m];
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

// quantise |X|^2 (already including the pair-separation factor through fs.b) to a byte of `acc`
template <bool PLUGIN, int BYTE>
__device__ __forceinline__ uint32_t put_index(float msq, const FastScale& fs, uint32_t acc) {
  if constexpr (PLUGIN) {
    return fast_plugin_byte<BYTE>(msq, fs, acc);
  } else {
    return fast_worker_byte<BYTE>(msq, fs, acc);
  }
}

}  // namespace fast1024
}  // namespace sono
#endif
