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

// Orders this wavefront's LDS traffic for the compiler.  The hardware executes one wavefront's LDS
// instructions in issue order, so data written by one lane is visible to a later read by another
// lane of the same wavefront without a workgroup barrier.
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

// quantise |X|^2 (already including the pair-separation factor through fs.b) to a byte of `acc`
template <bool PLUGIN, int BYTE>
__device__ __forceinline__ uint32_t put_index(float msq, const FastScale& fs, uint32_t acc) {
  if constexpr (PLUGIN) {
    return fast_plugin_byte<BYTE>(msq, fs, acc);
  } else {
    const float t = __builtin_floorf(__builtin_fmaf(__log2f(msq), fs.a, fs.b));
    return __builtin_amdgcn_cvt_pk_u8_f32(t, BYTE, acc);  // saturates to [0,255]; NaN -> 0
  }
}

}  // namespace fast1024
}  // namespace sono
#endif
