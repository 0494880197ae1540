// The loudness test of the kernels that pair frames under an asymmetric twiddle table (StftParams::loud_thr_raw).
#ifndef SONO_LOUD_H
#define SONO_LOUD_H

#include <hip/hip_runtime.h>

namespace sono {

// max(|a|, |b|, m) in one instruction (from fmaxf / fabsf the compiler builds three or four).
__device__ __forceinline__ float max3_abs(float a, float b, float m) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
  return r;
}

// Largest |v[i]| of a lane's K values (a NaN is ignored, as by fmaxf).
template <int K>
__device__ __forceinline__ float max_abs(const float (&v)[K]) {
  float m0 = 0.0f, m1 = 0.0f;  // two chains
#pragma unroll
  for (int i = 0; i + 3 < K; i += 4) {
    m0 = max3_abs(v[i], v[i + 1], m0);
    m1 = max3_abs(v[i + 2], v[i + 3], m1);
  }
#pragma unroll
  for (int i = K & ~3; i < K; ++i) m0 = max3_abs(v[i], v[i], m0);
  return __builtin_fmaxf(m0, m1);
}

// The same from fmaxf / fabsf, for the compiler to schedule as it likes (the 2048-point kernels run 2-3 % faster with
// this form, the others 1-4 % faster with the one above; measured on one box, interleaved runs).
template <int K>
__device__ __forceinline__ float max_abs_plain(const float (&v)[K]) {
  float m = 0.0f;
#pragma unroll
  for (int i = 0; i + 1 < K; i += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[i]), __builtin_fabsf(v[i + 1])));
  if (K & 1) m = __builtin_fmaxf(m, __builtin_fabsf(v[K - 1]));
  return m;
}

}  // namespace sono
#endif
