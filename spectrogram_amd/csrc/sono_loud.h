// The loudness test of the kernels that pair frames under an asymmetric twiddle table (StftParams::loud_thr_raw).
#ifndef SONO_LOUD_H
#define SONO_LOUD_H

#include <hip/hip_runtime.h>

namespace sono {

// Largest |v[i]| of a lane's K values.
template <int K>
__device__ __forceinline__ float max_abs(const float (&v)[K]) {
  float m = 0.0f;
#pragma unroll
  for (int i = 0; i + 1 < K; i += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[i]), __builtin_fabsf(v[i + 1])));
  if (K & 1) m = __builtin_fmaxf(m, __builtin_fabsf(v[K - 1]));
  return m;
}

}  // namespace sono
#endif
