// Marking and finding the loud units of StftParams::loud_flags (see there).
#ifndef SONO_LOUD_H
#define SONO_LOUD_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sono_internal.h"

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

// First launch: `loud` is this lane's verdict on the samples it holds; if any lane of the wavefront says so, the groups
// of four frames that [frame, frame + count) touches are marked (absolute frame numbers of the call).  Wave-uniform
// arguments; a NaN sample does not compare as loud and needs no second look (it poisons every bin in either kernel).
__device__ __forceinline__ void mark_loud(const StftParams& p, bool loud, uint64_t frame, uint32_t count) {
  if (__builtin_amdgcn_ballot_w64(loud) == 0) return;
  if ((threadIdx.x & 63) == 0) {
    const uint64_t g0 = frame >> 2, g1 = (frame + count - 1) >> 2;
    for (uint64_t g = g0; g <= g1; ++g) p.loud_flags[g & p.loud_mask] = 1;
  }
}

// Second launch: is any frame of [frame, frame + count) in a marked group?
__device__ __forceinline__ bool is_loud(const StftParams& p, uint64_t frame, uint32_t count) {
  const uint64_t g0 = frame >> 2, g1 = (frame + count - 1) >> 2;
  bool any = false;
  for (uint64_t g = g0; g <= g1; ++g) any = any || p.loud_flags[g & p.loud_mask] != 0;
  return any;
}

}  // namespace sono
#endif
