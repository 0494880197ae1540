// Host-side launch helpers shared by the kernel translation units: per-device facts and per-kernel attributes
// are looked up once, not on every launch (a 5 s file is ~10 us of kernel time; hipGetDeviceProperties and
// hipFuncSetAttribute cost more than that).
#ifndef SONO_LAUNCH_HELPERS_H
#define SONO_LAUNCH_HELPERS_H

#include <hip/hip_runtime.h>

#include <atomic>

#include "sonospec.h"

namespace sono {

constexpr int kMaxDevices = 64;

// Compute units of the current device (256 on MI355X), cached per device.
inline int device_cu_count() {
  static std::atomic<int> cached[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  cached[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

// Largest dynamic-LDS size already granted to one kernel, per device.
struct LdsGrant {
  std::atomic<int> bytes[kMaxDevices];
};

// hipFuncAttributeMaxDynamicSharedMemorySize for `kern` on the current device, only when `bytes` exceeds
// what an earlier launch asked for.
inline int grant_lds(LdsGrant& g, const void* kern, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return SONO_ERR_DEVICE;
  if (g.bytes[dev].load(std::memory_order_relaxed) >= bytes) return SONO_OK;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return SONO_ERR_DEVICE;
  g.bytes[dev].store(bytes, std::memory_order_relaxed);
  return SONO_OK;
}

}  // namespace sono

// One grant record per expansion site, i.e. per kernel instantiation.
#define SONO_GRANT_LDS(kern, bytes)                                                                  \
  do {                                                                                               \
    static ::sono::LdsGrant grant_;                                                                  \
    if (::sono::grant_lds(grant_, reinterpret_cast<const void*>(kern), (int)(bytes)) != SONO_OK)     \
      return SONO_ERR_DEVICE;                                                                        \
  } while (0)

#endif
