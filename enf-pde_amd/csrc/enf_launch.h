// enf_launch.h -- host-side launch helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds per DEVICE: `done` keeps one bit per device ordinal of the calling
// thread's current device (ordinals beyond 63 set the attribute on every launch).  Safe from concurrent host threads:
// setting the attribute twice is harmless, the bit is set only after it succeeded.
typedef std::atomic<unsigned long long> EnfAttrBits;
inline bool enf_lds_attr(const void* kern, int bytes, EnfAttrBits& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return true;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
  return true;
}
