// socmx_launch.h -- host-side launch helpers shared by the C-ABI translation units.
//
//  * launch(): hipLaunchKernel with its OWN status.  (hipLaunchKernelGGL + hipGetLastError reports whatever error was
//    pending in the process -- an earlier failure of unrelated asynchronous work would be blamed on this call.)
//  * ensure_max_lds(): hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KiB) once per (kernel, device) instead of
//    on every call; the only mutable state of the library is this per-process cache (mutex-protected).
//  * env_flag(): developer A/B switches are read once per process.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace socmx {

template <typename T>
struct same_as { typedef T type; };

template <typename... KArgs>
static inline int launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds_bytes, void* stream,
                         typename same_as<KArgs>::type... args) {
  void* ptrs[] = {(void*)&args...};
  return (int)hipLaunchKernel((const void*)kern, grid, block, ptrs, lds_bytes, (hipStream_t)stream);
}

static const int kLdsBytesPerCU = 160 * 1024;

static inline int ensure_max_lds(const void* kern) {
  static std::mutex mu;
  static std::unordered_map<const void*, uint64_t> done;   // kernel -> bit per device already configured
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) return (int)err;
  const uint64_t bit = 1ull << (dev & 63);
  std::lock_guard<std::mutex> lock(mu);
  uint64_t& mask = done[kern];
  if (mask & bit) return 0;
  err = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesPerCU);
  if (err != hipSuccess) return (int)err;
  mask |= bit;
  return 0;
}

template <typename... KArgs>
static inline int ensure_max_lds(void (*kern)(KArgs...)) { return ensure_max_lds((const void*)kern); }

}  // namespace socmx
