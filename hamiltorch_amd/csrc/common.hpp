// Shared host/device helpers for libhamiltorch_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/hamiltorch_amd.h"
#include "mh_rules.hpp"

namespace hta {

void set_error(const char* fmt, ...);
// names the kernel a C-ABI call dispatched to (read back by hta_last_route(); tests assert the route they claim to test)
void note_route(const char* fmt, ...);

#define HTA_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) {                              \
      ::hta::set_error(__VA_ARGS__);            \
      return HTA_ERR_INVALID;                   \
    }                                           \
  } while (0)

#define HTA_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      ::hta::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return HTA_ERR_LAUNCH;                                                          \
    }                                                                                 \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that samples on a second GPU
// must opt the kernel in there as well.  A DevOnce is a "done" flag per device of the calling thread's current device.
struct DevOnce {
  bool f[64] = {};
  static int dev() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : 63;
  }
  bool operator!() const { return !f[dev()]; }
  DevOnce& operator=(bool v) { f[dev()] = v; return *this; }
};

// ---- wave64 reductions (DPP via __shfl_xor butterflies) --------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over aligned groups of G lanes (G power of two <= 64); every lane of a group gets the total
template <int G, typename T> __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace hta
