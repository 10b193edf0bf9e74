// Shared host/device helpers for libmused_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define MUSED_OK 0
#define MUSED_ERR_ARG -1
#define MUSED_ERR_HIP -2
#define MUSED_ERR_STATE -3
#define MUSED_ERR_UNSUPPORTED -4

// dtype codes shared with include/mused_hip.h
#define MUSED_F32 0
#define MUSED_F64 1
#define MUSED_I64 2
#define MUSED_BITS 3

namespace mused {

void set_error(const char* fmt, ...);

#define MUSED_CHECK_HIP(expr)                                                          \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ::mused::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return MUSED_ERR_HIP;                                                            \
    }                                                                                  \
  } while (0)

#define MUSED_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::mused::set_error(__VA_ARGS__);      \
      return MUSED_ERR_ARG;                 \
    }                                       \
  } while (0)

#define MUSED_LAUNCH_CHECK() MUSED_CHECK_HIP(hipGetLastError())

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// order-preserving map double -> uint64 (ascending)
__device__ __forceinline__ unsigned long long f64_key(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace mused
