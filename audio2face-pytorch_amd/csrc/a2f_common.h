// Shared helpers for the gfx950 kernels of liba2f_hip.so (CDNA4: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/a2f_hip.h"

namespace a2f {

constexpr int kWave = 64;

// thread-local last-error text (C ABI: a2f_last_error)
char* err_buf();
int set_err(int code, const char* fmt, ...);

#define A2F_CHECK_ARG(cond, ...)                              \
  do {                                                        \
    if (!(cond)) return a2f::set_err(A2F_EINVAL, __VA_ARGS__); \
  } while (0)

#define A2F_CHECK_LAUNCH(what)                                                          \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) return a2f::set_err(A2F_ELAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
  } while (0)

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// wave64 butterfly reductions (every lane ends with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace a2f
