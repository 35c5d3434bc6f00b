// Measurement helpers of the C ABI (no model arithmetic): a float4 streaming copy and a 2-reads-1-write triad, used by
// bench.py to measure, in the same job as the step timings, the HBM rate this box gives plain streaming kernels.
#include "a2f_common.h"

namespace a2f {

__global__ __launch_bounds__(256) void stream_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void stream_triad_kernel(float4* __restrict__ dst, const float4* __restrict__ a,
                                                           const float4* __restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 x = a[i], y = b[i];
    dst[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

}  // namespace a2f

extern "C" int a2f_stream_copy(float* dst, const float* src, int64_t n, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dst && src && n > 0 && n % 4 == 0, "stream_copy: n=%lld must be a positive multiple of 4", (long long)n);
  A2F_CHECK_ARG((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "stream_copy: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (float4*)dst, (const float4*)src, n / 4);
  A2F_CHECK_LAUNCH("stream_copy");
  return A2F_OK;
}

extern "C" int a2f_stream_triad(float* dst, const float* a, const float* b, int64_t n, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dst && a && b && n > 0 && n % 4 == 0, "stream_triad: n=%lld must be a positive multiple of 4", (long long)n);
  A2F_CHECK_ARG((((uintptr_t)dst | (uintptr_t)a | (uintptr_t)b) & 15) == 0, "stream_triad: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(stream_triad_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (float4*)dst, (const float4*)a,
                     (const float4*)b, n / 4);
  A2F_CHECK_LAUNCH("stream_triad");
  return A2F_OK;
}
