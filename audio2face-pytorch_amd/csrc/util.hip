// Measurement helpers of the C ABI (no model arithmetic): a float4 streaming copy and a 2-reads-1-write triad, used by
// bench.py to measure, in the same job as the step timings, the HBM rate this box gives plain streaming kernels.
// Shape: ONE float4 per thread, one 256-thread workgroup per 4 KB (tools/streambench.hip ranked the shapes on this pool: 6.25 TB/s copy /
// 6.07 TB/s triad - the guide's 6.29 - against 4.9 / 5.2 for the persistent grid-stride form round 2 used as its yardstick and 5.5 / 5.7
// for persistent workgroups that each stream one contiguous chunk: with workgroups dispatched in address order the bytes in flight form
// one narrow moving window of DRAM pages, which persistent workgroups that drift apart do not).
#include "a2f_common.h"

namespace a2f {

__global__ __launch_bounds__(256) void stream_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n4) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n4) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void stream_triad_kernel(float4* __restrict__ dst, const float4* __restrict__ a,
                                                           const float4* __restrict__ b, int64_t n4) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n4) {
    const float4 x = a[i], y = b[i];
    dst[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

}  // namespace a2f

extern "C" int a2f_stream_copy(float* dst, const float* src, int64_t n, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dst && src && n > 0 && n % 4 == 0, "stream_copy: n=%lld must be a positive multiple of 4", (long long)n);
  A2F_CHECK_ARG((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "stream_copy: buffers must be 16-byte aligned");
  A2F_CHECK_ARG(n / 4 <= 256ll * 0x7FFFFFFF, "stream_copy: n too large for one float4 per thread");
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)dst, (const float4*)src, n / 4);
  A2F_CHECK_LAUNCH("stream_copy");
  return A2F_OK;
}

extern "C" int a2f_stream_triad(float* dst, const float* a, const float* b, int64_t n, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dst && a && b && n > 0 && n % 4 == 0, "stream_triad: n=%lld must be a positive multiple of 4", (long long)n);
  A2F_CHECK_ARG((((uintptr_t)dst | (uintptr_t)a | (uintptr_t)b) & 15) == 0, "stream_triad: buffers must be 16-byte aligned");
  A2F_CHECK_ARG(n / 4 <= 256ll * 0x7FFFFFFF, "stream_triad: n too large for one float4 per thread");
  hipLaunchKernelGGL(stream_triad_kernel, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)dst, (const float4*)a,
                     (const float4*)b, n / 4);
  A2F_CHECK_LAUNCH("stream_triad");
  return A2F_OK;
}
