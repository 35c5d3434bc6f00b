// Fused Adam step over one flat fp32 parameter buffer (torch.optim.Adam as configured at
// /root/reference/src/model/lightning_model.py:209-213: lr, weight_decay = lr/10 added to the gradient).
// Pure streaming: 16 B/element read (p, g, m, v) + 12 B written; float4 per lane, grid-stride.
#include "a2f_common.h"

namespace a2f {

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_c1, float b1, float b2,
                                         float inv_sqrt_c2, float eps, float wd, float gs) {
  g = g * gs + wd * p;
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_c2 + eps;
  p -= lr_c1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr_c1, float b1, float b2,
                                                   float inv_sqrt_c2, float eps, float wd, float gs) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, lr_c1, b1, b2, inv_sqrt_c2, eps, wd, gs);
    adam_one(pp.y, gg.y, mm.y, vv.y, lr_c1, b1, b2, inv_sqrt_c2, eps, wd, gs);
    adam_one(pp.z, gg.z, mm.z, vv.z, lr_c1, b1, b2, inv_sqrt_c2, eps, wd, gs);
    adam_one(pp.w, gg.w, mm.w, vv.w, lr_c1, b1, b2, inv_sqrt_c2, eps, wd, gs);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // scalar tail
  const int64_t tail = n4 << 2;
  const int64_t i = tail + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) adam_one(p[i], g[i], m[i], v[i], lr_c1, b1, b2, inv_sqrt_c2, eps, wd, gs);
}

}  // namespace a2f

extern "C" int a2f_adam_step(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float grad_scale, int32_t step, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(p && grad && m && v, "adam_step: null pointer");
  A2F_CHECK_ARG(n > 0 && step >= 1, "adam_step: n=%lld step=%d", (long long)n, step);
  A2F_CHECK_ARG((((uintptr_t)p | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  const double c1 = 1.0 - pow((double)beta1, (double)step);
  const double c2 = 1.0 - pow((double)beta2, (double)step);
  int64_t blocks = cdiv(cdiv(n, 4), 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, n, (float)(lr / c1), beta1,
                     beta2, (float)(1.0 / sqrt(c2)), eps, weight_decay, grad_scale);
  A2F_CHECK_LAUNCH("adam_step");
  return A2F_OK;
}
