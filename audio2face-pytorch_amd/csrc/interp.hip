// Linear interpolation along time, align_corners = True, of channels-last feature rows (N, L_in, C) -> (N, L_out, C):
// the 49 Hz -> frame-rate resampling of the wav2vec2 conv features in front of the FaceFormer decoder
// (/root/reference/src/model/wav2vec.py:76-84 `linear_interpolation`, called at :124-128).
//
// torch's F.interpolate(mode="linear") on the transposed (N, C, L) view costs 2.3 ms forward and 4.6 ms backward
// (atomics) per 16-sequence FaceFormer step on MI355X; both directions are pure streaming here: the forward reads two
// input rows per output row, the backward is the GATHER form (input row i sums the few output rows whose i0 or i1 is i,
// found by re-evaluating the forward's own index arithmetic, so the two directions agree bit for bit on indices and
// weights and the sum order is fixed: deterministic, no atomics).
//
// Index arithmetic = at::native::area_pixel_compute_source_index (align_corners): scale = (L_in - 1) / (L_out - 1) in
// fp32, src = scale * t, i0 = (int)src, i1 = i0 + (i0 < L_in - 1), w1 = src - i0, w0 = 1 - w1.
#include "a2f_common.h"

namespace a2f {

struct InterpArgs {
  const float* src; float* dst;
  int N, Lin, Lout, C4;                // C4 = C / 4 (float4 columns)
  float scale, inv_scale;
};

__device__ __forceinline__ void interp_index(float scale, int Lin, int t, int& i0, int& i1, float& w0, float& w1) {
  float s = scale * (float)t;
  asm volatile("" : "+v"(s));          // keep the ROUNDED product the reference computes: contracted into s - i0 (fma) it moves w1 by up to an ulp of s (1e-5 of the output at L = 600)
  i0 = (int)s;
  if (i0 > Lin - 1) i0 = Lin - 1;
  i1 = i0 + (i0 < Lin - 1 ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.f - w1;
}

__global__ __launch_bounds__(256) void interp_fwd_kernel(const InterpArgs a) {
  const int64_t total = (int64_t)a.N * a.Lout * a.C4;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % a.C4);
    const int64_t r = e / a.C4;
    const int t = (int)(r % a.Lout), n = (int)(r / a.Lout);
    int i0, i1; float w0, w1;
    interp_index(a.scale, a.Lin, t, i0, i1, w0, w1);
    const f32x4* s = reinterpret_cast<const f32x4*>(a.src) + (int64_t)n * a.Lin * a.C4 + c;
    const f32x4 x0 = s[(int64_t)i0 * a.C4], x1 = s[(int64_t)i1 * a.C4];
    reinterpret_cast<f32x4*>(a.dst)[e] = w0 * x0 + w1 * x1;
  }
}

// src = grad of the output (N, L_out, C), dst = grad of the input (N, L_in, C)
__global__ __launch_bounds__(256) void interp_bwd_kernel(const InterpArgs a) {
  const int64_t total = (int64_t)a.N * a.Lin * a.C4;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % a.C4);
    const int64_t r = e / a.C4;
    const int i = (int)(r % a.Lin), n = (int)(r / a.Lin);
    // output rows that can reference input row i: src = scale * t in (i - 1, i + 1); two rows of slack on either side
    int tlo = 0, thi = a.Lout - 1;
    if (a.scale > 0.f) {
      tlo = (int)floorf((float)(i - 1) * a.inv_scale) - 2;
      thi = (int)ceilf((float)(i + 1) * a.inv_scale) + 2;
      tlo = tlo < 0 ? 0 : tlo;
      thi = thi > a.Lout - 1 ? a.Lout - 1 : thi;
    }
    const f32x4* g = reinterpret_cast<const f32x4*>(a.src) + (int64_t)n * a.Lout * a.C4 + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = tlo; t <= thi; ++t) {
      int i0, i1; float w0, w1;
      interp_index(a.scale, a.Lin, t, i0, i1, w0, w1);
      const float w = (i0 == i ? w0 : 0.f) + (i1 == i ? w1 : 0.f);
      if (w != 0.f) acc += w * g[(int64_t)t * a.C4];
    }
    reinterpret_cast<f32x4*>(a.dst)[e] = acc;
  }
}

static int interp_launch(bool bwd, const float* src, int64_t N, int64_t Lin, int64_t C, int64_t Lout, float* dst, a2f_stream_t stream) {
  const char* what = bwd ? "interp_linear_bwd" : "interp_linear_fwd";
  A2F_CHECK_ARG(src && dst, "%s: null pointer", what);
  A2F_CHECK_ARG(N > 0 && Lin > 0 && Lout > 0 && C > 0 && (C % 4) == 0, "%s: N=%lld L_in=%lld L_out=%lld C=%lld (C must be a multiple of 4)", what,
                (long long)N, (long long)Lin, (long long)Lout, (long long)C);
  A2F_CHECK_ARG(Lin < (1 << 24) && Lout < (1 << 24), "%s: sequence too long for exact fp32 index arithmetic", what);
  A2F_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "%s: buffers must be 16-byte aligned", what);
  InterpArgs a;
  a.src = src; a.dst = dst; a.N = (int)N; a.Lin = (int)Lin; a.Lout = (int)Lout; a.C4 = (int)(C / 4);
  a.scale = Lout > 1 ? (float)(Lin - 1) / (float)(Lout - 1) : 0.f;
  a.inv_scale = a.scale > 0.f ? 1.f / a.scale : 0.f;
  const int64_t total = N * (bwd ? Lin : Lout) * (C / 4);
  int64_t blocks = cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  if (bwd) hipLaunchKernelGGL(interp_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(interp_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  A2F_CHECK_LAUNCH(what);
  return A2F_OK;
}

}  // namespace a2f

extern "C" int a2f_interp_linear_fwd(const float* x, int64_t N, int64_t L_in, int64_t C, int64_t L_out, float* y, a2f_stream_t stream) {
  return a2f::interp_launch(false, x, N, L_in, C, L_out, y, stream);
}

extern "C" int a2f_interp_linear_bwd(const float* dy, int64_t N, int64_t L_in, int64_t C, int64_t L_out, float* dx, a2f_stream_t stream) {
  return a2f::interp_launch(true, dy, N, L_in, C, L_out, dx, stream);
}
