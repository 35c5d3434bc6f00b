// Row-wise kernels of the wav2vec2 encoder path (HF Wav2Vec2Model as wrapped by /root/reference/src/model/wav2vec.py:
// the LayerNorms of feature projection / encoder / the 12 post-norm layers, bias gradients, the GELU derivative).
// All are HBM-streaming: one wave per row, float4 per lane, wave64 butterfly reductions; every cross-row reduction
// (dgamma, dbeta, column sums) goes through per-workgroup partial slabs folded in fixed order (deterministic).
#include "a2f_common.h"
#include "img.h"

namespace a2f {

constexpr int RO_MAXV = 4;            // float4 per lane: rows of up to 4 * 4 * 64 = 1024 floats
constexpr int RO_THREADS = 256;

__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma + beta, biased variance, two-pass in registers (torch's numerics)
// ---------------------------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(RO_THREADS) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t R, int C, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                                   float* __restrict__ mean, float* __restrict__ rstd, char* __restrict__ y_img) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (RO_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= R) return;
  const float4* xr = reinterpret_cast<const float4*>(x + row * C);
  const int nv = C >> 2;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    v[i] = c4 < nv ? xr[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  float4* yr = reinterpret_cast<float4*>(y + row * C);
  char* yi = y_img + row * C * 4;                          // split-32 image of y (the operand format of the image GEMMs): same row pitch as fp32
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
      const float4 o = make_float4((v[i].x - mu) * rs * g.x + b.x, (v[i].y - mu) * rs * g.y + b.y, (v[i].z - mu) * rs * g.z + b.z,
                                   (v[i].w - mu) * rs * g.w + b.w);
      if (y) yr[c4] = o;
      if (y_img) img::store4(yi, 4 * c4, f32x4{o.x, o.y, o.z, o.w});
    }
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; per-workgroup partials of
// dgamma = sum dy * xhat, dbeta = sum dy, and colsum(dx) (the bias gradient of the Linear that produced x) in `part`
// laid out [block][3][C]; layernorm_bwd_fold_kernel sums the blocks in order.
// ---------------------------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(RO_THREADS) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma, int64_t R, int C,
                                                                   float* __restrict__ dx, float* __restrict__ part, const unsigned* __restrict__ keep,
                                                                   int64_t keep_wpr, float keep_scale, char* __restrict__ dxm_img) {
  // third partial: column sums of what the Linear below receives = dx, or (keep given) the masked gradient keep ? dx * keep_scale : 0
  __shared__ float red[3][RO_THREADS / 64][NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = C >> 2;
  float4 g4[NV], ag[NV], ab[NV], ax[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    g4[i] = c4 < nv ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = ab[i] = ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // TWO rows per trip, both rows' loads issued before either row's reductions: one row at a time a wave has six 1-KB loads in flight and a CU (one workgroup of
  // four waves, 256 workgroups) 24 KB - the kernel ran at 2.8 TB/s.  The rows are still finished in the order row, row + stride, so every sum is unchanged.
  const int64_t wstride = (int64_t)gridDim.x * (RO_THREADS / 64);
  for (int64_t row0 = (int64_t)blockIdx.x * (RO_THREADS / 64) + wave; row0 < R; row0 += 2 * wstride) {
    float4 d[2][NV], xh[2][NV];
    float mu[2], rs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = row0 + u * wstride;
      const bool live = row < R;
      mu[u] = live ? mean[row] : 0.f;
      rs[u] = live ? rstd[row] : 0.f;
      const float4* dyr = reinterpret_cast<const float4*>(dy + (live ? row : row0) * C);
      const float4* xr = reinterpret_cast<const float4*>(x + (live ? row : row0) * C);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) { d[u][i] = dyr[c4]; xh[u][i] = xr[c4]; }
        else d[u][i] = xh[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = row0 + u * wstride;
      if (row >= R) break;                                   // wave-uniform
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
          const float4 xv = xh[u][i];
          xh[u][i] = make_float4((xv.x - mu[u]) * rs[u], (xv.y - mu[u]) * rs[u], (xv.z - mu[u]) * rs[u], (xv.w - mu[u]) * rs[u]);
        }
        const float gx = d[u][i].x * g4[i].x, gy = d[u][i].y * g4[i].y, gz = d[u][i].z * g4[i].z, gw = d[u][i].w * g4[i].w;
        s1 += (gx + gy) + (gz + gw);
        s2 += (gx * xh[u][i].x + gy * xh[u][i].y) + (gz * xh[u][i].z + gw * xh[u][i].w);
      }
      const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
      float4* dxr = reinterpret_cast<float4*>(dx + row * C);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
          float4 o;
          o.x = rs[u] * (d[u][i].x * g4[i].x - m1 - xh[u][i].x * m2);
          o.y = rs[u] * (d[u][i].y * g4[i].y - m1 - xh[u][i].y * m2);
          o.z = rs[u] * (d[u][i].z * g4[i].z - m1 - xh[u][i].z * m2);
          o.w = rs[u] * (d[u][i].w * g4[i].w - m1 - xh[u][i].w * m2);
          dxr[c4] = o;
          ag[i].x += d[u][i].x * xh[u][i].x; ag[i].y += d[u][i].y * xh[u][i].y; ag[i].z += d[u][i].z * xh[u][i].z; ag[i].w += d[u][i].w * xh[u][i].w;
          ab[i].x += d[u][i].x; ab[i].y += d[u][i].y; ab[i].z += d[u][i].z; ab[i].w += d[u][i].w;
          if (keep) {                                        // train-mode dropout of the Linear's output: its gradient passes the same mask
            const unsigned b = keep[row * keep_wpr + (c4 >> 3)] >> ((c4 & 7) * 4);
            o.x = (b & 1u) ? o.x * keep_scale : 0.f;
            o.y = (b & 2u) ? o.y * keep_scale : 0.f;
            o.z = (b & 4u) ? o.z * keep_scale : 0.f;
            o.w = (b & 8u) ? o.w * keep_scale : 0.f;
          }
          if (dxm_img) img::store4(dxm_img + row * C * 4, 4 * c4, f32x4{o.x, o.y, o.z, o.w});
          ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
        }
      }
    }
  }
  // fold the four waves of the workgroup through LDS (fixed order), one slab row per workgroup
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    *reinterpret_cast<float4*>(&red[0][wave][c]) = ag[i];
    *reinterpret_cast<float4*>(&red[1][wave][c]) = ab[i];
    *reinterpret_cast<float4*>(&red[2][wave][c]) = ax[i];
  }
  __syncthreads();
  float* out = part + (int64_t)blockIdx.x * 3 * C;
  for (int idx = threadIdx.x; idx < 3 * C; idx += RO_THREADS) {
    const int k = idx / C, c = idx - k * C;
    out[idx] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
  }
}

// out[k][c] = sum over blocks of part[b][k][c]   (k < nk), fixed order.  One workgroup per 16 output elements: thread t owns element t & 15 and the
// blocks b = (t >> 4), + 16, + 32, ... in four independent chains, so all of its loads (16 at 256 blocks) are in flight together - the previous form walked
// 64 dependent loads per lane, 22 us per call and 72 calls in a 64 x 600 FaceFormer step; the sixteen partials of an element are then added in order through LDS.
__global__ __launch_bounds__(256) void slab_fold_kernel(const float* __restrict__ part, int nblocks, int nk, int C, float* __restrict__ o0,
                                                        float* __restrict__ o1, float* __restrict__ o2) {
  __shared__ float sh[16][17];
  const int e = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + e;
  const int total = nk * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < total) {
    int b = ph;
    for (; b + 48 < nblocks; b += 64) {
      s0 += part[(int64_t)b * total + idx];
      s1 += part[(int64_t)(b + 16) * total + idx];
      s2 += part[(int64_t)(b + 32) * total + idx];
      s3 += part[(int64_t)(b + 48) * total + idx];
    }
    for (; b < nblocks; b += 16) s0 += part[(int64_t)b * total + idx];
  }
  sh[ph][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < 16 && idx < total) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += sh[g][threadIdx.x];
    const int k = idx / C, c = idx - k * C;
    float* o = k == 0 ? o0 : (k == 1 ? o1 : o2);
    if (o) o[c] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Column sums of a (R, C) matrix with row stride ld (bias gradients): partials [block][C], then slab_fold_kernel
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RO_THREADS) void colsum_partial_kernel(const float* __restrict__ x, int64_t R, int C, int64_t ld, float* __restrict__ part) {
  // thread t owns float4 column c4 = t % nv4 ... handled by a 2-D split: columns over (blockIdx.y, lanes), rows over blockIdx.x
  const int c4 = blockIdx.y * RO_THREADS + threadIdx.x;
  const int nv = C >> 2;
  if (c4 >= nv) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(x + row * ld + 4 * c4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * C + 4 * c4) = acc;
}

// the same over a split-32 image (values hi + lo): the bias gradient of a Linear whose output gradient only exists as an image
__global__ __launch_bounds__(RO_THREADS) void colsum_img_partial_kernel(const char* __restrict__ x, int64_t R, int C, int64_t ld, float* __restrict__ part) {
  const int c4 = blockIdx.y * RO_THREADS + threadIdx.x;
  const int nv = C >> 2;
  if (c4 >= nv) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t row = blockIdx.x; row < R; row += gridDim.x) acc += img::load4(x + row * ld * 4, 4 * c4);
  *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.x * C + 4 * c4) = acc;
}

// dz = dy * gelu'(z), float4 grid-stride (may run in place: out == dy)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ z, int64_t n4, float4* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += stride) {
    const float4 d = dy[i], v = z[i];
    out[i] = make_float4(d.x * gelu_grad(v.x), d.y * gelu_grad(v.y), d.z * gelu_grad(v.z), d.w * gelu_grad(v.w));
  }
}

// Persistent workgroups of the backward: TWO per CU (218 registers = two waves per SIMD, 48 KB of LDS each) - 96 KB of row loads in flight per CU instead of 48;
// 19 200 x 768 with keep bits and image output: 54.6 -> 47.1 us (4.3 -> 5.0 TB/s; 1024 workgroups: 50.5 us).  One slab row per workgroup in the workspace.
static int ln_blocks_bwd(int64_t R) {
  int64_t b = cdiv(R, RO_THREADS / 64);
  return (int)(b > 512 ? 512 : b);
}

}  // namespace a2f

extern "C" size_t a2f_layernorm_bwd_ws_bytes(int64_t R, int64_t C) { return (size_t)a2f::ln_blocks_bwd(R) * 3 * (size_t)C * sizeof(float); }

extern "C" int a2f_layernorm_fwd(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta, float eps, float* y, float* mean,
                                 float* rstd, a2f_stream_t stream) {
  A2F_CHECK_ARG(y != nullptr, "layernorm_fwd: null pointer");
  return a2f_layernorm_fwd_img(x, R, C, gamma, beta, eps, y, nullptr, mean, rstd, stream);
}

extern "C" int a2f_layernorm_fwd_img(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta, float eps, float* y, void* y_img_, float* mean,
                                     float* rstd, a2f_stream_t stream) {
  using namespace a2f;
  char* y_img = (char*)y_img_;
  A2F_CHECK_ARG(x && gamma && beta && (y || y_img) && mean && rstd, "layernorm_fwd: null pointer");
  A2F_CHECK_ARG(y_img == nullptr || (C % 32 == 0 && ((uintptr_t)y_img & 127) == 0), "layernorm_fwd: an image output needs C %% 32 == 0 and a 128-byte aligned buffer");
  A2F_CHECK_ARG(R > 0 && C > 0 && C % 4 == 0 && C <= RO_MAXV * 256, "layernorm_fwd: R=%lld C=%lld (C must be a multiple of 4, <= 1024)", (long long)R, (long long)C);
  A2F_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_fwd: buffers must be 16-byte aligned");
  const dim3 grid((unsigned)cdiv(R, RO_THREADS / 64));
  hipStream_t st = (hipStream_t)stream;
  if (C <= 256) hipLaunchKernelGGL(layernorm_fwd_kernel<1>, grid, dim3(RO_THREADS), 0, st, x, R, (int)C, gamma, beta, eps, y, mean, rstd, y_img);
  else if (C <= 512) hipLaunchKernelGGL(layernorm_fwd_kernel<2>, grid, dim3(RO_THREADS), 0, st, x, R, (int)C, gamma, beta, eps, y, mean, rstd, y_img);
  else hipLaunchKernelGGL(layernorm_fwd_kernel<4>, grid, dim3(RO_THREADS), 0, st, x, R, (int)C, gamma, beta, eps, y, mean, rstd, y_img);
  A2F_CHECK_LAUNCH("layernorm_fwd");
  return A2F_OK;
}

extern "C" int a2f_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t C,
                                 float* dx, float* dgamma, float* dbeta, float* dx_colsum, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  return a2f_layernorm_bwd_img(dy, x, mean, rstd, gamma, R, C, dx, dgamma, dbeta, dx_colsum, nullptr, 0, 1.f, nullptr, ws, ws_bytes, stream);
}

extern "C" int a2f_layernorm_bwd_img(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t C,
                                     float* dx, float* dgamma, float* dbeta, float* dxm_colsum, const uint32_t* keep_bits, int64_t keep_wpr, float keep_scale,
                                     void* dxm_img_, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  using namespace a2f;
  float* dx_colsum = dxm_colsum;
  char* dxm_img = (char*)dxm_img_;
  A2F_CHECK_ARG(dxm_img == nullptr || (C % 32 == 0 && ((uintptr_t)dxm_img & 127) == 0), "layernorm_bwd: an image output needs C %% 32 == 0 and a 128-byte aligned buffer");
  A2F_CHECK_ARG(keep_bits == nullptr || keep_wpr * 32 >= C, "layernorm_bwd: keep_wpr * 32 < C");
  A2F_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && ws, "layernorm_bwd: null pointer");
  A2F_CHECK_ARG(R > 0 && C > 0 && C % 4 == 0 && C <= RO_MAXV * 256, "layernorm_bwd: R=%lld C=%lld", (long long)R, (long long)C);
  A2F_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)ws) & 15) == 0, "layernorm_bwd: buffers must be 16-byte aligned");
  if (ws_bytes < a2f_layernorm_bwd_ws_bytes(R, C)) return set_err(A2F_EWORKSPACE, "layernorm_bwd: workspace too small");
  const int nb = ln_blocks_bwd(R);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  if (C <= 256) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(nb), dim3(RO_THREADS), 0, st, dy, x, mean, rstd, gamma, R, (int)C, dx, part, keep_bits, keep_wpr, keep_scale, dxm_img);
  else if (C <= 512) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(nb), dim3(RO_THREADS), 0, st, dy, x, mean, rstd, gamma, R, (int)C, dx, part, keep_bits, keep_wpr, keep_scale, dxm_img);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(nb), dim3(RO_THREADS), 0, st, dy, x, mean, rstd, gamma, R, (int)C, dx, part, keep_bits, keep_wpr, keep_scale, dxm_img);
  A2F_CHECK_LAUNCH("layernorm_bwd");
  hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)cdiv(3 * C, 16)), dim3(256), 0, st, part, nb, 3, (int)C, dgamma, dbeta, dx_colsum);
  A2F_CHECK_LAUNCH("layernorm_bwd fold");
  return A2F_OK;
}

extern "C" size_t a2f_colsum_ws_bytes(int64_t R, int64_t C) {
  int64_t b = R < 256 ? R : 256;
  return (size_t)b * (size_t)C * sizeof(float);
}

extern "C" int a2f_colsum(const float* x, int64_t R, int64_t C, int64_t ld, float* out, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(x && out && ws, "colsum: null pointer");
  A2F_CHECK_ARG(R > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "colsum: R=%lld C=%lld ld=%lld", (long long)R, (long long)C, (long long)ld);
  A2F_CHECK_ARG((((uintptr_t)x | (uintptr_t)ws) & 15) == 0, "colsum: buffers must be 16-byte aligned");
  if (ws_bytes < a2f_colsum_ws_bytes(R, C)) return set_err(A2F_EWORKSPACE, "colsum: workspace too small");
  const int nb = (int)(R < 256 ? R : 256);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb, (unsigned)cdiv(C / 4, RO_THREADS)), dim3(RO_THREADS), 0, st, x, R, (int)C, ld, (float*)ws);
  A2F_CHECK_LAUNCH("colsum");
  hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)cdiv(C, 16)), dim3(256), 0, st, (const float*)ws, nb, 1, (int)C, out, (float*)nullptr, (float*)nullptr);
  A2F_CHECK_LAUNCH("colsum fold");
  return A2F_OK;
}

extern "C" int a2f_colsum_img(const void* x, int64_t R, int64_t C, int64_t ld, float* out, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(x && out && ws, "colsum_img: null pointer");
  A2F_CHECK_ARG(R > 0 && C > 0 && C % 4 == 0 && ld % 32 == 0 && ld >= C && ((uintptr_t)x & 127) == 0, "colsum_img: R=%lld C=%lld ld=%lld", (long long)R, (long long)C, (long long)ld);
  if (ws_bytes < a2f_colsum_ws_bytes(R, C)) return set_err(A2F_EWORKSPACE, "colsum_img: workspace too small");
  const int nb = (int)(R < 256 ? R : 256);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_img_partial_kernel, dim3(nb, (unsigned)cdiv(C / 4, RO_THREADS)), dim3(RO_THREADS), 0, st, (const char*)x, R, (int)C, ld, (float*)ws);
  A2F_CHECK_LAUNCH("colsum_img");
  hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)cdiv(C, 16)), dim3(256), 0, st, (const float*)ws, nb, 1, (int)C, out, (float*)nullptr, (float*)nullptr);
  A2F_CHECK_LAUNCH("colsum_img fold");
  return A2F_OK;
}

extern "C" int a2f_gelu_bwd(const float* dy, const float* z, int64_t n, float* out, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dy && z && out && n > 0 && n % 4 == 0, "gelu_bwd: n=%lld must be a positive multiple of 4", (long long)n);
  A2F_CHECK_ARG((((uintptr_t)dy | (uintptr_t)z | (uintptr_t)out) & 15) == 0, "gelu_bwd: buffers must be 16-byte aligned");
  int64_t blocks = cdiv(n / 4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, (const float4*)z, n / 4, (float4*)out);
  A2F_CHECK_LAUNCH("gelu_bwd");
  return A2F_OK;
}
