// Layer 0 of the wav2vec2 feature encoder (HF Wav2Vec2FeatureEncoder as called at /root/reference/src/model/wav2vec.py:116):
//   conv1d(1 -> C channels, k = 10, stride 5, no bias)  ->  GroupNorm(C groups of one channel: statistics over time, per
//   sequence and channel, eps 1e-5, affine)  ->  GELU (erf form),
// written channels-last (S, L, C) for the implicit-GEMM convolutions that follow.  10 multiply-adds per output: the layer
// is pure HBM streaming of its (S, L, C) output (1 GB at 16 x 32 000 x 512), so the convolution is RECOMPUTED from the
// waveform wherever it is needed instead of being stored: forward = statistics pass (reads the waveform only) + apply pass
// (writes the output once); backward = reduction pass + apply pass over the incoming gradient.  All reductions go through
// per-workgroup partials folded in fixed order, in double where sums run over the whole sequence (deterministic).
#include "a2f_common.h"
#include "img.h"

namespace a2f {
namespace fe {

constexpr int KW = 10, STRIDE = 5;
constexpr int THREADS = 256;
constexpr int LB = 64;                    // positions per inner chunk
constexpr int XN = LB * STRIDE + KW;      // waveform samples a chunk touches (with slack)

// GELU and its derivative from ONE exponential and one reciprocal (the form of csrc/gemm_img.hip's epilogues): Phi(-|x|) = 0.5 erfc(|x| / sqrt 2) by
// Abramowitz-Stegun 7.1.26 (|error| <= 0.75e-7 on Phi; the small branch is never formed as 1 - (something near 1)), phi(x) = exp(-x^2 / 2) / sqrt(2 pi) shares
// the exponential.  These passes are bound by their vector arithmetic, not by HBM (10 multiply-adds of the recomputed convolution + the GroupNorm affine + GELU per
// element: ~43 vector operations per element with erff / expf, 0.57 ms of a 0.63-ms forward apply pass at 32 x 32 000 x 512), so the GELU's instruction count is what they cost
// (same box, forward 0.943 -> 0.857 ms, backward 1.879 -> 1.646 ms against erff / expf: tools/bench_featenc0.py).
__device__ __forceinline__ void gelu_cdf_pdf(float x, float& cdf, float& e) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);                                   // exp(-x^2 / 2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = 0.5f * p * t * e;                                                            // Phi(-|x|)
  cdf = x < 0.f ? q : 1.f - q;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gelu_cdf_pdf(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, e;
  gelu_cdf_pdf(x, cdf, e);
  return fmaf(x * 0.3989422804014327f, e, cdf);
}

struct Args {
  const float* wave;   // (S, N)
  const float* w;      // (C, KW)
  const float* gamma;  // (C)
  const float* beta;   // (C)
  const float* stats;  // (S, 2, C): mean, rstd
  const float* dout;   // (S, L, C)
  const float* sums;   // (S, 2, C): A = sum_l du, B = sum_l du * xhat
  float* out;          // (S, L, C)
  float* dy;           // optional (S, L, C): gradient w.r.t. the convolution output
  float* part;         // partial slabs
  int S, N, L, C, nchunk, chunks_per_block;
  int out_img;         // 1: out is written as a split-32 image (its only reader is the first convolution of the stack on image operands)
};

// thread -> 4 consecutive channels c4 = 4 (t % (C/4)); position phase ph = t / (C/4), PH phases
#define FE_SETUP                                                             \
  const int t = threadIdx.x;                                                 \
  const int cq = a.C >> 2;                                                   \
  const int PH = THREADS / cq;                                               \
  const int c4 = 4 * (t % cq), ph = t / cq;                                  \
  const int s = blockIdx.y;                                                  \
  __shared__ float xs[XN + 6];                                               \
  float wr[4][KW];                                                           \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                              \
      _Pragma("unroll") for (int k = 0; k < KW; ++k) wr[j][k] = a.w[(c4 + j) * KW + k];

__device__ __forceinline__ void load_chunk(const Args& a, int s, int l0, float* xs) {
  const long long base = (long long)s * a.N + (long long)l0 * STRIDE;
  for (int i = threadIdx.x; i < XN; i += THREADS) {
    const long long j = (long long)l0 * STRIDE + i;
    xs[i] = j < a.N ? a.wave[base + i] : 0.f;
  }
}
__device__ __forceinline__ void conv4(const float (&wr)[4][KW], const float* xp, float (&y)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = 0.f;
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const float xv = xp[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = fmaf(wr[j][k], xv, y[j]);
  }
}

// ---- forward pass 1: per-(block, channel) partial sum / sum of squares of the convolution output
__global__ __launch_bounds__(THREADS) void conv0_stats_kernel(const Args a) {
  FE_SETUP
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < a.chunks_per_block; ++ch) {
    const int l0 = (blockIdx.x * a.chunks_per_block + ch) * LB;
    if (l0 >= a.L) break;
    __syncthreads();
    load_chunk(a, s, l0, xs);
    __syncthreads();
    for (int p = ph; p < LB && l0 + p < a.L; p += PH) {
      float y[4];
      conv4(wr, xs + p * STRIDE, y);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += y[j];
        s2[j] = fmaf(y[j], y[j], s2[j]);
      }
    }
  }
  // part[s][block][phase][2][C]
  float* o = a.part + (((long long)s * gridDim.x + blockIdx.x) * PH + ph) * 2 * a.C;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[c4 + j] = s1[j];
    o[a.C + c4 + j] = s2[j];
  }
}

// stats[s][0][c] = mean, stats[s][1][c] = rstd from `nslab` partial (sum, sumsq) pairs per (s, c); double accumulation
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, int nslab, int C, int L, float eps, float* __restrict__ stats) {
  const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < nslab; ++b) {
    s1 += (double)part[((long long)s * nslab + b) * 2 * C + c];
    s2 += (double)part[((long long)s * nslab + b) * 2 * C + C + c];
  }
  const double mu = s1 / L;
  double var = s2 / L - mu * mu;
  if (var < 0.0) var = 0.0;
  stats[((long long)s * 2) * C + c] = (float)mu;
  stats[((long long)s * 2 + 1) * C + c] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- forward pass 2: out = gelu((conv - mean) * rstd * gamma + beta), channels-last float4 stores
__global__ __launch_bounds__(THREADS) void conv0_apply_kernel(const Args a) {
  FE_SETUP
  float mu[4], sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = a.stats[((long long)s * 2) * a.C + c4 + j];
    sc[j] = a.stats[((long long)s * 2 + 1) * a.C + c4 + j] * a.gamma[c4 + j];
    sh[j] = a.beta[c4 + j];
  }
  for (int ch = 0; ch < a.chunks_per_block; ++ch) {
    const int l0 = (blockIdx.x * a.chunks_per_block + ch) * LB;
    if (l0 >= a.L) break;
    __syncthreads();
    load_chunk(a, s, l0, xs);
    __syncthreads();
    for (int p = ph; p < LB && l0 + p < a.L; p += PH) {
      float y[4];
      conv4(wr, xs + p * STRIDE, y);
      float4 o;
      o.x = gelu_f((y[0] - mu[0]) * sc[0] + sh[0]);
      o.y = gelu_f((y[1] - mu[1]) * sc[1] + sh[1]);
      o.z = gelu_f((y[2] - mu[2]) * sc[2] + sh[2]);
      o.w = gelu_f((y[3] - mu[3]) * sc[3] + sh[3]);
      if (a.out_img) img::store4(reinterpret_cast<char*>(a.out) + ((long long)s * a.L + l0 + p) * a.C * 4, c4, f32x4{o.x, o.y, o.z, o.w});
      else *reinterpret_cast<float4*>(a.out + ((long long)s * a.L + l0 + p) * a.C + c4) = o;
    }
  }
}

// ---- backward pass 1: per-(block, channel) partials of A = sum du, B = sum du * xhat, du = dout * gelu'(u)
__global__ __launch_bounds__(THREADS) void conv0_bwd_reduce_kernel(const Args a) {
  FE_SETUP
  float mu[4], rs[4], g[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = a.stats[((long long)s * 2) * a.C + c4 + j];
    rs[j] = a.stats[((long long)s * 2 + 1) * a.C + c4 + j];
    g[j] = a.gamma[c4 + j];
    sh[j] = a.beta[c4 + j];
  }
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < a.chunks_per_block; ++ch) {
    const int l0 = (blockIdx.x * a.chunks_per_block + ch) * LB;
    if (l0 >= a.L) break;
    __syncthreads();
    load_chunk(a, s, l0, xs);
    __syncthreads();
    for (int p = ph; p < LB && l0 + p < a.L; p += PH) {
      float y[4];
      conv4(wr, xs + p * STRIDE, y);
      const float4 d = *reinterpret_cast<const float4*>(a.dout + ((long long)s * a.L + l0 + p) * a.C + c4);
      const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (y[j] - mu[j]) * rs[j];
        const float du = dd[j] * gelu_grad_f(xh * g[j] + sh[j]);
        sa[j] += du;
        sb[j] = fmaf(du, xh, sb[j]);
      }
    }
  }
  float* o = a.part + (((long long)s * gridDim.x + blockIdx.x) * PH + ph) * 2 * a.C;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[c4 + j] = sa[j];
    o[a.C + c4 + j] = sb[j];
  }
}

// sums[s][k][c] = sum over this sequence's slabs (double accumulation), one thread per (s, c)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part, int nslab, int C, float* __restrict__ sums) {
  const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (c >= C) return;
  double sa = 0.0, sb = 0.0;
  for (int b = 0; b < nslab; ++b) {
    sa += (double)part[((long long)s * nslab + b) * 2 * C + c];
    sb += (double)part[((long long)s * nslab + b) * 2 * C + C + c];
  }
  sums[((long long)s * 2) * C + c] = (float)sa;
  sums[((long long)s * 2 + 1) * C + c] = (float)sb;
}
// dbeta[c] = sum_s sums[s][0][c], dgamma[c] = sum_s sums[s][1][c]  (sequence order)
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ sums, int S, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double ga = 0.0, gb = 0.0;
  for (int s = 0; s < S; ++s) {
    ga += (double)sums[((long long)s * 2) * C + c];
    gb += (double)sums[((long long)s * 2 + 1) * C + c];
  }
  dbeta[c] = (float)ga;
  dgamma[c] = (float)gb;
}

// ---- backward pass 2: dy = rstd * gamma * (du - A/L - xhat * B/L); per-block partial dw[c][k] = sum dy * x[5 l + k]; optional dy store
__global__ __launch_bounds__(THREADS) void conv0_bwd_apply_kernel(const Args a) {
  FE_SETUP
  float mu[4], rs[4], g[4], sh[4], ma[4], mb[4];
  const float invL = 1.f / (float)a.L;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = a.stats[((long long)s * 2) * a.C + c4 + j];
    rs[j] = a.stats[((long long)s * 2 + 1) * a.C + c4 + j];
    g[j] = a.gamma[c4 + j];
    sh[j] = a.beta[c4 + j];
    ma[j] = a.sums[((long long)s * 2) * a.C + c4 + j] * invL;
    mb[j] = a.sums[((long long)s * 2 + 1) * a.C + c4 + j] * invL;
  }
  float dw[4][KW];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KW; ++k) dw[j][k] = 0.f;
  for (int ch = 0; ch < a.chunks_per_block; ++ch) {
    const int l0 = (blockIdx.x * a.chunks_per_block + ch) * LB;
    if (l0 >= a.L) break;
    __syncthreads();
    load_chunk(a, s, l0, xs);
    __syncthreads();
    for (int p = ph; p < LB && l0 + p < a.L; p += PH) {
      float y[4];
      const float* xp = xs + p * STRIDE;
      conv4(wr, xp, y);
      const long long off = ((long long)s * a.L + l0 + p) * a.C + c4;
      const float4 d = *reinterpret_cast<const float4*>(a.dout + off);
      const float dd[4] = {d.x, d.y, d.z, d.w};
      float dyv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (y[j] - mu[j]) * rs[j];
        const float du = dd[j] * gelu_grad_f(xh * g[j] + sh[j]);
        dyv[j] = rs[j] * g[j] * (du - ma[j] - xh * mb[j]);
      }
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const float xv = xp[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) dw[j][k] = fmaf(dyv[j], xv, dw[j][k]);
      }
      if (a.dy) *reinterpret_cast<float4*>(a.dy + off) = make_float4(dyv[0], dyv[1], dyv[2], dyv[3]);
    }
  }
  // part[s][block][phase][C][KW]
  float* o = a.part + (((long long)s * gridDim.x + blockIdx.x) * PH + ph) * a.C * KW;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KW; ++k) o[(c4 + j) * KW + k] = dw[j][k];
}

// out[i] = sum over slabs of part[b][i] (double accumulation): a wave owns 4 consecutive outputs, its lanes are 16 slab phases x 4
// outputs; the phase partials are folded by a butterfly over the phase bits (same order every run)
__global__ __launch_bounds__(256) void dw_fold_kernel(const float* __restrict__ part, int nslab, int n, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = (blockIdx.x * 4 + wave) * 4 + (lane & 3);
  double acc = 0.0;
  if (i < n)
    for (int b = lane >> 2; b < nslab; b += 16) acc += (double)part[(long long)b * n + i];
#pragma unroll
  for (int off = 4; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
  if (i < n && lane < 4) out[i] = (float)acc;
}

static int plan(int64_t L, int& nblk, int& cpb) {
  const int nchunk = (int)cdiv(L, LB);
  nblk = nchunk < 64 ? nchunk : 64;
  cpb = (int)cdiv(nchunk, nblk);
  nblk = (int)cdiv(nchunk, cpb);
  return nchunk;
}

}  // namespace fe
}  // namespace a2f

extern "C" size_t a2f_featenc0_ws_bytes(int64_t S, int64_t N, int64_t C) {
  using namespace a2f::fe;
  const int64_t L = (N - KW) / STRIDE + 1;
  int nblk, cpb;
  plan(L, nblk, cpb);
  const int PH = THREADS / (int)(C / 4);
  return ((size_t)S * nblk * PH * (size_t)C * KW + (size_t)S * 2 * C) * sizeof(float);   // weight-gradient partials + the (S,2,C) sums
}

extern "C" int a2f_featenc0_fwd(const float* wave, int64_t S, int64_t N, const float* w, const float* gamma, const float* beta, int64_t C,
                                float eps, float* out, float* stats, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  return a2f_featenc0_fwd_img(wave, S, N, w, gamma, beta, C, eps, out, 0, stats, ws, ws_bytes, stream);
}

extern "C" int a2f_featenc0_fwd_img(const float* wave, int64_t S, int64_t N, const float* w, const float* gamma, const float* beta, int64_t C,
                                    float eps, void* out_, int32_t out_format, float* stats, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  using namespace a2f;
  using namespace a2f::fe;
  float* out = (float*)out_;
  A2F_CHECK_ARG(out_format == 0 || (out_format == 1 && C % 32 == 0 && ((uintptr_t)out & 127) == 0), "featenc0_fwd: out_format %d (1 = image: C %% 32 == 0, 128-byte aligned)", out_format);
  A2F_CHECK_ARG(wave && w && gamma && beta && out && stats && ws, "featenc0_fwd: null pointer");
  A2F_CHECK_ARG(S > 0 && N >= KW && C % 4 == 0 && C >= 4 && THREADS % (C / 4) == 0 && C / 4 <= THREADS, "featenc0_fwd: S=%lld N=%lld C=%lld", (long long)S,
                (long long)N, (long long)C);
  A2F_CHECK_ARG(((uintptr_t)out & 15) == 0, "featenc0_fwd: out must be 16-byte aligned");
  if (ws_bytes < a2f_featenc0_ws_bytes(S, N, C)) return set_err(A2F_EWORKSPACE, "featenc0_fwd: workspace too small");
  Args a{};
  a.wave = wave; a.w = w; a.gamma = gamma; a.beta = beta; a.stats = stats; a.out = out; a.part = (float*)ws;
  a.out_img = out_format;
  a.S = (int)S; a.N = (int)N; a.L = (int)((N - KW) / STRIDE + 1); a.C = (int)C;
  int nblk, cpb;
  a.nchunk = plan(a.L, nblk, cpb);
  a.chunks_per_block = cpb;
  const int PH = THREADS / (int)(C / 4);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv0_stats_kernel, dim3(nblk, (unsigned)S), dim3(THREADS), 0, st, a);
  A2F_CHECK_LAUNCH("featenc0 stats");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)cdiv(C, 256), (unsigned)S), dim3(256), 0, st, (const float*)ws, nblk * PH, (int)C, a.L, eps, stats);
  A2F_CHECK_LAUNCH("featenc0 finalize");
  hipLaunchKernelGGL(conv0_apply_kernel, dim3(nblk, (unsigned)S), dim3(THREADS), 0, st, a);
  A2F_CHECK_LAUNCH("featenc0 apply");
  return A2F_OK;
}

extern "C" int a2f_featenc0_bwd(const float* wave, int64_t S, int64_t N, const float* w, const float* gamma, const float* beta, int64_t C,
                                const float* stats, const float* dout, float* dw, float* dgamma, float* dbeta, float* dy, void* ws,
                                size_t ws_bytes, a2f_stream_t stream) {
  using namespace a2f;
  using namespace a2f::fe;
  A2F_CHECK_ARG(wave && w && gamma && beta && stats && dout && dw && dgamma && dbeta && ws, "featenc0_bwd: null pointer");
  A2F_CHECK_ARG(S > 0 && N >= KW && C % 4 == 0 && C >= 4 && THREADS % (C / 4) == 0 && C / 4 <= THREADS, "featenc0_bwd: S=%lld N=%lld C=%lld", (long long)S,
                (long long)N, (long long)C);
  A2F_CHECK_ARG((((uintptr_t)dout | (uintptr_t)dy) & 15) == 0, "featenc0_bwd: buffers must be 16-byte aligned");
  if (ws_bytes < a2f_featenc0_ws_bytes(S, N, C)) return set_err(A2F_EWORKSPACE, "featenc0_bwd: workspace too small");
  Args a{};
  a.wave = wave; a.w = w; a.gamma = gamma; a.beta = beta; a.stats = stats; a.dout = dout; a.dy = dy;
  a.S = (int)S; a.N = (int)N; a.L = (int)((N - KW) / STRIDE + 1); a.C = (int)C;
  int nblk, cpb;
  a.nchunk = plan(a.L, nblk, cpb);
  a.chunks_per_block = cpb;
  const int PH = THREADS / (int)(C / 4);
  float* sums = (float*)ws;                                     // (S, 2, C)
  a.part = sums + (size_t)S * 2 * C;
  a.sums = sums;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv0_bwd_reduce_kernel, dim3(nblk, (unsigned)S), dim3(THREADS), 0, st, a);
  A2F_CHECK_LAUNCH("featenc0 bwd reduce");
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 256), (unsigned)S), dim3(256), 0, st, (const float*)a.part, nblk * PH, (int)C, sums);
  A2F_CHECK_LAUNCH("featenc0 bwd finalize");
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, (const float*)sums, (int)S, (int)C, dgamma, dbeta);
  A2F_CHECK_LAUNCH("featenc0 bwd params");
  hipLaunchKernelGGL(conv0_bwd_apply_kernel, dim3(nblk, (unsigned)S), dim3(THREADS), 0, st, a);
  A2F_CHECK_LAUNCH("featenc0 bwd apply");
  const int n = (int)(C * KW);
  hipLaunchKernelGGL(dw_fold_kernel, dim3((unsigned)cdiv(n, 16)), dim3(256), 0, st, (const float*)a.part, (int)(S * nblk * PH), n, dw);
  A2F_CHECK_LAUNCH("featenc0 bwd fold");
  return A2F_OK;
}
