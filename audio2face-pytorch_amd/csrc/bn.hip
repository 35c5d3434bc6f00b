// BatchNorm2d over channels-last rows x (R, C)  (/root/reference/src/model/audio2face.py:15-46, nn.BatchNorm2d
// semantics: training-mode batch statistics with biased variance for normalisation, unbiased for the running
// buffer, momentum 0.1, eps 1e-5; optional fused ReLU).
//
// HBM-bound elementwise + column reductions.  Reductions are deterministic two-stage: every workgroup reduces a
// slab of rows into double partials [block][2][C] in the workspace, one finishing workgroup folds them in fixed order.
// Channel index = flat index mod C, so consecutive lanes read consecutive addresses (fully coalesced) for any C.
// Round 3: (i) the backward no longer reads the saved output y for the ReLU mask - y > 0 <=> fma(x, invstd gamma, beta - mean invstd gamma) > 0,
// the forward's own expression (bn_affine) re-evaluated from x: one (R, C) read less in each of the two backward passes; (ii) the two
// apply kernels move one float4 per thread with one workgroup per 4 KB (the shape that streams at 6 TB/s on this chip, util.hip) instead of
// 4-byte elements in a persistent grid-stride loop with an integer modulo per element.
#include "a2f_common.h"

#include <algorithm>

namespace a2f {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_BLOCKS = 1024;
constexpr int BN_MAX_C = 1024;

__host__ __device__ inline int bn_blocks(int64_t R, int C) {
  // each block owns a contiguous slab of rows; lanes map to (row-in-pass, channel)
  const int rows_per_pass = BN_THREADS / C > 0 ? BN_THREADS / C : 1;
  int64_t b = (R + (int64_t)rows_per_pass * 64 - 1) / ((int64_t)rows_per_pass * 64);
  if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

// the normalise-scale-shift of the forward, shared with the backward's ReLU mask (same instruction, same rounding)
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return fmaf(x, sc, sh); }
__device__ __forceinline__ float bn_scale(float invstd, float gamma) { return invstd * gamma; }
__device__ __forceinline__ float bn_shift(float beta, float mean, float sc) { return fmaf(-mean, sc, beta); }

// Where x comes from.  Memory (a0 == nullptr), or RECOMPUTED (round 3, Audio2Mesh analysis layer 0, audio2face.py:13-17): x(r, c) = cbias[c] + sum_kh a0[2r - 1 + kh]
// wg[c][kh] (a0[2r - 1] = 0 at the first position of every group of P_out rows: the left padding) is the one-input-channel k = 3 / stride 2 / pad 1 convolution - three multiply-adds from an input 72x smaller than the (R, C)
// tensor, so neither the forward's two passes nor the backward's two ever read (or write) the convolution's 302-MB output at B = 1024.  Every pass evaluates the
// same expression (conv0_eval4), so the ReLU masks and x-hat of the backward are the forward's bit for bit.
struct BnSrc {
  const float* x;
  const float* a0;
  const float* wg;       // (C, 3)
  const float* cbias;    // (C)
  int P_out;
};
__device__ __forceinline__ float4 conv0_eval4(const float* __restrict__ a0, int64_t r, int P_out, const float* w12, const float4 b) {
  const float x1 = a0[2 * r], x2 = a0[2 * r + 1];
  const float x0 = (r % P_out) ? a0[2 * r - 1] : 0.f;
  return make_float4(((b.x + x0 * w12[0]) + x1 * w12[1]) + x2 * w12[2], ((b.y + x0 * w12[3]) + x1 * w12[4]) + x2 * w12[5],
                     ((b.z + x0 * w12[6]) + x1 * w12[7]) + x2 * w12[8], ((b.w + x0 * w12[9]) + x1 * w12[10]) + x2 * w12[11]);
}

// part[blk][0][c] = sum_r u(r,c) ; part[blk][1][c] = sum_r u(r,c)*w(r,c)
//   MODE 0 (fwd stats): u = x, w = x
//   MODE 1 (bwd):       u = dy' , w = xhat   (dy' = relu ? dy*(bn_affine(x) > 0) : dy; `y` = beta in this mode)
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                 const float* __restrict__ dy, int64_t R, int C,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, int relu, double* __restrict__ part) {
  const int t = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  const int64_t rows_per_block = (R + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  // C <= 256: lane -> (row offset t / C, channel t % C); C > 256: channels strided over lanes, one row per pass
  if (C <= BN_THREADS) {
    const int rpp = BN_THREADS / C;
    const int c = t % C, ro = t / C;
    if (ro < rpp) {
      float mu = 0.f, is = 0.f, sc = 0.f, sh = 1.f;                  // sc = 0, sh = 1: "always positive" when there is no ReLU
      if (MODE == 1) {
        mu = mean[c]; is = invstd[c];
        if (relu) { sc = bn_scale(is, gamma[c]); sh = bn_shift(y[c], mu, sc); }
      }
      // four rows per trip: the loads are independent, the double accumulation order stays fixed
      int64_t r = r0 + ro;
      for (; r + 3 * rpp < r1; r += 4 * rpp) {
        float xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t i = (r + (int64_t)u * rpp) * C + c;
          xv[u] = x[i];
          if (MODE == 1) gv[u] = dy[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (MODE == 0) { s0 += xv[u]; s1 += (double)xv[u] * xv[u]; }
          else { const float g = bn_affine(xv[u], sc, sh) > 0.f ? gv[u] : 0.f; s0 += g; s1 += (double)g * ((xv[u] - mu) * is); }
        }
      }
      for (; r < r1; r += rpp) {
        const int64_t i = r * C + c;
        if (MODE == 0) {
          const float v = x[i];
          s0 += v; s1 += (double)v * v;
        } else {
          const float xv1 = x[i];
          const float g = bn_affine(xv1, sc, sh) > 0.f ? dy[i] : 0.f;
          s0 += g; s1 += (double)g * ((xv1 - mu) * is);
        }
      }
    }
    // fold the rpp row-lanes of each channel through LDS
    __shared__ double sh[2][BN_THREADS];
    sh[0][t] = s0; sh[1][t] = s1;
    __syncthreads();
    if (t < C) {
      double a = 0.0, b = 0.0;
      for (int k = 0; k < rpp; ++k) { a += sh[0][t + k * C]; b += sh[1][t + k * C]; }
      part[((int64_t)blockIdx.x * 2 + 0) * C + t] = a;
      part[((int64_t)blockIdx.x * 2 + 1) * C + t] = b;
    }
  } else {
    for (int c = t; c < C; c += BN_THREADS) {
      s0 = 0.0; s1 = 0.0;
      float mu = 0.f, is = 0.f, sc = 0.f, sh = 1.f;
      if (MODE == 1) {
        mu = mean[c]; is = invstd[c];
        if (relu) { sc = bn_scale(is, gamma[c]); sh = bn_shift(y[c], mu, sc); }
      }
      for (int64_t r = r0; r < r1; ++r) {
        const int64_t i = r * C + c;
        if (MODE == 0) { const float v = x[i]; s0 += v; s1 += (double)v * v; }
        else { const float xv1 = x[i]; const float g = bn_affine(xv1, sc, sh) > 0.f ? dy[i] : 0.f; s0 += g; s1 += (double)g * ((xv1 - mu) * is); }
      }
      part[((int64_t)blockIdx.x * 2 + 0) * C + c] = s0;
      part[((int64_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
  }
}

// The same partials for C % 4 == 0 (every Audio2Mesh layer: 72, 108, 164, 244, 256) with 16-byte loads: lane -> (row offset t / (C/4), channel
// quad t % (C/4)), four rows per trip = up to twelve float4 loads in flight per lane (the scalar form above reaches 3.1-3.5 TB/s, this one the
// stream rate).  Same [block][2][C] double layout; the accumulation order differs from the scalar form but is fixed.
template <int MODE, bool CONV0 = false>
__global__ __launch_bounds__(BN_THREADS) void bn_partial4_kernel(const BnSrc src, const float* __restrict__ y,
                                                                  const float* __restrict__ dy, int64_t R, int C,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, int relu, double* __restrict__ part) {
  const int t = threadIdx.x;
  const int Q = C >> 2, rpp = BN_THREADS / Q;
  const int q = t % Q, ro = t / Q;
  const int64_t rows_per_block = (R + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  const float* __restrict__ x = src.x;
  float w12[12];
  float4 cb4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (CONV0) {
    if (ro < rpp) {
#pragma unroll
      for (int e = 0; e < 12; ++e) w12[e] = src.wg[12 * q + e];
      cb4 = *reinterpret_cast<const float4*>(src.cbias + 4 * q);
    }
  }
  auto load_x = [&](int64_t row) {
    if constexpr (CONV0) return conv0_eval4(src.a0, row, src.P_out, w12, cb4);
    else return *reinterpret_cast<const float4*>(x + row * C + 4 * q);
  };
  if (ro < rpp) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, sc = mu, sh = make_float4(1.f, 1.f, 1.f, 1.f);
    if (MODE == 1) {
      mu = *reinterpret_cast<const float4*>(mean + 4 * q); is = *reinterpret_cast<const float4*>(invstd + 4 * q);
      if (relu) {
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + 4 * q), b4 = *reinterpret_cast<const float4*>(y + 4 * q);     // y = beta in this mode
        sc = make_float4(bn_scale(is.x, g4.x), bn_scale(is.y, g4.y), bn_scale(is.z, g4.z), bn_scale(is.w, g4.w));
        sh = make_float4(bn_shift(b4.x, mu.x, sc.x), bn_shift(b4.y, mu.y, sc.y), bn_shift(b4.z, mu.z, sc.z), bn_shift(b4.w, mu.w, sc.w));
      }
    }
    auto acc = [&](const float4 xv, const float4 gv) {
      if (MODE == 0) {
        s0[0] += xv.x; s1[0] += (double)xv.x * xv.x; s0[1] += xv.y; s1[1] += (double)xv.y * xv.y;
        s0[2] += xv.z; s1[2] += (double)xv.z * xv.z; s0[3] += xv.w; s1[3] += (double)xv.w * xv.w;
      } else {
        const float g0 = bn_affine(xv.x, sc.x, sh.x) > 0.f ? gv.x : 0.f, g1 = bn_affine(xv.y, sc.y, sh.y) > 0.f ? gv.y : 0.f;
        const float g2 = bn_affine(xv.z, sc.z, sh.z) > 0.f ? gv.z : 0.f, g3 = bn_affine(xv.w, sc.w, sh.w) > 0.f ? gv.w : 0.f;
        s0[0] += g0; s1[0] += (double)g0 * ((xv.x - mu.x) * is.x);
        s0[1] += g1; s1[1] += (double)g1 * ((xv.y - mu.y) * is.y);
        s0[2] += g2; s1[2] += (double)g2 * ((xv.z - mu.z) * is.z);
        s0[3] += g3; s1[3] += (double)g3 * ((xv.w - mu.w) * is.w);
      }
    };
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    int64_t r = r0 + ro;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {
      float4 xv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = r + (int64_t)u * rpp;
        xv[u] = load_x(row);
        gv[u] = MODE == 1 ? *reinterpret_cast<const float4*>(dy + row * C + 4 * q) : one;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc(xv[u], gv[u]);
    }
    for (; r < r1; r += rpp) {
      const float4 xv = load_x(r);
      const float4 gv = MODE == 1 ? *reinterpret_cast<const float4*>(dy + r * C + 4 * q) : one;
      acc(xv, gv);
    }
  }
  __shared__ double sh[2][4][BN_THREADS];
#pragma unroll
  for (int e = 0; e < 4; ++e) { sh[0][e][t] = s0[e]; sh[1][e][t] = s1[e]; }
  __syncthreads();
  if (t < C) {                                                      // channel t = quad t / 4, element t % 4: fold its rpp row-lanes in fixed order
    const int qq = t >> 2, e = t & 3;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < rpp; ++k) { a += sh[0][e][qq + k * Q]; b += sh[1][e][qq + k * Q]; }
    part[((int64_t)blockIdx.x * 2 + 0) * C + t] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * C + t] = b;
  }
}

static inline bool bn_vec4_ok(const void* a, const void* b, const void* c, int64_t C) {
  return C % 4 == 0 && C / 4 <= BN_THREADS && C <= BN_THREADS && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

// Fold of the per-block double partials: 4 channels x 64 block-lanes per workgroup (a lane sums every 64th block with all its loads in flight at once, the 64
// lane sums are added in fixed order through LDS) - deterministic.  (16 channels x 16 lanes, i.e. up to 64 dependent L2 round trips per lane on 5-16 workgroups,
// made each of the twenty finish launches of an Audio2Mesh step 9.5 us long.)
constexpr int BN_FOLD_CH = 4, BN_FOLD_LANES = 64;
__device__ __forceinline__ void bn_fold_partials(const double* __restrict__ part, int nblk, int C, int c, int zg, bool valid,
                                                 double (&sh)[2][BN_FOLD_LANES][BN_FOLD_CH + 1], double& s, double& q) {
  double a[4] = {0.0, 0.0, 0.0, 0.0}, bq[4] = {0.0, 0.0, 0.0, 0.0};
  if (valid) {
    int b = zg;
    for (; b + 3 * BN_FOLD_LANES < nblk; b += 4 * BN_FOLD_LANES) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] += part[((int64_t)(b + u * BN_FOLD_LANES) * 2 + 0) * C + c];
        bq[u] += part[((int64_t)(b + u * BN_FOLD_LANES) * 2 + 1) * C + c];
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {                            // at most three blocks are left for this lane
      const int bb = b + u * BN_FOLD_LANES;
      if (bb < nblk) {
        a[u] += part[((int64_t)bb * 2 + 0) * C + c];
        bq[u] += part[((int64_t)bb * 2 + 1) * C + c];
      }
    }
  }
  const int o = threadIdx.x % BN_FOLD_CH;
  sh[0][zg][o] = (a[0] + a[1]) + (a[2] + a[3]);
  sh[1][zg][o] = (bq[0] + bq[1]) + (bq[2] + bq[3]);
  __syncthreads();
  s = 0.0; q = 0.0;
  if (zg == 0) {
#pragma unroll 8
    for (int g = 0; g < BN_FOLD_LANES; ++g) { s += sh[0][g][o]; q += sh[1][g][o]; }
  }
}

__global__ __launch_bounds__(256) void bn_finish_stats_kernel(const double* __restrict__ part, int nblk, int64_t R, int C, float momentum,
                                                               float eps, float* __restrict__ rmean, float* __restrict__ rvar,
                                                               float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  __shared__ double sh[2][BN_FOLD_LANES][BN_FOLD_CH + 1];
  const int c = blockIdx.x * BN_FOLD_CH + (threadIdx.x % BN_FOLD_CH), zg = threadIdx.x / BN_FOLD_CH;
  double s, q;
  bn_fold_partials(part, nblk, C, c, zg, c < C, sh, s, q);
  if (zg != 0 || c >= C) return;
  const double mean = s / (double)R;
  double var = q / (double)R - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  const double unb = R > 1 ? var * ((double)R / (double)(R - 1)) : var;
  rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
  rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = rmean[c];
  save_invstd[c] = (float)(1.0 / sqrt((double)rvar[c] + (double)eps));
}

__global__ __launch_bounds__(256) void bn_finish_bwd_kernel(const double* __restrict__ part, int nblk, int C, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
  __shared__ double sh[2][BN_FOLD_LANES][BN_FOLD_CH + 1];
  const int c = blockIdx.x * BN_FOLD_CH + (threadIdx.x % BN_FOLD_CH), zg = threadIdx.x / BN_FOLD_CH;
  double s, q;
  bn_fold_partials(part, nblk, C, c, zg, c < C, sh, s, q);
  if (zg != 0 || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
}

// (row, channel) of flat element i: 32-bit arithmetic whenever the tensor allows it (block-uniform) - the 64-bit modulo by a run-time C is ~50 instructions, half
// of what a thread of a kernel that streams at the HBM rate can afford
__device__ __forceinline__ void bn_row_col(int64_t i, int64_t total, int C, int64_t& row, int& c) {
  if (total < (1ll << 31)) {
    const unsigned r = (unsigned)i / (unsigned)C;
    row = r;
    c = (int)((unsigned)i - r * (unsigned)C);
  } else {
    row = i / C;
    c = (int)(i - row * C);
  }
}

// y = (x - mean) * invstd * gamma + beta  [relu]        VEC: one float4 per thread (C % 4 == 0, 16-byte aligned), else one element
template <bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const float* __restrict__ x, int64_t total, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               int relu, float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float sc[BN_MAX_C], sh[BN_MAX_C];
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    const float s = bn_scale(invstd[c], gamma[c]);
    sc[c] = s;
    sh[c] = bn_shift(beta[c], mean[c], s);
  }
  __syncthreads();
  const int64_t i = (blockIdx.x * (int64_t)BN_THREADS + threadIdx.x) * (VEC ? 4 : 1);
  if (i >= total) return;
  int64_t row;
  int c;
  bn_row_col(i, total, C, row, c);
  if constexpr (VEC) {
    const float4 xv = *reinterpret_cast<const float4*>(x + i);
    const float4 s4 = *reinterpret_cast<const float4*>(sc + c), h4 = *reinterpret_cast<const float4*>(sh + c);
    float4 v = make_float4(bn_affine(xv.x, s4.x, h4.x), bn_affine(xv.y, s4.y, h4.y), bn_affine(xv.z, s4.z, h4.z), bn_affine(xv.w, s4.w, h4.w));
    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    *reinterpret_cast<float4*>(y + i) = v;
  } else {
    float v = bn_affine(x[i], sc[c], sh[c]);
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
  }
}

// training: dx = gamma*invstd/R * (R*dy' - dbeta - xhat*dgamma) ; eval: dx = dy'*gamma*invstd        (dy' = dy where the forward output was > 0)
template <bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                                                   const float* __restrict__ dy, int64_t total, int C, float invR,
                                                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ dgamma,
                                                                   const float* __restrict__ dbeta, int training, int relu,
                                                                   float* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) float s_gi[BN_MAX_C], s_mu[BN_MAX_C], s_is[BN_MAX_C], s_db[BN_MAX_C], s_dg[BN_MAX_C], s_sh[BN_MAX_C];
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    const float gi = bn_scale(invstd[c], gamma[c]);           // the forward's scale, bit for bit
    s_gi[c] = gi;
    s_mu[c] = mean[c];
    s_is[c] = invstd[c];
    s_db[c] = training ? dbeta[c] * invR : 0.f;
    s_dg[c] = training ? dgamma[c] * invR : 0.f;
    s_sh[c] = relu ? bn_shift(beta[c], mean[c], gi) : 1.f;
  }
  __syncthreads();
  const int64_t i = (blockIdx.x * (int64_t)BN_THREADS + threadIdx.x) * (VEC ? 4 : 1);
  if (i >= total) return;
  int64_t row;
  int c;
  bn_row_col(i, total, C, row, c);
  auto one = [&](float xv, float g, int cc) {
    if (!(bn_affine(xv, relu ? s_gi[cc] : 0.f, s_sh[cc]) > 0.f)) g = 0.f;
    const float xhat = (xv - s_mu[cc]) * s_is[cc];
    return s_gi[cc] * (g - s_db[cc] - xhat * s_dg[cc]);
  };
  if constexpr (VEC) {
    const float4 xv = *reinterpret_cast<const float4*>(x + i), gv = *reinterpret_cast<const float4*>(dy + i);
    *reinterpret_cast<float4*>(dx + i) = make_float4(one(xv.x, gv.x, c), one(xv.y, gv.y, c + 1), one(xv.z, gv.z, c + 2), one(xv.w, gv.w, c + 3));
  } else {
    dx[i] = one(x[i], dy[i], c);
  }
}

// The two apply passes with x recomputed (CONV0), ROW-PERSISTENT: thread (row offset t / (C/4), channel quad t % (C/4)) keeps its quad's twelve convolution
// weights, bias and BatchNorm constants in registers and walks the rows of its workgroup's slab - no division, no LDS, no constant reloaded per element
// (a one-float4-per-thread form with the constants in LDS spent ~80 instructions per 16 bytes and streamed at 3 TB/s: 5.71 -> 5.50 ms per step).  A workgroup writes one
// contiguous slab of rows.  BWD = false: y = relu?(bn(x)); BWD = true: dx as in bn_bwd_apply_kernel.
// WG (backward only): the convolution's OWN weight / bias gradient is collected in the same pass - dwg[c][kh] = sum_r dx(r, c) a0[2r - 1 + kh], dcbias[c] = sum_r dx(r, c)
// (layer 0 has no data gradient, so dx has no other reader: with out == nullptr it is never written - 302 MB at B = 1024 - and the separate weight-gradient
// kernel that read it back is gone).  Per thread 16 fp32 accumulators over its rows, the row-offset threads of a channel quad added in order through LDS, one
// slab row [q][12 weights | 4 biases] per workgroup in wpart; bn_conv0_wgrad_fold_kernel adds the slab rows in double, fixed order.
template <bool BWD, bool WG = false>
__global__ __launch_bounds__(BN_THREADS) void bn_conv0_rows_kernel(const BnSrc src, const float* __restrict__ dy, int64_t R, int C, float invR,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta, int training, int relu,
                                                                    float* __restrict__ out, float* __restrict__ wpart = nullptr) {
  static_assert(BWD || !WG, "the weight gradient belongs to the backward pass");
  __shared__ float s_wg[WG ? BN_THREADS * 16 : 1];
  const int t = threadIdx.x;
  const int Q = C >> 2, rpp = BN_THREADS / Q;
  const int q = t % Q, ro = t / Q;
  if (!WG && ro >= rpp) return;
  float aw[12], ab[4];
#pragma unroll
  for (int e = 0; e < 12; ++e) aw[e] = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) ab[e] = 0.f;
  const int64_t rows_per_block = (R + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  float w12[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) w12[e] = src.wg[12 * q + e];
  const float4 cb4 = *reinterpret_cast<const float4*>(src.cbias + 4 * q);
  float gi[4], shf[4], mu[4], is[4], db[4], dg[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    gi[e] = bn_scale(invstd[c], gamma[c]);                    // the forward's scale, bit for bit
    shf[e] = bn_shift(beta[c], mean[c], gi[e]);
    mu[e] = mean[c];
    is[e] = invstd[c];
    db[e] = (BWD && training) ? dbeta[c] * invR : 0.f;
    dg[e] = (BWD && training) ? dgamma[c] * invR : 0.f;
  }
  for (int64_t r = r0 + ro; r < (ro < rpp ? r1 : r0); r += rpp) {
    const float4 xv = conv0_eval4(src.a0, r, src.P_out, w12, cb4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
    if constexpr (!BWD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = bn_affine(xs[e], gi[e], shf[e]);
        o[e] = relu ? fmaxf(v, 0.f) : v;
      }
    } else {
      const float4 gv = *reinterpret_cast<const float4*>(dy + r * C + 4 * q);
      const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g = gs[e];
        if (relu && !(bn_affine(xs[e], gi[e], shf[e]) > 0.f)) g = 0.f;
        const float xhat = (xs[e] - mu[e]) * is[e];
        o[e] = gi[e] * (g - db[e] - xhat * dg[e]);
      }
    }
    if (!WG || out) *reinterpret_cast<float4*>(out + r * C + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    if constexpr (WG) {
      const float i1 = src.a0[2 * r], i2 = src.a0[2 * r + 1];
      const float i0 = (r % src.P_out) ? src.a0[2 * r - 1] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        aw[3 * e] = fmaf(o[e], i0, aw[3 * e]);
        aw[3 * e + 1] = fmaf(o[e], i1, aw[3 * e + 1]);
        aw[3 * e + 2] = fmaf(o[e], i2, aw[3 * e + 2]);
        ab[e] += o[e];
      }
    }
  }
  if constexpr (WG) {
#pragma unroll
    for (int e = 0; e < 12; ++e) s_wg[t * 16 + e] = aw[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) s_wg[t * 16 + 12 + e] = ab[e];
    __syncthreads();
    for (int idx = t; idx < Q * 16; idx += BN_THREADS) {
      const int qq = idx >> 4, e = idx & 15;
      float v = 0.f;
      for (int o2 = 0; o2 < rpp; ++o2) v += s_wg[(o2 * Q + qq) * 16 + e];
      wpart[(int64_t)blockIdx.x * Q * 16 + idx] = v;
    }
  }
}

// dwg (C, 3), dcbias (C) from the slab rows of bn_conv0_rows_kernel<true, true>: one workgroup per 16 slab columns, thread (column t & 15, phase t >> 4) adds
// its blocks in double, the sixteen phases are then added in order
__global__ __launch_bounds__(256) void bn_conv0_wgrad_fold_kernel(const float* __restrict__ wpart, int nblk, int Q, float* __restrict__ dwg, float* __restrict__ dcbias) {
  __shared__ double sh[16][17];
  const int e = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int qq = blockIdx.x;                                  // one channel quad per workgroup: 16 columns [12 weights | 4 biases]
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;            // four chains: the loads of a trip are in flight together
  int b = ph;
  for (; b + 48 < nblk; b += 64) {
    const float x0 = wpart[((int64_t)b * Q + qq) * 16 + e], x1 = wpart[((int64_t)(b + 16) * Q + qq) * 16 + e];
    const float x2 = wpart[((int64_t)(b + 32) * Q + qq) * 16 + e], x3 = wpart[((int64_t)(b + 48) * Q + qq) * 16 + e];
    v0 += (double)x0; v1 += (double)x1; v2 += (double)x2; v3 += (double)x3;
  }
  for (; b < nblk; b += 16) v0 += (double)wpart[((int64_t)b * Q + qq) * 16 + e];
  const double v = (v0 + v1) + (v2 + v3);
  sh[ph][e] = v;
  __syncthreads();
  if (ph != 0) return;
  double tot = 0.0;
  for (int p2 = 0; p2 < 16; ++p2) tot += sh[p2][e];
  if (e < 12) dwg[12 * qq + e] = (float)tot;                 // (c = 4 qq + e / 3, kh = e % 3): the (C, 3) layout itself
  else dcbias[4 * qq + (e - 12)] = (float)tot;
}

}  // namespace a2f

extern "C" size_t a2f_bn_ws_bytes(int64_t R, int64_t C) {
  if (R <= 0 || C <= 0) return 16;
  return (size_t)a2f::bn_blocks(R, (int)C) * 2 * (size_t)C * sizeof(double);
}

// conv0: x is recomputed from (a0, wg, cbias, P_out) - a2f_bn_fwd_conv0 / a2f_bn_bwd_conv0 (C % 4 == 0 required there)
static int bn_fwd_impl(const a2f::BnSrc src, bool conv0, int64_t R, int64_t C, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, int32_t training, int32_t relu, float* y, float* save_mean, float* save_invstd,
                       void* ws, size_t ws_bytes, a2f_stream_t stream) {
  using namespace a2f;
  const float* x = src.x;
  A2F_CHECK_ARG((x || conv0) && gamma && beta && running_mean && running_var && y && save_mean && save_invstd, "bn_fwd: null pointer");
  A2F_CHECK_ARG(R > 0 && C > 0 && C <= BN_MAX_C, "bn_fwd: R=%lld C=%lld (C <= %d)", (long long)R, (long long)C, BN_MAX_C);
  hipStream_t s = (hipStream_t)stream;
  const int cb = (int)cdiv(C, 256);
  if (training) {
    A2F_CHECK_ARG(ws != nullptr, "bn_fwd: null workspace");
    if (ws_bytes < a2f_bn_ws_bytes(R, C)) return set_err(A2F_EWORKSPACE, "bn_fwd: workspace %zu < %zu", ws_bytes, a2f_bn_ws_bytes(R, C));
    const int nblk = bn_blocks(R, (int)C);
    if (conv0)
      hipLaunchKernelGGL((bn_partial4_kernel<0, true>), dim3(nblk), dim3(BN_THREADS), 0, s, src, (const float*)nullptr, (const float*)nullptr, R,
                         (int)C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (double*)ws);
    else if (bn_vec4_ok(x, nullptr, nullptr, C))
      hipLaunchKernelGGL((bn_partial4_kernel<0, false>), dim3(nblk), dim3(BN_THREADS), 0, s, src, (const float*)nullptr, (const float*)nullptr, R,
                         (int)C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (double*)ws);
    else
      hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(nblk), dim3(BN_THREADS), 0, s, x, (const float*)nullptr, (const float*)nullptr, R,
                         (int)C, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (double*)ws);
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3((unsigned)cdiv(C, BN_FOLD_CH)), dim3(256), 0, s, (const double*)ws, nblk, R, (int)C, momentum, eps,
                       running_mean, running_var, save_mean, save_invstd);
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cb), dim3(256), 0, s, running_mean, running_var, (int)C, eps, save_mean, save_invstd);
  }
  const int64_t total = R * C;
  A2F_CHECK_ARG(total / BN_THREADS < 0x7FFFFFFFll, "bn_fwd: R * C too large for one element per thread");
  if (conv0) {
    const int rpp = BN_THREADS / (int)(C / 4);
    const int64_t blocks = std::min<int64_t>(cdiv(R, (int64_t)rpp * 32), 8192);            // >= 32 rows per thread: the constants are loaded once per ~130 outputs
    hipLaunchKernelGGL((bn_conv0_rows_kernel<false>), dim3((unsigned)blocks), dim3(BN_THREADS), 0, s, src, (const float*)nullptr, R, (int)C, 0.f, gamma, beta,
                       save_mean, save_invstd, (const float*)nullptr, (const float*)nullptr, 0, (int)relu, y);
  } else if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3((unsigned)cdiv(total / 4, BN_THREADS)), dim3(BN_THREADS), 0, s, x, total, (int)C, gamma, beta, save_mean,
                       save_invstd, (int)relu, y);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3((unsigned)cdiv(total, BN_THREADS)), dim3(BN_THREADS), 0, s, x, total, (int)C, gamma, beta, save_mean,
                       save_invstd, (int)relu, y);
  A2F_CHECK_LAUNCH("bn_fwd");
  return A2F_OK;
}

static int bn_bwd_impl(const a2f::BnSrc src, bool conv0, const float* beta, const float* dy, int64_t R, int64_t C, const float* gamma,
                       const float* save_mean, const float* save_invstd, int32_t training, int32_t relu, float* dx, float* dgamma, float* dbeta, void* ws,
                       size_t ws_bytes, a2f_stream_t stream, float* dwg = nullptr, float* dcbias = nullptr) {
  using namespace a2f;
  const float* x = src.x;
  const bool wgrad = dwg != nullptr;
  A2F_CHECK_ARG((x || conv0) && dy && gamma && save_mean && save_invstd && (dx || wgrad) && dgamma && dbeta && ws && (beta || !relu), "bn_bwd: null pointer (beta is needed with relu)");
  A2F_CHECK_ARG(!wgrad || (conv0 && dcbias), "bn_bwd: the fused weight gradient exists for the conv0 form only and needs dcbias");
  A2F_CHECK_ARG(R > 0 && C > 0 && C <= BN_MAX_C, "bn_bwd: R=%lld C=%lld (C <= %d)", (long long)R, (long long)C, BN_MAX_C);
  if (ws_bytes < a2f_bn_ws_bytes(R, C)) return set_err(A2F_EWORKSPACE, "bn_bwd: workspace %zu < %zu", ws_bytes, a2f_bn_ws_bytes(R, C));
  hipStream_t s = (hipStream_t)stream;
  const int nblk = bn_blocks(R, (int)C);
  const bool vec_stats = (((uintptr_t)save_mean | (uintptr_t)save_invstd | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
  if (conv0) {
    A2F_CHECK_ARG(vec_stats && bn_vec4_ok(dy, nullptr, dx, C), "bn_bwd_conv0: dy, dx, gamma, beta and the saved statistics must be 16-byte aligned");
    hipLaunchKernelGGL((bn_partial4_kernel<1, true>), dim3(nblk), dim3(BN_THREADS), 0, s, src, beta, dy, R, (int)C, save_mean, save_invstd, gamma, (int)relu,
                       (double*)ws);
  } else if (bn_vec4_ok(x, nullptr, dy, C) && vec_stats)
    hipLaunchKernelGGL((bn_partial4_kernel<1, false>), dim3(nblk), dim3(BN_THREADS), 0, s, src, beta, dy, R, (int)C, save_mean, save_invstd, gamma, (int)relu,
                       (double*)ws);
  else
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(nblk), dim3(BN_THREADS), 0, s, x, beta, dy, R, (int)C, save_mean, save_invstd, gamma, (int)relu,
                       (double*)ws);
  hipLaunchKernelGGL(bn_finish_bwd_kernel, dim3((unsigned)cdiv(C, BN_FOLD_CH)), dim3(256), 0, s, (const double*)ws, nblk, (int)C, dgamma, dbeta);
  const int64_t total = R * C;
  A2F_CHECK_ARG(total / BN_THREADS < 0x7FFFFFFFll, "bn_bwd: R * C too large for one element per thread");
  if (conv0) {
    const int rpp = BN_THREADS / (int)(C / 4);
    int64_t blocks = std::min<int64_t>(cdiv(R, (int64_t)rpp * 32), 8192);
    if (!wgrad) {
      hipLaunchKernelGGL((bn_conv0_rows_kernel<true>), dim3((unsigned)blocks), dim3(BN_THREADS), 0, s, src, dy, R, (int)C, 1.0f / (float)R, gamma, beta, save_mean,
                         save_invstd, (const float*)dgamma, (const float*)dbeta, (int)training, (int)relu, dx, (float*)nullptr);
    } else {
      // the slab rows of the weight gradient take over the workspace (the statistics partials were consumed by bn_finish_bwd_kernel, earlier on this stream)
      blocks = std::min<int64_t>(blocks, (int64_t)(ws_bytes / ((size_t)C * 4 * sizeof(float))));
      A2F_CHECK_ARG(blocks >= 1, "bn_bwd_conv0: workspace too small for one weight-gradient slab row");
      hipLaunchKernelGGL((bn_conv0_rows_kernel<true, true>), dim3((unsigned)blocks), dim3(BN_THREADS), 0, s, src, dy, R, (int)C, 1.0f / (float)R, gamma, beta,
                         save_mean, save_invstd, (const float*)dgamma, (const float*)dbeta, (int)training, (int)relu, dx, (float*)ws);
      hipLaunchKernelGGL(bn_conv0_wgrad_fold_kernel, dim3((unsigned)(C / 4)), dim3(256), 0, s, (const float*)ws, (int)blocks, (int)(C / 4), dwg, dcbias);
    }
  } else if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3((unsigned)cdiv(total / 4, BN_THREADS)), dim3(BN_THREADS), 0, s, x, beta, dy, total, (int)C, 1.0f / (float)R,
                       gamma, save_mean, save_invstd, dgamma, dbeta, (int)training, (int)relu, dx);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3((unsigned)cdiv(total, BN_THREADS)), dim3(BN_THREADS), 0, s, x, beta, dy, total, (int)C, 1.0f / (float)R,
                       gamma, save_mean, save_invstd, dgamma, dbeta, (int)training, (int)relu, dx);
  A2F_CHECK_LAUNCH("bn_bwd");
  return A2F_OK;
}

extern "C" int a2f_bn_fwd(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float eps, int32_t training, int32_t relu, float* y,
                          float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  return bn_fwd_impl(a2f::BnSrc{x, nullptr, nullptr, nullptr, 1}, false, R, C, gamma, beta, running_mean, running_var, momentum, eps, training, relu, y, save_mean,
                     save_invstd, ws, ws_bytes, stream);
}

extern "C" int a2f_bn_bwd(const float* x, const float* beta, const float* dy, int64_t R, int64_t C, const float* gamma,
                          const float* save_mean, const float* save_invstd, int32_t training, int32_t relu, float* dx,
                          float* dgamma, float* dbeta, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  return bn_bwd_impl(a2f::BnSrc{x, nullptr, nullptr, nullptr, 1}, false, beta, dy, R, C, gamma, save_mean, save_invstd, training, relu, dx, dgamma, dbeta, ws, ws_bytes,
                     stream);
}

static int bn_conv0_check(const float* a0, const float* wg, const float* cbias, int64_t P_out, int64_t R, int64_t C, const float* gamma, const float* beta,
                          const float* y_or_dy, const char* who) {
  A2F_CHECK_ARG(a0 && wg && cbias, "%s: null convolution operand", who);
  A2F_CHECK_ARG(P_out > 0 && R % P_out == 0 && C % 4 == 0 && C <= a2f::BN_THREADS, "%s: R=%lld must be a multiple of P_out=%lld, C=%lld a multiple of 4 (<= %d)", who,
                (long long)R, (long long)P_out, (long long)C, a2f::BN_THREADS);
  A2F_CHECK_ARG((((uintptr_t)cbias | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y_or_dy) & 15) == 0, "%s: cbias, gamma, beta and the (R, C) tensors must be 16-byte aligned", who);
  return A2F_OK;
}

extern "C" int a2f_bn_fwd_conv0(const float* a0, const float* wg, const float* cbias, int64_t P_out, int64_t R, int64_t C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int32_t training, int32_t relu, float* y,
                                float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  if (int rc = bn_conv0_check(a0, wg, cbias, P_out, R, C, gamma, beta, y, "bn_fwd_conv0")) return rc;
  return bn_fwd_impl(a2f::BnSrc{nullptr, a0, wg, cbias, (int)P_out}, true, R, C, gamma, beta, running_mean, running_var, momentum, eps, training, relu, y, save_mean,
                     save_invstd, ws, ws_bytes, stream);
}

extern "C" int a2f_bn_bwd_conv0(const float* a0, const float* wg, const float* cbias, int64_t P_out, const float* beta, const float* dy, int64_t R, int64_t C,
                                const float* gamma, const float* save_mean, const float* save_invstd, int32_t training, int32_t relu, float* dx,
                                float* dgamma, float* dbeta, float* dwg, float* dcbias, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  if (int rc = bn_conv0_check(a0, wg, cbias, P_out, R, C, gamma, beta, dy, "bn_bwd_conv0")) return rc;
  A2F_CHECK_ARG((dwg == nullptr) == (dcbias == nullptr), "bn_bwd_conv0: dwg and dcbias come together");
  A2F_CHECK_ARG(dx == nullptr || (((uintptr_t)dx) & 15) == 0, "bn_bwd_conv0: dx must be 16-byte aligned");
  return bn_bwd_impl(a2f::BnSrc{nullptr, a0, wg, cbias, (int)P_out}, true, beta, dy, R, C, gamma, save_mean, save_invstd, training, relu, dx, dgamma, dbeta, ws, ws_bytes,
                     stream, dwg, dcbias);
}
