// Train-mode dropout of the wav2vec2 encoder behind FaceFormer (HF Wav2Vec2Model as wrapped by
// /root/reference/src/model/wav2vec.py:87-189; the reference trains it in train mode: feat_proj_dropout / hidden_dropout /
// attention_dropout / activation_dropout = 0.1 in facebook/wav2vec2-base-960h's configuration).
//
// Every dropout site is a bit-packed KEEP mask in HBM (1 bit per element: 1/32 of the fp32 traffic of the tensor it gates):
//   * produced once per step by a counter-based generator (Philox4x32-10 keyed by the step seed; counter = (group of eight
//     elements, row, site)), so a mask is a pure function of (seed, site, element) - reproducible, order-independent;
//   * consumed where the gated tensor is produced or used: in the a2f_mm epilogue (csrc/gemm_mfma.hip), inside the
//     flash-attention kernels on the probabilities (csrc/attention.hip), or by the streaming a2f_dropout_apply below;
//   * or injected: the parity tests feed the masks recorded from the reference's own nn.functional.dropout calls, bit for
//     bit (tests/golden/gen_golden.py G13), so the train-mode function is pinned although it is stochastic.
// Layouts: a matrix (R, C) has R rows of `wpr` 32-bit words, bit b of word (r, w) <-> element (r, 32 w + b).  Attention
// probabilities (SH, T, T) are kept twice, because the kernels hold them in two orientations: query-major words
// [sh][q][TW] (bit = key) for the forward / dQ kernels (query on the lane) and key-major words [sh][TW][32 TW] (word
// (qw, key), bit = query 32 qw + b) for the dK / dV kernel (key on the lane); TW = 2 ceil(T / 64).
#include "a2f_common.h"

namespace a2f {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the 32 keep bits of elements (row, 32 w .. 32 w + 31) of a row of `cols` elements.  One Philox call yields four 32-bit words = EIGHT 16-bit randoms:
// element 8 g + 2 e takes the low half of word e of group g = 4 w + j, element 8 g + 2 e + 1 the high half; an element keeps iff its 16-bit random
// >= thresh16 = round(p 2^16) (p = 0.1: 6554 / 65536 = 0.100006).  Half the Philox work of one 32-bit word per element: the mask kernels are
// bound by exactly that arithmetic (round 4: 5.2 ms of the 64 x 600 FaceFormer step with 32-bit randoms).
__device__ __forceinline__ unsigned keep_word(unsigned k0, unsigned k1, unsigned site, unsigned row, int w, int cols, unsigned thresh16) {
  unsigned word = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned r[4];
    philox4x32_10((unsigned)(4 * w + j), row, site, 0u, k0, k1, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      word |= ((r[e] & 0xFFFFu) >= thresh16 ? 1u : 0u) << (8 * j + 2 * e);
      word |= ((r[e] >> 16) >= thresh16 ? 1u : 0u) << (8 * j + 2 * e + 1);
    }
  }
  const int left = cols - 32 * w;
  if (left < 32) word = left <= 0 ? 0u : (word & ((1u << left) - 1u));
  return word;
}

// seed_dev (optional): the 64-bit step seed lives in device memory - a hipGraph-captured train step replays the same launch with a fresh seed written
// into that word before every replay (a seed passed by value would be baked into the captured node)
__global__ __launch_bounds__(256) void keep_bits_kernel(unsigned k0, unsigned k1, unsigned site, long long R, int C, int wpr, unsigned thresh,
                                                        unsigned* __restrict__ out, const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) { const unsigned long long s = *seed_dev; k0 = (unsigned)s; k1 = (unsigned)(s >> 32); }
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * wpr) return;
  const long long r = idx / wpr;
  const int w = (int)(idx - r * wpr);
  out[idx] = keep_word(k0, k1, site, (unsigned)r, w, C, thresh);
}

// grid (TW / 4, ceil(T / 64), SH), 256 threads: wave <-> key word kw, lane <-> query
__global__ __launch_bounds__(256) void keep_bits_attn_kernel(unsigned k0, unsigned k1, unsigned site, int T, int TW, unsigned thresh,
                                                             unsigned* __restrict__ qmaj, unsigned* __restrict__ kmaj, const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) { const unsigned long long s = *seed_dev; k0 = (unsigned)s; k1 = (unsigned)(s >> 32); }
  const int lane = threadIdx.x & 63, kw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q = blockIdx.y * 64 + lane;
  const long long sh = blockIdx.z;
  if (kw >= TW) return;                                                    // wave-uniform
  unsigned word = 0;
  if (q < T) {
    word = keep_word(k0, k1, site, (unsigned)(sh * T + q), kw, T, thresh);
    qmaj[(sh * T + q) * TW + kw] = word;
  }
  // transpose the 64 x 32 bit block held by the wave: ballot j = the column of bit j over the 64 queries
  unsigned tw = 0;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const unsigned long long b = __ballot((word >> j) & 1u);
    const unsigned mine = lane < 32 ? (unsigned)b : (unsigned)(b >> 32);
    tw = (lane & 31) == j ? mine : tw;
  }
  const int qw = blockIdx.y * 2 + (lane >> 5);                             // < TW (TW is even)
  kmaj[(sh * TW + qw) * (32ll * TW) + 32 * kw + (lane & 31)] = tw;
}

__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, const unsigned* __restrict__ bits, long long R, int C4, long long ld,
                                                            int wpr, float scale, float* __restrict__ y) {
  const long long total = R * C4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const unsigned w = bits[r * wpr + (c >> 5)] >> (c & 31);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + c);
    f32x4 o;
    o.x = (w & 1u) ? v.x * scale : 0.f;
    o.y = (w & 2u) ? v.y * scale : 0.f;
    o.z = (w & 4u) ? v.z * scale : 0.f;
    o.w = (w & 8u) ? v.w * scale : 0.f;
    *reinterpret_cast<f32x4*>(y + r * ld + c) = o;
  }
}

static unsigned keep_threshold(float p) {            // keep iff random u16 >= round(p * 2^16)
  const double t = (double)p * 65536.0 + 0.5;
  return t <= 0.0 ? 0u : t >= 65535.0 ? 65535u : (unsigned)t;
}

}  // namespace a2f

static int keep_bits_impl(uint64_t seed, const uint64_t* seed_dev, uint32_t site, int64_t R, int64_t C, float p, uint32_t* bits, int64_t wpr, a2f_stream_t stream);
extern "C" int a2f_keep_bits(uint64_t seed, uint32_t site, int64_t R, int64_t C, float p, uint32_t* bits, int64_t wpr, a2f_stream_t stream) {
  return keep_bits_impl(seed, nullptr, site, R, C, p, bits, wpr, stream);
}
extern "C" int a2f_keep_bits_dev(const uint64_t* seed_dev, uint32_t site, int64_t R, int64_t C, float p, uint32_t* bits, int64_t wpr, a2f_stream_t stream) {
  if (!seed_dev) return a2f::set_err(A2F_EINVAL, "keep_bits_dev: null seed pointer");
  return keep_bits_impl(0, seed_dev, site, R, C, p, bits, wpr, stream);
}
static int keep_bits_impl(uint64_t seed, const uint64_t* seed_dev, uint32_t site, int64_t R, int64_t C, float p, uint32_t* bits, int64_t wpr, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(bits && R > 0 && C > 0 && R < (1ll << 32) && C < (1ll << 31) && wpr * 32 >= C && wpr < (1 << 26), "keep_bits: R=%lld C=%lld wpr=%lld",
                (long long)R, (long long)C, (long long)wpr);
  A2F_CHECK_ARG(p >= 0.f && p < 1.f, "keep_bits: p=%g", (double)p);
  const long long n = R * wpr;
  hipLaunchKernelGGL(keep_bits_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (unsigned)seed, (unsigned)(seed >> 32), site,
                     (long long)R, (int)C, (int)wpr, keep_threshold(p), bits, (const unsigned long long*)seed_dev);
  A2F_CHECK_LAUNCH("keep_bits");
  return A2F_OK;
}

extern "C" int64_t a2f_attn_keep_words(int64_t T) { return 2 * ((T + 63) / 64); }

static int keep_bits_attn_impl(uint64_t seed, const uint64_t* seed_dev, uint32_t site, int64_t SH, int64_t T, float p, uint32_t* qmajor, uint32_t* kmajor, a2f_stream_t stream);
extern "C" int a2f_keep_bits_attn(uint64_t seed, uint32_t site, int64_t SH, int64_t T, float p, uint32_t* qmajor, uint32_t* kmajor, a2f_stream_t stream) {
  return keep_bits_attn_impl(seed, nullptr, site, SH, T, p, qmajor, kmajor, stream);
}
extern "C" int a2f_keep_bits_attn_dev(const uint64_t* seed_dev, uint32_t site, int64_t SH, int64_t T, float p, uint32_t* qmajor, uint32_t* kmajor, a2f_stream_t stream) {
  if (!seed_dev) return a2f::set_err(A2F_EINVAL, "keep_bits_attn_dev: null seed pointer");
  return keep_bits_attn_impl(0, seed_dev, site, SH, T, p, qmajor, kmajor, stream);
}
static int keep_bits_attn_impl(uint64_t seed, const uint64_t* seed_dev, uint32_t site, int64_t SH, int64_t T, float p, uint32_t* qmajor, uint32_t* kmajor, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(qmajor && kmajor && SH > 0 && SH < 65536 && T > 0 && SH * T < (1ll << 32), "keep_bits_attn: SH=%lld T=%lld", (long long)SH, (long long)T);
  A2F_CHECK_ARG(p >= 0.f && p < 1.f, "keep_bits_attn: p=%g", (double)p);
  const int TW = (int)a2f_attn_keep_words(T);
  hipLaunchKernelGGL(keep_bits_attn_kernel, dim3((unsigned)cdiv(TW, 4), (unsigned)cdiv(T, 64), (unsigned)SH), dim3(256), 0, (hipStream_t)stream,
                     (unsigned)seed, (unsigned)(seed >> 32), site, (int)T, TW, keep_threshold(p), qmajor, kmajor, (const unsigned long long*)seed_dev);
  A2F_CHECK_LAUNCH("keep_bits_attn");
  return A2F_OK;
}

extern "C" int a2f_dropout_apply(const float* x, const uint32_t* bits, int64_t R, int64_t C, int64_t ld, int64_t wpr, float scale, float* y,
                                 a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(x && bits && y && R > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C && wpr * 32 >= C, "dropout_apply: R=%lld C=%lld ld=%lld wpr=%lld",
                (long long)R, (long long)C, (long long)ld, (long long)wpr);
  A2F_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "dropout_apply: buffers must be 16-byte aligned");
  long long blocks = cdiv(R * (C / 4), 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bits, (long long)R, (int)(C / 4), (long long)ld,
                     (int)wpr, scale, y);
  A2F_CHECK_LAUNCH("dropout_apply");
  return A2F_OK;
}
