// Shared pieces of the bf16-MFMA GEMM family (csrc/gemm_mfma.hip): kernel argument block, tile order, epilogue, split / fragment helpers.
// Contract: include/a2f_hip.h (a2f_mm).
#pragma once
#include "a2f_common.h"
#include <stdlib.h>

namespace a2f {
namespace mm {

using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
using u32x2_t = __attribute__((ext_vector_type(2))) unsigned;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
using i16x4_t = __attribute__((ext_vector_type(4))) short;
using rsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int BK = 32;
#ifndef MM_TPAD
#define MM_TPAD 64            // bytes of padding per row of a [k][lines] image (64: transposing reads conflict-free)
#endif
constexpr int THREADS = 256;           // CFG 0 / 1; CFG 2 runs 512 threads (see gemm_mfma_kernel)

struct Op {
  const float* p;
  unsigned bytes;
  int ld, lps, off, seq_elems;
  int lps_shift;         // log2(lps) when lps is a power of two, else -1 (short sequences: division -> shift)
  long long seq_stride;
  long long bstride;
};

struct Args {
  Op A, B;
  float* C;
  float* C2;
  const float* bias;
  const float* aux;
  float* ws;
  long long ldc, c_seq_stride, c_bstride, bias_bstride, ldaux, aux_bstride;
  int c_rps;
  int M, N, K;
  int act, aux_mode, accumulate, split_k, batch;
  float alpha;
  float* a_rowsum;               // optional (A trans, plain operands, batch 1): a_rowsum[m] = sum_k A(m, k) - the bias gradient of a weight-gradient product
  int group_n;                   // tile order: 0 = row-major (m-tile, n-tile); > 0 = panels of group_n n-tiles, m fastest across a panel (tile_origin)
  int epi32;                     // every byte offset the epilogue forms (C, C2, aux, keep words, split-K slab) fits 32 bits: buffer-resource epilogue
  const unsigned* keep;          // dropout keep bits of C (csrc/dropout.hip: word (m, n >> 5), bit n & 31), applied after the activation
  long long keep_wpr;
  float keep_scale;
  float* c_colsum;                // image-operand k-contiguous products: c_colsum[n] = sum_m C[m][n] (partials in ws, folded in fixed order)
  int c_format;                   // 0: C is fp32; 1: C is a split-32 image (csrc/gemm_img.hip; image-operand kernels only)
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// one output element through the epilogue (shared by the GEMM kernel and the split-K fold)
__device__ __forceinline__ void epilogue_store(const Args& a, int bz, int m, int n, long long crow, float v) {
  v *= a.alpha;
  if (a.bias) v += a.bias[bz * a.bias_bstride + n];
  const long long ci = bz * a.c_bstride + crow + n;
  if (a.act == 4) {                        // GELU with its derivative saved for the backward (C2 = gelu'(v) instead of v)
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
    if (a.C2) a.C2[ci] = cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v);
    v *= cdf;
  } else {
    if (a.C2) a.C2[ci] = v;
    if (a.act == 1) v = fmaxf(v, 0.f);
    else if (a.act == 3) v = gelu_f(v);
  }
  if (a.keep) v = ((a.keep[(long long)m * a.keep_wpr + (n >> 5)] >> (n & 31)) & 1u) ? v * a.keep_scale : 0.f;
  if (a.aux_mode == 4) {
    v *= a.aux[ci];                        // aux laid out like C (same row map): a saved activation derivative
  } else if (a.aux_mode) {
    const float x = a.aux[bz * a.aux_bstride + (long long)m * a.ldaux + n];
    if (a.aux_mode == 1) v += x;
    else if (a.aux_mode == 2) v *= gelu_grad_f(x);
    else v *= (x > 0.f ? 1.f : 0.f);
  }
  if (a.accumulate) v += a.C[ci];
  a.C[ci] = v;
}

__device__ __forceinline__ long long c_row_offset(const Args& a, int m) {
  if (a.c_rps <= 0) return (long long)m * a.ldc;
  return (long long)(m / a.c_rps) * a.c_seq_stride + (long long)(m % a.c_rps) * a.ldc;
}

// Workgroup -> output tile.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8 labels the XCD group); the linear id is first
// remapped so that every XCD owns ONE contiguous run of the tile order (speed only: any placement computes the same tiles), then the
// order itself is either row-major - the n-tiles that share an A panel run together, right when the whole B operand fits the XCD's 4 MB
// L2 - or PANEL-major: n-tiles in groups of `group_n`, all m-tiles of a group before the next group.  With a wide B (N x K x 4 bytes
// beyond the L2: the 768 -> 2304 / 3072 projections) row-major streams the whole of B through every XCD once per ~3 m-tiles - PMC on
// 38400 x 3072 x 768: 2.7 GB fetched for 0.13 GB of operands, the kernel ran at the fabric's rate, not the matrix pipe's - whereas a panel's
// slice of B (<= 2 MB) stays resident while the A panels stream past it once per panel.
__device__ __forceinline__ void tile_origin(const Args& a, int BM, int BN, int& m0, int& n0) {
  const int tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  int tm, tn;
  if (a.group_n <= 0) {
    tm = wgid / tiles_n;
    tn = wgid - tm * tiles_n;
  } else {
    const int full = (tiles_n / a.group_n) * a.group_n;               // n-tiles inside full panels
    const int per_panel = tiles_m * a.group_n;
    if (wgid < (full / a.group_n) * per_panel) {
      const int p = wgid / per_panel, rem = wgid - p * per_panel;
      tm = rem / a.group_n;
      tn = p * a.group_n + (rem - tm * a.group_n);
    } else {                                                            // the last, narrower panel
      const int gw = tiles_n - full, rem = wgid - (full / a.group_n) * per_panel;
      tm = rem / gw;
      tn = full + (rem - tm * gw);
    }
  }
  m0 = tm * BM;
  n0 = tn * BN;
}

// NB: bit-cast the WHOLE vector first.  __builtin_bit_cast(float, raw.y) on a vector-element lvalue reads element 0 with
// hipcc 7.2 (all four values came out as raw.x).
__device__ __forceinline__ void split4(const u32x4_t raw, u32x2_t& hi, u32x2_t& lo) {
  const f32x4 f = __builtin_bit_cast(f32x4, raw);
  const f32x2_t v0 = {f.x, f.y};
  const f32x2_t v1 = {f.z, f.w};
  const bf16x2_t h0 = __builtin_convertvector(v0, bf16x2_t);
  const bf16x2_t h1 = __builtin_convertvector(v1, bf16x2_t);
  const f32x2_t r0 = v0 - __builtin_convertvector(h0, f32x2_t);
  const f32x2_t r1 = v1 - __builtin_convertvector(h1, f32x2_t);
  hi.x = __builtin_bit_cast(unsigned, h0);
  hi.y = __builtin_bit_cast(unsigned, h1);
  lo.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf16x2_t));
  lo.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
}
// x = hi + mid + lo (three bf16 pieces = 24 significand bits: every fp32 value is represented exactly unless it underflows)
__device__ __forceinline__ void split4x3(const u32x4_t raw, u32x2_t& hi, u32x2_t& mid, u32x2_t& lo) {
  const f32x4 f = __builtin_bit_cast(f32x4, raw);
  const f32x2_t v0 = {f.x, f.y};
  const f32x2_t v1 = {f.z, f.w};
  const bf16x2_t h0 = __builtin_convertvector(v0, bf16x2_t);
  const bf16x2_t h1 = __builtin_convertvector(v1, bf16x2_t);
  const f32x2_t r0 = v0 - __builtin_convertvector(h0, f32x2_t);
  const f32x2_t r1 = v1 - __builtin_convertvector(h1, f32x2_t);
  const bf16x2_t m0 = __builtin_convertvector(r0, bf16x2_t);
  const bf16x2_t m1 = __builtin_convertvector(r1, bf16x2_t);
  const f32x2_t q0 = r0 - __builtin_convertvector(m0, f32x2_t);
  const f32x2_t q1 = r1 - __builtin_convertvector(m1, f32x2_t);
  hi.x = __builtin_bit_cast(unsigned, h0);
  hi.y = __builtin_bit_cast(unsigned, h1);
  mid.x = __builtin_bit_cast(unsigned, m0);
  mid.y = __builtin_bit_cast(unsigned, m1);
  lo.x = __builtin_bit_cast(unsigned, __builtin_convertvector(q0, bf16x2_t));
  lo.y = __builtin_bit_cast(unsigned, __builtin_convertvector(q1, bf16x2_t));
}
__device__ __forceinline__ u32x2_t cvt4(const u32x4_t raw) {
  const f32x4 f = __builtin_bit_cast(f32x4, raw);
  const f32x2_t v0 = {f.x, f.y};
  const f32x2_t v1 = {f.z, f.w};
  u32x2_t r;
  r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf16x2_t));
  r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf16x2_t));
  return r;
}
__device__ __forceinline__ u32x2_t tr_read(const char* lds_addr) {
  return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)(lds_addr)));
}
__device__ __forceinline__ f32x16 mfma(u32x4_t a, u32x4_t b, f32x16 c) {
#ifdef MM_ABL_NOMFMA        // timing ablation only: operands stay live, no matrix instruction
  asm volatile("" ::"v"(a), "v"(b));
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}

// ---- epilogue of one workgroup tile: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// One 32-row band (accumulator row index I) per instantiation: the accumulator indices stay compile-time constants whatever the unroller
// decides about the surrounding loops (a runtime index into the f32x16 array sends the whole accumulator set to scratch: with four bands
// per wave the fully unrolled form was too large for the unroller and the 256 x 256 tile ran 5x slower).
template <int TM, int TN, int I>
__device__ __forceinline__ void tile_epilogue_band(const Args& a, const f32x16 (&acc)[TM][TN], int m0, int wm, int h, int bz, const int (&ncol)[TN],
                                                   const float (&bv)[TN], bool generic) {
#ifdef MM_ABL_NOEPI         // timing ablation only (wrong results): the accumulators stay live, nothing is stored
#pragma unroll
  for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[I][j]));
  if constexpr (I + 1 < TM) tile_epilogue_band<TM, TN, I + 1>(a, acc, m0, wm, h, bz, ncol, bv, generic);
  return;
#endif
  if (a.epi32) {
    // Buffer-resource epilogue: wave-uniform bases in SGPRs, one 32-bit byte offset per element, out-of-tile elements get offset 2^31
    // (= num_records: dropped / zero-filled by the hardware range check) - no 64-bit address arithmetic and no branch per element (the pointer form below
    // compiles to a compare-and-branch around every one of the 32 stores of a band).
    constexpr unsigned NOWHERE = 0x80000000u;          // == num_records: out of range without relying on how offset + size wraps
    auto rsrc_of = [](const void* p) {
      const unsigned long long v = (unsigned long long)p;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)0x80000000u, 0x00020000);
    };
    if (a.split_k > 1) {
      const rsrc_t rw = rsrc_of(a.ws + (long long)blockIdx.z * a.M * a.N);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + I) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const unsigned rowoff = (unsigned)m * (unsigned)a.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned off = (m < a.M && ncol[j] < a.N) ? (rowoff + (unsigned)ncol[j]) * 4u : NOWHERE;
          const float w = acc[I][j][r];       // a copy: __builtin_bit_cast applied to the vector-element expression itself reads element 0 whatever r is
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, w), rw, off, 0, 0);
        }
      }
    } else {
      const rsrc_t rc = rsrc_of(a.C + bz * a.c_bstride);
      const rsrc_t rc2 = rsrc_of(a.C2 ? a.C2 + bz * a.c_bstride : a.C);
      const rsrc_t rax = rsrc_of(a.aux ? (a.aux_mode == 4 ? a.aux + bz * a.c_bstride : a.aux + bz * a.aux_bstride) : a.C);
      const rsrc_t rk = rsrc_of(a.keep ? (const void*)a.keep : (const void*)a.C);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + I) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool mok = m < a.M;
        unsigned crow;
        if (a.c_rps <= 0) crow = (unsigned)m * (unsigned)a.ldc;
        else { const int sq = m / a.c_rps; crow = (unsigned)sq * (unsigned)a.c_seq_stride + (unsigned)(m - sq * a.c_rps) * (unsigned)a.ldc; }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bool ok = mok && ncol[j] < a.N;
          const unsigned off = ok ? (crow + (unsigned)ncol[j]) * 4u : NOWHERE;
          float v = acc[I][j][r] * a.alpha + bv[j];
          if (generic) {
            if (a.act == 4) {
              const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
              if (a.C2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v)), rc2, off, 0, 0);
              v *= cdf;
            } else {
              if (a.C2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc2, off, 0, 0);
              if (a.act == 1) v = fmaxf(v, 0.f);
              else if (a.act == 3) v = gelu_f(v);
            }
            if (a.keep) {
              const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rk, ok ? ((unsigned)m * (unsigned)a.keep_wpr + (unsigned)(ncol[j] >> 5)) * 4u : NOWHERE, 0, 0);
              v = ((w >> (ncol[j] & 31)) & 1u) ? v * a.keep_scale : 0.f;
            }
            if (a.aux_mode == 4) {
              v *= __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rax, off, 0, 0));
            } else if (a.aux_mode) {
              const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rax, ok ? ((unsigned)m * (unsigned)a.ldaux + (unsigned)ncol[j]) * 4u : NOWHERE, 0, 0));
              if (a.aux_mode == 1) v += x;
              else if (a.aux_mode == 2) v *= gelu_grad_f(x);
              else v *= (x > 0.f ? 1.f : 0.f);
            }
            if (a.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off, 0, 0));
          }
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, off, 0, 0);
        }
      }
    }
  } else if (a.split_k > 1) {
    float* wsz = a.ws + (long long)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * TM + I) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (m < a.M && ncol[j] < a.N) wsz[(long long)m * a.N + ncol[j]] = acc[I][j][r];
    }
  } else {
    float* Cz = a.C + bz * a.c_bstride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * TM + I) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m >= a.M) continue;
      const long long crow = c_row_offset(a, m);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (ncol[j] >= a.N) continue;
        float v = acc[I][j][r] * a.alpha + bv[j];
        if (generic) {
          if (a.act == 4) {
            const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
            if (a.C2) a.C2[bz * a.c_bstride + crow + ncol[j]] = cdf + v * 0.3989422804014327f * __expf(-0.5f * v * v);
            v *= cdf;
          } else {
            if (a.C2) a.C2[bz * a.c_bstride + crow + ncol[j]] = v;
            if (a.act == 1) v = fmaxf(v, 0.f);
            else if (a.act == 3) v = gelu_f(v);
          }
          if (a.keep) v = ((a.keep[(long long)m * a.keep_wpr + (ncol[j] >> 5)] >> (ncol[j] & 31)) & 1u) ? v * a.keep_scale : 0.f;
          if (a.aux_mode == 4) {
            v *= a.aux[bz * a.c_bstride + crow + ncol[j]];
          } else if (a.aux_mode) {
            const float x = a.aux[bz * a.aux_bstride + (long long)m * a.ldaux + ncol[j]];
            if (a.aux_mode == 1) v += x;
            else if (a.aux_mode == 2) v *= gelu_grad_f(x);
            else v *= (x > 0.f ? 1.f : 0.f);
          }
          if (a.accumulate) v += Cz[crow + ncol[j]];
        }
        Cz[crow + ncol[j]] = v;
      }
    }
  }
  if constexpr (I + 1 < TM) tile_epilogue_band<TM, TN, I + 1>(a, acc, m0, wm, h, bz, ncol, bv, generic);
}

template <int TM, int TN>
__device__ __forceinline__ void tile_epilogue(const Args& a, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, int bz) {
  const int h = lane >> 5;
  int ncol[TN];
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    ncol[j] = n0 + (wn * TN + j) * 32 + (lane & 31);
    bv[j] = (a.bias && a.split_k <= 1 && ncol[j] < a.N) ? a.bias[bz * a.bias_bstride + ncol[j]] : 0.f;
  }
  const bool generic = a.C2 || a.act || a.aux_mode || a.accumulate || a.keep;     // wave-uniform
  tile_epilogue_band<TM, TN, 0>(a, acc, m0, wm, h, bz, ncol, bv, generic);
}


}  // namespace mm
}  // namespace a2f
