// bf16x3 MFMA GEMM on operands that ARRIVE as split-bf16 images (a2f_mm_operand.format 1): 256 x 256 tiles, LDS-DMA staging, two wave
// groups in ping-pong.  Part of the a2f_mm family (entry point: csrc/gemm_mfma.hip).  What it replaces: the dense contractions of the
// 12 Transformer layers of the wav2vec2 encoder behind FaceFormer - nn.Linear forward, data and weight gradients of HF Wav2Vec2EncoderLayer,
// /root/reference/src/model/wav2vec.py:174-180 - which the reference hands to the vendor BLAS through torch.
//
// Image format ("split-32", 4 bytes per element like the fp32 tensor it stands for): a row of `ld` elements (ld % 32 == 0) is ld / 32 blocks of
// 128 bytes; block b holds the 32 elements k = 32 b .. 32 b + 31 as 32 bf16 `hi` (hi = RNE(x), 64 bytes) followed by 32 bf16 `lo`
// (lo = RNE(x - hi), 64 bytes).  A producer that holds four consecutive elements writes 8 bytes of hi and 8 bytes of lo; a 32-deep k-tile of one
// row is ONE 128-byte line - the unit the LDS-DMA moves.  Nothing is converted on chip (the fp32 path spends 80 of its 161 vector
// instructions per wave and k-tile on fp32 -> hi / lo, repeated by every consumer of an activation: DESIGN 3.3).
//
// NT kernel (both operands k-contiguous: Linear forward; data gradients through a transposed weight image):
//   * workgroup = 256 x 256 of C, 8 waves as 2 (m) x 4 (n), a wave owns 128 x 64 = 4 x 2 accumulator tiles of v_mfma_f32_32x32x16_bf16;
//     the MFMA's A operand is the tile of B (n) and its B operand the tile of A (m), so a lane's four consecutive accumulator registers are four
//     consecutive COLUMNS of one row of C: 16-byte stores / 8 + 8-byte image stores, bias / residual / saved derivative as float4, no shuffles;
//   * a k-tile (32 deep) of the 512 operand rows is 64 KB of LDS ([row][128 B], 16-byte chunks XORed with (row >> 1) & 7 on the SOURCE address:
//     every fragment is one conflict-free ds_read_b128), two stages; k-tile t + 1 travels global -> LDS by `buffer_load_dwordx4 ... lds`
//     (8 one-KB pieces per wave) while tile t is multiplied, retired by a vmcnt wait one tile later;
//   * ping-pong: waves 0-3 and waves 4-7 (one of each per SIMD) run the same program one barrier apart, so while one group issues its
//     12 fragment reads (+ the DMA) the other group's 24 MFMAs own the SIMD's matrix pipe - a k16 step of 24 MFMAs is 768 cycles of pipe against
//     12 KB of LDS reads per wave.
// TN kernel (both operands k-strided: weight gradients, K = rows of the batch): [k][256 lines] LDS images with 1088-byte pitch (one DMA piece
//   per k-row), fragments by two ds_read_b64_tr_b16, split-K slabs folded by gemm_mfma_fold_kernel.
#include "gemm_mfma.h"
#include "img.h"
#include <type_traits>

namespace a2f {
namespace mm {
namespace img {

constexpr int TB = 256;                 // tile edge (both m and n)
constexpr int TH = 512;
constexpr int ROWB = 128;               // bytes of one row of an R image: 32 hi + 32 lo
constexpr int PART = TB * ROWB;         // 32 KB: one operand's share of a stage
constexpr int STAGE = 2 * PART;
constexpr int TPITCH = 1024 + 64;       // T image: one k-row = 256 lines x 4 bytes, padded so that the four k-rows of a transposing read hit four bank quarters
constexpr int TPART = 32 * TPITCH;
constexpr int TSTAGE = 2 * TPART;

__device__ __forceinline__ rsrc_t rsrc_at(const float* p, unsigned bytes, long long adv_bytes) {
  const unsigned left = adv_bytes < (long long)bytes ? bytes - (unsigned)adv_bytes : 0u;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p) + adv_bytes), 0, (int)left, 0x00020000);
}
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lds_dma16(rsrc_t rs, char* lds, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
#endif
}
__device__ __forceinline__ void lds_dma16s(rsrc_t rs, char* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ u32x4_t lds_read16(const char* p) { return *reinterpret_cast<const u32x4_t*>(p); }

// GELU and its derivative from ONE exponential: Phi(-|x|) = 0.5 erfc(|x| / sqrt 2) by Abramowitz-Stegun 7.1.26 (|error| <= 0.75e-7 on Phi, relative
// accuracy kept in the negative tail because the small branch is never formed as 1 - (something near 1)), phi(x) = exp(-x^2 / 2) / sqrt(2 pi) shares
// the exponential.  ~18 vector instructions for both values against ~45 for erff + expf.
__device__ __forceinline__ void gelu_pair1(float x, float& g, float& dg) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);                       // exp(-x^2 / 2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = 0.5f * p * t * e;                                                            // Phi(-|x|)
  const float cdf = x < 0.f ? q : 1.f - q;
  g = x * cdf;
  dg = fmaf(x * 0.3989422804014327f, e, cdf);
}
__device__ __forceinline__ void gelu_pair(const f32x4 x, f32x4& g, f32x4& dg) {
  float g0, g1, g2, g3, d0, d1, d2, d3;
  gelu_pair1(x.x, g0, d0);
  gelu_pair1(x.y, g1, d1);
  gelu_pair1(x.z, g2, d2);
  gelu_pair1(x.w, g3, d3);
  g = f32x4{g0, g1, g2, g3};
  dg = f32x4{d0, d1, d2, d3};
}

__device__ __forceinline__ void split_store4(char* row_base, int n, const f32x4 v) { a2f::img::store4(row_base, n, v); }

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of one wave's 128 x 64 block (transposed accumulators: lane = row of C, four consecutive registers = four consecutive columns).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void epilogue_wave(const Args& a, const f32x16 (&acc)[4][2], int m0, int n0, int wm, int wn, int lane) {
  constexpr unsigned NOWHERE = 0x80000000u;
  const int h = lane >> 5;
  auto rsrc_of = [](const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(reinterpret_cast<const float*>(p))), 0, (int)0x80000000u, 0x00020000);
  };
  const bool generic = a.C2 || a.act || a.aux_mode || a.keep;        // wave-uniform
  const rsrc_t rc = rsrc_of(a.split_k > 1 ? a.ws + (long long)blockIdx.z * a.M * a.N : a.C);
  const rsrc_t rc2 = rsrc_of(a.C2 ? a.C2 : a.C);
  const rsrc_t rax = rsrc_of(a.aux ? a.aux : a.C);
  const rsrc_t rk = rsrc_of(a.keep ? (const void*)a.keep : (const void*)a.C);
  f32x4 bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
      bv[j][q] = (a.bias && a.split_k <= 1 && n < a.N) ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const unsigned ldc = a.split_k > 1 ? (unsigned)a.N : (unsigned)a.ldc;
  const bool do_cs = a.c_colsum != nullptr && a.split_k <= 1;        // wave-uniform
  f32x4 cs[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) cs[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 128 + i * 32 + (lane & 31);
    const bool mok = m < a.M;
    unsigned crow;
    if (a.c_rps <= 0 || a.split_k > 1) crow = (unsigned)m * ldc;
    else { const int sq = m / a.c_rps; crow = (unsigned)sq * (unsigned)a.c_seq_stride + (unsigned)(m - sq * a.c_rps) * ldc; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = n0 + wn * 64 + j * 32;
      unsigned kw = 0xFFFFFFFFu;
      if (generic && a.keep) kw = __builtin_amdgcn_raw_buffer_load_b32(rk, (mok && nb < a.N) ? ((unsigned)m * (unsigned)a.keep_wpr + (unsigned)(nb >> 5)) * 4u : NOWHERE, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nb + 8 * q + 4 * h;
        const bool ok = mok && n < a.N;
        const unsigned off = ok ? (crow + (unsigned)n) * 4u : NOWHERE;
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        if (a.split_k > 1) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rc, off, 0, 0);
          continue;
        }
        v = v * a.alpha + bv[j][q];
        if (generic) {
          if (a.act == 4) {
            f32x4 g, dg;
            gelu_pair(v, g, dg);
            if (a.C2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, dg), rc2, off, 0, 0);
            v = g;
          } else {
            if (a.C2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rc2, off, 0, 0);
            if (a.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            else if (a.act == 3) {
              f32x4 g, dg;
              gelu_pair(v, g, dg);
              v = g;
            }
          }
          if (a.keep) {
            const unsigned b = kw >> (8 * q + 4 * h);
            v.x = (b & 1u) ? v.x * a.keep_scale : 0.f;
            v.y = (b & 2u) ? v.y * a.keep_scale : 0.f;
            v.z = (b & 4u) ? v.z * a.keep_scale : 0.f;
            v.w = (b & 8u) ? v.w * a.keep_scale : 0.f;
          }
          if (a.aux_mode == 4) {
            v *= __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rax, off, 0, 0));
          } else if (a.aux_mode == 1) {
            v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rax, ok ? ((unsigned)m * (unsigned)a.ldaux + (unsigned)n) * 4u : NOWHERE, 0, 0));
          }
        }
        if (do_cs && ok) cs[j][q] += v;
        if (a.c_format == 1) {
          if (ok) split_store4(reinterpret_cast<char*>(a.C) + (size_t)crow * 4u, n, v);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rc, off, 0, 0);
        }
      }
    }
  }
  // column sums of the stored values over this wave's 128 rows (the bias gradient of the Linear whose input gradient C is): a fixed butterfly over the 32
  // row-lanes, one partial row per (m-tile, wave row) in the workspace, folded in order by colsum_fold_kernel - deterministic, no atomics
  if (do_cs) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = cs[j][q];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          v.x += __shfl_xor(v.x, o, 64);
          v.y += __shfl_xor(v.y, o, 64);
          v.z += __shfl_xor(v.z, o, 64);
          v.w += __shfl_xor(v.w, o, 64);
        }
        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
        if ((lane & 31) == 0 && n < a.N) *reinterpret_cast<f32x4*>(a.ws + (long long)((m0 >> 8) * 2 + wm) * a.N + n) = v;
      }
  }
}

__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, int P, int N, float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;               // four chains, fixed order
  int p = 0;
  for (; p + 4 <= P; p += 4) {
    v0 += part[(long long)p * N + n];
    v1 += part[(long long)(p + 1) * N + n];
    v2 += part[(long long)(p + 2) * N + n];
    v3 += part[(long long)(p + 3) * N + n];
  }
  for (; p < P; ++p) v0 += part[(long long)p * N + n];
  out[n] = (v0 + v1) + (v2 + v3);
}

// ---------------------------------------------------------------------------------------------------------------------
// The kernel.  TN = false: C[m, n] = epi(sum_k A[m, k] B[n, k]), A / B row images (k contiguous).
//              TN = true:  C[m, n] = sum_k A[k, m] B[k, n] over this workgroup's k-range (split-K slabs), A / B images with the m / n index contiguous.
// ---------------------------------------------------------------------------------------------------------------------
template <bool TN>
__global__ __launch_bounds__(TH, 2) void gemm_img_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int STG = TN ? TSTAGE : STAGE;
  constexpr int PRT = TN ? TPART : PART;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int m0, n0;
  tile_origin(a, TB, TB, m0, n0);
  int kbeg = 0, kend = a.K;
  if (TN && a.split_k > 1) {
    const int sk = __builtin_amdgcn_readfirstlane((int)blockIdx.z);
    const int ktiles = (a.K + 31) >> 5;
    const int per = (ktiles + a.split_k - 1) / a.split_k;
    kbeg = sk * per * 32;
    kend = min(a.K, kbeg + per * 32);
  }
  const int nkt = kend > kbeg ? (kend - kbeg + 31) >> 5 : 0;
  const float* pA = uniform_ptr(a.A.p);
  const float* pB = uniform_ptr(a.B.p);

  // ---- the 8 DMA pieces (1 KB each) of this wave per k-tile: piece pc = wave + 8 i, operand pc >> 5.
  // NT: piece = rows 8 (pc & 31) .. + 7 of the operand's 256; lane -> row lane >> 3, LDS chunk lane & 7, which holds SOURCE chunk (lane & 7) ^ ((row >> 1) & 7).
  // TN: piece = k-row pc & 31, all 256 lines (8 blocks of 128 bytes); lane -> bytes 16 lane .. of that run.
  unsigned voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pc = wave + 8 * i;
    const bool isB = i >= 4;
    const Op& op = isB ? a.B : a.A;
    const int x0 = isB ? n0 : m0, cnt = isB ? a.N : a.M;
    if constexpr (!TN) {
      const int row = (pc & 31) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      const int g = x0 + row;
      unsigned rb;
      if (op.lps > 0) {
        const int s = g / op.lps, q = g - s * op.lps;
        rb = (unsigned)s * (unsigned)(op.seq_stride * 4) + (unsigned)(q * op.ld + op.off) * 4u;
      } else {
        rb = (unsigned)g * (unsigned)op.ld * 4u;
      }
      voff[i] = g < cnt ? rb + (unsigned)(c * 16) : 0xFFFFFFFFu;
    } else {
      const int krow = pc & 31;
      const int x = x0 + (lane >> 3) * 32;                     // first line of this lane's 128-byte block
      // windowed operand (conv weight gradient: a k-row is one output position, its lines the window starting at q * ld + off): the row's byte offset is
      // wave-uniform and travels in the instruction's scalar offset, re-derived per k-tile from (sequence, position)
      const unsigned rowpart = op.lps > 0 ? (unsigned)op.off : (unsigned)krow * (unsigned)op.ld;
      voff[i] = x < cnt ? (rowpart + (unsigned)x0) * 4u + (unsigned)(lane * 16) : 0xFFFFFFFFu;
    }
  }
  // TN, windowed B: (sequence, position) of the four B k-rows this wave moves, for the current k-tile
  int bs[4] = {0, 0, 0, 0}, bq[4] = {0, 0, 0, 0};
  const bool bwin = TN && a.B.lps > 0;                         // wave-uniform
  if (bwin) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = kbeg + ((wave + 8 * (i + 4)) & 31);
      bs[i] = __builtin_amdgcn_readfirstlane(r / a.B.lps);
      bq[i] = __builtin_amdgcn_readfirstlane(r - (r / a.B.lps) * a.B.lps);
    }
  }
  auto dma = [&](int kt, int st) {
    rsrc_t ra, rb;
    if constexpr (!TN) {
      ra = rsrc_at(pA, a.A.bytes, (long long)kt * 128);
      rb = rsrc_at(pB, a.B.bytes, (long long)kt * 128);
    } else {                                                   // rows beyond kend are out of range of the re-based resource: zeros
      const long long k0 = kbeg + kt * 32;
      const long long la = (long long)a.A.ld * 4, lb = (long long)a.B.ld * 4;
      const long long enda = min((long long)a.A.bytes, kend * la), endb = min((long long)a.B.bytes, kend * lb);
      ra = rsrc_at(pA, (unsigned)enda, k0 * la);
      rb = rsrc_at(pB, (unsigned)endb, k0 * lb);
    }
    char* base = smem + st * STG;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pc = wave + 8 * i;
      char* dst = TN ? base + (i >= 4 ? PRT : 0) + (pc & 31) * TPITCH : base + pc * 1024;
      if (TN && i >= 4 && bwin) {
        // rows at or beyond the end of this k-range must read as zero whatever lies there (their A rows are zero, but 0 x NaN is not): out-of-range offset
        const bool live = kbeg + kt * 32 + (pc & 31) < kend;
        const unsigned so = live ? (unsigned)bs[i - 4] * (unsigned)(a.B.seq_stride * 4) + (unsigned)bq[i - 4] * (unsigned)(a.B.ld * 4) : 0u;
        lds_dma16s(rsrc_at(pB, a.B.bytes, 0), dst, live ? voff[i] : 0xFFFFFFFFu, so);
      } else {
        lds_dma16(i >= 4 ? rb : ra, dst, voff[i]);
      }
    }
    if (TN && bwin) {                                          // next k-tile: 32 positions on (scalar arithmetic; short sequences wrap more than once)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bq[i] += 32;
        while (bq[i] >= a.B.lps) { bq[i] -= a.B.lps; bs[i] += 1; }
      }
    }
  };

  // ---- fragment addresses
  // NT: row = (lane & 31) of a 32-row block, 16-byte chunk (ks * 2 + (lane >> 5) + 4 lo) ^ ((row >> 1) & 7)
  // TN: two transposing 8-byte reads: k-rows ks * 16 + 8 (g >> 1) + (i >> 2) and + 4, lines 16 (g & 1) + 4 (i & 3) .. + 3 of the block (g = lane >> 4, i = lane & 15)
  int aA, aB;
  if constexpr (!TN) {
    const int fx = ((lane & 31) >> 1) & 7;
    const int ch0 = ((lane >> 5) ^ fx) << 4;
    aA = (wm * 128 + (lane & 31)) * ROWB + ch0;
    aB = PRT + (wn * 64 + (lane & 31)) * ROWB + ch0;
  } else {
    const int g = lane >> 4, i = lane & 15;
    const int lo0 = (8 * (g >> 1) + (i >> 2)) * TPITCH + (16 * (g & 1) + 4 * (i & 3)) * 2;
    aA = lo0 + wm * 4 * 128;
    aB = PRT + lo0 + wn * 2 * 128;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4_t fa[2][4], fb[2][2];
  auto reads = [&](int st, int ks) {
    const char* sb = smem + st * STG;
#pragma unroll
    for (int lo = 0; lo < 2; ++lo) {
      if constexpr (!TN) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[lo][j] = lds_read16(sb + (aB ^ (ks << 5) ^ (lo << 6)) + j * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[lo][i] = lds_read16(sb + (aA ^ (ks << 5) ^ (lo << 6)) + i * 4096);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const char* p = sb + aB + ks * 16 * TPITCH + lo * 64 + j * 128;
          const u32x2_t r0 = tr_read(p), r1 = tr_read(p + 4 * TPITCH);
          fb[lo][j] = u32x4_t{r0.x, r0.y, r1.x, r1.y};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const char* p = sb + aA + ks * 16 * TPITCH + lo * 64 + i * 128;
          const u32x2_t r0 = tr_read(p), r1 = tr_read(p + 4 * TPITCH);
          fa[lo][i] = u32x4_t{r0.x, r0.y, r1.x, r1.y};
        }
      }
    }
  };
  auto mfmas = [&]() {      // small terms first; D = (tile of B) x (tile of A)^T: lane = m, registers = n
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma(fb[1][j], fa[0][i], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma(fb[0][j], fa[1][i], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma(fb[0][j], fa[0][i], acc[i][j]);
  };

  // Slots between barriers (group 0 = waves 0-3, group 1 = waves 4-7, one wave of each per SIMD; group 1 runs one barrier behind):
  //   slot 4t:     group 0 reads k16 step 0 of tile t and issues the DMA of tile t + 1   | group 1 multiplies step 1 of tile t - 1
  //   slot 4t + 1: group 0 multiplies step 0                                             | group 1 reads step 0 of tile t, issues its DMA of tile t + 1
  //   slot 4t + 2: group 0 reads step 1                                                  | group 1 multiplies step 0
  //   slot 4t + 3: group 0 multiplies step 1, then waits for its DMA                     | group 1 reads step 1, then waits for its DMA
  // Every wave's fragment reads are complete (lgkmcnt 0) before the barrier that ends its read slot, so when slot 4t begins nobody reads tile t - 1's
  // stage any more (write-after-read), and both groups' DMA of tile t + 1 has landed before the barrier that ends slot 4t + 3 (read-after-write).
  if (nkt > 0) dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    reads(st, 0);
    if (kt + 1 < nkt) dma(kt + 1, st ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mfmas();
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    reads(st, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mfmas();
    __builtin_amdgcn_s_setprio(0);
    if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  epilogue_wave(a, acc, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// The k-contiguous kernel on v_mfma_f32_16x16x32_bf16: the same tile, staging, ping-pong and LDS images; a wave's 128 x 64 block is 8 x 4 accumulator
// tiles of 16 x 16 (the same 128 registers), one MFMA spans the whole 32-deep k-tile.  Same cycles per flop as the 32 x 32 x 16 form, but the chip holds a
// higher clock on this shape under a dense bf16 loop (MI355X guide, DVFS give-back item 7), and with the transposed accumulators (lane = row l & 15, registers =
// columns 4 (l >> 4) .. + 3) a store instruction covers 16 rows x 64 contiguous bytes instead of 32 rows x 32.
// Clusters per k-tile: (a) all 8 B fragments + A fragments of m-tiles 0-3, 48 MFMAs; (b) A fragments of m-tiles 4-7 (B fragments stay), 48 MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
using f32x4v = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4v mfma16(u32x4_t a, u32x4_t b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ void epilogue_wave16(const Args& a, const f32x4v (&acc)[8][4], int m0, int n0, int wm, int wn, int lane) {
  constexpr unsigned NOWHERE = 0x80000000u;
  constexpr int AHEAD = 4;                                            // row blocks whose aux loads are in flight ahead of the one being finished
  const int lr = lane & 15, lc = lane >> 4;
  auto rsrc_of = [](const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(reinterpret_cast<const float*>(p))), 0, (int)0x80000000u, 0x00020000);
  };
  const bool generic = a.C2 || a.act || a.aux_mode || a.keep;        // wave-uniform
  const rsrc_t rc = rsrc_of(a.C);
  const rsrc_t rc2 = rsrc_of(a.C2 ? a.C2 : a.C);
  const rsrc_t rax = rsrc_of(a.aux ? a.aux : a.C);
  const rsrc_t rk = rsrc_of(a.keep ? (const void*)a.keep : (const void*)a.C);
  const unsigned ldc = (unsigned)a.ldc;
  // columns: the DMA permutes the rows of the B tile inside every 32-row block (LDS row r holds row 8 ((r & 15) >> 2) + 4 (r >> 4) + (r & 3)), so accumulator
  // tile j = 2 w + half of this lane covers columns 32 w + 8 lc + 4 half .. + 3: tiles 2 w and 2 w + 1 together are EIGHT consecutive columns - 32 contiguous bytes
  // of an fp32 row (a full 128-byte line per row from the four lane groups), 16 + 16 contiguous bytes of an image row
  const int nw = n0 + wn * 64 + 8 * lc;
  auto col_of = [&](int j) { return nw + 32 * (j >> 1) + 4 * (j & 1); };
  auto row_of = [&](int i) { return m0 + wm * 128 + i * 16 + lr; };
  auto crow_of = [&](int m) -> unsigned {
    if (a.c_rps <= 0) return (unsigned)m * ldc;
    const int sq = m / a.c_rps;
    return (unsigned)sq * (unsigned)a.c_seq_stride + (unsigned)(m - sq * a.c_rps) * ldc;
  };
  // Every load the epilogue needs is issued well ahead of its use (a load waited for right where it is used costs one L2 / HBM round trip per float4 with
  // nothing else in flight): all keep words first, the aux float4s AHEAD row blocks ahead through a register ring.
  unsigned kw[8][2];
  if (a.keep) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = row_of(i);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int nb = n0 + wn * 64 + w * 32;
        kw[i][w] = __builtin_amdgcn_raw_buffer_load_b32(rk, (m < a.M && nb < a.N) ? ((unsigned)m * (unsigned)a.keep_wpr + (unsigned)(nb >> 5)) * 4u : NOWHERE, 0, 0);
      }
    }
  }
  u32x4_t ax[AHEAD][4];
  auto load_aux = [&](int i, u32x4_t (&dst)[4]) {
    const int m = row_of(i);
    const unsigned base = a.aux_mode == 4 ? crow_of(m) : (unsigned)m * (unsigned)a.ldaux;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col_of(j);
      dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rax, (m < a.M && n < a.N) ? (base + (unsigned)n) * 4u : NOWHERE, 0, 0);
    }
  };
  if (a.aux_mode) {
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) load_aux(i, ax[i]);
  }
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = col_of(j);
    bv[j] = (a.bias && n < a.N) ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_cs = a.c_colsum != nullptr;
  f32x4 cs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = row_of(i);
    const bool mok = m < a.M;
    const unsigned crow = crow_of(m);
    f32x4 held = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col_of(j);
      const bool ok = mok && n < a.N;
      const unsigned off = ok ? (crow + (unsigned)n) * 4u : NOWHERE;
      f32x4 v = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      v = v * a.alpha + bv[j];
      if (generic) {
        if (a.act == 4) {
          f32x4 g, dg;
          gelu_pair(v, g, dg);
          if (a.C2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, dg), rc2, off, 0, 0);
          v = g;
        } else {
          if (a.C2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rc2, off, 0, 0);
          if (a.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          else if (a.act == 3) {
            f32x4 g, dg;
            gelu_pair(v, g, dg);
            v = g;
          }
        }
        if (a.keep) {
          const unsigned b = kw[i][j >> 1] >> (8 * lc + 4 * (j & 1));
          v.x = (b & 1u) ? v.x * a.keep_scale : 0.f;
          v.y = (b & 2u) ? v.y * a.keep_scale : 0.f;
          v.z = (b & 4u) ? v.z * a.keep_scale : 0.f;
          v.w = (b & 8u) ? v.w * a.keep_scale : 0.f;
        }
        if (a.aux_mode == 4) v *= __builtin_bit_cast(f32x4, ax[i % AHEAD][j]);
        else if (a.aux_mode == 1) v += __builtin_bit_cast(f32x4, ax[i % AHEAD][j]);
      }
      if (do_cs && ok) cs[j] += v;
      if (a.c_format == 1) {
        if constexpr (true) {
          if ((j & 1) == 0) held = v;                              // first half of the eight columns: wait for the second
          else if (ok) a2f::img::store8(reinterpret_cast<char*>(a.C) + (size_t)crow * 4u, n - 4, held, v);
        }
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rc, off, 0, 0);
      }
    }
    if (a.aux_mode && i + AHEAD < 8) load_aux(i + AHEAD, ax[i % AHEAD]);
  }
  if (do_cs) {       // column sums over the wave's 128 rows: butterfly over the 16 row-lanes, one partial row per (m-tile, wave row), folded in order
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = cs[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        v.x += __shfl_xor(v.x, o, 64);
        v.y += __shfl_xor(v.y, o, 64);
        v.z += __shfl_xor(v.z, o, 64);
        v.w += __shfl_xor(v.w, o, 64);
      }
      const int n = col_of(j);
      if (lr == 0 && n < a.N) *reinterpret_cast<f32x4*>(a.ws + (long long)((m0 >> 8) * 2 + wm) * a.N + n) = v;
    }
  }
}

// CHK: the windows of A may reach outside their sequence (zero padding: the two-row windows of the k = 3 / stride 2 data gradient in gather form) - every A
// piece then tests its element offset against [0, seq_elems) per k-tile and addresses the un-rebased buffer (negative offsets cannot ride on a moved base).
template <bool CHK>
__global__ __launch_bounds__(TH, 2) void gemm_img16_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int m0, n0;
  tile_origin(a, TB, TB, m0, n0);
  const int nkt = a.K >> 5;
  const float* pA = uniform_ptr(a.A.p);
  const float* pB = uniform_ptr(a.B.p);
  unsigned voff[8];
  int e0[CHK ? 4 : 1];                                              // CHK: element offset of this lane's A row inside its sequence at k = 0 (may be negative)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pc = wave + 8 * i;
    const bool isB = i >= 4;
    const Op& op = isB ? a.B : a.A;
    const int x0 = isB ? n0 : m0, cnt = isB ? a.N : a.M;
    const int row = (pc & 31) * 8 + (lane >> 3);                                    // LDS row
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    // B: the rows of every 32-block are permuted on their way in, so that a lane's accumulators are eight consecutive columns (epilogue_wave16)
    const int srow = isB ? (row & ~31) + 8 * ((row & 15) >> 2) + 4 * ((row >> 4) & 1) + (row & 3) : row;
    const int g = x0 + srow;
    unsigned rb;
    if (op.lps > 0) {
      const int s = g / op.lps, q = g - s * op.lps;
      if (CHK && !isB) {
        e0[i & 3] = g < cnt ? q * op.ld + op.off : -0x40000000;      // a row beyond M: never in range
        rb = (unsigned)s * (unsigned)(op.seq_stride * 4);
      } else {
        rb = (unsigned)s * (unsigned)(op.seq_stride * 4) + (unsigned)(q * op.ld + op.off) * 4u;
      }
    } else {
      rb = (unsigned)g * (unsigned)op.ld * 4u;
    }
    voff[i] = g < cnt ? rb + (unsigned)(c * 16) : 0xFFFFFFFFu;
  }
  auto dma = [&](int kt, int st) {
    const rsrc_t ra = rsrc_at(pA, a.A.bytes, CHK ? 0ll : (long long)kt * 128);
    const rsrc_t rb = rsrc_at(pB, a.B.bytes, (long long)kt * 128);
    char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned vo = voff[i];
      if constexpr (CHK) {
        if (i < 4) {
          const unsigned e = (unsigned)(e0[i] + kt * 32);
          vo = e < (unsigned)a.A.seq_elems ? vo + e * 4u : 0xFFFFFFFFu;
        }
      }
      lds_dma16(i >= 4 ? rb : ra, base + i * 8192, vo);
    }
  };
  // fragment: row lane & 15 of a 16-row block, 16-byte chunk ((lane >> 4) + 4 lo) ^ ((row >> 1) & 7)
  const int lr = lane & 15;
  const int ch0 = ((lane >> 4) ^ ((lr >> 1) & 7)) << 4;
  const int aA = (wm * 128 + lr) * ROWB + ch0;
  const int aB = PART + (wn * 64 + lr) * ROWB + ch0;

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  u32x4_t fa[2][4], fb[2][4];
  auto reads_b = [&](int st) {
    const char* sb = smem + st * STAGE;
#pragma unroll
    for (int lo = 0; lo < 2; ++lo)
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[lo][j] = lds_read16(sb + (aB ^ (lo << 6)) + j * 2048);
  };
  auto reads_a = [&](int st, int half) {
    const char* sb = smem + st * STAGE;
#pragma unroll
    for (int lo = 0; lo < 2; ++lo)
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[lo][i] = lds_read16(sb + (aA ^ (lo << 6)) + (half * 4 + i) * 2048);
  };
  auto mfmas = [&](auto half_c) {
    constexpr int half = decltype(half_c)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = mfma16(fb[1][j], fa[0][i], acc[half * 4 + i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = mfma16(fb[0][j], fa[1][i], acc[half * 4 + i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[half * 4 + i][j] = mfma16(fb[0][j], fa[0][i], acc[half * 4 + i][j]);
  };
  if (nkt > 0) dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    reads_b(st);
    reads_a(st, 0);
    if (kt + 1 < nkt) dma(kt + 1, st ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mfmas(std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    reads_a(st, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mfmas(std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_setprio(0);
    if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  epilogue_wave16(a, acc, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 -> split-32 image (optionally transposed: weights W (R, C) -> image of W^T (C, R)); a streaming pass for operands no producer emits as images
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_img_kernel(const float* __restrict__ x, long long R, int C, long long ld, char* __restrict__ out, long long ldo) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  const int c4 = C >> 2;
  if (idx >= R * c4) return;
  const long long r = idx / c4;
  const int n = (int)(idx - r * c4) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + n);
  split_store4(out + r * ldo * 4, n, v);
}
// transposed: out row c (length R) = column c of x; a 32 x 32 tile per workgroup through LDS
__global__ __launch_bounds__(256) void split_img_t_kernel(const float* __restrict__ x, int R, int C, long long ld, char* __restrict__ out, long long ldo) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? x[(long long)r * ld + c] : 0.f;
  }
  __syncthreads();
  // thread -> output row c0 + (t >> 3), four consecutive r
  const int oc = threadIdx.x >> 3, q = (threadIdx.x & 7) * 4;
  if (c0 + oc < C && r0 + q < R) {
    const f32x4 v = {tile[q][oc], tile[q + 1][oc], tile[q + 2][oc], tile[q + 3][oc]};
    split_store4(out + (long long)(c0 + oc) * ldo * 4, r0 + q, v);
  }
}

}  // namespace img

template <bool CHK>
static int launch_img16(const Args& a, hipStream_t st) {
  using namespace img;
  constexpr int smem = 2 * STAGE;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_img16_kernel<CHK>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return set_err(A2F_ELAUNCH, "a2f_mm (image operands): cannot reserve %d bytes of LDS", smem);
    attr_done = true;
  }
  dim3 grid((unsigned)(cdiv(a.N, TB) * cdiv(a.M, TB)), 1, 1);
  hipLaunchKernelGGL(gemm_img16_kernel<CHK>, grid, dim3(TH), smem, st, a);
  return A2F_OK;
}

template <bool TN>
static int launch_img_t(const Args& a, hipStream_t st) {
  using namespace img;
  constexpr int smem = 2 * (TN ? TSTAGE : STAGE);
  auto kern = gemm_img_kernel<TN>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return set_err(A2F_ELAUNCH, "a2f_mm (image operands): cannot reserve %d bytes of LDS", smem);
    attr_done = true;
  }
  dim3 grid((unsigned)(cdiv(a.N, TB) * cdiv(a.M, TB)), 1, (unsigned)(TN ? a.split_k : 1));
  hipLaunchKernelGGL(kern, grid, dim3(TH), smem, st, a);
  return A2F_OK;
}

// both operands format 1; trans flags equal (NT: both k-contiguous, TN: both line-contiguous); checked by the caller (a2f_mm)
int launch_img(const Args& a, bool tn, hipStream_t st) {
  const bool chk = !tn && a.A.lps > 0 && (a.A.off < 0 || (long long)(a.A.lps - 1) * a.A.ld + a.A.off + a.K > a.A.seq_elems);      // padded windows of A
  const int rc = tn ? launch_img_t<true>(a, st) : (chk ? launch_img16<true>(a, st) : launch_img16<false>(a, st));
  if (rc) return rc;
  if (!tn && a.c_colsum) {
    const int P = 2 * (int)cdiv(a.M, img::TB);
    hipLaunchKernelGGL(img::colsum_fold_kernel, dim3((unsigned)cdiv(a.N, 256)), dim3(256), 0, st, a.ws, P, a.N, a.c_colsum);
  }
  return A2F_OK;
}

}  // namespace mm
}  // namespace a2f

extern "C" int a2f_split_img(const float* x, int64_t R, int64_t C, int64_t ld, int32_t transpose, void* out, int64_t ldo, a2f_stream_t stream) {
  using namespace a2f;
  using namespace a2f::mm::img;
  A2F_CHECK_ARG(x && out && R > 0 && C > 0, "a2f_split_img: null / empty");
  A2F_CHECK_ARG(C % 4 == 0 && ld % 4 == 0 && ldo % 32 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0, "a2f_split_img: C, ld multiples of 4, ldo of 32, 16-byte aligned buffers");
  if (!transpose) {
    A2F_CHECK_ARG(ldo >= C, "a2f_split_img: ldo < C");
    const long long n = R * (C / 4);
    hipLaunchKernelGGL(split_img_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)R, (int)C, (long long)ld, (char*)out, (long long)ldo);
  } else {
    A2F_CHECK_ARG(ldo >= R && R % 4 == 0 && R < (1ll << 31) && C < (1ll << 31), "a2f_split_img (transpose): ldo < R or R %% 4 != 0");
    hipLaunchKernelGGL(split_img_t_kernel, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream, x, (int)R, (int)C, (long long)ld, (char*)out,
                       (long long)ldo);
  }
  A2F_CHECK_LAUNCH("a2f_split_img");
  return A2F_OK;
}
