// Generic fused fp32 GEMM for gfx950 on the f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// C[m,n] = epi(sum_k A(m,k) B(k,n)) with element-strided / windowed / masked operands: see
// include/a2f_hip.h (a2f_gemm_f32).  One kernel serves every Linear and every strided (3,1)/(1,3)/(4,1)
// convolution (as implicit GEMM over overlapping activation windows) of the reference models,
// forward, data-gradient and weight-gradient (split-K, fp32 atomics).
//
// Tiling (wave64): workgroup = 256 threads = 4 waves stacked along M; tile BM=128 x BN in {32,64,96,128},
// BK=16; each wave owns 32 rows x BN columns = BN/32 MFMA 32x32 accumulators (16 VGPR each).
// LDS images are k-major ([k][m], [k][n], row pitch +4 floats) so the A/B fragment reads
// (lane -> m or n, lane>>5 -> k) are conflict-free ds_read_b32; global->LDS goes through registers with the
// next tile's loads issued before the current tile's MFMAs (two LDS buffers, one barrier per k-tile).
// Exact fp32: the f32 MFMA is a k-ordered fmaf chain, so results match torch CPU fp32 to rounding order.
#include "a2f_common.h"

namespace a2f {

constexpr int G_BM = 128;
constexpr int G_BK = 16;
constexpr int G_THREADS = 256;

// Epilogue shared by the fp32 and the split-precision kernel.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int NT>
__device__ __forceinline__ void gemm_epilogue(const a2f_gemm_desc& d, const f32x16 (&acc)[NT], int m0, int n0, int wave, int lane, int M, int N,
                                              int Neff) {
  const bool atomic = d.split_k > 1;
  if (atomic && d.splitk_ws) {            // slab mode: this split's partial tile -> ws[z][row][col], pitch Neff
    float* slab = d.splitk_ws + (int64_t)blockIdx.z * M * Neff;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + j * 32 + (lane & 31);
      if (col >= Neff) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) slab[(int64_t)row * Neff + col] = acc[j][r];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 32 + (lane & 31);
    if (col >= Neff) continue;
    const float bias = (d.bias && col < N) ? d.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= M) continue;
      float v = acc[j][r];
      if (col == N) {                       // virtual ones column -> bias gradient
        if (atomic) atomicAdd(d.ones_out + row, v);
        else d.ones_out[row] = d.accumulate ? d.ones_out[row] + v : v;
        continue;
      }
      float* cp = d.C + (int64_t)row * d.ldc + col;
      if (atomic) { atomicAdd(cp, v); continue; }
      v += bias;
      if (d.accumulate) v += *cp;
      if (d.act == 1) v = fmaxf(v, 0.f);
      else if (d.act == 2) v = tanhf(v);
      if (d.aux_mode) {
        const float ax = d.aux[(int64_t)row * d.ldaux + col];
        if (d.aux_mode == 1) v = ax > 0.f ? v : 0.f;
        else if (d.aux_mode == 2) v *= (1.f - ax * ax);
        else v += ax;
      }
      *cp = v;
    }
  }
}

template <int BN>
__global__ __launch_bounds__(G_THREADS) void gemm_f32_kernel(const a2f_gemm_desc d) {
  constexpr int NT = BN / 32;
  constexpr int LDA = G_BM + 4;
  constexpr int LDB = BN + 4;
  constexpr int A_PER = G_BK * G_BM / G_THREADS;   // 8
  constexpr int B_PER = (G_BK * BN + G_THREADS - 1) / G_THREADS;
  __shared__ float smem[2 * G_BK * LDA + 2 * G_BK * LDB];
  float* As = smem;                       // [2][BK][LDA]
  float* Bs = smem + 2 * G_BK * LDA;      // [2][BK][LDB]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int M = (int)d.M, N = (int)d.N, K = (int)d.K;
  const int Neff = N + (d.ones_out ? 1 : 0);
  const int m0 = blockIdx.x * G_BM;
  const int n0 = blockIdx.y * BN;

  // split-K range
  int kbeg = 0, kend = K;
  if (d.split_k > 1) {
    const int ktiles = (K + G_BK - 1) / G_BK;
    const int per = (ktiles + d.split_k - 1) / d.split_k;
    kbeg = (int)blockIdx.z * per * G_BK;
    kend = min(K, kbeg + per * G_BK);
    if (kbeg >= kend) return;
  }
  const int nkt = (kend - kbeg + G_BK - 1) / G_BK;

  const bool a_kfast = (d.a_rs != 1);     // K-contiguous (or generic) rows vs M-contiguous storage
  const bool b_kfast = (d.b_cs != 1);     // N-contiguous storage unless the column stride is not 1
  const float* __restrict__ Ap = d.A + d.a_off;
  const float* __restrict__ Bp = d.B + d.b_off;

  // per-thread element coordinates inside a tile (fixed across k-tiles)
  int a_mm[A_PER], a_kk[A_PER];
  unsigned a_rowmask = 0;                 // bit i: row of element i is a masked (padding) row
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int e = t + i * G_THREADS;
    if (a_kfast) { a_kk[i] = e % G_BK; a_mm[i] = e / G_BK; }
    else         { a_mm[i] = e % G_BM; a_kk[i] = e / G_BM; }
    if (d.mask_operand == 1) {
      const int m = m0 + a_mm[i];
      if (m % d.mask_period == d.mask_phase) a_rowmask |= (1u << i);
    }
  }
  int b_nn[B_PER], b_kk[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int e = t + i * G_THREADS;
    if (b_kfast) { b_kk[i] = e % G_BK; b_nn[i] = e / G_BK; }
    else         { b_nn[i] = e % BN;   b_kk[i] = e / BN; }
  }

  float ra[A_PER], rb[B_PER];

  auto load_tile = [&](int kt) {
    const int k0 = kbeg + kt * G_BK;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int m = m0 + a_mm[i];
      const int k = k0 + a_kk[i];
      bool ok = (m < M) && (k < kend);
      if ((a_rowmask >> i) & 1u) ok = ok && !(k >= d.mask_lo && k < d.mask_hi);
      ra[i] = ok ? Ap[(int64_t)m * d.a_rs + (int64_t)k * d.a_cs] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int e = t + i * G_THREADS;
      float v = 0.f;
      if (e < G_BK * BN) {
        const int n = n0 + b_nn[i];
        const int k = k0 + b_kk[i];
        if (k < kend) {
          if (n < N) {
            bool ok = true;
            if (d.mask_operand == 2) ok = !((k % d.mask_period == d.mask_phase) && n >= d.mask_lo && n < d.mask_hi);
            if (ok) v = Bp[(int64_t)k * d.b_rs + (int64_t)n * d.b_cs];
          } else if (n == N && d.ones_out) {
            v = 1.f;
          }
        }
      }
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float* as = As + buf * G_BK * LDA;
    float* bs = Bs + buf * G_BK * LDB;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) as[a_kk[i] * LDA + a_mm[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int e = t + i * G_THREADS;
      if (e < G_BK * BN) bs[b_kk[i] * LDB + b_nn[i]] = rb[i];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frag_row = wave * 32 + (lane & 31);
  const int frag_k = lane >> 5;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const float* as = As + buf * G_BK * LDA;
    const float* bs = Bs + buf * G_BK * LDB;
#pragma unroll
    for (int kk = 0; kk < G_BK; kk += 2) {
      const float a = as[(kk + frag_k) * LDA + frag_row];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float b = bs[(kk + frag_k) * LDB + j * 32 + (lane & 31)];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<NT>(d, acc, m0, n0, wave, lane, M, N, Neff);
}

// =====================================================================================================================
// Split-precision variants (d.precision == 1 / 2): same tiling, same operand contract, same epilogue, but every product on
// the bf16 matrix pipe.  precision 1: x = hi + lo, three bf16 MFMAs per product (lo*hi' + hi*lo' + hi*hi'), <= 2^-16
// relative error per product, 5.3x less matrix time than v_mfma_f32_32x32x2_f32 (see head_fused.hip).  precision 2:
// x = hi + mid + lo (24 significant bits), the six products whose pieces multiply to >= 2^-16 of the full product
// (hh', hm', mh', hl', lh', mm'), fp32 accumulation: error <= ~2^-23 per product, i.e. fp32-class results at 2.7x less
// matrix time.  Used for the convolution GEMMs of Audio2Mesh, which are matrix-pipe bound in fp32.
// Operands are split when they are stored to LDS: a thread owns PAIRS of consecutive k, so one v_cvt_pk_bf16_f32 pair
// yields a packed dword per image; the images are [row][16 k] bf16 with a 48-byte pitch, which makes every fragment of
// v_mfma_f32_32x32x16_bf16 (8 consecutive k of one row) ONE conflict-free ds_read_b128.
using gx_bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using gx_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using gx_f32x2 = __attribute__((ext_vector_type(2))) float;
using gx_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
// (a, b) -> NS packed bf16 pair images: piece q = bf16(x - sum of the earlier pieces); 2 pieces carry 16 bits, 3 carry 24
template <int NS>
__device__ __forceinline__ void gx_split(float a, float b, unsigned (&img)[NS]) {
  gx_f32x2 r = {a, b};
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const gx_bf16x2 p = __builtin_convertvector(r, gx_bf16x2);
    img[q] = __builtin_bit_cast(unsigned, p);
    if (q + 1 < NS) r = r - __builtin_convertvector(p, gx_f32x2);
  }
}
__device__ __forceinline__ f32x16 gx_mfma(gx_u32x4 a, gx_u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gx_bf16x8, a), __builtin_bit_cast(gx_bf16x8, b), c, 0, 0, 0);
}
constexpr int GX_PITCH = 48;          // bytes per image row: 16 bf16 + 16 B pad

template <int BN, int NS>
__global__ __launch_bounds__(G_THREADS) void gemm_split_kernel(const a2f_gemm_desc d) {
  constexpr int NT = BN / 32;
  constexpr int KP = G_BK / 2;                                   // k-pairs per tile
  constexpr int A_PAIRS = G_BM * KP / G_THREADS;                 // 4
  constexpr int B_PAIRS = (BN * KP + G_THREADS - 1) / G_THREADS;
  constexpr int A_IMG = G_BM * GX_PITCH, B_IMG = BN * GX_PITCH;  // one bf16 image (hi or lo)
  constexpr int BUF_BYTES = NS * (A_IMG + B_IMG);                 // [A images hi, (mid,) lo | B images hi, (mid,) lo]
  extern __shared__ __attribute__((aligned(16))) char smem[];    // two buffers

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int M = (int)d.M, N = (int)d.N, K = (int)d.K;
  const int Neff = N + (d.ones_out ? 1 : 0);
  const int m0 = blockIdx.x * G_BM;
  const int n0 = blockIdx.y * BN;

  int kbeg = 0, kend = K;
  if (d.split_k > 1) {
    const int ktiles = (K + G_BK - 1) / G_BK;
    const int per = (ktiles + d.split_k - 1) / d.split_k;
    kbeg = (int)blockIdx.z * per * G_BK;
    kend = min(K, kbeg + per * G_BK);
    if (kbeg >= kend) return;
  }
  const int nkt = (kend - kbeg + G_BK - 1) / G_BK;

  const bool a_kfast = (d.a_rs != 1);
  const bool b_kfast = (d.b_cs != 1);
  const float* __restrict__ Ap = d.A + d.a_off;
  const float* __restrict__ Bp = d.B + d.b_off;

  // per-thread pair coordinates inside a tile: (row, k-pair); fixed across k-tiles
  int a_mm[A_PAIRS], a_kp[A_PAIRS];
  unsigned a_rowmask = 0;
#pragma unroll
  for (int i = 0; i < A_PAIRS; ++i) {
    const int e = t + i * G_THREADS;
    if (a_kfast) { a_kp[i] = e % KP; a_mm[i] = e / KP; }
    else         { a_mm[i] = e % G_BM; a_kp[i] = e / G_BM; }
    if (d.mask_operand == 1 && (m0 + a_mm[i]) % d.mask_period == d.mask_phase) a_rowmask |= (1u << i);
  }
  int b_nn[B_PAIRS], b_kp[B_PAIRS];
#pragma unroll
  for (int i = 0; i < B_PAIRS; ++i) {
    const int e = t + i * G_THREADS;
    if (b_kfast) { b_kp[i] = e % KP; b_nn[i] = e / KP; }
    else         { b_nn[i] = e % BN; b_kp[i] = e / BN; }
  }

  float ra[A_PAIRS][2], rb[B_PAIRS][2];
  auto load_tile = [&](int kt) {
    const int k0 = kbeg + kt * G_BK;
#pragma unroll
    for (int i = 0; i < A_PAIRS; ++i) {
      const int m = m0 + a_mm[i];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = k0 + 2 * a_kp[i] + q;
        bool ok = (m < M) && (k < kend);
        if ((a_rowmask >> i) & 1u) ok = ok && !(k >= d.mask_lo && k < d.mask_hi);
#ifdef GX_ABL_NOLOAD      // timing ablation only (wrong results)
        ra[i][q] = ok ? 1.f : 0.f;
#else
        ra[i][q] = ok ? Ap[(int64_t)m * d.a_rs + (int64_t)k * d.a_cs] : 0.f;
#endif
      }
    }
#pragma unroll
    for (int i = 0; i < B_PAIRS; ++i) {
      const int e = t + i * G_THREADS;
      const int n = n0 + b_nn[i];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = k0 + 2 * b_kp[i] + q;
        float v = 0.f;
        if (e < BN * KP && k < kend) {
          if (n < N) {
            bool ok = true;
            if (d.mask_operand == 2) ok = !((k % d.mask_period == d.mask_phase) && n >= d.mask_lo && n < d.mask_hi);
#ifdef GX_ABL_NOLOAD
            if (ok) v = 1.f;
#else
            if (ok) v = Bp[(int64_t)k * d.b_rs + (int64_t)n * d.b_cs];
#endif
          } else if (n == N && d.ones_out) {
            v = 1.f;
          }
        }
        rb[i][q] = v;
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < A_PAIRS; ++i) {
      unsigned img[NS];
      gx_split<NS>(ra[i][0], ra[i][1], img);
      const int off = a_mm[i] * GX_PITCH + a_kp[i] * 4;
#pragma unroll
      for (int q = 0; q < NS; ++q) *reinterpret_cast<unsigned*>(base + q * A_IMG + off) = img[q];
    }
#pragma unroll
    for (int i = 0; i < B_PAIRS; ++i) {
      const int e = t + i * G_THREADS;
      if (e < BN * KP) {
        unsigned img[NS];
        gx_split<NS>(rb[i][0], rb[i][1], img);
        const int off = NS * A_IMG + b_nn[i] * GX_PITCH + b_kp[i] * 4;
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<unsigned*>(base + q * B_IMG + off) = img[q];
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  // fragment of a 16-k tile: lane (row = lane&31, h = lane>>5) takes k = 8h .. 8h+7 of its row
  const int frag_a = (wave * 32 + (lane & 31)) * GX_PITCH + (lane >> 5) * 16;
  const int frag_b = NS * A_IMG + (lane & 31) * GX_PITCH + (lane >> 5) * 16;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* base = smem + buf * BUF_BYTES;
    gx_u32x4 af[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) af[q] = *reinterpret_cast<const gx_u32x4*>(base + q * A_IMG + frag_a);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      gx_u32x4 bf[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) bf[q] = *reinterpret_cast<const gx_u32x4*>(base + q * B_IMG + frag_b + j * 32 * GX_PITCH);
      // every product of pieces whose magnitudes multiply to >= 2^-8(NS-1) of the full product, smallest terms first
#pragma unroll
      for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
        for (int qa = 0; qa <= sum; ++qa) {
#ifdef GX_ABL_NOMFMA     // timing ablation only (wrong results): one product instead of NS (NS + 1) / 2
          if (sum == 0)
#endif
          acc[j] = gx_mfma(af[qa], bf[sum - qa], acc[j]);
        }
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }
  gemm_epilogue<NT>(d, acc, m0, n0, wave, lane, M, N, Neff);
}

// fold split-K slabs in fixed order and apply the descriptor's epilogue: C[row][col] = epi(sum_z slab[z][row][col]);
// virtual ones column -> ones_out.  With the slabs, split-K is deterministic AND compatible with every epilogue, so the
// host layer also uses it to spread short-M GEMMs (a handful of 128-row tiles with K = 768..1024) over the chip.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const a2f_gemm_desc d, int splits, int Neff) {
  const int M = (int)d.M, N = (int)d.N;
  const int64_t total = (int64_t)M * Neff;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float* __restrict__ ws = d.splitk_ws;
  float v = 0.f;
  for (int z = 0; z < splits; ++z) v += ws[(int64_t)z * total + i];
  const int row = (int)(i / Neff), col = (int)(i - (int64_t)row * Neff);
  if (col == N) {
    d.ones_out[row] = d.accumulate ? d.ones_out[row] + v : v;
    return;
  }
  float* cp = d.C + (int64_t)row * d.ldc + col;
  if (d.bias) v += d.bias[col];
  if (d.accumulate) v += *cp;
  if (d.act == 1) v = fmaxf(v, 0.f);
  else if (d.act == 2) v = tanhf(v);
  if (d.aux_mode) {
    const float ax = d.aux[(int64_t)row * d.ldaux + col];
    if (d.aux_mode == 1) v = ax > 0.f ? v : 0.f;
    else if (d.aux_mode == 2) v *= (1.f - ax * ax);
    else v += ax;
  }
  *cp = v;
}

template <int BN, int NS>
static int split_launch(const a2f_gemm_desc& d, dim3 grid, hipStream_t s) {
  const size_t lds = 2 * (size_t)NS * (G_BM + BN) * GX_PITCH;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<BN, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_err(A2F_ELAUNCH, "gemm: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_split_kernel<BN, NS>), grid, dim3(G_THREADS), lds, s, d);
  return A2F_OK;
}

}  // namespace a2f

extern "C" int a2f_gemm_f32(const a2f_gemm_desc* dp, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dp != nullptr, "gemm: null descriptor");
  const a2f_gemm_desc d = *dp;
  A2F_CHECK_ARG(d.A && d.B && d.C, "gemm: null operand pointer");
  A2F_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "gemm: empty problem M=%lld N=%lld K=%lld", (long long)d.M, (long long)d.N, (long long)d.K);
  A2F_CHECK_ARG(d.M < (1ll << 30) && d.N < (1ll << 30) && d.K < (1ll << 30), "gemm: dimension too large");
  A2F_CHECK_ARG(d.ldc >= d.N, "gemm: ldc %lld < N %lld", (long long)d.ldc, (long long)d.N);
  A2F_CHECK_ARG(d.act >= 0 && d.act <= 2 && d.aux_mode >= 0 && d.aux_mode <= 3, "gemm: bad act/aux_mode");
  A2F_CHECK_ARG(d.aux_mode == 0 || d.aux != nullptr, "gemm: aux_mode without aux");
  A2F_CHECK_ARG(d.mask_operand >= 0 && d.mask_operand <= 2, "gemm: bad mask_operand");
  A2F_CHECK_ARG(d.mask_operand == 0 || d.mask_period > 0, "gemm: mask_period must be > 0");
  const int split = d.split_k > 1 ? d.split_k : 1;
  if (split > 1 && !d.splitk_ws)
    A2F_CHECK_ARG(d.act == 0 && d.aux_mode == 0 && d.bias == nullptr && !d.accumulate,
                  "gemm: atomic split_k (no splitk_ws) only with a linear epilogue into a zeroed C");
  A2F_CHECK_ARG(split <= 65535, "gemm: split_k too large");
  const int64_t neff = d.N + (d.ones_out ? 1 : 0);
  const int bn = neff <= 32 ? 32 : neff <= 64 ? 64 : neff <= 96 ? 96 : 128;
  dim3 grid((unsigned)cdiv(d.M, G_BM), (unsigned)cdiv(neff, bn), (unsigned)split);
  A2F_CHECK_ARG(grid.y <= 65535, "gemm: N too large for grid.y");
  hipStream_t s = (hipStream_t)stream;
  A2F_CHECK_ARG(d.precision >= 0 && d.precision <= 2, "gemm: precision must be 0 (exact fp32 MFMA), 1 (bf16x3) or 2 (bf16x6 split products)");
  if (d.precision == 1 || d.precision == 2) {
    const int rc = d.precision == 1 ? (bn == 32 ? split_launch<32, 2>(d, grid, s) : bn == 64 ? split_launch<64, 2>(d, grid, s) : bn == 96 ? split_launch<96, 2>(d, grid, s) : split_launch<128, 2>(d, grid, s))
                                    : (bn == 32 ? split_launch<32, 3>(d, grid, s) : bn == 64 ? split_launch<64, 3>(d, grid, s) : bn == 96 ? split_launch<96, 3>(d, grid, s) : split_launch<128, 3>(d, grid, s));
    if (rc != A2F_OK) return rc;
  } else {
    switch (bn) {
      case 32: hipLaunchKernelGGL(gemm_f32_kernel<32>, grid, dim3(G_THREADS), 0, s, d); break;
      case 64: hipLaunchKernelGGL(gemm_f32_kernel<64>, grid, dim3(G_THREADS), 0, s, d); break;
      case 96: hipLaunchKernelGGL(gemm_f32_kernel<96>, grid, dim3(G_THREADS), 0, s, d); break;
      default: hipLaunchKernelGGL(gemm_f32_kernel<128>, grid, dim3(G_THREADS), 0, s, d); break;
    }
  }
  if (split > 1 && d.splitk_ws) {
    // a split whose K range is empty stores nothing: clamp the fold to the splits that exist
    const int ktiles = (int)cdiv(d.K, G_BK);
    const int per = (ktiles + split - 1) / split;
    const int live = (ktiles + per - 1) / per;
    const int64_t total = d.M * neff;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, d, live, (int)neff);
  }
  A2F_CHECK_LAUNCH("gemm_f32");
  return A2F_OK;
}
