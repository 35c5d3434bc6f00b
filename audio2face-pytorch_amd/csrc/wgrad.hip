// Weight-gradient GEMMs for tiny outputs and huge contractions (gfx950):
//     C[M,N] = sum_k A(m,k) B(k,n),  M <= 128, N+1 <= 224, K = frames x positions (1e4 .. 2e5)
// i.e. dW = dY^T X (+ dbias through a virtual ones column) of every Linear / implicit-GEMM convolution of the VOCA trunk.
//
// The generic tiled GEMM is the wrong shape for this: one 128x128 tile, K split over hundreds of workgroups, three
// quarters of every tile masked, and a serial fold of the K-splits.  Here the whole M x N output is ONE register tile
// set per WAVE (MT x NT accumulators of the 32x32x2 f32 MFMA), every wave of every workgroup owns a slice of K, and both
// operands are read from global memory directly in MFMA operand layout - A is stored (K, M) with M contiguous and B is
// stored (K, N) with N contiguous, so a wave-instruction fetches two full 128-byte row segments: coalesced, no LDS, no
// barriers in the K loop.  The loop is latency-bound (one wave per SIMD), so operand loads are double-buffered in
// registers: 2 x WG_U k-steps are in flight while the previous set feeds the matrix pipe; loads are raw buffer loads whose
// out-of-range offset (-1) returns 0, which is the K tail, the M/N edge and the convolution's left zero padding at once.
// The four waves fold through LDS into one partial slab per workgroup; the fold kernel sums the slabs in fixed order
// (deterministic) and can store a convolution gradient directly in nn.Conv2d's (Co, Ci, KH, 1) layout.
//
// Up to A2F_WGRAD_BATCH_MAX problems run in ONE launch pair (all seven trunk layers of a VOCA step): the workgroup index
// selects its problem from a table in the kernel arguments.
#include "a2f_common.h"

#include <string.h>

#include <algorithm>

namespace a2f {

constexpr int WG_THREADS = 256;
#ifdef WG_FOLD_ROW
constexpr int WG_NT_MAX = 7;
#elif defined(WG_LDS_BIG)
constexpr int WG_NT_MAX = 7;
#else
constexpr int WG_NT_MAX = 1;
#endif
#ifndef WG_U_SET
#define WG_U_SET 4
#endif
#ifndef WG_TARGET
#define WG_TARGET 800
#endif
constexpr int WG_U = WG_U_SET;               // k-steps (of 2 rows) per register set; two sets are in flight (<= 63 loads: the vmcnt limit)

using wrsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ wrsrc_t wg_rsrc(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float wg_load(wrsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// one problem of a batch, as the kernels see it (offsets folded into the pointers)
struct WgradProb {
  const float* A;                     // (K, M): element (k, m) at A[k * a_cs + m]
  const float* B;                     // (K, N): element (k, n) at B[k * b_rs + n]
  float* ws;                          // slabs [blocks][M][Neff]
  float* C;
  float* ones_out;
  int a_cs, b_rs, ldc;
  int M, N, K, Neff;
  int mask_period, mask_phase, mask_lo, mask_hi;   // mask_period == 0: no window mask
  int rows_per_wave, blocks, shape;                // shape = MT * 8 + NT of the instantiated tile set
  int unpack_kh;                                   // > 0: column kh*Ci+ci is stored at ci*KH+kh (Conv2d weight layout)
  int m_off, n_off, n_glob;                        // this sub-problem's corner in the caller's output, and its full width
  int blk_beg, fold_beg;                           // first workgroup of this problem in the GEMM / fold launch
  int x3;                                          // a2f_gemm_desc.precision == 1: bf16x3 split products on the bf16 matrix pipe
};

// A caller's problem is cut into sub-problems of at most 2 x 4 accumulator tiles (64 rows x 128 columns of the output):
// 128 accumulator registers, so the kernel fits 256 VGPRs and TWO workgroups share a CU - the loop is bound by the loads a
// CU keeps in flight, and the extra operand re-reads (A once per column range, B once per row range) are small.
constexpr int WG_MAX_SUBS = 16;
struct WgradBatch {
  int count;
  WgradProb p[WG_MAX_SUBS];
};

// the problem a workgroup belongs to: wave-uniform select chain over the kernel arguments (no scratch copy)
__device__ __forceinline__ WgradProb pick_problem(const WgradBatch& b, int blk, bool fold) {
  WgradProb pr = b.p[0];
#pragma unroll
  for (int i = 1; i < WG_MAX_SUBS; ++i)
    if (i < b.count && blk >= (fold ? b.p[i].fold_beg : b.p[i].blk_beg)) pr = b.p[i];
  return pr;
}

// fold the four waves of the workgroup through LDS and leave ONE partial slab per workgroup: [m][n], pitch Neff.
template <int MT, int NT>
__device__ __forceinline__ void wgrad_store_slab(const WgradProb& pr, int blk, f32x16 (&acc)[MT][NT], float (*red)[WG_NT_MAX * 1024], int t,
                                                 int wave, int l31, int h, int M, int Neff) {
  // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* slab = pr.ws + (int64_t)blk * M * Neff;
#ifdef WG_FOLD_ROW
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][j * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) {
      const int e = t + q * WG_THREADS;                    // tile j = e / 1024, row = (e % 1024) / 32, col = e % 32
      const int m = i * 32 + ((e & 1023) >> 5), n = (e >> 10) * 32 + (e & 31);
      if (m < M && n < Neff) slab[(int64_t)m * Neff + n] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
  }
#else
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[i][j][r];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = t + q * WG_THREADS;                  // 0..1023: row = e / 32, col = e % 32
        const int m = i * 32 + (e >> 5), n = j * 32 + (e & 31);
        if (m < M && n < Neff) slab[(int64_t)m * Neff + n] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
      }
    }
#endif
}

template <int MT, int NT>
__device__ __forceinline__ void wgrad_body(const WgradProb& pr, int blk, float (*red)[WG_NT_MAX * 1024]) {
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = pr.M, N = pr.N, Neff = pr.Neff;
  const int kbeg = (blk * 4 + wave) * pr.rows_per_wave;
  const int kend = kbeg + pr.rows_per_wave;              // a multiple of one register-set pair; rows >= K read as 0
  // exact operand spans: rows >= K (the tail of the last wave) and bytes past the last row read as 0; columns >= M / N
  // of an interior row read the neighbouring row - finite garbage that only reaches output rows / columns never stored
  const wrsrc_t ra = wg_rsrc(pr.A, ((pr.K - 1) * pr.a_cs + M) * 4), rb = wg_rsrc(pr.B, ((pr.K - 1) * pr.b_rs + N) * 4);
  const bool ones = Neff > N;
  const int lane_a = (h * pr.a_cs + l31) * 4, lane_b = (h * pr.b_rs + l31) * 4;   // + (k0 + 2u) * stride: row k0+2u+h
  const int step_a = 2 * pr.a_cs * 4, step_b = 2 * pr.b_rs * 4;
  // left zero padding of the convolution window: rows with (k % period) == phase drop columns [mask_lo, mask_hi)
  const bool masked = pr.mask_period > 0;
  const int period = masked ? pr.mask_period : 1, step_pos = 2 % period;
  int pos = (kbeg + h) % period;                           // advanced by 2 rows per k-step, in load order
  bool bmask[NT], bone[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 32 + l31;
    bmask[j] = masked && n >= pr.mask_lo && n < pr.mask_hi;
    bone[j] = ones && n == N;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_set = [&](int k0, float (&av)[WG_U][MT], float (&bv)[WG_U][NT]) {
    const int va0 = lane_a + k0 * pr.a_cs * 4, vb0 = lane_b + k0 * pr.b_rs * 4;
#pragma unroll
    for (int u = 0; u < WG_U; ++u) {
      const int va = va0 + u * step_a, vb = vb0 + u * step_b;
      const bool padrow = masked && pos == pr.mask_phase;
      const bool inside = k0 + 2 * u + h < pr.K;
      pos += step_pos;
      if (pos >= period) pos -= period;
#pragma unroll
      for (int i = 0; i < MT; ++i) av[u][i] = wg_load(ra, va + i * 128);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float v = wg_load(rb, (padrow && bmask[j]) ? -1 : vb + j * 128);
        bv[u][j] = (bone[j] && inside) ? 1.f : v;
      }
    }
  };
  auto mfma_set = [&](float (&av)[WG_U][MT], float (&bv)[WG_U][NT]) {
#pragma unroll
    for (int u = 0; u < WG_U; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#ifdef WG_ABL_NOMFMA     // timing ablation only
          if (u == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i] + av[(u + 1) % WG_U][i] + av[(u + 3) % WG_U][i] + av[(u + 5) % WG_U][i] + av[(u+2) % WG_U][i] + av[(u+4) % WG_U][i]+ av[(u+6) % WG_U][i]+ av[(u+7) % WG_U][i], bv[u][j]+bv[(u+1)%WG_U][j]+bv[(u+2)%WG_U][j]+bv[(u+3)%WG_U][j]+bv[(u+4)%WG_U][j]+bv[(u+5)%WG_U][j]+bv[(u+6)%WG_U][j]+bv[(u+7)%WG_U][j], acc[i][j], 0, 0, 0);
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
#endif
        }
  };

  float a0[WG_U][MT], b0[WG_U][NT], a1[WG_U][MT], b1[WG_U][NT];
  load_set(kbeg, a0, b0);
  for (int k0 = kbeg; k0 < kend; k0 += 4 * WG_U) {
    load_set(k0 + 2 * WG_U, a1, b1);
    mfma_set(a0, b0);
    if (k0 + 4 * WG_U < kend) load_set(k0 + 4 * WG_U, a0, b0);
    mfma_set(a1, b1);
  }

  wgrad_store_slab<MT, NT>(pr, blk, acc, red, t, wave, l31, h, M, Neff);
}

// ---- split-precision variant (a2f_gemm_desc.precision == 1) ---------------------------------------------------------
// The fp32 MFMA (32x32x2, 64 cycles per 2 rows) is what the loop above spends its issue time on; here every operand value is
// split x = hi + lo (bf16 each, round to nearest) and a product is lo*hi' + hi*lo' + hi*hi' on v_mfma_f32_32x32x16_bf16
// (16 rows per 3 x 32 cycles: 5.3x less matrix time, <= 2^-16 relative error per product, fp32 accumulation).  A register
// set is 16 rows: lane (l31, h) reads rows k0 + 8h + u (u = 0..7) - the k order of the bf16 operand fragments - converts
// them to packed hi / lo fragments, requests the next set into the same registers and multiplies.
using wbf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using wbf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using wf32x2_t = __attribute__((ext_vector_type(2))) float;
using wu32x4_t = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ void wx_split2(float a, float b, unsigned& hi, unsigned& lo) {
  const wf32x2_t v = {a, b};
  const wbf16x2_t hb = __builtin_convertvector(v, wbf16x2_t);
  const wf32x2_t r = v - __builtin_convertvector(hb, wf32x2_t);
  hi = __builtin_bit_cast(unsigned, hb);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wbf16x2_t));
}
__device__ __forceinline__ f32x16 wx_mfma(wu32x4_t a, wu32x4_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8_t, a), __builtin_bit_cast(wbf16x8_t, b), c, 0, 0, 0);
}

template <int MT, int NT>
__device__ __forceinline__ void wgrad_body_x3(const WgradProb& pr, int blk, float (*red)[WG_NT_MAX * 1024]) {
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = pr.M, N = pr.N, Neff = pr.Neff;
  const int kbeg = (blk * 4 + wave) * pr.rows_per_wave;
  const int kend = kbeg + pr.rows_per_wave;              // a multiple of 16; rows >= K read as 0
  const wrsrc_t ra = wg_rsrc(pr.A, ((pr.K - 1) * pr.a_cs + M) * 4), rb = wg_rsrc(pr.B, ((pr.K - 1) * pr.b_rs + N) * 4);
  const bool ones = Neff > N;
  const int lane_a = (8 * h * pr.a_cs + l31) * 4, lane_b = (8 * h * pr.b_rs + l31) * 4;     // + (k0 + u) * stride: row k0 + 8h + u
  const bool masked = pr.mask_period > 0;
  const int period = masked ? pr.mask_period : 1, adv8 = 8 % period;
  int pos = (kbeg + 8 * h) % period;                       // (row of the next load) mod period, in load order
  bool bmask[NT], bone[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 32 + l31;
    bmask[j] = masked && n >= pr.mask_lo && n < pr.mask_hi;
    bone[j] = ones && n == N;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float av[8][MT], bv[8][NT];
  auto load_set = [&](int k0) {
    const int va0 = lane_a + k0 * pr.a_cs * 4, vb0 = lane_b + k0 * pr.b_rs * 4;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int va = va0 + u * pr.a_cs * 4, vb = vb0 + u * pr.b_rs * 4;
      const bool padrow = masked && pos == pr.mask_phase;
      const bool inside = k0 + 8 * h + u < pr.K;
      pos = pos + 1 == period ? 0 : pos + 1;
#pragma unroll
      for (int i = 0; i < MT; ++i) av[u][i] = wg_load(ra, va + i * 128);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float v = wg_load(rb, (padrow && bmask[j]) ? -1 : vb + j * 128);
        bv[u][j] = (bone[j] && inside) ? 1.f : v;
      }
    }
    pos += adv8;                                           // this lane's next set starts 16 rows on: 8 counted above
    if (pos >= period) pos -= period;
  };
  load_set(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    wu32x4_t ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int u = 0; u < 8; u += 2) { unsigned hi, lo; wx_split2(av[u][i], av[u + 1][i], hi, lo); ah[i][u >> 1] = hi; al[i][u >> 1] = lo; }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < 8; u += 2) { unsigned hi, lo; wx_split2(bv[u][j], bv[u + 1][j], hi, lo); bh[j][u >> 1] = hi; bl[j][u >> 1] = lo; }
    if (k0 + 16 < kend) load_set(k0 + 16);                 // in flight under the products below
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] = wx_mfma(al[i], bh[j], acc[i][j]);
        acc[i][j] = wx_mfma(ah[i], bl[j], acc[i][j]);
        acc[i][j] = wx_mfma(ah[i], bh[j], acc[i][j]);
      }
  }
  wgrad_store_slab<MT, NT>(pr, blk, acc, red, t, wave, l31, h, M, Neff);
}

#ifndef WG_LB
#define WG_LB 2
#endif
__global__ __launch_bounds__(WG_THREADS, WG_LB) void wgrad_batch_kernel(const WgradBatch b) {
  __shared__ float red[4][WG_NT_MAX * 1024];
  const WgradProb pr = pick_problem(b, blockIdx.x, false);
  const int blk = blockIdx.x - pr.blk_beg;
  switch (pr.shape) {
    case 1 * 8 + 3: wgrad_body<1, 3>(pr, blk, red); break;
    case 1 * 8 + 4: wgrad_body<1, 4>(pr, blk, red); break;
    case 2 * 8 + 1: wgrad_body<2, 1>(pr, blk, red); break;
    case 2 * 8 + 3: wgrad_body<2, 3>(pr, blk, red); break;
    default: wgrad_body<2, 4>(pr, blk, red); break;
  }
}

// the split-precision launch: sub-problems are cut at 96 columns (2 x 3 tiles: 96 accumulators + two 40-register operand sets)
__global__ __launch_bounds__(WG_THREADS, WG_LB) void wgrad_batch_x3_kernel(const WgradBatch b) {
  __shared__ float red[4][WG_NT_MAX * 1024];
  const WgradProb pr = pick_problem(b, blockIdx.x, false);
  const int blk = blockIdx.x - pr.blk_beg;
  switch (pr.shape) {
    case 1 * 8 + 3: wgrad_body_x3<1, 3>(pr, blk, red); break;
    case 2 * 8 + 1: wgrad_body_x3<2, 1>(pr, blk, red); break;
    default: wgrad_body_x3<2, 3>(pr, blk, red); break;
  }
}

// out[i] = sum_z slab[z][i]: 16 consecutive outputs x 16 z-lanes per workgroup (64-byte coalesced segments), every lane
// keeps 4 independent partial sums so its loads are in flight together; fixed summation order (deterministic).
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const WgradBatch b) {
  __shared__ float sh[16][17];
  const WgradProb pr = pick_problem(b, blockIdx.x, true);
  const float* __restrict__ ws = pr.ws;
  const int slabs = pr.blocks;
  const int t = threadIdx.x, o = t & 15, zg = t >> 4;
  const int64_t total = (int64_t)pr.M * pr.Neff;
  const int64_t i = (int64_t)(blockIdx.x - pr.fold_beg) * 16 + o;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (i < total) {
    int z = zg;
    for (; z + 48 < slabs; z += 64) {
      v0 += ws[(int64_t)z * total + i];
      v1 += ws[(int64_t)(z + 16) * total + i];
      v2 += ws[(int64_t)(z + 32) * total + i];
      v3 += ws[(int64_t)(z + 48) * total + i];
    }
    for (; z < slabs; z += 16) v0 += ws[(int64_t)z * total + i];
  }
  sh[zg][o] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (t < 16 && i < total) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += sh[g][t];
    const int row = (int)(i / pr.Neff), col = (int)(i - (int64_t)row * pr.Neff);
    if (col < pr.N) {
      int c = pr.n_off + col;
      if (pr.unpack_kh > 0) { const int ci_n = pr.n_glob / pr.unpack_kh, kh = c / ci_n; c = (c - kh * ci_n) * pr.unpack_kh + kh; }
      pr.C[(int64_t)(pr.m_off + row) * pr.ldc + c] = s;
    } else {
      pr.ones_out[pr.m_off + row] = s;
    }
  }
}

struct WgradPlan { int rows_per_wave, blocks; };

// smallest instantiated tile set that covers a sub-problem of (mt <= 2, nt <= 4) tiles
static int wgrad_shape(int mt, int nt) {
  if (mt == 1) return nt <= 3 ? 1 * 8 + 3 : 1 * 8 + 4;
  return nt == 1 ? 2 * 8 + 1 : nt <= 3 ? 2 * 8 + 3 : 2 * 8 + 4;
}

// The loop moves (MT+NT) x 256 B per k-step and wave, so a workgroup that owns ~800/(MT+NT) rows per wave moves ~0.4 MB:
// the trunk problems of a 19 200-frame step then make several hundred equally heavy workgroups.  Short contractions are cut
// into at least ~64 workgroups; a wave's slice is a multiple of one register set pair.  All sub-problems of a caller's
// problem share its plan (same number of slabs), so the workspace bound blocks x M x (N+1) holds for their union.
static WgradPlan wgrad_plan(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = std::min<int64_t>((M + 31) / 32, 2) + std::min<int64_t>((N + 1 + 31) / 32, 4);
  int64_t rpw = std::min<int64_t>(WG_TARGET / tiles, cdiv(K, 4 * 64));
  rpw = std::max<int64_t>(4 * WG_U, cdiv(rpw, 4 * WG_U) * (4 * WG_U));
  WgradPlan p;
  p.rows_per_wave = (int)rpw;
  p.blocks = (int)cdiv(K, rpw * 4);
  return p;
}

}  // namespace a2f

extern "C" size_t a2f_wgrad_ws_floats(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  return (size_t)a2f::wgrad_plan(M, N, K).blocks * (size_t)M * (size_t)(N + 1);
}

extern "C" int a2f_wgrad_supported(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  return (M <= 128 && N + 1 <= 224) ? 1 : 0;
}

extern "C" int a2f_wgrad_batch_f32(const a2f_gemm_desc* descs, int32_t count, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(descs != nullptr && count >= 1 && count <= A2F_WGRAD_BATCH_MAX, "wgrad: 1..%d descriptors per batch", A2F_WGRAD_BATCH_MAX);
  hipStream_t s = (hipStream_t)stream;
  WgradBatch b;
  memset(&b, 0, sizeof(b));
  int blk = 0, fold = 0;
  bool cur_x3 = false;
  auto flush = [&]() {
    if (b.count == 0) return;
    if (cur_x3) hipLaunchKernelGGL(wgrad_batch_x3_kernel, dim3(blk), dim3(WG_THREADS), 0, s, b);
    else hipLaunchKernelGGL(wgrad_batch_kernel, dim3(blk), dim3(WG_THREADS), 0, s, b);
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3(fold), dim3(256), 0, s, b);
    memset(&b, 0, sizeof(b));
    blk = fold = 0;
  };
  for (int i = 0; i < count; ++i) {
    const a2f_gemm_desc& d = descs[i];
    A2F_CHECK_ARG(d.A && d.B && d.C && d.splitk_ws, "wgrad[%d]: null pointer (A, B, C and splitk_ws are required)", i);
    A2F_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0 && d.K < (1ll << 30), "wgrad[%d]: bad shape", i);
    A2F_CHECK_ARG(a2f_wgrad_supported(d.M, d.N), "wgrad[%d]: output %lld x %lld too large for the register-tile kernel", i, (long long)d.M,
                  (long long)d.N);
    A2F_CHECK_ARG(d.a_rs == 1 && d.b_cs == 1, "wgrad[%d]: needs A stored (K,M) with M contiguous and B stored (K,N) with N contiguous", i);
    A2F_CHECK_ARG(d.a_cs > 0 && d.b_rs > 0 && (d.K * d.a_cs + d.M) * 4 < 0x7FFFFFF0ll && (d.K * d.b_rs + d.N) * 4 < 0x7FFFFFF0ll,
                  "wgrad[%d]: operand spans 2 GB or more (use a2f_gemm_f32)", i);
    A2F_CHECK_ARG(d.act == 0 && d.aux_mode == 0 && d.bias == nullptr && !d.accumulate, "wgrad[%d]: linear epilogue only", i);
    A2F_CHECK_ARG(d.mask_operand == 0 || (d.mask_operand == 2 && d.mask_period > 0), "wgrad[%d]: only a B-operand window mask is supported", i);
    A2F_CHECK_ARG(d.ldc >= d.N, "wgrad[%d]: ldc < N", i);
    A2F_CHECK_ARG(d.c_unpack_kh >= 0 && (d.c_unpack_kh == 0 || d.N % d.c_unpack_kh == 0), "wgrad[%d]: c_unpack_kh must divide N", i);
    const WgradPlan pl = wgrad_plan(d.M, d.N, d.K);
    const int M = (int)d.M, N = (int)d.N, Neff = N + (d.ones_out ? 1 : 0);
    float* ws = d.splitk_ws;
    const bool x3 = d.precision == 1;
    if (b.count > 0 && x3 != cur_x3) flush();              // one precision per launch
    cur_x3 = x3;
    const int ncut = x3 ? 96 : 128;
    for (int m0 = 0; m0 < M; m0 += 64)
      for (int n0 = 0; n0 < Neff; n0 += ncut) {
        if (b.count == WG_MAX_SUBS) flush();
        const int m1 = std::min(M, m0 + 64), n1 = std::min(Neff, n0 + ncut);
        WgradProb& p = b.p[b.count++];
        p.A = d.A + d.a_off + m0;
        p.B = d.B + d.b_off + n0;
        p.ws = ws;
        p.C = d.C;
        p.ones_out = d.ones_out;
        p.a_cs = (int)d.a_cs; p.b_rs = (int)d.b_rs; p.ldc = (int)d.ldc;
        p.M = m1 - m0;
        p.Neff = n1 - n0;
        p.N = std::min(N, n1) - n0;                       // real columns of this range (the ones column, if any, is the last of the last range)
        p.K = (int)d.K;
        if (d.mask_operand == 2) { p.mask_period = d.mask_period; p.mask_phase = d.mask_phase; p.mask_lo = d.mask_lo - n0; p.mask_hi = d.mask_hi - n0; }
        p.rows_per_wave = pl.rows_per_wave;
        p.blocks = pl.blocks;
        p.shape = wgrad_shape((p.M + 31) / 32, (p.Neff + 31) / 32);
        p.unpack_kh = d.c_unpack_kh;
        p.x3 = x3 ? 1 : 0;
        p.m_off = m0; p.n_off = n0; p.n_glob = N;
        p.blk_beg = blk;
        p.fold_beg = fold;
        blk += pl.blocks;
        fold += (int)cdiv((int64_t)p.M * p.Neff, 16);
        ws += (int64_t)pl.blocks * p.M * p.Neff;
      }
  }
  flush();
  A2F_CHECK_LAUNCH("wgrad_batch_f32");
  return A2F_OK;
}

extern "C" int a2f_wgrad_f32(const a2f_gemm_desc* dp, a2f_stream_t stream) {
  A2F_CHECK_ARG(dp != nullptr, "wgrad: null descriptor");
  return a2f_wgrad_batch_f32(dp, 1, stream);
}
