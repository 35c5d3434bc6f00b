// Weight-gradient GEMM for tiny outputs and huge contractions (gfx950):
//     C[M,N] = sum_k A(m,k) B(k,n),  M <= 128, N+1 <= 224, K = frames x positions (1e4 .. 2e5)
// i.e. dW = dY^T X (+ dbias through a virtual ones column) of every Linear / implicit-GEMM convolution of the VOCA trunk.
//
// The generic tiled GEMM is the wrong shape for this: one 128x128 tile, K split over hundreds of workgroups, three
// quarters of every tile masked, and a serial fold of the K-splits.  Here the whole M x N output is ONE register tile
// set per WAVE (MT x NT accumulators of the 32x32x2 f32 MFMA), every wave of every workgroup owns a slice of K, and both
// operands are read from global memory directly in MFMA operand layout - A is stored (K, M) with M contiguous and B is
// stored (K, N) with N contiguous, so a wave-instruction fetches two full 128-byte row segments: coalesced, no LDS, no
// barriers in the K loop.  The four waves fold through LDS into one partial slab per workgroup; `wgrad_fold_kernel` sums the
// slabs in fixed order (deterministic).
#include "a2f_common.h"

namespace a2f {

constexpr int WG_THREADS = 256;
constexpr int WG_UNROLL = 4;          // k-steps (of 2 rows) whose loads are in flight together

template <int MT, int NT>
__global__ __launch_bounds__(WG_THREADS, 1) void wgrad_kernel(const a2f_gemm_desc d, int rows_per_wave) {
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = (int)d.M, N = (int)d.N, K = (int)d.K;
  const int Neff = N + (d.ones_out ? 1 : 0);
  const int kbeg = (blockIdx.x * 4 + wave) * rows_per_wave;
  const int kend = min(K, kbeg + rows_per_wave);
  const float* __restrict__ Ap = d.A + d.a_off;
  const float* __restrict__ Bp = d.B + d.b_off;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += 2 * WG_UNROLL) {
    float av[WG_UNROLL][MT], bv[WG_UNROLL][NT];
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) {
      const int k = k0 + 2 * u + h;                        // this lane's contraction row of k-step u
      const bool kok = k < kend;
      const bool pad = (d.mask_operand == 2) && (k % d.mask_period == d.mask_phase);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = i * 32 + l31;
        av[u][i] = (kok && m < M) ? Ap[(int64_t)k * d.a_cs + m] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + l31;
        float v = 0.f;
        if (kok) {
          if (n < N) { if (!(pad && n >= d.mask_lo && n < d.mask_hi)) v = Bp[(int64_t)k * d.b_rs + n]; }
          else if (n == N && d.ones_out) v = 1.f;
        }
        bv[u][j] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
  }

  // fold the four waves of the workgroup through LDS, tile by tile, and leave ONE partial slab per workgroup:
  // [m][n], pitch Neff.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  __shared__ float red[4][1024];
  float* slab = d.splitk_ws + (int64_t)blockIdx.x * M * Neff;
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
}

// out[i] = sum_z slab[z][i]: 16 consecutive outputs x 16 z-lanes per workgroup (64-byte coalesced segments), every lane
// keeps 4 independent partial sums so its loads are in flight together; fixed summation order (deterministic).
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* __restrict__ ws, int slabs, int M, int N, int Neff,
                                                          float* __restrict__ C, int64_t ldc, float* __restrict__ ones_out) {
  __shared__ float sh[16][17];
  const int t = threadIdx.x, o = t & 15, zg = t >> 4;
  const int64_t total = (int64_t)M * Neff;
  const int64_t i = (int64_t)blockIdx.x * 16 + o;
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
    const int row = (int)(i / Neff), col = (int)(i - (int64_t)row * Neff);
    if (col < N) C[(int64_t)row * ldc + col] = s;
    else ones_out[row] = s;
  }
}

struct WgradPlan { int rows_per_wave, blocks, slabs; };
static WgradPlan wgrad_plan(int64_t K) {
  // aim at ~300 workgroups (256 CUs, one 512-register wave per SIMD), at least 32 rows per wave
  int64_t rpw = cdiv(cdiv(K, 300 * 4), 8) * 8;
  if (rpw < 32) rpw = 32;
  WgradPlan p;
  p.rows_per_wave = (int)rpw;
  p.blocks = (int)cdiv(K, rpw * 4);
  p.slabs = p.blocks;                 // one slab per workgroup (waves folded in LDS)
  return p;
}

template <int MT, int NT>
static void wgrad_launch(const a2f_gemm_desc& d, const WgradPlan& p, hipStream_t s) {
  hipLaunchKernelGGL((wgrad_kernel<MT, NT>), dim3(p.blocks), dim3(WG_THREADS), 0, s, d, p.rows_per_wave);
}

}  // namespace a2f

extern "C" size_t a2f_wgrad_ws_floats(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  return (size_t)a2f::wgrad_plan(K).slabs * (size_t)M * (size_t)(N + 1);
}

extern "C" int a2f_wgrad_supported(int64_t M, int64_t N) {
  const int mt = (int)((M + 31) / 32), nt = (int)((N + 1 + 31) / 32);
  return ((mt == 1 || mt == 2) ? nt <= 7 : mt == 3 ? nt <= 4 : mt == 4 ? nt <= 3 : false) ? 1 : 0;
}

extern "C" int a2f_wgrad_f32(const a2f_gemm_desc* dp, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dp != nullptr, "wgrad: null descriptor");
  const a2f_gemm_desc d = *dp;
  A2F_CHECK_ARG(d.A && d.B && d.C && d.splitk_ws, "wgrad: null pointer (A, B, C and splitk_ws are required)");
  A2F_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0 && d.K < (1ll << 30), "wgrad: bad shape");
  A2F_CHECK_ARG(a2f_wgrad_supported(d.M, d.N), "wgrad: output %lld x %lld too large for the register-tile kernel", (long long)d.M, (long long)d.N);
  A2F_CHECK_ARG(d.a_rs == 1 && d.b_cs == 1, "wgrad: needs A stored (K,M) with M contiguous and B stored (K,N) with N contiguous");
  A2F_CHECK_ARG(d.act == 0 && d.aux_mode == 0 && d.bias == nullptr && !d.accumulate, "wgrad: linear epilogue only");
  A2F_CHECK_ARG(d.mask_operand == 0 || (d.mask_operand == 2 && d.mask_period > 0), "wgrad: only a B-operand window mask is supported");
  A2F_CHECK_ARG(d.ldc >= d.N, "wgrad: ldc < N");
  const WgradPlan p = wgrad_plan(d.K);
  hipStream_t s = (hipStream_t)stream;
  const int mt = (int)((d.M + 31) / 32), nt = (int)((d.N + (d.ones_out ? 1 : 0) + 31) / 32);
  // smallest instantiated tile set that covers (mt, nt)
  if (mt == 1 && nt <= 4) wgrad_launch<1, 4>(d, p, s);
  else if (mt == 1) wgrad_launch<1, 7>(d, p, s);
  else if (mt == 2 && nt <= 4) wgrad_launch<2, 4>(d, p, s);
  else if (mt == 2 && nt <= 5) wgrad_launch<2, 5>(d, p, s);
  else if (mt == 2) wgrad_launch<2, 7>(d, p, s);
  else if (mt == 3 && nt <= 3) wgrad_launch<3, 3>(d, p, s);
  else if (mt == 3) wgrad_launch<3, 4>(d, p, s);
  else if (nt <= 3) wgrad_launch<4, 3>(d, p, s);
  else return set_err(A2F_EINVAL, "wgrad: no tile set for %d x %d tiles", mt, nt);
  const int64_t neff = d.N + (d.ones_out ? 1 : 0);
  const int64_t total = d.M * neff;
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)cdiv(total, 16)), dim3(256), 0, s, (const float*)d.splitk_ws, p.slabs, (int)d.M, (int)d.N,
                     (int)neff, d.C, d.ldc, d.ones_out);
  A2F_CHECK_LAUNCH("wgrad_f32");
  return A2F_OK;
}
