// bf16-MFMA GEMM on operands that arrive as bf16 hi / lo images (a2f_mm_operand.format 1): LDS-DMA staging, no conversion on chip.
// Part of the a2f_mm family (entry point and the fp32-operand kernels: csrc/gemm_mfma.hip; reference work replaced: the dense
// contractions of the wav2vec2 encoder behind FaceFormer, /root/reference/src/model/wav2vec.py:116,147,174-180).
#include "gemm_mfma.h"

namespace a2f {
namespace mm {

// =====================================================================================================================
// Operands that ARRIVE as bf16 hi / lo images (a2f_mm_operand.format 1; produced by a2f_split_bf16 or by a producing kernel's
// epilogue): nothing is converted here and nothing passes through registers - the k-tile is copied global -> LDS by the LDS-DMA
// form of the buffer load (buffer_load_dwordx4 ... lds: a wave instruction fills 1 KB of LDS, lane l at + 16 l), which removes
// the 40 conversion instructions, the eight ds_writes and the 16 staging registers per thread and k-tile of the fp32 path
// (its measured price: 30-45 % of the kernel).  An LDS-DMA cannot scatter, so the swizzles live on the SOURCE side:
//   R image (k contiguous), tile [X lines][32 k] in 64-byte rows: LDS slot (line, s) holds source chunk s ^ ((line >> 2) & 3) - the
//     layout the fp32 path writes, read back by the same one-ds_read_b128 fragment;
//   T image (lines contiguous), tile [32 k][X lines] with UNPADDED rows of 2 X bytes (a padded row would put lanes into the pad): the
//     64-byte blocks of row k are XORed with (k & 3), so the four k-rows a transposing read touches per lane group land in four
//     different bank quarters (conflict-free, checked exhaustively off line like the other layouts).
// Plain operands only (no windows), K a multiple of 32; one barrier per k-tile: wait for the own DMA of tile t, barrier (everybody's
// tile t is in LDS and everybody is done reading tile t - 1), issue the DMA of tile t + 1 into the other stage, multiply tile t.
// =====================================================================================================================
template <bool T, int X, int TH>
struct StageImg {
  static constexpr int IMG = X * 64;                          // bytes of one bf16 image tile (R: X x 64 B; T: 32 x 2 X B)
  static constexpr int NI = IMG / (TH * 16);                  // DMA instructions per thread and image
  static constexpr int PITCH = T ? 2 * X : 64;
  static_assert(NI >= 1 && IMG % (TH * 16) == 0, "image tile must be a whole number of 1-KB wave pieces per wave");
  unsigned voff[NI];                                          // byte offset of this lane's 16-byte chunk from the k-tile's base
  int piece[NI];                                              // wave-uniform byte offset of the 1-KB piece inside the image

  __device__ __forceinline__ void init(const Op& op, int x0, int xcount, int t) {
    const int wave = t >> 6, p = t & 63;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int pc = wave + (TH / 64) * i;
      piece[i] = __builtin_amdgcn_readfirstlane(pc * 1024);
      if constexpr (!T) {
        const int line = pc * 16 + (p >> 2), slot = p & 3;
        const int c = slot ^ ((line >> 2) & 3);
        const int gl = x0 + line;
        voff[i] = gl < xcount ? ((unsigned)gl * (unsigned)op.ld + (unsigned)(c * 8)) * 2u : 0xFFFFFFFFu;
      } else {
        constexpr int CPR = 2 * X / 16;                       // 16-byte chunks per k-row
        const int krow = pc * (64 / CPR) + p / CPR, slot = p % CPR;
        const int c16 = ((((slot >> 2) ^ (krow & 3)) << 2) | (slot & 3));
        const int col = x0 + c16 * 8;
        voff[i] = col < xcount ? ((unsigned)krow * (unsigned)op.ld + (unsigned)col) * 2u : 0xFFFFFFFFu;
      }
    }
  }
  // buffer resources of the hi / lo images for the k-tile starting at k0 (size 0 past the k-range: such a tile reads as zeros)
  __device__ __forceinline__ static rsrc_t tile_rsrc(const void* img, const Op& op, int k0, int kend) {
    const long long adv = (T ? (long long)k0 * op.ld : (long long)k0) * 2;             // bytes; wave-uniform
    const unsigned left = k0 < kend ? op.bytes - (unsigned)adv : 0u;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(img) + adv), 0, (int)left, 0x00020000);
  }
  template <int PREC>
  __device__ __forceinline__ void issue(const Op& op, int k0, int kend, char* stage) const {
#if defined(__HIP_DEVICE_COMPILE__)          // the LDS address-space cast only parses in the device pass
    const rsrc_t rh = tile_rsrc(op.p, op, k0, kend);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (__attribute__((address_space(3))) void*)(stage + piece[i]), 16, voff[i], 0, 0, 0);
    if constexpr (PREC == 1) {
      const rsrc_t rl = tile_rsrc(op.p_lo, op, k0, kend);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (__attribute__((address_space(3))) void*)(stage + IMG + piece[i]), 16, voff[i], 0, 0, 0);
    }
#endif
  }
  __device__ __forceinline__ static u32x4_t frag(const char* img, int rb, int ks, int lane) {
    if constexpr (!T) {
      const int line = rb + (lane & 31);
      const int chunk = ks * 2 + (lane >> 5);
      return *reinterpret_cast<const u32x4_t*>(img + line * 64 + ((chunk ^ ((line >> 2) & 3)) << 4));
    } else {
      const int g = lane >> 4, i = lane & 15;
      const int krow = ks * 16 + 8 * (g >> 1) + (i >> 2);
      const int byte = (rb + 16 * (g & 1) + 4 * (i & 3)) * 2;
      const char* p = img + krow * PITCH + ((((byte >> 6) ^ (krow & 3)) << 6) | (byte & 63));
      const u32x2_t r0 = tr_read(p);
      const u32x2_t r1 = tr_read(p + 4 * PITCH);              // (krow + 4) & 3 == krow & 3: same block permutation
      return u32x4_t{r0.x, r0.y, r1.x, r1.y};
    }
  }
};

// Tile geometries of the image kernel (MODE):
//   0: 128 x 128 on 8 waves of 32 x 64 (the CFG 3 geometry, four waves per SIMD);  1: 128 x 128 on 4 waves of 64 x 64 (two waves per SIMD,
//   twice the MFMAs per fragment read and per barrier);  2: MODE 0 with three LDS stages;
//   3: 256 x 256 on 8 waves of 128 x 64 (one workgroup per CU, 128 KB of LDS): per k-tile and wave 48 MFMAs against 24 fragment reads, and
//      half the staged bytes per output of the 128 x 128 tile - the LDS array (fragment reads + DMA writes) drops from ~80 % to ~40 % of
//      the matrix time;  4: 256 x 128 on 8 waves of 64 x 64.
template <int MODE>
struct ImgCfg {
  static constexpr int BM = MODE >= 3 ? 256 : 128, BN = MODE == 3 ? 256 : 128;
  static constexpr int TH = MODE == 1 ? 256 : 512;
  static constexpr int WGN = MODE == 3 ? 4 : 2;
  static constexpr int TM = MODE == 3 ? 4 : (MODE == 1 || MODE == 4) ? 2 : 1, TN = 2;
  static constexpr int WPS = MODE == 0 ? 4 : 2;               // waves per SIMD the register budget is set for
  static constexpr int STAGES = MODE == 2 ? 3 : 2;
};

template <bool AT, bool BT, int PREC, int MODE>
__global__ __launch_bounds__(ImgCfg<MODE>::TH, ImgCfg<MODE>::WPS) void gemm_img_kernel(const Args a) {
  constexpr bool ST3 = MODE == 2;        // three LDS stages, one workgroup per CU: the DMA of tile t + 2 is in flight across the barrier of tile t + 1
  constexpr int BM = ImgCfg<MODE>::BM, BN = ImgCfg<MODE>::BN, TH = ImgCfg<MODE>::TH, WGN = ImgCfg<MODE>::WGN, TM = ImgCfg<MODE>::TM, TN = ImgCfg<MODE>::TN;
  constexpr int NIMG = PREC == 1 ? 2 : 1;
  using SA = StageImg<AT, BM, TH>;
  using SB = StageImg<BT, BN, TH>;
  constexpr int A_BYTES = NIMG * SA::IMG, B_BYTES = NIMG * SB::IMG, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  int m0, n0;
  tile_origin(a, BM, BN, m0, n0);
  const int bz = 0;
  const int sk = __builtin_amdgcn_readfirstlane((int)blockIdx.z);
  int kbeg = 0, kend = a.K;
  if (a.split_k > 1) {
    const int ktiles = (a.K + BK - 1) / BK;
    const int per = (ktiles + a.split_k - 1) / a.split_k;
    kbeg = sk * per * BK;
    kend = min(a.K, kbeg + per * BK);
  }
  const int nkt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  SA sa;
  SB sb;
  sa.init(a.A, m0, a.M, t);
  sb.init(a.B, n0, a.N, t);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  sa.template issue<PREC>(a.A, kbeg, kend, smem);
  sb.template issue<PREC>(a.B, kbeg, kend, smem + A_BYTES);
  if constexpr (ST3) {                                          // second tile in flight before the loop (zeros / no traffic past the k-range)
    sa.template issue<PREC>(a.A, kbeg + BK, kend, smem + STAGE);
    sb.template issue<PREC>(a.B, kbeg + BK, kend, smem + STAGE + A_BYTES);
  }
  int st_cur = 0;                                               // ST3: stage of tile kt
  for (int kt = 0; kt < nkt; ++kt) {
    const char* cur;
    if constexpr (ST3) {
      // tile kt has landed when at most the instructions of tile kt + 1 are outstanding (counted wait: never 0 inside the loop);
      // the raw barrier then publishes it - and says everybody is done reading tile kt - 1, whose stage the next DMA overwrites
      constexpr int PER_TILE = (SA::NI + SB::NI) * NIMG;
      if constexpr (PER_TILE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (PER_TILE == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int st_nxt = st_cur + 2; st_nxt = st_nxt >= 3 ? st_nxt - 3 : st_nxt;
      sa.template issue<PREC>(a.A, kbeg + (kt + 2) * BK, kend, smem + st_nxt * STAGE);
      sb.template issue<PREC>(a.B, kbeg + (kt + 2) * BK, kend, smem + st_nxt * STAGE + A_BYTES);
      cur = smem + st_cur * STAGE;
      st_cur = st_cur == 2 ? 0 : st_cur + 1;
    } else {
      __syncthreads();                                          // the compiler drains the DMA (vmcnt 0) in front of the barrier
      if (kt + 1 < nkt) {
        char* nst = smem + ((kt + 1) & 1) * STAGE;
        sa.template issue<PREC>(a.A, kbeg + (kt + 1) * BK, kend, nst);
        sb.template issue<PREC>(a.B, kbeg + (kt + 1) * BK, kend, nst + A_BYTES);
      }
      cur = smem + (kt & 1) * STAGE;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t fa[NIMG][TM], fb[NIMG][TN];
#pragma unroll
      for (int q = 0; q < NIMG; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[q][i] = SA::frag(cur + q * SA::IMG, (wm * TM + i) * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[q][j] = SB::frag(cur + A_BYTES + q * SB::IMG, (wn * TN + j) * 32, ks, lane);
      }
      if constexpr (PREC == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma(fa[1][i], fb[0][j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma(fa[0][i], fb[1][j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma(fa[0][i], fb[0][j], acc[i][j]);
    }
  }
  tile_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, lane, bz);
}

template <bool AT, bool BT, int PREC, int MODE>
static int launch_img(const Args& a, hipStream_t st) {
  constexpr int NIMG = PREC == 1 ? 2 : 1;
  constexpr int TH = ImgCfg<MODE>::TH, BM = ImgCfg<MODE>::BM, BN = ImgCfg<MODE>::BN;
  constexpr int smem = ImgCfg<MODE>::STAGES * NIMG * (StageImg<AT, BM, TH>::IMG + StageImg<BT, BN, TH>::IMG);
  auto kern = gemm_img_kernel<AT, BT, PREC, MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return set_err(A2F_ELAUNCH, "a2f_mm: cannot reserve %d bytes of LDS", smem);
    attr_done = true;
  }
  dim3 grid((unsigned)(cdiv(a.N, BN) * cdiv(a.M, BM)), 1, (unsigned)a.split_k);
  hipLaunchKernelGGL(kern, grid, dim3(TH), smem, st, a);
  return A2F_OK;
}

template <int PREC, int MODE>
static int dispatch_img_mode(const Args& a, bool at, bool bt, hipStream_t st) {
  if (!at && !bt) return launch_img<false, false, PREC, MODE>(a, st);
  if (!at && bt) return launch_img<false, true, PREC, MODE>(a, st);
  if (at && bt) return launch_img<true, true, PREC, MODE>(a, st);
  return launch_img<true, false, PREC, MODE>(a, st);
}

template <int PREC>
static int dispatch_img(const Args& a, bool at, bool bt, hipStream_t st) {
  static const int forced = [] { const char* e = getenv("A2F_MM_IMG_MODE"); return e ? e[0] - '0' : -1; }();    // A-B hook: 0 .. 4
  int mode = forced;
  if (mode < 0) mode = 0;
  switch (mode) {
    case 1: return dispatch_img_mode<PREC, 1>(a, at, bt, st);
    case 2: return dispatch_img_mode<PREC, 2>(a, at, bt, st);
    case 3: return dispatch_img_mode<PREC, 3>(a, at, bt, st);
    case 4: return dispatch_img_mode<PREC, 4>(a, at, bt, st);
    default: return dispatch_img_mode<PREC, 0>(a, at, bt, st);
  }
}

int dispatch_img_precision(const Args& a, int precision, bool at, bool bt, hipStream_t st) {
  return precision == 1 ? dispatch_img<1>(a, at, bt, st) : dispatch_img<0>(a, at, bt, st);
}

}  // namespace mm
}  // namespace a2f
