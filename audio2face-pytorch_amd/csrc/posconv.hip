// The grouped positional convolution of the wav2vec2 encoder (HF Wav2Vec2PositionalConvEmbedding behind /root/reference/src/model/wav2vec.py:91-189:
// nn.Conv1d(768, 768, kernel_size = 128, padding = 64, groups = 16), last output frame dropped, GELU, added to its input) as bf16x3 MFMA kernels of its own.
// What it replaces: the batched windowed a2f_mm launches of rounds 2-4 (16 groups x [rows x 48 x 6144] on the 128 x 64 fp32-operand tile: 48 of 64 tile
// columns in use, every window element re-read and re-converted once per tap - 0.26 of the pipe forward / data gradient, 0.15 weight gradient).
//
// A group's convolution over one stretch of frames reads (frames + 127) x 48 inputs: that block is converted to bf16 hi / lo ONCE per workgroup and kept in LDS
// ([frame][48 hi | 48 lo], 208-byte pitch: 16 consecutive frames start in 16 different bank quads), and all 128 taps take their MFMA fragments from it at a row
// offset.
//   posconv_nt_kernel    y[s, t, g, co] = epi(sum_{kk, ci} x[s, t + off + kk, g, ci] * W[g][co][kk * 48 + ci])   forward (off = -64: bias, GELU, + input) and data
//                        gradient (off = -63, flipped / transposed weights, + dy).  Workgroup = (sequence, group, block of <= 384 frames); W arrives as a split-32
//                        image ([g][48 rows][6144]), 32-deep k-tiles of 6 KB by LDS-DMA; a k-tile is 32 consecutive k of a row = channels of one tap or the tails
//                        of two (8 consecutive k never straddle a tap: every lane group addresses its own frame).  8 waves = 4 (frame tiles) x 2 (even / odd
//                        k-tiles, ping-pong one barrier apart: one group's fragment reads under the other's MFMAs), summed through LDS at the end.
//   posconv_wgrad_kernel dW[g][kk][ci][co] = sum_{s, t} x[s, t - 64 + kk, g, ci] * dz[s, t, g, co].  Workgroup = (group, 8 taps) over ALL frames - no split-K slabs,
//                        fixed order.  8 waves = 4 (tap pairs: the dz fragments of a k-tile serve both taps) x 2 (even / odd k-tiles of a block), two 48 x 48
//                        accumulators (18 tiles) per wave, the k-parity halves summed through LDS at the end; per 128-frame block the dz rows and the x rows
//                        the eight taps need sit in LDS (double-buffered), both operands by transposing reads (ds_read_b64_tr_b16).
#include "a2f_common.h"
#include "gemm_mfma.h"
#include "img.h"

namespace a2f {
namespace pc {

using mm::u32x4_t;
using mm::u32x2_t;
using mm::bf16x8_t;
using mm::rsrc_t;
using mm::tr_read;
using f32x4v = __attribute__((ext_vector_type(4))) float;

constexpr int CG = 48;                  // channels per group
constexpr int TAPS = 128;
constexpr int KTOT = TAPS * CG;         // 6144
constexpr int NKT = KTOT / 32;          // 192 k-tiles
constexpr int PITCH = 208;              // bytes of one frame row in LDS: 96 hi + 96 lo + 16 pad
constexpr int TH = 512;
constexpr int MAXBT = 384;              // frames per workgroup (nt kernel)
constexpr int MAXM = 6;                 // 16-frame tiles per wave: 24 tiles over 4 wave rows
constexpr int XB_BYTES = ((MAXBT + TAPS - 1) * PITCH + 1023) / 1024 * 1024;
constexpr int WST = 48 * 128;           // one k-tile of W in LDS: [co][32 hi | 32 lo]
constexpr int NT_LDS = XB_BYTES + 4 * WST;
static_assert(NT_LDS <= 160 * 1024, "LDS");

__device__ __forceinline__ f32x4v mfma16(u32x4_t a, u32x4_t b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lds_dma16(rsrc_t rs, char* lds, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
#endif
}
// four fp32 -> 8 bytes of hi at p, 8 bytes of lo at p + 96
__device__ __forceinline__ void put4(char* p, const f32x4 v) {
  using img::bf16x2_t;
  using img::f32x2_t;
  const f32x2_t v0 = {v.x, v.y}, v1 = {v.z, v.w};
  const bf16x2_t h0 = __builtin_convertvector(v0, bf16x2_t), h1 = __builtin_convertvector(v1, bf16x2_t);
  const f32x2_t r0 = v0 - __builtin_convertvector(h0, f32x2_t), r1 = v1 - __builtin_convertvector(h1, f32x2_t);
  u32x2_t hi, lo;
  hi.x = __builtin_bit_cast(unsigned, h0);
  hi.y = __builtin_bit_cast(unsigned, h1);
  lo.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf16x2_t));
  lo.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  *reinterpret_cast<u32x2_t*>(p) = hi;
  *reinterpret_cast<u32x2_t*>(p + 96) = lo;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

struct NtArgs {
  const float* x;        // (S, T, C) fp32
  const char* wimg;      // split-32 image of (G * 48, 6144)
  const float* bias;     // (C) or null
  const float* aux;      // (S, T, C) added after the activation, or null
  float* y;              // (S, T, C)
  float* z;              // (S, T, C) pre-activation, or null
  int S, T, C, G, off, BT, nblk, act;
};

__global__ __launch_bounds__(TH, 2) void posconv_nt_kernel(const NtArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* xb = smem;
  char* wst = smem + XB_BYTES;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 3, wk = wave >> 2;
  // two groups per XCD (workgroup b runs on XCD b & 7): their weight images (2 x 1.2 MB) stay in that XCD's L2
  const int bid = blockIdx.x;
  const int g = (bid & 7) * 2 + ((bid >> 3) & 1);
  const int sb = bid >> 4;
  const int s = sb / a.nblk, blk = sb - s * a.nblk;
  const int t0 = blk * a.BT;
  const int mt = a.BT >> 4;
  const int mlo = wm * mt / 4, mcnt = (wm + 1) * mt / 4 - mlo;     // this wave row's 16-frame tiles

  // W k-tile pieces: a k-tile = 48 rows x 128 bytes = six 1-KB pieces (8 rows each); each k-parity group moves its own k-tiles: waves wm = 0, 1 two pieces, 2, 3 one
  const char* wbase = uniform_ptr(a.wimg + (size_t)g * CG * (KTOT * 4));
  const int npc = wm < 2 ? 2 : 1;
  unsigned voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wm + 4 * i;                                      // piece 0..5
    const int row = pc * 8 + (lane >> 3);
    voff[i] = (unsigned)row * (unsigned)(KTOT * 4) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
  }
  auto dma = [&](int kt, int st) {                                  // k-tile kt (of this wave's parity) -> stage st, sub-stage wk
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase + (size_t)kt * 128), 0, CG * KTOT * 4 - kt * 128, 0x00020000);
    char* base = wst + (st * 2 + wk) * WST;
    lds_dma16(rs, base + wm * 1024, voff[0]);
    if (npc == 2) lds_dma16(rs, base + (wm + 4) * 1024, voff[1]);
  };
  dma(wk, 0);

  // the block of input frames t0 + off .. t0 + off + BT + 126 (zero outside the sequence), fp32 -> hi / lo
  const int rx = a.BT + TAPS - 1;
  const float* xs = a.x + ((size_t)s * a.T) * a.C + g * CG;
  for (int e = t; e < rx * 12; e += TH) {
    const int r = e / 12, c4 = e - r * 12;
    const int tt = t0 + a.off + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tt >= 0 && tt < a.T) v = *reinterpret_cast<const f32x4*>(xs + (size_t)tt * a.C + 4 * c4);
    put4(xb + r * PITCH + c4 * 8, v);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int lr = lane & 15, q = lane >> 4;
  // B fragment (W): row co = 16 j + lr of the stage, chunk (q + 4 lo) ^ ((co >> 1) & 7); (16 j >> 1) & 7 = 0 for every j: the swizzle is that of lr
  const int bfrag = lr * 128 + ((q ^ ((lr >> 1) & 7)) << 4);
  const int arow = (mlo * 16 + lr) * PITCH;

  f32x4v acc[MAXM][3];
#pragma unroll
  for (int i = 0; i < MAXM; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  u32x4_t fa[2][MAXM], fb[2][3];

  auto reads = [&](int kt, int st) {
    const char* sb_ = wst + (st * 2 + wk) * WST;
#pragma unroll
    for (int j = 0; j < 3; ++j) {                                   // hi / lo chunk = bit 2 of the (swizzled) chunk index: an XOR of byte 64
      fb[0][j] = *reinterpret_cast<const u32x4_t*>(sb_ + bfrag + j * 2048);
      fb[1][j] = *reinterpret_cast<const u32x4_t*>(sb_ + (bfrag ^ 64) + j * 2048);
    }
    const int k0 = kt * 32 + q * 8;                                 // this lane group's eight k: one tap, eight channels
    const int tap = k0 / CG, ci = k0 - tap * CG;
    const char* ap = xb + arow + tap * PITCH + ci * 2;
#pragma unroll
    for (int i = 0; i < MAXM; ++i) {
      if (i < mcnt) {
        fa[0][i] = *reinterpret_cast<const u32x4_t*>(ap + i * 16 * PITCH);
        fa[1][i] = *reinterpret_cast<const u32x4_t*>(ap + i * 16 * PITCH + 96);
      }
    }
  };
  auto mfmas = [&]() {     // small terms first; D = (tile of W) x (tile of x)^T: lane = frame, registers = four consecutive output channels
#pragma unroll
    for (int i = 0; i < MAXM; ++i) {
      if (i < mcnt) {
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(fb[1][j], fa[0][i], acc[i][j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(fb[0][j], fa[1][i], acc[i][j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma16(fb[0][j], fa[0][i], acc[i][j]);
      }
    }
  };

  // ping-pong: the odd-k group runs one barrier behind the even-k group (its reads under the other's MFMAs).  Each group stages its own k-tiles: the DMA of
  // pair p + 1 goes out in the read slot of pair p into the stage this group last read in pair p - 1 (two barriers ago) and is waited for before the barrier
  // that ends the MFMA slot.
  constexpr int NP = NKT / 2;
  if (wk == 1) __builtin_amdgcn_s_barrier();
  for (int p = 0; p < NP; ++p) {
    const int st = p & 1;
    reads(2 * p + wk, st);
    if (p + 1 < NP) dma(2 * (p + 1) + wk, st ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mfmas();
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wk == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();

  // the odd-k group's sums travel through LDS ([wave row][register][lane]: the frame block is no longer needed)
  float* red = reinterpret_cast<float*>(xb) + wm * (MAXM * 12 * 64);
  if (wk == 1) {
#pragma unroll
    for (int i = 0; i < MAXM; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((i * 3 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (wk == 1) return;
  const size_t row0 = (size_t)s * a.T;
#pragma unroll
  for (int i = 0; i < MAXM; ++i) {
    if (i >= mcnt) continue;
    const int tt = t0 + (mlo + i) * 16 + lr;
    if (tt >= a.T) continue;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = g * CG + j * 16 + q * 4;
      f32x4 v;
      v.x = acc[i][j][0] + red[((i * 3 + j) * 4 + 0) * 64 + lane];
      v.y = acc[i][j][1] + red[((i * 3 + j) * 4 + 1) * 64 + lane];
      v.z = acc[i][j][2] + red[((i * 3 + j) * 4 + 2) * 64 + lane];
      v.w = acc[i][j][3] + red[((i * 3 + j) * 4 + 3) * 64 + lane];
      const size_t o = (row0 + tt) * a.C + c;
      if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + c);
      if (a.z) *reinterpret_cast<f32x4*>(a.z + o) = v;
      if (a.act) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      if (a.aux) v += *reinterpret_cast<const f32x4*>(a.aux + o);
      *reinterpret_cast<f32x4*>(a.y + o) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WG_BK = 128;                                    // frames per staged block: 4 k-tiles, two per k-parity wave group
constexpr int WG_TAPS = 8;                                    // taps per workgroup: four wave columns x two taps
constexpr int WG_DZ = WG_BK * PITCH;                          // 26 624 B
constexpr int WG_X = (WG_BK + WG_TAPS) * PITCH;               // 136 rows (135 used): 28 288 B
constexpr int WG_STAGE = WG_DZ + WG_X;
constexpr int WG_LDS = 2 * WG_STAGE;
static_assert(WG_LDS <= 160 * 1024 && WG_BK % 64 == 0, "LDS / k-tiles");
constexpr int WG_NDZ = WG_BK * 12, WG_NX = (WG_BK + WG_TAPS - 1) * 12;          // float4s per block
constexpr int WG_PER = (WG_NDZ + WG_NX + TH - 1) / TH;                           // 7 per thread
static_assert(4 * 72 * 64 * 4 <= WG_LDS, "the k-parity reduction reuses the stages");

struct WgArgs {
  const float* x;        // (S, T, C)
  const float* dz;       // (S, T, C)
  float* dw;             // (G, 128, 48, 48): [g][tap][ci][co]
  int S, T, C, G, pad, nb;
};

__global__ __launch_bounds__(TH, 2) void posconv_wgrad_kernel(const WgArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wp = wave & 3, wk = wave >> 2;                     // tap pair of the slice; k-tile parity inside a block
  const int bid = blockIdx.x;                                  // 16 groups x 16 tap slices; a group's slices on one XCD pair share its x / dz rows in L2
  const int g = (bid & 7) * 2 + ((bid >> 3) & 1);
  const int slice = bid >> 4;
  const int tap0 = slice * WG_TAPS;

  // this thread's float4s of a block: index e < WG_NDZ -> dz row e / 12; else x row (e - WG_NDZ) / 12
  f32x4 stage[WG_PER];
  auto fetch = [&](int it) {
    const int s = it / a.nb, b = it - s * a.nb;
    const int t0 = b * WG_BK;
    const size_t row0 = (size_t)s * a.T;
#pragma unroll
    for (int i = 0; i < WG_PER; ++i) {
      const int e = t + i * TH;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < WG_NDZ) {
        const int r = e / 12, c4 = e - r * 12;
        const int tt = t0 + r;
        if (tt < a.T) v = *reinterpret_cast<const f32x4*>(a.dz + (row0 + tt) * a.C + g * CG + 4 * c4);
      } else if (e < WG_NDZ + WG_NX) {
        const int e2 = e - WG_NDZ;
        const int r = e2 / 12, c4 = e2 - r * 12;
        const int tt = t0 + r + tap0 - a.pad;
        if (tt >= 0 && tt < a.T) v = *reinterpret_cast<const f32x4*>(a.x + (row0 + tt) * a.C + g * CG + 4 * c4);
      }
      stage[i] = v;
    }
  };
  auto park = [&](int st) {
    char* base = smem + st * WG_STAGE;
#pragma unroll
    for (int i = 0; i < WG_PER; ++i) {
      const int e = t + i * TH;
      if (e < WG_NDZ) {
        const int r = e / 12, c4 = e - r * 12;
        put4(base + r * PITCH + c4 * 8, stage[i]);
      } else if (e < WG_NDZ + WG_NX) {
        const int e2 = e - WG_NDZ;
        const int r = e2 / 12, c4 = e2 - r * 12;
        put4(base + WG_DZ + r * PITCH + c4 * 8, stage[i]);
      }
    }
  };

  // transposing fragment: lane (q = lane >> 4, i = lane & 15) -> k-rows 8 q + (i >> 2) and + 4, columns 4 (i & 3) .. + 3 of a 16-column tile
  const int q = lane >> 4, li = lane & 15;
  const int frag0 = (8 * q + (li >> 2)) * PITCH + 8 * (li & 3);
  f32x4v acc[2][3][3];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[tp][i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  auto frag = [](const char* p) {
    const u32x2_t r0 = tr_read(p), r1 = tr_read(p + 4 * PITCH);
    return u32x4_t{r0.x, r0.y, r1.x, r1.y};
  };

  const int nit = a.S * a.nb;
  fetch(0);
  park(0);
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const int st = it & 1;
    if (it + 1 < nit) fetch(it + 1);                          // the next block's rows are on their way while this one is multiplied
    const char* dzb = smem + st * WG_STAGE + frag0;
    const char* xbk = smem + st * WG_STAGE + WG_DZ + 2 * wp * PITCH + frag0;      // tap 2 wp (+ 1): x row of frame r is row r + tap
#pragma unroll
    for (int kk = 0; kk < WG_BK / 64; ++kk) {
      const int kt = 2 * kk + wk;
      u32x4_t fz[2][3];                                       // the dz fragments serve both taps of this wave
#pragma unroll
      for (int lo = 0; lo < 2; ++lo)
#pragma unroll
        for (int j = 0; j < 3; ++j) fz[lo][j] = frag(dzb + kt * 32 * PITCH + lo * 96 + j * 32);
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        u32x4_t fx[2][3];
#pragma unroll
        for (int lo = 0; lo < 2; ++lo)
#pragma unroll
          for (int i = 0; i < 3; ++i) fx[lo][i] = frag(xbk + (kt * 32 + tp) * PITCH + lo * 96 + i * 32);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[tp][i][j] = mfma16(fz[1][j], fx[0][i], acc[tp][i][j]);
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[tp][i][j] = mfma16(fz[0][j], fx[1][i], acc[tp][i][j]);
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[tp][i][j] = mfma16(fz[0][j], fx[0][i], acc[tp][i][j]);
        }
      }
    }
    if (it + 1 < nit) park(st ^ 1);                           // stage st ^ 1 was last read in iteration it - 1: the barrier below that iteration
    __syncthreads();
  }
  // the odd k-tiles' sums join the even ones through LDS ([wave column][register][lane]), then D = lane ci (li of tile i), registers co 4 q .. + 3 of tile j
  float* red = reinterpret_cast<float*>(smem) + wp * (72 * 64);
  if (wk == 1) {
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(((tp * 3 + i) * 3 + j) * 4 + r) * 64 + lane] = acc[tp][i][j][r];
  }
  __syncthreads();
  if (wk == 1) return;
#pragma unroll
  for (int tp = 0; tp < 2; ++tp) {
    float* out = a.dw + ((size_t)g * TAPS + tap0 + 2 * wp + tp) * (CG * CG);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 v;
        v.x = acc[tp][i][j][0] + red[(((tp * 3 + i) * 3 + j) * 4 + 0) * 64 + lane];
        v.y = acc[tp][i][j][1] + red[(((tp * 3 + i) * 3 + j) * 4 + 1) * 64 + lane];
        v.z = acc[tp][i][j][2] + red[(((tp * 3 + i) * 3 + j) * 4 + 2) * 64 + lane];
        v.w = acc[tp][i][j][3] + red[(((tp * 3 + i) * 3 + j) * 4 + 3) * 64 + lane];
        *reinterpret_cast<f32x4*>(out + (i * 16 + li) * CG + j * 16 + q * 4) = v;
      }
  }
}

}  // namespace pc
}  // namespace a2f

// include/a2f_hip.h: a2f_posconv_nt
extern "C" int a2f_posconv_nt(const float* x, const void* w_img, const float* bias, const float* aux, float* y, float* z, int64_t S, int64_t T, int64_t C,
                              int64_t groups, int64_t taps, int64_t off, int32_t act, a2f_stream_t stream) {
  using namespace a2f;
  using namespace a2f::pc;
  A2F_CHECK_ARG(x && w_img && y && S > 0 && T > 0, "a2f_posconv_nt: null / empty");
  A2F_CHECK_ARG(C == groups * CG && taps == TAPS && groups == 16, "a2f_posconv_nt: built for 16 groups of 48 channels and 128 taps (got C = %lld, groups = %lld, taps = %lld)",
                (long long)C, (long long)groups, (long long)taps);
  A2F_CHECK_ARG((((uintptr_t)x | (uintptr_t)w_img | (uintptr_t)y | (uintptr_t)z | (uintptr_t)bias | (uintptr_t)aux) & 15) == 0, "a2f_posconv_nt: 16-byte aligned buffers");
  A2F_CHECK_ARG(off <= 0 && off > -TAPS && S * T * C < (1ll << 40), "a2f_posconv_nt: window offset in (-128, 0]");
  // blocks per sequence: as few as fit the LDS (the 127-frame halo is re-staged per block), but at least enough workgroups for the chip when there are few
  // sequences (the reference's batch-1 step: 16 workgroups per block), down to 64-frame blocks
  int64_t nblk = cdiv(T, MAXBT);
  const int64_t want = cdiv(256, S * groups);
  if (want > nblk) nblk = want < cdiv(T, 64) ? want : cdiv(T, 64);
  if (nblk < cdiv(T, MAXBT)) nblk = cdiv(T, MAXBT);
  const int BT = (int)cdiv(cdiv(T, nblk), 16) * 16;
  nblk = cdiv(T, BT);                                         // rounding the block up to whole 16-frame tiles may have made the last block empty
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(posconv_nt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS) != hipSuccess)
      return set_err(A2F_ELAUNCH, "a2f_posconv_nt: cannot reserve %d bytes of LDS", NT_LDS);
    attr_done = true;
  }
  NtArgs a{x, (const char*)w_img, bias, aux, y, z, (int)S, (int)T, (int)C, (int)groups, (int)off, BT, (int)nblk, act};
  hipLaunchKernelGGL(posconv_nt_kernel, dim3((unsigned)(S * nblk * groups)), dim3(TH), NT_LDS, (hipStream_t)stream, a);
  A2F_CHECK_LAUNCH("a2f_posconv_nt");
  return A2F_OK;
}

// include/a2f_hip.h: a2f_posconv_wgrad
extern "C" int a2f_posconv_wgrad(const float* x, const float* dz, float* dw, int64_t S, int64_t T, int64_t C, int64_t groups, int64_t taps, int64_t pad,
                                 a2f_stream_t stream) {
  using namespace a2f;
  using namespace a2f::pc;
  A2F_CHECK_ARG(x && dz && dw && S > 0 && T > 0, "a2f_posconv_wgrad: null / empty");
  A2F_CHECK_ARG(C == groups * CG && taps == TAPS && groups == 16, "a2f_posconv_wgrad: built for 16 groups of 48 channels and 128 taps");
  A2F_CHECK_ARG((((uintptr_t)x | (uintptr_t)dz | (uintptr_t)dw) & 15) == 0 && pad >= 0 && pad < TAPS, "a2f_posconv_wgrad: 16-byte aligned buffers, 0 <= pad < 128");
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(posconv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS) != hipSuccess)
      return set_err(A2F_ELAUNCH, "a2f_posconv_wgrad: cannot reserve %d bytes of LDS", WG_LDS);
    attr_done = true;
  }
  WgArgs a{x, dz, dw, (int)S, (int)T, (int)C, (int)groups, (int)pad, (int)cdiv(T, WG_BK)};
  hipLaunchKernelGGL(posconv_wgrad_kernel, dim3((unsigned)(groups * (TAPS / WG_TAPS))), dim3(TH), WG_LDS, (hipStream_t)stream, a);
  A2F_CHECK_LAUNCH("a2f_posconv_wgrad");
  return A2F_OK;
}
