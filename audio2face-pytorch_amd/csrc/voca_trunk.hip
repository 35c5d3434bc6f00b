// Fused VOCA trunk forward (gfx950): one-hot tiling + concat -> 4 x (Conv(3,1)/s2 + ReLU) -> Linear 72->72 -> Linear
// 72->128 + tanh -> Linear 128->50, all for a block of 32 frames inside one workgroup
// (/root/reference/src/model/voca.py:40-48, everything except the 15069-wide head).
//
// The per-layer implicit-GEMM launches of this stack (8 + 7 kernels, 0.25 ms at 19 200 frames) are latency bound: every
// layer is a tiny tall-skinny product.  Here the activations of a frame block never leave the CU: they live in LDS in
// channels-last rows with ONE leading zero row per frame (the left zero padding of the k=3/s=2/p=1 convolution), so the
// im2col row of output position p is the contiguous window starting at row 2p - no masks, no im2col.  Each layer is a set
// of 32x32 f32-MFMA tiles (A fragment = window row, B fragment = weight row, both conflict-free ds_read_b32 thanks to odd
// pitches) distributed over the 4 waves.  Every activation is also written to global memory once
// (channels-last, exactly the buffers the backward kernels read) - 3.8 KB per frame.
//
// LDS (117 504 B): R0 = input windows [32][17][37] (later: conv2/conv3 outputs, concat, fc activations),
//                  R1 = conv1 output [32][9][32] (+1).
// The weights never touch LDS: each wave keeps the B fragments of every tile it is responsible for (288 floats per
// lane) in REGISTERS for the life of the persistent kernel - one wave per SIMD owns the full 512-entry register file -
// so a layer is nothing but A-fragment LDS reads feeding back-to-back MFMAs, with one barrier per layer.
#include "a2f_common.h"
#include <stdlib.h>

namespace a2f {

// diagnostic build only (-DVT_STAMPS): s_memtime totals per phase of wave 0 of workgroup 0, read by a2f_debug_trunk_stamps
#ifdef VT_STAMPS
__device__ unsigned long long vt_dbg[2][16];
#define VT_STAMP(k, i)                                                              \
  do {                                                                              \
    unsigned long long now__;                                                       \
    __builtin_amdgcn_sched_barrier(0);                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now__)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) vt_dbg[k][i] += now__ - last__;        \
    last__ = now__;                                                                 \
  } while (0)
#define VT_STAMP_INIT()                                                             \
  unsigned long long last__;                                                        \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last__)::"memory")
#else
#define VT_STAMP(k, i) do { } while (0)
#define VT_STAMP_INIT() do { } while (0)
#endif

// Global memory is touched only through raw buffer instructions whose resource covers exactly the B frames of a tensor:
// a frame >= B (tail block) is out of range, so loads return 0 and stores are dropped - no per-element predicate.  (With
// ordinary predicated loads hipcc wraps every load in a branch and waits for it at the join: 96 serialised latencies.)
using vt_rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ vt_rsrc_t vt_rsrc(const float* p, int frames, int floats_per_frame) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, frames * floats_per_frame * 4, 0x00020000);
}
__device__ __forceinline__ float vt_ld(vt_rsrc_t r, int float_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, float_off * 4, 0, 0));
}
__device__ __forceinline__ void vt_st(vt_rsrc_t r, int float_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, float_off * 4, 0, 0);
}
constexpr int VT_MAX_FRAMES = 800000;                      // keeps every byte offset below 2^31

constexpr int VT_THREADS = 256;
constexpr int VT_F = 32;                                   // frames per block
constexpr int VT_S0 = 17 * 37;                             // 629: frame stride of the input windows (odd)
constexpr int VT_S1 = 9 * 32 + 1;                          // 289
constexpr int VT_S2 = 5 * 32 + 1;                          // 161
constexpr int VT_S3 = 3 * 64 + 1;                          // 193
constexpr int VT_SC = 73, VT_SZ2 = 129;
constexpr int VT_R0 = VT_F * VT_S0;                        // 20128 floats
constexpr int VT_A2 = 0, VT_A3 = VT_A2 + VT_F * VT_S2, VT_CAT = VT_A3 + VT_F * VT_S3, VT_Z1 = VT_CAT + VT_F * VT_SC,
              VT_Z2 = VT_Z1 + VT_F * VT_SC;                // ends at exactly VT_R0
static_assert(VT_Z2 + VT_F * VT_SZ2 == VT_R0, "R0 carve");
constexpr int VT_R1 = VT_F * VT_S1;                        // 9248
constexpr size_t VT_LDS_BYTES = (size_t)(VT_R0 + VT_R1) * sizeof(float);

struct TrunkArgs {
  const float* x; const float* one_hot;
  const float* cw[4]; const float* cb[4];                  // time_conv.{0,2,4,6} weight (Co,Ci,3[,1]) / bias
  const float* fw[3]; const float* fb[3];                  // decoder.{0,1,3}
  float* a0; float* a1; float* a2; float* a3; float* cat; float* z1; float* z2; float* h3;
  int B, no, NB;
  int T;                                                   // 0: x holds (B,29,16) windows; > 0: x holds (B/T, T, 29) per-frame features
};

// B-operand fragments of one 32-column weight tile (contraction length K), kept in REGISTERS for the whole (persistent) kernel.
//   exact form (X3 = false, v_mfma_f32_32x32x2f32):  w[s] = W(k = 2s + h) of this lane's column, h = lane >> 5
//   split form (X3 = true, v_mfma_f32_32x32x16_bf16): k-step s covers k = 16s .. 16s + 15, lane half h owns k = 16s + 8h + j (j < 8);
//     every weight is x = hi + lo (bf16 each, round to nearest) and a product is lo*hi' + hi*lo' + hi*hi' with fp32 accumulation
//     (<= 2^-16 relative per product, the device of head_fused.hip / wgrad.hip / gemm_mfma.hip): 3 x 32 matrix-pipe cycles per 16 k
//     instead of 8 x 64 - the exact kernel spends two thirds of its time waiting on the fp32 matrix pipe (tools/trunk_stamps.py).
// Both forms hold K/2 registers per tile (K rounded up to 16 in the split form); weights beyond K are zero.
// WHERE the split form is used: the three Linear layers of the forward and the whole data-gradient chain.  The four convolutions of the
// forward stay exact: their pre-activations decide ReLU masks, and a pre-activation that rounds across zero switches a whole gradient path
// on or off - at 2^-16 per product that happens ~50x more often than at fp32 rounding (test_voca_vs_oracle_seeded at B = 1024: one flipped
// unit = 4e-3 on time_conv.0.weight.grad; the all-split forward ran 69.6 us instead of 90.4 and was withdrawn for that reason).
using vt_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using vt_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using vt_bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using vt_f32x2 = __attribute__((ext_vector_type(2))) float;

// One storage type for both forms (conv4 and decoder.3 share one register array across waves but not the form): K/2 dwords, K rounded up to 16.
template <int K> struct VtFrag { unsigned r[(K + 15) / 16 * 8]; };

__device__ __forceinline__ void vt_split2(float a, float b, unsigned& hi, unsigned& lo) {
  const vt_f32x2 v = {a, b};
  const vt_bf16x2 hb = __builtin_convertvector(v, vt_bf16x2);
  const vt_f32x2 r = v - __builtin_convertvector(hb, vt_f32x2);
  hi = __builtin_bit_cast(unsigned, hb);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, vt_bf16x2));
}

// wt(k) = the weight of this lane's column at contraction index k < K (the caller returns 0 for columns beyond the layer's width)
template <bool X3, int K, class F>
__device__ __forceinline__ void vt_load_frags(VtFrag<K>& fr, int h, F wt) {
  if constexpr (!X3) {
#pragma unroll
    for (int s = 0; s < (K + 1) / 2; ++s) {
      const int k = 2 * s + h;
      const float w = k < K ? wt(k) : 0.f;
      fr.r[s] = __builtin_bit_cast(unsigned, w);
    }
  } else {
#pragma unroll
    for (int s = 0; s < (K + 15) / 16; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {               // dwords 8s .. 8s+3: hi pairs, 8s+4 .. 8s+7: lo pairs
        const int k = 16 * s + 8 * h + 2 * j;
        vt_split2(k < K ? wt(k) : 0.f, k + 1 < K ? wt(k + 1) : 0.f, fr.r[8 * s + j], fr.r[8 * s + 4 + j]);
      }
  }
}

__device__ __forceinline__ f32x16 vt_mfma_bf16(vt_u32x4 a, vt_u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vt_bf16x8, a), __builtin_bit_cast(vt_bf16x8, b), c, 0, 0, 0);
}

// one 32-row x 32-col tile over the first KU <= K contraction indices: A fragment from LDS (ap = this lane's row, fp32, k contiguous),
// B fragments from registers
template <bool X3, int KU, int K>
__device__ __forceinline__ f32x16 vt_tile(const float* ap, int h, const VtFrag<K>& fr) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (!X3) {
    ap += h;
#pragma unroll
    for (int s = 0; s < (KU + 1) / 2; ++s) {
      const unsigned w = fr.r[s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], __builtin_bit_cast(float, w), acc, 0, 0, 0);
    }
  } else {
    ap += 8 * h;
#pragma unroll
    for (int s = 0; s < (KU + 15) / 16; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = ap[16 * s + j];
        // the rows are KU floats long: what lies behind them (the neighbouring frame, another layer's tile) meets a zero weight, but 0 x inf / NaN
        // of a neighbouring frame must not reach this one - select, only in the step that crosses KU
        if (16 * s + 8 + j >= KU) v[j] = (16 * s + 8 * h + j < KU) ? v[j] : 0.f;
      }
      unsigned hh[4], ll[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) vt_split2(v[2 * j], v[2 * j + 1], hh[j], ll[j]);
      const vt_u32x4 ah = {hh[0], hh[1], hh[2], hh[3]}, al = {ll[0], ll[1], ll[2], ll[3]};
      const vt_u32x4 bh = {fr.r[8 * s], fr.r[8 * s + 1], fr.r[8 * s + 2], fr.r[8 * s + 3]};
      const vt_u32x4 bl = {fr.r[8 * s + 4], fr.r[8 * s + 5], fr.r[8 * s + 6], fr.r[8 * s + 7]};
      acc = vt_mfma_bf16(al, bh, acc);
      acc = vt_mfma_bf16(ah, bl, acc);
      acc = vt_mfma_bf16(ah, bh, acc);
    }
  }
  return acc;
}

// SEQ: the input is the per-frame feature tensor (a.T = sequence length); a separate instantiation so the window-mode
// kernel keeps its register allocation; X3: the Linear layers on the split-bf16 matrix instructions (default) or the exact fp32 ones (A2F_TRUNK_EXACT=1)
template <bool SEQ, bool X3>
__global__ __launch_bounds__(VT_THREADS, 1) void voca_trunk_fwd_kernel(const TrunkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* R0 = lds;
  float* R1 = lds + VT_R0;
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int B = a.B;

  // ---- every weight this wave will ever need, as MFMA B fragments in registers (one wave per SIMD) ----
  // conv4 runs on waves 0,1 and decoder.3 on waves 2,3: their 64 fragments share one array (w47), 288 VGPRs in all
  VtFrag<111> w1;
  VtFrag<96> w2, w3;
  VtFrag<128> w47;
  VtFrag<72> w5, w6;
  // conv weights are in the reference layout w[n][ci][kh], contraction index k = kh * Ci + ci
  vt_load_frags<false>(w1, h, [&](int k) { return a.cw[0][(l31 * 37 + k % 37) * 3 + k / 37]; });
  vt_load_frags<false>(w2, h, [&](int k) { return a.cw[1][(l31 * 32 + k % 32) * 3 + k / 32]; });
  vt_load_frags<false>(w3, h, [&](int k) { return a.cw[2][(((wave & 1) * 32 + l31) * 32 + k % 32) * 3 + k / 32]; });
  if (wave < 2) vt_load_frags<false>(w47, h, [&](int k) { return a.cw[3][((wave * 32 + l31) * 64 + k % 64) * 3 + 1 + k / 64]; });   // taps 1,2: tap 0 always meets the padding
  else { const int n = (wave - 2) * 32 + l31; vt_load_frags<X3>(w47, h, [&](int k) { return n < 50 ? a.fw[2][n * 128 + k] : 0.f; }); }
  { const int n = (wave < 3 ? wave : 0) * 32 + l31; vt_load_frags<X3>(w5, h, [&](int k) { return n < 72 ? a.fw[0][n * 72 + k] : 0.f; }); }
  vt_load_frags<X3>(w6, h, [&](int k) { return a.fw[1][(wave * 32 + l31) * 72 + k]; });
  const float cb1 = a.cb[0][l31], cb2 = a.cb[1][l31], cb3 = a.cb[2][(wave & 1) * 32 + l31], cb4 = a.cb[3][(wave & 1) * 32 + l31];
  const int c5 = (wave < 3 ? wave : 0) * 32 + l31, c7 = (wave & 1) * 32 + l31;
  const float fb5 = c5 < 72 ? a.fb[0][c5] : 0.f, fb6 = a.fb[1][wave * 32 + l31], fb7 = c7 < 50 ? a.fb[2][c7] : 0.f;

  // zero rows of R1 (never overwritten)
  for (int e = t; e < VT_F * 33; e += VT_THREADS) { const int f = e / 33, c = e - f * 33; R1[f * VT_S1 + (c < 32 ? c : 288)] = 0.f; }

  // the x tile of a block (32 frames x 464 floats = 58 per lane) is fetched one block ahead into registers, so its HBM
  // latency hides under the previous block's layers
  constexpr int XR = VT_F * 464 / VT_THREADS;              // 58
  static_assert(XR * VT_THREADS == VT_F * 464, "x tile must divide evenly");
  float px[XR];
  float poh;                                               // one_hot[f0 + t/8][t%8] of the same block
  // sequence mode (a.T > 0): the input is the per-frame feature tensor (S, T, 29) and the 16-frame window
  // of frame F = s T + t is the CONTIGUOUS run of 464 floats starting at row F - 8, rows outside [0, T) of the frame's own
  // sequence reading as zeros (= feature_windows() / the zero-padded slices of vocaset.py) - 116 B per frame from HBM
  // instead of 1 856 B, and no window tensor to build on the host
  const vt_rsrc_t rx = vt_rsrc(a.x, B, SEQ ? 29 : 464), roh = vt_rsrc(a.one_hot, B, a.no);
  const vt_rsrc_t ra0 = vt_rsrc(a.a0, B, 592), ra1 = vt_rsrc(a.a1, B, 256), ra2 = vt_rsrc(a.a2, B, 128), ra3 = vt_rsrc(a.a3, B, 128);
  const vt_rsrc_t rcat = vt_rsrc(a.cat, B, 72), rz1 = vt_rsrc(a.z1, B, 72), rz2 = vt_rsrc(a.z2, B, 128), rh3 = vt_rsrc(a.h3, B, 50);
  auto fetch_x = [&](int blk) {
    if constexpr (!SEQ) {
      int voff = blk * VT_F * 464 + t;
      asm volatile("" : "+v"(voff));                       // keep ONE address register, the rest are immediates / adds
#pragma unroll
      for (int u = 0; u < XR; ++u) px[u] = vt_ld(rx, voff + u * VT_THREADS);
    } else {
      const int T = a.T, tf = (blk * VT_F) % T;            // frame index of the block's first frame inside its sequence
      int to = t;
      asm volatile("" : "+v"(to));
#pragma unroll
      for (int u = 0; u < XR; ++u) {
        const int e = to + u * VT_THREADS, f = e / 464, r = e - f * 464, w = r / 29;
        int tt = tf + f;                                   // T >= 32 = VT_F: a block crosses at most one sequence boundary
        tt = tt >= T ? tt - T : tt;
        const bool ok = (unsigned)(tt - 8 + w) < (unsigned)T;
        px[u] = vt_ld(rx, ok ? (blk * VT_F + f - 8) * 29 + r : -1);
      }
    }
    poh = vt_ld(roh, (blk * VT_F + (t >> 3)) * a.no + (t & 7));
  };
  if ((int)blockIdx.x < a.NB) fetch_x(blockIdx.x);
  VT_STAMP_INIT();

  for (int blk = blockIdx.x; blk < a.NB; blk += gridDim.x) {
    const int f0 = blk * VT_F;
    __syncthreads();                                       // previous block finished with R0
    VT_STAMP(0, 0);
    // ---- stage the input windows: R0[f][1+h][c] = x[f][c][h] (c<29) | one_hot[f][h%8] ; row 0 = zeros ----
    {
      int to = t;
      asm volatile("" : "+v"(to));                         // recompute the 58 scatter addresses per block instead of keeping them live
      if constexpr (!SEQ) {
#pragma unroll
        for (int u = 0; u < XR; ++u) {
          const int e = to + u * VT_THREADS, f = e / 464, r = e - f * 464, c = r >> 4, hh = r & 15;
          R0[f * VT_S0 + (1 + hh) * 37 + c] = px[u];
        }
      } else {                                             // feature rows are already channels-last: r = 29 w + c
#pragma unroll
        for (int u = 0; u < XR; ++u) {
          const int e = to + u * VT_THREADS, f = e / 464, r = e - f * 464, w = r / 29, c = r - 29 * w;
          R0[f * VT_S0 + (1 + w) * 37 + c] = px[u];
        }
      }
    }
    // one-hot channels (voca.py:40-45 repeat/view quirk): window row h carries one_hot[f][h % 8] in its 8 trailing
    // channels; lane (f = t/8, j = t%8) owns that value and fills rows j and j+8
    const float ohv = poh;
    {
      const int f = t >> 3, j = t & 7;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 8; ++q) R0[f * VT_S0 + (1 + j + 8 * b) * 37 + 29 + q] = ohv;
    }
    if (blk + (int)gridDim.x < a.NB) fetch_x(blk + gridDim.x);
    for (int e = t; e < VT_F * 37; e += VT_THREADS) { const int f = e / 37, c = e - f * 37; R0[f * VT_S0 + c] = 0.f; }
    __syncthreads();
    VT_STAMP(0, 1);
    // a0 to global (B,16,37): a frame is 592 contiguous floats in LDS and in HBM -> 148 float4 stores per frame
    {
      int to = t;
      asm volatile("" : "+v"(to));
#pragma unroll
      for (int i = 0; i < (VT_F * 148 + VT_THREADS - 1) / VT_THREADS; ++i) {
        const int e4 = to + i * VT_THREADS;
        if (e4 < VT_F * 148) {
          const int f = e4 / 148, q = e4 - f * 148;
          const float* src = R0 + f * VT_S0 + 37 + 4 * q;
          const f32x4 v = {src[0], src[1], src[2], src[3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ra0,
                                                 ((f0 + f) * 592 + 4 * q) * 4, 0, 0);
        }
      }
    }
    VT_STAMP(0, 9);
    // ---- conv1: rows (f,p<8) = 256 = 8 tiles, 2 per wave; window = R0[f*629 + 2p*37 ..+111) ----
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int mt = wave + 4 * it;
      const int row = mt * 32 + l31, f = row >> 3, p = row & 7;
      const f32x16 acc = vt_tile<false, 111>(R0 + f * VT_S0 + 2 * p * 37, h, w1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, of = orow >> 3, op = orow & 7;
        const float v = fmaxf(acc[r] + cb1, 0.f);
        R1[of * VT_S1 + (1 + op) * 32 + l31] = v;
        vt_st(ra1, ((f0 + of) * 8 + op) * 32 + l31, v);
      }
    }
    __syncthreads();                                       // R0 (input windows) is dead; conv1 output complete
    VT_STAMP(0, 2);
    for (int e = t; e < VT_F * 33; e += VT_THREADS) { const int f = e / 33, c = e - f * 33; R0[VT_A2 + f * VT_S2 + (c < 32 ? c : 160)] = 0.f; }
    for (int e = t; e < VT_F * 65; e += VT_THREADS) { const int f = e / 65, c = e - f * 65; R0[VT_A3 + f * VT_S3 + (c < 64 ? c : 192)] = 0.f; }
    // one-hot tail of the concat vector (voca.py:47): cat[f][64+q] = one_hot[f][q]
    {
      const int f = t >> 3, q = t & 7;
      R0[VT_CAT + f * VT_SC + 64 + q] = ohv;
      vt_st(rcat, (f0 + f) * 72 + 64 + q, ohv);
    }
    // ---- conv2: rows (f,p<4) = 128 = 4 tiles, 1 per wave (reads R1, writes the A2 area of R0: no hazard with the zeroing) ----
    {
      const int row = wave * 32 + l31, f = row >> 2, p = row & 3;
      const f32x16 acc = vt_tile<false, 96>(R1 + f * VT_S1 + 2 * p * 32, h, w2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, of = orow >> 2, op = orow & 3;
        const float v = fmaxf(acc[r] + cb2, 0.f);
        R0[VT_A2 + of * VT_S2 + (1 + op) * 32 + l31] = v;
        vt_st(ra2, ((f0 + of) * 4 + op) * 32 + l31, v);
      }
    }
    __syncthreads();
    VT_STAMP(0, 3);
    // ---- conv3: rows (f,p<2) = 64 = 2 M-tiles x 2 N-tiles ----
    {
      const int mt = wave >> 1, nt = wave & 1;
      const int row = mt * 32 + l31, f = row >> 1, p = row & 1;
      const f32x16 acc = vt_tile<false, 96>(R0 + VT_A2 + f * VT_S2 + 2 * p * 32, h, w3);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, of = orow >> 1, op = orow & 1;
        const float v = fmaxf(acc[r] + cb3, 0.f);
        R0[VT_A3 + of * VT_S3 + (1 + op) * 64 + nt * 32 + l31] = v;
        vt_st(ra3, ((f0 + of) * 2 + op) * 64 + nt * 32 + l31, v);
      }
    }
    __syncthreads();
    VT_STAMP(0, 4);
    // ---- conv4: rows f = 32 = 1 M-tile x 2 N-tiles (waves 0,1); window = positions 0,1 (taps 1,2) ----
    if (wave < 2) {
      const f32x16 acc = vt_tile<false, 128>(R0 + VT_A3 + l31 * VT_S3 + 64, h, w47);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = fmaxf(acc[r] + cb4, 0.f);
        R0[VT_CAT + of * VT_SC + wave * 32 + l31] = v;
        vt_st(rcat, (f0 + of) * 72 + wave * 32 + l31, v);
      }
    }
    __syncthreads();
    VT_STAMP(0, 5);
    // ---- decoder.0: 72 -> 72 (3 N-tiles, waves 0..2), no activation ----
    if (wave < 3) {
      const f32x16 acc = vt_tile<X3, 72>(R0 + VT_CAT + l31 * VT_SC, h, w5);
      if (c5 < 72) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = acc[r] + fb5;
          R0[VT_Z1 + of * VT_SC + c5] = v;
          vt_st(rz1, (f0 + of) * 72 + c5, v);
        }
      }
    }
    __syncthreads();
    VT_STAMP(0, 6);
    // ---- decoder.1: 72 -> 128 + tanh (4 N-tiles) ----
    {
      const f32x16 acc = vt_tile<X3, 72>(R0 + VT_Z1 + l31 * VT_SC, h, w6);
      const int col = wave * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = tanhf(acc[r] + fb6);
        R0[VT_Z2 + of * VT_SZ2 + col] = v;
        vt_st(rz2, (f0 + of) * 128 + col, v);
      }
    }
    __syncthreads();
    VT_STAMP(0, 7);
    // ---- decoder.3: 128 -> 50 (2 N-tiles, waves 2,3), output to global only ----
    if (wave >= 2) {
      const f32x16 acc = vt_tile<X3, 128>(R0 + VT_Z2 + l31 * VT_SZ2, h, w47);
      if (c7 < 50) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
          vt_st(rh3, (f0 + of) * 50 + c7, acc[r] + fb7);
        }
      }
    }
    VT_STAMP(0, 8);
  }
}

// =====================================================================================================================
// Fused data-gradient chain of the trunk (backward of voca.py:46-48 w.r.t. activations): from d(h3) down to the gradient
// of conv1's pre-activation, for a block of 32 frames per workgroup, same machinery as the forward kernel: gradients of a
// frame block stay in LDS, every weight is an MFMA B fragment in registers (269 VGPRs per lane), one barrier per layer.
// The k=3/s=2/p=1 data gradient in gather form: even input rows take tap 1 of one output row; odd input rows take tap 2 of
// output row p and tap 0 of output row p+1 - with ONE trailing zero row per frame in the LDS tile that is the contiguous
// window [row p | row p+1] times the stacked weights [tap 2; tap 0].  Activation-function derivatives (tanh', ReLU masks
// of the saved forward activations) are applied in the epilogues.  Outputs are the pre-activation gradients the weight
// gradient kernels (wgrad.hip) contract with the saved activations.
constexpr int VB_DH = 0;                                   // [32][51]
constexpr int VB_Z2 = VB_DH + VT_F * 51;                   // [32][129]
constexpr int VB_Z1 = VB_Z2 + VT_F * 129;                  // [32][73]
constexpr int VB_C4 = VB_Z1 + VT_F * 73;                   // [32][65]
constexpr int VB_C3 = VB_C4 + VT_F * 65;                   // [32][3][64] + 1  (rows 0,1 + trailing zero row)
constexpr int VB_C2 = VB_C3 + VT_F * 193;                  // [32][5][32] + 1  (rows 0..3 + trailing zero row)
constexpr int VB_END = VB_C2 + VT_F * 161;
constexpr size_t VB_LDS_BYTES = (size_t)VB_END * sizeof(float);

struct TrunkBwdArgs {
  const float* dh3; const float* z2; const float* cat; const float* a3; const float* a2; const float* a1;
  const float* fw3; const float* fw1; const float* fw0; const float* cw4; const float* cw3; const float* cw2;
  float* g_z2; float* g_z1; float* g_c4; float* g_c3; float* g_c2; float* g_c1;
  int B, NB;
};

template <bool X3>
__global__ __launch_bounds__(VT_THREADS, 1) void voca_trunk_bwd_data_kernel(const TrunkBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int B = a.B;

  // ---- weight fragments (B operand: lane -> output column n, lane>>5 -> contraction parity) ----
  VtFrag<50> q3;
  VtFrag<128> q1, q5;
  VtFrag<72> q0;
  VtFrag<64> q4, q6o;
  VtFrag<32> q6e;
  vt_load_frags<X3>(q3, h, [&](int k) { return a.fw3[k * 128 + wave * 32 + l31]; });                               // fc3': n < 128, k = j < 50: W3[j][n]
  { const int n = (wave < 3 ? wave : 0) * 32 + l31; vt_load_frags<X3>(q1, h, [&](int k) { return n < 72 ? a.fw1[k * 72 + n] : 0.f; }); }   // fc1': n < 72, k < 128: W1[k][n]
  vt_load_frags<X3>(q0, h, [&](int k) { return a.fw0[k * 72 + (wave & 1) * 32 + l31]; });                          // fc0': n < 64 (conv features only), k < 72: W0[k][n]
  { const int hp = wave >> 1, ci = (wave & 1) * 32 + l31; vt_load_frags<X3>(q4, h, [&](int k) { return a.cw4[(k * 64 + ci) * 3 + hp + 1]; }); }   // conv4': column (h', ci), k = co
  if (wave < 2) vt_load_frags<X3>(q5, h, [&](int k) { return k < 64 ? a.cw3[(k * 32 + l31) * 3 + 1] : 0.f; });      // conv3' even rows: tap 1, k = co < 64 (the tile runs KU = 64)
  else vt_load_frags<X3>(q5, h, [&](int k) { return k < 64 ? a.cw3[(k * 32 + l31) * 3 + 2] : a.cw3[((k - 64) * 32 + l31) * 3 + 0]; });   // odd rows: [tap 2 ; tap 0], k < 128
  vt_load_frags<X3>(q6e, h, [&](int k) { return a.cw2[(k * 32 + l31) * 3 + 1]; });                                 // conv2' even: tap 1, k = co < 32
  vt_load_frags<X3>(q6o, h, [&](int k) { return k < 32 ? a.cw2[(k * 32 + l31) * 3 + 2] : a.cw2[((k - 32) * 32 + l31) * 3 + 0]; });        // conv2' odd: [tap 2 ; tap 0], k < 64
  // trailing zero rows of the two convolution-gradient tiles (never overwritten)
  for (int e = t; e < VT_F * 65; e += VT_THREADS) { const int f = e / 65, c = e - f * 65; lds[VB_C3 + f * 193 + 128 + c] = 0.f; }
  for (int e = t; e < VT_F * 33; e += VT_THREADS) { const int f = e / 33, c = e - f * 33; lds[VB_C2 + f * 161 + 128 + c] = 0.f; }

  // d(h3) of a block (32 x 50 floats) is fetched one block ahead; the saved forward activations that gate the gradients
  // (tanh', ReLU masks: 96 floats per lane) are all requested at the top of the block, so one HBM latency is paid per
  // block instead of one per layer
  float pdh[7];
  const vt_rsrc_t rdh = vt_rsrc(a.dh3, B, 50), rz2 = vt_rsrc(a.z2, B, 128), rcat = vt_rsrc(a.cat, B, 72), ra3 = vt_rsrc(a.a3, B, 128),
                  ra2 = vt_rsrc(a.a2, B, 128), ra1 = vt_rsrc(a.a1, B, 256);
  const vt_rsrc_t rgz2 = vt_rsrc(a.g_z2, B, 128), rgz1 = vt_rsrc(a.g_z1, B, 72), rgc4 = vt_rsrc(a.g_c4, B, 64), rgc3 = vt_rsrc(a.g_c3, B, 128),
                  rgc2 = vt_rsrc(a.g_c2, B, 128), rgc1 = vt_rsrc(a.g_c1, B, 256);
  auto fetch_dh = [&](int blk) {
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int e = t + u * VT_THREADS;
      pdh[u] = vt_ld(rdh, e < VT_F * 50 ? blk * VT_F * 50 + e : -1);
    }
  };
  if ((int)blockIdx.x < a.NB) fetch_dh(blockIdx.x);
  VT_STAMP_INIT();

  for (int blk = blockIdx.x; blk < a.NB; blk += gridDim.x) {
    const int f0 = blk * VT_F;
    __syncthreads();
    VT_STAMP(1, 0);
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int e = t + u * VT_THREADS;
      if (e < VT_F * 50) { const int f = e / 50, j = e - f * 50; lds[VB_DH + f * 51 + j] = pdh[u]; }
    }
    // activation masks of this block, requested up front (accumulator row of register r: rho(r) + 4h).  One opaque base
    // offset per tensor keeps the 96 addresses out of registers (each load adds a constant).
    float mz[16], mc[16], m3[16], m2[16], m1[2][16];
    {
      int bz = (f0 + 4 * h) * 128 + wave * 32 + l31;
      int bc = wave < 2 ? (f0 + 4 * h) * 72 + wave * 32 + l31 : -0x20000000;   // waves 2,3: far out of range -> 0
      int b3 = ((f0 + 4 * h) * 2 + (wave >> 1)) * 64 + (wave & 1) * 32 + l31;
      asm volatile("" : "+v"(bz), "+v"(bc), "+v"(b3));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int r0 = (r & 3) + 8 * (r >> 2);
        mz[r] = vt_ld(rz2, bz + r0 * 128);
        mc[r] = vt_ld(rcat, bc + r0 * 72);
        m3[r] = vt_ld(ra3, b3 + r0 * 128);
      }
      // conv3' rows: orow = (wave&1)*32 + rho + 4h -> frame of = orow/2, input row hin = 2*(orow&1) + (wave>>1)
      // conv2' rows: orow = wave*32 + rho + 4h     -> frame of = orow/4, input row hin = 2*(orow&3) + odd
      // both are affine in orow: a2 offset = (f0*4 + 2*orow + (wave>>1))*32 + l31 ; a1 offset = (f0*8 + 2*orow + odd)*32 + l31
      int b2 = (f0 * 4 + 2 * ((wave & 1) * 32 + 4 * h) + (wave >> 1)) * 32 + l31;
      int b1 = (f0 * 8 + 2 * (wave * 32 + 4 * h)) * 32 + l31;
      asm volatile("" : "+v"(b2), "+v"(b1));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int r0 = (r & 3) + 8 * (r >> 2);
        m2[r] = vt_ld(ra2, b2 + 2 * r0 * 32);
        m1[0][r] = vt_ld(ra1, b1 + 2 * r0 * 32);
        m1[1][r] = vt_ld(ra1, b1 + 2 * r0 * 32 + 32);
      }
    }
    if (blk + (int)gridDim.x < a.NB) fetch_dh(blk + gridDim.x);
    __syncthreads();
    VT_STAMP(1, 1);
    // ---- decoder.3 backward: dz2pre = (dh3 W3) * (1 - z2^2) ----
    {
      const f32x16 acc = vt_tile<X3, 50>(lds + VB_DH + l31 * 51, h, q3);
      const int col = wave * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float z = mz[r];
        const float v = acc[r] * (1.f - z * z);
        lds[VB_Z2 + of * 129 + col] = v;
        vt_st(rgz2, (f0 + of) * 128 + col, v);
      }
    }
    __syncthreads();
    VT_STAMP(1, 2);
    // ---- decoder.1 backward: dz1 = dz2pre W1 ----
    if (wave < 3) {
      const f32x16 acc = vt_tile<X3, 128>(lds + VB_Z2 + l31 * 129, h, q1);
      const int col = wave * 32 + l31;
      if (col < 72) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
          lds[VB_Z1 + of * 73 + col] = acc[r];
          vt_st(rgz1, (f0 + of) * 72 + col, acc[r]);
        }
      }
    }
    __syncthreads();
    VT_STAMP(1, 3);
    // ---- decoder.0 backward (conv features only) + ReLU mask of conv4's output ----
    if (wave < 2) {
      const f32x16 acc = vt_tile<X3, 72>(lds + VB_Z1 + l31 * 73, h, q0);
      const int col = wave * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = mc[r] > 0.f ? acc[r] : 0.f;
        lds[VB_C4 + of * 65 + col] = v;
        vt_st(rgc4, (f0 + of) * 64 + col, v);
      }
    }
    __syncthreads();
    VT_STAMP(1, 4);
    // ---- conv4 data gradient: d3[f][h'][ci] = sum_co g4[f][co] w4[co][ci][h'+1], masked by a3 > 0 ----
    {
      const f32x16 acc = vt_tile<X3, 64>(lds + VB_C4 + l31 * 65, h, q4);
      const int hp = wave >> 1, ci = (wave & 1) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int of = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = m3[r] > 0.f ? acc[r] : 0.f;
        lds[VB_C3 + of * 193 + hp * 64 + ci] = v;
        vt_st(rgc3, ((f0 + of) * 2 + hp) * 64 + ci, v);
      }
    }
    __syncthreads();
    VT_STAMP(1, 5);
    // ---- conv3 data gradient: rows (f, p<2); waves 0,1: even input rows 2p (tap 1); waves 2,3: odd rows 2p+1 ----
    {
      const int mt = wave & 1, odd = wave >> 1;
      const int row = mt * 32 + l31, f = row >> 1, p = row & 1;
      const float* ap = lds + VB_C3 + f * 193 + p * 64;
      const f32x16 acc = odd ? vt_tile<X3, 128>(ap, h, q5) : vt_tile<X3, 64>(ap, h, q5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, of = orow >> 1, hin = 2 * (orow & 1) + odd;
        const float v = m2[r] > 0.f ? acc[r] : 0.f;
        lds[VB_C2 + of * 161 + hin * 32 + l31] = v;
        vt_st(rgc2, ((f0 + of) * 4 + hin) * 32 + l31, v);
      }
    }
    __syncthreads();
    VT_STAMP(1, 6);
    // ---- conv2 data gradient: rows (f, p<4) = 4 tiles, one per wave: even rows 2p, then odd rows 2p+1 ----
    {
      const int row = wave * 32 + l31, f = row >> 2, p = row & 3;
      const float* ap = lds + VB_C2 + f * 161 + p * 32;
#pragma unroll
      for (int odd = 0; odd < 2; ++odd) {
        const f32x16 acc = odd ? vt_tile<X3, 64>(ap, h, q6o) : vt_tile<X3, 32>(ap, h, q6e);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int orow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, of = orow >> 2, hin = 2 * (orow & 3) + odd;
          vt_st(rgc1, ((f0 + of) * 8 + hin) * 32 + l31, m1[odd][r] > 0.f ? acc[r] : 0.f);
        }
      }
    }
    VT_STAMP(1, 7);
  }
}

}  // namespace a2f

// A2F_TRUNK_EXACT=1: the exact fp32 matrix instructions instead of the split-bf16 ones (A-B hook and the reference point of the kernel
// tests; read per call, so a captured graph keeps the form it was captured with)
static bool trunk_exact() {
  const char* e = getenv("A2F_TRUNK_EXACT");
  return e && e[0] == '1';
}

static int voca_trunk_fwd_impl(const float* x, int64_t seq_len, const float* one_hot, int64_t B, int64_t n_onehot, const float* const* conv_w,
                               const float* const* conv_b, const float* const* fc_w, const float* const* fc_b, float* a0, float* a1,
                               float* a2, float* a3, float* cat72, float* z1, float* z2, float* h3, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(x && one_hot && conv_w && conv_b && fc_w && fc_b && a0 && a1 && a2 && a3 && cat72 && z1 && z2 && h3, "voca_trunk_fwd: null pointer");
  A2F_CHECK_ARG(B > 0 && B <= VT_MAX_FRAMES && n_onehot >= 8 && n_onehot <= 512, "voca_trunk_fwd: B=%lld (1..%d frames per call) n_onehot=%lld", (long long)B, VT_MAX_FRAMES, (long long)n_onehot);
  TrunkArgs a;
  a.x = x; a.one_hot = one_hot;
  for (int i = 0; i < 4; ++i) { A2F_CHECK_ARG(conv_w[i] && conv_b[i], "voca_trunk_fwd: null conv parameter"); a.cw[i] = conv_w[i]; a.cb[i] = conv_b[i]; }
  for (int i = 0; i < 3; ++i) { A2F_CHECK_ARG(fc_w[i] && fc_b[i], "voca_trunk_fwd: null decoder parameter"); a.fw[i] = fc_w[i]; a.fb[i] = fc_b[i]; }
  a.a0 = a0; a.a1 = a1; a.a2 = a2; a.a3 = a3; a.cat = cat72; a.z1 = z1; a.z2 = z2; a.h3 = h3;
  a.B = (int)B; a.no = (int)n_onehot; a.NB = (int)cdiv(B, VT_F); a.T = (int)seq_len;
  static bool attr_set = false;
  if (!attr_set) {
    const void* kernels[4] = {reinterpret_cast<const void*>(&voca_trunk_fwd_kernel<false, false>), reinterpret_cast<const void*>(&voca_trunk_fwd_kernel<false, true>),
                              reinterpret_cast<const void*>(&voca_trunk_fwd_kernel<true, false>), reinterpret_cast<const void*>(&voca_trunk_fwd_kernel<true, true>)};
    for (const void* k : kernels) {
      const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VT_LDS_BYTES);
      if (e != hipSuccess) return set_err(A2F_ELAUNCH, "voca_trunk_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    attr_set = true;
  }
  const int grid = a.NB < 256 ? a.NB : 256;
  const bool seq = a.T > 0;
  auto* kernel = trunk_exact() ? (seq ? &voca_trunk_fwd_kernel<true, false> : &voca_trunk_fwd_kernel<false, false>)
                               : (seq ? &voca_trunk_fwd_kernel<true, true> : &voca_trunk_fwd_kernel<false, true>);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(VT_THREADS), VT_LDS_BYTES, (hipStream_t)stream, a);
  A2F_CHECK_LAUNCH("voca_trunk_fwd");
  return A2F_OK;
}

extern "C" int a2f_voca_trunk_fwd(const float* x, const float* one_hot, int64_t B, int64_t n_onehot, const float* const* conv_w,
                                  const float* const* conv_b, const float* const* fc_w, const float* const* fc_b, float* a0, float* a1,
                                  float* a2, float* a3, float* cat72, float* z1, float* z2, float* h3, a2f_stream_t stream) {
  return voca_trunk_fwd_impl(x, 0, one_hot, B, n_onehot, conv_w, conv_b, fc_w, fc_b, a0, a1, a2, a3, cat72, z1, z2, h3, stream);
}

extern "C" int a2f_voca_trunk_fwd_seq(const float* feats, int64_t n_seq, int64_t seq_len, const float* one_hot, int64_t n_onehot,
                                      const float* const* conv_w, const float* const* conv_b, const float* const* fc_w,
                                      const float* const* fc_b, float* a0, float* a1, float* a2, float* a3, float* cat72, float* z1,
                                      float* z2, float* h3, a2f_stream_t stream) {
  A2F_CHECK_ARG(n_seq > 0 && seq_len >= 32 && n_seq * seq_len <= a2f::VT_MAX_FRAMES,
                "voca_trunk_fwd_seq: n_seq=%lld seq_len=%lld (sequences of >= 32 frames, <= %d frames per call)", (long long)n_seq, (long long)seq_len, a2f::VT_MAX_FRAMES);
  return voca_trunk_fwd_impl(feats, seq_len, one_hot, n_seq * seq_len, n_onehot, conv_w, conv_b, fc_w, fc_b, a0, a1, a2, a3, cat72, z1, z2, h3, stream);
}

extern "C" int a2f_voca_trunk_bwd_data(const float* dh3, const float* z2, const float* cat72, const float* a3, const float* a2,
                                       const float* a1, const float* fw3, const float* fw1, const float* fw0, const float* cw4,
                                       const float* cw3, const float* cw2, int64_t B, float* g_z2, float* g_z1, float* g_c4,
                                       float* g_c3, float* g_c2, float* g_c1, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(dh3 && z2 && cat72 && a3 && a2 && a1 && fw3 && fw1 && fw0 && cw4 && cw3 && cw2 && g_z2 && g_z1 && g_c4 && g_c3 && g_c2 && g_c1,
                "voca_trunk_bwd_data: null pointer");
  A2F_CHECK_ARG(B > 0 && B <= VT_MAX_FRAMES, "voca_trunk_bwd_data: B=%lld (1..%d frames per call)", (long long)B, VT_MAX_FRAMES);
  TrunkBwdArgs a;
  a.dh3 = dh3; a.z2 = z2; a.cat = cat72; a.a3 = a3; a.a2 = a2; a.a1 = a1;
  a.fw3 = fw3; a.fw1 = fw1; a.fw0 = fw0; a.cw4 = cw4; a.cw3 = cw3; a.cw2 = cw2;
  a.g_z2 = g_z2; a.g_z1 = g_z1; a.g_c4 = g_c4; a.g_c3 = g_c3; a.g_c2 = g_c2; a.g_c1 = g_c1;
  a.B = (int)B; a.NB = (int)cdiv(B, VT_F);
  static bool attr_set = false;
  if (!attr_set) {
    const void* kernels[2] = {reinterpret_cast<const void*>(&voca_trunk_bwd_data_kernel<false>), reinterpret_cast<const void*>(&voca_trunk_bwd_data_kernel<true>)};
    for (const void* k : kernels) {
      const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VB_LDS_BYTES);
      if (e != hipSuccess) return set_err(A2F_ELAUNCH, "voca_trunk_bwd_data: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    attr_set = true;
  }
  const int grid = a.NB < 256 ? a.NB : 256;
  auto* kernel = trunk_exact() ? &voca_trunk_bwd_data_kernel<false> : &voca_trunk_bwd_data_kernel<true>;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(VT_THREADS), VB_LDS_BYTES, (hipStream_t)stream, a);
  A2F_CHECK_LAUNCH("voca_trunk_bwd_data");
  return A2F_OK;
}

#ifdef VT_STAMPS
// diagnostic build only: copy and clear the phase totals (cycles of the 100 MHz s_memtime counter)
extern "C" int a2f_debug_trunk_stamps(unsigned long long* out32) {
  unsigned long long z[2][16] = {};
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(a2f::vt_dbg), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(a2f::vt_dbg), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
