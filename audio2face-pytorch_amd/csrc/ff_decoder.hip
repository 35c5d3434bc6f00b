// FaceFormer decoder recurrence, forward and reverse-time backward (gfx950).
//
// Reference: /root/reference/src/model/faceformer.py:154-185 re-runs the whole 1-layer decoder and the 64->15069 map on
// the growing prefix at every step (O(T^3)).  Causal masking makes row i of every pass equal to row i computed at
// step i, the diagonal enc-dec mask (faceformer.py:58-66) reduces cross-attention to a per-frame vector, and
// vertice_map(vertice_map_r(y)) collapses to A y + c, so the same function is evaluated here incrementally with a
// K/V cache: O(T^2) work, gradients identical (SURVEY §3.3 identities 1-5; tests/ pin both against the reference).
//
// Mapping: ONE workgroup per sequence (forward: 8 waves, the upper four only help with the attention sweep - two waves per head, alternate
// trips of keys, partial softmaxes merged through LDS: 5.9 -> 5.2 ms at T = 600, 74 -> 54 ms at T = 3000; backward, default = RECORD FORM:
// ff_decoder_bwd2_kernel, 8 waves, leaves per-frame gradient factors and the host entry turns them into the weight gradients with five a2f_mm
// products over all frames; from 256 frames on it runs in blocks of 128 steps and the attention terms that cross a block boundary are computed
// on the whole chip by ff_bwd_prep_kernel / ff_bwd_rect_kernel: 10.6 -> 5.5 ms at T = 600, 176 -> 30 ms at T = 3000.  ff_decoder_bwd_kernel is
// the round-1 form, kept selectable: 4 waves, 480-500 registers of weight-gradient accumulators, one wave per SIMD); sequences are independent,
// so a batch of S sequences fills S CUs.
// The step is latency-bound (a chain of 64-wide mat-vecs, LayerNorms and a softmax), so everything a step touches
// stays on chip:
//   * LDS: the self-attention / feedback weight matrices in the orientation that makes lane = output index
//     conflict-free (forward: transposed [in][out]; backward: row-major [out][in] for the transposed products);
//   * registers: the FFN weight slices (forward) / FFN weight columns + every weight-gradient accumulator (backward);
//   * K/V cache rows (256 B each) in global memory, L2-resident, read as float4 by 4 lanes per (key, head);
//   * wave64 butterfly reductions for softmax statistics and LayerNorm moments; one wave per attention head.
// Attention scores are never stored and every K/V row is read ONCE per step: the forward keeps a running maximum
// (online softmax) and merges the 16 key slots of a head at the end; the backward needs no separate pass for
// sum_j a_j da_j because that sum equals dctx . ctx (ctx is saved), so one sweep forms a, da, ds, dq, dK and dV.  LDS use
// does not grow with T (T = 3000 runs in the same kernel); several key slots per trip keep the L2 latency covered.
// ALiBi-style bias in closed form: -2^{-2(h+1)} * floor((i-j)/period) (faceformer.py:22-54).
#include "a2f_common.h"

namespace a2f {

constexpr int FD = 64;                 // model width
constexpr int FH = 4;                  // heads
constexpr int FHD = 16;                // head dim
constexpr int FFF = 128;               // feed-forward width
constexpr int FF_THREADS = 256;
constexpr int FF_FWD_THREADS = 512;   // forward: waves 4..7 only help with the self-attention sweep (two waves per head, every other trip of keys each)
constexpr int FF_BWD2_THREADS = 512;   // backward (record form): waves 4..7 help with the attention sweep
constexpr int FB2_U = 4;               // key slots per trip and wave in that form
// gradient record of one frame (floats): what the weight-gradient products need beside the saved activations
constexpr int GR_QKV = 0, GR_SA = 192, GR_HP = 256, GR_F = 384, GR_EMB = 448, GR_Y2 = 512, GR_Y = 576, GR_DC = 640, GR_RS = 704, FF_GREC = 720;
// blocked form of the backward (sequences of FF_BLK_MIN frames and more): the recurrence runs in blocks of FF_BLK steps, attention terms that cross a block
// boundary are batched over the whole chip (ff_bwd_prep_kernel / ff_bwd_rect_kernel below).  FF_NREC floats per frame: Nt[h][a'][b], then n[h][b]
constexpr int FF_BLK = 128, FF_BLK_MIN = FF_BLK + 1, FF_NREC = 4 * 272, NR_N = 1024;
static_assert(FF_BLK == 2 * 16 * FB2_U, "a block of the backward is one sweep trip of a head's two waves (ff_decoder_bwd2_kernel, LACC)");
#ifndef FA_U_SET
#define FA_U_SET 1
#endif
#ifndef FB_U_SET
#define FB_U_SET 8
#endif
constexpr int FA_U = FA_U_SET, FB_U = FB_U_SET;   // keys per lane and attention trip, forward (lane = key: 64 FA_U keys per wave and trip) / key slots of 16 keys, backward
// per-step save row (floats); K/V/dK/dV sub-rows are 256-byte aligned
constexpr int SV_X = 0, SV_Q = 64, SV_CTX = 128, SV_XH1 = 192, SV_XH2 = 256, SV_H = 320, SV_XH3 = 448, SV_ST = 512;
constexpr int SV_K = 576, SV_V = 640, SV_DK = 704, SV_DV = 768, SV_STRIDE = 896;
// stats sub-row: rstd1, rstd2, rstd3, -, m[4], l[4]
constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Sum / maximum over the 64 lanes of a wave on the DPP network (every lane gets the result; all lanes must be active): two quad permutes and two
// row mirrors leave each row of 16 with its total, four v_readlane join the rows in a fixed order.  The butterfly of six __shfl_xor (ds_bpermute: an
// LDS round trip each) that a2f_common.h's wave_sum compiles to is ~4x the latency, and these reductions sit on the recurrence's critical path
// (three LayerNorms and the softmax merge per step).  ff_dpp<0xB1> / <0x4E> alone are __shfl_xor(v, 1) / (v, 2) as register moves (the compiler
// lowers every __shfl_xor to ds_bpermute, constant mask or not).
template <int CTRL>
__device__ __forceinline__ float ff_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float ff_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float ffw_sum(float v) {
  v += ff_dpp<0xB1>(v);                    // quad_perm [1,0,3,2]
  v += ff_dpp<0x4E>(v);                    // quad_perm [2,3,0,1]
  v += ff_dpp<0x141>(v);                   // row_half_mirror
  v += ff_dpp<0x140>(v);                   // row_mirror
  return (ff_lane(v, 0) + ff_lane(v, 16)) + (ff_lane(v, 32) + ff_lane(v, 48));     // (two __shfl_xor instead: 4.3 vs 3.3 ms per 64 x 600 forward)
}
__device__ __forceinline__ float ffw_max(float v) {
  v = fmaxf(v, ff_dpp<0xB1>(v));
  v = fmaxf(v, ff_dpp<0x4E>(v));
  v = fmaxf(v, ff_dpp<0x141>(v));
  v = fmaxf(v, ff_dpp<0x140>(v));
  return fmaxf(fmaxf(ff_lane(v, 0), ff_lane(v, 16)), fmaxf(ff_lane(v, 32), ff_lane(v, 48)));
}

// y[col] = bias + sum_k Wt[k * stride + col] * x[k], k < 64, both operands in LDS (Wt transposed: lanes read consecutive columns; x broadcast as float4).
// All 80 LDS reads are issued before the first multiply-add (sched_barrier): left to itself the scheduler - register-shy in the kernels that do not
// fit a third wave per SIMD anyway - put every read directly in front of its use, one LDS round trip per k (s_memtime stamps of the train-mode
// forward: linear1 2 600 clk instead of 1 190).  Four partial sums by k mod 4.
__device__ __forceinline__ float ff_matvec64(const float* __restrict__ Wt, int stride, int col, const float* __restrict__ x, float bias) {
  float w[FD];
  float4 xv[FD / 4];
#pragma unroll
  for (int k = 0; k < FD; ++k) w[k] = Wt[k * stride + col];
#pragma unroll
  for (int q = 0; q < FD / 4; ++q) xv[q] = *reinterpret_cast<const float4*>(x + 4 * q);
  __builtin_amdgcn_sched_barrier(0);
  float a0 = bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int q = 0; q < FD / 4; ++q) {
    a0 += w[4 * q] * xv[q].x; a1 += w[4 * q + 1] * xv[q].y;
    a2 += w[4 * q + 2] * xv[q].z; a3 += w[4 * q + 3] * xv[q].w;
  }
  return (a0 + a1) + (a2 + a3);
}

// LayerNorm over the 64 lanes of one wave: returns normalised value, writes rstd
__device__ __forceinline__ float wave_layernorm(float u, float& rstd) {
  const float mean = ffw_sum(u) * (1.f / FD);
  const float d = u - mean;
  const float var = ffw_sum(d * d) * (1.f / FD);
  rstd = 1.f / sqrtf(var + LN_EPS);
  return d * rstd;
}
// LayerNorm backward over one wave: dxhat -> du
__device__ __forceinline__ float wave_layernorm_bwd(float dxhat, float xhat, float rstd) {
  const float m1 = ffw_sum(dxhat) * (1.f / FD);
  const float m2 = ffw_sum(dxhat * xhat) * (1.f / FD);
  return rstd * (dxhat - m1 - xhat * m2);
}

// Train-mode dropout (DROP variants).  The reference's decoder / PPE dropout (p = 0.1: faceformer.py:73,88; nn.TransformerDecoderLayer's
// dropout, dropout1..3 and the attention-probability dropout of its MultiheadAttention) redraws every mask at EVERY step of the O(T^3)
// loop, for all rows of the growing prefix - row j at step i then differs from row j at step j, and that function has no incremental
// evaluation.  Here a mask belongs to its ROW (frame), not to the loop step: the frame keeps the mask it drew when it was first computed.
// Per step the marginal distribution is the reference's; with masks fixed per row the reference's own loop is causal-consistent again and
// equals this recurrence, which is how the variant is pinned (tests/golden/gen_golden.py G14: the reference run with its dropout calls
// replaced by per-row masks, the same masks injected here).  Keep bits (csrc/dropout.hip layout): `row` (S*T, 10 words): columns 0..63 PPE
// output, 64..127 self-attention output (dropout1), 128..255 feed-forward activation, 256..319 feed-forward output (dropout3); `attn`
// (S*4*T, TW words): frame i, head h -> bit j = key j.  Cross-attention dropout and dropout2 act on the `cross` rows before this kernel.
struct FFDrop {
  const unsigned* row;
  const unsigned* attn;
  int TW;
  float scale;
};
constexpr int FF_ROW_WORDS = 10;

// =====================================================================================================================
// forward
// =====================================================================================================================
// -DFF_STAMPS: timing build only - thread 0 of sequence 0 leaves s_memtime stamps of step FF_STAMP_STEP after every barrier in the (then unused)
// A / c block of its gradient slab (tools/ff_stamps.py)
#ifdef FF_STAMPS
__device__ long long ffb_dbg[16];           // backward: stamps of step FF_STAMP_STEP of sequence 0 (a2f_debug_ff_bwd_stamps, tools/ff_stamps_bwd.py)
#define FFB_STAMP(n) do { if (t == 0 && s == 0 && i == FF_STAMP_STEP) ffb_dbg[n] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define FF_STAMP(n) do { if (t == 0 && s == 0 && i == FF_STAMP_STEP) stamp_out[n] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define FFB_STAMP(n) do { } while (0)
#define FF_STAMP(n) do { } while (0)
#endif
// amdgpu_waves_per_eu(2, 2): one workgroup of 8 waves per CU is all there ever is, so the scheduler may spend the 256 registers of a wave on
// latency.  Without the attribute it scheduled the train-mode variant for a third wave per SIMD it could not reach anyway (181 registers, limit
// 168): every LDS read of linear1 / the out-projection directly in front of its use, one at a time - those phases ran 2x longer than in the
// variant that happened to fit 168 (s_memtime stamps: 2 600 vs 1 190 and 3 460 vs 2 050 clk per step).
template <bool DROP>
__global__ __launch_bounds__(FF_FWD_THREADS, 1) __attribute__((amdgpu_waves_per_eu(2, 2))) void ff_decoder_fwd_kernel(const float* __restrict__ P, const float* __restrict__ pe,
                                                                        const float* __restrict__ cross, const float* __restrict__ style,
                                                                        int T, int period, float* __restrict__ y, float* __restrict__ save,
                                                                        float* __restrict__ kvp, const FFDrop dm) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __builtin_amdgcn_s_setprio(3);          // a latency-bound chain: when it shares a CU with GEMM waves of another stream (the pipelined step), its waves issue first
  float* WqkvT = lds;                       // [64][192]  WqkvT[k*192 + o] = Wqkv[o][k]
  float* WoT = WqkvT + FD * 192;            // [64][64]
  float* AT = WoT + FD * FD;                // [64][64]
  float* xs = AT + FD * FD;                 // [64]
  float* qs = xs + FD;                      // [64]  (pre-scaled by 1/sqrt(16))
  float* cs = qs + FD;                      // [64]  attention context
  float* y2s = cs + FD;                     // [64]
  float* hs = y2s + FD;                     // [128]
  float* part = hs + FFF;                   // [4][64]
  float* ys = part + 4 * FD;                // [64]
  float* mrg_c = ys + FD;                   // [2][64] unnormalised context partials of the two waves of each head
  float* mrg_s = mrg_c + 2 * FD;            // [2][4][2] their running maximum / exponential sum
  float* sweep_red = mrg_s + 2 * FH * 2;    // [8 waves][16][64] the waves' per-lane context partials on their way to one sum per component
  float* W1T = sweep_red + 8 * FHD * 64;    // [64][128]  W1T[k*128 + o] = W1[o][k]
  float* nx_pe = W1T + FD * FFF;            // [2][64]  positional-encoding row of step i, in buffer i & 1 (written during step i - 1 by the helper wave)
  float* nx_cross = nx_pe + 2 * FD;         // [2][64]  cross-attention row of step i
  unsigned* nx_kw = reinterpret_cast<unsigned*>(nx_cross + 2 * FD);   // [2][16] the frame's keep words (DROP; FF_ROW_WORDS = 10 used)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int head = wave & 3, apart = wave >> 2;     // self-attention: head and which half of the key trips this wave sweeps
  const int64_t s = blockIdx.x;
  float* svbase = save + s * (int64_t)T * SV_STRIDE;
  // The sweep's own copy of K and V (kvp; the records keep theirs for the backward): per (sequence, head) four planes of float4 per key,
  // [c][T][4] with head dimension d = 4c + e - lane = key, so one load instruction of a wave is 1 KB of consecutive keys (the records are 3.5 KB apart).
  float* kp_seq = kvp + s * (int64_t)T * (2 * FD);               // K planes of the four heads, then V planes
  float* vp_seq = kp_seq + (int64_t)T * FD;
#ifdef FF_STAMPS
  long long* stamp_out = reinterpret_cast<long long*>(save + (int64_t)gridDim.x * T * SV_STRIDE + A2F_FF_A);
#endif

  for (int e = t; e < 192 * FD; e += FF_FWD_THREADS) { const int o = e / FD, k = e - o * FD; WqkvT[k * 192 + o] = P[A2F_FF_WQKV + e]; }
  for (int e = t; e < FD * FD; e += FF_FWD_THREADS) {
    const int o = e / FD, k = e - o * FD;
    WoT[k * FD + o] = P[A2F_FF_WO + e];
    AT[k * FD + o] = P[A2F_FF_A + e];
  }
  // FFN weights: linear1 transposed in LDS (W1T[k][o]: its 64 registers per lane are what lets the sweep keep two keys per lane in flight);
  // linear2 slice in registers: output o = t & 63, inputs [32*(t>>6), +32)
  float w2r[32];
  for (int e = t; e < FFF * FD; e += FF_FWD_THREADS) { const int o = e / FD, k = e - o * FD; W1T[k * FFF + o] = P[A2F_FF_W1 + e]; }
#pragma unroll
  for (int k = 0; k < 32; ++k) w2r[k] = t < FF_THREADS ? P[A2F_FF_W2 + (t & 63) * FFF + (t >> 6) * 32 + k] : 0.f;
  const float bqkv = t < 192 ? P[A2F_FF_BQKV + t] : 0.f;
  const float b1 = t < FFF ? P[A2F_FF_B1 + t] : 0.f;
  // wave 0 lane constants
  const float bo = P[A2F_FF_BO + lane], b2 = P[A2F_FF_B2 + lane], cfb = P[A2F_FF_C + lane];
  const float g1 = P[A2F_FF_LN + 0 * FD + lane], be1 = P[A2F_FF_LN + 1 * FD + lane];
  const float g2 = P[A2F_FF_LN + 2 * FD + lane], be2 = P[A2F_FF_LN + 3 * FD + lane];
  const float g3 = P[A2F_FF_LN + 4 * FD + lane], be3 = P[A2F_FF_LN + 5 * FD + lane];
  const float sty = style[s * FD + lane];
  const float slope = exp2f(-2.f * (float)(head + 1));
  const float inv_period = 1.f / (float)period;
  const int js = lane >> 2, dq = lane & 3;                      // attention lanes: 16 key slots x 4 quarter-heads
  float emb = sty;                                              // wave 0: embedding fed to the next step
  float y2 = 0.f;
  // Inputs of a step that do not depend on the recurrence (positional-encoding row, cross-attention row, keep words) reach their consumers through
  // LDS: at the top of step i the HELPER wave (wave 7: it only takes part in the sweep) parks the rows of step i + 1, requested one step earlier,
  // in buffer (i + 1) & 1 and requests those of step i + 2 - ONE branch, its registers carried around the loop.  Why so roundabout: a vector-memory
  // load whose result is used (or whose destination register is re-used) somewhere in the chain makes the compiler place a vmcnt wait there for
  // EVERY wave, and for the waves on the critical path that wait is for their own global stores (the save rows): an L2 write acknowledgement per
  // phase (s_memtime stamps, train mode: linear1 2 736 clk with the keep word loaded by its own wave - or merely sharing a register with a
  // prefetch of the helper - against 1 184 without).
  constexpr int HELPER = FF_FWD_THREADS / 64 - 1;
  int ph_nx = T > 1 ? 1 % period : 0;                           // helper: row of the pe table for the step its registers hold
  float pf_pe = 0.f, pf_cross = 0.f;
  unsigned pf_kw = 0;
  if (wave == HELPER) {
    nx_pe[lane] = pe[lane];
    nx_cross[lane] = cross[(s * T) * FD + lane];
    if constexpr (DROP) { if (lane < FF_ROW_WORDS) nx_kw[lane] = dm.row[(s * T) * FF_ROW_WORDS + lane]; }
    const int i1 = min(1, T - 1);
    pf_pe = pe[ph_nx * FD + lane];
    pf_cross = cross[(s * T + i1) * FD + lane];
    if constexpr (DROP) pf_kw = dm.row[(s * T + i1) * FF_ROW_WORDS + min(lane, FF_ROW_WORDS - 1)];
  }
  __syncthreads();

  for (int i = 0; i < T; ++i) {
    FF_STAMP(0);
    float* sv = svbase + (int64_t)i * SV_STRIDE;
    const int buf = i & 1;
    float cross_i = 0.f;
    unsigned kw_d1 = 0, kw_d3 = 0, kw_act = 0;
    if (wave == HELPER) {
      nx_pe[(buf ^ 1) * FD + lane] = pf_pe;                     // rows of step i + 1 (buffer (i + 1) & 1 was last read during step i - 1)
      nx_cross[(buf ^ 1) * FD + lane] = pf_cross;
      if constexpr (DROP) { if (lane < FF_ROW_WORDS) nx_kw[(buf ^ 1) * 16 + lane] = pf_kw; }
      const int i2 = min(i + 2, T - 1);                         // rows of step i + 2 (past the end: the last step's, never consumed)
      if (++ph_nx == period) ph_nx = 0;
      pf_pe = pe[ph_nx * FD + lane];                            // the table has `period` rows: always in range
      pf_cross = cross[(s * T + i2) * FD + lane];
      if constexpr (DROP) pf_kw = dm.row[(s * T + i2) * FF_ROW_WORDS + min(lane, FF_ROW_WORDS - 1)];
    }
    if (wave == 0) {
      const float pe_i = nx_pe[buf * FD + lane];
      cross_i = nx_cross[buf * FD + lane];                      // used after the self-attention
      unsigned kw_x = 0;
      if constexpr (DROP) { kw_x = nx_kw[buf * 16 + (lane >> 5)]; kw_d1 = nx_kw[buf * 16 + 2 + (lane >> 5)]; kw_d3 = nx_kw[buf * 16 + 8 + (lane >> 5)]; }
      float x = emb + pe_i;                                     // periodic positional encoding
      if constexpr (DROP) x = ((kw_x >> (lane & 31)) & 1u) ? x * dm.scale : 0.f;
      xs[lane] = x;
      sv[SV_X + lane] = x;
    }
    if constexpr (DROP) { if (t < FFF) kw_act = nx_kw[buf * 16 + 4 + (t >> 5)]; }
    __syncthreads();
    FF_STAMP(1);
    if (t < 192) {                                              // q | k | v projections
      float acc = ff_matvec64(WqkvT, 192, t, xs, bqkv);
      if (t < 64) { acc *= 0.25f; qs[t] = acc; sv[SV_Q + t] = acc; }
      else {
        const int d = t & 63;                                    // head = d >> 4, plane c = (d >> 2) & 3, e = d & 3
        float* plane = (t < 128 ? kp_seq : vp_seq) + ((int64_t)(d >> 2) * T + i) * 4 + (d & 3);
        *plane = acc;
        sv[(t < 128 ? SV_K : SV_V) + d] = acc;
      }
    }
    __syncthreads();
    FF_STAMP(2);                                            // K/V row i visible to the whole workgroup
    {
      // ---- self-attention for `head` over keys j <= i: lane = key.  A wave takes FA_U x 64 consecutive keys per trip, the head's two waves alternate
      // trips; every lane runs its own online softmax (running maximum, sum, 16 context components) over its keys - no cross-lane traffic and one
      // exponential per key inside the loop (the (16 key slots x 4 quarter-heads) form computed every exponential four times and spent two shuffles per
      // key: ~190 vector instructions per 64 keys against ~65 here; s_memtime stamps at T = 3000: the sweep was 70-82 % of a step) - and the 64 partial
      // softmaxes of a wave are merged once per step.  A key past i re-reads key i (valid memory) and is masked out.
      float q[FHD];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 qv = *reinterpret_cast<const float4*>(qs + head * FHD + c * 4);
        q[4 * c] = qv.x; q[4 * c + 1] = qv.y; q[4 * c + 2] = qv.z; q[4 * c + 3] = qv.w;
      }
      const float* kph = kp_seq + (int64_t)head * 4 * T * 4;
      const float* vph = vp_seq + (int64_t)head * 4 * T * 4;
      float mrun = -INFINITY, lsum = 0.f;
      float cacc[FHD];
#pragma unroll
      for (int d = 0; d < FHD; ++d) cacc[d] = 0.f;
      const unsigned* abits = DROP ? dm.attn + ((s * FH + head) * T + i) * dm.TW : nullptr;     // this frame's, this head's keys
      for (int j0 = apart * 64 * FA_U; j0 <= i; j0 += 2 * 64 * FA_U) {
        float4 kk[FA_U][4], vv[FA_U][4];
        unsigned kw[FA_U];
#pragma unroll
        for (int u = 0; u < FA_U; ++u) {
          const int jj = min(j0 + 64 * u + lane, i);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            kk[u][c] = *reinterpret_cast<const float4*>(kph + ((int64_t)c * T + jj) * 4);
            vv[u][c] = *reinterpret_cast<const float4*>(vph + ((int64_t)c * T + jj) * 4);
          }
          if constexpr (DROP) kw[u] = abits[jj >> 5];
        }
        __builtin_amdgcn_sched_barrier(0);        // every load of the trip is issued HERE: the machine scheduler otherwise sinks the V loads below the scores (a second exposed round trip per trip)
        // (requesting the NEXT trip before evaluating this one - two register sets - spilled 59 / 93 registers as written; not kept)
        // scores of the trip first, ONE rescale of the running sums per trip (trip maximum), one exponential per key; the periodic bias
        // floor((i - j) / period) through a float reciprocal ((d + 0.5) / period is never within 0.5 / period of an integer: exact for T < 2^21)
        float pu[FA_U];
        float mt = mrun;
#pragma unroll
        for (int u = 0; u < FA_U; ++u) {
          const int j = j0 + 64 * u + lane;
          float pc[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) pc[c] = dot4(make_float4(q[4 * c], q[4 * c + 1], q[4 * c + 2], q[4 * c + 3]), kk[u][c]);
          float p = (pc[0] + pc[1]) + (pc[2] + pc[3]);
          p = j <= i ? p - slope * floorf(((float)(i - j) + 0.5f) * inv_period) : -INFINITY;
          pu[u] = p;
          mt = fmaxf(mt, p);
        }
        const float sc = mt == -INFINITY ? 1.f : expf(mrun - mt);
        lsum *= sc;
#pragma unroll
        for (int d = 0; d < FHD; ++d) cacc[d] *= sc;
#pragma unroll
        for (int u = 0; u < FA_U; ++u) {
          const int j = j0 + 64 * u + lane;
          const float e = pu[u] == -INFINITY ? 0.f : expf(pu[u] - mt);
          lsum += e;                                               // the normaliser keeps every key; only the value sum is thinned
          float ev = e;
          if constexpr (DROP) ev = ((kw[u] >> (j & 31)) & 1u) ? e * dm.scale : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            cacc[4 * c] += ev * vv[u][c].x; cacc[4 * c + 1] += ev * vv[u][c].y;
            cacc[4 * c + 2] += ev * vv[u][c].z; cacc[4 * c + 3] += ev * vv[u][c].w;
          }
        }
        mrun = mt;
      }
      // merge the wave's 64 partial softmaxes in a fixed order (deterministic).  The 16 context components go through a [component][lane] tile in LDS:
      // lane (d = lane >> 2, quarter = lane & 3) adds 16 lanes' worth of component d, two shuffles join the quarters (18 butterfly reductions of 6
      // shuffles each cost ~7 000 clk per step - more than the keys of a T = 600 sequence)
      const float m = ffw_max(mrun);
      const float scw = mrun == -INFINITY ? 0.f : expf(mrun - m);           // lanes that saw no key contribute nothing
      lsum = ffw_sum(lsum * scw);
      float* red = sweep_red + wave * (FHD * 64);
#pragma unroll
      for (int d = 0; d < FHD; ++d) red[d * 64 + lane] = cacc[d] * scw;
      // (same wave: LDS operations complete in order, no barrier)
      float tot = 0.f;
      {
        const float* src = red + (lane >> 2) * 64 + (lane & 3) * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 v4 = *reinterpret_cast<const float4*>(src + 4 * c);
          tot += (v4.x + v4.y) + (v4.z + v4.w);
        }
        tot += ff_dpp<0xB1>(tot);
        tot += ff_dpp<0x4E>(tot);
      }
      if ((lane & 3) == 0) mrg_c[apart * FD + head * FHD + (lane >> 2)] = tot;          // relative to this wave's maximum m
      if (lane == 0) {
        mrg_s[(apart * FH + head) * 2] = m;
        mrg_s[(apart * FH + head) * 2 + 1] = lsum;
      }
    }
    __syncthreads();
    FF_STAMP(3);
    if (wave == 0) {
      // merge the two partial softmaxes of every head (a wave that saw no key has m = -inf, l = 0): lane = context component, head = lane >> 4 - in the
      // wave that consumes the context, so the merge needs no phase and no barrier of its own (same wave: its LDS accesses complete in order)
      {
        const int hh = lane >> 4;
        const float m0 = mrg_s[hh * 2], l0 = mrg_s[hh * 2 + 1], m1 = mrg_s[(FH + hh) * 2], l1 = mrg_s[(FH + hh) * 2 + 1];
        const float mm = fmaxf(m0, m1);
        const float w0 = m0 == -INFINITY ? 0.f : expf(m0 - mm), w1 = m1 == -INFINITY ? 0.f : expf(m1 - mm);
        const float lt = l0 * w0 + l1 * w1;
        const float inv = 1.f / lt;
        const float c = (mrg_c[lane] * w0 + mrg_c[FD + lane] * w1) * inv;
        cs[lane] = c;
        sv[SV_CTX + lane] = c;
        if ((lane & 15) == 0) { sv[SV_ST + 4 + hh] = mm; sv[SV_ST + 8 + hh] = lt; }
      }
      FF_STAMP(4);
      float sa = ff_matvec64(WoT, FD, lane, cs, bo);
      if constexpr (DROP) sa = ((kw_d1 >> (lane & 31)) & 1u) ? sa * dm.scale : 0.f;     // dropout1
      float rstd;
      float xh = wave_layernorm(xs[lane] + sa, rstd);                       // norm1(x + self_attn)
      sv[SV_XH1 + lane] = xh;
      if (lane == 0) sv[SV_ST + 0] = rstd;
      const float y1 = xh * g1 + be1;
      xh = wave_layernorm(y1 + cross_i, rstd);                              // norm2(y1 + cross-attention row)
      sv[SV_XH2 + lane] = xh;
      if (lane == 0) sv[SV_ST + 1] = rstd;
      y2 = xh * g2 + be2;
      y2s[lane] = y2;
    }
    __syncthreads();
    FF_STAMP(5);
    if (t < FFF) {                                                          // linear1 + ReLU
      float acc = ff_matvec64(W1T, FFF, t, y2s, b1);
      acc = fmaxf(acc, 0.f);
      if constexpr (DROP) acc = ((kw_act >> (t & 31)) & 1u) ? acc * dm.scale : 0.f;          // dropout after the activation
      hs[t] = acc;
      sv[SV_H + t] = acc;
    }
    __syncthreads();
    FF_STAMP(6);
    if (wave < 4) {                                                         // linear2, 4 partial sums per output
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) acc += w2r[k] * hs[wave * 32 + k];
      part[wave * FD + lane] = acc;
    }
    __syncthreads();
    FF_STAMP(7);
    if (wave == 0) {
      float f = b2 + part[lane] + part[FD + lane] + part[2 * FD + lane] + part[3 * FD + lane];
      if constexpr (DROP) f = ((kw_d3 >> (lane & 31)) & 1u) ? f * dm.scale : 0.f;         // dropout3
      float rstd;
      const float xh = wave_layernorm(y2 + f, rstd);                        // norm3(y2 + ffn)
      sv[SV_XH3 + lane] = xh;
      if (lane == 0) sv[SV_ST + 2] = rstd;
      const float y3 = xh * g3 + be3;
      y[(s * T + i) * FD + lane] = y3;
      ys[lane] = y3;
    }
    __syncthreads();
    FF_STAMP(8);
    if (wave == 0) {                                                        // feedback: emb = A y + c + style
      emb = ff_matvec64(AT, FD, lane, ys, cfb + sty);
    }
  }
}

// =====================================================================================================================
// backward (reverse time)
// =====================================================================================================================
// LACC (launches that cover at most FF_BLK steps: every launch of the blocked form): the dK / dV accumulators of the launch's own keys live in LDS
// ([key - i_lo][64] floats each, the 16-byte chunk index XORed with 4 (key & 3) so that the four key slots a 16-lane group touches hit four bank
// quarters) instead of being read, added to and written back in global memory by every step - that round trip through L2 was 5.3 k of the 13.9 k clocks
// of a step (round-3 stamps).  What earlier-launched (later-time) blocks left in the global rows through ff_bwd_rect_kernel is complete before this
// launch; the helper wave parks it beside the step's saved rows and it is added when row i is finished.
template <bool DROP>
__global__ __launch_bounds__(FF_BWD2_THREADS, 1) void ff_decoder_bwd2_kernel(const float* __restrict__ P, const float* __restrict__ dy, int T,
                                                                              int period, float* __restrict__ save, float* __restrict__ grec,
                                                                              float* __restrict__ dcross, float* __restrict__ dstyle, const FFDrop dm,
                                                                              const float* __restrict__ nrec, int i_lo, int i_hi) {
  constexpr bool LACC = true;      // every launch covers at most FF_BLK steps (the global read-add-write form of rounds 2-3 was deleted in round 5)
  // steps i_hi-1 .. i_lo of the reverse-time recurrence (the whole sequence: i_lo = 0, i_hi = T).  Blocked form: the attention sweep covers the keys
  // j >= i_lo only; the keys of earlier blocks enter dq through the precomputed Nt / n (nrec) and receive their dK / dV from ff_bwd_rect_kernel; the
  // state carried from launch to launch (feedback carry, LayerNorm / style sums) lives in this sequence's slab.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __builtin_amdgcn_s_setprio(3);          // a latency-bound chain: when it shares a CU with GEMM waves of another stream (the pipelined step), its waves issue first
  float* Wqkv = lds;                        // [192][64] row-major: lane = input index for the transposed products
  float* Wo = Wqkv + 192 * FD;              // [64][64]
  float* Am = Wo + FD * FD;                 // [64][64]
  float* xs = Am + FD * FD;                 // [64]  x_i
  float* cs = xs + FD;                      // [64]  ctx_i
  float* hs = cs + FD;                      // [128] h_i (post-ReLU)
  float* y2s = hs + FFF;                    // [64]
  float* ys = y2s + FD;                     // [64]  y_i (decoder output row)
  float* dfs = ys + FD;                     // [64]
  float* dhps = dfs + FD;                   // [128]
  float* pbuf = dhps + FFF;                 // [4][64] / [2][128] partial sums
  float* dsas = pbuf + 4 * FD;              // [64]
  float* dcs = dsas + FD;                   // [64]
  float* dqkvs = dcs + FD;                  // [192]
  float* dembs = dqkvs + 192;               // [64] gradient w.r.t. emb_{i+1} from the previous (later-time) iteration
  float* qs = dembs + FD;                   // [64]
  float* mrg = qs + FD;                     // [64] dq partial of the heads' second waves
  // Saved rows of a step (the first 576 floats of its record: x, q, ctx, xh1, xh2, h, xh3, stats), its dy row and its keep words, parked here one
  // step ahead by the helper wave (see the loop top): [2][FB_STG]
  constexpr int FB_STG = 576 + FD + 16 + (LACC ? 2 * FD : 0);
  float* stg = mrg + FD;
  float* dkv = stg + 2 * FB_STG;            // LACC: [FF_BLK][64] dK, then [FF_BLK][64] dV

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int head = wave & 3, apart = wave >> 2;     // attention sweep: head and which half of the key trips
  const int64_t s = blockIdx.x;
  float* svbase = save + s * (int64_t)T * SV_STRIDE;

  for (int e = t; e < 192 * FD; e += FF_BWD2_THREADS) Wqkv[e] = P[A2F_FF_WQKV + e];
  for (int e = t; e < FD * FD; e += FF_BWD2_THREADS) { Wo[e] = P[A2F_FF_WO + e]; Am[e] = P[A2F_FF_A + e]; }
  if (t < FD) dembs[t] = 0.f;
  if constexpr (LACC)
    for (int e = t; e < 2 * FF_BLK * FD; e += FF_BWD2_THREADS) dkv[e] = 0.f;
  // zero this sequence's dK / dV accumulators (first launch of the sequence)
  if (i_hi == T)
    for (int64_t e = t; e < (int64_t)T * 128; e += FF_BWD2_THREADS) svbase[(e >> 7) * SV_STRIDE + SV_DK + (e & 127)] = 0.f;

  // weight columns for the transposed FFN products, in registers
  //   W1^T dhp : output k = t & 63, rows [32*(t>>6), +32)          -> w1c[r] = W1[32*(t>>6)+r][k]
  //   W2^T df  : output u = t & 127, rows [32*(t>>7), +32)         -> w2c[r] = W2[32*(t>>7)+r][u]
  float w1c[32], w2c[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    w1c[r] = t < FF_THREADS ? P[A2F_FF_W1 + ((t >> 6) * 32 + r) * FD + (t & 63)] : 0.f;
    w2c[r] = t < FF_THREADS ? P[A2F_FF_W2 + ((t >> 7) * 32 + r) * FFF + (t & 127)] : 0.f;
  }
  // The weight gradients are NOT accumulated here: every step leaves its factors (dqkv, dsa, dhp, df, demb, y2) in the gradient record
  // `grec` (FF_GREC floats per frame) and the host entry forms dW = sum_frames g (x) activation as a2f_mm products afterwards - that
  // frees ~230 registers per thread, which is what lets a second wave per head join the attention sweep.  LayerNorm gains / biases
  // and the style gradient are per-lane scalars and stay in registers.
  float gg1 = 0.f, gb1n = 0.f, gg2 = 0.f, gb2n = 0.f, gg3 = 0.f, gb3n = 0.f, gsty = 0.f;   // wave 0 lanes
  float* gbase = grec + s * (int64_t)T * FF_GREC;

  const float ln_g1 = P[A2F_FF_LN + 0 * FD + lane], ln_b1 = P[A2F_FF_LN + 1 * FD + lane];
  const float ln_g2 = P[A2F_FF_LN + 2 * FD + lane], ln_b2 = P[A2F_FF_LN + 3 * FD + lane];
  const float ln_g3 = P[A2F_FF_LN + 4 * FD + lane], ln_b3 = P[A2F_FF_LN + 5 * FD + lane];
  const float slope = exp2f(-2.f * (float)(head + 1));
  const float inv_period = 1.f / (float)period;
  const int js = lane >> 2, dq = lane & 3;
  float carry = 0.f;                                   // wave 0: (A^T demb_{i+1})[lane]
  float* dP = save + (int64_t)gridDim.x * T * SV_STRIDE + s * (int64_t)A2F_FF_PARAM_FLOATS;      // this sequence's slab
  if (i_hi < T && wave == 0) {                         // resume: state left by the launch of the later block
    gg1 = dP[A2F_FF_LN + 0 * FD + lane]; gb1n = dP[A2F_FF_LN + 1 * FD + lane];
    gg2 = dP[A2F_FF_LN + 2 * FD + lane]; gb2n = dP[A2F_FF_LN + 3 * FD + lane];
    gg3 = dP[A2F_FF_LN + 4 * FD + lane]; gb3n = dP[A2F_FF_LN + 5 * FD + lane];
    carry = dP[A2F_FF_A + lane];
    gsty = dP[A2F_FF_A + FD + lane];
  }
  __syncthreads();

  // The saved rows a step reads do not depend on the recurrence: the HELPER wave (wave 7: it only takes part in the sweep) requests those of step
  // i - 2 at the top of step i and parks those of step i - 1 (requested one step earlier) in buffer (i - 1) & 1 - one branch, registers carried
  // around the loop (cf. the forward).  Read where they are used they cost an L2 round trip at the head of S1, S5 and S7 each (s_memtime stamps:
  // S1 2 444 clk of a 13.9 k step).
  constexpr int HELPER = FF_BWD2_THREADS / 64 - 1;
  float pf[9], pf_dy = 0.f, pf_dk = 0.f, pf_dv = 0.f;
  unsigned pf_kw = 0;
  auto request_rows = [&](int step) {
    const float* r = svbase + (int64_t)step * SV_STRIDE;
#pragma unroll
    for (int q = 0; q < 9; ++q) pf[q] = r[q * 64 + lane];
    if constexpr (LACC) { pf_dk = r[SV_DK + lane]; pf_dv = r[SV_DV + lane]; }
    pf_dy = dy[(s * T + step) * FD + lane];
    if constexpr (DROP) pf_kw = dm.row[(s * T + step) * FF_ROW_WORDS + min(lane, FF_ROW_WORDS - 1)];
  };
  auto park_rows = [&](int step) {
    float* d = stg + (step & 1) * FB_STG;
#pragma unroll
    for (int q = 0; q < 9; ++q) d[q * 64 + lane] = pf[q];
    d[576 + lane] = pf_dy;
    if constexpr (DROP) { if (lane < FF_ROW_WORDS) reinterpret_cast<unsigned*>(d)[576 + FD + lane] = pf_kw; }
    if constexpr (LACC) { d[576 + FD + 16 + lane] = pf_dk; d[576 + FD + 16 + FD + lane] = pf_dv; }
  };
  if (wave == HELPER) {
    request_rows(i_hi - 1);
    park_rows(i_hi - 1);
    request_rows(max(i_hi - 2, 0));
  }
  __syncthreads();

  for (int i = i_hi - 1; i >= i_lo; --i) {
    FFB_STAMP(0);
    float* sv = svbase + (int64_t)i * SV_STRIDE;
    float* gr = gbase + (int64_t)i * FF_GREC;
    float dy2 = 0.f, xh2 = 0.f, xh1 = 0.f;
    if (wave == HELPER) {
      park_rows(i - 1);                                // i = 0: buffer 1, never read
      request_rows(max(i - 2, 0));
    }
    const float* sr = stg + (i & 1) * FB_STG;          // this step's rows (parked at the top of step i + 1, or before the loop)
    const unsigned* rowbits = reinterpret_cast<const unsigned*>(sr) + 576 + FD;
    // ---- S1: stage saved vectors; wave 0: LayerNorm-3 backward ----
    if (wave == 0) {
      const float xh3 = sr[SV_XH3 + lane];
      const float rstd3 = sr[SV_ST + 2];
      const float dy3 = sr[576 + lane] + carry;
      gg3 += dy3 * xh3;
      gb3n += dy3;
      const float du3 = wave_layernorm_bwd(dy3 * ln_g3, xh3, rstd3);
      dy2 = du3;                                       // residual branch of norm3(y2 + ffn)
      float df = du3;                                  // the feed-forward branch sees it through dropout3
      if constexpr (DROP) df = ((rowbits[8 + (lane >> 5)] >> (lane & 31)) & 1u) ? du3 * dm.scale : 0.f;
      dfs[lane] = df;
      gr[GR_F + lane] = df;
      const float yv = xh3 * ln_g3 + ln_b3;
      ys[lane] = yv;
      gr[GR_Y + lane] = yv;
      xh2 = sr[SV_XH2 + lane];
      const float y2v = xh2 * ln_g2 + ln_b2;
      y2s[lane] = y2v;
      gr[GR_Y2 + lane] = y2v;
    } else if (wave == 1) {
      hs[lane] = sr[SV_H + lane];
      hs[64 + lane] = sr[SV_H + 64 + lane];
    } else if (wave == 2) {
      xs[lane] = sr[SV_X + lane];
      cs[lane] = sr[SV_CTX + lane];
    } else if (wave == 3) {
      qs[lane] = sr[SV_Q + lane];
    }
    __syncthreads();
    FFB_STAMP(1);
    // ---- S2: dW2 += df (x) h ; dh = W2^T df (partials) ; dA += demb_{i+1} (x) y_i ; dc += demb_{i+1} ----
    if (t < FF_THREADS) {
      float acc = 0.f;
      const int half = t >> 7;
#pragma unroll
      for (int r = 0; r < 32; ++r) acc += w2c[r] * dfs[half * 32 + r];
      pbuf[half * FFF + (t & 127)] = acc;
    }
    __syncthreads();
    FFB_STAMP(2);
    // ---- S3: ReLU backward ; dW1 += dhp (x) y2 ----
    if (t < FFF) {
      const float dh = pbuf[t] + pbuf[FFF + t];
      // hs is the activation AFTER its dropout: a dropped unit is 0 there, a kept one relu * scale - so (hs > 0) is both masks at once
      const float dhp = hs[t] > 0.f ? (DROP ? dh * dm.scale : dh) : 0.f;
      dhps[t] = dhp;
      gr[GR_HP + t] = dhp;
    }
    __syncthreads();
    FFB_STAMP(3);
    // ---- S4: W1^T dhp (4 partial sums per output) ----
    {
      float acc = 0.f;
      if (wave < 4) {
#pragma unroll
        for (int r = 0; r < 32; ++r) acc += w1c[r] * dhps[wave * 32 + r];
      }
      __syncthreads();                                 // pbuf readers of S3 are done
      FFB_STAMP(4);
      if (wave < 4) pbuf[wave * FD + lane] = acc;
    }
    __syncthreads();
    FFB_STAMP(5);
    // ---- S5: wave 0: LayerNorm-2 and LayerNorm-1 backward ----
    float dx = 0.f;
    if (wave == 0) {
      dy2 += pbuf[lane] + pbuf[FD + lane] + pbuf[2 * FD + lane] + pbuf[3 * FD + lane];
      const float rstd2 = sr[SV_ST + 1], rstd1 = sr[SV_ST + 0];
      gg2 += dy2 * xh2;
      gb2n += dy2;
      const float du2 = wave_layernorm_bwd(dy2 * ln_g2, xh2, rstd2);
      dcross[(s * T + i) * FD + lane] = du2;           // u2 = y1 + cross_i
      xh1 = sr[SV_XH1 + lane];
      gg1 += du2 * xh1;
      gb1n += du2;
      const float du1 = wave_layernorm_bwd(du2 * ln_g1, xh1, rstd1);
      dx = du1;                                        // u1 = x + dropout1(self_attn(x))
      float dsa = du1;
      if constexpr (DROP) dsa = ((rowbits[2 + (lane >> 5)] >> (lane & 31)) & 1u) ? du1 * dm.scale : 0.f;
      dsas[lane] = dsa;
      gr[GR_SA + lane] = dsa;
    }
    __syncthreads();
    FFB_STAMP(6);
    // ---- S6: out-projection: dWo += dsa (x) ctx (wave 3) ; dctx = Wo^T dsa (wave 0) ----
    if (wave == 0) {
      float acc = 0.f;
#pragma unroll 16
      for (int r = 0; r < FD; ++r) acc += Wo[r * FD + lane] * dsas[r];
      dcs[lane] = acc;
      gr[GR_DC + lane] = acc;
    }
    __syncthreads();
    FFB_STAMP(7);
    // ---- S7: attention backward for head = wave ----
    {
      const float4 qv = *reinterpret_cast<const float4*>(qs + head * FHD + dq * 4);
      const float4 dcv = *reinterpret_cast<const float4*>(dcs + head * FHD + dq * 4);
      const float m = sr[SV_ST + 4 + head];
      const float invl = 1.f / sr[SV_ST + 8 + head];
      float* kbase = svbase + SV_K + head * FHD + dq * 4;
      float* vbase = svbase + SV_V + head * FHD + dq * 4;
      float* dkbase = svbase + SV_DK + head * FHD + dq * 4;
      float* dvbase = svbase + SV_DV + head * FHD + dq * 4;
      // sum_j a_j da_j = dctx . ctx (ctx = sum_j a_j v_j is saved), so ONE pass over the keys does everything: a, da, ds,
      // dq, dK += ds q, dV += a dctx.  FB_U key slots per trip, all loads issued before use (L2 latency bound);
      // a slot past i re-reads row i and neither accumulates nor stores.
      float rowsum = dot4(dcv, *reinterpret_cast<const float4*>(cs + head * FHD + dq * 4));
      rowsum += ff_dpp<0xB1>(rowsum);
      rowsum += ff_dpp<0x4E>(rowsum);
      float4 dqa = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0 && apart == 0) gr[GR_RS + head] = rowsum;
      if (i_lo > 0 && apart == 0) {                        // keys of the earlier blocks: dq += Nt^T dctx - rowsum n  (lane: a' = js, b = 4 dq ..)
        const float* nr = nrec + (s * T + i) * FF_NREC;
        const float4 nt = *reinterpret_cast<const float4*>(nr + head * 256 + js * 16 + dq * 4);
        const float dca = dcs[head * FHD + js];
        dqa = make_float4(nt.x * dca, nt.y * dca, nt.z * dca, nt.w * dca);
        if (js == 0) {
          const float4 nn = *reinterpret_cast<const float4*>(nr + NR_N + head * FHD + dq * 4);
          dqa.x -= rowsum * nn.x; dqa.y -= rowsum * nn.y; dqa.z -= rowsum * nn.z; dqa.w -= rowsum * nn.w;
        }
      }
      const unsigned* abits = DROP ? dm.attn + ((s * FH + head) * T + i) * dm.TW : nullptr;
      constexpr int FB_U = FB2_U;                          // two waves per head: half the slots per trip, alternate trips
      // LACC: a launch covers at most FF_BLK = 2 x 16 x FB_U keys, so the sweep is ONE trip (no loop-carried state for the compiler to keep)
      for (int j0 = i_lo + js + apart * 16 * FB_U; j0 <= i; j0 += LACC ? (1 << 30) : 2 * 16 * FB_U) {
        float4 kk[FB_U], vv[FB_U], dk[FB_U], dv[FB_U];
        unsigned kw[FB_U];
#pragma unroll
        for (int u = 0; u < FB_U; ++u) {
          const int jj = min(j0 + 16 * u, i);
          const int64_t off = (int64_t)jj * SV_STRIDE;
          kk[u] = *reinterpret_cast<const float4*>(kbase + off);
          vv[u] = *reinterpret_cast<const float4*>(vbase + off);
          if constexpr (LACC) {
            const int lk = jj - i_lo;
            const int a = lk * FD + (((head * 4 + dq) ^ ((lk & 3) << 2)) << 2);
            dk[u] = *reinterpret_cast<const float4*>(dkv + a);
            dv[u] = *reinterpret_cast<const float4*>(dkv + FF_BLK * FD + a);
          } else {
            dk[u] = *reinterpret_cast<const float4*>(dkbase + off);
            dv[u] = *reinterpret_cast<const float4*>(dvbase + off);
          }
          if constexpr (DROP) kw[u] = abits[jj >> 5];
        }
#pragma unroll
        for (int u = 0; u < FB_U; ++u) {
          const int j = j0 + 16 * u;
          const bool valid = j <= i;
          float p = dot4(qv, kk[u]);
          p += ff_dpp<0xB1>(p);
          p += ff_dpp<0x4E>(p);
          float da = dot4(dcv, vv[u]);
          da += ff_dpp<0xB1>(da);
          da += ff_dpp<0x4E>(da);
          const float a = valid ? expf(p - slope * floorf(((float)(i - min(j, i)) + 0.5f) * inv_period) - m) * invl : 0.f;   // exact floor for T < 2^21
          float ad = a;                                // ctx = sum_j a_j D_j v_j: dV sees a D, da_j = D_j (dctx . v_j); sum_j a_j da_j = dctx . ctx still
          if constexpr (DROP) {
            const float dj = ((kw[u] >> (min(j, i) & 31)) & 1u) ? dm.scale : 0.f;
            ad = a * dj;
            da *= dj;
          }
          const float ds = a * (da - rowsum);
          dqa.x += ds * kk[u].x; dqa.y += ds * kk[u].y; dqa.z += ds * kk[u].z; dqa.w += ds * kk[u].w;
          if (valid) {
            const float4 ndk = make_float4(dk[u].x + ds * qv.x, dk[u].y + ds * qv.y, dk[u].z + ds * qv.z, dk[u].w + ds * qv.w);
            const float4 ndv = make_float4(dv[u].x + ad * dcv.x, dv[u].y + ad * dcv.y, dv[u].z + ad * dcv.z, dv[u].w + ad * dcv.w);
            if constexpr (LACC) {
              const int lk = j - i_lo;
              const int a = lk * FD + (((head * 4 + dq) ^ ((lk & 3) << 2)) << 2);
              *reinterpret_cast<float4*>(dkv + a) = ndk;
              *reinterpret_cast<float4*>(dkv + FF_BLK * FD + a) = ndv;
            } else {
              const int64_t off = (int64_t)j * SV_STRIDE;
              *reinterpret_cast<float4*>(dkbase + off) = ndk;
              *reinterpret_cast<float4*>(dvbase + off) = ndv;
            }
          }
        }
      }
      // fold the 16 key slots into lanes 0..3 (js == 0: the only readers): row rotations by 4 and 8 pair those lanes exactly like the xor butterfly
      // did ((v[l] + v[l+4]) + (v[l+8] + v[l+12])), the rows are joined by two shuffles
      dqa.x += ff_dpp<0x124>(dqa.x); dqa.y += ff_dpp<0x124>(dqa.y); dqa.z += ff_dpp<0x124>(dqa.z); dqa.w += ff_dpp<0x124>(dqa.w);
      dqa.x += ff_dpp<0x128>(dqa.x); dqa.y += ff_dpp<0x128>(dqa.y); dqa.z += ff_dpp<0x128>(dqa.z); dqa.w += ff_dpp<0x128>(dqa.w);
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        dqa.x += __shfl_xor(dqa.x, off, 64); dqa.y += __shfl_xor(dqa.y, off, 64);
        dqa.z += __shfl_xor(dqa.z, off, 64); dqa.w += __shfl_xor(dqa.w, off, 64);
      }
      if (js == 0 && apart == 1) *reinterpret_cast<float4*>(mrg + head * FHD + dq * 4) = dqa;
      __syncthreads();                                 // dK / dV row i complete (includes this step's j = i term); second-wave dq partials visible
      FFB_STAMP(8);
      if (js == 0 && apart == 0) {                     // q was stored pre-scaled: d(q_raw) = 0.25 d(q_scaled)
        const float4 o = *reinterpret_cast<const float4*>(mrg + head * FHD + dq * 4);
        const float4 r = make_float4(0.25f * (dqa.x + o.x), 0.25f * (dqa.y + o.y), 0.25f * (dqa.z + o.z), 0.25f * (dqa.w + o.w));
        *reinterpret_cast<float4*>(dqkvs + head * FHD + dq * 4) = r;
        *reinterpret_cast<float4*>(gr + GR_QKV + head * FHD + dq * 4) = r;
      }
    }
    if (t >= 64 && t < 192) {                                                                               // SV_DV follows SV_DK: columns 64..127 are dV
      float g;
      if constexpr (LACC) {
        const int e = t - 64, lk = i - i_lo, col = e & 63;
        g = dkv[(e >> 6) * FF_BLK * FD + lk * FD + ((((col >> 2) ^ ((lk & 3) << 2)) << 2) | (col & 3))] + sr[576 + FD + 16 + e];
      } else {
        g = sv[SV_DK + (t - 64)];
      }
      dqkvs[t] = g;
      gr[GR_QKV + t] = g;
    }
    __syncthreads();
    FFB_STAMP(9);
    // ---- S9: dWqkv += dqkv (x) x ; dx += Wqkv^T dqkv (4 partial sums per output) ----
    if (wave < 4) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;        // four partial sums, the broadcast operand as float4 (as in the forward's projection)
#pragma unroll
      for (int r = 0; r < 48; r += 4) {
        const float4 gv = *reinterpret_cast<const float4*>(dqkvs + wave * 48 + r);
        a0 += Wqkv[(wave * 48 + r) * FD + lane] * gv.x; a1 += Wqkv[(wave * 48 + r + 1) * FD + lane] * gv.y;
        a2 += Wqkv[(wave * 48 + r + 2) * FD + lane] * gv.z; a3 += Wqkv[(wave * 48 + r + 3) * FD + lane] * gv.w;
      }
      pbuf[wave * FD + lane] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    FFB_STAMP(10);
    // ---- S10: wave 0: gradient w.r.t. emb_i; carry A^T demb to step i-1 ----
    if (wave == 0) {
      float demb = dx + pbuf[lane] + pbuf[FD + lane] + pbuf[2 * FD + lane] + pbuf[3 * FD + lane];
      if constexpr (DROP) demb = ((rowbits[lane >> 5] >> (lane & 31)) & 1u) ? demb * dm.scale : 0.f;      // through the PPE dropout
      gsty += demb;                                    // style enters every emb (emb_0 = style, emb_i = A y + c + style)
      dembs[lane] = demb;
      gr[GR_EMB + lane] = demb;
    }
    __syncthreads();
    FFB_STAMP(11);
    if (wave == 0) {
      float acc = 0.f;
#pragma unroll 16
      for (int r = 0; r < FD; ++r) acc += Am[r * FD + lane] * dembs[r];
      carry = acc;
    }
    // dembs stays valid for the dA / dc accumulation of the next (earlier-time) iteration
  }
  // demb_0 belongs to emb_0 = style only: it must not reach dA / dc (no feedback produced emb_0) - nothing to do.

  // ---- the per-lane scalars of this sequence: LayerNorm gains / biases into its slab (folded in sequence order), style gradient
  if (wave == 0) {
    dP[A2F_FF_A + lane] = carry;                       // (state for the launch of the earlier block; the A / c gradients themselves come from the products)
    dP[A2F_FF_A + FD + lane] = gsty;
    dP[A2F_FF_LN + 0 * FD + lane] = gg1;
    dP[A2F_FF_LN + 1 * FD + lane] = gb1n;
    dP[A2F_FF_LN + 2 * FD + lane] = gg2;
    dP[A2F_FF_LN + 3 * FD + lane] = gb2n;
    dP[A2F_FF_LN + 4 * FD + lane] = gg3;
    dP[A2F_FF_LN + 5 * FD + lane] = gb3n;
    if (i_lo == 0) dstyle[s * FD + lane] = gsty;
  }
}

// Blocked backward, attention terms across block boundaries (a_ij = exp(q_i . k_j - bias - m_i) / l_i is known from the forward):
//   ff_bwd_prep_kernel : for every step i of a block that starts at i0 > 0:  Nt_i[a'][b] = sum_{j < i0} a_ij D_ij v_j[a'] k_j[b],  n_i[b] = sum_{j < i0} a_ij k_j[b]
//                        so that the recurrence gets  sum_{j < i0} ds_ij k_j = Nt_i^T dctx_i - (dctx_i . ctx_i) n_i  without touching those keys;
//   ff_bwd_rect_kernel : after the recurrence of block [i0, i1) has left dctx_i and dctx_i . ctx_i in the records: for every key j < i0
//                        dK_j += sum_i ds_ij q_i,  dV_j += sum_i a_ij D_ij dctx_i  (one thread per (key, head): plain read-add-write, fixed order).
// Both run over the whole chip; what stays in the one-workgroup-per-sequence recurrence is the sweep over the <= FF_BLK keys of the own block.
template <bool DROP>
__global__ __launch_bounds__(256) void ff_bwd_prep_kernel(const float* __restrict__ save, float* __restrict__ nrec, int T, int period, const FFDrop dm) {
  const int quarter = threadIdx.x >> 6, lane = threadIdx.x & 63;       // wave = which four b of k_j[b]
  const int head = blockIdx.y;
  const int64_t s = blockIdx.z;
  const int i_base = FF_BLK + blockIdx.x * 64;                          // 64 consecutive steps: one block of the recurrence (FF_BLK is a multiple of 64)
  const int i0 = (i_base / FF_BLK) * FF_BLK;
  const int i = i_base + lane, ii = min(i, T - 1);
  const float* svbase = save + s * (int64_t)T * SV_STRIDE;
  const float* svi = svbase + (int64_t)ii * SV_STRIDE;
  float q[FHD];
#pragma unroll
  for (int c = 0; c < FHD; ++c) q[c] = svi[SV_Q + head * FHD + c];
  const float m = svi[SV_ST + 4 + head], invl = 1.f / svi[SV_ST + 8 + head];
  const float slope = exp2f(-2.f * (float)(head + 1));
  const unsigned* abits = DROP ? dm.attn + ((s * FH + head) * T + ii) * dm.TW : nullptr;
  float acc[FHD][4], nacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < FHD; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
  int qd = ii / period, r = ii - qd * period;                           // (i - j) / period and % period at j = 0
  unsigned word = 0;
  for (int j = 0; j < i0; ++j) {
    const float* svj = svbase + (int64_t)j * SV_STRIDE;                 // wave-uniform row
    float p = 0.f;
#pragma unroll
    for (int c = 0; c < FHD; ++c) p += q[c] * svj[SV_K + head * FHD + c];
    const float a = expf(p - slope * (float)qd - m) * invl;
    float ad = a;
    if constexpr (DROP) {
      if ((j & 31) == 0) word = abits[j >> 5];
      ad = ((word >> (j & 31)) & 1u) ? a * dm.scale : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float kb = svj[SV_K + head * FHD + quarter * 4 + c];
      nacc[c] += a * kb;
      const float w = ad * kb;
#pragma unroll
      for (int a2 = 0; a2 < FHD; ++a2) acc[a2][c] += w * svj[SV_V + head * FHD + a2];
    }
    if (--r < 0) { r = period - 1; --qd; }
  }
  if (i < T) {
    float* nr = nrec + (s * T + i) * FF_NREC;
#pragma unroll
    for (int a2 = 0; a2 < FHD; ++a2)
      *reinterpret_cast<float4*>(nr + head * 256 + a2 * 16 + quarter * 4) = make_float4(acc[a2][0], acc[a2][1], acc[a2][2], acc[a2][3]);
    *reinterpret_cast<float4*>(nr + NR_N + head * FHD + quarter * 4) = make_float4(nacc[0], nacc[1], nacc[2], nacc[3]);
  }
}

template <bool DROP>
__global__ __launch_bounds__(256) void ff_bwd_rect_kernel(float* __restrict__ save, const float* __restrict__ grec, int T, int period, int i_lo, int i_hi,
                                                          const FFDrop dm) {
  // the block's per-step vectors (q_i, dctx_i of this head, softmax statistics, dctx_i . ctx_i) staged in LDS once: read from global inside the
  // step loop, every iteration waited for an L2 round trip (223 us per launch at T = 3000 where the arithmetic needs 10)
  __shared__ __attribute__((aligned(16))) float qs[FF_BLK][FHD], ds_[FF_BLK][FHD];
  __shared__ float ms[FF_BLK], ils[FF_BLK], rss[FF_BLK];
  const int t = threadIdx.x;
  const int j = blockIdx.x * 256 + t;
  const int head = blockIdx.y;
  const int64_t s = blockIdx.z;
  float* svbase = save + s * (int64_t)T * SV_STRIDE;
  const float* gbase = grec + s * (int64_t)T * FF_GREC;
  const int ni = i_hi - i_lo;                                           // <= FF_BLK
  for (int e = t; e < ni * FHD; e += 256) {
    const int ii = e >> 4, c = e & 15;
    qs[ii][c] = svbase[(int64_t)(i_lo + ii) * SV_STRIDE + SV_Q + head * FHD + c];
    ds_[ii][c] = gbase[(int64_t)(i_lo + ii) * FF_GREC + GR_DC + head * FHD + c];
  }
  if (t < ni) {
    const float* svi = svbase + (int64_t)(i_lo + t) * SV_STRIDE;
    ms[t] = svi[SV_ST + 4 + head];
    ils[t] = 1.f / svi[SV_ST + 8 + head];
    rss[t] = gbase[(int64_t)(i_lo + t) * FF_GREC + GR_RS + head];
  }
  __syncthreads();
  if (j >= i_lo) return;
  float* svj = svbase + (int64_t)j * SV_STRIDE;
  float k[FHD], v[FHD], dk[FHD], dv[FHD];
#pragma unroll
  for (int c = 0; c < FHD; ++c) { k[c] = svj[SV_K + head * FHD + c]; v[c] = svj[SV_V + head * FHD + c]; dk[c] = 0.f; dv[c] = 0.f; }
  const float slope = exp2f(-2.f * (float)(head + 1));
  int qd = (i_lo - j) / period, r = (i_lo - j) - qd * period;
  const unsigned* abits = DROP ? dm.attn + ((s * FH + head) * T + i_lo) * dm.TW + (j >> 5) : nullptr;
  for (int ii = 0; ii < ni; ++ii) {
    float q[FHD], dc[FHD];
#pragma unroll
    for (int c = 0; c < FHD; c += 4) {
      const float4 a4 = *reinterpret_cast<const float4*>(&qs[ii][c]);
      const float4 b4 = *reinterpret_cast<const float4*>(&ds_[ii][c]);
      q[c] = a4.x; q[c + 1] = a4.y; q[c + 2] = a4.z; q[c + 3] = a4.w;
      dc[c] = b4.x; dc[c + 1] = b4.y; dc[c + 2] = b4.z; dc[c + 3] = b4.w;
    }
    float p = 0.f, da = 0.f;
#pragma unroll
    for (int c = 0; c < FHD; ++c) { p += q[c] * k[c]; da += dc[c] * v[c]; }
    const float a = expf(p - slope * (float)qd - ms[ii]) * ils[ii];
    float ad = a;
    if constexpr (DROP) {
      const float dj = ((abits[(int64_t)ii * dm.TW] >> (j & 31)) & 1u) ? dm.scale : 0.f;
      ad = a * dj;
      da *= dj;
    }
    const float ds = a * (da - rss[ii]);
#pragma unroll
    for (int c = 0; c < FHD; ++c) { dk[c] += ds * q[c]; dv[c] += ad * dc[c]; }
    if (++r == period) { r = 0; ++qd; }
  }
#pragma unroll
  for (int c = 0; c < FHD; ++c) { svj[SV_DK + head * FHD + c] += dk[c]; svj[SV_DV + head * FHD + c] += dv[c]; }
}

// dparams[i] = sum over sequences (fixed order) of the per-sequence slabs the backward kernel left behind `save`
constexpr size_t FF_FWD_LDS = (size_t)(FD * 192 + 2 * FD * FD + 4 * FD + FFF + 4 * FD + FD + 2 * FD + 2 * FH * 2 + 8 * FHD * 64 + FD * FFF + 4 * FD + 32) * sizeof(float);
// dparams[off + i] = sum over sequences (fixed order) of the per-sequence slabs the backward kernel left behind `save`
__global__ __launch_bounds__(256) void ff_decoder_fold_range_kernel(const float* __restrict__ slabs, int S, float* __restrict__ dP, int off, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += slabs[(int64_t)s * A2F_FF_PARAM_FLOATS + off + i];
  dP[off + i] = acc;
}

constexpr int FF_GEMM_SPLIT_MAX = 64;
constexpr size_t FF_GEMM_WS_FLOATS = (size_t)FF_GEMM_SPLIT_MAX * (192 * 64 + 192);

constexpr size_t FF_BWD_LDS = (size_t)(192 * FD + 2 * FD * FD + 2 * FD + FFF + 3 * FD + FFF + 4 * FD + 2 * FD + 192 + 2 * FD) * sizeof(float);

}  // namespace a2f

// floats in front of the forward sweep's K / V planes (the last region of the save buffer): per-step save rows + one gradient slab per sequence
// (written by the backward kernel, folded in sequence order) + the backward's per-frame gradient records and the split-K workspace of its
// weight-gradient products
static size_t ff_kv_planes_offset(int64_t S, int64_t T) {
  return (size_t)S * (size_t)T * a2f::SV_STRIDE + (size_t)S * A2F_FF_PARAM_FLOATS + (size_t)S * (size_t)T * a2f::FF_GREC + a2f::FF_GEMM_WS_FLOATS +
         (T >= a2f::FF_BLK_MIN ? (size_t)S * (size_t)T * a2f::FF_NREC : 0);
}

extern "C" size_t a2f_ff_save_bytes(int64_t S, int64_t T) {
  if (S <= 0 || T <= 0) return 16;
  return (ff_kv_planes_offset(S, T) + (size_t)S * (size_t)T * 2 * a2f::FD) * sizeof(float);
}

static int ff_check_drop(const uint32_t* row_bits, const uint32_t* attn_bits, float keep_scale, const char* who) {
  using namespace a2f;
  A2F_CHECK_ARG(row_bits && attn_bits && keep_scale >= 1.f, "%s: null keep bits / keep_scale %g", who, (double)keep_scale);
  return A2F_OK;
}

static int ff_fwd_impl(const float* params, const float* pe, const float* cross, const float* style, int64_t S, int64_t T, int32_t period, float* y,
                       void* save, size_t save_bytes, const uint32_t* row_bits, const uint32_t* attn_bits, float keep_scale, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(params && pe && cross && style && y && save, "ff_decoder_fwd: null pointer");
  A2F_CHECK_ARG(S > 0 && S < (1 << 20) && T > 0 && T < (1 << 20) && period > 0, "ff_decoder_fwd: S=%lld T=%lld period=%d", (long long)S, (long long)T, period);
  A2F_CHECK_ARG((((uintptr_t)save) & 255) == 0, "ff_decoder_fwd: save buffer must be 256-byte aligned");
  if (save_bytes < a2f_ff_save_bytes(S, T)) return set_err(A2F_EWORKSPACE, "ff_decoder_fwd: save buffer %zu < %zu", save_bytes, a2f_ff_save_bytes(S, T));
  static bool attr_set[2] = {false, false};
  const bool drop = row_bits != nullptr;
  if (!attr_set[drop]) {
    const void* k = drop ? reinterpret_cast<const void*>(&ff_decoder_fwd_kernel<true>) : reinterpret_cast<const void*>(&ff_decoder_fwd_kernel<false>);
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FF_FWD_LDS);
    if (e != hipSuccess) return set_err(A2F_ELAUNCH, "ff_decoder_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set[drop] = true;
  }
  const FFDrop dm{row_bits, attn_bits, (int)a2f_attn_keep_words(T), keep_scale};
  if (drop)
    hipLaunchKernelGGL(ff_decoder_fwd_kernel<true>, dim3((unsigned)S), dim3(FF_FWD_THREADS), FF_FWD_LDS, (hipStream_t)stream, params, pe, cross, style, (int)T,
                       (int)period, y, (float*)save, (float*)save + ff_kv_planes_offset(S, T), dm);
  else
    hipLaunchKernelGGL(ff_decoder_fwd_kernel<false>, dim3((unsigned)S), dim3(FF_FWD_THREADS), FF_FWD_LDS, (hipStream_t)stream, params, pe, cross, style, (int)T,
                       (int)period, y, (float*)save, (float*)save + ff_kv_planes_offset(S, T), dm);
  A2F_CHECK_LAUNCH("ff_decoder_fwd");
  return A2F_OK;
}

// Record form of the backward: the recurrence kernel (8 waves) leaves per-frame gradient factors in `grec`; the weight gradients are five
// a2f_mm products over all S*T frames (bias gradients ride along as row sums), LayerNorm gradients are folded from the per-sequence slabs.
static int ff_bwd2_run(const float* params, const float* dy, int64_t S, int64_t T, int32_t period, float* save, float* dparams, float* dcross,
                       float* dstyle, const a2f::FFDrop& dm, bool drop, hipStream_t st) {
  using namespace a2f;
  constexpr size_t LDS_LACC = FF_BWD_LDS + (64 + 2 * (576 + FD + 16 + 2 * FD) + 2 * FF_BLK * FD) * sizeof(float);      // + parked dK / dV rows + the block's accumulators
  static_assert(LDS_LACC <= 160 * 1024, "the LDS-accumulator form of the backward recurrence must fit one CU");
  static bool attr_set[2] = {false, false};
  float* slabs = save + (size_t)S * T * SV_STRIDE;
  float* grec = slabs + (size_t)S * A2F_FF_PARAM_FLOATS;
  float* ws = grec + (size_t)S * T * FF_GREC;
  float* nrec = ws + FF_GEMM_WS_FLOATS;
  int launch_rc = A2F_OK;
  auto recurrence = [&](int lo, int hi) {
    if (!attr_set[drop]) {
      const void* k = drop ? reinterpret_cast<const void*>(&ff_decoder_bwd2_kernel<true>) : reinterpret_cast<const void*>(&ff_decoder_bwd2_kernel<false>);
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LACC);
      if (e != hipSuccess) { launch_rc = set_err(A2F_ELAUNCH, "ff_decoder_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return; }
      attr_set[drop] = true;
    }
    if (drop)
      hipLaunchKernelGGL((ff_decoder_bwd2_kernel<true>), dim3((unsigned)S), dim3(FF_BWD2_THREADS), LDS_LACC, st, params, dy, (int)T, (int)period, save, grec, dcross, dstyle, dm,
                         (const float*)nrec, lo, hi);
    else
      hipLaunchKernelGGL((ff_decoder_bwd2_kernel<false>), dim3((unsigned)S), dim3(FF_BWD2_THREADS), LDS_LACC, st, params, dy, (int)T, (int)period, save, grec, dcross, dstyle, dm,
                         (const float*)nrec, lo, hi);
  };
  if (T < FF_BLK_MIN) {
    recurrence(0, (int)T);
  } else {
    const dim3 pg((unsigned)((T - FF_BLK + 63) / 64), FH, (unsigned)S);
    if (drop) hipLaunchKernelGGL(ff_bwd_prep_kernel<true>, pg, dim3(256), 0, st, (const float*)save, nrec, (int)T, (int)period, dm);
    else hipLaunchKernelGGL(ff_bwd_prep_kernel<false>, pg, dim3(256), 0, st, (const float*)save, nrec, (int)T, (int)period, dm);
    A2F_CHECK_LAUNCH("ff_decoder_bwd (cross-block dq terms)");
    for (int hi = (int)T; hi > 0;) {
      const int lo = ((hi - 1) / FF_BLK) * FF_BLK;
      recurrence(lo, hi);
      if (lo > 0) {
        const dim3 rg((unsigned)((lo + 255) / 256), FH, (unsigned)S);
        if (drop) hipLaunchKernelGGL(ff_bwd_rect_kernel<true>, rg, dim3(256), 0, st, save, (const float*)grec, (int)T, (int)period, lo, hi, dm);
        else hipLaunchKernelGGL(ff_bwd_rect_kernel<false>, rg, dim3(256), 0, st, save, (const float*)grec, (int)T, (int)period, lo, hi, dm);
      }
      hi = lo;
    }
  }
  if (launch_rc != A2F_OK) return launch_rc;
  A2F_CHECK_LAUNCH("ff_decoder_bwd (recurrence)");
  hipLaunchKernelGGL(ff_decoder_fold_range_kernel, dim3(2), dim3(256), 0, st, (const float*)slabs, (int)S, dparams, (int)A2F_FF_LN, 6 * FD);
  A2F_CHECK_LAUNCH("ff_decoder_bwd (LayerNorm fold)");
  const int64_t R = S * T;
  auto product = [&](const float* Abase, int64_t a_ld, int64_t a_rows, const float* Bbase, int64_t b_ld, int64_t b_rows, int64_t M, int64_t N, int64_t K,
                     int64_t lps, int64_t a_off, int64_t b_off, float* Cdst, float* rowsum) -> int {
    a2f_mm_desc d{};
    d.A.p = Abase; d.A.bytes = a_rows * a_ld * 4; d.A.ld = a_ld; d.A.trans = 1;
    d.B.p = Bbase; d.B.bytes = b_rows * b_ld * 4; d.B.ld = b_ld; d.B.trans = 1;
    if (lps > 0) {                               // rows q + 1 of A against rows q of B inside every sequence (the feedback pairs demb_{i+1}, y_i)
      d.A.lines_per_seq = lps; d.A.seq_stride = T * a_ld; d.A.off = a_off; d.A.seq_elems = T * a_ld;
      d.B.lines_per_seq = lps; d.B.seq_stride = T * b_ld; d.B.off = b_off; d.B.seq_elems = T * b_ld;
    }
    d.C = Cdst; d.ldc = N; d.M = M; d.N = N; d.K = K; d.precision = 1; d.batch = 1; d.alpha = 1.f;
    int64_t split = K / 512; split = split < 1 ? 1 : split > FF_GEMM_SPLIT_MAX ? FF_GEMM_SPLIT_MAX : split;
    d.split_k = (int32_t)split; d.ws = ws; d.a_rowsum = rowsum;
    return a2f_mm(&d, (a2f_stream_t)st);
  };
  int rc;
  // plain operands address element (row, column) at base + row * ld + column: pass base + column offset, rows = R
  if ((rc = product(grec + GR_QKV, FF_GREC, R, save + SV_X, SV_STRIDE, R, 192, 64, R, 0, 0, 0, dparams + A2F_FF_WQKV, dparams + A2F_FF_BQKV))) return rc;
  if ((rc = product(grec + GR_SA, FF_GREC, R, save + SV_CTX, SV_STRIDE, R, 64, 64, R, 0, 0, 0, dparams + A2F_FF_WO, dparams + A2F_FF_BO))) return rc;
  if ((rc = product(grec + GR_HP, FF_GREC, R, grec + GR_Y2, FF_GREC, R, 128, 64, R, 0, 0, 0, dparams + A2F_FF_W1, dparams + A2F_FF_B1))) return rc;
  if ((rc = product(grec + GR_F, FF_GREC, R, save + SV_H, SV_STRIDE, R, 64, 128, R, 0, 0, 0, dparams + A2F_FF_W2, dparams + A2F_FF_B2))) return rc;
  if (T > 1) {
    if ((rc = product(grec, FF_GREC, R, grec, FF_GREC, R, 64, 64, S * (T - 1), T - 1, GR_EMB + FF_GREC, GR_Y, dparams + A2F_FF_A, dparams + A2F_FF_C))) return rc;
  } else {
    if (hipMemsetAsync(dparams + A2F_FF_A, 0, (64 * 64 + 64) * sizeof(float), st) != hipSuccess) return set_err(A2F_ELAUNCH, "ff_decoder_bwd: memset");
  }
  return A2F_OK;
}

static int ff_bwd_impl(const float* params, const float* dy, int64_t S, int64_t T, int32_t period, void* save, size_t save_bytes, float* dparams,
                       float* dcross, float* dstyle, const uint32_t* row_bits, const uint32_t* attn_bits, float keep_scale, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(params && dy && save && dparams && dcross && dstyle, "ff_decoder_bwd: null pointer");
  A2F_CHECK_ARG(S > 0 && S < (1 << 20) && T > 0 && T < (1 << 20) && period > 0, "ff_decoder_bwd: S=%lld T=%lld period=%d", (long long)S, (long long)T, period);
  A2F_CHECK_ARG((((uintptr_t)save) & 255) == 0, "ff_decoder_bwd: save buffer must be 256-byte aligned");
  if (save_bytes < a2f_ff_save_bytes(S, T)) return set_err(A2F_EWORKSPACE, "ff_decoder_bwd: save buffer %zu < %zu", save_bytes, a2f_ff_save_bytes(S, T));
  const bool drop = row_bits != nullptr;
  const FFDrop dm{row_bits, attn_bits, (int)a2f_attn_keep_words(T), keep_scale};
  // record form: the recurrence leaves per-frame gradient factors, the weight gradients are products over all frames; blocks of FF_BLK steps from FF_BLK_MIN frames on
  A2F_CHECK_ARG((uint64_t)S * T * SV_STRIDE * 4ull < 0xFFFFFFF0ull, "ff_decoder_bwd: S * T = %lld frames exceed the 4-GiB span of the saved rows (split the batch)",
                (long long)(S * T));
  return ff_bwd2_run(params, dy, S, T, period, (float*)save, dparams, dcross, dstyle, dm, drop, (hipStream_t)stream);
}

extern "C" int a2f_ff_decoder_fwd(const float* params, const float* pe, const float* cross, const float* style, int64_t S, int64_t T,
                                  int32_t period, float* y, void* save, size_t save_bytes, a2f_stream_t stream) {
  return ff_fwd_impl(params, pe, cross, style, S, T, period, y, save, save_bytes, nullptr, nullptr, 1.f, stream);
}

extern "C" int a2f_ff_decoder_bwd(const float* params, const float* dy, int64_t S, int64_t T, int32_t period, void* save, size_t save_bytes,
                                  float* dparams, float* dcross, float* dstyle, a2f_stream_t stream) {
  return ff_bwd_impl(params, dy, S, T, period, save, save_bytes, dparams, dcross, dstyle, nullptr, nullptr, 1.f, stream);
}

extern "C" int a2f_ff_decoder_fwd_drop(const float* params, const float* pe, const float* cross, const float* style, int64_t S, int64_t T,
                                       int32_t period, float* y, void* save, size_t save_bytes, const uint32_t* row_bits, const uint32_t* attn_bits,
                                       float keep_scale, a2f_stream_t stream) {
  int rc = ff_check_drop(row_bits, attn_bits, keep_scale, "ff_decoder_fwd_drop");
  if (rc) return rc;
  return ff_fwd_impl(params, pe, cross, style, S, T, period, y, save, save_bytes, row_bits, attn_bits, keep_scale, stream);
}

extern "C" int a2f_ff_decoder_bwd_drop(const float* params, const float* dy, int64_t S, int64_t T, int32_t period, void* save, size_t save_bytes,
                                       float* dparams, float* dcross, float* dstyle, const uint32_t* row_bits, const uint32_t* attn_bits,
                                       float keep_scale, a2f_stream_t stream) {
  int rc = ff_check_drop(row_bits, attn_bits, keep_scale, "ff_decoder_bwd_drop");
  if (rc) return rc;
  return ff_bwd_impl(params, dy, S, T, period, save, save_bytes, dparams, dcross, dstyle, row_bits, attn_bits, keep_scale, stream);
}

#ifdef FF_STAMPS
// diagnostic build only: the backward recurrence's stamps (s_memtime after every barrier of step FF_STAMP_STEP, sequence 0)
extern "C" int a2f_debug_ff_bwd_stamps(long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(a2f::ffb_dbg), 16 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
