// MFMA flash attention for the wav2vec2 encoder behind FaceFormer: 12 heads x 64, full (unmasked) self-attention over the
// T frames of one sequence (T = 600 at BASELINE configs[2], 3000 at configs[4]), forward and backward.
// Replaces softmax(q k^T) v of HF Wav2Vec2Attention (the encoder layers called at /root/reference/src/model/wav2vec.py:174-180;
// q is pre-scaled by head_dim^-0.5 as HF does) and its autograd backward.
//
// Storage is fp32 in the fused (rows, 3*768) q|k|v projection buffer; operands are converted to bf16 on chip and multiplied
// with v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 softmax).  PREC 1 = bf16x3 split products (x = hi + lo; lo*hi' +
// hi*lo' + hi*hi'; fp32-class: the 1e-4 parity setting), PREC 0 = plain bf16 operands.
//
// Structure (all three kernels): a workgroup = 4 waves; the streamed operand tiles (64 rows x 64) are fetched with 16-byte
// raw buffer loads (a resource per sequence: rows beyond T read as zero), converted in registers and kept in LDS as bf16
// images with 128-byte rows whose 16-byte chunks are XOR-swizzled so that BOTH the row reads (ds_read_b128, MFMA operands
// along d) and the transposing reads (ds_read_b64_tr_b16, operands along the row index) are bank-conflict free; double
// buffered, loads of tile t+1 in flight under the MFMAs of tile t, one barrier per tile.
//   forward : wave = 32 queries; S^T = K Q^T with the query on the lane -> online softmax per lane (one cross-half exchange
//             per tile); P feeds O^T += V^T P straight from the accumulator registers (permuted-k operand order);
//   dK / dV : wave = 32 keys (K, V fragments in registers), streams Q / dO tiles; P and dS feed dV^T, dK^T from registers;
//   dQ      : wave = 32 queries (Q, dO fragments in registers), streams K / V tiles; dS^T feeds dQ^T from registers.
// No atomics anywhere: every output element has exactly one owner (deterministic).
#include "a2f_common.h"
#include "img.h"
#include <type_traits>

namespace a2f {
namespace attn {

using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
using u32x2_t = __attribute__((ext_vector_type(2))) unsigned;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
using i16x4_t = __attribute__((ext_vector_type(4))) short;
using rsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int D = 64;            // head dimension
constexpr int TILE = 64;         // rows of a streamed tile
constexpr int IMG = TILE * 128;  // bytes of one bf16 image of a 64 x 64 tile
constexpr int THREADS = 256;

struct Args {
  const float* qkv;      // (S*T, ld): q at column 0, k at koff, v at voff; head h at + 64 h
  const float* ctx;      // forward output / backward input (S*T, ldo)
  const float* dctx;     // (S*T, ldo)
  const float* lse;      // (S, H, T) log-sum-exp of the scaled scores
  const float* delta;    // (S, H, T) rowsum(dctx * ctx) per head
  float* out_ctx;
  float* out_lse;
  float* out_delta;
  float* dqkv;           // (S*T, ld)
  char* out_ctx_img;     // optional: the context also as a split-32 image (S*T, ldo) - the operand format of the out-projection GEMM and of its weight gradient
  int dqkv_img;          // 1: dqkv is written as a split-32 image (its only readers are the q|k|v weight-gradient / data-gradient GEMMs)
  int qkv_img;           // 1: qkv ARRIVES as a split-32 image (written by the q|k|v projection's GEMM epilogue): tiles and fragments are copied, nothing is converted,
                         //    and q is not pre-scaled - the scale rides in the exponent's fused multiply-add (scale is a power of two for head_dim 64: exact)
  int dctx_img;          // 1: dctx arrives as a split-32 image (written by the out-projection's data-gradient GEMM)
  int ld, ldo, koff, voff;
  int S, T, H;
  float scale;
  // train-mode dropout of the probabilities (HF Wav2Vec2Attention: nn.functional.dropout(attn_weights, p = attention_dropout)): keep bits,
  // query-major words [sh][q][TW] / key-major words [sh][TW][32 TW] (csrc/dropout.hip), kept values are multiplied by keep_scale = 1 / (1 - p)
  const unsigned* keep_q;
  const unsigned* keep_k;
  int TW;
  float keep_scale;
};

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int swz(int row) {           // chunk XOR of the dual-use image
  const int x = (row >> 1) & 7;
  return ((x & 1) << 2) | (x >> 1);
}
__device__ __forceinline__ int img_off(int row, int c16) { return row * 128 + ((c16 ^ swz(row)) << 4); }

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  const f32x2_t v = {a, b};
  const bf16x2_t hb = __builtin_convertvector(v, bf16x2_t);
  const f32x2_t r = v - __builtin_convertvector(hb, f32x2_t);
  hi = __builtin_bit_cast(unsigned, hb);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}
__device__ __forceinline__ unsigned cvt2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ u32x2_t tr_read(const char* lds_addr) {
  return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)(lds_addr)));
}
__device__ __forceinline__ f32x16 mfma(u32x4_t a, u32x4_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// PREC 2 (forward only): fp8 e4m3 operands on v_mfma_f32_32x32x16_fp8_fp8 - BASELINE configs[4] names "fp8 MFMA attention".  The
// LDS images stay bf16 (as PREC 0); a fragment is narrowed to eight e4m3 bytes in registers (elements .x/.y of the first vector)
// right before the product.  Probabilities are multiplied by 2^8 before narrowing (1 / T is below e4m3's normal range) and the
// factor is taken out again with the normaliser.  Measured accuracy: tests/test_gpu_encoder_kernels.py, DESIGN 7.
constexpr float FP8_P_SCALE = 256.f;
__device__ __forceinline__ u32x4_t bf16x8_to_fp8(const u32x4_t v) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xFFFF0000u), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xFFFF0000u), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_bit_cast(float, v.z << 16), __builtin_bit_cast(float, v.z & 0xFFFF0000u), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_bit_cast(float, v.w << 16), __builtin_bit_cast(float, v.w & 0xFFFF0000u), hi, true);
  return u32x4_t{(unsigned)lo, (unsigned)hi, 0u, 0u};
}
__device__ __forceinline__ f32x16 mfma8(const u32x4_t a, const u32x4_t b, f32x16 c) {
  const long la = (long)(((unsigned long long)a.y << 32) | a.x), lb = (long)(((unsigned long long)b.y << 32) | b.x);
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, c, 0, 0, 0);
}
// c += (ah + al)(bh + bl) without lo*lo; small terms first
template <int PREC>
__device__ __forceinline__ f32x16 mac(const u32x4_t (&a)[2], const u32x4_t (&b)[2], f32x16 c) {
  if constexpr (PREC == 2) return mfma8(a[0], b[0], c);
  if constexpr (PREC == 1) {
    c = mfma(a[1], b[0], c);
    c = mfma(a[0], b[1], c);
  }
  return mfma(a[0], b[0], c);
}
// x if bit BP of `word` is set, else +0 (a bit-field-extract + AND form in pinned inline asm measured the same as this select: the kernels are not bound by
// their vector instructions, profiles/r04_t_attention_ab.txt)
template <int BP>
__device__ __forceinline__ float keep_if(float x, unsigned word) { return ((word >> BP) & 1u) ? x : 0.f; }
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ rsrc_t seq_rsrc(const float* base, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(base)), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// ---- a streamed 64 x 64 fp32 tile: 4 float4 per thread (row = (t >> 4) + 16 i, columns 4 (t & 15) ..), bf16 image(s) in LDS
template <int PREC>
struct Tile {
  f32x4 reg[4];
  __device__ __forceinline__ void load(rsrc_t rs, int row0, int ld, int col0, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + (t >> 4) + 16 * i;
      reg[i] = ld4(rs, (unsigned)(row * ld + col0 + 4 * (t & 15)) * 4u);       // rows beyond the sequence are out of range: zeros
    }
  }
  __device__ __forceinline__ void commit(char* img, int t, float mul = 1.f) const {
    const int c4 = t & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (t >> 4) + 16 * i;
      char* p = img + img_off(row, c4 >> 1) + ((c4 & 1) << 3);
      u32x2_t hi, lo;
      if constexpr (PREC == 1) {
        unsigned h0, l0, h1, l1;
        split2(reg[i].x * mul, reg[i].y * mul, h0, l0);
        split2(reg[i].z * mul, reg[i].w * mul, h1, l1);
        hi = u32x2_t{h0, h1};
        lo = u32x2_t{l0, l1};
        *reinterpret_cast<u32x2_t*>(p) = hi;
        *reinterpret_cast<u32x2_t*>(p + IMG) = lo;
      } else {
        hi.x = cvt2(reg[i].x * mul, reg[i].y * mul);
        hi.y = cvt2(reg[i].z * mul, reg[i].w * mul);
        *reinterpret_cast<u32x2_t*>(p) = hi;
      }
    }
  }
};

// The same tile from a split-32 IMAGE (img.h: per 32 elements 32 bf16 hi then 32 bf16 lo): a head's 64 columns of a row are 256 bytes = 16 chunks of
// 16 bytes = 8 bf16 each, chunk c: block c >> 3, lo part if (c >> 2) & 1, elements 32 (c >> 3) + 8 (c & 3) .. + 7.  A chunk IS a 16-byte piece of the
// LDS image row (hi or lo image): thread t copies chunk t & 15 of rows (t >> 4) + 16 i - no conversion, no multiply.
template <int PREC>
struct TileImg {
  u32x4_t reg[4];
  __device__ __forceinline__ void load(rsrc_t rs, int row0, int ld, int col0, int t) {
    const int c = t & 15;
    if (PREC != 1 && ((c >> 2) & 1)) return;                   // single-term products never read the lo parts
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + (t >> 4) + 16 * i;
      reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(row * ld + col0) * 4u + (unsigned)((c >> 3) * 128 + ((c >> 2) & 1) * 64 + (c & 3) * 16), 0, 0);
    }
  }
  __device__ __forceinline__ void commit(char* img, int t) const {
    const int c = t & 15;
    if (PREC != 1 && ((c >> 2) & 1)) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (t >> 4) + 16 * i;
      *reinterpret_cast<u32x4_t*>(img + img_off(row, 4 * (c >> 3) + (c & 3)) + (((c >> 2) & 1) ? IMG : 0)) = reg[i];
    }
  }
};

// MFMA operand along d (k = d): lane (r, h) <- row (rb + r), d = 16 ks + 8 h .. + 7
template <int PREC>
__device__ __forceinline__ void frag_rows(const char* img, int rb, int ks, int lane, u32x4_t (&f)[2]) {
  const int row = rb + (lane & 31);
  const int off = img_off(row, 2 * ks + (lane >> 5));
  f[0] = *reinterpret_cast<const u32x4_t*>(img + off);
  if constexpr (PREC == 1) f[1] = *reinterpret_cast<const u32x4_t*>(img + IMG + off);
  if constexpr (PREC == 2) f[0] = bf16x8_to_fp8(f[0]);
}
// MFMA A operand along the tile's row index (k = row), rows taken in the PERMUTED order of an accumulator tile used as the
// other operand: element j of lane half h <-> row rb + 16 s + 8 (j >> 2) + 4 h + (j & 3); lane r <-> column cb + r
template <int PREC>
__device__ __forceinline__ void frag_cols(const char* img, int rb, int s, int cb, int lane, u32x4_t (&f)[2]) {
  const int g = lane >> 4, i = lane & 15;
  const int row = rb + 16 * s + 4 * (g >> 1) + (i >> 2);
  const int col = cb + 16 * (g & 1) + 4 * (i & 3);
  const int o0 = img_off(row, col >> 3) + ((col & 7) << 1);
  const int o1 = img_off(row + 8, col >> 3) + ((col & 7) << 1);
  const u32x2_t a0 = tr_read(img + o0), a1 = tr_read(img + o1);
  f[0] = u32x4_t{a0.x, a0.y, a1.x, a1.y};
  if constexpr (PREC == 1) {
    const u32x2_t b0 = tr_read(img + IMG + o0), b1 = tr_read(img + IMG + o1);
    f[1] = u32x4_t{b0.x, b0.y, b1.x, b1.y};
  }
  if constexpr (PREC == 2) f[0] = bf16x8_to_fp8(f[0]);
}
// registers 8 s .. 8 s + 7 of an accumulator tile as the bf16 operand of k-step s (permuted-k order, see frag_cols)
template <int PREC>
__device__ __forceinline__ void acc_operand(const f32x16& x, int s, u32x4_t (&f)[2]) {
  if constexpr (PREC == 2) {                  // probabilities: scaled into e4m3's normal range, narrowed straight from fp32
    int lo8 = 0, hi8 = 0;
    lo8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[8 * s + 0] * FP8_P_SCALE, x[8 * s + 1] * FP8_P_SCALE, lo8, false);
    lo8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[8 * s + 2] * FP8_P_SCALE, x[8 * s + 3] * FP8_P_SCALE, lo8, true);
    hi8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[8 * s + 4] * FP8_P_SCALE, x[8 * s + 5] * FP8_P_SCALE, hi8, false);
    hi8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[8 * s + 6] * FP8_P_SCALE, x[8 * s + 7] * FP8_P_SCALE, hi8, true);
    f[0] = u32x4_t{(unsigned)lo8, (unsigned)hi8, 0u, 0u};
    return;
  }
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (PREC == 1) split2(x[8 * s + 2 * j], x[8 * s + 2 * j + 1], hi[j], lo[j]);
    else hi[j] = cvt2(x[8 * s + 2 * j], x[8 * s + 2 * j + 1]);
  }
  f[0] = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
  if constexpr (PREC == 1) f[1] = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
}
// per-lane fragments of a (row = lane & 31) x 64 operand held in registers for the whole kernel: k-step ks <- d = 16 ks + 8 h ..
template <int PREC>
__device__ __forceinline__ void reg_operand(rsrc_t rs, int row, int ld, int col0, int lane, float mul, u32x4_t (&f)[4][2]) {
  const int h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned off = (unsigned)(row * ld + col0 + 16 * ks + 8 * h) * 4u;
    const f32x4 a = ld4(rs, off), b = ld4(rs, off + 16u);
    unsigned hi[4], lo[4];
    if constexpr (PREC == 1) {
      split2(a.x * mul, a.y * mul, hi[0], lo[0]);
      split2(a.z * mul, a.w * mul, hi[1], lo[1]);
      split2(b.x * mul, b.y * mul, hi[2], lo[2]);
      split2(b.z * mul, b.w * mul, hi[3], lo[3]);
      f[ks][1] = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
    } else {
      hi[0] = cvt2(a.x * mul, a.y * mul);
      hi[1] = cvt2(a.z * mul, a.w * mul);
      hi[2] = cvt2(b.x * mul, b.y * mul);
      hi[3] = cvt2(b.z * mul, b.w * mul);
    }
    f[ks][0] = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
    if constexpr (PREC == 2) f[ks][0] = bf16x8_to_fp8(f[ks][0]);
  }
}

// the same from a split-32 image: d = 16 ks + 8 h .. + 7 is one 16-byte chunk of the hi part and one of the lo part
template <int PREC>
__device__ __forceinline__ void reg_operand_img(rsrc_t rs, int row, int ld, int col0, int lane, u32x4_t (&f)[4][2]) {
  const int h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int d0 = 16 * ks + 8 * h;
    const unsigned off = (unsigned)(row * ld + col0) * 4u + (unsigned)((d0 >> 5) * 128 + ((d0 & 31) >> 3) * 16);
    f[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    if constexpr (PREC == 1) f[ks][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 64u, 0, 0);
    if constexpr (PREC == 2) f[ks][0] = bf16x8_to_fp8(f[ks][0]);
  }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
// QIMG: qkv arrives as a split-32 image (Args.qkv_img)
template <int PREC, bool DROP, bool QIMG>
__global__ __launch_bounds__(THREADS, 2) void attn_fwd_kernel(const Args a) {
  constexpr int NIMG = PREC == 1 ? 2 : 1;
  constexpr int STAGE = 2 * NIMG * IMG;                       // K images, V images
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5;
  const int head = blockIdx.y, seq = blockIdx.z;
  const int T = a.T;
  const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const rsrc_t rs = seq_rsrc(a.qkv + (long long)seq * T * a.ld, (long long)T * a.ld * 4);
  const int hc = head * D;

  u32x4_t qf[4][2];
  if constexpr (QIMG) reg_operand_img<PREC>(rs, q, a.ld, hc, lane, qf);
  else reg_operand<PREC>(rs, q, a.ld, hc, lane, a.scale, qf);
  const float sl2 = (QIMG ? a.scale : 1.f) * LOG2E;             // scores -> units of log2 (image operands are not pre-scaled)

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const unsigned* keepq = nullptr;                             // this lane's query: TW words, bit = key
  if constexpr (DROP) keepq = a.keep_q + (((long long)seq * a.H + head) * T + min(q, T - 1)) * a.TW;

  const int nt = (T + TILE - 1) / TILE;
  std::conditional_t<QIMG, TileImg<PREC>, Tile<PREC>> kt_, vt_;
  kt_.load(rs, 0, a.ld, a.koff + hc, t);
  vt_.load(rs, 0, a.ld, a.voff + hc, t);
  kt_.commit(smem, t);
  vt_.commit(smem + NIMG * IMG, t);
  // the keep words of tile it + 1 are requested with that tile's rows, one iteration ahead: read where they are used they put an s_waitcnt vmcnt(0) in front
  // of the softmax, which also waits for the K / V rows just requested - an exposed global round trip per tile in train mode
  u32x2_t keep_nx = {0u, 0u};
  if constexpr (DROP) keep_nx = *reinterpret_cast<const u32x2_t*>(keepq);
  __syncthreads();

  for (int it = 0; it < nt; ++it) {
    const char* kimg = smem + (it & 1) * STAGE;
    const char* vimg = kimg + NIMG * IMG;
    const u32x2_t keep = keep_nx >> (unsigned)(4 * h);          // bit (r & 3) + 8 (r >> 2) <-> register r
    if (it + 1 < nt) {
      kt_.load(rs, (it + 1) * TILE, a.ld, a.koff + hc, t);
      vt_.load(rs, (it + 1) * TILE, a.ld, a.voff + hc, t);
      if constexpr (DROP) keep_nx = *reinterpret_cast<const u32x2_t*>(keepq + 2 * (it + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- S^T = K Q^T: keys on the registers, this lane's query on the lane
    // the K fragment of step n + 1 is requested before the three MFMAs of step n (left alone, the compiler waited for every ds_read_b128 right where it
    // issued it: sixteen exposed LDS round trips per tile)
    f32x16 st[2];
    {
      u32x4_t kfr[2][2];
      frag_rows<PREC>(kimg, 0, 0, lane, kfr[0]);
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int kb = n >> 2, ks = n & 3;
        if (n + 1 < 8) frag_rows<PREC>(kimg, 32 * ((n + 1) >> 2), (n + 1) & 3, lane, kfr[(n + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);                      // the reads above stay above the products below
        if (ks == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
        }
        st[kb] = mac<PREC>(kfr[n & 1], qf[ks], st[kb]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if ((it + 1) * TILE > T) {                                  // ragged last tile: keys beyond T do not exist
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (it * TILE + 32 * kb + rho(r, h) >= T) st[kb][r] = -INFINITY;
    }
    // ---- online softmax (per lane = per query; the two lane halves hold disjoint keys of the same query).  The running maximum is kept in
    // units of log2: p = exp2(s log2e - m2) is one fused multiply-add and one v_exp_f32 per score (exp(s - m) is a subtract, a multiply and the v_exp_f32)
    float mx = st[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx * sl2);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[kb][r] = __builtin_amdgcn_exp2f(fmaf(st[kb][r], sl2, -mn));
        ps += st[kb][r];
      }
    l = l * alpha + ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) {                              // (skipping this while the maximum stands still - a wave-uniform branch - measured no gain)
      o[0][r] *= alpha;
      o[1][r] *= alpha;
    }
    if constexpr (DROP) {                                       // the normaliser l keeps every key; only the value sum is thinned
#define A2F_KEEP_FWD(R)                                                  \
      st[0][R] = keep_if<((R) & 3) + 8 * ((R) >> 2)>(st[0][R], keep.x); \
      st[1][R] = keep_if<((R) & 3) + 8 * ((R) >> 2)>(st[1][R], keep.y);
      A2F_KEEP_FWD(0) A2F_KEEP_FWD(1) A2F_KEEP_FWD(2) A2F_KEEP_FWD(3) A2F_KEEP_FWD(4) A2F_KEEP_FWD(5) A2F_KEEP_FWD(6) A2F_KEEP_FWD(7)
      A2F_KEEP_FWD(8) A2F_KEEP_FWD(9) A2F_KEEP_FWD(10) A2F_KEEP_FWD(11) A2F_KEEP_FWD(12) A2F_KEEP_FWD(13) A2F_KEEP_FWD(14) A2F_KEEP_FWD(15)
#undef A2F_KEEP_FWD
    }
    // ---- O^T += V^T P: P straight from the accumulator registers
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4_t pf[2];
        acc_operand<PREC>(st[kb], s, pf);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          u32x4_t vf[2];
          frag_cols<PREC>(vimg, 32 * kb, s, 32 * dt, lane, vf);
          o[dt] = mac<PREC>(vf, pf, o[dt]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < nt) {
      char* nk = smem + ((it + 1) & 1) * STAGE;
      kt_.commit(nk, t);
      vt_.commit(nk + NIMG * IMG, t);
    }
    __syncthreads();
  }

  // ---- epilogue: O = O^T / l, rows = this lane's query, 4 consecutive d per register quad
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = (DROP ? a.keep_scale : 1.f) / (PREC == 2 ? lt * FP8_P_SCALE : lt);
  if (q < T) {
    float* orow = a.out_ctx + ((long long)seq * T + q) * a.ldo + hc;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 v = {o[dt][4 * gq] * inv, o[dt][4 * gq + 1] * inv, o[dt][4 * gq + 2] * inv, o[dt][4 * gq + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * gq + 4 * h) = v;
        if (a.out_ctx_img) img::store4(a.out_ctx_img + ((long long)seq * T + q) * a.ldo * 4, hc + 32 * dt + 8 * gq + 4 * h, v);
      }
    if (h == 0) a.out_lse[((long long)seq * a.H + head) * T + q] = m * LN2 + __logf(lt);
  }
}

// delta[s,h,q] = sum_d dctx[q, 64 h + d] * ctx[q, 64 h + d]: one wave per row of (S*T), lane l <-> columns 4 l .. (ldo <= 1024)
__global__ __launch_bounds__(THREADS) void attn_delta_kernel(const Args a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
  if (row >= (long long)a.S * a.T) return;
  const int seq = (int)(row / a.T), q = (int)(row - (long long)seq * a.T);
  const int ncol = a.H * D;
  for (int c = 4 * lane; c < ncol; c += 256) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(a.ctx + row * a.ldo + c);
    const f32x4 y = a.dctx_img ? img::load4(reinterpret_cast<const char*>(a.dctx) + row * a.ldo * 4, c) : *reinterpret_cast<const f32x4*>(a.dctx + row * a.ldo + c);
    float s = (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    // 16 lanes share a head (64 columns = 16 float4)
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if ((lane & 15) == 0) a.out_delta[((long long)seq * a.H + c / D) * a.T + q] = s;
  }
}

// =====================================================================================================================
// backward, dK and dV: wave = 32 keys (registers), streams Q / dO tiles of 64 queries
//   S[q][key] = Qs K^T (query on the registers, key on the lane);  P = exp(S - lse[q]);  dP = dO V^T;  dS = P (dP - delta[q])
//   dV^T[d][key] += dO^T P;   dK^T[d][key] += Qs^T dS           (Qs = scale * Q, so dK needs no further factor)
// =====================================================================================================================
// INIMG: qkv AND dctx arrive as split-32 images (Args.qkv_img, Args.dctx_img)
template <int PREC, bool DROP, bool INIMG>
__global__ __launch_bounds__(THREADS, 2) void attn_bwd_dkv_kernel(const Args a) {
  constexpr int NIMG = PREC == 1 ? 2 : 1;
  constexpr int STAGE = 2 * NIMG * IMG;                       // Q images, dO images
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5;
  const int head = blockIdx.y, seq = blockIdx.z;
  const int T = a.T;
  const int key = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const rsrc_t rs = seq_rsrc(a.qkv + (long long)seq * T * a.ld, (long long)T * a.ld * 4);
  const rsrc_t rd = seq_rsrc(a.dctx + (long long)seq * T * a.ldo, (long long)T * a.ldo * 4);
  const int hc = head * D;
  const float* lse = a.lse + ((long long)seq * a.H + head) * T;
  const float* dlt = a.delta + ((long long)seq * a.H + head) * T;
  // this lane's key: word (qw) at + qw * 32 TW, bit = query 32 qw + b.  Through a buffer resource of the (sequence, head) block: a 32-bit lane offset
  // instead of a 64-bit pointer per lane (the kernel sits at its 256 registers)
  rsrc_t rkeep = rs;
  unsigned keep_wg = 0u;
  if constexpr (DROP) {
    rkeep = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(reinterpret_cast<const unsigned*>(uniform_ptr(reinterpret_cast<const float*>(
                a.keep_k + ((long long)seq * a.H + head) * a.TW * (32ll * a.TW))))), 0, a.TW * 32 * a.TW * 4, 0x00020000);
    keep_wg = (unsigned)(blockIdx.x * 128) * 4u;                 // word (qw, key) at qw * 32 TW + key; keys beyond 32 TW are out of range of the resource: 0
  }

  u32x4_t kf[4][2], vf[4][2];
  if constexpr (INIMG) {
    reg_operand_img<PREC>(rs, key, a.ld, a.koff + hc, lane, kf);
    reg_operand_img<PREC>(rs, key, a.ld, a.voff + hc, lane, vf);
  } else {
    reg_operand<PREC>(rs, key, a.ld, a.koff + hc, lane, 1.f, kf);
    reg_operand<PREC>(rs, key, a.ld, a.voff + hc, lane, 1.f, vf);
  }
  const float sl2 = (INIMG ? a.scale : 1.f) * LOG2E;              // image Q tiles are not pre-scaled: the scale rides in the exponent, and on dK at the end

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) dk[0][r] = dk[1][r] = dv[0][r] = dv[1][r] = 0.f;

  const int nt = (T + TILE - 1) / TILE;
  std::conditional_t<INIMG, TileImg<PREC>, Tile<PREC>> qt_, dt_;
  // lse (in units of log2: times log2e) and delta of a tile's 64 queries travel with the tile through LDS ([stage][lse2 64 | delta 64]): read back as four
  // float4 per 32-query sub-tile instead of 64 scalar global loads per tile and wave.  A query beyond T gets lse2 = +inf: its probability is exp2(-inf) = 0.
  float* lsd = reinterpret_cast<float*>(smem + 2 * STAGE);
  unsigned* kws = reinterpret_cast<unsigned*>(lsd + 256);       // train mode: [stage][sub-tile][key of the workgroup] keep words, the same way (one word per thread)
  float aux_r = 0.f;
  unsigned aux_k = 0u;
  auto aux_load = [&](int tile) {
    if (t < 128) {
      const int qq = tile * TILE + (t & 63);
      aux_r = t < 64 ? (qq < T ? lse[qq] * LOG2E : INFINITY) : (qq < T ? dlt[qq] : 0.f);
    }
    if constexpr (DROP) aux_k = __builtin_amdgcn_raw_buffer_load_b32(rkeep, keep_wg + (unsigned)(t & 127) * 4u, (unsigned)((2 * tile + (t >> 7)) * 32 * a.TW * 4), 0);
  };
  auto aux_commit = [&](int stage) {
    if (t < 128) lsd[stage * 128 + t] = aux_r;
    if constexpr (DROP) kws[stage * 256 + t] = aux_k;
  };
  qt_.load(rs, 0, a.ld, hc, t);
  dt_.load(rd, 0, a.ldo, hc, t);
  aux_load(0);
  if constexpr (INIMG) qt_.commit(smem, t); else qt_.commit(smem, t, a.scale);
  dt_.commit(smem + NIMG * IMG, t);
  aux_commit(0);
  __syncthreads();

  for (int it = 0; it < nt; ++it) {
    const char* qimg = smem + (it & 1) * STAGE;
    const char* dimg = qimg + NIMG * IMG;
    const float* lq = lsd + (it & 1) * 128;
    if (it + 1 < nt) {
      qt_.load(rs, (it + 1) * TILE, a.ld, hc, t);
      dt_.load(rd, (it + 1) * TILE, a.ldo, hc, t);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {                          // two 32-query sub-tiles (not unrolled: interleaved, the two sub-tiles' values do not fit the register file)
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
      {                                                        // fragments of k-step ks + 1 requested before the six MFMAs of step ks (cf. the forward)
        u32x4_t qa[2][2], da[2][2];
        frag_rows<PREC>(qimg, 32 * qb, 0, lane, qa[0]);
        frag_rows<PREC>(dimg, 32 * qb, 0, lane, da[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks + 1 < 4) {
            frag_rows<PREC>(qimg, 32 * qb, ks + 1, lane, qa[(ks + 1) & 1]);
            frag_rows<PREC>(dimg, 32 * qb, ks + 1, lane, da[(ks + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          s = mac<PREC>(qa[ks & 1], kf[ks], s);
          dp = mac<PREC>(da[ks & 1], vf[ks], dp);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // rows = queries it*64 + 32 qb + rho(r, h); P = exp(S - lse), dS = P (dP - delta)
      unsigned keep = 0u;
      if constexpr (DROP) keep = kws[(it & 1) * 256 + qb * 128 + wave * 32 + (lane & 31)] >> (unsigned)(4 * h);
      if constexpr (DROP) {          // dP = (dO V^T) o D first, as a pass of its own (one live mask word; fused into the exponential pass below the kernel spilled 98 registers)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = ((keep >> ((r & 3) + 8 * (r >> 2))) & 1u) ? dp[r] * a.keep_scale : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = 32 * qb + rho(r, h);
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lq[ql]));
        dp[r] = p * (dp[r] - lq[64 + ql]);
        s[r] = p;
      }
      if constexpr (DROP) {          // ctx = (P o D) V: dV sees the thinned P (its 1 / (1 - p) at the end)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = ((keep >> ((r & 3) + 8 * (r >> 2))) & 1u) ? s[r] : 0.f;
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        u32x4_t pf[2], sf[2];
        acc_operand<PREC>(s, ss, pf);
        acc_operand<PREC>(dp, ss, sf);
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          u32x4_t oa[2], qa[2];
          frag_cols<PREC>(dimg, 32 * qb, ss, 32 * dd, lane, oa);
          frag_cols<PREC>(qimg, 32 * qb, ss, 32 * dd, lane, qa);
          dv[dd] = mac<PREC>(oa, pf, dv[dd]);
          dk[dd] = mac<PREC>(qa, sf, dk[dd]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < nt) {
      char* nq = smem + ((it + 1) & 1) * STAGE;
      if constexpr (INIMG) qt_.commit(nq, t); else qt_.commit(nq, t, a.scale);
      dt_.commit(nq + NIMG * IMG, t);
      aux_load(it + 1);                                      // 128 L2-resident floats, requested here so that they hold no register across the products above
      aux_commit((it + 1) & 1);
    }
    __syncthreads();
  }
  if (key < T) {
    float* krow = a.dqkv + ((long long)seq * T + key) * a.ld + hc;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float kscl = INIMG ? a.scale : 1.f;
        const f32x4 x = {dk[dd][4 * gq] * kscl, dk[dd][4 * gq + 1] * kscl, dk[dd][4 * gq + 2] * kscl, dk[dd][4 * gq + 3] * kscl};
        const float vs = DROP ? a.keep_scale : 1.f;
        const f32x4 y = {dv[dd][4 * gq] * vs, dv[dd][4 * gq + 1] * vs, dv[dd][4 * gq + 2] * vs, dv[dd][4 * gq + 3] * vs};
        if (a.dqkv_img) {
          char* kimg_row = reinterpret_cast<char*>(a.dqkv) + ((long long)seq * T + key) * a.ld * 4;
          img::store4(kimg_row, hc + a.koff + 32 * dd + 8 * gq + 4 * h, x);
          img::store4(kimg_row, hc + a.voff + 32 * dd + 8 * gq + 4 * h, y);
        } else {
          *reinterpret_cast<f32x4*>(krow + a.koff + 32 * dd + 8 * gq + 4 * h) = x;
          *reinterpret_cast<f32x4*>(krow + a.voff + 32 * dd + 8 * gq + 4 * h) = y;
        }
      }
  }
}

// =====================================================================================================================
// backward, dQ: wave = 32 queries (Qs, dO fragments in registers), streams K / V tiles of 64 keys
//   S^T[key][q] = K Qs^T;  P^T = exp(S^T - lse[q]);  dP^T = V dO^T;  dS^T = P^T (dP^T - delta[q]);  dQ^T[d][q] += K^T dS^T
//   dQ = scale * dQ^T^T
// =====================================================================================================================
template <int PREC, bool DROP, bool INIMG>
__global__ __launch_bounds__(THREADS, 2) void attn_bwd_dq_kernel(const Args a) {
  constexpr int NIMG = PREC == 1 ? 2 : 1;
  constexpr int STAGE = 2 * NIMG * IMG;                       // K images, V images
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5;
  const int head = blockIdx.y, seq = blockIdx.z;
  const int T = a.T;
  const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const rsrc_t rs = seq_rsrc(a.qkv + (long long)seq * T * a.ld, (long long)T * a.ld * 4);
  const rsrc_t rd = seq_rsrc(a.dctx + (long long)seq * T * a.ldo, (long long)T * a.ldo * 4);
  const int hc = head * D;
  const bool qok = q < T;
  const float ls2 = qok ? a.lse[((long long)seq * a.H + head) * T + q] * LOG2E : 0.f;       // a query beyond T has zero q and dO rows: its dS is zero whatever ls2 is
  const float de = qok ? a.delta[((long long)seq * a.H + head) * T + q] : 0.f;
  const unsigned* keepq = nullptr;
  if constexpr (DROP) keepq = a.keep_q + (((long long)seq * a.H + head) * T + min(q, T - 1)) * a.TW;

  u32x4_t qf[4][2], of[4][2];
  if constexpr (INIMG) {
    reg_operand_img<PREC>(rs, q, a.ld, hc, lane, qf);
    reg_operand_img<PREC>(rd, q, a.ldo, hc, lane, of);
  } else {
    reg_operand<PREC>(rs, q, a.ld, hc, lane, a.scale, qf);
    reg_operand<PREC>(rd, q, a.ldo, hc, lane, 1.f, of);
  }
  const float sl2 = (INIMG ? a.scale : 1.f) * LOG2E;

  f32x16 dq[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[0][r] = dq[1][r] = 0.f;

  const int nt = (T + TILE - 1) / TILE;
  std::conditional_t<INIMG, TileImg<PREC>, Tile<PREC>> kt_, vt_;
  kt_.load(rs, 0, a.ld, a.koff + hc, t);
  vt_.load(rs, 0, a.ld, a.voff + hc, t);
  kt_.commit(smem, t);
  vt_.commit(smem + NIMG * IMG, t);
  u32x2_t keep_nx = {0u, 0u};                                   // keep words one tile ahead (cf. the forward)
  if constexpr (DROP) keep_nx = *reinterpret_cast<const u32x2_t*>(keepq);
  __syncthreads();

  for (int it = 0; it < nt; ++it) {
    const char* kimg = smem + (it & 1) * STAGE;
    const char* vimg = kimg + NIMG * IMG;
    const u32x2_t keep2 = keep_nx >> (unsigned)(4 * h);
    if (it + 1 < nt) {
      kt_.load(rs, (it + 1) * TILE, a.ld, a.koff + hc, t);
      vt_.load(rs, (it + 1) * TILE, a.ld, a.voff + hc, t);
      if constexpr (DROP) keep_nx = *reinterpret_cast<const u32x2_t*>(keepq + 2 * (it + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
      {
        u32x4_t ka[2][2], va[2][2];
        frag_rows<PREC>(kimg, 32 * kb, 0, lane, ka[0]);
        frag_rows<PREC>(vimg, 32 * kb, 0, lane, va[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks + 1 < 4) {
            frag_rows<PREC>(kimg, 32 * kb, ks + 1, lane, ka[(ks + 1) & 1]);
            frag_rows<PREC>(vimg, 32 * kb, ks + 1, lane, va[(ks + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          s = mac<PREC>(ka[ks & 1], qf[ks], s);
          dp = mac<PREC>(va[ks & 1], of[ks], dp);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      const unsigned kw = kb == 0 ? keep2.x : keep2.y;
#define A2F_DQ_ELEM(R)                                                                                              \
      {                                                                                                           \
        const float p = __builtin_amdgcn_exp2f(fmaf(s[R], sl2, -ls2));                                            \
        if constexpr (DROP) dp[R] = p * (keep_if<((R) & 3) + 8 * ((R) >> 2)>(dp[R] * a.keep_scale, kw) - de);     \
        else dp[R] = p * (dp[R] - de);                                                                            \
      }
      A2F_DQ_ELEM(0) A2F_DQ_ELEM(1) A2F_DQ_ELEM(2) A2F_DQ_ELEM(3) A2F_DQ_ELEM(4) A2F_DQ_ELEM(5) A2F_DQ_ELEM(6) A2F_DQ_ELEM(7)
      A2F_DQ_ELEM(8) A2F_DQ_ELEM(9) A2F_DQ_ELEM(10) A2F_DQ_ELEM(11) A2F_DQ_ELEM(12) A2F_DQ_ELEM(13) A2F_DQ_ELEM(14) A2F_DQ_ELEM(15)
#undef A2F_DQ_ELEM
      if ((it + 1) * TILE > T) {                               // ragged last tile (wave-uniform): keys beyond T do not exist
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (it * TILE + 32 * kb + rho(r, h) >= T) dp[r] = 0.f;
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        u32x4_t sf[2];
        acc_operand<PREC>(dp, ss, sf);
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          u32x4_t ka[2];
          frag_cols<PREC>(kimg, 32 * kb, ss, 32 * dd, lane, ka);
          dq[dd] = mac<PREC>(ka, sf, dq[dd]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < nt) {
      char* nk = smem + ((it + 1) & 1) * STAGE;
      kt_.commit(nk, t);
      vt_.commit(nk + NIMG * IMG, t);
    }
    __syncthreads();
  }
  if (qok) {
    float* qrow = a.dqkv + ((long long)seq * T + q) * a.ld + hc;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x4 x = {dq[dd][4 * gq] * a.scale, dq[dd][4 * gq + 1] * a.scale, dq[dd][4 * gq + 2] * a.scale, dq[dd][4 * gq + 3] * a.scale};
        if (a.dqkv_img) img::store4(reinterpret_cast<char*>(a.dqkv) + ((long long)seq * T + q) * a.ld * 4, hc + 32 * dd + 8 * gq + 4 * h, x);
        else *reinterpret_cast<f32x4*>(qrow + 32 * dd + 8 * gq + 4 * h) = x;
      }
  }
}

template <typename K>
static int set_lds(K kern, int bytes, bool& done) {
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      return set_err(A2F_ELAUNCH, "attention: cannot reserve %d bytes of LDS", bytes);
    done = true;
  }
  return A2F_OK;
}

static int check_common(const float* qkv, int64_t S, int64_t T, int64_t H, int64_t ld, int64_t ldo, int precision) {
  A2F_CHECK_ARG(qkv != nullptr, "attention: null qkv");
  A2F_CHECK_ARG(S > 0 && T > 0 && H > 0 && H * 64 <= ldo && 3 * H * 64 <= ld, "attention: S=%lld T=%lld H=%lld ld=%lld ldo=%lld", (long long)S,
                (long long)T, (long long)H, (long long)ld, (long long)ldo);
  A2F_CHECK_ARG(ld % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)qkv & 15) == 0, "attention: strides must be multiples of 4 floats, buffers 16-byte aligned");
  A2F_CHECK_ARG(T * ld * 4 < (1ll << 31) && H * 64 <= 1024, "attention: one sequence of the q|k|v buffer must span < 2 GiB");
  A2F_CHECK_ARG(precision >= 0 && precision <= 2, "attention: precision %d (0 bf16, 1 bf16x3, 2 fp8 forward)", precision);
  return A2F_OK;
}

}  // namespace attn
}  // namespace a2f

static int attn_fwd_impl(const float* qkv, int64_t S, int64_t T, int64_t H, int64_t ld, float scale, int32_t precision, float* ctx, int64_t ldo,
                         float* lse, const uint32_t* keep_q, float keep_scale, a2f_stream_t stream, void* ctx_img = nullptr, int32_t qkv_format = 0) {
  using namespace a2f;
  using namespace a2f::attn;
  int rc = check_common(qkv, S, T, H, ld, ldo, precision);
  if (rc) return rc;
  A2F_CHECK_ARG(ctx && lse && ((uintptr_t)ctx & 15) == 0, "attn_fwd: null / misaligned output");
  Args a{};
  a.qkv = qkv; a.out_ctx = ctx; a.out_lse = lse;
  A2F_CHECK_ARG(ctx_img == nullptr || (ldo % 32 == 0 && ((uintptr_t)ctx_img & 127) == 0), "attn_fwd: an image output needs ldo %% 32 == 0 and a 128-byte aligned buffer");
  a.out_ctx_img = (char*)ctx_img;
  A2F_CHECK_ARG(qkv_format == 0 || (qkv_format == 1 && ld % 32 == 0 && ((uintptr_t)qkv & 127) == 0), "attn_fwd: qkv_format %d (1 = image: ld %% 32 == 0, 128-byte aligned)", qkv_format);
  a.qkv_img = qkv_format;
  a.ld = (int)ld; a.ldo = (int)ldo; a.koff = (int)(H * 64); a.voff = (int)(2 * H * 64);
  a.S = (int)S; a.T = (int)T; a.H = (int)H; a.scale = scale;
  a.keep_q = keep_q; a.TW = (int)a2f_attn_keep_words(T); a.keep_scale = keep_scale;
  const dim3 grid((unsigned)cdiv(T, 128), (unsigned)H, (unsigned)S);
  hipStream_t st = (hipStream_t)stream;
  static bool done[2][6] = {{false, false, false, false, false, false}, {false, false, false, false, false, false}};
  const int lds = precision == 1 ? 2 * 4 * IMG : 2 * 2 * IMG;
#define A2F_ATTN_FWD(P, D)                                                                  \
  do {                                                                                      \
    if (a.qkv_img) {                                                                        \
      if ((rc = set_lds(attn_fwd_kernel<P, D, true>, lds, done[1][2 * P + D]))) return rc;  \
      hipLaunchKernelGGL((attn_fwd_kernel<P, D, true>), grid, dim3(THREADS), lds, st, a);   \
    } else {                                                                                \
      if ((rc = set_lds(attn_fwd_kernel<P, D, false>, lds, done[0][2 * P + D]))) return rc; \
      hipLaunchKernelGGL((attn_fwd_kernel<P, D, false>), grid, dim3(THREADS), lds, st, a);  \
    }                                                                                       \
  } while (0)
  if (precision == 1) { if (keep_q) A2F_ATTN_FWD(1, true); else A2F_ATTN_FWD(1, false); }
  else if (precision == 2) { if (keep_q) A2F_ATTN_FWD(2, true); else A2F_ATTN_FWD(2, false); }
  else { if (keep_q) A2F_ATTN_FWD(0, true); else A2F_ATTN_FWD(0, false); }
#undef A2F_ATTN_FWD
  A2F_CHECK_LAUNCH("attn_fwd");
  return A2F_OK;
}

static int attn_bwd_impl(const float* qkv, const float* ctx, const float* dctx, const float* lse, int64_t S, int64_t T, int64_t H, int64_t ld,
                         int64_t ldo, float scale, int32_t precision, float* dqkv, float* delta_ws, const uint32_t* keep_q, const uint32_t* keep_k,
                         float keep_scale, a2f_stream_t stream, int32_t dqkv_format = 0, int32_t in_format = 0) {
  using namespace a2f;
  using namespace a2f::attn;
  int rc = check_common(qkv, S, T, H, ld, ldo, precision);
  if (rc) return rc;
  A2F_CHECK_ARG(ctx && dctx && lse && dqkv && delta_ws, "attn_bwd: null pointer");
  A2F_CHECK_ARG(dqkv_format == 0 || (dqkv_format == 1 && ld % 32 == 0 && ((uintptr_t)dqkv & 127) == 0), "attn_bwd: dqkv_format %d (1 = image: ld %% 32 == 0, 128-byte aligned)", dqkv_format);
  A2F_CHECK_ARG((((uintptr_t)ctx | (uintptr_t)dctx | (uintptr_t)dqkv) & 15) == 0, "attn_bwd: buffers must be 16-byte aligned");
  A2F_CHECK_ARG((keep_q == nullptr) == (keep_k == nullptr), "attn_bwd: both keep-bit layouts or neither");
  Args a{};
  a.qkv = qkv; a.ctx = ctx; a.dctx = dctx; a.lse = lse; a.delta = delta_ws; a.out_delta = delta_ws; a.dqkv = dqkv;
  a.ld = (int)ld; a.ldo = (int)ldo; a.koff = (int)(H * 64); a.voff = (int)(2 * H * 64);
  a.S = (int)S; a.T = (int)T; a.H = (int)H; a.scale = scale;
  a.keep_q = keep_q; a.keep_k = keep_k; a.TW = (int)a2f_attn_keep_words(T); a.keep_scale = keep_scale;
  a.dqkv_img = dqkv_format;
  A2F_CHECK_ARG(in_format == 0 || (in_format == 1 && ld % 32 == 0 && ldo % 32 == 0 && (((uintptr_t)qkv | (uintptr_t)dctx) & 127) == 0),
                "attn_bwd: in_format %d (1 = qkv and dctx are images: ld, ldo %% 32 == 0, 128-byte aligned)", in_format);
  a.qkv_img = a.dctx_img = in_format;
  if (precision == 2) precision = 0;          // fp8 is a forward mode: gradients would need dynamic scaling; the backward runs its bf16 kernels
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdiv(S * T, THREADS / 64)), dim3(THREADS), 0, st, a);
  A2F_CHECK_LAUNCH("attn_bwd delta");
  const dim3 grid((unsigned)cdiv(T, 128), (unsigned)H, (unsigned)S);
  static bool dk[2][4] = {{false, false, false, false}, {false, false, false, false}}, dq[2][4] = {{false, false, false, false}, {false, false, false, false}};
  const int lds = precision == 1 ? 2 * 4 * IMG : 2 * 2 * IMG;
#define A2F_ATTN_BWD_I(P, D, I)                                                                     \
  do {                                                                                              \
    if ((rc = set_lds(attn_bwd_dkv_kernel<P, D, I>, lds + 3072, dk[I][2 * P + D]))) return rc;      \
    if ((rc = set_lds(attn_bwd_dq_kernel<P, D, I>, lds, dq[I][2 * P + D]))) return rc;              \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<P, D, I>), grid, dim3(THREADS), lds + 3072, st, a);     \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<P, D, I>), grid, dim3(THREADS), lds, st, a);             \
  } while (0)
#define A2F_ATTN_BWD(P, D) do { if (a.qkv_img) A2F_ATTN_BWD_I(P, D, true); else A2F_ATTN_BWD_I(P, D, false); } while (0)
  if (precision == 1) { if (keep_q) A2F_ATTN_BWD(1, true); else A2F_ATTN_BWD(1, false); }
  else { if (keep_q) A2F_ATTN_BWD(0, true); else A2F_ATTN_BWD(0, false); }
#undef A2F_ATTN_BWD_I
#undef A2F_ATTN_BWD
  A2F_CHECK_LAUNCH("attn_bwd");
  return A2F_OK;
}

extern "C" int a2f_attn_fwd(const float* qkv, int64_t S, int64_t T, int64_t H, int64_t ld, float scale, int32_t precision, float* ctx, int64_t ldo,
                            float* lse, a2f_stream_t stream) {
  return attn_fwd_impl(qkv, S, T, H, ld, scale, precision, ctx, ldo, lse, nullptr, 1.f, stream);
}

extern "C" int a2f_attn_bwd(const float* qkv, const float* ctx, const float* dctx, const float* lse, int64_t S, int64_t T, int64_t H, int64_t ld,
                            int64_t ldo, float scale, int32_t precision, float* dqkv, float* delta_ws, a2f_stream_t stream) {
  return attn_bwd_impl(qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, precision, dqkv, delta_ws, nullptr, nullptr, 1.f, stream);
}

extern "C" int a2f_attn_fwd_drop(const float* qkv, int64_t S, int64_t T, int64_t H, int64_t ld, float scale, int32_t precision, float* ctx, int64_t ldo,
                                 float* lse, const uint32_t* keep_qmajor, float keep_scale, a2f_stream_t stream) {
  if (!keep_qmajor) return a2f::set_err(A2F_EINVAL, "attn_fwd_drop: null keep bits");
  return attn_fwd_impl(qkv, S, T, H, ld, scale, precision, ctx, ldo, lse, keep_qmajor, keep_scale, stream);
}

extern "C" int a2f_attn_bwd_drop(const float* qkv, const float* ctx, const float* dctx, const float* lse, int64_t S, int64_t T, int64_t H, int64_t ld,
                                 int64_t ldo, float scale, int32_t precision, float* dqkv, float* delta_ws, const uint32_t* keep_qmajor,
                                 const uint32_t* keep_kmajor, float keep_scale, a2f_stream_t stream) {
  if (!keep_qmajor || !keep_kmajor) return a2f::set_err(A2F_EINVAL, "attn_bwd_drop: null keep bits");
  return attn_bwd_impl(qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, precision, dqkv, delta_ws, keep_qmajor, keep_kmajor, keep_scale, stream);
}

extern "C" int a2f_attn_fwd_img(const void* qkv, int32_t qkv_format, int64_t S, int64_t T, int64_t H, int64_t ld, float scale, int32_t precision, float* ctx,
                                int64_t ldo, float* lse, const uint32_t* keep_qmajor, float keep_scale, void* ctx_img, a2f_stream_t stream) {
  return attn_fwd_impl((const float*)qkv, S, T, H, ld, scale, precision, ctx, ldo, lse, keep_qmajor, keep_scale, stream, ctx_img, qkv_format);
}

extern "C" int a2f_attn_bwd_img(const void* qkv, const float* ctx, const void* dctx, int32_t in_format, const float* lse, int64_t S, int64_t T, int64_t H,
                                int64_t ld, int64_t ldo, float scale, int32_t precision, void* dqkv, float* delta_ws, const uint32_t* keep_qmajor,
                                const uint32_t* keep_kmajor, float keep_scale, int32_t dqkv_format, a2f_stream_t stream) {
  return attn_bwd_impl((const float*)qkv, ctx, (const float*)dctx, lse, S, T, H, ld, ldo, scale, precision, (float*)dqkv, delta_ws, keep_qmajor, keep_kmajor, keep_scale,
                       stream, dqkv_format, in_format);
}
