// bf16-MFMA GEMM family for gfx950 (C ABI: a2f_mm, contract in include/a2f_hip.h).
//
// What it replaces: every dense contraction of the wav2vec2 encoder behind FaceFormer (HF Wav2Vec2Model as wrapped by
// /root/reference/src/model/wav2vec.py:91-189: conv feature encoder as implicit GEMMs, feature projection, grouped
// positional convolution, q/k/v/out projections, feed-forward layers; forward, data and weight gradients) and the
// Audio2Mesh convolutions (audio2face.py:13-47), which the reference dispatches to rocBLAS / MIOpen through torch.
//
// Shape of the kernel (one workgroup = 4 waves = one 128 x 128 (or 128 x 64) tile of C, 32-deep k-tiles):
//   * storage is fp32; a k-tile of A and B is fetched with 16-byte raw buffer loads (range-checked: out-of-range and masked
//     elements read as zero, no branches), converted to bf16 IN REGISTERS and written to LDS as bf16 images - one image for
//     precision 0, a hi and a lo image for bf16x3 (x = hi + lo, hi = RNE(x), lo = RNE(x - hi));
//   * k-contiguous operands ("R" images, [line][32 k], 64-byte rows, 16-byte chunks XOR-swizzled with (line >> 2) & 3) give
//     each v_mfma_f32_32x32x16_bf16 fragment as ONE conflict-free ds_read_b128; operands whose contiguous axis is m / n
//     ("T" images, [k][lines], pitch 2*lines + 64 bytes) are stored as they arrive (coalesced 8-byte writes) and read with
//     the transposing ds_read_b64_tr_b16 (two per fragment, conflict-free with that pitch);
//   * the loads of k-tile t+1 are in flight while tile t is multiplied (register staging, LDS double buffer, one barrier
//     per k-tile); a wave owns a 64 x 64 (32 x 64) block of C = 2 x 2 (1 x 2) accumulator tiles, so a 16-deep k-step is
//     8 (6) fragment reads for 12 (6) MFMAs in bf16x3;
//   * split-K slabs + fixed-order fold for weight gradients (K = rows of the batch), batch index for grouped convolutions.
#include "gemm_mfma.h"

namespace a2f {
namespace mm {

// ---------------------------------------------------------------------------------------------------------------------
// One operand's staging: X = number of lines (R) or contiguous extent (T) a tile spans (BM for A, BN for B).
// ---------------------------------------------------------------------------------------------------------------------
template <bool T, int X, int TH = THREADS>
struct Stage {
  static constexpr int NL = T ? (BK * X / 4) / TH : (X * 8) / TH;                // float4 units per thread
  static constexpr int PITCH = T ? (2 * X + MM_TPAD) : 64;                       // bytes
  static constexpr int IMG = T ? BK * PITCH : X * 64;                            // bytes of one bf16 image
  // R: fixed lines, the loads walk along k.  T: fixed contiguous offset, the lines (= k) advance with the k-tiles.
  // All address arithmetic is 32-bit and branch-free (selects): byte offsets are taken modulo 2^32 from the operand base
  // (the span is < 4 GiB), an element offset inside a sequence is < 2^31.
  unsigned boff[NL];       // R: byte offset of (line i, k = 0);  T: unused
  int e0[NL];              // R: element offset inside the sequence of (line i, k = 0);  T: local k-line of unit i
  unsigned okmask;         // R: bit i = line i exists;  T: bit 0 = this thread's float4 lies inside the contiguous extent
  int lds_w[NL];           // byte offset of this thread's 8-byte slot in an image
  int kq4;                 // R: k offset of this thread's float4 inside a k-tile
  int q0;                  // T: position (inside its sequence) of the current k-tile's first line
  unsigned s0bytes;        // T: byte offset of that sequence (mod 2^32)
  int jx;                  // T: contiguous element index of this thread's float4 (+ the window offset)

  __device__ __forceinline__ void init(const Op& op, int x0, int xcount, int kbeg, int t) {
    okmask = 0;
    if constexpr (!T) {
      const int kq = t & 7;
      kq4 = kq * 4;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int ll = (t >> 3) + (TH / 8) * i;
        const int line = x0 + ll;
        int s = 0, q = line;
        if (op.lps > 0) { s = line / op.lps; q = line - s * op.lps; }
        const int e = q * op.ld + op.off + kq4;
        e0[i] = e;
        boff[i] = (unsigned)s * (unsigned)(op.seq_stride * 4) + (unsigned)e * 4u;
        if (line < xcount) okmask |= (1u << i);
        lds_w[i] = ll * 64 + ((((kq >> 1) ^ ((ll >> 2) & 3))) << 4) + ((kq & 1) << 3);
      }
    } else {
      constexpr int PER_ROW = X / 4;
      const int nq = t % PER_ROW;
      jx = x0 + nq * 4;
      if (jx < xcount) okmask = 1;
      jx += op.off;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int kl = t / PER_ROW + (TH / PER_ROW) * i;
        e0[i] = kl;
        lds_w[i] = kl * PITCH + nq * 8;
      }
      q0 = kbeg;
      s0bytes = 0;
      if (op.lps > 0) {
        const int s = kbeg / op.lps;
        q0 = kbeg - s * op.lps;
        s0bytes = (unsigned)s * (unsigned)(op.seq_stride * 4);
      }
    }
  }

  // ---- plain operands (no sequence windows: lines_per_seq == 0, K a multiple of the k-tile): every per-thread offset is
  // loop-invariant.  The k-tile is selected by re-basing the buffer resource (scalar ALU: base += k0 (* ld), size -= the same;
  // size 0 past the end of the k-range), the vector offsets stay put: ZERO vector instructions per load in the steady state
  // (the windowed path spends ~25 per load on positions, sequence wraps and range masks).
  __device__ __forceinline__ void init_plain(const Op& op, int x0, int xcount, int t) {
    okmask = 0;
    if constexpr (!T) {
      const int kq = t & 7;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int ll = (t >> 3) + (TH / 8) * i;
        const int line = x0 + ll;
        boff[i] = line < xcount ? ((unsigned)line * (unsigned)op.ld + (unsigned)(kq * 4)) * 4u : 0xFFFFFFFFu;
        lds_w[i] = ll * 64 + ((((kq >> 1) ^ ((ll >> 2) & 3))) << 4) + ((kq & 1) << 3);
      }
    } else {
      constexpr int PER_ROW = X / 4;
      const int nq = t % PER_ROW;
      const int j = x0 + nq * 4;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int kl = t / PER_ROW + (TH / PER_ROW) * i;
        boff[i] = j < xcount ? ((unsigned)kl * (unsigned)op.ld + (unsigned)j) * 4u : 0xFFFFFFFFu;
        lds_w[i] = kl * PITCH + nq * 8;
      }
    }
  }
  __device__ __forceinline__ static rsrc_t tile_rsrc(const Op& op, int k0, int kend) {
    const long long adv = T ? (long long)k0 * op.ld : (long long)k0;                    // elements; wave-uniform
    const unsigned left = k0 < kend ? op.bytes - (unsigned)(adv * 4) : 0u;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.p + adv), 0, (int)left, 0x00020000);
  }
  __device__ __forceinline__ void load_plain(rsrc_t rs, u32x4_t (&reg)[NL]) const {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#ifdef MM_ABL_NOLOAD
      reg[i] = u32x4_t{boff[i], boff[i], boff[i], boff[i]};
#else
      reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, boff[i], 0, 0);
#endif
    }
  }

  // issue the loads of the k-tile starting at k0 (kend = end of this split's k-range); T operands: call advance() after it
  __device__ __forceinline__ void load(const Op& op, rsrc_t rs, int k0, int kend, u32x4_t (&reg)[NL]) {
    if constexpr (!T) {
      const bool kin = (k0 + kq4) < kend;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const unsigned e = (unsigned)(e0[i] + k0);                       // negative offsets wrap to huge values: out of range
        const bool ok = kin & (((okmask >> i) & 1u) != 0) & (e < (unsigned)op.seq_elems);
        const unsigned off = boff[i] + (unsigned)k0 * 4u;
#ifdef MM_ABL_NOLOAD      // timing ablation only (wrong results): no global traffic
        reg[i] = u32x4_t{off, off, off, off};
        (void)ok;
#else
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : 0xFFFFFFFFu, 0, 0);
#endif
      }
    } else {
      const unsigned sstep = (unsigned)(op.seq_stride * 4);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int kl = e0[i];
        int q = q0 + kl;
        unsigned sb = s0bytes;
        if (op.lps > 0) {
          if (op.lps >= BK) {                                             // at most one sequence boundary inside a k-tile
            const bool wrap = q >= op.lps;
            q = wrap ? q - op.lps : q;
            sb = wrap ? sb + sstep : sb;
          } else {                                                        // short sequences (Audio2Mesh rows of 1..16 positions)
            const int w = op.lps_shift >= 0 ? (q >> op.lps_shift) : q / op.lps;
            q -= w * op.lps;
            sb += (unsigned)w * sstep;
          }
        }
        const unsigned e = (unsigned)(q * op.ld + jx);
        const bool ok = (okmask != 0) & ((k0 + kl) < kend) & (e < (unsigned)op.seq_elems);
#ifdef MM_ABL_NOLOAD
        reg[i] = u32x4_t{e, e, e, e};
        (void)ok;
#else
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? sb + e * 4u : 0xFFFFFFFFu, 0, 0);
#endif
      }
    }
  }

  __device__ __forceinline__ void advance(const Op& op) {
    if constexpr (T) {
      q0 += BK;
      if (op.lps > 0) {
        if (op.lps >= BK) {
          const bool wrap = q0 >= op.lps;
          q0 = wrap ? q0 - op.lps : q0;
          s0bytes = wrap ? s0bytes + (unsigned)(op.seq_stride * 4) : s0bytes;
        } else {
          const int w = op.lps_shift >= 0 ? (q0 >> op.lps_shift) : q0 / op.lps;
          q0 -= w * op.lps;
          s0bytes += (unsigned)w * (unsigned)(op.seq_stride * 4);
        }
      }
    }
  }

  template <int PREC>
  __device__ __forceinline__ void commit(char* img, const u32x4_t (&reg)[NL]) const {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if constexpr (PREC == 1) {
        u32x2_t hi, lo;
        split4(reg[i], hi, lo);
        *reinterpret_cast<u32x2_t*>(img + lds_w[i]) = hi;
        *reinterpret_cast<u32x2_t*>(img + IMG + lds_w[i]) = lo;
      } else if constexpr (PREC == 2) {
        u32x2_t hi, mid, lo;
        split4x3(reg[i], hi, mid, lo);
        *reinterpret_cast<u32x2_t*>(img + lds_w[i]) = hi;
        *reinterpret_cast<u32x2_t*>(img + IMG + lds_w[i]) = mid;
        *reinterpret_cast<u32x2_t*>(img + 2 * IMG + lds_w[i]) = lo;
      } else {
        *reinterpret_cast<u32x2_t*>(img + lds_w[i]) = cvt4(reg[i]);
      }
    }
  }

  // fragment of the 32-line tile starting at local line `rb`, k16 step ks
  __device__ __forceinline__ static u32x4_t frag(const char* img, int rb, int ks, int lane) {
    if constexpr (!T) {
      const int line = rb + (lane & 31);
      const int chunk = ks * 2 + (lane >> 5);
      return *reinterpret_cast<const u32x4_t*>(img + line * 64 + ((chunk ^ ((line >> 2) & 3)) << 4));
    } else {
      const int g = lane >> 4, i = lane & 15;
      const int krow = ks * 16 + 8 * (g >> 1) + (i >> 2);
      const char* p = img + krow * PITCH + (rb + 16 * (g & 1) + 4 * (i & 3)) * 2;
      const u32x2_t r0 = tr_read(p);
      const u32x2_t r1 = tr_read(p + 4 * PITCH);
      return u32x4_t{r0.x, r0.y, r1.x, r1.y};
    }
  }
};

// CFG 0: 128 x 128 tile, 4 waves as 2 x 2, 2 x 2 accumulator tiles per wave.  CFG 1: 128 x 64 tile, waves 4 x 1, 1 x 2 tiles.
// CFG 2: 256 x 128 tile, 8 waves as 4 x 2 (512 threads, one workgroup per CU): the same two waves per SIMD as two CFG-0
// workgroups, but 384 instead of 512 operand rows are fetched / converted / written to LDS per 32 768 outputs and k-tile.
template <int CFG>
struct Cfg {
  static constexpr int BM = CFG == 2 ? 256 : 128, BN = CFG == 1 ? 64 : 128;
  static constexpr int TH = CFG >= 2 ? 512 : 256;
  static constexpr int WGN = CFG == 1 ? 1 : 2;
  static constexpr int TM = (CFG == 1 || CFG == 3) ? 1 : 2, TN = 2;
  // waves per SIMD the register budget is set for: CFG 3 = 128 x 128 tile on 8 waves (32 x 64 per wave), two workgroups per CU
  // -> four waves per SIMD (<= 128 VGPRs) to cover the LDS-read / barrier latencies of one another
  static constexpr int WPS = CFG == 3 ? 4 : (CFG == 2 ? 2 : 2);
};

// The second launch bound is WAVES PER SIMD (not workgroups per CU).  bf16x6 (three images per operand) at 128 x 128 on 8 waves: 4 = a 128-register budget and
// ONE LDS stage (48 KB) put two workgroups on a CU; with the budget of 2 the compiler took 161 registers and a CU held one workgroup whatever its LDS use
// (round 5: Audio2Mesh's four analysis convolutions 312 / 316 / 225 / 155 -> 265 / 271 / 201 / 135 us at B = 1024, tools/bench_a2m_conv.py).
template <bool AT, bool BT, int PREC, int CFG, bool PLAIN, bool ONE_STAGE = false>
__global__ __launch_bounds__(Cfg<CFG>::TH, PREC == 2 ? (CFG == 3 ? 4 : 1) : Cfg<CFG>::WPS) void gemm_mfma_kernel(const Args a) {
  constexpr int BM = Cfg<CFG>::BM, BN = Cfg<CFG>::BN, TH = Cfg<CFG>::TH;
  constexpr int WGN = Cfg<CFG>::WGN;
  constexpr int TM = Cfg<CFG>::TM, TN = Cfg<CFG>::TN;
  constexpr int NIMG = PREC == 2 ? 3 : PREC == 1 ? 2 : 1;
  using SA = Stage<AT, BM, TH>;
  using SB = Stage<BT, BN, TH>;
  constexpr int A_BYTES = NIMG * SA::IMG, B_BYTES = NIMG * SB::IMG;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];          // two stages (ONE_STAGE: one)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  // XCD-contiguous runs of the row- or panel-major tile order: tile_origin (gemm_mfma.h)
  int m0, n0;
  tile_origin(a, BM, BN, m0, n0);
  // readfirstlane: the batch / split indices come out of an integer division (vector ALU); telling the compiler they are
  // wave-uniform keeps the buffer resources in SGPRs (otherwise every buffer load is wrapped in a waterfall loop)
  const int bz = __builtin_amdgcn_readfirstlane((int)(blockIdx.z / a.split_k));
  const int sk = __builtin_amdgcn_readfirstlane((int)(blockIdx.z % a.split_k));

  int kbeg = 0, kend = a.K;
  if (a.split_k > 1) {
    const int ktiles = (a.K + BK - 1) / BK;
    const int per = (ktiles + a.split_k - 1) / a.split_k;
    kbeg = sk * per * BK;
    kend = min(a.K, kbeg + per * BK);
  }
  const int nkt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  Op opA = a.A, opB = a.B;
  // the operand bases as provably wave-uniform values: the buffer resources then live in SGPRs (no waterfall loops)
  auto uniform_ptr = [](const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
  };
  opA.p = uniform_ptr(opA.p + bz * opA.bstride);
  opB.p = uniform_ptr(opB.p + bz * opB.bstride);
  const rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opA.p), 0, opA.bytes, 0x00020000);
  const rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opB.p), 0, opB.bytes, 0x00020000);

  SA sa;
  SB sb;
  if constexpr (PLAIN) {
    sa.init_plain(opA, m0, a.M, t);
    sb.init_plain(opB, n0, a.N, t);
  } else {
    sa.init(opA, m0, a.M, kbeg, t);
    sb.init(opB, n0, a.N, kbeg, t);
  }
  // the loads of the k-tile that starts at k0 into one register set
  auto fetch = [&](int k0, u32x4_t (&qa)[SA::NL], u32x4_t (&qb)[SB::NL]) {
    if constexpr (PLAIN) {
      sa.load_plain(SA::tile_rsrc(opA, k0, kend), qa);
      sb.load_plain(SB::tile_rsrc(opB, k0, kend), qb);
    } else {
      sa.load(opA, rsA, k0, kend, qa);
      sb.load(opB, rsB, k0, kend, qb);
      sa.advance(opA);
      sb.advance(opB);
    }
  };

  f32x16 acc[TM][TN];

  // Two register sets: while k-tile kt is multiplied, tile kt+1 (fetched one iteration earlier) is converted and written
  // to the other LDS stage and the loads of tile kt+2 are issued - a fetch has two multiply phases to land (an L2 / HBM
  // round trip is longer than one 32-deep multiply phase).
  u32x4_t ra0[SA::NL], rb0[SB::NL], ra1[SA::NL], rb1[SB::NL];
  fetch(kbeg, ra0, rb0);
  fetch(kbeg + BK, ra1, rb1);                                  // out of range (zeros) when there is no second tile
  // Row sums of A beside the product (weight-gradient layout: A = dY^T, so sum_k A(m, k) is the bias gradient): every unit of a thread
  // covers the same four m, so one float4 per thread collects them as the tiles pass through the registers on their way to LDS.
  constexpr bool CAN_ROWSUM = AT && BT;                        // masked / out-of-range loads arrive as zeros on both addressing paths
  const bool do_rowsum = CAN_ROWSUM && a.a_rowsum != nullptr;  // wave-uniform
  f32x4 rsum = {0.f, 0.f, 0.f, 0.f};
  auto commit_a = [&](char* img, const u32x4_t (&q)[SA::NL]) {
    if constexpr (CAN_ROWSUM) {
      if (do_rowsum) {
#pragma unroll
        for (int i = 0; i < SA::NL; ++i) rsum += __builtin_bit_cast(f32x4, q[i]);
      }
    }
    sa.template commit<PREC>(img, q);
  };
  auto compute = [&](const char* cur) {
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
      if constexpr (PREC == 2) {           // bf16x6: the six products of order <= 2^-16, smallest first
        constexpr int PA[5] = {2, 0, 1, 1, 0}, PB[5] = {0, 2, 1, 0, 1};
#pragma unroll
        for (int p = 0; p < 5; ++p)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma(fa[PA[p]][i], fb[PB[p]][j], acc[i][j]);
      }
      if constexpr (PREC == 1) {           // small terms first; consecutive MFMAs hit different accumulators
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
  };
  // one steady-state iteration: tile kt is multiplied, `cur` registers (tile kt+1) go to LDS, `nxt` registers receive tile kt+2
  auto body = [&](int kt, u32x4_t (&ca)[SA::NL], u32x4_t (&cb)[SB::NL], u32x4_t (&na)[SA::NL], u32x4_t (&nb)[SB::NL]) {
    char* nst = smem + ((kt + 1) & 1) * STAGE;
    const int k2 = kbeg + (kt + 2) * BK;
    fetch(k2, na, nb);                                          // `nxt` held tile kt, which went to LDS one iteration ago
    if constexpr (ONE_STAGE) {
      // ONE LDS stage (bf16x6: three images per operand are 48 KB per stage at 128 x 128 - two stages leave room for one workgroup per CU): tile kt is
      // multiplied, then - behind a barrier - tile kt + 1 overwrites it.  The conversion no longer overlaps this workgroup's own MFMAs; the CU's other
      // workgroup's MFMAs run under it instead, and so do its prologue and epilogue (short-K products run 7-23 k-tiles per tile of C).
      __builtin_amdgcn_sched_barrier(0);
      compute(smem);
      __syncthreads();
      commit_a(smem, ca);
      sb.template commit<PREC>(smem + A_BYTES, cb);
      __syncthreads();
      return;
    }
    // keep the load issue HERE: without the fence the machine scheduler sinks the buffer loads below the MFMA block, next
    // to their first use, and the fetch no longer overlaps the multiplication.  Below it the conversion / LDS writes of
    // tile kt+1 (other stage) and the MFMAs of tile kt are one scheduling region: the compiler interleaves VALU and MFMA.
    __builtin_amdgcn_sched_barrier(0);
#ifdef MM_ABL_NOCOMMIT      // timing ablation only: loaded registers stay live, no conversion, no LDS writes
#pragma unroll
    for (int i = 0; i < SA::NL; ++i) asm volatile("" ::"v"(ca[i]));
#pragma unroll
    for (int i = 0; i < SB::NL; ++i) asm volatile("" ::"v"(cb[i]));
#else
    commit_a(nst, ca);
    sb.template commit<PREC>(nst + A_BYTES, cb);
#endif
    compute(smem + (kt & 1) * STAGE);
    __syncthreads();
  };
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  commit_a(smem, ra0);
  sb.template commit<PREC>(smem + A_BYTES, rb0);
  __syncthreads();
  int kt = 0;
  for (; kt + 2 < nkt; kt += 2) {
    body(kt, ra1, rb1, ra0, rb0);
    body(kt + 1, ra0, rb0, ra1, rb1);
  }
  if (kt + 1 < nkt) {
    body(kt, ra1, rb1, ra0, rb0);
    ++kt;
  }
  if (nkt > 0) compute(smem + (ONE_STAGE ? 0 : ((nkt - 1) & 1) * STAGE));

  if constexpr (CAN_ROWSUM) {
    if (do_rowsum && n0 == 0) {                                // one column of workgroups owns the sums (block-uniform)
      constexpr int PER_ROW = BM / 4, GROUPS = TH / PER_ROW;   // threads t, t + PER_ROW, ... hold partial sums of the same four rows
      __syncthreads();                                         // every wave is done with the operand images
      float* red = reinterpret_cast<float*>(smem);
      *reinterpret_cast<f32x4*>(red + (t / PER_ROW) * BM + (t % PER_ROW) * 4) = rsum;
      __syncthreads();
      if (t < BM && m0 + t < a.M) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) v += red[g * BM + t];
        if (a.split_k > 1) a.ws[(long long)a.batch * a.split_k * a.M * a.N + (long long)sk * a.M + m0 + t] = v;     // slabs behind the C slabs
        else a.a_rowsum[m0 + t] = v;
      }
    }
  }

  tile_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, lane, bz);
}

// The weight-gradient fold (plain store, batch 1, N a multiple of 4: every split-K product of a Linear / convolution weight gradient): one float4 of C per
// thread, EIGHT slabs in flight per trip (the scalar form below ran 45 us per call on 66 MB of slabs - a quarter of the HBM rate - 60 calls per FaceFormer
// step), same fixed order of additions per element as the scalar kernel: ((s0 + s1) + (s2 + s3)) per group of four, groups in order.
__global__ __launch_bounds__(256) void gemm_mfma_fold4_kernel(const Args a) {
  const long long mn4 = (long long)a.M * a.N / 4;
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i < mn4) {
    const f32x4* w = reinterpret_cast<const f32x4*>(a.ws) + i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= a.split_k; s += 8) {
      f32x4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(w + (long long)(s + u) * mn4);
      v += (x[0] + x[1]) + (x[2] + x[3]);
      v += (x[4] + x[5]) + (x[6] + x[7]);
    }
    for (; s + 4 <= a.split_k; s += 4) {
      const f32x4 x0 = w[(long long)s * mn4], x1 = w[(long long)(s + 1) * mn4], x2 = w[(long long)(s + 2) * mn4], x3 = w[(long long)(s + 3) * mn4];
      v += (x0 + x1) + (x2 + x3);
    }
    for (; s < a.split_k; ++s) v += w[(long long)s * mn4];
    const long long e = i * 4;
    const int m = (int)(e / a.N), n = (int)(e - (long long)m * a.N);
    *reinterpret_cast<f32x4*>(a.C + (long long)m * a.ldc + n) = v * a.alpha;
  }
  if (a.a_rowsum) {
    const float* rs = a.ws + (long long)a.split_k * a.M * a.N;
    for (long long m = blockIdx.x * 256ll + threadIdx.x; m < a.M; m += (long long)gridDim.x * 256) {
      float v = 0.f;
      int s = 0;
      for (; s + 4 <= a.split_k; s += 4)
        v += (rs[(long long)s * a.M + m] + rs[(long long)(s + 1) * a.M + m]) + (rs[(long long)(s + 2) * a.M + m] + rs[(long long)(s + 3) * a.M + m]);
      for (; s < a.split_k; ++s) v += rs[(long long)s * a.M + m];
      a.a_rowsum[m] = v;
    }
  }
}

__global__ __launch_bounds__(256) void gemm_mfma_fold_kernel(const Args a) {
  const long long total = (long long)a.batch * a.M * a.N;
  const long long mn = (long long)a.M * a.N;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int bz = (int)(idx / mn);
    const long long r = idx - bz * mn;
    const int m = (int)(r / a.N), n = (int)(r - (long long)m * a.N);
    // four slabs per trip, their loads in flight together (one load - wait - add per slab was up to 64 dependent L2 round trips per element); the order of the
    // additions is fixed: ((s0 + s1) + (s2 + s3)) per trip, trips in order
    const float* w = a.ws + (long long)bz * a.split_k * mn + r;
    float v = 0.f;
    int s = 0;
    for (; s + 4 <= a.split_k; s += 4) {
      const float x0 = w[(long long)s * mn], x1 = w[(long long)(s + 1) * mn], x2 = w[(long long)(s + 2) * mn], x3 = w[(long long)(s + 3) * mn];
      v += (x0 + x1) + (x2 + x3);
    }
    for (; s < a.split_k; ++s) v += w[(long long)s * mn];
    epilogue_store(a, bz, m, n, c_row_offset(a, m), v);
  }
  if (a.a_rowsum) {                                            // the row-sum slabs of the split product, same fixed order
    const float* rs = a.ws + (long long)a.batch * a.split_k * mn;
    for (long long m = blockIdx.x * 256ll + threadIdx.x; m < a.M; m += (long long)gridDim.x * 256) {
      float v = 0.f;
      int s = 0;
      for (; s + 4 <= a.split_k; s += 4)
        v += (rs[(long long)s * a.M + m] + rs[(long long)(s + 1) * a.M + m]) + (rs[(long long)(s + 2) * a.M + m] + rs[(long long)(s + 3) * a.M + m]);
      for (; s < a.split_k; ++s) v += rs[(long long)s * a.M + m];
      a.a_rowsum[m] = v;
    }
  }
}

template <bool AT, bool BT, int PREC, int CFG, bool PLAIN>
static int launch(const Args& a, hipStream_t st) {
  constexpr int BM = Cfg<CFG>::BM, BN = Cfg<CFG>::BN, TH = Cfg<CFG>::TH;
  constexpr int NIMG = PREC == 2 ? 3 : PREC == 1 ? 2 : 1;
  constexpr bool ONE_STAGE = PREC == 2 && CFG == 3;         // bf16x6 at 128 x 128: one 48-KB stage, two workgroups per CU (see the kernel's body())
  constexpr int smem = (ONE_STAGE ? 1 : 2) * NIMG * (Stage<AT, BM, TH>::IMG + Stage<BT, BN, TH>::IMG);
  auto kern = gemm_mfma_kernel<AT, BT, PREC, CFG, PLAIN, ONE_STAGE>;
  static bool attr_done = false;                        // set-once per instantiation (idempotent; safe to repeat)
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return set_err(A2F_ELAUNCH, "a2f_mm: cannot reserve %d bytes of LDS", smem);
    attr_done = true;
  }
  dim3 grid((unsigned)(cdiv(a.N, BN) * cdiv(a.M, BM)), 1, (unsigned)(a.batch * a.split_k));
  hipLaunchKernelGGL(kern, grid, dim3(TH), smem, st, a);
  return A2F_OK;
}

template <int PREC, int CFG>
static int dispatch_layout(const Args& a, bool at, bool bt, bool plain, hipStream_t st) {
  // plain operands (the Transformer layers' Linear forward / data gradient / weight gradient): the loop-invariant addressing path, built
  // for the tile shapes those layouts run with
  if constexpr (CFG == 3) {
    if (plain && !at && !bt) return launch<false, false, PREC, CFG, true>(a, st);
    if (plain && !at && bt) return launch<false, true, PREC, CFG, true>(a, st);
  }
  if constexpr (CFG == 0 || CFG == 2) {
    if (plain && at && bt) return launch<true, true, PREC, CFG, true>(a, st);
  }
  if (!at && !bt) return launch<false, false, PREC, CFG, false>(a, st);
  if (!at && bt) return launch<false, true, PREC, CFG, false>(a, st);
  if (at && bt) return launch<true, true, PREC, CFG, false>(a, st);
  return launch<true, false, PREC, CFG, false>(a, st);
}

static int fill_op(const a2f_mm_operand& o, int64_t bstride, int64_t contiguous_extent, Op& k, const char* name) {
  A2F_CHECK_ARG(o.p != nullptr, "a2f_mm: operand %s is null", name);
  A2F_CHECK_ARG((o.format == 0 || o.format == 1) && o.p_lo == nullptr, "a2f_mm: operand %s: format %d (0 = fp32, 1 = split-32 image; p_lo is reserved and must be null)", name, o.format);
  A2F_CHECK_ARG(o.format == 0 || (o.ld % 32 == 0 && o.off % 32 == 0 && o.seq_stride % 32 == 0 && ((uintptr_t)o.p & 127) == 0),
                "a2f_mm: image operand %s: ld, off and seq_stride must be multiples of 32 elements and the base 128-byte aligned", name);
  A2F_CHECK_ARG(((uintptr_t)o.p & 15) == 0, "a2f_mm: operand %s must be 16-byte aligned", name);
  A2F_CHECK_ARG(o.bytes > 0 && o.bytes < (int64_t)0xFFFFFFF0ll, "a2f_mm: operand %s spans %lld bytes (must be < 4 GiB)", name, (long long)o.bytes);
  A2F_CHECK_ARG(o.ld % 4 == 0 && o.off % 4 == 0 && o.seq_stride % 4 == 0 && bstride % 4 == 0 && contiguous_extent % 4 == 0,
                "a2f_mm: operand %s: ld, off, seq_stride, batch stride and the contiguous extent must be multiples of 4 elements", name);
  A2F_CHECK_ARG(o.lines_per_seq >= 0 && o.lines_per_seq < (1ll << 31) && o.ld < (1ll << 31) && llabs(o.off) < (1ll << 31),
                "a2f_mm: operand %s: window parameters out of range", name);
  k.p = o.p;
  k.bytes = (unsigned)o.bytes;
  k.ld = (int)o.ld;
  k.lps = (int)o.lines_per_seq;
  k.lps_shift = -1;
  if (k.lps > 0 && (k.lps & (k.lps - 1)) == 0) {
    k.lps_shift = 0;
    while ((1 << k.lps_shift) < k.lps) ++k.lps_shift;
  }
  k.off = o.lines_per_seq > 0 ? (int)o.off : 0;
  k.seq_elems = o.lines_per_seq > 0 ? (int)(o.seq_elems < 0x7FFFFFF0ll ? o.seq_elems : 0x7FFFFFF0ll) : 0x7FFFFFF0;
  k.seq_stride = o.lines_per_seq > 0 ? o.seq_stride : 0;
  k.bstride = bstride;
  return A2F_OK;
}

int launch_img(const Args& a, bool tn, hipStream_t st);      // csrc/gemm_img.hip

static void launch_fold(const Args& a, hipStream_t st) {
  const bool plain = a.batch == 1 && !a.bias && !a.act && !a.aux_mode && !a.keep && !a.accumulate && !a.C2 && a.c_rps <= 0 && a.N % 4 == 0 && a.ldc % 4 == 0 &&
                     (((uintptr_t)a.C | (uintptr_t)a.ws) & 15) == 0;
  if (plain) {
    const long long mn4 = (long long)a.M * a.N / 4;
    hipLaunchKernelGGL(gemm_mfma_fold4_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, st, a);
    return;
  }
  const long long total = (long long)a.batch * a.M * a.N;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gemm_mfma_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
}

}  // namespace mm
}  // namespace a2f

extern "C" int a2f_mm(const a2f_mm_desc* d, a2f_stream_t stream) {
  using namespace a2f;
  using namespace a2f::mm;
  A2F_CHECK_ARG(d != nullptr, "a2f_mm: null descriptor");
  A2F_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->M < (1ll << 31) && d->N < (1ll << 31) && d->K < (1ll << 31),
                "a2f_mm: M=%lld N=%lld K=%lld", (long long)d->M, (long long)d->N, (long long)d->K);
  A2F_CHECK_ARG(d->C != nullptr, "a2f_mm: C is null");
  A2F_CHECK_ARG(d->precision >= 0 && d->precision <= 2, "a2f_mm: precision %d (0 = bf16, 1 = bf16x3, 2 = bf16x6)", d->precision);
  A2F_CHECK_ARG(d->act == 0 || d->act == 1 || d->act == 3 || d->act == 4, "a2f_mm: act %d", d->act);
  A2F_CHECK_ARG(d->aux_mode >= 0 && d->aux_mode <= 4 && (d->aux_mode == 0 || d->aux != nullptr), "a2f_mm: aux_mode %d / aux", d->aux_mode);
  const int batch = d->batch > 0 ? d->batch : 1;
  const int split = d->split_k > 1 ? d->split_k : 1;
  A2F_CHECK_ARG(split == 1 || d->ws != nullptr, "a2f_mm: split_k needs a workspace");
  A2F_CHECK_ARG((int64_t)batch * split <= 65535, "a2f_mm: batch * split_k = %lld exceeds the grid limit", (long long)batch * split);
  Args a;
  int rc = fill_op(d->A, d->a_bstride, d->A.trans ? d->M : d->K, a.A, "A");
  if (rc) return rc;
  rc = fill_op(d->B, d->b_bstride, d->B.trans ? d->N : d->K, a.B, "B");
  if (rc) return rc;
  a.C = d->C;
  a.C2 = d->C2;
  a.bias = d->bias;
  a.aux = d->aux;
  a.ws = d->ws;
  a.ldc = d->ldc;
  a.c_seq_stride = d->c_seq_stride;
  a.c_bstride = d->c_bstride;
  a.bias_bstride = d->bias_bstride;
  a.ldaux = d->ldaux;
  a.aux_bstride = d->aux_bstride;
  a.c_rps = (int)d->c_rows_per_seq;
  a.M = (int)d->M;
  a.N = (int)d->N;
  a.K = (int)d->K;
  a.act = d->act;
  a.aux_mode = d->aux_mode;
  a.accumulate = d->accumulate;
  a.split_k = split;
  a.batch = batch;
  a.alpha = d->alpha == 0.f ? 1.f : d->alpha;
  A2F_CHECK_ARG(d->keep_bits == nullptr || (batch == 1 && d->keep_wpr * 32 >= d->N), "a2f_mm: keep_bits need batch == 1 and keep_wpr * 32 >= N");
  a.a_rowsum = d->a_rowsum;
  A2F_CHECK_ARG(d->a_rowsum == nullptr || (batch == 1 && d->A.trans && d->B.trans), "a2f_mm: a_rowsum needs the weight-gradient layout (both operands trans), batch 1");
  a.keep = d->keep_bits;
  a.keep_wpr = d->keep_wpr;
  a.keep_scale = d->keep_scale;
  const bool at = d->A.trans != 0, bt = d->B.trans != 0;
  a.c_format = d->c_format;
  a.c_colsum = d->c_colsum;
  A2F_CHECK_ARG(d->c_colsum == nullptr || (d->A.format == 1 && !at && d->ws != nullptr), "a2f_mm: c_colsum needs image operands (k-contiguous) and a workspace");
  A2F_CHECK_ARG(d->A.format == d->B.format, "a2f_mm: both operands must have the same format (%d vs %d)", d->A.format, d->B.format);
  A2F_CHECK_ARG(d->c_format == 0 || (d->c_format == 1 && d->A.format == 1), "a2f_mm: c_format %d (1 = image output, image-operand products only)", d->c_format);
  if (d->A.format == 1) {
    // operands arrive as split-32 images: the 256 x 256 LDS-DMA kernels of csrc/gemm_img.hip (bf16x3 only)
    A2F_CHECK_ARG(d->precision == 1 && batch == 1 && !d->accumulate && at == bt && d->a_rowsum == nullptr,
                  "a2f_mm (image operands): precision 1, batch 1, no accumulate, no a_rowsum, both operands with the same `trans`");
    A2F_CHECK_ARG(d->aux_mode == 0 || d->aux_mode == 1 || d->aux_mode == 4, "a2f_mm (image operands): aux_mode %d", d->aux_mode);
    A2F_CHECK_ARG(d->N % 4 == 0 && d->ldc % 4 == 0 && d->ldaux % 4 == 0 && d->c_seq_stride % 4 == 0 && (((uintptr_t)d->C | (uintptr_t)d->C2 | (uintptr_t)d->aux | (uintptr_t)d->bias) & 15) == 0,
                  "a2f_mm (image operands): N, ldc, ldaux multiples of 4 and 16-byte aligned C / C2 / aux / bias (float4 epilogue)");
    const long long lim = 0x80000000ll / 4;
    const long long M1 = d->M - 1;
    const long long c_max = (d->c_rows_per_seq > 0 ? (M1 / d->c_rows_per_seq) * d->c_seq_stride + (M1 % d->c_rows_per_seq) * d->ldc : M1 * d->ldc) + d->N;
    A2F_CHECK_ARG(c_max < lim && d->M * d->N < lim && M1 * d->ldaux + d->N < lim && d->M * d->keep_wpr < lim, "a2f_mm (image operands): output spans must stay below 2 GiB");
    if (!at) {
      A2F_CHECK_ARG(d->K % 32 == 0 && split == 1, "a2f_mm (image operands, k-contiguous): K %% 32 == 0, no split-K");
      A2F_CHECK_ARG(a.B.lps == 0, "a2f_mm (image operands): B must be a plain matrix");
      A2F_CHECK_ARG(a.A.lps == 0 || llabs(d->A.off) + d->A.lines_per_seq * d->A.ld + d->K < (1ll << 30), "a2f_mm (image operands): window offsets must stay below 2^30 elements");
    } else {
      A2F_CHECK_ARG(a.A.lps == 0 && d->c_format == 0 && d->c_rows_per_seq == 0, "a2f_mm (image operands, weight-gradient layout): plain A, fp32 C");
      A2F_CHECK_ARG(a.B.lps == 0 || (d->B.off >= 0 && d->B.seq_stride >= 0), "a2f_mm (image operands, weight-gradient layout): B windows need off >= 0 and seq_stride >= 0");
      A2F_CHECK_ARG(d->M % 32 == 0 && d->N % 32 == 0, "a2f_mm (image operands, weight-gradient layout): M, N multiples of 32");
    }
    A2F_CHECK_ARG(d->c_format == 0 || (d->ldc % 32 == 0 && d->c_seq_stride % 32 == 0 && d->N % 8 == 0 && ((uintptr_t)d->C & 127) == 0),
                  "a2f_mm: image output needs ldc and c_seq_stride %% 32 == 0, N %% 8 == 0 and a 128-byte aligned C");
    a.epi32 = 1;
    a.group_n = 0;
    {
      const long long b_bytes = (long long)d->N * d->K * 4, a_bytes = (long long)d->M * d->K * 4;
      const long long tiles_n = cdiv(d->N, 256), tiles_m = cdiv(d->M, 256);
      static const int forced = [] { const char* e = getenv("A2F_MM_GROUP_N"); return e ? atoi(e) : -1; }();
      if (forced >= 0) a.group_n = forced;
      else if (!at && b_bytes > (5ll << 19) && tiles_n >= 4) {
        long long g = (2ll << 20) / (256 * d->K * 4);
        g = g < 1 ? 1 : (g > tiles_n ? tiles_n : g);
        const long long rows_live = 32 / tiles_n > 0 ? 32 / tiles_n : 1;
        const long long row_major = a_bytes + b_bytes * cdiv(tiles_m, rows_live);
        const long long panel = a_bytes * cdiv(tiles_n, g) + 8 * b_bytes;
        if (panel < row_major) a.group_n = (int)g;
      }
      if (a.group_n >= tiles_n) a.group_n = 0;
    }
    rc = launch_img(a, at, (hipStream_t)stream);
    if (rc) return rc;
    A2F_CHECK_LAUNCH("a2f_mm (image operands)");
    if (split > 1) {
      launch_fold(a, (hipStream_t)stream);
      A2F_CHECK_LAUNCH("a2f_mm fold");
    }
    return A2F_OK;
  }
  // 32-bit epilogue offsets (gemm_mfma.h tile_epilogue_band): every element offset of C / C2 / like-C aux, of the (m, ldaux) aux, of the keep words and
  // of one split-K slab, in bytes from its (per-batch) base, below 2^31
  {
    static const bool off32 = [] { const char* e = getenv("A2F_MM_EPI32"); return !(e && e[0] == '0'); }();       // TEST HOOK: 0 forces the 64-bit pointer form (the >= 2 GiB fallback) on small operands
    const long long lim = 0x80000000ll / 4;
    const long long M1 = d->M - 1;
    const long long c_max = (d->c_rows_per_seq > 0 ? (M1 / d->c_rows_per_seq) * d->c_seq_stride + (M1 % d->c_rows_per_seq) * d->ldc : M1 * d->ldc) + d->N;
    bool ok = off32 && c_max < lim && d->ldc >= 0 && d->c_seq_stride >= 0 && d->M * d->N < lim;
    if (d->aux_mode >= 1 && d->aux_mode <= 3) ok = ok && d->ldaux >= 0 && M1 * d->ldaux + d->N < lim;
    if (d->keep_bits) ok = ok && d->M * d->keep_wpr < lim;
    a.epi32 = ok ? 1 : 0;
  }
  // tile order (tile_origin): panel-major when the B operand does not fit an XCD's L2 and panels re-read A less than row-major re-reads B
  a.group_n = 0;
  {
    static const int forced = [] { const char* e = getenv("A2F_MM_GROUP_N"); return e ? atoi(e) : -1; }();      // TEST HOOK: 0 = always row-major (the two orders must give the same bits)
    const long long bn = d->N <= 64 ? 64 : 128;
    const long long b_bytes = (long long)d->N * d->K * 4, a_bytes = (long long)d->M * d->K * 4;
    const long long tiles_n = cdiv(d->N, bn), tiles_m = cdiv(d->M, 128);
    if (forced >= 0) a.group_n = forced;
    else if (batch == 1 && split == 1 && b_bytes > (5ll << 19) && tiles_n >= 4) {                                 // B beyond 2.5 MB
      long long g = (2ll << 20) / (bn * d->K * 4);                                                                 // a panel's slice of B: <= 2 MB
      g = g < 1 ? 1 : (g > tiles_n ? tiles_n : g);
      const long long rows_live = 64 / tiles_n > 0 ? 64 / tiles_n : 1;                                             // m-tiles an XCD holds at a time (64 workgroups)
      const long long row_major = a_bytes + b_bytes * cdiv(tiles_m, rows_live);
      const long long panel = a_bytes * cdiv(tiles_n, g) + 8 * b_bytes;
      if (panel < row_major) a.group_n = (int)g;
    }
    if (a.group_n >= tiles_n) a.group_n = 0;
  }
  // tile choice (measured, tools/bench_mm.py): narrow outputs -> 128 x 64; weight-gradient layout (both operands k-strided: 80 KB of
  // LDS per 128 x 128 workgroup, so only one 8-wave workgroup would fit) -> 256 x 128 on 8 waves when there are enough tiles, else
  // 128 x 128 on 4 waves; everything else -> 128 x 128 on 8 waves, two workgroups per CU = four waves per SIMD (+10-25 % over 4 waves)
  int cfg = d->N <= 64 ? 1 : 3;
  // (bf16x6: the 8-wave 128 x 128 form too, on one LDS stage and a 128-register budget = two workgroups per CU: see gemm_mfma_kernel)
  // (a weight gradient of at most 128 output rows - Audio2Mesh's analysis layer 1, 108 x 216 over 524 288 rows - keeps the 8-wave 128 x 128 form: half of a 256-row
  // tile would be empty and the 4-wave form leaves one wave per SIMD: 261 (256 x 128) / 232 (4 waves) / 198 us (8 waves, 128 x 128), same box)
  if (cfg == 3 && at && bt && d->precision != 2 && d->M > 128) cfg = (cdiv(d->M, 256) * cdiv(d->N, 128) * batch * split >= 200) ? 2 : 0;
  hipStream_t st = (hipStream_t)stream;
  const bool plain = a.A.lps == 0 && a.B.lps == 0 && (d->K % BK) == 0;
  if (d->precision == 2) {                 // three images per operand: the 256 x 128 tile does not fit
    if (cfg == 2) cfg = 3;
    rc = cfg == 0 ? dispatch_layout<2, 0>(a, at, bt, plain, st) : cfg == 1 ? dispatch_layout<2, 1>(a, at, bt, plain, st) : dispatch_layout<2, 3>(a, at, bt, plain, st);
  } else if (d->precision == 1)
    rc = cfg == 0 ? dispatch_layout<1, 0>(a, at, bt, plain, st) : cfg == 1 ? dispatch_layout<1, 1>(a, at, bt, plain, st)
       : cfg == 2 ? dispatch_layout<1, 2>(a, at, bt, plain, st) : dispatch_layout<1, 3>(a, at, bt, plain, st);
  else
    rc = cfg == 0 ? dispatch_layout<0, 0>(a, at, bt, plain, st) : cfg == 1 ? dispatch_layout<0, 1>(a, at, bt, plain, st)
       : cfg == 2 ? dispatch_layout<0, 2>(a, at, bt, plain, st) : dispatch_layout<0, 3>(a, at, bt, plain, st);
  if (rc) return rc;
  A2F_CHECK_LAUNCH("a2f_mm");
  if (split > 1) {
    launch_fold(a, st);
    A2F_CHECK_LAUNCH("a2f_mm fold");
  }
  return A2F_OK;
}
