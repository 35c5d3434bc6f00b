// Input packing / weight re-layout kernels (tiny, HBM-trivial).
//
// The strided temporal convolutions run as implicit GEMMs over CHANNELS-LAST activations
// act[b][h][c]: the im2col row of output position (b,ho) is then the contiguous window
// act[b][2ho-1 .. 2ho+1][:] (3C floats), so no im2col buffer exists.  These kernels build the first
// channels-last activation from the caller's tensors and reproduce the reference's one-hot tiling
// quirks (voca.py:40-45, audio2face.py:59-62).
#include "a2f_common.h"

namespace a2f {

// x (B,29,16), one_hot (B,no) -> a0 (B,16,37): a0[b,h,c<29] = x[b,c,h]; a0[b,h,29+r] = one_hot[b, h%8]
// (embedding = one_hot[:, :8].repeat(1,16).view(bs,1,8,16): flat index r*16+h of the repeated vector -> (r*16+h)%8 = h%8)
__global__ void voca_pack_kernel(const float* __restrict__ x, const float* __restrict__ one_hot, int64_t B, int no,
                                 float* __restrict__ a0, float* __restrict__ cat72) {
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;                 // 256 threads
  __shared__ float xs[29 * 16];
  for (int i = t; i < 29 * 16; i += blockDim.x) xs[i] = x[b * 29 * 16 + i];
  __syncthreads();
  for (int i = t; i < 16 * 37; i += blockDim.x) {
    const int h = i / 37, c = i % 37;
    a0[b * 16 * 37 + i] = c < 29 ? xs[c * 16 + h] : one_hot[b * no + (h & 7)];
  }
  if (t < 8) cat72[b * 72 + 64 + t] = one_hot[b * no + t];
}

// x (B,52,32), one_hot (B,no) -> a0 (B,64,32) [row h][col w] (NCHW with C=1 == (B,H=64,W=32)):
// rows 0..51 = x; row 52+r, col c = one_hot[b, (32r+c) % no]     (audio2face.py:59-62)
__global__ void a2m_pack_kernel(const float* __restrict__ x, const float* __restrict__ one_hot, int64_t B, int no,
                                float* __restrict__ a0, float* __restrict__ cat, int64_t cat_ld, int64_t cat_off) {
  const int64_t b = blockIdx.x;
  const int t = threadIdx.x;
  const int rows = 52 + no;
  for (int i = t; i < rows * 32; i += blockDim.x) {
    const int h = i / 32, c = i % 32;
    a0[b * rows * 32 + i] = h < 52 ? x[b * 52 * 32 + i] : one_hot[b * no + ((32 * (h - 52) + c) % no)];
  }
  if (t < no) cat[b * cat_ld + cat_off + t] = one_hot[b * no + t];
}

// dir 0: w (Co,Ci,KH) -> wg (Co, KH*Ci) with k = kh*Ci + ci ; dir 1: the inverse
__global__ void conv_weight_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t total, int Ci, int KH, int dir) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int per = Ci * KH;
  const int64_t co = i / per;
  const int r = (int)(i % per);
  if (dir == 0) {            // i indexes dst (wg): r = kh*Ci + ci
    const int kh = r / Ci, ci = r % Ci;
    dst[i] = src[co * per + ci * KH + kh];
  } else {                   // i indexes dst (w): r = ci*KH + kh
    const int ci = r / KH, kh = r % KH;
    dst[i] = src[co * per + kh * Ci + ci];
  }
}

constexpr int RL_MAX = 32, RL_BLOCKS = 64;
struct RelayoutBatch { a2f_relayout_item it[RL_MAX]; };

// grid (RL_BLOCKS, n): the workgroups of column y walk item y's DESTINATION elements
__global__ __launch_bounds__(256) void relayout_batch_kernel(const RelayoutBatch b) {
  const a2f_relayout_item& it = b.it[blockIdx.y];
  const int Co = it.Co, Ci = it.Ci, KH = it.KH, Cop = it.Cop, Cip = it.Cip;
  const int kp = KH * Cip;
  const int64_t total = it.mode == 2 ? (int64_t)Co * Ci * KH : (int64_t)Cop * kp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)RL_BLOCKS * 256) {
    if (it.mode == 2) {                                     // i = (co * Ci + ci) * KH + kh
      const int kh = (int)(i % KH);
      const int64_t r = i / KH;
      const int ci = (int)(r % Ci), co = (int)(r / Ci);
      it.dst[i] = it.src[(int64_t)co * kp + kh * Cip + ci];
    } else {
      int co, k;
      if (it.mode == 0) { co = (int)(i / kp); k = (int)(i - (int64_t)co * kp); }
      else { k = (int)(i / Cop); co = (int)(i - (int64_t)k * Cop); }
      const int kh = k / Cip, ci = k - kh * Cip;
      it.dst[i] = (co < Co && ci < Ci) ? it.src[((int64_t)co * Ci + ci) * KH + kh] : it.pad_value;
    }
  }
}

}  // namespace a2f

extern "C" int a2f_relayout_batch(const a2f_relayout_item* items, int32_t n, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(items && n >= 1 && n <= RL_MAX, "relayout_batch: n=%d (1 .. %d items)", (int)n, RL_MAX);
  RelayoutBatch b;
  for (int i = 0; i < n; ++i) {
    const a2f_relayout_item& it = items[i];
    A2F_CHECK_ARG(it.src && it.dst && it.src != it.dst, "relayout_batch: item %d: null or aliased pointers", i);
    A2F_CHECK_ARG(it.Co >= 1 && it.Ci >= 1 && it.KH >= 1 && it.Cop >= it.Co && it.Cip >= it.Ci && it.mode >= 0 && it.mode <= 2 &&
                  (int64_t)it.Cop * it.KH * it.Cip < (1ll << 31), "relayout_batch: item %d: Co=%d Ci=%d KH=%d Cop=%d Cip=%d mode=%d", i, it.Co, it.Ci, it.KH, it.Cop, it.Cip, it.mode);
    b.it[i] = it;
  }
  for (int i = n; i < RL_MAX; ++i) b.it[i] = items[0];
  hipLaunchKernelGGL(relayout_batch_kernel, dim3(RL_BLOCKS, (unsigned)n), dim3(256), 0, (hipStream_t)stream, b);
  A2F_CHECK_LAUNCH("relayout_batch");
  return A2F_OK;
}

extern "C" int a2f_voca_pack(const float* x, const float* one_hot, int64_t B, int64_t n_onehot, float* a0, float* cat72,
                             a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(x && one_hot && a0 && cat72, "voca_pack: null pointer");
  A2F_CHECK_ARG(B > 0 && B < (1ll << 31), "voca_pack: bad B");
  A2F_CHECK_ARG(n_onehot >= 8, "voca_pack: one_hot needs >= 8 columns (voca.py:40 slices [:, :8])");
  hipLaunchKernelGGL(voca_pack_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, one_hot, B, (int)n_onehot, a0, cat72);
  A2F_CHECK_LAUNCH("voca_pack");
  return A2F_OK;
}

extern "C" int a2f_a2m_pack(const float* x, const float* one_hot, int64_t B, int64_t n_onehot, float* a0, float* cat,
                            int64_t cat_ld, int64_t cat_off, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(x && one_hot && a0 && cat, "a2m_pack: null pointer");
  A2F_CHECK_ARG(B > 0 && B < (1ll << 31), "a2m_pack: bad B");
  A2F_CHECK_ARG(n_onehot >= 1 && n_onehot <= 256, "a2m_pack: bad n_onehot");
  hipLaunchKernelGGL(a2m_pack_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, one_hot, B, (int)n_onehot, a0, cat, cat_ld, cat_off);
  A2F_CHECK_LAUNCH("a2m_pack");
  return A2F_OK;
}

extern "C" int a2f_conv_weight_pack(const float* src, float* dst, int64_t Co, int64_t Ci, int64_t KH, int32_t dir, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(src && dst, "conv_weight_pack: null pointer");
  A2F_CHECK_ARG(Co > 0 && Ci > 0 && KH > 0 && (dir == 0 || dir == 1), "conv_weight_pack: bad arguments");
  const int64_t total = Co * Ci * KH;
  hipLaunchKernelGGL(conv_weight_pack_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, total, (int)Ci, (int)KH, (int)dir);
  A2F_CHECK_LAUNCH("conv_weight_pack");
  return A2F_OK;
}
