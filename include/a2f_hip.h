/*
 * a2f_hip.h - C ABI of liba2f_hip.so: the MI355X (gfx950) hot path of the audio -> face-mesh
 * regression models (VOCA / Audio2Mesh / FaceFormer decoder + 15069-wide vertex head + per-vertex
 * L2/velocity loss + Adam), forward and backward.
 *
 * The reference (xtliu97/audio2face-pytorch) is pure Python on torch and has no FFI of its own; each
 * entry point below names the reference code (file:line under /root/reference) whose arithmetic it
 * replaces.  The drop-in boundary a reference maintainer sees is the Python nn.Module API
 * (src/model/lightning_model.py:50-73,96,111-117,155); this C ABI is what those modules call through
 * ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch tensor .data_ptr()), fp32 unless noted,
 *    dense row-major unless a stride is given; nothing is retained after the call returns;
 *  - work is enqueued asynchronously on `stream` (a hipStream_t, passed as void*); no entry point synchronises or
 *    allocates device memory (hipGraph-capturable).  Process-wide state the library does keep, all of it idempotent or
 *    opt-in: one "LDS attribute set" flag per kernel instantiation (hipFuncSetAttribute is issued once), the head-kernel
 *    shape selected by the environment variable A2F_HEAD_CFG (read once), the tile override A2F_MM_CFG (read per call), and
 *    the measurement ring of a2f_profile_enable (off by default).  Nothing else survives a call;
 *  - return value: 0 on success, negative A2F_E* on failure; a2f_last_error() gives a thread-local message;
 *  - `ws` is caller-provided scratch of at least the matching *_ws_bytes(); contents are undefined on entry.
 */
#ifndef A2F_HIP_H
#define A2F_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A2F_VERSION 100          /* 0.1.0 */
#define A2F_OK 0
#define A2F_EINVAL (-1)          /* bad shape / null pointer / unsupported combination */
#define A2F_ELAUNCH (-2)         /* hipLaunch / runtime error */
#define A2F_EWORKSPACE (-3)      /* workspace too small */

typedef void* a2f_stream_t;      /* hipStream_t */

int a2f_version(void);
const char* a2f_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Generic fused fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain):
 *     C[m,n] = epi( sum_k A(m,k) * B(k,n) )          m<M, n<N, k<K
 * A(m,k) = A[a_off + m*a_rs + k*a_cs],  B(k,n) = B[b_off + k*b_rs + n*b_cs]  (element strides; either
 * may alias overlapping windows of an activation buffer, which is how the (3,1)/(1,3) strided
 * convolutions of voca.py:19-28 / audio2face.py:13-47 run as implicit GEMMs without im2col).
 * Window mask (zero padding of those convolutions): with mask_operand = 1 (A) elements with
 * (m % mask_period == mask_phase) and mask_lo <= k < mask_hi read as 0; mask_operand = 2 (B) does the
 * same on (k % mask_period == mask_phase), mask_lo <= n < mask_hi.
 * epi: v = acc; if bias: v += bias[n]; if accumulate: v += C_old; act (0 none, 1 relu, 2 tanh);
 *      aux_mode 1: v *= (aux[m*ldaux+n] > 0)      (ReLU backward)
 *      aux_mode 2: v *= 1 - aux^2                 (tanh backward)
 *      aux_mode 3: v += aux                       (template add, voca.py:49)
 * ones_out != NULL appends a virtual all-ones column n == N to B and routes it to ones_out[m]
 * (bias gradients as part of the weight-gradient GEMM).
 * split_k > 1 partitions K over grid.z.
 *   splitk_ws == NULL: partial tiles are added with fp32 atomics into a C (and ones_out) the caller has zeroed; only
 *   valid with act = aux_mode = 0, bias = NULL, accumulate = 0;
 *   splitk_ws != NULL: (>= split_k*M*(N+1) floats) every split stores its partial tile there and a second
 *   kernel folds the splits in fixed order and applies the full epilogue into C / ones_out: deterministic, free of the
 *   atomic contention that dominates when the output is tiny (weight gradients), and valid with every epilogue - also
 *   the way to spread a GEMM with few 128-row tiles and a long K over the whole chip.
 * Replaces torch.nn.Linear / Conv2d forward+backward as dispatched by the reference models.
 */
typedef struct a2f_gemm_desc {
  const float* A; int64_t a_off, a_rs, a_cs;
  const float* B; int64_t b_off, b_rs, b_cs;
  float* C; int64_t ldc;
  int64_t M, N, K;
  const float* bias;
  const float* aux; int64_t ldaux;
  float* ones_out;
  int32_t act, aux_mode, accumulate, split_k;
  int32_t mask_operand, mask_period, mask_phase, mask_lo, mask_hi;
  int32_t c_unpack_kh;   /* a2f_wgrad_* only: > 0 stores column kh*Ci+ci (Ci = N/c_unpack_kh) at ci*KH+kh, i.e. C in nn.Conv2d's (Co,Ci,KH,1) layout */
  float* splitk_ws;
  int32_t precision;     /* a2f_gemm_f32 only: 0 = exact fp32 MFMA (default); 1 = bf16x3 split products (x = hi + lo in bf16; three bf16 MFMAs, fp32 accumulate; <= 2^-16 per product); 2 = bf16x6 (x = hi + mid + lo; six bf16 MFMAs; <= ~2^-23 per product, fp32-class) */
  int32_t reserved;
} a2f_gemm_desc;

int a2f_gemm_f32(const a2f_gemm_desc* d, a2f_stream_t stream);

/* Weight-gradient specialisation of the same contract for tiny outputs and huge contractions (dW = dY^T X of the
 * Linear / implicit-GEMM convolution layers, voca.py:19-36): M <= 128, N+1 <= 224 (a2f_wgrad_supported), A stored (K,M)
 * with a_rs == 1, B stored (K,N) with b_cs == 1, optional B window mask and ones column, linear epilogue, C (and
 * ones_out) OVERWRITTEN.  One register tile set per wave, operands read from global memory in MFMA layout, per-wave
 * partial slabs in d->splitk_ws (>= a2f_wgrad_ws_floats(M,N,K) floats) folded in fixed order (deterministic).
 * a2f_wgrad_batch_f32 runs up to A2F_WGRAD_BATCH_MAX independent problems (e.g. all seven trunk layers of a VOCA step)
 * in one launch pair; each descriptor carries its own workspace.  Operands must span less than 2 GB each.
 * d->precision: 0 = exact fp32 MFMA; 1 = bf16x3 split products (x = hi + lo, lo*hi' + hi*lo' + hi*hi' on the bf16 matrix pipe,
 * fp32 accumulation, <= 2^-16 relative error per product; 16-row register sets, 96-column sub-problems); other values = 0. */
#define A2F_WGRAD_BATCH_MAX 8
int a2f_wgrad_supported(int64_t M, int64_t N);
size_t a2f_wgrad_ws_floats(int64_t M, int64_t N, int64_t K);
int a2f_wgrad_f32(const a2f_gemm_desc* d, a2f_stream_t stream);
int a2f_wgrad_batch_f32(const a2f_gemm_desc* descs, int32_t count, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * VocaLoss (src/loss/loss.py:24-55); FaceFormerLoss (loss.py:4-17) is the same call on the
 * even-length prefix of the (T, n) frame matrix.
 *   rec = mean_{b,v} sum_xyz (p-g)^2 ; vel = same on (row 2k+1 - row 2k) differences ; loss = k_rec*rec + k_vel*vel
 * pred, gt: (B, n) with n = 3*n_verts, B even. out3 = {loss, rec, vel}.
 * fwd is deterministic (fixed-order two-stage reduction through ws).
 * bwd writes grad_pred (B, n) = g_loss*dloss/dpred + g_rec*drec/dpred + g_vel*dvel/dpred, upstream
 * scalars read from the DEVICE array g3 = {g_loss, g_rec, g_vel}.
 */
size_t a2f_vocaloss_ws_bytes(int64_t B, int64_t n);
int a2f_vocaloss_fwd(const float* pred, const float* gt, int64_t B, int64_t n, float k_rec, float k_vel,
                     float* out3, void* ws, size_t ws_bytes, a2f_stream_t stream);
int a2f_vocaloss_bwd(const float* pred, const float* gt, const float* g3, int64_t B, int64_t n,
                     float k_rec, float k_vel, float* grad_pred, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused vertex head + loss + head backward in ONE pass over the (B, n) frame matrix
 * (decoder.4 / output_net.4 / vertice_map_r: voca.py:35,48-49, audio2face.py:54,65-66,
 *  faceformer.py:129,181,187; loss.py:24-55):
 *   pred[b,:] = H[b,:] Wt + bias + template[b or 0,:]              (written once)
 *   partial sums of rec / vel                                      -> out3 = {loss, rec, vel}
 *   G = dloss/dpred (never stored);  dH = G W ; dW = G^T H ; dbias = colsum G
 * H (B,K) K<=64; W (n,K) torch Linear layout; template (template_rows, n): template_rows == 1 (one row for all frames),
 * == B (one row per frame) or a divisor S of B - frame b takes row b / (B / S): the S sequences of a FaceFormer step
 * (batch = 1 each in the reference, train.py:33) in one launch, each with its own subject template;
 * gt (B,n).  B must be even.  dH (B,K), dW (n,K), dbias (n) are overwritten.
 * `Bloss` frames (even, <= B) enter the loss (FaceFormerLoss drops an odd tail frame, loss.py:13-15);
 * all B rows of pred are written.  pred may be NULL on the default (x3) kernel: no prediction is written - a training step that only
 * reads the loss (the reference's training_step never reads pred: mse_error = rec_loss / 3); 122 456 instead of 182 732 algorithmic bytes per frame.
 */
size_t a2f_head_fused_ws_bytes(int64_t B, int64_t n, int64_t K);
int a2f_head_fused(const float* H, const float* W, const float* bias, const float* templ, int64_t template_rows,
                   const float* gt, int64_t B, int64_t Bloss, int64_t n, int64_t K, float k_rec, float k_vel,
                   float* pred, float* out3, float* dH, float* dW, float* dbias,
                   void* ws, size_t ws_bytes, a2f_stream_t stream);

/* Measurement hook (off by default; the only mutable global state of the library, and only while enabled): records
 * HIP events on the launch stream right around the main kernel of a2f_head_fused (not its slab folds) for up to 64
 * launches; a2f_profile_head_ms returns their mean duration and resets the ring (synchronise the device first). */
int a2f_profile_enable(int32_t on);
int a2f_profile_head_ms(float* mean_ms, int32_t* launches);

/* ------------------------------------------------------------------------------------------------
 * VOCA input packing (voca.py:40-45): x (B,29,16), one_hot (B,n_onehot>=8) ->
 *   a0 (B,16,37) channels-last window [h][c]: c<29: x[b,c,h]; c>=29: one_hot[b, h%8]  (repeat(1,16).view quirk)
 *   cat72[:,64:72] = one_hot[:, :8]                                (voca.py:47; cat72 is (B,72))
 * Audio2Mesh input packing (audio2face.py:59-62): x (B,52,32), one_hot (B,12) ->
 *   a0 (B,64,32) rows 0..51 = x, row 52+r col c = one_hot[b,(32r+c)%12]; cat268[:,256:268] = one_hot
 */
int a2f_voca_pack(const float* x, const float* one_hot, int64_t B, int64_t n_onehot, float* a0, float* cat72,
                  a2f_stream_t stream);
int a2f_a2m_pack(const float* x, const float* one_hot, int64_t B, int64_t n_onehot, float* a0, float* cat,
                 int64_t cat_ld, int64_t cat_off, a2f_stream_t stream);

/* Fused VOCA trunk forward (voca.py:40-48 up to the 50-wide bottleneck) for all B frames in one launch: activations of a
 * 32-frame block stay in LDS across the four temporal convolutions and the three Linear layers; every activation is
 * also stored once, channels-last, in exactly the layout the backward entry points read:
 *   a0 (B,16,37) a1 (B,8,32) a2 (B,4,32) a3 (B,2,64) cat72 (B,72) z1 (B,72) z2 (B,128, post-tanh) h3 (B,50).
 * conv_w[i]: time_conv.{0,2,4,6}.weight (Co,Ci,3,1) in the reference layout, conv_b[i] its bias; fc_w/fc_b: decoder.{0,1,3}.
 * The four pointer arrays are HOST arrays of device pointers.
 * Arithmetic: the four convolutions in exact fp32 (their pre-activations decide ReLU masks); the three Linear layers, and every product of
 * a2f_voca_trunk_bwd_data, in split bf16 (x = hi + lo; lo*hi' + hi*lo' + hi*hi' on the bf16 matrix instructions, fp32 accumulation:
 * <= 2^-16 relative per product, the same device as the head and a2f_mm's precision 1).  The environment variable A2F_TRUNK_EXACT=1
 * (read per call by both entry points) selects the exact fp32 matrix instructions everywhere. */
int a2f_voca_trunk_fwd(const float* x, const float* one_hot, int64_t B, int64_t n_onehot, const float* const* conv_w,
                       const float* const* conv_b, const float* const* fc_w, const float* const* fc_b, float* a0, float* a1,
                       float* a2, float* a3, float* cat72, float* z1, float* z2, float* h3, a2f_stream_t stream);

/* The same trunk fed from the per-frame feature tensor instead of materialised windows (the window
 * semantics are those of the reference's data side, /root/reference/src/dataset/vocaset.py:408-430 zero-padded slices,
 * restated in a2f_amd.data.feature_windows): feats (n_seq, seq_len, 29); frame t of a sequence sees feature rows
 * t-8 .. t+7 of ITS sequence, rows outside [0, seq_len) read as zeros.  B = n_seq * seq_len frames, one_hot (B, n_onehot)
 * and every output as in a2f_voca_trunk_fwd (a0 holds the windows, so the backward entry points are unchanged).
 * seq_len >= 32.  HBM reads of the input drop from 1 856 to 116 bytes per frame. */
int a2f_voca_trunk_fwd_seq(const float* feats, int64_t n_seq, int64_t seq_len, const float* one_hot, int64_t n_onehot,
                           const float* const* conv_w, const float* const* conv_b, const float* const* fc_w,
                           const float* const* fc_b, float* a0, float* a1, float* a2, float* a3, float* cat72, float* z1,
                           float* z2, float* h3, a2f_stream_t stream);

/* Fused data-gradient chain of the same trunk: d(h3) (B,50) -> pre-activation gradients of every layer, one launch:
 *   g_z2 (B,128) = d(decoder.1 pre-tanh), g_z1 (B,72) = d(decoder.0 output), g_c4 (B,64), g_c3 (B,2,64), g_c2 (B,4,32),
 *   g_c1 (B,8,32) = d(conv pre-ReLU outputs).  Reads the forward's saved activations for the tanh' / ReLU masks and the
 *   weights in the reference layouts.  The weight gradients are then a2f_wgrad_f32 contractions of these with the saved
 *   activations (dW3 = dh3^T z2, dW1 = g_z2^T z1, dW0 = g_z1^T cat, conv l: g_cl^T windows(a_{l-1})). */
int a2f_voca_trunk_bwd_data(const float* dh3, const float* z2, const float* cat72, const float* a3, const float* a2,
                            const float* a1, const float* fw3, const float* fw1, const float* fw0, const float* cw4,
                            const float* cw3, const float* cw2, int64_t B, float* g_z2, float* g_z1, float* g_c4,
                            float* g_c3, float* g_c2, float* g_c1, a2f_stream_t stream);

/* Conv weight re-layout  w (Co,Ci,KH) <-> wg (Co, KH*Ci) [k = kh*Ci+ci]; dir 0: w->wg, 1: wg->w(+=0 overwrite) */
int a2f_conv_weight_pack(const float* src, float* dst, int64_t Co, int64_t Ci, int64_t KH, int32_t dir, a2f_stream_t stream);

/* A batch of such re-layouts in ONE launch (Audio2Mesh refreshes nine packed convolution weights, their transposes for the data-gradient products and eight
 * padded BatchNorm vectors every step, and unpacks nine weight gradients: ~60 launches of ~5 us as separate copies).  The logical tensor is w (Co, Ci, KH) in the
 * reference's parameter layout (a vector: Ci = KH = 1); the packed side has padded extents Cop >= Co, Cip >= Ci:
 *   mode 0: dst = wg  (Cop, KH*Cip), wg[co][kh*Cip + ci] = w[co][ci][kh], every padded element = pad_value (the whole of dst is written)
 *   mode 1: dst = wgT (KH*Cip, Cop), the transpose of mode 0's result (both operands of a data-gradient product k-contiguous), padded elements = pad_value
 *   mode 2: dst = w   (Co, Ci, KH) from src = wg (Cop, KH*Cip): the gradient's way back (padding dropped)
 * items: HOST array of n <= 32 entries (copied into the kernel arguments); src / dst of an entry must not overlap. */
typedef struct a2f_relayout_item {
  const float* src;
  float* dst;
  int32_t Co, Ci, KH, Cop, Cip, mode;
  float pad_value;
  int32_t reserved;
} a2f_relayout_item;
int a2f_relayout_batch(const a2f_relayout_item* items, int32_t n, a2f_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d over channels-last rows (audio2face.py:15-46), training-mode batch statistics:
 *   x (R, C) rows = (b,h,w) positions.  stats: mean[c], invstd[c] (biased var, eps); running buffers
 *   updated with momentum and the unbiased variance exactly like nn.BatchNorm2d.
 *   y = (x-mean)*invstd*gamma + beta, optional ReLU.
 * eval mode: pass training = 0, statistics come from running_mean / running_var.
 */
size_t a2f_bn_ws_bytes(int64_t R, int64_t C);
int a2f_bn_fwd(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta,
               float* running_mean, float* running_var, float momentum, float eps, int32_t training, int32_t relu,
               float* y, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, a2f_stream_t stream);
/* dy is the gradient w.r.t. the (post-ReLU if relu) output.  The ReLU mask is re-derived from x (y > 0 <=> the forward's own
 * fma(x, invstd gamma, beta - mean invstd gamma) > 0), so the saved output is not read: pass beta (needed when relu != 0). */
int a2f_bn_bwd(const float* x, const float* beta, const float* dy, int64_t R, int64_t C, const float* gamma,
               const float* save_mean, const float* save_invstd, int32_t training, int32_t relu,
               float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, a2f_stream_t stream);
/* The same two entries for the BatchNorm that follows Audio2Mesh's analysis layer 0 (audio2face.py:13-17), with the convolution FUSED IN: x is never stored -
 * every pass re-evaluates x(r, c) = cbias[c] + sum_kh a0[2r - 1 + kh] wg[c][kh] (Conv2d(1, C, (1,3), stride (1,2), padding (0,1)); a0[2r - 1] = 0 at the first
 * position of a group; exact fp32, bit for bit the same in all four passes) from
 * the one-channel input a0, so the forward moves R*C*4 bytes (the output) instead of 4x that and the backward 3x instead of 5x.  a0: R / P_out groups of 2 P_out
 * values; C a multiple of 4 (<= 256); cbias, gamma, beta, the saved statistics and the (R, C) tensors 16-byte aligned. */
int a2f_bn_fwd_conv0(const float* a0, const float* wg, const float* cbias, int64_t P_out, int64_t R, int64_t C, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int32_t training, int32_t relu,
                     float* y, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, a2f_stream_t stream);
/* a2f_bn_bwd_conv0 can also produce the convolution's OWN parameter gradients in the same pass: dwg (C, 3) = sum_r dx(r, c) a0[2r - 1 + kh] and dcbias (C) = sum_r
 * dx(r, c) (both NULL: not computed).  Layer 0 has no data gradient, so with dwg given dx may be NULL - the (R, C) pre-activation gradient is then never written
 * (302 MB at B = 1024) and no separate weight-gradient pass reads it back.  Sums: fp32 per thread over its rows, double across workgroups, fixed order. */
int a2f_bn_bwd_conv0(const float* a0, const float* wg, const float* cbias, int64_t P_out, const float* beta, const float* dy, int64_t R, int64_t C,
                     const float* gamma, const float* save_mean, const float* save_invstd, int32_t training, int32_t relu,
                     float* dx, float* dgamma, float* dbeta, float* dwg, float* dcbias, void* ws, size_t ws_bytes, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * FaceFormer decoder recurrence (faceformer.py:154-185 evaluated incrementally):
 * one workgroup per sequence, S sequences of T frames each.  Inputs per sequence s:
 *   cross (S,T,64)  = out_proj(v_proj(memory)) of the diagonal cross-attention (identity 3)
 *   style (S,64)    = obj_vector(one_hot)
 * params: packed decoder parameter block (layout in a2f_ff_param_offsets), A (64x64) and c (64): collapsed feedback.
 * Outputs: y (S,T,64) decoder rows; saved activations for backward in `save` (a2f_ff_save_bytes).
 */
#define A2F_FF_D 64
#define A2F_FF_HEADS 4
#define A2F_FF_FF 128
/* float offsets inside the packed parameter block */
enum {
  A2F_FF_WQKV = 0,                       /* (192,64) self_attn.in_proj_weight */
  A2F_FF_BQKV = A2F_FF_WQKV + 192 * 64,  /* (192) */
  A2F_FF_WO = A2F_FF_BQKV + 192,         /* (64,64) self_attn.out_proj.weight */
  A2F_FF_BO = A2F_FF_WO + 64 * 64,       /* (64) */
  A2F_FF_W1 = A2F_FF_BO + 64,            /* (128,64) linear1 */
  A2F_FF_B1 = A2F_FF_W1 + 128 * 64,      /* (128) */
  A2F_FF_W2 = A2F_FF_B1 + 128,           /* (64,128) linear2 */
  A2F_FF_B2 = A2F_FF_W2 + 64 * 128,      /* (64) */
  A2F_FF_LN = A2F_FF_B2 + 64,            /* norm1.w, norm1.b, norm2.w, norm2.b, norm3.w, norm3.b (6 x 64) */
  A2F_FF_A = A2F_FF_LN + 6 * 64,         /* (64,64) collapsed feedback A = Wm Wr */
  A2F_FF_C = A2F_FF_A + 64 * 64,         /* (64)    c = Wm br + bm */
  A2F_FF_PARAM_FLOATS = A2F_FF_C + 64
};
size_t a2f_ff_save_bytes(int64_t S, int64_t T);
/* pe: (period, 64) periodic positional-encoding table (faceformer.py:74-80), computed by the caller exactly as the
 * reference builds it. */
int a2f_ff_decoder_fwd(const float* params, const float* pe, const float* cross, const float* style, int64_t S, int64_t T,
                       int32_t period, float* y, void* save, size_t save_bytes, a2f_stream_t stream);
/* dy (S,T,64) -> dparams (A2F_FF_PARAM_FLOATS, OVERWRITTEN), dcross (S,T,64), dstyle (S,64).  Deterministic, no atomics: the
 * reverse-time recurrence kernel leaves 640 gradient factors per frame and per-sequence LayerNorm-gradient slabs behind the
 * per-step rows of `save`; weight and bias gradients are then five bf16x3 products over all S*T frames launched on the same
 * stream by this call (split-K slabs folded in fixed order), the slabs are summed in sequence order.  `save` is the forward's
 * buffer (a2f_ff_save_bytes covers records, slabs, the products' workspace and - last - the forward sweep's own K / V planes,
 * 128 floats per frame laid out [sequence][K | V][head][4 planes][T][4]); its dK/dV accumulator columns are scratch
 * for this call.  From 256 frames on the recurrence is launched in blocks of 128 steps; attention terms across block
 * boundaries are computed by two chip-wide kernels (see ff_decoder.hip).  Environment A2F_FF_BWD (A-B hook, read per call):
 * 1 = the single-kernel form with register accumulators, 2 = record form in one launch. */
int a2f_ff_decoder_bwd(const float* params, const float* dy, int64_t S, int64_t T, int32_t period, void* save,
                       size_t save_bytes, float* dparams, float* dcross, float* dstyle, a2f_stream_t stream);
/* Train-mode variants: dropout 0.1 of the PPE output (faceformer.py:73,88) and of nn.TransformerDecoderLayer's self-attention
 * probabilities, dropout1, feed-forward activation dropout and dropout3, with ONE mask per frame row (the reference redraws the
 * masks of all prefix rows at every loop step, which has no incremental evaluation; per-row masks keep its per-step distribution
 * and make its own loop causal-consistent - fixture G14 pins this variant on the reference run with per-row masks injected).
 * Keep bits (a2f_keep_bits layout): row_bits (S*T, 10 words): columns 0..63 PPE output, 64..127 dropout1, 128..255 activation,
 * 256..319 dropout3; attn_bits (S*4*T, a2f_attn_keep_words(T) words): row (s*4 + head)*T + frame, bit j = key j.  Cross-attention
 * dropout / dropout2 act on the `cross` rows before the call.  keep_scale = 1 / (1 - p). */
int a2f_ff_decoder_fwd_drop(const float* params, const float* pe, const float* cross, const float* style, int64_t S, int64_t T,
                            int32_t period, float* y, void* save, size_t save_bytes, const uint32_t* row_bits, const uint32_t* attn_bits,
                            float keep_scale, a2f_stream_t stream);
int a2f_ff_decoder_bwd_drop(const float* params, const float* dy, int64_t S, int64_t T, int32_t period, void* save, size_t save_bytes,
                            float* dparams, float* dcross, float* dstyle, const uint32_t* row_bits, const uint32_t* attn_bits,
                            float keep_scale, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Linear interpolation along time (align_corners = True) of channels-last feature rows: x (N, L_in, C) -> y (N, L_out, C),
 * C a multiple of 4, 16-byte aligned buffers.  Replaces F.interpolate(features.transpose(1, 2), size=L_out,
 * align_corners=True, mode="linear").transpose(1, 2) of /root/reference/src/model/wav2vec.py:76-84 (the 49 Hz -> frame
 * rate resampling of the wav2vec2 conv features; same index arithmetic as torch: scale = (L_in-1)/(L_out-1) in fp32).
 * _bwd: dy (N, L_out, C) -> dx (N, L_in, C), gather form, deterministic. */
int a2f_interp_linear_fwd(const float* x, int64_t N, int64_t L_in, int64_t C, int64_t L_out, float* y, a2f_stream_t stream);
int a2f_interp_linear_bwd(const float* dy, int64_t N, int64_t L_in, int64_t C, int64_t L_out, float* dx, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused Adam step on flat fp32 buffers (torch.optim.Adam semantics as configured at
 * lightning_model.py:209-213: L2-style weight decay added to the gradient, bias correction, eps outside sqrt):
 *   g = grad*grad_scale + wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 */
int a2f_adam_step(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float grad_scale, int32_t step, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement helpers (no model arithmetic): float4 streaming copy (1 read + 1 write) and triad dst = a + b (2 reads +
 * 1 write, the access mix of the fused head kernel) over n fp32 elements (n % 4 == 0, 16-byte aligned buffers).
 * bench.py times them with HIP events to report the HBM rate the box gives plain streaming kernels next to the
 * nominal 8 TB/s. */
int a2f_stream_copy(float* dst, const float* src, int64_t n, a2f_stream_t stream);
int a2f_stream_triad(float* dst, const float* a, const float* b, int64_t n, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * bf16-MFMA GEMM family (csrc/gemm_mfma.hip): the matrix products of the wav2vec2 encoder that FaceFormer trains through
 * (HF Wav2Vec2Model as wrapped by /root/reference/src/model/wav2vec.py:91-189: the 7-layer convolutional feature encoder
 * as implicit GEMMs, feature projection, grouped positional convolution, the q/k/v/out projections and the feed-forward
 * layers of the 12 Transformer layers - forward, data gradients and weight gradients) and of the Audio2Mesh convolutions
 * (audio2face.py:13-47).
 *     C[m,n] = epi( alpha * sum_k A(m,k) * B(k,n) )
 * Storage is fp32; operands are converted to bf16 while they are staged into LDS and multiplied with
 * v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  precision 1 (bf16x3): x = hi + lo, lo*hi' + hi*lo' + hi*hi', <= 2^-16
 * relative error per product - the setting the 1e-4 parity bar is met with; precision 0: plain bf16 operands (the
 * reference's `percision: 16-mixed`); precision 2 (bf16x6): x = hi + mid + lo, the six products down to 2^-16 of the leading
 * one - fp32-GEMM class (2e-6 on the result), three LDS images per operand: the forward convolutions of the Audio2Mesh trunk, whose
 * rounding reaches the batch statistics of the next BatchNorm (bf16x3 there puts the weight gradients 1e-2 off the reference).
 * Operands (a2f_mm_operand): `trans` = 0: a line is a row of A (column of B) and k is the contiguous axis; `trans` = 1: a
 * line is one k and the m (n) index is contiguous.  Line i of an operand with lines_per_seq > 0 belongs to sequence
 * s = i / lines_per_seq, position q = i % lines_per_seq; element j of the line sits at element offset
 * e = q*ld + j + off inside the sequence and reads as ZERO unless 0 <= e < seq_elems; its address is
 * p + s*seq_stride + e.  lines_per_seq = 0: plain matrix (address p + i*ld + j).  This is how strided convolutions over
 * channels-last (sequence, position, channel) activations run as implicit GEMMs with no im2col buffer: overlapping windows
 * are lines with ld < K, zero padding and sequence ends are the bounds test.  `bytes` = addressable span from p (< 4 GiB):
 * anything beyond reads as zero.  All of p, ld, off, seq_stride and the contiguous extent must be multiples of 4 elements
 * (16-byte vector loads).
 * `format` 1 (both operands; `p_lo` stays null): the operand arrives as a SPLIT-32 IMAGE of the fp32 matrix it stands for - same shape, same 4 bytes
 * per element, same `ld` / `off` / `bytes` units: a row is ld / 32 blocks of 128 bytes, block b = the 32 elements 32 b .. 32 b + 31 as 32 bf16 `hi`
 * (RNE(x)) followed by 32 bf16 `lo` (RNE(x - hi)); ld, off, seq_stride multiples of 32, base 128-byte aligned.  Producers write it directly
 * (a2f_mm with c_format 1, a2f_layernorm_fwd / _bwd image outputs, the attention kernels, a2f_split_img for everything else), so no consumer converts:
 * such products run on the 256 x 256 LDS-DMA kernels of csrc/gemm_img.hip (bf16x3 only: precision 1, batch 1, no accumulate / a_rowsum; both operands
 * k-contiguous - plain or un-padded windows of A, K % 32 == 0, no split-K - or both line-contiguous = the weight-gradient layout with split-K, M and N
 * multiples of 32; aux_mode 0 / 1 / 4; N, ldc, ldaux multiples of 4; GELU there is the Abramowitz-Stegun 7.1.26 form, |error| < 1e-7).
 * c_format 1 (k-contiguous image products only): C itself is written as a split-32 image (ldc % 32 == 0); C2 stays fp32.
 * c_colsum (k-contiguous image products only): the column sums of the values stored to C - when C is the input gradient of a Linear this is its bias
 * gradient - from per-(m-tile, wave-row) partial rows in `ws`, folded in fixed order (deterministic).
 * Tile order: workgroups walk the (m-tile, n-tile) grid row-major, or - when B (N x K x 4 bytes) outgrows an XCD's 4 MB L2 - in panels of n-tiles
 * whose slice of B stays resident while the A panels stream past (same results bit for bit; A2F_MM_GROUP_N=0 forces row-major).
 * Epilogue, per element: v = alpha*acc; if bias: v += bias[n]; C2 (optional) = v;  act: 0 none, 1 ReLU, 3 GELU (erf form), 4 GELU with
 * C2 = gelu'(v) instead of v (the derivative the backward multiplies by: no erf / exp pass there);
 * keep_bits (optional, batch == 1): v = bit (m, n) of the keep mask ? v * keep_scale : 0 (train-mode dropout of the layer's output,
 * a2f_keep_bits layout with keep_wpr words per row);
 * aux_mode 1: v += aux[m*ldaux+n]; 2: v *= gelu'(aux); 3: v *= (aux > 0); 4: v *= aux at C's own index (aux laid out like C, row map
 * included: a saved derivative); accumulate: v += C_old.  Row m of C (and C2) is
 * stored at C + (m / c_rows_per_seq)*c_seq_stride + (m % c_rows_per_seq)*ldc + n  (c_rows_per_seq = 0: C + m*ldc + n).
 * a_rowsum (optional; both operands `trans`, batch 1): a_rowsum[m] = sum_k A(m,k), collected from the A tiles as
 * they pass through the registers - with A = dY^T (weight gradient of a Linear) this is the bias gradient, at no extra pass over dY; with
 * split_k > 1 `ws` needs split_k*M more floats.
 * split_k > 1: the contraction is cut into split_k slabs stored in `ws` (>= batch*split_k*M*N floats) and folded in fixed
 * order by a second kernel that applies the epilogue (deterministic; how weight gradients with K = 10^4 rows fill the chip).
 * batch > 1: independent problems at p + b*{a,b,c,bias}_bstride, aux + b*aux_bstride (the 16 groups of the positional
 * convolution; C2 follows C).
 */
typedef struct a2f_mm_operand {
  const float* p;
  int64_t bytes;
  int64_t ld;
  int32_t trans, format;          /* format: 0 = fp32; 1 = split-32 image (see below) */
  int64_t lines_per_seq, seq_stride, off, seq_elems;
  const void* p_lo;               /* reserved: must be null */
} a2f_mm_operand;

typedef struct a2f_mm_desc {
  a2f_mm_operand A, B;
  float* C; int64_t ldc, c_rows_per_seq, c_seq_stride;
  float* C2;
  int64_t M, N, K;
  const float* bias;
  const float* aux; int64_t ldaux, aux_bstride;
  int32_t act, aux_mode, precision, split_k;
  float* ws;
  int32_t batch, accumulate;
  int64_t a_bstride, b_bstride, c_bstride, bias_bstride;
  float alpha; int32_t c_format;                    /* 0: C is fp32; 1: C is a split-32 image (image-operand products only) */
  const uint32_t* keep_bits; int64_t keep_wpr;     /* optional dropout of C: see a2f_keep_bits */
  float keep_scale; int32_t reserved2;
  float* a_rowsum;                                 /* optional: a_rowsum[m] = sum_k A(m,k), see below */
  float* c_colsum;                                 /* optional (image operands, k-contiguous): c_colsum[n] = sum_m of the values stored to C; ws >= 2 ceil(M/256) N floats */
} a2f_mm_desc;

int a2f_mm(const a2f_mm_desc* d, a2f_stream_t stream);

/* fp32 (R, C) rows at stride ld -> split-32 image (a2f_mm_operand.format 1) with row stride ldo elements (ldo % 32 == 0, ldo >= C);
 * transpose != 0: the image of the TRANSPOSE, (C, R) rows at stride ldo >= R (R % 4 == 0): how data gradients read a weight k-contiguously.
 * A streaming pass for operands no producer emits as images (weights once per step; activations in tests).  C % 4 == 0, ld % 4 == 0. */
int a2f_split_img(const float* x, int64_t R, int64_t C, int64_t ld, int32_t transpose, void* out, int64_t ldo, a2f_stream_t stream);

/* The grouped positional convolution of the wav2vec2 encoder as kernels of its own (csrc/posconv.hip; replaces what HF's
 * Wav2Vec2PositionalConvEmbedding - nn.Conv1d(768, 768, kernel_size 128, padding 64, groups 16) behind /root/reference/src/model/wav2vec.py:91-189 -
 * hands to MIOpen).  Activations are channels-last fp32 (S, T, C), C = groups * 48, groups = 16, taps = 128 (anything else: A2F_EINVAL); bf16x3 products.
 *   a2f_posconv_nt:    y[s,t,g*48+co] = [aux +] act(bias + sum_{kk<taps, ci<48} x[s, t+off+kk, g*48+ci] * W[g*48+co][kk*48+ci]), frames outside [0, T) read as zero.
 *                      w_img = the split-32 image (a2f_split_img) of W as a (C, taps*48) matrix.  act: 0 none, 1 exact GELU; z (optional) receives the
 *                      pre-activation (bias added).  Forward: off = -taps/2, bias, act 1, aux = x.  Data gradient: x = dz, W = flipped / transposed weights
 *                      [g*48+ci][kk'*48+co] = w[g*48+co][ci][taps-1-kk'], off = -(taps/2 - 1), aux = dy.
 *   a2f_posconv_wgrad: dw[g][kk][ci][co] = sum_{s,t} x[s, t-pad+kk, g*48+ci] * dz[s, t, g*48+co]   (G, taps, 48, 48), one pass, fixed summation order. */
int a2f_posconv_nt(const float* x, const void* w_img, const float* bias, const float* aux, float* y, float* z, int64_t S, int64_t T, int64_t C,
                   int64_t groups, int64_t taps, int64_t off, int32_t act, a2f_stream_t stream);
int a2f_posconv_wgrad(const float* x, const float* dz, float* dw, int64_t S, int64_t T, int64_t C, int64_t groups, int64_t taps, int64_t pad,
                      a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row-wise kernels of the wav2vec2 encoder path (csrc/rowops.hip; HF Wav2Vec2Model as wrapped by
 * /root/reference/src/model/wav2vec.py:91-189: nn.LayerNorm of feature projection / encoder / the 12 post-norm layers,
 * bias gradients, GELU derivative).  Rows of C floats, C % 4 == 0, C <= 1024, 16-byte aligned buffers.
 *   layernorm_fwd: y = (x - mean) * rstd * gamma + beta (biased variance, eps inside the sqrt); saves mean / rstd per row.
 *   layernorm_bwd: dx, dgamma = sum dy * xhat, dbeta = sum dy, and (optional, may be NULL) dx_colsum = column sums of dx
 *                  (= the bias gradient of the Linear whose output was normalised); deterministic two-stage reductions.
 *   colsum:        out[c] = sum_r x[r*ld + c]  (bias gradients), deterministic.
 *   gelu_bwd:      out = dy * gelu'(z), erf form (may run in place, out == dy).
 */
int a2f_layernorm_fwd(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta, float eps, float* y, float* mean,
                      float* rstd, a2f_stream_t stream);
/* the same with a split-32 image of y beside (or, y == NULL, instead of) the fp32 rows: the operand format of the image-operand a2f_mm (C % 32 == 0) */
int a2f_layernorm_fwd_img(const float* x, int64_t R, int64_t C, const float* gamma, const float* beta, float eps, float* y, void* y_img, float* mean,
                          float* rstd, a2f_stream_t stream);
size_t a2f_layernorm_bwd_ws_bytes(int64_t R, int64_t C);
int a2f_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t C,
                      float* dx, float* dgamma, float* dbeta, float* dx_colsum, void* ws, size_t ws_bytes, a2f_stream_t stream);
/* layernorm_bwd that also hands the Linear below its output gradient as an image: dxm = keep_bits ? (bit ? dx * keep_scale : 0) : dx  (the
 * train-mode dropout that sat on that Linear's output, a2f_keep_bits layout) -> dxm_img (split-32 image, may be NULL) and dxm_colsum = column sums of
 * dxm (= that Linear's bias gradient; may be NULL); dx itself (unmasked: the residual branch) is always written as fp32. */
int a2f_layernorm_bwd_img(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t C,
                          float* dx, float* dgamma, float* dbeta, float* dxm_colsum, const uint32_t* keep_bits, int64_t keep_wpr, float keep_scale,
                          void* dxm_img, void* ws, size_t ws_bytes, a2f_stream_t stream);
size_t a2f_colsum_ws_bytes(int64_t R, int64_t C);
/* colsum over a split-32 image (values hi + lo); ld % 32 == 0; workspace of a2f_colsum_ws_bytes(R, C) */
int a2f_colsum_img(const void* x, int64_t R, int64_t C, int64_t ld, float* out, void* ws, size_t ws_bytes, a2f_stream_t stream);
int a2f_colsum(const float* x, int64_t R, int64_t C, int64_t ld, float* out, void* ws, size_t ws_bytes, a2f_stream_t stream);
int a2f_gelu_bwd(const float* dy, const float* z, int64_t n, float* out, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Train-mode dropout of the wav2vec2 encoder (csrc/dropout.hip; HF Wav2Vec2Model in train mode as the reference runs it:
 * feat_proj_dropout / hidden_dropout / attention_dropout / activation_dropout, /root/reference/src/model/wav2vec.py:87-189 and
 * HF modeling_wav2vec2 Wav2Vec2FeatureProjection / Wav2Vec2Encoder / Wav2Vec2EncoderLayer / Wav2Vec2FeedForward / eager attention).
 * A dropout site is a bit-packed KEEP mask: (R, C) elements = R rows of `wpr` 32-bit words (wpr * 32 >= C), bit b of word (r, w)
 * <-> element (r, 32 w + b), 1 = keep.  a2f_keep_bits fills one from Philox4x32-10: key = seed, counter = (element / 8 within the row,
 * row, site, 0); the four output words are eight 16-bit randoms (element 8 g + 2 e: low half of word e, 8 g + 2 e + 1: its high half) and an
 * element keeps iff its random >= round(p * 2^16) - a pure function of (seed, site, row, column); the drop probability is exact to 2^-17.
 * a2f_keep_bits_attn: the (SH, T, T) mask of the attention probabilities in the two layouts the kernels read: query-major words
 * [sh][q][TW] (bit = key) and key-major words [sh][TW][32 TW] (word (qw, key): bit b = query 32 qw + b); TW = a2f_attn_keep_words(T)
 * = 2 ceil(T / 64); row index of the counter = sh * T + q.  a2f_dropout_apply: y = keep ? x * scale : 0 (streaming; y may be x).
 * The consumers never generate masks themselves, so tests inject the masks recorded from the reference's own dropout calls.
 */
int a2f_keep_bits(uint64_t seed, uint32_t site, int64_t R, int64_t C, float p, uint32_t* bits, int64_t wpr, a2f_stream_t stream);
/* the same mask with the seed read from device memory at run time (one uint64): what a hipGraph-captured train step launches, so that every replay draws
 * fresh masks from the seed the host wrote there (same function of (seed, site, element) as the by-value form) */
int a2f_keep_bits_dev(const uint64_t* seed_dev, uint32_t site, int64_t R, int64_t C, float p, uint32_t* bits, int64_t wpr, a2f_stream_t stream);
int64_t a2f_attn_keep_words(int64_t T);
int a2f_keep_bits_attn(uint64_t seed, uint32_t site, int64_t SH, int64_t T, float p, uint32_t* qmajor, uint32_t* kmajor, a2f_stream_t stream);
int a2f_keep_bits_attn_dev(const uint64_t* seed_dev, uint32_t site, int64_t SH, int64_t T, float p, uint32_t* qmajor, uint32_t* kmajor, a2f_stream_t stream);
int a2f_dropout_apply(const float* x, const uint32_t* bits, int64_t R, int64_t C, int64_t ld, int64_t wpr, float scale, float* y,
                      a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MFMA flash attention of the wav2vec2 encoder layers (csrc/attention.hip): H heads of 64, full self-attention inside each
 * of the S sequences of T frames; replaces softmax(q k^T) v of HF Wav2Vec2Attention (the encoder layers called at
 * /root/reference/src/model/wav2vec.py:174-180) and its backward.
 *   qkv (S*T, ld): row = frame; q at column 0, k at column 64 H, v at column 128 H, head h at + 64 h (the output of the fused
 *   q|k|v projection; q NOT yet scaled: `scale` = head_dim^-0.5 is applied on chip).
 *   fwd: ctx (S*T, ldo) = softmax(scale q k^T) v per head; lse (S, H, T) = log-sum-exp of the scaled scores (saved for bwd).
 *   bwd: dqkv (S*T, ld) from dctx (S*T, ldo), recomputing the probabilities from q, k and lse; delta_ws: S*H*T floats scratch.
 *   precision 1 = bf16x3 split products (fp32-class), 0 = bf16 operands, 2 = fp8 e4m3 operands on v_mfma_f32_32x32x16_fp8_fp8 for the
 *   FORWARD products (probabilities pre-scaled by 2^8; the backward of this mode runs the bf16 kernels; context error 5e-2: a labelled variant,
 *   not a setting of BASELINE configs[4]'s 1e-3).  Rounds 2-3 also carried a three-term e4m3 form that met 1e-3 (six term products per k-step);
 *   it costs the MFMA work of bf16x3 at best - even on the block-scaled K = 64 instruction, 2x rate for 6 products against 3 - plus the
 *   conversions, measured +3 % per step, and was removed in round 4: bf16x3 IS configs[4]'s attention setting.
 *   fp32 softmax and accumulation always.
 * Deterministic (no atomics).  T*ld*4 must be < 2 GiB.
 */
int a2f_attn_fwd(const float* qkv, int64_t S, int64_t T, int64_t H, int64_t ld, float scale, int32_t precision, float* ctx, int64_t ldo,
                 float* lse, a2f_stream_t stream);
int a2f_attn_bwd(const float* qkv, const float* ctx, const float* dctx, const float* lse, int64_t S, int64_t T, int64_t H, int64_t ld,
                 int64_t ldo, float scale, int32_t precision, float* dqkv, float* delta_ws, a2f_stream_t stream);
/* Image forms (keep bits optional: NULL = no dropout): a2f_attn_fwd_img also writes the context as a split-32 image (ctx_img, (S*T, ldo), may be NULL);
 * a2f_attn_bwd_img with dqkv_format 1 writes dqkv as a split-32 image INSTEAD of fp32 (its readers are the image-operand GEMMs of the q|k|v projection).
 * qkv_format / in_format 1: the INPUTS arrive as split-32 images - qkv as the q|k|v projection's GEMM writes it (c_format 1), and (backward) dctx as the
 * out-projection's data-gradient GEMM writes it; ctx stays fp32.  Tiles and MFMA fragments are then copied, not converted, and q is not pre-scaled
 * (the scale multiplies the scores inside the exponent's fused multiply-add; exact for the power-of-two scale of head_dim 64). */
int a2f_attn_fwd_img(const void* qkv, int32_t qkv_format, int64_t S, int64_t T, int64_t H, int64_t ld, float scale, int32_t precision, float* ctx,
                     int64_t ldo, float* lse, const uint32_t* keep_qmajor, float keep_scale, void* ctx_img, a2f_stream_t stream);
int a2f_attn_bwd_img(const void* qkv, const float* ctx, const void* dctx, int32_t in_format, const float* lse, int64_t S, int64_t T, int64_t H,
                     int64_t ld, int64_t ldo, float scale, int32_t precision, void* dqkv, float* delta_ws, const uint32_t* keep_qmajor,
                     const uint32_t* keep_kmajor, float keep_scale, int32_t dqkv_format, a2f_stream_t stream);
/* The same with dropout on the probabilities (HF eager attention: softmax -> nn.functional.dropout(p = attention_dropout) -> @ v):
 * ctx = (P o D) v with D = keep ? keep_scale : 0; lse is that of the un-thinned scores.  keep bits as a2f_keep_bits_attn lays them out. */
int a2f_attn_fwd_drop(const float* qkv, int64_t S, int64_t T, int64_t H, int64_t ld, float scale, int32_t precision, float* ctx, int64_t ldo,
                      float* lse, const uint32_t* keep_qmajor, float keep_scale, a2f_stream_t stream);
int a2f_attn_bwd_drop(const float* qkv, const float* ctx, const float* dctx, const float* lse, int64_t S, int64_t T, int64_t H, int64_t ld,
                      int64_t ldo, float scale, int32_t precision, float* dqkv, float* delta_ws, const uint32_t* keep_qmajor,
                      const uint32_t* keep_kmajor, float keep_scale, a2f_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Layer 0 of the wav2vec2 feature encoder (csrc/featenc.hip; HF Wav2Vec2FeatureEncoder layer 0 as called at
 * /root/reference/src/model/wav2vec.py:116): conv1d(1 -> C, k = 10, stride 5, no bias) -> GroupNorm(C groups: per sequence and
 * channel statistics over time, biased variance, eps) -> GELU (erf), written channels-last.
 *   wave (S, N) fp32, w (C, 1, 10), gamma / beta (C); L = (N - 10) / 5 + 1;  out (S, L, C);  stats (S, 2, C) = mean, rstd (saved).
 *   bwd: dout (S, L, C) -> dw (C, 1, 10), dgamma (C), dbeta (C); dy (S, L, C), optional (NULL to skip) = the gradient w.r.t.
 *   the convolution output, from which the caller forms d/d(wave) when the waveform itself needs a gradient.
 * The convolution is recomputed from the waveform in every pass (10 multiply-adds per output) instead of being stored.
 * C % 4 == 0 and 256 % (C / 4) == 0 (C = 512 here).  Deterministic.  ws: a2f_featenc0_ws_bytes(S, N, C) bytes.
 */
size_t a2f_featenc0_ws_bytes(int64_t S, int64_t N, int64_t C);
int a2f_featenc0_fwd(const float* wave, int64_t S, int64_t N, const float* w, const float* gamma, const float* beta, int64_t C, float eps,
                     float* out, float* stats, void* ws, size_t ws_bytes, a2f_stream_t stream);
/* the same with the output written as a split-32 image (out_format 1; 0 = fp32): the operand format of the convolution stack on image operands */
int a2f_featenc0_fwd_img(const float* wave, int64_t S, int64_t N, const float* w, const float* gamma, const float* beta, int64_t C, float eps,
                         void* out, int32_t out_format, float* stats, void* ws, size_t ws_bytes, a2f_stream_t stream);
int a2f_featenc0_bwd(const float* wave, int64_t S, int64_t N, const float* w, const float* gamma, const float* beta, int64_t C,
                     const float* stats, const float* dout, float* dw, float* dgamma, float* dbeta, float* dy, void* ws, size_t ws_bytes,
                     a2f_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* A2F_HIP_H */
