// VocaLoss forward / backward on the (B, n) frame matrix  (/root/reference/src/loss/loss.py:24-55).
//
// Pure HBM streaming: every pred/gt element is read once.  One workgroup walks consecutive frame PAIRS
// (rows 2k, 2k+1 are contiguous in memory, 2n floats) so the velocity term (loss.py:32-40) needs no
// second pass.  n = 15069 is odd, so rows are only 4-byte aligned: accesses are dword per lane,
// 256 contiguous bytes per wave-instruction, four independent row streams in flight per thread.
// Forward reduction is deterministic: per-workgroup double partials in the workspace, then one
// fixed-order finishing workgroup.
#include "a2f_common.h"

namespace a2f {

constexpr int L_THREADS = 256;
constexpr int L_MAX_BLOCKS = 4096;

__global__ __launch_bounds__(L_THREADS) void vocaloss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                      int64_t pairs, int n, double* __restrict__ part) {
  const int t = threadIdx.x;
  float rec = 0.f, vel = 0.f;
  for (int64_t p = blockIdx.x; p < pairs; p += gridDim.x) {
    const float* p0 = pred + 2 * p * (int64_t)n;
    const float* g0 = gt + 2 * p * (int64_t)n;
    const float* p1 = p0 + n;
    const float* g1 = g0 + n;
    for (int c = t; c < n; c += L_THREADS) {
      const float d0 = p0[c] - g0[c];
      const float d1 = p1[c] - g1[c];
      const float dv = d1 - d0;
      rec += d0 * d0 + d1 * d1;
      vel += dv * dv;
    }
  }
  __shared__ double sh[2][L_THREADS / 64];
  const double r = wave_sum_d((double)rec), v = wave_sum_d((double)vel);
  if ((t & 63) == 0) { sh[0][t >> 6] = r; sh[1][t >> 6] = v; }
  __syncthreads();
  if (t == 0) {
    double rs = 0, vs = 0;
    for (int w = 0; w < L_THREADS / 64; ++w) { rs += sh[0][w]; vs += sh[1][w]; }
    part[2 * blockIdx.x] = rs;
    part[2 * blockIdx.x + 1] = vs;
  }
}

// out3 = {loss, rec, vel}; rec = sum_rec / (B*V), vel = sum_vel / ((B/2)*V), V = n/3
__global__ void vocaloss_finish_kernel(const double* __restrict__ part, int nparts, double inv_rec, double inv_vel,
                                       float k_rec, float k_vel, float* __restrict__ out3) {
  const int t = threadIdx.x;
  double r = 0, v = 0;
  for (int i = t; i < nparts; i += 64) { r += part[2 * i]; v += part[2 * i + 1]; }
  r = wave_sum_d(r);
  v = wave_sum_d(v);
  if (t == 0) {
    const float rec = (float)(r * inv_rec), vel = (float)(v * inv_vel);
    out3[0] = rec * k_rec + vel * k_vel;
    out3[1] = rec;
    out3[2] = vel;
  }
}

__global__ __launch_bounds__(L_THREADS) void vocaloss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                  const float* __restrict__ g3, int64_t pairs, int n,
                                                                  float k_rec, float k_vel, float inv_rec, float inv_vel,
                                                                  float* __restrict__ grad) {
  // d loss_total / d pred with loss_total = g_loss*(k_rec rec + k_vel vel) + g_rec*rec + g_vel*vel
  const float cr = (g3[0] * k_rec + g3[1]) * 2.f * inv_rec;
  const float cv = (g3[0] * k_vel + g3[2]) * 2.f * inv_vel;
  const int t = threadIdx.x;
  for (int64_t p = blockIdx.x; p < pairs; p += gridDim.x) {
    const int64_t base = 2 * p * (int64_t)n;
    for (int c = t; c < n; c += L_THREADS) {
      const float d0 = pred[base + c] - gt[base + c];
      const float d1 = pred[base + n + c] - gt[base + n + c];
      const float dv = cv * (d1 - d0);
      grad[base + c] = cr * d0 - dv;
      grad[base + n + c] = cr * d1 + dv;
    }
  }
}

}  // namespace a2f

extern "C" size_t a2f_vocaloss_ws_bytes(int64_t B, int64_t n) {
  (void)B; (void)n;
  return (size_t)a2f::L_MAX_BLOCKS * 2 * sizeof(double);
}

extern "C" int a2f_vocaloss_fwd(const float* pred, const float* gt, int64_t B, int64_t n, float k_rec, float k_vel,
                                float* out3, void* ws, size_t ws_bytes, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(pred && gt && out3 && ws, "vocaloss_fwd: null pointer");
  A2F_CHECK_ARG(B >= 2 && (B % 2) == 0, "vocaloss_fwd: B=%lld must be even and >= 2 (loss.py:33-35 pairs rows)", (long long)B);
  A2F_CHECK_ARG(n >= 3 && (n % 3) == 0 && n < (1ll << 30), "vocaloss_fwd: n=%lld must be a multiple of 3", (long long)n);
  if (ws_bytes < a2f_vocaloss_ws_bytes(B, n)) return set_err(A2F_EWORKSPACE, "vocaloss_fwd: workspace %zu < %zu", ws_bytes, a2f_vocaloss_ws_bytes(B, n));
  const int64_t pairs = B / 2;
  const int blocks = (int)(pairs < L_MAX_BLOCKS ? pairs : L_MAX_BLOCKS);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(vocaloss_partial_kernel, dim3(blocks), dim3(L_THREADS), 0, s, pred, gt, pairs, (int)n, (double*)ws);
  const double V = (double)(n / 3);
  hipLaunchKernelGGL(vocaloss_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, blocks, 1.0 / ((double)B * V),
                     1.0 / ((double)pairs * V), k_rec, k_vel, out3);
  A2F_CHECK_LAUNCH("vocaloss_fwd");
  return A2F_OK;
}

extern "C" int a2f_vocaloss_bwd(const float* pred, const float* gt, const float* g3, int64_t B, int64_t n, float k_rec,
                                float k_vel, float* grad_pred, a2f_stream_t stream) {
  using namespace a2f;
  A2F_CHECK_ARG(pred && gt && g3 && grad_pred, "vocaloss_bwd: null pointer");
  A2F_CHECK_ARG(B >= 2 && (B % 2) == 0, "vocaloss_bwd: B=%lld must be even and >= 2", (long long)B);
  A2F_CHECK_ARG(n >= 3 && (n % 3) == 0 && n < (1ll << 30), "vocaloss_bwd: n=%lld must be a multiple of 3", (long long)n);
  const int64_t pairs = B / 2;
  const int blocks = (int)(pairs < 8192 ? pairs : 8192);
  const double V = (double)(n / 3);
  hipLaunchKernelGGL(vocaloss_bwd_kernel, dim3(blocks), dim3(L_THREADS), 0, (hipStream_t)stream, pred, gt, g3, pairs, (int)n,
                     k_rec, k_vel, (float)(1.0 / ((double)B * V)), (float)(1.0 / ((double)pairs * V)), grad_pred);
  A2F_CHECK_LAUNCH("vocaloss_bwd");
  return A2F_OK;
}
