// The C ABI used WITHOUT torch or Python: a plain host program links liba2f_hip.so, owns its device buffers through the
// HIP runtime and calls the fused head entry (a2f_head_fused: Linear 50 -> n + template + VocaLoss + head backward) and the
// standalone loss entry, checking both against a straightforward double-precision host evaluation of the reference formulas
// (voca.py:48-49, loss.py:24-55).  Build: hipcc -O2 tools/cabi_smoke.cpp -Iinclude -Laudio2face-pytorch_amd -la2f_hip
//         -Wl,-rpath,'$ORIGIN/../../audio2face-pytorch_amd' -o tools/bin/cabi_smoke      (done by __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "a2f_hip.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main() {
  const int64_t B = 70, n = 1536, K = 50;
  if (a2f_version() != 100) { fprintf(stderr, "unexpected a2f_version %d\n", a2f_version()); return 1; }
  unsigned seed = 12345u;
  std::vector<float> H(B * K), W(n * K), bias(n), templ(B * n), gt(B * n);
  for (auto& v : H) v = frand(seed);
  for (auto& v : W) v = 0.3f * frand(seed);
  for (auto& v : bias) v = frand(seed);
  for (auto& v : templ) v = 5.f * frand(seed);
  for (size_t i = 0; i < gt.size(); ++i) gt[i] = templ[i] + 0.5f * frand(seed);

  // host evaluation in double
  std::vector<double> pred(B * n);
  for (int64_t b = 0; b < B; ++b)
    for (int64_t c = 0; c < n; ++c) {
      double s = bias[c] + templ[b * n + c];
      for (int64_t k = 0; k < K; ++k) s += (double)H[b * K + k] * W[c * K + k];
      pred[b * n + c] = s;
    }
  const double V = n / 3.0;
  double rec = 0, vel = 0;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t c = 0; c < n; ++c) { const double d = pred[b * n + c] - gt[b * n + c]; rec += d * d; }
  for (int64_t b = 0; b + 1 < B; b += 2)
    for (int64_t c = 0; c < n; ++c) {
      const double d = (pred[(b + 1) * n + c] - pred[b * n + c]) - ((double)gt[(b + 1) * n + c] - gt[b * n + c]);
      vel += d * d;
    }
  rec /= B * V; vel /= (B / 2) * V;
  const double loss = rec + 10.0 * vel;

  float *dH, *dWt, *dbias, *dtempl, *dgt, *dpred, *dout3, *gH, *gW, *gb;
  HIPCHK(hipMalloc(&dH, H.size() * 4)); HIPCHK(hipMalloc(&dWt, W.size() * 4)); HIPCHK(hipMalloc(&dbias, n * 4));
  HIPCHK(hipMalloc(&dtempl, templ.size() * 4)); HIPCHK(hipMalloc(&dgt, gt.size() * 4)); HIPCHK(hipMalloc(&dpred, B * n * 4));
  HIPCHK(hipMalloc(&dout3, 16)); HIPCHK(hipMalloc(&gH, B * K * 4)); HIPCHK(hipMalloc(&gW, n * K * 4)); HIPCHK(hipMalloc(&gb, n * 4));
  HIPCHK(hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dWt, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dbias, bias.data(), n * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dtempl, templ.data(), templ.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dgt, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
  const size_t wsb = a2f_head_fused_ws_bytes(B, n, K);
  void* ws; HIPCHK(hipMalloc(&ws, wsb));
  hipStream_t st; HIPCHK(hipStreamCreate(&st));
  int rc = a2f_head_fused(dH, dWt, dbias, dtempl, B, dgt, B, B, n, K, 1.0f, 10.0f, dpred, dout3, gH, gW, gb, ws, wsb, (a2f_stream_t)st);
  if (rc != 0) { fprintf(stderr, "a2f_head_fused: %d %s\n", rc, a2f_last_error()); return 1; }
  HIPCHK(hipStreamSynchronize(st));
  std::vector<float> hp(B * n); float out3[3];
  HIPCHK(hipMemcpy(hp.data(), dpred, hp.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out3, dout3, 12, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (size_t i = 0; i < hp.size(); ++i) { const double o = pred[i] - templ[i]; const double e = (hp[i] - templ[i]) - o; num += e * e; den += o * o; }
  const double rel_pred = sqrt(num / den), rel_loss = fabs(out3[0] - loss) / loss;
  printf("a2f_head_fused: vertex-offset rel L2 %.2e, loss %.6f (host %.6f, rel %.1e), rec %.6f vel %.6f\n", rel_pred, out3[0], loss, rel_loss, out3[1], out3[2]);

  // the standalone loss entry on the prediction the library just wrote
  const size_t lwb = a2f_vocaloss_ws_bytes(B, n);
  void* lws; HIPCHK(hipMalloc(&lws, lwb));
  rc = a2f_vocaloss_fwd(dpred, dgt, B, n, 1.0f, 10.0f, dout3, lws, lwb, (a2f_stream_t)st);
  if (rc != 0) { fprintf(stderr, "a2f_vocaloss_fwd: %d %s\n", rc, a2f_last_error()); return 1; }
  HIPCHK(hipStreamSynchronize(st));
  float l3[3]; HIPCHK(hipMemcpy(l3, dout3, 12, hipMemcpyDeviceToHost));
  const double rel_loss2 = fabs(l3[0] - loss) / loss;
  printf("a2f_vocaloss_fwd: loss %.6f (rel %.1e)\n", l3[0], rel_loss2);
  // error contract: bad arguments are reported, not executed
  rc = a2f_head_fused(dH, dWt, dbias, dtempl, B, dgt, B, B + 1, n, K, 1.0f, 10.0f, dpred, dout3, gH, gW, gb, ws, wsb, (a2f_stream_t)st);
  if (rc == 0) { fprintf(stderr, "odd Bloss was accepted\n"); return 1; }
  const bool ok = rel_pred < 1e-4 && rel_loss < 1e-4 && rel_loss2 < 1e-4;
  printf("%s\n", ok ? "CABI_SMOKE_OK" : "CABI_SMOKE_FAILED");
  return ok ? 0 : 1;
}
