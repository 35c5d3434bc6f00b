// Access-pattern micro-benchmark for the (B, n) frame matrix (n = 15069, rows 4-byte aligned only).
//  mode 0: MFMA C-layout tile reads: one wave-instruction = 2 rows x 128 B (lanes 0-31 row r, lanes 32-63 row r+4)
//  mode 1: row-contiguous: one wave-instruction = 1 row x 256 B
// Same grid / tile walk as head_fused (30 column groups x 8 row groups, 8 waves x 64 columns, 32-row blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int WRITE, int NW>
__global__ __launch_bounds__(64 * NW) void k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int B, int n, int NRB, int RG, int CG) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int cg = blockIdx.x % CG, rg = blockIdx.x / CG;
  const int rb0 = rg * NRB / RG, rb1 = (rg + 1) * NRB / RG;
  float acc = 0.f;
  for (int rb = rb0; rb < rb1; ++rb) {
    if (MODE == 0) {
      for (int ct = 0; ct < 16 / NW; ++ct) {
        const int col = cg * 512 + wave * (512 / NW) + ct * 32 + l31;
        float va[16], vb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool ok = col < n && row < B;
          const long off = ok ? (long)row * n + col : 0;
          va[r] = a[off]; vb[r] = b[off];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc += va[r] * vb[r];
          if (WRITE) { const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; if (col < n && row < B) o[(long)row * n + col] = va[r] + vb[r]; }
        }
      }
    } else {
      for (int cc = 0; cc < 8 / NW; ++cc) {
      const int col = cg * 512 + wave * (512 / NW) + cc * 64 + lane;
      float va[32], vb[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int row = rb * 32 + r;
        const bool ok = col < n && row < B;
        const long off = ok ? (long)row * n + col : 0;
        va[r] = a[off]; vb[r] = b[off];
      }
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        acc += va[r] * vb[r];
        if (WRITE) { const int row = rb * 32 + r; if (col < n && row < B) o[(long)row * n + col] = va[r] + vb[r]; }
      }
      }
    }
  }
  if (acc == 12345.678f) o[0] = acc;
}
int main() {
  const int B = 19200, n = 15069, CG = 30, RG = 8, NRB = 600;
  size_t bytes = (size_t)B * n * 4;
  float *a, *b, *o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, bytes);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int wr, int nw = 8) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(CG * RG), dim3(64 * nw), 0, 0, a, b, o, B, n, NRB, RG, CG);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(CG * RG), dim3(64 * nw), 0, 0, a, b, o, B, n, NRB, RG, CG);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-36s %.3f ms  %.0f GB/s\n", name, ms, (2 + wr) * (double)bytes / ms / 1e6);
  };
  run(k<0, 0, 8>, "8 waves tile 2x128B read-only", 0, 8);
  run(k<1, 0, 8>, "8 waves row 256B read-only", 0, 8);
  run(k<0, 1, 8>, "8 waves tile 2x128B read+write", 1, 8);
  run(k<1, 1, 8>, "8 waves row 256B read+write", 1, 8);
  run(k<0, 0, 4>, "4 waves tile 2x128B read-only", 0, 4);
  run(k<0, 1, 4>, "4 waves tile 2x128B read+write", 1, 4);
  run(k<1, 1, 4>, "4 waves row 256B read+write", 1, 4);
  return 0;
}
