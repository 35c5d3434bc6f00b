// I/O-pattern floor of the fused head pass at its real geometry: pred = templ + bias-ish, loss-ish sum over (pred - gt),
// (B, n) = (19200, 15069) fp32, rows 4-byte aligned only.  Every mode moves the same 3 x B x n x 4 bytes (2 reads + 1 write)
// with the head kernel's grid (59 column groups x 4 row groups, 512 threads, 32-row blocks) unless stated, next block's
// loads in flight while the current block is stored.
//   mode 0  MFMA C-layout dword: wave = 32 columns, one instruction = 2 rows x 128 B          (what head_x3 does)
//   mode 1  row-contiguous dword: wave = 64 columns x 16 rows, one instruction = 1 row x 256 B  (permlane32_swap form)
//   mode 2  quad dwordx4: wave = 32 columns, one instruction = 8 rows x 128 B (4-lane transposes afterwards)
//   mode 3  workgroup rows: one instruction = 1 row x 1 KB (dwordx4 per lane, whole 256-column group)
//   mode 4  mode 1 with dwordx2 per lane: wave = 128 columns x 8 rows, one instruction = 1 row x 512 B
// XCD = 1: blockIdx remapped so the 8 XCDs own contiguous (row group, column group) runs.
// build: hipcc -O3 --offload-arch=gfx950 tools/membench2.hip -o gpurun_out/membench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
__device__ __forceinline__ rsrc_t mk(const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7FFFFFF0, 0x00020000); }
template <int AUX> __device__ __forceinline__ float ld1(rsrc_t r, int off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX)); }
template <int AUX> __device__ __forceinline__ void st1(rsrc_t r, int off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, AUX); }
template <int AUX> __device__ __forceinline__ u32x4 ld4(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX); }
template <int AUX> __device__ __forceinline__ void st4(rsrc_t r, int off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX); }
template <int AUX> __device__ __forceinline__ u32x2 ld2(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX); }
template <int AUX> __device__ __forceinline__ void st2(rsrc_t r, int off, u32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, AUX); }
__device__ __forceinline__ float f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned u(float x) { return __builtin_bit_cast(unsigned, x); }

struct Args { const float* t; const float* g; float* p; float* sink; int B, n, CG, RG, NRB, xcd, nblk; };

__device__ __forceinline__ void block_id(const Args& a, int& cg, int& rg) {
  int v = blockIdx.x;
  if (a.xcd) {                      // XCD x = blockIdx % 8 gets the contiguous run [start_x, start_x + cnt_x)
    const int x = v & 7, i = v >> 3, q = a.nblk >> 3, rem = a.nblk & 7;
    v = x * q + (x < rem ? x : rem) + i;
  }
  cg = v % a.CG; rg = v / a.CG;
}

// ---- mode 0 ----
template <int LA, int SA>
__global__ __launch_bounds__(512, 2) void k0(const Args a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
  int cg, rg; block_id(a, cg, rg);
  const int n = a.n, col = cg * 256 + wave * 32 + l31;
  const int rb0 = rg * a.NRB / a.RG, rb1 = (rg + 1) * a.NRB / a.RG;
  const int mcol = (col - n) >> 31;
  float tc[16], gc[16], acc = 0.f;
  auto issue = [&](int rb) {
    const rsrc_t tb = mk(a.t + ((int64_t)rb * 32 * n + cg * 256 + wave * 32)), gb = mk(a.g + ((int64_t)rb * 32 * n + cg * 256 + wave * 32));
    const int m = mcol & ((rb - rb1) >> 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int off = (4 * (rr * n + l31)) | ~(m & ((rb * 32 + rr - a.B) >> 31));
      tc[r] = ld1<LA>(tb, off); gc[r] = ld1<LA>(gb, off);
    }
  };
  issue(rb0);
  for (int rb = rb0; rb < rb1; ++rb) {
    const rsrc_t pb = mk(a.p + ((int64_t)rb * 32 * n + cg * 256 + wave * 32));
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { pv[r] = tc[r] + 1.f; const float d = pv[r] - gc[r]; acc += d * d; }
    issue(rb + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
      st1<SA>(pb, (4 * (rr * n + l31)) | ~(mcol & ((rb * 32 + rr - a.B) >> 31)), pv[r]);
    }
  }
  if (acc == 12345.678f) a.sink[0] = acc;
}

// ---- mode 1: wave w = column pair (w & 3) x row half (w >> 2): 64 columns x 16 rows ----
template <int LA, int SA>
__global__ __launch_bounds__(512, 2) void k1(const Args a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int cg, rg; block_id(a, cg, rg);
  const int n = a.n, cb = cg * 256 + (wave & 3) * 64, col = cb + lane, r0 = (wave >> 2) * 16;
  const int rb0 = rg * a.NRB / a.RG, rb1 = (rg + 1) * a.NRB / a.RG;
  const int mcol = (col - n) >> 31;
  float tc[16], gc[16], acc = 0.f;
  auto issue = [&](int rb) {
    const rsrc_t tb = mk(a.t + ((int64_t)(rb * 32 + r0) * n + cb)), gb = mk(a.g + ((int64_t)(rb * 32 + r0) * n + cb));
    const int m = mcol & ((rb - rb1) >> 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = (4 * (r * n + lane)) | ~(m & ((rb * 32 + r0 + r - a.B) >> 31));
      tc[r] = ld1<LA>(tb, off); gc[r] = ld1<LA>(gb, off);
    }
  };
  issue(rb0);
  for (int rb = rb0; rb < rb1; ++rb) {
    const rsrc_t pb = mk(a.p + ((int64_t)(rb * 32 + r0) * n + cb));
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { pv[r] = tc[r] + 1.f; const float d = pv[r] - gc[r]; acc += d * d; }
    issue(rb + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) st1<SA>(pb, (4 * (r * n + lane)) | ~(mcol & ((rb * 32 + r0 + r - a.B) >> 31)), pv[r]);
  }
  if (acc == 12345.678f) a.sink[0] = acc;
}

// ---- mode 2: wave = 32 columns x 32 rows, lane (q = l31 >> 2, i = l31 & 3, h) loads row 8m + i + 4h, columns 4q..4q+3 ----
template <int LA, int SA>
__global__ __launch_bounds__(512, 2) void k2(const Args a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5, q = l31 >> 2, i = l31 & 3;
  int cg, rg; block_id(a, cg, rg);
  const int n = a.n, cb = cg * 256 + wave * 32, col = cb + 4 * q;
  const int rb0 = rg * a.NRB / a.RG, rb1 = (rg + 1) * a.NRB / a.RG;
  const bool full = col + 3 < n;
  u32x4 tc[4], gc[4]; float acc = 0.f;
  // boundary quads (only in the last column group) fall back to nothing: the bench only needs the traffic shape
  auto issue = [&](int rb) {
    const rsrc_t tb = mk(a.t + ((int64_t)rb * 32 * n + cb)), gb = mk(a.g + ((int64_t)rb * 32 * n + cb));
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int rr = 8 * m + i + 4 * h;
      const int off = (full && rb < rb1 && rb * 32 + rr < a.B) ? 4 * (rr * n + 4 * q) : -1;
      tc[m] = ld4<LA>(tb, off); gc[m] = ld4<LA>(gb, off);
    }
  };
  issue(rb0);
  for (int rb = rb0; rb < rb1; ++rb) {
    const rsrc_t pb = mk(a.p + ((int64_t)rb * 32 * n + cb));
    u32x4 pv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float p = f(tc[m][j]) + 1.f; pv[m][j] = u(p); const float d = p - f(gc[m][j]); acc += d * d; }
    issue(rb + 1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int rr = 8 * m + i + 4 * h;
      st4<SA>(pb, (full && rb * 32 + rr < a.B) ? 4 * (rr * n + 4 * q) : -1, pv[m]);
    }
  }
  if (acc == 12345.678f) a.sink[0] = acc;
}

// ---- mode 3: one instruction = one 1-KB row of the column group; wave w owns rows 4w..4w+3 of the block ----
template <int LA, int SA>
__global__ __launch_bounds__(512, 2) void k3(const Args a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int cg, rg; block_id(a, cg, rg);
  const int n = a.n, cb = cg * 256, col = cb + 4 * lane;
  const int rb0 = rg * a.NRB / a.RG, rb1 = (rg + 1) * a.NRB / a.RG;
  const bool full = col + 3 < n;
  u32x4 tc[4], gc[4]; float acc = 0.f;
  auto issue = [&](int rb) {
    const rsrc_t tb = mk(a.t + ((int64_t)rb * 32 * n + cb)), gb = mk(a.g + ((int64_t)rb * 32 * n + cb));
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int rr = 4 * wave + m;
      const int off = (full && rb < rb1 && rb * 32 + rr < a.B) ? 4 * (rr * n + 4 * lane) : -1;
      tc[m] = ld4<LA>(tb, off); gc[m] = ld4<LA>(gb, off);
    }
  };
  issue(rb0);
  for (int rb = rb0; rb < rb1; ++rb) {
    const rsrc_t pb = mk(a.p + ((int64_t)rb * 32 * n + cb));
    u32x4 pv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float p = f(tc[m][j]) + 1.f; pv[m][j] = u(p); const float d = p - f(gc[m][j]); acc += d * d; }
    issue(rb + 1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int rr = 4 * wave + m;
      st4<SA>(pb, (full && rb * 32 + rr < a.B) ? 4 * (rr * n + 4 * lane) : -1, pv[m]);
    }
  }
  if (acc == 12345.678f) a.sink[0] = acc;
}

// ---- mode 4: dwordx2 per lane, wave = (column half w & 1: 128 columns) x (row quarter w >> 1: 8 rows) ----
template <int LA, int SA>
__global__ __launch_bounds__(512, 2) void k4(const Args a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int cg, rg; block_id(a, cg, rg);
  const int n = a.n, cb = cg * 256 + (wave & 1) * 128, col = cb + 2 * lane, r0 = (wave >> 1) * 8;
  const int rb0 = rg * a.NRB / a.RG, rb1 = (rg + 1) * a.NRB / a.RG;
  const bool full = col + 1 < n;
  u32x2 tc[8], gc[8]; float acc = 0.f;
  auto issue = [&](int rb) {
    const rsrc_t tb = mk(a.t + ((int64_t)(rb * 32 + r0) * n + cb)), gb = mk(a.g + ((int64_t)(rb * 32 + r0) * n + cb));
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int off = (full && rb < rb1 && rb * 32 + r0 + m < a.B) ? 4 * (m * n + 2 * lane) : -1;
      tc[m] = ld2<LA>(tb, off); gc[m] = ld2<LA>(gb, off);
    }
  };
  issue(rb0);
  for (int rb = rb0; rb < rb1; ++rb) {
    const rsrc_t pb = mk(a.p + ((int64_t)(rb * 32 + r0) * n + cb));
    u32x2 pv[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j) { const float p = f(tc[m][j]) + 1.f; pv[m][j] = u(p); const float d = p - f(gc[m][j]); acc += d * d; }
    issue(rb + 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) st2<SA>(pb, (full && rb * 32 + r0 + m < a.B) ? 4 * (m * n + 2 * lane) : -1, pv[m]);
  }
  if (acc == 12345.678f) a.sink[0] = acc;
}

// ---- modes 5 / 6: loads as mode 3 (1-KB rows), staged through LDS into the MFMA C layout (per-wave chunks [32 rows][32 cols]);
// stores either straight from the C layout (2 x 128 B, STROW = 0) or back through the LDS chunk as 1-KB rows (STROW = 1) ----
template <int LA, int SA, int STROW>
__global__ __launch_bounds__(512, 2) void k5(const Args a) {
  __shared__ __attribute__((aligned(16))) float stage[8 * 2 * 1024];      // [chunk][T | G][32][32]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
  int cg, rg; block_id(a, cg, rg);
  const int n = a.n, cb = cg * 256;
  const int rb0 = rg * a.NRB / a.RG, rb1 = (rg + 1) * a.NRB / a.RG;
  const bool full = cb + 4 * lane + 3 < n;
  const int colc = cb + wave * 32 + l31, mcol = (colc - n) >> 31;
  u32x4 tc[4], gc[4]; float acc = 0.f;
  auto issue = [&](int rb) {
    const rsrc_t tb = mk(a.t + ((int64_t)rb * 32 * n + cb)), gb = mk(a.g + ((int64_t)rb * 32 * n + cb));
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int rr = 4 * wave + m;
      const int off = (full && rb < rb1 && rb * 32 + rr < a.B) ? 4 * (rr * n + 4 * lane) : -1;
      tc[m] = ld4<LA>(tb, off); gc[m] = ld4<LA>(gb, off);
    }
  };
  float* mychunk = stage + wave * 2048;
  float* wr = stage + (lane >> 3) * 2048 + (lane & 7) * 4;      // + row * 32 (+ 1024 for G)
  issue(rb0);
  for (int rb = rb0; rb < rb1; ++rb) {
    __syncthreads();                                   // A: previous block's chunk reads are complete
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      *reinterpret_cast<u32x4*>(wr + (4 * wave + m) * 32) = tc[m];
      *reinterpret_cast<u32x4*>(wr + 1024 + (4 * wave + m) * 32) = gc[m];
    }
    __syncthreads();                                   // S
    issue(rb + 1);
    const rsrc_t pb = mk(a.p + ((int64_t)rb * 32 * n + cb + wave * 32));
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
      pv[r] = mychunk[rr * 32 + l31] + 1.f; const float d = pv[r] - mychunk[1024 + rr * 32 + l31]; acc += d * d;
    }
    if (!STROW) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        st1<SA>(pb, (4 * (rr * n + l31)) | ~(mcol & ((rb * 32 + rr - a.B) >> 31)), pv[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) mychunk[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = pv[r];
      __syncthreads();                                 // P
      const rsrc_t pr = mk(a.p + ((int64_t)rb * 32 * n + cb));
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int rr = 4 * wave + m;
        st4<SA>(pr, (full && rb * 32 + rr < a.B) ? 4 * (rr * n + 4 * lane) : -1, *reinterpret_cast<const u32x4*>(wr + rr * 32));
      }
    }
  }
  if (acc == 12345.678f) a.sink[0] = acc;
}

// plain aligned float4 triad over the same byte count (calibration of the box: 2 reads + 1 write)
__global__ __launch_bounds__(256) void triad(const float4* __restrict__ x, const float4* __restrict__ y, float4* __restrict__ o, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 a = x[i], b = y[i];
    o[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

int main() {
  const int B = 19200, n = 15069;
  const size_t bytes = (size_t)B * n * 4;
  float *t, *g, *p, *sink;
  hipMalloc(&t, bytes + 64); hipMalloc(&g, bytes + 64); hipMalloc(&p, bytes + 64); hipMalloc(&sink, 64);
  hipMemset(t, 0, bytes); hipMemset(g, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto launch, const char* name) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-58s %.3f ms  %.0f GB/s\n", name, ms, 3.0 * bytes / ms / 1e6); fflush(stdout);
  };
  timeit([&] { hipLaunchKernelGGL(triad, dim3(256 * 8), dim3(256), 0, 0, (const float4*)t, (const float4*)g, (float4*)p, bytes / 16); }, "float4 triad, 2048 blocks (aligned, calibration)");
  auto run = [&](auto kern, const char* name, int CG, int RG, int xcd) {
    Args a{t, g, p, sink, B, n, CG, RG, B / 32, xcd, CG * RG};
    char nm[128]; snprintf(nm, sizeof nm, "%s grid %dx%d%s", name, CG, RG, xcd ? " xcd" : "");
    timeit([&] { hipLaunchKernelGGL(kern, dim3(CG * RG), dim3(512), 0, 0, a); }, nm);
  };
  run(k0<0, 0>, "m0 C-layout 2x128B", 59, 4, 0);
  run(k0<0, 0>, "m0 C-layout 2x128B", 59, 4, 1);
  run(k0<0, 0>, "m0 C-layout 2x128B", 59, 8, 0);
  run(k0<2, 2>, "m0 C-layout 2x128B nt/nt", 59, 4, 0);
  run(k0<0, 2>, "m0 C-layout 2x128B -/nt", 59, 4, 0);
  run(k1<0, 0>, "m1 row 256B", 59, 4, 0);
  run(k1<0, 0>, "m1 row 256B", 59, 4, 1);
  run(k1<0, 0>, "m1 row 256B", 59, 8, 0);
  run(k1<2, 2>, "m1 row 256B nt/nt", 59, 4, 0);
  run(k2<0, 0>, "m2 quad x4 8x128B", 59, 4, 0);
  run(k2<0, 0>, "m2 quad x4 8x128B", 59, 4, 1);
  run(k3<0, 0>, "m3 wg row 1KB x4", 59, 4, 0);
  run(k3<0, 0>, "m3 wg row 1KB x4", 59, 4, 1);
  run(k3<0, 0>, "m3 wg row 1KB x4", 59, 8, 0);
  run(k3<2, 2>, "m3 wg row 1KB x4 nt/nt", 59, 4, 0);
  run(k3<0, 2>, "m3 wg row 1KB x4 -/nt", 59, 4, 0);
  run(k5<0, 0, 0>, "m5 ld 1KB rows via LDS, st C-layout", 59, 4, 0);
  run(k5<0, 0, 0>, "m5 ld 1KB rows via LDS, st C-layout", 59, 4, 1);
  run(k5<0, 0, 1>, "m6 ld+st 1KB rows via LDS", 59, 4, 0);
  run(k5<0, 0, 1>, "m6 ld+st 1KB rows via LDS", 59, 4, 1);
  run(k5<0, 2, 1>, "m6 ld+st 1KB rows via LDS -/nt", 59, 4, 1);
  run(k4<0, 0>, "m4 row 512B x2", 59, 4, 0);
  run(k4<0, 0>, "m4 row 512B x2", 59, 4, 1);
  return 0;
}
