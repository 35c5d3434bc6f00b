// What does this box give a plain streaming kernel?  float4 copy (1R + 1W) and triad (2R + 1W) on 1 GiB buffers in several shapes,
// to settle the yardstick the fused head kernel is judged against (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).
//   build: hipcc -O3 --offload-arch=gfx950 tools/streambench.hip -o tools/bin/streambench        run: tools/bin/streambench [reps]
// Variants: U = independent 16-byte loads in flight per thread before the first store; NT = nontemporal loads + stores;
// grid = workgroups of 256 threads; "chunk" = every workgroup streams one contiguous region instead of a grid-stride interleave.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using f4 = __attribute__((ext_vector_type(4))) float;

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int U, bool NT, bool TRIAD, bool CHUNK>
__global__ __launch_bounds__(256) void stream_kernel(f4* __restrict__ dst, const f4* __restrict__ a, const f4* __restrict__ b, long long n4) {
  long long i, stride, end;
  if (CHUNK) {                       // workgroup w owns [w * per, (w + 1) * per): threads interleave inside it
    const long long per = (n4 + gridDim.x - 1) / gridDim.x;
    i = blockIdx.x * per + threadIdx.x;
    end = min(n4, (long long)(blockIdx.x + 1) * per);
    stride = 256;
  } else {
    i = blockIdx.x * 256ll + threadIdx.x;
    end = n4;
    stride = (long long)gridDim.x * 256;
  }
  for (; i + (U - 1) * stride < end; i += U * stride) {
    f4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = ld<NT>(a + i + u * stride); if (TRIAD) y[u] = ld<NT>(b + i + u * stride); }
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(dst + i + u * stride, TRIAD ? x[u] + y[u] : x[u]);
  }
  for (; i < end; i += stride) st<NT>(dst + i, TRIAD ? ld<NT>(a + i) + ld<NT>(b + i) : ld<NT>(a + i));
}

struct Res { double best, med; };

template <typename F>
static Res time_it(F launch, int reps, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) launch();
  std::vector<double> gbs;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    gbs.push_back(bytes / (ms * 1e-3) / 1e9);
  }
  std::sort(gbs.begin(), gbs.end());
  return {gbs.back(), gbs[gbs.size() / 2]};
}

template <int U, bool NT, bool CHUNK>
static void run(const char* name, int grid, f4* d, f4* a, f4* b, long long n4, int reps) {
  const double by = (double)n4 * 16;
  Res c = time_it([&] { hipLaunchKernelGGL((stream_kernel<U, NT, false, CHUNK>), dim3(grid), dim3(256), 0, 0, d, a, b, n4); }, reps, 2 * by);
  Res t = time_it([&] { hipLaunchKernelGGL((stream_kernel<U, NT, true, CHUNK>), dim3(grid), dim3(256), 0, 0, d, a, b, n4); }, reps, 3 * by);
  printf("{\"variant\": \"%s\", \"unroll\": %d, \"nontemporal\": %s, \"chunked\": %s, \"grid\": %d, \"copy_GBs_best\": %.1f, \"copy_GBs_median\": %.1f, "
         "\"triad_GBs_best\": %.1f, \"triad_GBs_median\": %.1f}\n", name, U, NT ? "true" : "false", CHUNK ? "true" : "false", grid, c.best, c.med, t.best, t.med);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const long long n4 = 1ll << 26;     // 2^26 float4 = 1 GiB per buffer: beyond the 256 MiB Infinity Cache
  f4 *a, *b, *d;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&d, n4 * 16));
  CK(hipMemset(a, 0x3c, n4 * 16)); CK(hipMemset(b, 0x3d, n4 * 16)); CK(hipMemset(d, 0, n4 * 16));
  CK(hipDeviceSynchronize());
  hipMemcpy(d, a, 1, hipMemcpyDeviceToDevice);
  {                                   // the runtime's own device-to-device copy, for reference
    Res m = time_it([&] { CK(hipMemcpyAsync(d, a, n4 * 16, hipMemcpyDeviceToDevice, 0)); }, reps, 2.0 * n4 * 16);
    printf("{\"variant\": \"hipMemcpyAsync D2D\", \"copy_GBs_best\": %.1f, \"copy_GBs_median\": %.1f}\n", m.best, m.med);
  }
  run<1, false, false>("grid-stride (the library's stream_copy / stream_triad of round 2)", 2048, d, a, b, n4, reps);
  run<1, false, false>("grid-stride", 8192, d, a, b, n4, reps);
  run<1, false, false>("grid-stride, one float4 per thread", (int)(n4 / 256), d, a, b, n4, reps);
  run<4, false, false>("grid-stride", 2048, d, a, b, n4, reps);
  run<4, false, false>("grid-stride", 4096, d, a, b, n4, reps);
  run<8, false, false>("grid-stride", 2048, d, a, b, n4, reps);
  run<4, true, false>("grid-stride", 2048, d, a, b, n4, reps);
  run<8, true, false>("grid-stride", 2048, d, a, b, n4, reps);
  run<1, true, false>("grid-stride", 8192, d, a, b, n4, reps);
  run<4, false, true>("chunked", 2048, d, a, b, n4, reps);
  run<4, true, true>("chunked", 2048, d, a, b, n4, reps);
  run<8, true, true>("chunked", 1024, d, a, b, n4, reps);
  run<4, true, true>("chunked", 8192, d, a, b, n4, reps);
  return 0;
}
