// Stand-alone two-kernel reproducer for DESIGN 2.4 (the cross-queue irreproducibility of rounds 4-5).
//
// What tools/hash_bisect.py traced in the FaceFormer step: ONE launch (the LayerNorm backward that is running when the side stream's decoder recurrence ends
// and its chip-wide fused head starts) computes x-hat = (x - mean) * rstd with `v_pk_add_f32 vD, vX, v[MU0:MU1] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]`
// (hipcc's SLP vectoriser packs the four subtractions of a float4 into two packed-fp32 instructions; for the second row of a trip the subtrahend is the HIGH
// register of a pair, selected for the LOW result by op_sel[1] = 1).  In the bad launches exactly one LOW result of one such instruction came out as x - 0
// in lanes 48..63 of one wave (registers, loads, reductions all right: per-lane dumps), ~1 launch in 10 under the overlap, never without it, and never once
// the kernel was compiled without packed-fp32 (-fno-slp-vectorize: 0 / 80 against 9 / 60).
//
//   stream A (victim):    waves loop over that instruction (inline asm) against the two scalar subtractions, on changing data, counting lanes that differ.
//   stream B (aggressor): chip-wide workgroups of 512 threads with 150 KB of LDS running bf16 MFMAs + streaming loads, started again and again (every start is
//                         the "head kernel begins" event).
// Result on MI355X (ROCm 7.2): 0 wrong results in 3e10 executions under every aggressor setting - the instruction alone, even beside chip-wide 150-KB-LDS MFMA
// kernels, does not misread; the full three-queue step does (tools/hash_bisect.py).  Kept as the record of what does NOT trigger it.
// usage: pkadd_repro [seconds] [aggr_wgs] [aggr_lds_kb] [aggr_us] [gap_launches]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

using f2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;

__global__ __launch_bounds__(256) void victim_kernel(const float4* __restrict__ src, size_t n4, float4* __restrict__ dst, unsigned long long* __restrict__ bad, unsigned seed,
                                                     long long ticks) {
  __shared__ float pad[12288];                                   // 48 KB like layernorm_bwd_kernel<4>: at most three such workgroups per CU
  const unsigned lane = threadIdx.x & 63;
  pad[threadIdx.x] = (float)seed;
  const long long t0 = wall_clock64();
  size_t q = ((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)seed * 7919u) % n4;
  unsigned long long nbad = 0, rounds = 0;
  unsigned first = 0;
  float mlo = 0.001f * (float)(lane + 1), mhi = -0.0017f;
  while (wall_clock64() - t0 < ticks) {
    const float4 a = src[q], b = src[(q + 64) % n4];             // real loads in flight, as in the product kernel
    q += (size_t)gridDim.x * 256;
    if (q >= n4) q -= n4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int r = 0; r < 32; ++r) {
      const float x0 = a.x + 0.37f * (float)r, x1 = a.y - 0.11f * (float)r, x2 = b.z + 1.3f, x3 = b.w * 0.5f + (float)r;
      mhi = -0.0017f - 1e-5f * (float)((r + seed) & 31);
      f2 xa = {x0, x1}, xb = {x2, x3}, m = {mlo, mhi}, oa, ob;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(oa) : "v"(xa), "v"(m));
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(ob) : "v"(xb), "v"(m));
      float r0, r1, r2, r3;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(mhi));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(mhi));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r2) : "v"(x2), "v"(mhi));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r3) : "v"(x3), "v"(mhi));
      const unsigned w = (__float_as_uint(oa.x) != __float_as_uint(r0) ? 1u : 0u) | (__float_as_uint(oa.y) != __float_as_uint(r1) ? 2u : 0u) |
                         (__float_as_uint(ob.x) != __float_as_uint(r2) ? 4u : 0u) | (__float_as_uint(ob.y) != __float_as_uint(r3) ? 8u : 0u);
      if (w) { if (!nbad) first = (w << 8) | lane; ++nbad; }
      acc.x += oa.x; acc.y += oa.y; acc.z += ob.x; acc.w += ob.y;
    }
    dst[((size_t)blockIdx.x * 256 + threadIdx.x)] = acc;         // stores in flight too
    ++rounds;
  }
  if (nbad) {
    atomicAdd(&bad[0], nbad);
    bad[1] = first;
    bad[2] = ((unsigned long long)blockIdx.x << 32) | threadIdx.x;
    atomicAdd(&bad[5 + (lane >> 4)], nbad);
  }
  if (threadIdx.x == 0) atomicAdd(&bad[3], rounds * 32ull);
  if (pad[(threadIdx.x * 7) % 12288] == -1.f) bad[4] = 1;
}

__global__ __launch_bounds__(512) void aggressor_kernel(const float4* __restrict__ src, size_t n4, float* __restrict__ sink, long long ticks, int lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const long long t0 = wall_clock64();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  bf16x8_t fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.001f * (float)(threadIdx.x + i)); fb[i] = (__bf16)(0.002f * (float)(i + 1)); }
  size_t q = ((size_t)blockIdx.x * 512 + threadIdx.x) % n4;
  int li = threadIdx.x * 4;
  while (wall_clock64() - t0 < ticks) {
#pragma unroll 4
    for (int u = 0; u < 4; ++u) {
      const float4 v = src[q];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      q += (size_t)gridDim.x * 512;
      if (q >= n4) q -= n4;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {                                // a dense burst on the matrix pipe
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fa, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fb, c3, 0, 0, 0);
    }
    *reinterpret_cast<float4*>(lds + li) = acc;
    li += 2048;
    if (li + 4 > lds_floats) li = threadIdx.x * 4;
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = acc.x + acc.y + acc.z + acc.w + c0[0] + c1[1] + c2[2] + c3[3] + lds[(blockIdx.x * 7) % lds_floats];
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
  const int awgs = argc > 2 ? atoi(argv[2]) : 256;
  const int alds_kb = argc > 3 ? atoi(argv[3]) : 150;
  const int aus = argc > 4 ? atoi(argv[4]) : 400;
  const int gap = argc > 5 ? atoi(argv[5]) : 4;
  unsigned long long* bad;
  CK(hipMalloc(&bad, 16 * 8));
  CK(hipMemset(bad, 0, 16 * 8));
  const size_t src_n4 = (size_t)1 << 26;
  float4 *src = nullptr, *dst = nullptr;
  float* sink = nullptr;
  CK(hipMalloc(&src, src_n4 * 16));
  CK(hipMemset(src, 0x3c, src_n4 * 16));
  CK(hipMalloc(&dst, (size_t)1024 * 256 * 16));
  CK(hipMalloc(&sink, 4096 * 4));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  int plo, phi;
  CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, phi));
  const int lds_bytes = alds_kb * 1024;
  if (awgs > 0) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  int clk_khz = 100000;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
  const long long aticks = (long long)aus * clk_khz / 1000, vticks = 150ll * clk_khz / 1000;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned n = 0;
  long launches_a = 0, launches_b = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int k = 0; k < 32; ++k) {
      if (awgs > 0 && gap > 0 && (k % gap) == 0) { hipLaunchKernelGGL(aggressor_kernel, dim3(awgs), dim3(512), lds_bytes, sb, src, src_n4, sink, aticks, lds_bytes / 4); ++launches_b; }
      hipLaunchKernelGGL(victim_kernel, dim3(256), dim3(256), 0, sa, src, src_n4, dst, bad, ++n, vticks);
      ++launches_a;
    }
    CK(hipStreamSynchronize(sa));
  }
  CK(hipDeviceSynchronize());
  unsigned long long h[16];
  CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("%.0f s, victim %ld launches (%llu packed subtractions per lane-pair checked), aggressor %ld launches of %d wgs x %d KB LDS x %d us every %d victim launches: WRONG results %llu"
         " by lane quarter [%llu %llu %llu %llu]", seconds, launches_a, h[3] * 2, launches_b, awgs, alds_kb, aus, gap, h[0], h[5], h[6], h[7], h[8]);
  if (h[0]) printf("  (first: halves mask %#llx lane %llu, block %llu thread %llu)", (h[1] >> 8) & 15, h[1] & 255, h[2] >> 32, h[2] & 0xFFFFFFFFull);
  printf("\n");
  return h[0] ? 1 : 0;
}
