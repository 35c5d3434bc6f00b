// Split-32 images (a2f_mm_operand.format 1, include/a2f_hip.h): the store / load helpers every producer and consumer of images shares.
// A row of ld elements (ld % 32 == 0) is ld / 32 blocks of 128 bytes: 32 bf16 hi = RNE(x), then 32 bf16 lo = RNE(x - hi).
#pragma once
#include "a2f_common.h"

namespace a2f {
namespace img {

using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
using u32x2_t = __attribute__((ext_vector_type(2))) unsigned;

// byte offset of element n of an image row from the row's base
__device__ __forceinline__ int elem_byte(int n) { return (n >> 5) * 128 + (n & 31) * 2; }

// four consecutive elements n .. n + 3 (n % 4 == 0) of the image row at row_base
__device__ __forceinline__ void store4(char* row_base, int n, const f32x4 v) {
  const f32x2_t v0 = {v.x, v.y}, v1 = {v.z, v.w};
  const bf16x2_t h0 = __builtin_convertvector(v0, bf16x2_t), h1 = __builtin_convertvector(v1, bf16x2_t);
  const f32x2_t r0 = v0 - __builtin_convertvector(h0, f32x2_t), r1 = v1 - __builtin_convertvector(h1, f32x2_t);
  u32x2_t hi, lo;
  hi.x = __builtin_bit_cast(unsigned, h0);
  hi.y = __builtin_bit_cast(unsigned, h1);
  lo.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf16x2_t));
  lo.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  char* p = row_base + elem_byte(n);
  *reinterpret_cast<u32x2_t*>(p) = hi;
  *reinterpret_cast<u32x2_t*>(p + 64) = lo;
}

// eight consecutive elements n .. n + 7 (n % 8 == 0): 16 bytes of hi, 16 bytes of lo
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ void store8(char* row_base, int n, const f32x4 a, const f32x4 b) {
  const f32x2_t v[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
  unsigned hi[4], lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2_t h = __builtin_convertvector(v[i], bf16x2_t);
    const f32x2_t r = v[i] - __builtin_convertvector(h, f32x2_t);
    hi[i] = __builtin_bit_cast(unsigned, h);
    lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
  }
  char* p = row_base + elem_byte(n);
  *reinterpret_cast<u32x4_t*>(p) = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
  *reinterpret_cast<u32x4_t*>(p + 64) = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
}

// the values hi + lo of four consecutive elements
__device__ __forceinline__ f32x4 load4(const char* row_base, int n) {
  const char* p = row_base + elem_byte(n);
  const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(p), lo = *reinterpret_cast<const u32x2_t*>(p + 64);
  f32x4 v;
  v.x = __builtin_bit_cast(float, hi.x << 16) + __builtin_bit_cast(float, lo.x << 16);
  v.y = __builtin_bit_cast(float, hi.x & 0xFFFF0000u) + __builtin_bit_cast(float, lo.x & 0xFFFF0000u);
  v.z = __builtin_bit_cast(float, hi.y << 16) + __builtin_bit_cast(float, lo.y << 16);
  v.w = __builtin_bit_cast(float, hi.y & 0xFFFF0000u) + __builtin_bit_cast(float, lo.y & 0xFFFF0000u);
  return v;
}

}  // namespace img
}  // namespace a2f
