// Shared device helpers of the LLM serving kernels (llm_*.hip). See llm_attention.hip for the cache
// layouts and the MFMA operand mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klm_ops.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace klm_llm {

constexpr int HD = 128;               // head dim
constexpr int BLK = 16;               // tokens per KV block
constexpr int BLK_ELEMS = HD * BLK;   // elements of one (block, kv head) slab

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ void unpack8(const uint4& a, float* v) {
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 r;
  r.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  r.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  r.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  r.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  return r;
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// 8 e4m3 bytes (two dwords, k order = byte order) -> one bf16 MFMA fragment
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(uint32_t a, uint32_t b) {
  const bf16x2 p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a, 1.0f, false);
  const bf16x2 p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a, 1.0f, true);
  const bf16x2 p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(b, 1.0f, false);
  const bf16x2 p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(b, 1.0f, true);
  bf16x8 r;
  r[0] = p0[0]; r[1] = p0[1]; r[2] = p1[0]; r[3] = p1[1];
  r[4] = p2[0]; r[5] = p2[1]; r[6] = p3[0]; r[7] = p3[1];
  return r;
}
// one float -> one e4m3 byte (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint8_t f2fp8(float x) {
  x = fminf(fmaxf(x, -448.f), 448.f);
  return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false) & 0xff);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int group_shift(int G) {
  switch (G) {
    case 1: return 0;
    case 2: return 1;
    case 4: return 2;
    case 8: return 3;
    case 16: return 4;
    default: return -1;
  }
}
inline int launch_status() { return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP; }


}  // namespace klm_llm
