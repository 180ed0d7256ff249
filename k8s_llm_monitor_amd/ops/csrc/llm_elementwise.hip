// LLM serving kernels for MI355X (gfx950): RMSNorm (+residual), SwiGLU, RoPE + paged-cache write.
// No reference counterpart (vLLM is a README bullet in the reference, README.md:73).
#include "llm_common.h"

using namespace klm_llm;

namespace {

// ------------------------------------------------------------------------------------
// RMSNorm (optionally fused with the residual add): one 256-thread block per token, the
// row stays in registers between the sum of squares and the scaled write (H <= 8192).
//   RES: residual <- bf16(x + residual); out <- rmsnorm(residual) * w
//  !RES:                                 out <- rmsnorm(x) * w
// ------------------------------------------------------------------------------------
template <bool RES>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const uint16_t* __restrict__ x,
                                                      uint16_t* __restrict__ res,
                                                      const uint16_t* __restrict__ w,
                                                      uint16_t* __restrict__ out, int H, float eps) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * H;
  const int tid = threadIdx.x;
  float v[4][8];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = (c * 256 + tid) * 8;
    if (e < H) {
      const uint4 a = *reinterpret_cast<const uint4*>(x + base + e);
      unpack8(a, v[c]);
      if (RES) {
        const uint4 r = *reinterpret_cast<const uint4*>(res + base + e);
        float rv[8];
        unpack8(r, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = bf2f(f2bf(v[c][j] + rv[j]));
        *reinterpret_cast<uint4*>(res + base + e) = pack8(v[c]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[c][j] * v[c][j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  ss = red[0] + red[1] + red[2] + red[3];
  const float inv = rsqrtf(ss / (float)H + eps);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = (c * 256 + tid) * 8;
    if (e < H) {
      const uint4 wa = *reinterpret_cast<const uint4*>(w + e);
      float wv[8], o[8];
      unpack8(wa, wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = v[c][j] * inv * wv[j];
      *reinterpret_cast<uint4*>(out + base + e) = pack8(o);
    }
  }
}

// ------------------------------------------------------------------------------------
// SwiGLU activation on the fused gate_up projection, whose output columns are interleaved in
// groups of 8 (columns 16j..16j+7 = gate 8j..8j+7, columns 16j+8..16j+15 = up 8j..8j+7; the
// weight rows are stored in that order so that the skinny kernel's 16-column tile holds both):
//   out[t][8j + c] = silu(gu[t][16j + c]) * gu[t][16j + 8 + c]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ gu,
                                                       uint16_t* __restrict__ out, size_t total) {
  // `total` = T * F / 8 output vectors of 8; vector i reads the 32 contiguous bytes at gu + 16 i
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const uint4 g = *reinterpret_cast<const uint4*>(gu + i * 16);
    const uint4 u = *reinterpret_cast<const uint4*>(gu + i * 16 + 8);
    float gv[8], uv[8], o[8];
    unpack8(g, gv);
    unpack8(u, uv);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = gv[j] / (1.f + __expf(-gv[j])) * uv[j];
    *reinterpret_cast<uint4*>(out + i * 8) = pack8(o);
  }
}

// ------------------------------------------------------------------------------------
// RoPE (rotate-half pairing i <-> i+64, cos/sin from a table) on the q and k heads of a fused
// qkv activation, in place, and the write of k / v into the paged cache. One wave per
// (token, head).
// ------------------------------------------------------------------------------------
// KV8: the cache holds OCP e4m3 bytes (unit scale) instead of bf16; q and the in-place k stay bf16.
template <bool KV8>
__global__ __launch_bounds__(256) void rope_cache_kernel(uint16_t* __restrict__ qkv,
                                                         const float* __restrict__ cos_sin,
                                                         const int* __restrict__ positions,
                                                         const int* __restrict__ slots,
                                                         void* __restrict__ kcv,
                                                         void* __restrict__ vcv, int T, int n_q,
                                                         int n_kv, int max_pos, int num_slots) {
  uint16_t* kc = reinterpret_cast<uint16_t*>(kcv);
  uint16_t* vc = reinterpret_cast<uint16_t*>(vcv);
  uint8_t* kc8 = reinterpret_cast<uint8_t*>(kcv);
  uint8_t* vc8 = reinterpret_cast<uint8_t*>(vcv);
  const int lane = threadIdx.x & 63;
  const int heads = n_q + 2 * n_kv;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= (long)T * heads) return;
  const int t = (int)(gw / heads), hs = (int)(gw - (long)t * heads);
  uint16_t* row = qkv + ((size_t)t * heads + hs) * HD;
  int pos = positions[t];
  pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
  const int slot = slots[t];
  const bool cache_ok = slot >= 0 && slot < num_slots;
  const uint16_t r1 = row[lane], r2 = row[lane + 64];
  if (hs < n_q + n_kv) {
    const float x1 = bf2f(r1), x2 = bf2f(r2);
    const float c = cos_sin[(size_t)pos * HD + lane], s = cos_sin[(size_t)pos * HD + 64 + lane];
    const uint16_t b1 = f2bf(x1 * c - x2 * s), b2 = f2bf(x2 * c + x1 * s);
    row[lane] = b1;
    row[lane + 64] = b2;
    if (hs >= n_q && cache_ok) {
      const size_t off = ((size_t)(slot / BLK) * n_kv + (hs - n_q)) * BLK_ELEMS + (slot % BLK) * HD;
      if (KV8) {
        kc8[off + lane] = f2fp8(bf2f(b1));
        kc8[off + lane + 64] = f2fp8(bf2f(b2));
      } else {
        kc[off + lane] = b1;
        kc[off + lane + 64] = b2;
      }
    }
  } else if (cache_ok) {
    const size_t off = ((size_t)(slot / BLK) * n_kv + (hs - n_q - n_kv)) * BLK_ELEMS + (slot % BLK);
    if (KV8) {
      vc8[off + lane * BLK] = f2fp8(bf2f(r1));
      vc8[off + (lane + 64) * BLK] = f2fp8(bf2f(r2));
    } else {
      vc[off + lane * BLK] = r1;
      vc[off + (lane + 64) * BLK] = r2;
    }
  }
}

}  // namespace

extern "C" {

int klm_rmsnorm(const void* x, void* residual, const void* weight, void* out, int T, int H, float eps,
                void* stream) {
  if (T <= 0 || H <= 0 || H % 8 != 0 || H > 8192) return KLM_ERR_SHAPE;
  if (!x || !weight || !out) return KLM_ERR_ARG;
  if (!aligned16(x) || !aligned16(weight) || !aligned16(out) || !aligned16(residual)) return KLM_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (residual)
    hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(T), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)residual,
                       (const uint16_t*)weight, (uint16_t*)out, H, eps);
  else
    hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3(T), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)nullptr,
                       (const uint16_t*)weight, (uint16_t*)out, H, eps);
  return launch_status();
}

int klm_silu_mul(const void* gate_up, void* out, int T, int F, void* stream) {
  if (T <= 0 || F <= 0 || F % 8 != 0) return KLM_ERR_SHAPE;
  if (!gate_up || !out) return KLM_ERR_ARG;
  if (!aligned16(gate_up) || !aligned16(out)) return KLM_ERR_ALIGN;
  const size_t total = (size_t)T * (F / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const uint16_t*)gate_up, (uint16_t*)out, total);
  return launch_status();
}

int klm_rope_cache(void* qkv, const float* cos_sin, const int32_t* positions, const int32_t* slots,
                   void* k_cache, void* v_cache, int T, int n_q, int n_kv, int max_pos, int64_t num_slots,
                   int kv_fp8, void* stream) {
  if (T <= 0 || n_q <= 0 || n_kv <= 0 || max_pos <= 0 || num_slots <= 0 || num_slots > INT32_MAX)
    return KLM_ERR_SHAPE;
  if (!qkv || !cos_sin || !positions || !slots || !k_cache || !v_cache) return KLM_ERR_ARG;
  const long waves = (long)T * (n_q + 2 * n_kv);
  const long blocks = (waves + 3) / 4;
  if (blocks > INT32_MAX) return KLM_ERR_SHAPE;
  if (kv_fp8)
    hipLaunchKernelGGL(rope_cache_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (uint16_t*)qkv, cos_sin, positions, slots, k_cache, v_cache, T, n_q, n_kv, max_pos,
                       (int)num_slots);
  else
    hipLaunchKernelGGL(rope_cache_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (uint16_t*)qkv, cos_sin, positions, slots, k_cache, v_cache, T, n_q, n_kv, max_pos,
                       (int)num_slots);
  return launch_status();
}

}  // extern "C"
