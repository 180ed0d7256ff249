// LLM serving kernels for MI355X (gfx950): RMSNorm (+residual), SwiGLU, RoPE + paged-cache write.
// No reference counterpart (vLLM is a README bullet in the reference, README.md:73).
#include "llm_common.h"

using namespace klm_llm;

namespace {

// ------------------------------------------------------------------------------------
// RMSNorm (optionally fused with the residual add): one 256-thread block per token, the
// row stays in registers between the sum of squares and the scaled write: CH vectors of 8 per
// thread, CH = 4 for H <= 8192, 8 up to 16384 (the 405B hidden size).
//   RES: residual <- bf16(x + residual); out <- rmsnorm(residual) * w
//  !RES:                                 out <- rmsnorm(x) * w
// ------------------------------------------------------------------------------------
template <bool RES, int CH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const uint16_t* __restrict__ x,
                                                      uint16_t* __restrict__ res,
                                                      const uint16_t* __restrict__ w,
                                                      uint16_t* __restrict__ out, int H, float eps) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * H;
  const int tid = threadIdx.x;
  float v[CH][8];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
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
  for (int c = 0; c < CH; ++c) {
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
// Eight lanes per (token, head): lane j of the group owns dims 8j..8j+7 and their partners 64+8j..,
// so every global access of the activation is a 16-byte vector (a 2-byte-per-lane version ran the
// 4096-token prefill's rope at 2.2 TB/s); only the transposed V cache write stays element-wise.
template <bool KV8, bool PART>
__global__ __launch_bounds__(256) void rope_cache_kernel(uint16_t* __restrict__ qkv,
                                                         const float* __restrict__ cos_sin,
                                                         const int* __restrict__ positions,
                                                         const int* __restrict__ slots,
                                                         void* __restrict__ kcv,
                                                         void* __restrict__ vcv, int T, int n_q,
                                                         int n_kv, int max_pos, int num_slots,
                                                         const float* __restrict__ part, int ks) {
  uint16_t* kc = reinterpret_cast<uint16_t*>(kcv);
  uint16_t* vc = reinterpret_cast<uint16_t*>(vcv);
  uint8_t* kc8 = reinterpret_cast<uint8_t*>(kcv);
  uint8_t* vc8 = reinterpret_cast<uint8_t*>(vcv);
  const int heads = n_q + 2 * n_kv;
  const long pair = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;       // (token, head) index
  if (pair >= (long)T * heads) return;
  const int j = threadIdx.x & 7;
  const int t = (int)(pair / heads), hs = (int)(pair - (long)t * heads);
  uint16_t* row = qkv + ((size_t)t * heads + hs) * HD;
  int pos = positions[t];
  pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
  const int slot = slots[t];
  const bool cache_ok = slot >= 0 && slot < num_slots;
  float x1[8], x2[8];
  if (PART) {
    // the projection's K splits are summed here (mid-batch kernel, llm_mid.hip): fp32 [ks][T][heads * 128], rounded to
    // bf16 once, exactly what the separate epilogue kernel would have stored
    const size_t split = (size_t)T * heads * HD;
    const float* p = part + ((size_t)t * heads + hs) * HD + j * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) x1[i] = x2[i] = 0.f;
#pragma unroll 4
    for (int k = 0; k < ks; ++k) {
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(p + k * split), u1 = *reinterpret_cast<const f32x4*>(p + k * split + 4);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(p + k * split + 64), w1 = *reinterpret_cast<const f32x4*>(p + k * split + 68);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x1[i] += u0[i]; x1[4 + i] += u1[i];
        x2[i] += w0[i]; x2[4 + i] += w1[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x1[i] = bf2f(f2bf(x1[i]));
      x2[i] = bf2f(f2bf(x2[i]));
    }
  } else {
    unpack8(*reinterpret_cast<const uint4*>(row + j * 8), x1);
    unpack8(*reinterpret_cast<const uint4*>(row + 64 + j * 8), x2);
  }
  if (hs < n_q + n_kv) {
    const float* cs = cos_sin + (size_t)pos * HD + j * 8;
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs), c1 = *reinterpret_cast<const f32x4*>(cs + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + 64), s1 = *reinterpret_cast<const f32x4*>(cs + 68);
    float y1[8], y2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float c = i < 4 ? c0[i & 3] : c1[i & 3], sn = i < 4 ? s0[i & 3] : s1[i & 3];
      y1[i] = bf2f(f2bf(x1[i] * c - x2[i] * sn));
      y2[i] = bf2f(f2bf(x2[i] * c + x1[i] * sn));
    }
    const uint4 o1 = pack8(y1), o2 = pack8(y2);
    *reinterpret_cast<uint4*>(row + j * 8) = o1;
    *reinterpret_cast<uint4*>(row + 64 + j * 8) = o2;
    if (hs >= n_q && cache_ok) {
      const size_t off = ((size_t)(slot / BLK) * n_kv + (hs - n_q)) * BLK_ELEMS + (slot % BLK) * HD + j * 8;
      if (KV8) {
        uint2 p1, p2;
        p1.x = f2fp8(y1[0]) | (f2fp8(y1[1]) << 8) | (f2fp8(y1[2]) << 16) | ((uint32_t)f2fp8(y1[3]) << 24);
        p1.y = f2fp8(y1[4]) | (f2fp8(y1[5]) << 8) | (f2fp8(y1[6]) << 16) | ((uint32_t)f2fp8(y1[7]) << 24);
        p2.x = f2fp8(y2[0]) | (f2fp8(y2[1]) << 8) | (f2fp8(y2[2]) << 16) | ((uint32_t)f2fp8(y2[3]) << 24);
        p2.y = f2fp8(y2[4]) | (f2fp8(y2[5]) << 8) | (f2fp8(y2[6]) << 16) | ((uint32_t)f2fp8(y2[7]) << 24);
        *reinterpret_cast<uint2*>(kc8 + off) = p1;
        *reinterpret_cast<uint2*>(kc8 + off + 64) = p2;
      } else {
        *reinterpret_cast<uint4*>(kc + off) = o1;
        *reinterpret_cast<uint4*>(kc + off + 64) = o2;
      }
    }
  } else {
    if (PART) {                                    // keep the qkv buffer complete (the v columns are not rotated)
      *reinterpret_cast<uint4*>(row + j * 8) = pack8(x1);
      *reinterpret_cast<uint4*>(row + 64 + j * 8) = pack8(x2);
    }
    if (!cache_ok) return;
    const size_t off = ((size_t)(slot / BLK) * n_kv + (hs - n_q - n_kv)) * BLK_ELEMS + (slot % BLK);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d1 = j * 8 + i, d2 = 64 + j * 8 + i;
      if (KV8) {
        vc8[off + d1 * BLK] = f2fp8(x1[i]);
        vc8[off + d2 * BLK] = f2fp8(x2[i]);
      } else {
        vc[off + d1 * BLK] = f2bf(x1[i]);
        vc[off + d2 * BLK] = f2bf(x2[i]);
      }
    }
  }
}

}  // namespace

extern "C" {

int klm_rmsnorm(const void* x, void* residual, const void* weight, void* out, int T, int H, float eps,
                void* stream) {
  if (T <= 0 || H <= 0 || H % 8 != 0 || H > 16384) return KLM_ERR_SHAPE;
  if (!x || !weight || !out) return KLM_ERR_ARG;
  if (!aligned16(x) || !aligned16(weight) || !aligned16(out) || !aligned16(residual)) return KLM_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(T), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)residual, (const uint16_t*)weight,
                       (uint16_t*)out, H, eps);
  };
  if (H <= 8192)
    residual ? go(rmsnorm_kernel<true, 4>) : go(rmsnorm_kernel<false, 4>);
  else
    residual ? go(rmsnorm_kernel<true, 8>) : go(rmsnorm_kernel<false, 8>);
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

static int rope_cache_launch(void* qkv, const float* cos_sin, const int32_t* positions, const int32_t* slots, void* k_cache,
                             void* v_cache, int T, int n_q, int n_kv, int max_pos, int64_t num_slots, int kv_fp8,
                             const float* part, int ks, void* stream) {
  if (T <= 0 || n_q <= 0 || n_kv <= 0 || max_pos <= 0 || num_slots <= 0 || num_slots > INT32_MAX)
    return KLM_ERR_SHAPE;
  if (!qkv || !cos_sin || !positions || !slots || !k_cache || !v_cache) return KLM_ERR_ARG;
  if (!aligned16(qkv) || !aligned16(cos_sin) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(part)) return KLM_ERR_ALIGN;
  if (part && ks <= 0) return KLM_ERR_SHAPE;
  const long pairs = (long)T * (n_q + 2 * n_kv);       // 8 lanes per (token, head): 32 pairs per 256-thread block
  const long blocks = (pairs + 31) / 32;
  if (blocks > INT32_MAX) return KLM_ERR_SHAPE;
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (uint16_t*)qkv, cos_sin,
                       positions, slots, k_cache, v_cache, T, n_q, n_kv, max_pos, (int)num_slots, part, ks);
  };
  if (part)
    kv_fp8 ? go(rope_cache_kernel<true, true>) : go(rope_cache_kernel<false, true>);
  else
    kv_fp8 ? go(rope_cache_kernel<true, false>) : go(rope_cache_kernel<false, false>);
  return launch_status();
}

int klm_rope_cache(void* qkv, const float* cos_sin, const int32_t* positions, const int32_t* slots,
                   void* k_cache, void* v_cache, int T, int n_q, int n_kv, int max_pos, int64_t num_slots,
                   int kv_fp8, void* stream) {
  return rope_cache_launch(qkv, cos_sin, positions, slots, k_cache, v_cache, T, n_q, n_kv, max_pos, num_slots, kv_fp8, nullptr, 0,
                           stream);
}

// The same, fed by the fp32 K-split partials of the qkv projection (llm_mid.hip) instead of a finished qkv tensor:
// part[ks][T][(n_q + 2 n_kv) * 128]. Sums the splits, rotates, fills `qkv` and the caches: one launch instead of two.
int klm_rope_cache_from_partials(const float* part, int ksplit, void* qkv, const float* cos_sin, const int32_t* positions,
                                 const int32_t* slots, void* k_cache, void* v_cache, int T, int n_q, int n_kv, int max_pos,
                                 int64_t num_slots, int kv_fp8, void* stream) {
  if (!part) return KLM_ERR_ARG;
  return rope_cache_launch(qkv, cos_sin, positions, slots, k_cache, v_cache, T, n_q, n_kv, max_pos, num_slots, kv_fp8, part,
                           ksplit, stream);
}

}  // extern "C"
