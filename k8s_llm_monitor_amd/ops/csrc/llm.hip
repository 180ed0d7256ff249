// LLM serving kernels for MI355X (gfx950, CDNA4): the GPU side of the vLLM-compatible
// load generator (k8s_llm_monitor_amd/serving), used to validate the dashboard's vLLM panel
// (TTFT, tokens/s, KV-cache usage) against a server whose ground truth is known.
//
// No reference counterpart: vLLM appears in the reference only as a README bullet
// (README.md:73; SURVEY.md 2.4 N7, BASELINE.json config 4).
//
// All tensors are bf16, math is fp32. head_dim is 128, a KV block holds 16 tokens.
//
//   K cache  [num_blocks][n_kv][16 tokens][128 dims]   one token row = 256 contiguous bytes
//   V cache  [num_blocks][n_kv][128 dims][16 tokens]   TRANSPOSED: the 16 tokens of one dim are
//                                                      32 contiguous bytes
//
// The V layout is what makes the attention kernels free of LDS traffic and of cross-lane
// data movement: with 64-wide waves and v_mfma_f32_16x16x32_bf16 (first operand X: lane
// holds row lane&15, k = (lane>>4)*8..+8; result D[x_row = (lane>>4)*4+i][y_row = lane&15]):
//
//   S^T tile = mfma(X = K tile (16 tokens x 32 dims), Y = Q (16 query rows x 32 dims))
//       lane (fr, fq) receives, for query row fr, the scores of token rows fq*4+i.
//       K tile T of a 32-token step holds token (r/4)*8 + T*4 + r%4 in its row r, so the two
//       tiles together give lane (fr, fq) the scores of the 8 CONSECUTIVE tokens fq*8 .. fq*8+7,
//   O^T tile = mfma(X = V^T tile (16 dims x 32 tokens), Y = P (16 query rows x 32 tokens))
//       needs, in lane (fr, fq), P[row fr][tokens fq*8..+8]  -> exactly the registers the
//       softmax just produced, and V^T[dim fr][tokens fq*8..+8] -> one 16-byte load from the
//       transposed cache.
//
// A query "row" is a (token, head-of-the-GQA-group) pair: one kv head serves G = n_q/n_kv
// query heads, so a 16-row tile is 16/G tokens x G heads and K/V are read once per group.
// Decode (one token per sequence) uses G of the 16 rows; the kernel is HBM-bound and the
// MFMA time is noise (16 MFMAs per 16 KiB of cache read).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klm_ops.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

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

// ------------------------------------------------------------------------------------
// Attention over the paged cache.
// ------------------------------------------------------------------------------------
struct KvView {
  const uint16_t* kc;   // K cache + kv head offset (bf16 cache; an fp8 cache is addressed through kc8 / vc8)
  const uint16_t* vc;   // V cache + kv head offset
  const uint8_t* kc8;
  const uint8_t* vc8;
  const int* bt;        // block table row of this sequence
  size_t blk_stride;    // n_kv * BLK_ELEMS, in elements
  int last_blk;         // index (in the table) of the last block that holds a token
  int nb_total;         // blocks in the cache (ids are clamped to it)
};

struct Frags {
  bf16x8 k[2][4];
  bf16x8 v[8];
};

// KV8 (e4m3 cache): a lane's 16-byte K load carries 16 dims = two MFMA k-steps, so the k order of the
// contraction is dims (h2*4+fq)*16 + half*8 .. +8 for k-step 2*h2+half (load_q<true> orders q alike);
// V^T rows are 16 bytes per block, a lane takes its 8 tokens with one 8-byte load. Both are turned
// into bf16 fragments in registers: half the cache bytes per token, the same MFMAs.
template <bool KV8>
__device__ __forceinline__ void load_step(const KvView& kv, int step, int fr, int fq, Frags& f) {
  int i0 = step * 2, i1 = step * 2 + 1;
  i0 = i0 > kv.last_blk ? kv.last_blk : i0;
  i1 = i1 > kv.last_blk ? kv.last_blk : i1;
  int b0 = kv.bt[i0], b1 = kv.bt[i1];
  b0 = b0 < 0 ? 0 : (b0 >= kv.nb_total ? kv.nb_total - 1 : b0);
  b1 = b1 < 0 ? 0 : (b1 >= kv.nb_total ? kv.nb_total - 1 : b1);
  if (KV8) {
    const uint8_t* kp8 = kv.kc8 + (size_t)((fr >> 3) ? b1 : b0) * kv.blk_stride +
                         ((((fr >> 2) & 1) << 3) + (fr & 3)) * HD + fq * 16;
    const uint8_t* vp8 = kv.vc8 + (size_t)((fq >> 1) ? b1 : b0) * kv.blk_stride + fr * BLK + (fq & 1) * 8;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const uint4 raw = *reinterpret_cast<const uint4*>(kp8 + t * 4 * HD + h2 * 64);
        f.k[t][2 * h2] = fp8x8_to_bf16(raw.x, raw.y);
        f.k[t][2 * h2 + 1] = fp8x8_to_bf16(raw.z, raw.w);
      }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const uint2 raw = *reinterpret_cast<const uint2*>(vp8 + n * 16 * BLK);
      f.v[n] = fp8x8_to_bf16(raw.x, raw.y);
    }
    return;
  }
  const uint16_t* kp = kv.kc + (size_t)((fr >> 3) ? b1 : b0) * kv.blk_stride +
                       ((((fr >> 2) & 1) << 3) + (fr & 3)) * HD + fq * 8;
  const uint16_t* vp = kv.vc + (size_t)((fq >> 1) ? b1 : b0) * kv.blk_stride + fr * BLK + (fq & 1) * 8;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      f.k[t][s] = *reinterpret_cast<const bf16x8*>(kp + t * 4 * HD + s * 32);
#pragma unroll
  for (int n = 0; n < 8; ++n) f.v[n] = *reinterpret_cast<const bf16x8*>(vp + n * 16 * BLK);
}

struct Acc {
  f32x4 o[8];
  float m, l;
};

// One 32-token step of online softmax attention for the wave's 16 query rows.
// qpos: position of this lane's query row (-1 = row not in use); keys at positions <= qpos count.
// lim (wave-uniform): every row IN USE has qpos >= lim, so a step whose last key is <= lim needs no
// mask at all (rows not in use carry q = 0: their scores stay finite and their output is dropped).
// The softmax keeps m in the scaled log2 domain and costs one max, one FMA and one exp2 per score.
__device__ __forceinline__ void qk_scores(const bf16x8 (&k)[2][4], const bf16x8 (&qf)[4], f32x4& s0, f32x4& s1) {
  s0 = f32x4{0.f, 0.f, 0.f, 0.f};
  s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[0][s], qf[s], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[1][s], qf[s], s1, 0, 0, 0);
  }
}

__device__ __forceinline__ void softmax_pv(const f32x4& s0, const f32x4& s1, const bf16x8 (&v)[8], int step, int fq,
                                           int qpos, int lim, float scale_log2, Acc& a) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[j] = s0[j];
    x[j + 4] = s1[j];
  }
  if (step * 32 + 31 > lim) {                      // wave-uniform: only the steps on the causal edge
    const int tk0 = step * 32 + fq * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (tk0 + j <= qpos) ? x[j] : -INFINITY;
  }
  float mx = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float m_new = fmaxf(a.m, mx * scale_log2);  // scale > 0; -inf stays -inf
  const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
  const float alpha = fast_exp2(a.m - m_use);
  bf16x8 p;
  float rs = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float pj = fast_exp2(fmaf(x[j], scale_log2, -m_use));
    p[j] = (__bf16)pj;
    rs += pj;
  }
  rs += __shfl_xor(rs, 16);
  rs += __shfl_xor(rs, 32);
  a.l = a.l * alpha + rs;
  a.m = m_new;
  // the running maximum settles after the first few steps: skip the 32 rescaling multiplies
  // whenever no row of the wave moved its maximum (wave-uniform branch)
  if (__ballot(alpha != 1.f) != 0) {
#pragma unroll
    for (int n = 0; n < 8; ++n) a.o[n] *= alpha;
  }
#pragma unroll
  for (int n = 0; n < 8; ++n) a.o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[n], p, a.o[n], 0, 0, 0);
}

__device__ __forceinline__ void attend_step(const Frags& f, const bf16x8 (&qf)[4], int step, int fq,
                                            int qpos, int lim, float scale_log2, Acc& a) {
  f32x4 s0, s1;
  qk_scores(f.k, qf, s0, s1);
  softmax_pv(s0, s1, f.v, step, fq, qpos, lim, scale_log2, a);
}

template <bool KV8 = false>
__device__ __forceinline__ void load_q(const uint16_t* qrow, bool valid, int fq, bf16x8 (&qf)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (valid) {
      // the dims of k-step s must be the ones load_step puts there (see load_step<true>)
      qf[s] = *reinterpret_cast<const bf16x8*>(qrow + (KV8 ? (s >> 1) * 64 + fq * 16 + (s & 1) * 8 : s * 32 + fq * 8));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (__bf16)0.f;
    }
  }
}

// Runs steps st0, st0+stride, ... < st_end with the loads of the next step in flight.
template <bool KV8>
__device__ __forceinline__ void attend_range(const KvView& kv, const bf16x8 (&qf)[4], int st0, int st_end,
                                             int stride, int fr, int fq, int qpos, int lim, float scale_log2,
                                             Acc& a) {
  if (st0 >= st_end) return;
  Frags cur;
  load_step<KV8>(kv, st0, fr, fq, cur);
  for (int st = st0; st < st_end; st += stride) {
    Frags nxt;
    const int nx = st + stride;
    load_step<KV8>(kv, nx < st_end ? nx : st, fr, fq, nxt);   // last iteration re-reads (cached), unused
    attend_step(cur, qf, st, fq, qpos, lim, scale_log2, a);
    cur = nxt;
  }
}

constexpr int OPAD = HD + 4;

// Decode: one token per sequence. Block = (kv head, sequence, split); its 4 waves take the
// 32-token steps of the split round-robin and are combined through LDS.
template <bool KV8>
__global__ __launch_bounds__(256) void paged_attn_decode_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    const int* __restrict__ block_tables, const int* __restrict__ ctx_lens, uint16_t* __restrict__ out,
    float* __restrict__ tmp_o, float* __restrict__ tmp_ml, int q_stride, int n_q, int n_kv, int G,
    int max_blocks, int nb_total, int num_splits, float scale_log2) {
  __shared__ float sm_o[4][16][OPAD];
  __shared__ float sm_m[4][16], sm_l[4][16];
  const int h = blockIdx.x, b = blockIdx.y, split = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int ctx = ctx_lens[b];
  ctx = ctx < 0 ? 0 : (ctx > max_blocks * BLK ? max_blocks * BLK : ctx);
  const int total_steps = (ctx + 31) >> 5;
  const int sps = (total_steps + num_splits - 1) / num_splits;
  const int s_begin = split * sps;
  const int s_end = (s_begin + sps) < total_steps ? (s_begin + sps) : total_steps;

  KvView kv;
  kv.kc = kc + (size_t)h * BLK_ELEMS;
  kv.vc = vc + (size_t)h * BLK_ELEMS;
  kv.kc8 = reinterpret_cast<const uint8_t*>(kc) + (size_t)h * BLK_ELEMS;
  kv.vc8 = reinterpret_cast<const uint8_t*>(vc) + (size_t)h * BLK_ELEMS;
  kv.bt = block_tables + (size_t)b * max_blocks;
  kv.blk_stride = (size_t)n_kv * BLK_ELEMS;
  kv.last_blk = ctx > 0 ? (ctx - 1) / BLK : 0;
  kv.nb_total = nb_total;

  const bool valid = fr < G;
  bf16x8 qf[4];
  load_q<KV8>(q + (size_t)b * q_stride + ((size_t)h * G + (valid ? fr : 0)) * HD, valid, fq, qf);
  const int qpos = valid ? ctx - 1 : -1;

  Acc a;
#pragma unroll
  for (int n = 0; n < 8; ++n) a.o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  a.m = -INFINITY;
  a.l = 0.f;
  attend_range<KV8>(kv, qf, s_begin + w, s_end, 4, fr, fq, qpos, ctx - 1, scale_log2, a);

#pragma unroll
  for (int n = 0; n < 8; ++n) *reinterpret_cast<f32x4*>(&sm_o[w][fr][n * 16 + fq * 4]) = a.o[n];
  if (fq == 0) {
    sm_m[w][fr] = a.m;
    sm_l[w][fr] = a.l;
  }
  __syncthreads();

  const int row = tid >> 4, d0 = (tid & 15) * 8;
  if (row < G) {
    float mt = fmaxf(fmaxf(sm_m[0][row], sm_m[1][row]), fmaxf(sm_m[2][row], sm_m[3][row]));
    const float mu = (mt == -INFINITY) ? 0.f : mt;
    float L = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const float wgt = fast_exp2(sm_m[ww][row] - mu);
      L += sm_l[ww][row] * wgt;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += sm_o[ww][row][d0 + j] * wgt;
    }
    const size_t head = (size_t)b * n_q + (size_t)h * G + row;
    if (num_splits == 1) {
      const float inv = L > 0.f ? 1.f / L : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] *= inv;
      *reinterpret_cast<uint4*>(out + head * HD + d0) = pack8(o);
    } else {
      float* po = tmp_o + (head * num_splits + split) * HD + d0;
      *reinterpret_cast<f32x4*>(po) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(po + 4) = f32x4{o[4], o[5], o[6], o[7]};
      if ((tid & 15) == 0) {
        tmp_ml[(head * num_splits + split) * 2] = mt;
        tmp_ml[(head * num_splits + split) * 2 + 1] = L;
      }
    }
  }
}

// Merges the partial results of the splits: one 128-thread block per (sequence, query head).
// Only the splits that hold tokens are read (the split count is fixed per hipGraph, the context
// length is not); their (m, l) pairs are loaded by different threads and the dim loop runs four
// independent loads deep, so the kernel costs one memory round trip rather than one per split.
__global__ __launch_bounds__(128) void paged_attn_reduce_kernel(const float* __restrict__ tmp_o,
                                                                const float* __restrict__ tmp_ml,
                                                                const int* __restrict__ ctx_lens,
                                                                uint16_t* __restrict__ out, int n_q,
                                                                int max_blocks, int num_splits) {
  __shared__ float sw[256];
  __shared__ float red[4];
  const size_t head = blockIdx.x;
  const int d = threadIdx.x, wv = d >> 6;
  int ctx = ctx_lens[head / n_q];
  ctx = ctx < 0 ? 0 : (ctx > max_blocks * BLK ? max_blocks * BLK : ctx);
  const int total_steps = (ctx + 31) >> 5;
  const int sps = (total_steps + num_splits - 1) / num_splits;
  const int active = sps > 0 ? (total_steps + sps - 1) / sps : 0;
  const float* ml = tmp_ml + head * num_splits * 2;
  float m0 = -INFINITY, m1 = -INFINITY, l0 = 0.f, l1 = 0.f;
  if (d < active) { m0 = ml[d * 2]; l0 = ml[d * 2 + 1]; }
  if (d + 128 < active) { m1 = ml[(d + 128) * 2]; l1 = ml[(d + 128) * 2 + 1]; }
  float mt = fmaxf(m0, m1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mt = fmaxf(mt, __shfl_xor(mt, o));
  if ((d & 63) == 0) red[wv] = mt;
  __syncthreads();
  mt = fmaxf(red[0], red[1]);
  const float mu = (mt == -INFINITY) ? 0.f : mt;
  const float w0 = fast_exp2(m0 - mu), w1 = fast_exp2(m1 - mu);
  if (d < active) sw[d] = w0;
  if (d + 128 < active) sw[d + 128] = w1;
  float L = l0 * w0 + l1 * w1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) L += __shfl_xor(L, o);
  if ((d & 63) == 0) red[2 + wv] = L;
  __syncthreads();
  L = red[2] + red[3];
  const float* po = tmp_o + head * num_splits * HD + d;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int sidx = 0;
  for (; sidx + 4 <= active; sidx += 4) {
    a0 += po[(sidx + 0) * HD] * sw[sidx + 0];
    a1 += po[(sidx + 1) * HD] * sw[sidx + 1];
    a2 += po[(sidx + 2) * HD] * sw[sidx + 2];
    a3 += po[(sidx + 3) * HD] * sw[sidx + 3];
  }
  for (; sidx < active; ++sidx) a0 += po[sidx * HD] * sw[sidx];
  const float o = (a0 + a1) + (a2 + a3);
  out[head * HD + d] = f2bf(L > 0.f ? o / L : 0.f);
}

// Prefill / chunked prefill: the queries are the LAST q_len tokens of a context of ctx_len
// tokens that is already in the cache (rope_cache runs first). Block = (tile, kv head); a tile
// is 4 waves x QT x 16/G consecutive query tokens of one sequence. A wave keeps QT 16-row query
// tiles in registers and feeds every K / V fragment it loads to all of them (QT x the MFMA work
// per byte loaded: the kernel is bound by those fragment loads, not by the matrix pipe); each
// tile stops at its own causal limit, and there is no cross-wave exchange.
//   seq_info[s] = {q_start (row of the first query in q/out), q_len, ctx_len}
//   tiles[t]    = {sequence, first query token of the tile (0-based inside the sequence's queries)}
template <int QT, bool KV8>
__global__ __launch_bounds__(256) void paged_attn_prefill_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    const int* __restrict__ block_tables, const int* __restrict__ seq_info, const int* __restrict__ tiles,
    uint16_t* __restrict__ out, int n_seqs, int q_stride, int n_q, int n_kv, int G, int g_shift,
    int max_blocks, int nb_total, float scale_log2) {
  const int h = blockIdx.x, tile = blockIdx.y;   // heads fastest: the tile table is sorted heaviest first
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int seq = tiles[tile * 2];
  seq = seq < 0 ? 0 : (seq >= n_seqs ? n_seqs - 1 : seq);
  const int tok0 = tiles[tile * 2 + 1];
  const int q_start = seq_info[seq * 3], q_len = seq_info[seq * 3 + 1];
  int ctx = seq_info[seq * 3 + 2];
  ctx = ctx < 0 ? 0 : (ctx > max_blocks * BLK ? max_blocks * BLK : ctx);
  const int tpt = 16 >> g_shift;                    // query tokens per 16-row tile
  const int wtok0 = tok0 + w * tpt * QT;
  if (tok0 < 0 || wtok0 >= q_len || q_len > ctx) return;   // wave-uniform

  KvView kv;
  kv.kc = kc + (size_t)h * BLK_ELEMS;
  kv.vc = vc + (size_t)h * BLK_ELEMS;
  kv.kc8 = reinterpret_cast<const uint8_t*>(kc) + (size_t)h * BLK_ELEMS;
  kv.vc8 = reinterpret_cast<const uint8_t*>(vc) + (size_t)h * BLK_ELEMS;
  kv.bt = block_tables + (size_t)seq * max_blocks;
  kv.blk_stride = (size_t)n_kv * BLK_ELEMS;
  kv.last_blk = ctx > 0 ? (ctx - 1) / BLK : 0;
  kv.nb_total = nb_total;

  const size_t qhead = (size_t)h * G + (fr & (G - 1));
  bf16x8 qf[QT][4];
  Acc a[QT];
  int qpos[QT], nst[QT], lim[QT];
  size_t qrow[QT];
  bool valid[QT];
  int n_steps = 0;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int t0 = wtok0 + t * tpt;
    const int tq = t0 + (fr >> g_shift);
    valid[t] = tq < q_len;
    qrow[t] = (size_t)(q_start + (valid[t] ? tq : wtok0));
    load_q<KV8>(q + qrow[t] * q_stride + qhead * HD, valid[t], fq, qf[t]);
    qpos[t] = valid[t] ? ctx - q_len + tq : -1;
    lim[t] = ctx - q_len + t0;                       // position of the tile's first query
    int last_tq = t0 + tpt - 1;
    last_tq = last_tq < q_len ? last_tq : q_len - 1;
    nst[t] = t0 < q_len ? ((ctx - q_len + last_tq) >> 5) + 1 : 0;     // wave-uniform
    n_steps = nst[t] > n_steps ? nst[t] : n_steps;
#pragma unroll
    for (int n = 0; n < 8; ++n) a[t].o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    a[t].m = -INFINITY;
    a[t].l = 0.f;
  }

  Frags cur;
  load_step<KV8>(kv, 0, fr, fq, cur);
  for (int st = 0; st < n_steps; ++st) {
    Frags nxt;
    load_step<KV8>(kv, st + 1 < n_steps ? st + 1 : st, fr, fq, nxt);
#pragma unroll
    for (int t = 0; t < QT; ++t)
      if (st < nst[t]) attend_step(cur, qf[t], st, fq, qpos[t], lim[t], scale_log2, a[t]);
    cur = nxt;
  }

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (valid[t]) {
      const float inv = a[t].l > 0.f ? 1.f / a[t].l : 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (__bf16)(a[t].o[n][i] * inv);
        *reinterpret_cast<bf16x4*>(out + (qrow[t] * n_q + qhead) * HD + n * 16 + fq * 4) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Prefill for long prompts: the same math, K / V staged ONCE per block through an LDS ring.
// The register kernel above makes each of the 4 waves fetch the whole 16 KiB step itself with one
// step of lookahead: at long context it waits on L2 latency (about 2 us per step measured). Here
// the 4 waves of a block split the fetch (4 KiB each, LDS-DMA: no registers, no VALU), 3 steps
// stay in flight across one raw s_barrier per step (counted s_waitcnt vmcnt(8)), and every wave
// reads its MFMA fragments from LDS:
//   stage = K [32 tokens][256 B], 16-byte chunk c of token row r stored at chunk c ^ key(r),
//           key(r) = (r & 15) ^ ((r >> 4) << 2)  -> the 16 rows of a fragment read hit 16 different
//           chunk columns (conflict-free ds_read_b128); the swizzle is applied on the DMA source
//           address, the LDS image itself is lane-linear as the DMA writes it
//         | V [2 blocks][128 dims][16 tokens], a straight copy of the transposed cache blocks
// ------------------------------------------------------------------------------------
constexpr int PF_STAGES = 4;
constexpr int PF_STAGE_BYTES = 16384;

#define KLM_AS1 __attribute__((address_space(1)))
#define KLM_AS3 __attribute__((address_space(3)))
__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const KLM_AS1 void*)g, (KLM_AS3 void*)lds, 16, 0, 0);
}

__device__ __forceinline__ void pf_issue(const KvView& kv, int st, char* stage, int w, int lane) {
  int i0 = st * 2, i1 = st * 2 + 1;
  i0 = i0 > kv.last_blk ? kv.last_blk : i0;
  i1 = i1 > kv.last_blk ? kv.last_blk : i1;
  int b0 = kv.bt[i0], b1 = kv.bt[i1];
  b0 = b0 < 0 ? 0 : (b0 >= kv.nb_total ? kv.nb_total - 1 : b0);
  b1 = b1 < 0 ? 0 : (b1 >= kv.nb_total ? kv.nb_total - 1 : b1);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = 2 * w + j;                         // piece: K token rows 4p..4p+3, V bytes [1024 p, +1024)
    const size_t base = (size_t)(p >= 4 ? b1 : b0) * kv.blk_stride;
    const int row = p * 4 + (lane >> 4);
    const int key = (row & 15) ^ ((row >> 4) << 2);
    glds16(kv.kc + base + (row & 15) * HD + (((lane & 15) ^ key) << 3), stage + p * 1024);
    glds16(kv.vc + base + (p & 3) * 512 + lane * 8, stage + 8192 + p * 1024);
  }
}

__device__ __forceinline__ void pf_read_k(const char* stage, int fr, int fq, bf16x8 (&k)[2][4]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tok = ((fr >> 2) << 3) + t * 4 + (fr & 3);
    const int key = (tok & 15) ^ ((tok >> 4) << 2);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      k[t][s] = *reinterpret_cast<const bf16x8*>(stage + tok * 256 + (((s * 4 + fq) ^ key) << 4));
  }
}

__device__ __forceinline__ void pf_read_v(const char* stage, int fr, int fq, bf16x8 (&v)[8]) {
  const char* vp = stage + 8192 + (fq >> 1) * 4096 + fr * 32 + (fq & 1) * 16;
#pragma unroll
  for (int n = 0; n < 8; ++n) v[n] = *reinterpret_cast<const bf16x8*>(vp + n * 512);
}

template <int QT>
__global__ __launch_bounds__(256, QT <= 2 ? 2 : 1) void paged_attn_prefill_lds_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    const int* __restrict__ block_tables, const int* __restrict__ seq_info, const int* __restrict__ tiles,
    uint16_t* __restrict__ out, int n_seqs, int q_stride, int n_q, int n_kv, int G, int g_shift,
    int max_blocks, int nb_total, float scale_log2) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int h = blockIdx.x, tile = blockIdx.y;   // heads fastest: the tile table is sorted heaviest first
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int seq = tiles[tile * 2];
  seq = seq < 0 ? 0 : (seq >= n_seqs ? n_seqs - 1 : seq);
  const int tok0 = tiles[tile * 2 + 1];
  const int q_start = seq_info[seq * 3], q_len = seq_info[seq * 3 + 1];
  int ctx = seq_info[seq * 3 + 2];
  ctx = ctx < 0 ? 0 : (ctx > max_blocks * BLK ? max_blocks * BLK : ctx);
  const int tpt = 16 >> g_shift;
  if (tok0 < 0 || tok0 >= q_len || q_len > ctx) return;          // block-uniform
  const int wtok0 = tok0 + w * tpt * QT;

  KvView kv;
  kv.kc = kc + (size_t)h * BLK_ELEMS;
  kv.vc = vc + (size_t)h * BLK_ELEMS;
  kv.kc8 = reinterpret_cast<const uint8_t*>(kc) + (size_t)h * BLK_ELEMS;
  kv.vc8 = reinterpret_cast<const uint8_t*>(vc) + (size_t)h * BLK_ELEMS;
  kv.bt = block_tables + (size_t)seq * max_blocks;
  kv.blk_stride = (size_t)n_kv * BLK_ELEMS;
  kv.last_blk = ctx > 0 ? (ctx - 1) / BLK : 0;
  kv.nb_total = nb_total;

  int blk_last = tok0 + 4 * QT * tpt - 1;
  blk_last = blk_last < q_len ? blk_last : q_len - 1;
  const int n_steps = ((ctx - q_len + blk_last) >> 5) + 1;        // block-uniform: every wave loads and syncs

  const size_t qhead = (size_t)h * G + (fr & (G - 1));
  bf16x8 qf[QT][4];
  Acc a[QT];
  int qpos[QT], nst[QT], lim[QT];
  size_t qrow[QT];
  bool valid[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int t0 = wtok0 + t * tpt;
    const int tq = t0 + (fr >> g_shift);
    valid[t] = tq < q_len;
    qrow[t] = (size_t)(q_start + (valid[t] ? tq : tok0));
    load_q(q + qrow[t] * q_stride + qhead * HD, valid[t], fq, qf[t]);
    qpos[t] = valid[t] ? ctx - q_len + tq : -1;
    lim[t] = ctx - q_len + t0;                       // position of the tile's first query
    int last_tq = t0 + tpt - 1;
    last_tq = last_tq < q_len ? last_tq : q_len - 1;
    nst[t] = t0 < q_len ? ((ctx - q_len + last_tq) >> 5) + 1 : 0;
#pragma unroll
    for (int n = 0; n < 8; ++n) a[t].o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    a[t].m = -INFINITY;
    a[t].l = 0.f;
  }

#pragma unroll
  for (int sidx = 0; sidx < PF_STAGES - 1; ++sidx)
    pf_issue(kv, sidx < n_steps ? sidx : n_steps - 1, ring + sidx * PF_STAGE_BYTES, w, lane);
  for (int st = 0; st < n_steps; ++st) {
    // 4 DMA instructions per wave and step, issued in order: at most the 2 newer steps may be pending
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nx = st + PF_STAGES - 1;
    pf_issue(kv, nx < n_steps ? nx : n_steps - 1, ring + (nx & (PF_STAGES - 1)) * PF_STAGE_BYTES, w, lane);
    bool any = false;
#pragma unroll
    for (int t = 0; t < QT; ++t) any = any || (st < nst[t]);
    if (any) {
      Frags f;
      const char* stage = ring + (st & (PF_STAGES - 1)) * PF_STAGE_BYTES;
      pf_read_k(stage, fr, fq, f.k);
      pf_read_v(stage, fr, fq, f.v);
#pragma unroll
      for (int t = 0; t < QT; ++t)
        if (st < nst[t]) attend_step(f, qf[t], st, fq, qpos[t], lim[t], scale_log2, a[t]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant tail DMAs before the LDS goes away

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (valid[t]) {
      const float inv = a[t].l > 0.f ? 1.f / a[t].l : 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (__bf16)(a[t].o[n][i] * inv);
        *reinterpret_cast<bf16x4*>(out + (qrow[t] * n_q + qhead) * HD + n * 16 + fq * 4) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Skinny linear layer for decode at small batch (M <= 16 rows): out[M][N] = x[M][K] . W[N][K]^T.
// The weight matrix is streamed from HBM exactly once, straight into MFMA operand registers
// (no LDS staging): a block owns 16 output columns (16 rows of W), its waves split K, every
// wave issues 16-byte loads of W[n0 + lane&15][k + (lane>>4)*8 ..] and of the matching x
// fragment (L1/L2 resident, M*K*2 bytes), and one v_mfma_f32_16x16x32_bf16 per 1 KiB of W keeps
// the math invisible next to the memory time. There is NO per-element VALU work in the loop:
//   RMS      : the row's sum of squares is a second MFMA, x . x^T, whose diagonal is read at the
//              end; the product is then scaled by rsqrt(mean(x^2) + eps). The norm's gain is
//              folded into W's columns when the model is built.
//   EPI_RES  : out = res + product  (the residual add; out may alias res)
//   EPI_SWIGLU: W rows are gate/up interleaved by 8 (see silu_mul_kernel), the tile's 8 gate and
//              8 up columns meet through one cross-lane exchange: out[M][N/2] = silu(g) * u
// so a decoder layer at small batch is 4 of these + rope + attention. The K split is reduced
// through LDS (deterministic, no atomics).
// ------------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_RES = 1, EPI_SWIGLU = 2 };



// RT = 16-row tiles of W per block: the x fragment is loaded once per RT weight fragments, which
// matters when M is large enough for the x traffic (M*K*2 bytes per block, from L2) to rival the
// block's weight traffic (16*RT*K*2 bytes).
//
// W8: the weights are OCP e4m3 bytes with one fp32 scale per output row (W8A16). A lane's 16-byte
// load then carries 16 k-values; v_cvt_scalef32_pk_bf16_fp8 turns them into two bf16 MFMA fragments
// in registers (8 VALU instructions per KiB-instruction of weights: still far from the memory time),
// x stays bf16, and the row scale is applied to the fp32 result. Half the bytes per token.
template <bool RMS, int EPI, int RT, bool W8>
__global__ __launch_bounds__(RT == 1 ? 1024 : 512) void skinny_linear_kernel(
    const uint16_t* __restrict__ x, int x_stride, const void* __restrict__ Wv, const float* __restrict__ wscale,
    uint16_t* out, const uint16_t* res, int M, int N, int K, int kslice, float eps, int tiled) {
  constexpr int MAXW = RT == 1 ? 16 : 8;
  __shared__ f32x4 part[MAXW][RT][64];
  __shared__ float ssq[MAXW][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * 16 * RT;
  const int k0 = w * kslice + fq * (W8 ? 16 : 8);
  // Weight layout. Row-major [N][K]: a lane's 16 bytes sit at row n0+fr, k offset k0, and the next
  // MFMA step is 64 bytes further along the row. Tiled (`tiled` != 0): the 1 KiB that one wave
  // instruction loads (16 rows x 64 bytes) is stored contiguously in lane order, tiles ordered
  // [row tile][k step], so a wave streams one contiguous run of kslice*16 elements - the access
  // pattern HBM likes best. Both layouts have the same distance between row tiles (16*K elements).
  const size_t wlane = tiled ? ((size_t)(n0 >> 4) * K + (size_t)w * kslice) * 16 + lane * (W8 ? 16 : 8)
                             : (size_t)(n0 + fr) * K + k0;
  const int wstep = tiled ? (W8 ? 1024 : 512) : (W8 ? 64 : 32);    // elements between consecutive MFMA steps
  const uint16_t* wp = reinterpret_cast<const uint16_t*>(Wv) + wlane;      // bf16
  const uint8_t* wp8 = reinterpret_cast<const uint8_t*>(Wv) + wlane;       // fp8
  const size_t tile_stride = (size_t)16 * K;
  const bool valid = fr < M;
  const uint16_t* xp = x + (size_t)(valid ? fr : 0) * x_stride + k0;
  f32x4 acc[RT], sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 zero;
#pragma unroll
  for (int j = 0; j < 8; ++j) zero[j] = (__bf16)0.f;

#define KLM_SKINNY_CHUNK(c)                                                                        \
  {                                                                                                \
    bf16x8 wf[RT][4], xf[4];                                                                       \
    _Pragma("unroll") for (int r = 0; r < RT; ++r)                                                 \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                \
        wf[r][s] = *reinterpret_cast<const bf16x8*>(wp + r * tile_stride + ((c) / 32 + s) * wstep); \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                  \
        xf[s] = valid ? *reinterpret_cast<const bf16x8*>(xp + (c) + s * 32) : zero;                \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                \
      _Pragma("unroll") for (int r = 0; r < RT; ++r)                                               \
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[r][s], xf[s], acc[r], 0, 0, 0);        \
      if (RMS) sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[s], xf[s], sq, 0, 0, 0);            \
    }                                                                                              \
  }
  // fp8 weights: 64 k-values per step; the lane's 16 bytes become fragments A0 (values 0..7) and A1
  // (8..15), matched by the two 16-byte x loads at the same k offsets (any k order is fine for a
  // contraction as long as both operands use it)
#define KLM_SKINNY_CHUNK8(c)                                                                       \
  {                                                                                                \
    uint4 raw[RT];                                                                                 \
    _Pragma("unroll") for (int r = 0; r < RT; ++r)                                                 \
        raw[r] = *reinterpret_cast<const uint4*>(wp8 + r * tile_stride + ((c) / 64) * wstep);       \
    const bf16x8 x0 = valid ? *reinterpret_cast<const bf16x8*>(xp + (c)) : zero;                   \
    const bf16x8 x1 = valid ? *reinterpret_cast<const bf16x8*>(xp + (c) + 8) : zero;               \
    _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                               \
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16(raw[r].x, raw[r].y), x0, acc[r], 0, 0, 0); \
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16(raw[r].z, raw[r].w), x1, acc[r], 0, 0, 0); \
    }                                                                                              \
    if (RMS) {                                                                                     \
      sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, x0, sq, 0, 0, 0);                           \
      sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, x1, sq, 0, 0, 0);                           \
    }                                                                                              \
  }
  int c = 0;
  if (W8) {
    if (RT == 1) {
      for (; c + 512 <= kslice; c += 512) {          // 8 weight loads in flight per wave, as in the bf16 loop
        KLM_SKINNY_CHUNK8(c)
        KLM_SKINNY_CHUNK8(c + 64)
        KLM_SKINNY_CHUNK8(c + 128)
        KLM_SKINNY_CHUNK8(c + 192)
        KLM_SKINNY_CHUNK8(c + 256)
        KLM_SKINNY_CHUNK8(c + 320)
        KLM_SKINNY_CHUNK8(c + 384)
        KLM_SKINNY_CHUNK8(c + 448)
      }
      for (; c + 256 <= kslice; c += 256) {
        KLM_SKINNY_CHUNK8(c)
        KLM_SKINNY_CHUNK8(c + 64)
        KLM_SKINNY_CHUNK8(c + 128)
        KLM_SKINNY_CHUNK8(c + 192)
      }
    }
    for (; c + 128 <= kslice; c += 128) {
      KLM_SKINNY_CHUNK8(c)
      KLM_SKINNY_CHUNK8(c + 64)
    }
  } else {
    if (RT == 1) {
      for (; c + 256 <= kslice; c += 256) {
        KLM_SKINNY_CHUNK(c)
        KLM_SKINNY_CHUNK(c + 128)
      }
    }
    for (; c < kslice; c += 128) KLM_SKINNY_CHUNK(c)
  }
#undef KLM_SKINNY_CHUNK
#undef KLM_SKINNY_CHUNK8

#pragma unroll
  for (int r = 0; r < RT; ++r) part[w][r][lane] = acc[r];
  if (RMS) {
    // sq = x . x^T: D[row fq*4+i][col fr]; the diagonal element of row fr sits in lane (fr, fr/4), i = fr%4
    const int i = fr & 3;
    const float d = i == 0 ? sq[0] : (i == 1 ? sq[1] : (i == 2 ? sq[2] : sq[3]));
    if (fq == (fr >> 2)) ssq[w][fr] = d;
  }
  __syncthreads();
  // wave r of the block finishes row tile r (RT <= number of waves is checked by the launcher)
  if (w < RT) {
    f32x4 t = part[0][w][lane];
    for (int ww = 1; ww < nw; ++ww) t += part[ww][w][lane];
    if (RMS) {
      float tot = 0.f;
      for (int ww = 0; ww < nw; ++ww) tot += ssq[ww][fr];
      t *= rsqrtf(tot / (float)K + eps);
    }
    const int nt = n0 + w * 16;
    if (W8) {
      const f32x4 ws = *reinterpret_cast<const f32x4*>(wscale + nt + fq * 4);
      t *= ws;
    }
    if (EPI == EPI_SWIGLU) {
      // lanes fq 0,1 hold gate columns fq*4+i, lanes fq 2,3 the matching up columns
      f32x4 u;
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = __shfl_xor(t[i], 32);
      if (valid && fq < 2) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (__bf16)(t[i] / (1.f + __expf(-t[i])) * u[i]);
        *reinterpret_cast<bf16x4*>(out + (size_t)fr * (N >> 1) + (nt >> 1) + fq * 4) = o;
      }
    } else if (valid) {
      const size_t idx = (size_t)fr * N + nt + fq * 4;
      if (EPI == EPI_RES) {
        const uint2 ra = *reinterpret_cast<const uint2*>(res + idx);
        t[0] += __uint_as_float(ra.x << 16);
        t[1] += __uint_as_float(ra.x & 0xffff0000u);
        t[2] += __uint_as_float(ra.y << 16);
        t[3] += __uint_as_float(ra.y & 0xffff0000u);
      }
      bf16x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (__bf16)t[i];
      *reinterpret_cast<bf16x4*>(out + idx) = o;
    }
  }
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

int klm_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                          const int32_t* ctx_lens, void* out, float* tmp_o, float* tmp_ml, int B, int q_stride,
                          int n_q, int n_kv, int max_blocks, int num_blocks_total, int num_splits, float scale,
                          int kv_fp8, void* stream) {
  if (q_stride < n_q * HD || q_stride % 8 != 0) return KLM_ERR_SHAPE;
  if (B <= 0 || B > 65535 || n_q <= 0 || n_kv <= 0 || n_q % n_kv != 0 || max_blocks <= 0 || num_blocks_total <= 0)
    return KLM_ERR_SHAPE;
  const int G = n_q / n_kv;
  if (group_shift(G) < 0) return KLM_ERR_SHAPE;
  if (num_splits < 1 || num_splits > 256) return KLM_ERR_ARG;
  if (!q || !k_cache || !v_cache || !block_tables || !ctx_lens || !out) return KLM_ERR_ARG;
  if (num_splits > 1 && (!tmp_o || !tmp_ml)) return KLM_ERR_ARG;
  if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out) || !aligned16(tmp_o))
    return KLM_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float scale_log2 = scale * 1.4426950408889634f;
  if (kv_fp8)
    hipLaunchKernelGGL(paged_attn_decode_kernel<true>, dim3(n_kv, B, num_splits), dim3(256), 0, s, (const uint16_t*)q,
                       (const uint16_t*)k_cache, (const uint16_t*)v_cache, block_tables, ctx_lens, (uint16_t*)out,
                       tmp_o, tmp_ml, q_stride, n_q, n_kv, G, max_blocks, num_blocks_total, num_splits, scale_log2);
  else
    hipLaunchKernelGGL(paged_attn_decode_kernel<false>, dim3(n_kv, B, num_splits), dim3(256), 0, s, (const uint16_t*)q,
                       (const uint16_t*)k_cache, (const uint16_t*)v_cache, block_tables, ctx_lens, (uint16_t*)out,
                       tmp_o, tmp_ml, q_stride, n_q, n_kv, G, max_blocks, num_blocks_total, num_splits, scale_log2);
  if (hipGetLastError() != hipSuccess) return KLM_ERR_HIP;
  if (num_splits > 1)
    hipLaunchKernelGGL(paged_attn_reduce_kernel, dim3((unsigned)B * n_q), dim3(128), 0, s, tmp_o, tmp_ml, ctx_lens,
                       (uint16_t*)out, n_q, max_blocks, num_splits);
  return launch_status();
}

int klm_paged_attn_prefill(const void* q, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                           const int32_t* seq_info, const int32_t* tiles, void* out, int n_tiles, int n_seqs,
                           int q_stride, int n_q, int n_kv, int max_blocks, int num_blocks_total, float scale,
                           int q_tiles, int variant, int kv_fp8, void* stream) {
  if ((q_tiles != 1 && q_tiles != 2 && q_tiles != 4) || variant < 0 || variant > 1) return KLM_ERR_VARIANT;
  if (kv_fp8) variant = 0;        // the LDS-ring kernel reads a bf16 cache only
  if (q_stride < n_q * HD || q_stride % 8 != 0) return KLM_ERR_SHAPE;
  if (n_tiles <= 0 || n_tiles > 65535 || n_seqs <= 0 || n_q <= 0 || n_kv <= 0 || n_q % n_kv != 0 ||
      max_blocks <= 0 || num_blocks_total <= 0)
    return KLM_ERR_SHAPE;
  const int G = n_q / n_kv, gs = group_shift(G);
  if (gs < 0) return KLM_ERR_SHAPE;
  if (!q || !k_cache || !v_cache || !block_tables || !seq_info || !tiles || !out) return KLM_ERR_ARG;
  if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out)) return KLM_ERR_ALIGN;
  const float scale_log2 = scale * 1.4426950408889634f;
#define KLM_PREFILL_K(QT, F8)                                                                                    \
  hipLaunchKernelGGL((paged_attn_prefill_kernel<QT, F8>), dim3(n_kv, n_tiles), dim3(256), 0,                     \
                     static_cast<hipStream_t>(stream), (const uint16_t*)q, (const uint16_t*)k_cache,             \
                     (const uint16_t*)v_cache, block_tables, seq_info, tiles, (uint16_t*)out, n_seqs, q_stride,  \
                     n_q, n_kv, G, gs, max_blocks, num_blocks_total, scale_log2)
#define KLM_PREFILL(QT)             \
  do {                              \
    if (kv_fp8) KLM_PREFILL_K(QT, true); \
    else KLM_PREFILL_K(QT, false);  \
  } while (0)
#define KLM_PREFILL_LDS(QT)                                                                                        \
  do {                                                                                                           \
    static bool attr_done = false;                                                                               \
    if (!attr_done) {                                                                                            \
      if (hipFuncSetAttribute((const void*)paged_attn_prefill_lds_kernel<QT>,                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, PF_STAGES * PF_STAGE_BYTES) != hipSuccess) \
        return KLM_ERR_HIP;                                                                                      \
      attr_done = true;                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL((paged_attn_prefill_lds_kernel<QT>), dim3(n_kv, n_tiles), dim3(256),                      \
                       PF_STAGES * PF_STAGE_BYTES, static_cast<hipStream_t>(stream), (const uint16_t*)q,         \
                       (const uint16_t*)k_cache, (const uint16_t*)v_cache, block_tables, seq_info, tiles,        \
                       (uint16_t*)out, n_seqs, q_stride, n_q, n_kv, G, gs, max_blocks, num_blocks_total,         \
                       scale_log2);                                                                              \
  } while (0)
  if (variant == 1) {
    if (q_tiles == 1) KLM_PREFILL_LDS(1);
    else if (q_tiles == 2) KLM_PREFILL_LDS(2);
    else KLM_PREFILL_LDS(4);
  } else if (q_tiles == 1) KLM_PREFILL(1);
  else if (q_tiles == 2) KLM_PREFILL(2);
  else KLM_PREFILL(4);
#undef KLM_PREFILL_LDS
#undef KLM_PREFILL
#undef KLM_PREFILL_K
  return launch_status();
}

static int skinny_launch(const void* x, int x_stride, const void* W, const float* wscale, bool w8, void* out,
                         const void* residual, int M, int N, int K, int rms, int epilogue, float eps, int waves,
                         int row_tiles, int tiled, void* stream) {
  if (M <= 0 || M > 16 || N <= 0 || N % 16 != 0 || K <= 0 || K % 128 != 0) return KLM_ERR_SHAPE;
  if (epilogue < 0 || epilogue > 2) return KLM_ERR_VARIANT;
  if (!x || !W || !out || ((epilogue == EPI_RES) != (residual != nullptr)) || (w8 && !wscale)) return KLM_ERR_ARG;
  if (x_stride % 8 != 0 || x_stride < K) return KLM_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(W) || !aligned16(out) || !aligned16(residual) || !aligned16(wscale))
    return KLM_ERR_ALIGN;
  // rows of W per block: 64 when the x traffic would otherwise rival the weight traffic (M > 4)
  // and N is large enough to still give every CU several blocks
  int rt = row_tiles;
  if (rt != 1 && rt != 4) rt = (M > 4 && N % 64 == 0 && N / 64 >= 384) ? 4 : 1;
  if (rt == 4 && N % 64 != 0) rt = 1;
  // measured on MI355X (scripts/gpu_skinny_bench.py): 8 waves split K up to 8192, 16 beyond; 4 with 64-row blocks
  int nw = (waves == 1 || waves == 2 || waves == 4 || waves == 8 || waves == 16) ? waves
                                                                                 : (rt == 4 ? 4 : (K > 8192 ? 16 : 8));
  if (rt == 4 && nw > 8) nw = 8;
  while (nw > 1 && K % (128 * nw) != 0) nw >>= 1;
  if (nw < rt) rt = 1;
  const int kslice = K / nw;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(N / (16 * rt)), block(64 * nw);
  const uint16_t* xx = (const uint16_t*)x;
  uint16_t* oo = (uint16_t*)out;
  const uint16_t* rr = (const uint16_t*)residual;
#define KLM_SKINNY_L(R, E, T, Q)                                                                            \
  hipLaunchKernelGGL((skinny_linear_kernel<R, E, T, Q>), grid, block, 0, s, xx, x_stride, W, wscale, oo, rr, M, N, \
                     K, kslice, eps, tiled)
#define KLM_SKINNY(R, E)                         \
  do {                                           \
    if (w8) {                                    \
      if (rt == 4) KLM_SKINNY_L(R, E, 4, true);  \
      else KLM_SKINNY_L(R, E, 1, true);          \
    } else {                                     \
      if (rt == 4) KLM_SKINNY_L(R, E, 4, false); \
      else KLM_SKINNY_L(R, E, 1, false);         \
    }                                            \
  } while (0)
  if (rms) {
    if (epilogue == EPI_NONE) KLM_SKINNY(true, EPI_NONE);
    else if (epilogue == EPI_RES) KLM_SKINNY(true, EPI_RES);
    else KLM_SKINNY(true, EPI_SWIGLU);
  } else {
    if (epilogue == EPI_NONE) KLM_SKINNY(false, EPI_NONE);
    else if (epilogue == EPI_RES) KLM_SKINNY(false, EPI_RES);
    else KLM_SKINNY(false, EPI_SWIGLU);
  }
#undef KLM_SKINNY
#undef KLM_SKINNY_L
  return launch_status();
}

int klm_skinny_linear(const void* x, int x_stride, const void* W, void* out, const void* residual, int M, int N,
                      int K, int rms, int epilogue, float eps, int waves, int row_tiles, int tiled, void* stream) {
  return skinny_launch(x, x_stride, W, nullptr, false, out, residual, M, N, K, rms, epilogue, eps, waves, row_tiles,
                       tiled, stream);
}

int klm_skinny_linear_w8(const void* x, int x_stride, const void* W8, const float* wscale, void* out,
                         const void* residual, int M, int N, int K, int rms, int epilogue, float eps, int waves,
                         int row_tiles, int tiled, void* stream) {
  return skinny_launch(x, x_stride, W8, wscale, true, out, residual, M, N, K, rms, epilogue, eps, waves, row_tiles,
                       tiled, stream);
}

}  // extern "C"
