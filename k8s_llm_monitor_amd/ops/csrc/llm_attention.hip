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
#include "klm_launch.h"
#include "llm_common.h"

using namespace klm_llm;

namespace {

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

// RoPE inside the decode kernel (bf16 cache): the query rows are rotated in registers, and the wave that owns the step
// holding the newest token rotates that token's key, writes key and value into the cache and patches them into the
// fragments it has just loaded, so no separate rope / cache-write launch runs before the attention.
struct RopeArgs {
  const float* cos_sin;   // [max_pos][128]: cos of the 64 pairs, then sin
  const int* slots;       // [B] cache slot of the newest token (< 0: write nothing)
  int max_pos;
  int num_slots;
};

// x[s] holds dims s*32 + fq*8 .. +8: the partner of a dim d < 64 is d + 64 = fragment s + 2 of the same lane.
// Same arithmetic and rounding as rope_cache_kernel (llm_elementwise.hip).
__device__ __forceinline__ void rope_frags(bf16x8 (&x)[4], const float* cs, int fq) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* c = cs + s * 32 + fq * 8;
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(c + 64), s1 = *reinterpret_cast<const f32x4*>(c + 68);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float cc = i < 4 ? c0[i & 3] : c1[i & 3], sn = i < 4 ? s0[i & 3] : s1[i & 3];
      const float a = (float)x[s][i], b = (float)x[s + 2][i];
      x[s][i] = (__bf16)(a * cc - b * sn);
      x[s + 2][i] = (__bf16)(b * cc + a * sn);
    }
  }
}

// Decode: one token per sequence. Block = (kv head, sequence, split); its 4 waves take the
// 32-token steps of the split round-robin and are combined through LDS.
template <bool KV8, bool ROPE = false>
__global__ __launch_bounds__(256) void paged_attn_decode_kernel(
    const uint16_t* __restrict__ q, const uint16_t* kc, const uint16_t* vc,
    const int* __restrict__ block_tables, const int* __restrict__ ctx_lens, uint16_t* __restrict__ out,
    float* __restrict__ tmp_o, float* __restrict__ tmp_ml, int q_stride, int n_q, int n_kv, int G,
    int max_blocks, int nb_total, int num_splits, float scale_log2, RopeArgs ra) {
  static_assert(!(KV8 && ROPE), "the fused rope path writes a bf16 cache");
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
  if (ROPE) {
    int pos = ctx - 1;
    pos = pos < 0 ? 0 : (pos >= ra.max_pos ? ra.max_pos - 1 : pos);
    const float* cs = ra.cos_sin + (size_t)pos * HD;
    if (valid) rope_frags(qf, cs, fq);
    const int last = total_steps - 1;
    // wave w takes steps s_begin + w, + 4, ...: exactly one wave of one split owns the step of the newest token
    const bool owner = ctx > 0 && last >= s_begin && last < s_end && ((last - s_begin) & 3) == w;
    if (!owner) {
      attend_range<KV8>(kv, qf, s_begin + w, s_end, 4, fr, fq, qpos, ctx - 1, scale_log2, a);
    } else {
      attend_range<KV8>(kv, qf, s_begin + w, last, 4, fr, fq, qpos, ctx - 1, scale_log2, a);
      Frags f;
      load_step<KV8>(kv, last, fr, fq, f);       // the newest token's slot holds stale bytes: patched below
      const int idx = (ctx - 1) & 31, tok = idx & 15;
      const int slot = ra.slots[b];
      const bool store = slot >= 0 && slot < ra.num_slots;
      const size_t cbase = ((size_t)(store ? slot / BLK : 0) * n_kv + h) * BLK_ELEMS;
      const uint16_t* krow = q + (size_t)b * q_stride + ((size_t)n_q + h) * HD;
      const uint16_t* vrow = krow + (size_t)n_kv * HD;
      // K: load_step gives lane fr the token rows t*4 + ((fr>>2)&1)*8 + (fr&3) of block fr>>3
      if (fr == (idx >> 4) * 8 + (tok >> 3) * 4 + (tok & 3)) {
        bf16x8 kn[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) kn[s4] = *reinterpret_cast<const bf16x8*>(krow + s4 * 32 + fq * 8);
        rope_frags(kn, cs, fq);
        uint16_t* kw = const_cast<uint16_t*>(kc) + cbase + (size_t)(slot % BLK) * HD + fq * 8;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          if ((tok >> 2) & 1) f.k[1][s4] = kn[s4]; else f.k[0][s4] = kn[s4];
          if (store) *reinterpret_cast<bf16x8*>(kw + s4 * 32) = kn[s4];
        }
      }
      // V^T: lane (fr, fq) holds dims n*16 + fr for the 8 tokens (fq&1)*8 .. +8 of block fq>>1
      if (fq == (idx >> 4) * 2 + (tok >> 3)) {
        uint16_t* vw = const_cast<uint16_t*>(vc) + cbase + (slot % BLK);
        const int e = tok & 7;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const uint16_t raw = vrow[n * 16 + fr];
          const __bf16 val = __builtin_bit_cast(__bf16, raw);
#pragma unroll
          for (int j = 0; j < 8; ++j) f.v[n][j] = (j == e) ? val : f.v[n][j];
          if (store) vw[(size_t)(n * 16 + fr) * BLK] = raw;
        }
      }
      attend_step(f, qf, last, fq, qpos, ctx - 1, scale_log2, a);
    }
  } else {
    attend_range<KV8>(kv, qf, s_begin + w, s_end, 4, fr, fq, qpos, ctx - 1, scale_log2, a);
  }

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

}  // namespace

extern "C" {

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
  const RopeArgs none{nullptr, nullptr, 0, 0};
  if (kv_fp8)
    hipLaunchKernelGGL((paged_attn_decode_kernel<true, false>), dim3(n_kv, B, num_splits), dim3(256), 0, s, (const uint16_t*)q,
                       (const uint16_t*)k_cache, (const uint16_t*)v_cache, block_tables, ctx_lens, (uint16_t*)out,
                       tmp_o, tmp_ml, q_stride, n_q, n_kv, G, max_blocks, num_blocks_total, num_splits, scale_log2, none);
  else
    hipLaunchKernelGGL((paged_attn_decode_kernel<false, false>), dim3(n_kv, B, num_splits), dim3(256), 0, s, (const uint16_t*)q,
                       (const uint16_t*)k_cache, (const uint16_t*)v_cache, block_tables, ctx_lens, (uint16_t*)out,
                       tmp_o, tmp_ml, q_stride, n_q, n_kv, G, max_blocks, num_blocks_total, num_splits, scale_log2, none);
  if (hipGetLastError() != hipSuccess) return KLM_ERR_HIP;
  if (num_splits > 1)
    hipLaunchKernelGGL(paged_attn_reduce_kernel, dim3((unsigned)B * n_q), dim3(128), 0, s, tmp_o, tmp_ml, ctx_lens,
                       (uint16_t*)out, n_q, max_blocks, num_splits);
  return launch_status();
}

// Decode attention straight from the un-rotated qkv activation [B][(n_q + 2 n_kv) * 128] (bf16 cache): RoPE on q and on
// the newest key, and the cache write of the newest key / value, happen inside the attention kernel. The position of
// sequence b's token is ctx_lens[b] - 1; `slots[b]` is its cache slot (< 0: padded row, nothing is written).
int klm_paged_attn_decode_rope(const void* qkv, void* k_cache, void* v_cache, const int32_t* block_tables,
                               const int32_t* ctx_lens, const int32_t* slots, const float* cos_sin, void* out, float* tmp_o,
                               float* tmp_ml, int B, int q_stride, int n_q, int n_kv, int max_blocks, int num_blocks_total,
                               int num_splits, float scale, int max_pos, void* stream) {
  if (q_stride < (n_q + 2 * n_kv) * HD || q_stride % 8 != 0) return KLM_ERR_SHAPE;
  if (B <= 0 || B > 65535 || n_q <= 0 || n_kv <= 0 || n_q % n_kv != 0 || max_blocks <= 0 || num_blocks_total <= 0 || max_pos <= 0 ||
      (int64_t)num_blocks_total * BLK > INT32_MAX)
    return KLM_ERR_SHAPE;
  const int G = n_q / n_kv;
  if (group_shift(G) < 0) return KLM_ERR_SHAPE;
  if (num_splits < 1 || num_splits > 256) return KLM_ERR_ARG;
  if (!qkv || !k_cache || !v_cache || !block_tables || !ctx_lens || !slots || !cos_sin || !out) return KLM_ERR_ARG;
  if (num_splits > 1 && (!tmp_o || !tmp_ml)) return KLM_ERR_ARG;
  if (!aligned16(qkv) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out) || !aligned16(tmp_o) || !aligned16(cos_sin))
    return KLM_ERR_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float scale_log2 = scale * 1.4426950408889634f;
  const RopeArgs ra{cos_sin, slots, max_pos, num_blocks_total * BLK};
  hipLaunchKernelGGL((paged_attn_decode_kernel<false, true>), dim3(n_kv, B, num_splits), dim3(256), 0, s, (const uint16_t*)qkv,
                     (const uint16_t*)k_cache, (const uint16_t*)v_cache, block_tables, ctx_lens, (uint16_t*)out, tmp_o, tmp_ml,
                     q_stride, n_q, n_kv, G, max_blocks, num_blocks_total, num_splits, scale_log2, ra);
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
    if (klm_ensure_dyn_lds<paged_attn_prefill_lds_kernel<QT>>(PF_STAGES * PF_STAGE_BYTES) != KLM_OK) return KLM_ERR_HIP;     \
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

}  // extern "C"
