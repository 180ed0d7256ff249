// LLM serving kernels for MI355X (gfx950): the weight-streaming "skinny" linear layer of decode at
// batch <= 16 (bf16 or e4m3 weights, fused RMSNorm / residual / SwiGLU).
// No reference counterpart (vLLM is a README bullet in the reference, README.md:73).
#include "llm_common.h"

using namespace klm_llm;

namespace {

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
    uint16_t* out, const uint16_t* res, int M, int N, int K, int kslice, float eps, int tiled, int w_stride) {
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
  // pattern HBM likes best. Both layouts have the same distance between row tiles (16*w_stride elements).
  const size_t wlane = tiled ? ((size_t)(n0 >> 4) * K + (size_t)w * kslice) * 16 + lane * (W8 ? 16 : 8)
                             : (size_t)(n0 + fr) * w_stride + k0;
  const int wstep = tiled ? (W8 ? 1024 : 512) : (W8 ? 64 : 32);    // elements between consecutive MFMA steps
  const uint16_t* wp = reinterpret_cast<const uint16_t*>(Wv) + wlane;      // bf16
  const uint8_t* wp8 = reinterpret_cast<const uint8_t*>(Wv) + wlane;       // fp8
  const size_t tile_stride = (size_t)16 * w_stride;          // w_stride == K unless row-major rows are padded
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

}  // namespace

extern "C" {

static int skinny_launch(const void* x, int x_stride, const void* W, const float* wscale, bool w8, void* out,
                         const void* residual, int M, int N, int K, int rms, int epilogue, float eps, int waves,
                         int row_tiles, int tiled, int w_stride, void* stream) {
  if (w_stride == 0) w_stride = K;
  if (w_stride < K || w_stride % (w8 ? 16 : 8) != 0 || (tiled && w_stride != K)) return KLM_ERR_SHAPE;
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
                     K, kslice, eps, tiled, w_stride)
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
                      int K, int rms, int epilogue, float eps, int waves, int row_tiles, int tiled, int w_stride,
                      void* stream) {
  return skinny_launch(x, x_stride, W, nullptr, false, out, residual, M, N, K, rms, epilogue, eps, waves, row_tiles,
                       tiled, w_stride, stream);
}

int klm_skinny_linear_w8(const void* x, int x_stride, const void* W8, const float* wscale, void* out,
                         const void* residual, int M, int N, int K, int rms, int epilogue, float eps, int waves,
                         int row_tiles, int tiled, int w_stride, void* stream) {
  return skinny_launch(x, x_stride, W8, wscale, true, out, residual, M, N, K, rms, epilogue, eps, waves, row_tiles,
                       tiled, w_stride, stream);
}

}  // extern "C"
