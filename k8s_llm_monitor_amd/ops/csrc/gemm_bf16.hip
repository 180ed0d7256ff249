// bf16 MFMA GEMM burner kernels for MI355X (gfx950, CDNA4).
//
//   C[M][N] (bf16) = A[M][K] (bf16) * B[N][K]^T (bf16), fp32 accumulate.
//
// No reference counterpart (the reference has no native code):
// this is the "known, steady GFX load" of BASELINE.json north_star item 4.
//
// Two kernels:
//  * gemm_bf16_128: 128x128x64 tile, 4 waves (2x2), one 32 KiB LDS buffer, LDS-DMA
//    (global_load_lds_dwordx4) staging, two barriers per K step. Several blocks per CU
//    hide the staging latency. Small-shape / fallback kernel.
//  * gemm_bf16_256: 256x256x64 tile, 8 waves (2x4), 128 KiB LDS as a ring of 2 stages x 4
//    half-tiles. Each K tile runs in 4 phases of 16 MFMAs per wave; each phase issues
//    the LDS-DMA of one half-tile of the NEXT K tile, and waits only with counted
//    `s_waitcnt vmcnt(N)` so 3 half-tiles stay in flight across the raw `s_barrier`s.
//    The two wave groups (waves 0-3 / 4-7 = the two waves of each SIMD) run staggered by
//    one barrier, so one reads LDS while its SIMD partner issues MFMAs.
//
// LDS images are lane-linear (what LDS-DMA writes); the bank-conflict swizzle
// (16-byte chunk index XOR row&7 inside each 128-byte row) is applied to the per-lane
// SOURCE address and again on the ds_read side.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klm_ops.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define KLM_AS1 __attribute__((address_space(1)))
#define KLM_AS3 __attribute__((address_space(3)))

// One wave-instruction: 64 lanes x 16 B from per-lane global addresses into 1 KiB of LDS
// starting at the wave-uniform address `lds` (+ lane*16, added by the hardware).
__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const KLM_AS1 void*)g, (KLM_AS3 void*)lds, 16, 0, 0);
}

#define KLM_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// Block id -> output tile. (1) XCD-aware: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each of the 8 residue classes one contiguous chunk of tile ids
// (bijective for any grid size). (2) Inside the chunk walk tiles in groups of GM tile
// rows, column by column, so that the blocks resident on an XCD at one time form a
// compact patch that shares A and B panels through that XCD's L2.
template <int GM>
__device__ __forceinline__ void tile_of_block(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int per_group = GM * tiles_n;
  const int g = id / per_group;
  const int first_m = g * GM;
  const int gsize = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int in_g = id - g * per_group;
  tm = first_m + in_g % gsize;
  tn = in_g / gsize;
}

__device__ __forceinline__ void store_c4(uint16_t* C, size_t idx, const f32x4& v) {
  bf16x4 o;
  o[0] = (__bf16)v[0];
  o[1] = (__bf16)v[1];
  o[2] = (__bf16)v[2];
  o[3] = (__bf16)v[3];
  *reinterpret_cast<bf16x4*>(C + idx) = o;
}

// ------------------------------------------------------------------------------------
// 128x128x64, 4 waves.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_bf16_128(const uint16_t* __restrict__ A,
                                                     const uint16_t* __restrict__ B,
                                                     uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // A [128][64] | B [128][64]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  int tm, tn;
  tile_of_block<8>(blockIdx.x, M >> 7, N >> 7, tm, tn);
  const int brow = tm << 7, bcol = tn << 7;

  // staging: wave `wid` fills rows [wid*32, wid*32+32) of both tiles, 8 rows per piece
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const uint16_t* ga = A + (size_t)(brow + wid * 32 + srow) * K + schunk * 8;
  const uint16_t* gb = B + (size_t)(bcol + wid * 32 + srow) * K + schunk * 8;
  char* la = smem + wid * 4096;
  char* lb = smem + 16384 + wid * 4096;

  // fragment reads: row = w*64 + t*16 + (lane&15); 16-byte chunk = ks*4 + (lane>>4)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = ((fq ^ (lane & 7)) << 4);
  const int a_off0 = (wr * 64 + fr) * 128 + sw0, a_off1 = a_off0 ^ 64;
  const int b_off0 = 16384 + (wc * 64 + fr) * 128 + sw0, b_off1 = b_off0 ^ 64;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < K; k0 += 64) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(ga + (size_t)i * 8 * K + k0, la + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(gb + (size_t)i * 8 * K + k0, lb + i * 1024);
    KLM_VMCNT(0);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ao = ks ? a_off1 : a_off0, bo = ks ? b_off1 : b_off0;
      bf16x8 a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const bf16x8*>(smem + ao + t * 2048);
        b[t] = *reinterpret_cast<const bf16x8*>(smem + bo + t * 2048);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          // X = B rows (n), Y = A rows (m): D[n][m]; lane holds 4 consecutive n of one m
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int m = brow + wr * 64 + mi * 16 + fr;
      const int n = bcol + wc * 64 + ni * 16 + fq * 4;
      store_c4(C, (size_t)m * N + n, acc[mi][ni]);
    }
}

// ------------------------------------------------------------------------------------
// 256x256x64, 8 waves, LDS-DMA ring with counted vmcnt.
//
// LDS: stage s in {0,1} at s*64 KiB; inside a stage four 16 KiB half-tile slots
//   A0 (+0), B0 (+16 KiB), B1 (+32 KiB), A1 (+48 KiB).
// Half-tiles are cut so that EVERY wave needs them in the same order:
//   A-half h = rows {wr*128 + h*64 + r : wr in 0..1, r in 0..63}   (slot row = wr*64 + r)
//   B-half h = cols {wc*64  + h*32 + r : wc in 0..3, r in 0..31}   (slot row = wc*32 + r)
// A wave (wr, wc) computes its 128x64 output as four quadrants, one per phase:
//   phase 0: (a0,b0)  reads A0,B0     issues A0,B0 of tile t+1   then vmcnt -> B1(t)
//   phase 1: (a0,b1)  reads B1        issues B1 of tile t+1      then vmcnt -> A1(t)
//   phase 2: (a1,b1)  reads A1        issues A1 of tile t+1
//   phase 3: (a1,b0)  (b0 kept)                                  then vmcnt -> A0,B0(t+1)
// A slot is rewritten 4 phases (8 barriers) after its last read, and read only after a
// barrier that follows every wave's counted wait for it (one phase earlier, because the
// wave groups are staggered by one barrier).
// ------------------------------------------------------------------------------------
#define KLM_SLOT_A0 0
#define KLM_SLOT_B0 16384
#define KLM_SLOT_B1 32768
#define KLM_SLOT_A1 49152

//
// SPLIT (decode at batch 129 - 256, ops/llm.py `_mid_partials`): the grid's y index picks a K range of `Kc`, the result
// goes out as fp32 partial sums P[split][M][N] (what the mid-batch epilogues and the RoPE kernel consume), A may have
// any number of rows <= 256 * tiles (rows past M re-read row M-1 and are not stored) and both operands a row stride.
// The burner instantiation (SPLIT = false) compiles to the code it had before the template existed.
template <bool SPLIT, bool SWIGLU = false>
__global__ __launch_bounds__(512) void gemm_bf16_256_t(const uint16_t* __restrict__ A,
                                                       const uint16_t* __restrict__ B,
                                                       void* __restrict__ Cv, int M, int N, int K, int lda, int ldb, int Kc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* const C = reinterpret_cast<uint16_t*>(Cv);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  int tm, tn;
  tile_of_block<8>(blockIdx.x, SPLIT ? (M + 255) >> 8 : M >> 8, N >> 8, tm, tn);
  const int brow = tm << 8, bcol = tn << 8;

  // ---- staging addresses. Wave `wid` fills slot rows [wid*16, wid*16+16): 2 pieces.
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const uint16_t* pa = A + (size_t)(brow + (wid >> 2) * 128 + (wid & 3) * 16 + srow) * K + schunk * 8;
  const uint16_t* pb = B + (size_t)(bcol + (wid >> 1) * 64 + (wid & 1) * 16 + srow) * (SPLIT ? ldb : K) + schunk * 8;
  const size_t rowsK8 = (size_t)8 * (SPLIT ? ldb : K);
  const size_t ldB = SPLIT ? (size_t)ldb : (size_t)K;
  char* const lw = smem + wid * 2048;  // + stage*65536 + slot + i*1024
  const uint16_t* pas[2][2] = {{pa, pa}, {pa, pa}};   // SPLIT: the four A rows of this lane, clamped to the last real row
  if (SPLIT) {
    const int kbeg = blockIdx.y * Kc;
    pb += kbeg;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int r = brow + (wid >> 2) * 128 + (wid & 3) * 16 + srow + h * 64 + j * 8;
        r = r < M ? r : M - 1;
        pas[h][j] = A + (size_t)r * lda + schunk * 8 + kbeg;
      }
  }

#define KLM_ISSUE_A(h, stage, k0)                                                              \
  do {                                                                                         \
    if (SPLIT) {                                                                               \
      glds16(pas[h][0] + (k0), lw + (stage) * 65536 + ((h) ? KLM_SLOT_A1 : KLM_SLOT_A0));      \
      glds16(pas[h][1] + (k0), lw + (stage) * 65536 + ((h) ? KLM_SLOT_A1 : KLM_SLOT_A0) + 1024); \
      break;                                                                                   \
    }                                                                                          \
    glds16(pa + (size_t)(h) * 64 * K + (k0), lw + (stage) * 65536 + ((h) ? KLM_SLOT_A1 : KLM_SLOT_A0)); \
    glds16(pa + (size_t)(h) * 64 * K + rowsK8 + (k0),                                          \
           lw + (stage) * 65536 + ((h) ? KLM_SLOT_A1 : KLM_SLOT_A0) + 1024);                   \
  } while (0)
#define KLM_ISSUE_B(h, stage, k0)                                                              \
  do {                                                                                         \
    glds16(pb + (size_t)(h) * 32 * ldB + (k0), lw + (stage) * 65536 + ((h) ? KLM_SLOT_B1 : KLM_SLOT_B0)); \
    glds16(pb + (size_t)(h) * 32 * ldB + rowsK8 + (k0),                                        \
           lw + (stage) * 65536 + ((h) ? KLM_SLOT_B1 : KLM_SLOT_B0) + 1024);                   \
  } while (0)

  // ---- fragment read offsets inside a slot
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = ((fq ^ (lane & 7)) << 4);
  const int a_off0 = (wr * 64 + fr) * 128 + sw0, a_off1 = a_off0 ^ 64;
  const int b_off0 = (wc * 32 + fr) * 128 + sw0, b_off1 = b_off0 ^ 64;

  f32x4 acc[2][2][4][2];  // [a half][b half][m tile][n tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 a[2][4], b0[2][2], b1[2][2];  // [ks][tile]

#define KLM_READ_A(slot_base)                                                            \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                     \
    a[0][mi] = *reinterpret_cast<const bf16x8*>((slot_base) + a_off0 + mi * 2048);       \
    a[1][mi] = *reinterpret_cast<const bf16x8*>((slot_base) + a_off1 + mi * 2048);       \
  }
#define KLM_READ_B(bv, slot_base)                                                        \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                     \
    bv[0][ni] = *reinterpret_cast<const bf16x8*>((slot_base) + b_off0 + ni * 2048);      \
    bv[1][ni] = *reinterpret_cast<const bf16x8*>((slot_base) + b_off1 + ni * 2048);      \
  }
#define KLM_MFMA16(ah, bh, bv)                                                           \
  do {                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                     \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                     \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                     \
      acc[ah][bh][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                     \
          bv[ks][ni], a[ks][mi], acc[ah][bh][mi][ni], 0, 0, 0);                          \
    __builtin_amdgcn_s_setprio(0);                                                       \
  } while (0)
#define KLM_BAR() __builtin_amdgcn_s_barrier()

#ifdef KLM_EXP_NO_GLOBAL   // timing-only: the loop issues no global->LDS traffic (outputs are wrong)
#undef KLM_ISSUE_A
#undef KLM_ISSUE_B
#define KLM_ISSUE_A(h, stage, k0) do { } while (0)
#define KLM_ISSUE_B(h, stage, k0) do { } while (0)
#endif
  const int nt = (SPLIT ? Kc : K) >> 6;

  // ---- prologue: stage K tile 0 into stage 0 in need order A0,B0 | B1 | A1
  KLM_ISSUE_A(0, 0, 0);
  KLM_ISSUE_B(0, 0, 0);
  KLM_ISSUE_B(1, 0, 0);
  KLM_ISSUE_A(1, 0, 0);
  KLM_VMCNT(4);
  KLM_BAR();
  if (wr == 1) KLM_BAR();  // stagger: group 1 runs one barrier behind group 0

  for (int t = 0; t < nt - 1; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    const char* sb = smem + cur * 65536;
    const int k1 = (t + 1) << 6;
    // phase 0
    KLM_READ_A(sb + KLM_SLOT_A0);
    KLM_READ_B(b0, sb + KLM_SLOT_B0);
    KLM_ISSUE_A(0, nxt, k1);
    KLM_ISSUE_B(0, nxt, k1);
    KLM_VMCNT(6);
    KLM_BAR();
    KLM_MFMA16(0, 0, b0);
    KLM_BAR();
    // phase 1
#ifndef KLM_EXP_SKIP_B1
    KLM_READ_B(b1, sb + KLM_SLOT_B1);
#endif
    KLM_ISSUE_B(1, nxt, k1);
    KLM_VMCNT(6);
    KLM_BAR();
    KLM_MFMA16(0, 1, b1);
    KLM_BAR();
    // phase 2
#ifndef KLM_EXP_SKIP_A1
    KLM_READ_A(sb + KLM_SLOT_A1);
#endif
    KLM_ISSUE_A(1, nxt, k1);
    KLM_BAR();
    KLM_MFMA16(1, 1, b1);
    KLM_BAR();
    // phase 3
    KLM_VMCNT(4);
    KLM_BAR();
    KLM_MFMA16(1, 0, b0);
    KLM_BAR();
  }
  {  // last K tile: nothing left to issue
    const char* sb = smem + ((nt - 1) & 1) * 65536;
    KLM_READ_A(sb + KLM_SLOT_A0);
    KLM_READ_B(b0, sb + KLM_SLOT_B0);
    KLM_VMCNT(2);
    KLM_BAR();
    KLM_MFMA16(0, 0, b0);
    KLM_BAR();
    KLM_READ_B(b1, sb + KLM_SLOT_B1);
    KLM_VMCNT(0);
    KLM_BAR();
    KLM_MFMA16(0, 1, b1);
    KLM_BAR();
    KLM_READ_A(sb + KLM_SLOT_A1);
    KLM_BAR();
    KLM_MFMA16(1, 1, b1);
    KLM_BAR();
    KLM_BAR();
    KLM_MFMA16(1, 0, b0);
    KLM_BAR();
  }
  if (wr == 0) KLM_BAR();  // matches group 1's extra barrier in the prologue

#pragma unroll
  for (int ah = 0; ah < 2; ++ah)
#pragma unroll
    for (int bh = 0; bh < 2; ++bh)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int m = brow + wr * 128 + ah * 64 + mi * 16 + fr;
          const int n = bcol + wc * 64 + bh * 32 + ni * 16 + fq * 4;
          if (SWIGLU) {
            // gate / up rows of B are interleaved by 8: in a 16-column tile lanes fq 0,1 hold gate, fq 2,3 the matching up
            // columns (lane ^ 32). One K range only: the activation goes out in bf16, no partial sums, no epilogue launch.
            f32x4 v = acc[ah][bh][mi][ni], o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = __shfl_xor(v[i], 32);
            if (fq < 2 && m < M) {
              f32x4 y;
#pragma unroll
              for (int i = 0; i < 4; ++i) y[i] = v[i] / (1.f + __expf(-v[i])) * o[i];
              store_c4(C, (size_t)m * (N >> 1) + ((n - fq * 4) >> 1) + fq * 4, y);
            }
          } else if (SPLIT) {
            float* const P = reinterpret_cast<float*>(Cv) + (size_t)blockIdx.y * M * N;
            if (m < M) *reinterpret_cast<f32x4*>(P + (size_t)m * N + n) = acc[ah][bh][mi][ni];
          } else {
            store_c4(C, (size_t)m * N + n, acc[ah][bh][mi][ni]);
          }
        }
}

extern "C" int klm_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                             void* stream) {
  if (!A || !B || !C) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
  if (M <= 0 || N <= 0 || K <= 0) return KLM_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (variant == 0) {
    if ((M & 127) || (N & 127) || (K & 63)) return KLM_ERR_SHAPE;
    const int grid = (M >> 7) * (N >> 7);
    hipLaunchKernelGGL(gemm_bf16_128, dim3(grid), dim3(256), 32768, s, (const uint16_t*)A,
                       (const uint16_t*)B, (uint16_t*)C, M, N, K);
  } else if (variant == 1) {
    if ((M & 255) || (N & 255) || (K & 63) || K < 128) return KLM_ERR_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)gemm_bf16_256_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              131072) != hipSuccess)
        return KLM_ERR_HIP;
      attr_set = true;
    }
    const int grid = (M >> 8) * (N >> 8);
    hipLaunchKernelGGL(gemm_bf16_256_t<false>, dim3(grid), dim3(512), 131072, s, (const uint16_t*)A,
                       (const uint16_t*)B, (void*)C, M, N, K, K, K, K);
  } else {
    return KLM_ERR_VARIANT;
  }
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}

// K split chosen for `P[ks][M][N] = A[M][K] . B[N][K]^T` on the 256 x 256 kernel: as many splits as still fit one wave of
// blocks on the 256 CUs, at most 8 (every split writes M x N fp32 partials) and at least 256 k per block.
extern "C" int klm_gemm_bf16_splitk_plan(int M, int N, int K, int* ksplit) {
  if (!ksplit) return KLM_ERR_ARG;
  if (M <= 0 || N <= 0 || (N & 255) || K < 256 || (K & 63)) return KLM_ERR_SHAPE;
  const int tiles = ((M + 255) >> 8) * (N >> 8);
  int ks = *ksplit;
  if (ks <= 0) {
    ks = 1;
    while (ks < 8 && tiles * ks * 2 <= 256 && K % (ks * 2 * 64) == 0 && K / (ks * 2) >= 256) ks *= 2;
  }
  if (K % (ks * 64) != 0 || K / ks < 128) return KLM_ERR_SHAPE;
  *ksplit = ks;
  return KLM_OK;
}

// fp32 partial sums of a K-split GEMM for activations of up to a few hundred rows (decode at batch 129 - 256):
// part[ksplit][M][N]; A bf16 [M][>= K] with row stride lda, B bf16 [N][>= K] with row stride ldb, N % 256 == 0.
extern "C" int klm_gemm_bf16_splitk(const void* A, int lda, const void* B, int ldb, float* part, int M, int N, int K,
                                    int ksplit, void* stream) {
  if (!A || !B || !part) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)part) & 15) return KLM_ERR_ALIGN;
  if (M <= 0 || N <= 0 || (N & 255) || lda < K || ldb < K || (lda & 7) || (ldb & 7)) return KLM_ERR_SHAPE;
  int ks = ksplit;
  const int rc = klm_gemm_bf16_splitk_plan(M, N, K, &ks);
  if (rc != KLM_OK) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_256_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) !=
        hipSuccess)
      return KLM_ERR_HIP;
    attr_set = true;
  }
  const int tiles = ((M + 255) >> 8) * (N >> 8);
  hipLaunchKernelGGL(gemm_bf16_256_t<true>, dim3(tiles, ks), dim3(512), 131072, (hipStream_t)stream, (const uint16_t*)A,
                     (const uint16_t*)B, (void*)part, M, N, K, lda, ldb, K / ks);
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}

// act[M][N/2] = silu(gate) * up of A . B^T with B's gate / up rows interleaved by 8 (ops.llm.interleave_gate_up), for shapes
// whose tiles fill the chip without a K split (Llama-3-70B gate_up: 224 tiles): SwiGLU in the tile epilogue, bf16 out.
extern "C" int klm_gemm_bf16_swiglu(const void* A, int lda, const void* B, int ldb, void* act, int M, int N, int K, void* stream) {
  if (!A || !B || !act) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)act) & 15) return KLM_ERR_ALIGN;
  if (M <= 0 || N <= 0 || (N & 255) || K < 128 || (K & 63) || lda < K || ldb < K || (lda & 7) || (ldb & 7)) return KLM_ERR_SHAPE;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bf16_256_t<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) !=
        hipSuccess)
      return KLM_ERR_HIP;
    attr_set = true;
  }
  const int tiles = ((M + 255) >> 8) * (N >> 8);
  hipLaunchKernelGGL((gemm_bf16_256_t<true, true>), dim3(tiles, 1), dim3(512), 131072, (hipStream_t)stream, (const uint16_t*)A,
                     (const uint16_t*)B, act, M, N, K, lda, ldb, K);
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}
