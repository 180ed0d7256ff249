// MX-format MFMA GEMM burner kernels for MI355X (gfx950, CDNA4): fp8 (OCP e4m3) and fp4 (e2m1).
//
//   C[M][N] (bf16) = A[M][K] * B[N][K]^T, fp32 accumulate, on the block-scaled MFMA
//   v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 scales (2^0 for every 32-element block).
//   fp8: 2x the bf16 MFMA rate;  fp4 (two elements per byte): 4x.
//
// No reference counterpart: the low-precision variants of the "known,
// steady GFX load" of BASELINE.json north_star item 4.
//
// Same structure as gemm_bf16_256 (gemm_bf16.hip): 256x256 output tile, 8 waves (2x4),
// 128 KiB LDS as a ring of 2 stages x 4 half-tile slots, LDS-DMA staging with counted
// vmcnt, two wave groups staggered by one barrier. A K tile is 128 BYTES per row in all three
// kernels, so staging, slot layout and swizzle are byte-identical; what differs is how many
// elements those 128 bytes hold and how many MFMAs consume them per 16x16 tile:
//   bf16:  64 elements, 2 x mfma_16x16x32        (16 bytes per lane each)
//   fp8:  128 elements, 1 x mfma_scale_16x16x128 (32 bytes per lane)
//   fp4:  256 elements, 2 x mfma_scale_16x16x128 (16 bytes per lane each)
//
// Operand k-mapping: lane l (row l&15, lane group g = l>>4) supplies 16-byte chunk(s) of the
// 128-byte row, for A and for B alike: fp8 chunks g and g+4, fp4 chunk 4*ks+g for MFMA ks.
// The hardware multiplies A's (lane group, byte j) with B's (lane group, byte j), so any
// assignment of k to (group, byte) gives the same sum as long as both operands use it (and,
// for fp4, whichever nibble of a byte is "first"). These assignments make the LDS read pattern
// identical to the bf16 kernel's (chunk index = 4*h + g), which rocprofv3 shows free of bank
// conflicts; "32 consecutive bytes per lane" (chunks 2g, 2g+1) measured 0.64 conflict
// cycles per active instruction (profiles/r1_prof/pmc_summary.json).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klm_ops.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define KLM_AS1 __attribute__((address_space(1)))
#define KLM_AS3 __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const KLM_AS1 void*)g, (KLM_AS3 void*)lds, 16, 0, 0);
}

#define KLM_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define KLM_BAR() __builtin_amdgcn_s_barrier()

// XCD-aware, bijective block -> tile map (see gemm_bf16.hip).
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

#define KLM_SLOT_A0 0
#define KLM_SLOT_B0 16384
#define KLM_SLOT_B1 32768
#define KLM_SLOT_A1 49152
#define KLM_UNIT_SCALE 0x7f7f7f7f  // four E8M0 scales of 2^0

// FMT: the MFMA's operand format code, 0 = fp8 e4m3, 4 = fp4 e2m1. K is the row length in BYTES.
template <int FMT>
__global__ __launch_bounds__(512) void gemm_mx_256(const uint8_t* __restrict__ A,
                                                   const uint8_t* __restrict__ B,
                                                   uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  int tm, tn;
  tile_of_block<8>(blockIdx.x, M >> 8, N >> 8, tm, tn);
  const int brow = tm << 8, bcol = tn << 8;

  // staging: a row of a K tile is 128 bytes = 8 chunks of 16; lane -> (row lane>>3, chunk lane&7)
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const uint8_t* pa = A + (size_t)(brow + (wid >> 2) * 128 + (wid & 3) * 16 + srow) * K + schunk * 16;
  const uint8_t* pb = B + (size_t)(bcol + (wid >> 1) * 64 + (wid & 1) * 16 + srow) * K + schunk * 16;
  const size_t rowsK8 = (size_t)8 * K;
  char* const lw = smem + wid * 2048;

#define KLM_ISSUE_A(h, stage, k0)                                                              \
  do {                                                                                         \
    glds16(pa + (size_t)(h) * 64 * K + (k0), lw + (stage) * 65536 + ((h) ? KLM_SLOT_A1 : KLM_SLOT_A0)); \
    glds16(pa + (size_t)(h) * 64 * K + rowsK8 + (k0),                                          \
           lw + (stage) * 65536 + ((h) ? KLM_SLOT_A1 : KLM_SLOT_A0) + 1024);                   \
  } while (0)
#define KLM_ISSUE_B(h, stage, k0)                                                              \
  do {                                                                                         \
    glds16(pb + (size_t)(h) * 32 * K + (k0), lw + (stage) * 65536 + ((h) ? KLM_SLOT_B1 : KLM_SLOT_B0)); \
    glds16(pb + (size_t)(h) * 32 * K + rowsK8 + (k0),                                          \
           lw + (stage) * 65536 + ((h) ? KLM_SLOT_B1 : KLM_SLOT_B0) + 1024);                   \
  } while (0)

  // fragment reads: row = base + t*16 + (lane&15); 16-byte chunks fq and fq+4 (swizzled)
  const int fr = lane & 15, fq = lane >> 4;
  const int key = lane & 7;
  const int c0 = (fq ^ key) << 4, c1 = c0 ^ 64;
  const int a_row = (wr * 64 + fr) * 128, b_row = (wc * 32 + fr) * 128;

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragments as [ks][tile]: fp8 uses ks = 0 only and all 8 registers (chunks g, g+4); fp4 uses
  // ks = 0, 1 with the low 4 registers each (chunk 4*ks+g); the compiler drops what a format
  // does not touch (the fp4 form of the instruction reads v[0:3] only).
  i32x8 a[2][4], b0[2][2], b1[2][2];

#define KLM_LOAD_FRAG(dst, base)                                                 \
  do {                                                                           \
    const i32x4 lo_ = *reinterpret_cast<const i32x4*>((base) + c0);              \
    const i32x4 hi_ = *reinterpret_cast<const i32x4*>((base) + c1);              \
    if (FMT == 0) {                                                              \
      dst[0] = i32x8{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]}; \
    } else {                                                                     \
      dst[0] = i32x8{lo_[0], lo_[1], lo_[2], lo_[3], 0, 0, 0, 0};                \
      dst[1] = i32x8{hi_[0], hi_[1], hi_[2], hi_[3], 0, 0, 0, 0};                \
    }                                                                            \
  } while (0)
#define KLM_READ_A(slot_base)                                              \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                       \
    i32x8 f_[2];                                                           \
    KLM_LOAD_FRAG(f_, (slot_base) + a_row + mi * 2048);                    \
    a[0][mi] = f_[0];                                                      \
    if (FMT != 0) a[1][mi] = f_[1];                                        \
  }
#define KLM_READ_B(bv, slot_base)                                          \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                       \
    i32x8 f_[2];                                                           \
    KLM_LOAD_FRAG(f_, (slot_base) + b_row + ni * 2048);                    \
    bv[0][ni] = f_[0];                                                     \
    if (FMT != 0) bv[1][ni] = f_[1];                                       \
  }
#define KLM_MFMA8(ah, bh, bv)                                                            \
  do {                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                       \
    _Pragma("unroll") for (int ks = 0; ks < (FMT == 0 ? 1 : 2); ++ks)                    \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                     \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                     \
      acc[ah][bh][mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(            \
          bv[ks][ni], a[ks][mi], acc[ah][bh][mi][ni], FMT, FMT, 0, KLM_UNIT_SCALE, 0, KLM_UNIT_SCALE); \
    __builtin_amdgcn_s_setprio(0);                                                       \
  } while (0)

  const int nt = K >> 7;

  KLM_ISSUE_A(0, 0, 0);
  KLM_ISSUE_B(0, 0, 0);
  KLM_ISSUE_B(1, 0, 0);
  KLM_ISSUE_A(1, 0, 0);
  KLM_VMCNT(4);
  KLM_BAR();
  if (wr == 1) KLM_BAR();

  for (int t = 0; t < nt - 1; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    const char* sb = smem + cur * 65536;
    const int k1 = (t + 1) << 7;
    // phase 0
    KLM_READ_A(sb + KLM_SLOT_A0);
    KLM_READ_B(b0, sb + KLM_SLOT_B0);
    KLM_ISSUE_A(0, nxt, k1);
    KLM_ISSUE_B(0, nxt, k1);
    KLM_VMCNT(6);
    KLM_BAR();
    KLM_MFMA8(0, 0, b0);
    KLM_BAR();
    // phase 1
    KLM_READ_B(b1, sb + KLM_SLOT_B1);
    KLM_ISSUE_B(1, nxt, k1);
    KLM_VMCNT(6);
    KLM_BAR();
    KLM_MFMA8(0, 1, b1);
    KLM_BAR();
    // phase 2
    KLM_READ_A(sb + KLM_SLOT_A1);
    KLM_ISSUE_A(1, nxt, k1);
    KLM_BAR();
    KLM_MFMA8(1, 1, b1);
    KLM_BAR();
    // phase 3
    KLM_VMCNT(4);
    KLM_BAR();
    KLM_MFMA8(1, 0, b0);
    KLM_BAR();
  }
  {
    const char* sb = smem + ((nt - 1) & 1) * 65536;
    KLM_READ_A(sb + KLM_SLOT_A0);
    KLM_READ_B(b0, sb + KLM_SLOT_B0);
    KLM_VMCNT(2);
    KLM_BAR();
    KLM_MFMA8(0, 0, b0);
    KLM_BAR();
    KLM_READ_B(b1, sb + KLM_SLOT_B1);
    KLM_VMCNT(0);
    KLM_BAR();
    KLM_MFMA8(0, 1, b1);
    KLM_BAR();
    KLM_READ_A(sb + KLM_SLOT_A1);
    KLM_BAR();
    KLM_MFMA8(1, 1, b1);
    KLM_BAR();
    KLM_BAR();
    KLM_MFMA8(1, 0, b0);
    KLM_BAR();
  }
  if (wr == 0) KLM_BAR();

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
          store_c4(C, (size_t)m * N + n, acc[ah][bh][mi][ni]);
        }
}

template <int FMT>
static int launch_mx(const void* A, const void* B, void* C, int M, int N, int Kbytes, void* stream) {
  static bool attr_set = false;          // one flag per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_mx_256<FMT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            131072) != hipSuccess)
      return KLM_ERR_HIP;
    attr_set = true;
  }
  const int grid = (M >> 8) * (N >> 8);
  hipLaunchKernelGGL((gemm_mx_256<FMT>), dim3(grid), dim3(512), 131072, (hipStream_t)stream,
                     (const uint8_t*)A, (const uint8_t*)B, (uint16_t*)C, M, N, Kbytes);
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}

static int check_mx(const void* A, const void* B, const void* C, int M, int N, int Kbytes, int variant) {
  if (!A || !B || !C) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
  if (variant != 0) return KLM_ERR_VARIANT;
  if (M <= 0 || N <= 0 || Kbytes <= 0 || (M & 255) || (N & 255) || (Kbytes & 127) || Kbytes < 256) return KLM_ERR_SHAPE;
  return KLM_OK;
}

extern "C" int klm_gemm_fp8(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                            void* stream) {
  if (int e = check_mx(A, B, C, M, N, K, variant)) return e;
  return launch_mx<0>(A, B, C, M, N, K, stream);
}

// K counts fp4 ELEMENTS; rows are K/2 bytes, two elements per byte.
extern "C" int klm_gemm_fp4(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                            void* stream) {
  if (K <= 0 || (K & 1)) return KLM_ERR_SHAPE;
  if (int e = check_mx(A, B, C, M, N, K >> 1, variant)) return e;
  return launch_mx<4>(A, B, C, M, N, K >> 1, stream);
}
