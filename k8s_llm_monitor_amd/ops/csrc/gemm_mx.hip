// MX-format MFMA GEMM burner kernels for MI355X (gfx950, CDNA4): fp8 (OCP e4m3) and fp4 (e2m1).
//
//   C[M][N] (bf16) = A[M][K] * B[N][K]^T, fp32 accumulate, on the block-scaled MFMA
//   v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 scales (2^0 for every 32-element block).
//   fp8: 2x the bf16 MFMA rate;  fp4 (two elements per byte): 4x.
//
// No reference counterpart: the low-precision variants of the "known,
// steady GFX load" of BASELINE.json north_star item 4.
//
// The pipeline is gemm_ring256.h (shared with gemm_bf16_256_t): 256x256 output tile, 8 waves (2x4),
// 128 KiB LDS as a ring of 2 stages x 4 half-tile slots, LDS-DMA staging with counted vmcnt, two wave
// groups staggered by one barrier. A K tile is 128 BYTES per row in all three kernels, so staging, slot
// layout and swizzle are byte-identical; this file supplies the operand policy: how many elements those
// 128 bytes hold and how many MFMAs consume them per 16x16 tile:
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
// cycles per active instruction (round 1; profiles/r2_prof/pmc_summary.json has the current layout).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_p128.h"
#include "gemm_quad_asm.h"
#include "gemm_ring256.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define KLM_UNIT_SCALE 0x7f7f7f7f  // four E8M0 scales of 2^0

// FMT: the MFMA's operand format code, 0 = fp8 e4m3, 4 = fp4 e2m1.
// Fragments as [ks][tile]: fp8 uses ks = 0 only and all 8 registers (chunks g, g+4); fp4 uses ks = 0, 1 with the
// low 4 registers each (chunk 4*ks+g); the compiler drops what a format does not touch (the fp4 form of the
// instruction reads v[0:3] only).
// The MFMA is written as inline asm with the accumulator TIED (destination = source C): as a builtin, hipcc (ROCm 7.2)
// leaves the two untied for this instruction, picks other destination registers inside the unrolled loop and copies
// the accumulators back at the loop's end (259 v_mov_b32 per two K tiles = 4 per MFMA, measured 0.68 MFMA-busy for
// fp8 where the bf16 kernel of the same pipeline reaches 0.80). Inline asm is invisible to the hazard recognizer:
// dependent MFMAs on one accumulator are 8 MFMAs apart here, and the kernel drains the pipe (s_nop) before the
// epilogue reads the accumulators.
template <int FMT>
struct klm_mx_frag;
template <>
struct klm_mx_frag<0> { typedef i32x8 type; };     // fp8: 32 bytes per lane (K = 128 in one instruction)
template <>
struct klm_mx_frag<4> { typedef i32x4 type; };     // fp4: 16 bytes per lane per k step (the instruction reads v[0:3])

template <int FMT>
struct klm_op_mx {
  typedef typename klm_mx_frag<FMT>::type frag;
  f32x4 acc[2][2][4][2];
  frag a[2][4];
  frag b[2][2][2];        // [register set][ks][n tile]
  int c0, c1, a_row, b_row, scale;

  __device__ __forceinline__ void init(int lane, int wr, int wc) {
    const int fr = lane & 15, fq = lane >> 4, key = lane & 7;
    c0 = (fq ^ key) << 4;
    c1 = c0 ^ 64;
    a_row = (wr * 64 + fr) * 128;
    b_row = (wc * 32 + fr) * 128;
    scale = KLM_UNIT_SCALE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[i][j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // fp8 uses ks = 0 only (chunks g and g + 4 side by side in one 8-register fragment); fp4 one 4-register fragment per ks
  __device__ __forceinline__ void load_frag(frag& f0, frag& f1, const char* base) const {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(base + c0);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(base + c1);
    if constexpr (FMT == 0) {
      f0 = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else {
      f0 = lo;
      f1 = hi;
    }
  }
  __device__ __forceinline__ void read_a(const char* slot) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) load_frag(a[0][mi], a[1][mi], slot + a_row + mi * 2048);
  }
  template <int SET>
  __device__ __forceinline__ void read_b(const char* slot) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) load_frag(b[SET][0][ni], b[SET][1][ni], slot + b_row + ni * 2048);
  }
  __device__ __forceinline__ void mfma1(f32x4& c, const frag& x, const frag& y) const {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass of hipcc parses this body too and knows no "v" registers
    if constexpr (FMT == 0)
      asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(x), "v"(y), "v"(scale));
    else
      asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                   : "+v"(c) : "v"(x), "v"(y), "v"(scale));
#endif
  }
  template <int AH, int BH, int SET>
  __device__ __forceinline__ void mfma() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < (FMT == 0 ? 1 : 2); ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) mfma1(acc[AH][BH][mi][ni], b[SET][ks][ni], a[ks][mi]);
    __builtin_amdgcn_s_setprio(0);
  }
};

// The operand policy of the 128 x 128 eight-wave ring kernel (gemm_p128.h) for the MX formats: the same chunks as klm_op_mx
// above; an accumulator is touched once (fp8) or twice, eight MFMAs apart (fp4), per K tile.
template <int FMT, int NI_>
struct klm_p128_mx {
  static constexpr int NI = NI_;
  typedef typename klm_mx_frag<FMT>::type frag;
  struct frags {
    frag a[2][4], b[2][NI_];     // [ks][tile]; fp8 uses ks = 0 only
  };
  f32x4 acc[4][NI_];
  int c0, c1, a_row, b_row;
  __device__ __forceinline__ void init(int lane, int wr, int wc, int b_base) {
    const int fr = lane & 15, fq = lane >> 4, key = lane & 7;
    c0 = (fq ^ key) << 4;
    c1 = c0 ^ 64;
    a_row = (wr * 64 + fr) * 128;
    b_row = b_base + (wc * 16 * NI_ + fr) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NI_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void load_frag(frag& f0, frag& f1, const char* base) const {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(base + c0);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(base + c1);
    if constexpr (FMT == 0) {
      f0 = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else {
      f0 = lo;
      f1 = hi;
    }
  }
  __device__ __forceinline__ void read(frags& f, const char* st) const {
#pragma unroll
    for (int ni = 0; ni < NI_; ++ni) load_frag(f.b[0][ni], f.b[1][ni], st + b_row + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) load_frag(f.a[0][mi], f.a[1][mi], st + a_row + mi * 2048);
  }
  // The BUILTIN, not the tied-accumulator asm of klm_op_mx: this kernel's two tails (odd / even K tile count) make the compiler
  // copy accumulators between register sets, and a VALU read of an accumulator an asm MFMA is still writing is a hazard only
  // the compiler's recognizer can pad (seen on hardware: wrong sums for odd tile counts). With literal zero scale operands the
  // compiler selects the plain v_mfma_f32_16x16x128_f8f6f4 (scales 2^0 by definition); 19 register moves in the whole kernel.
  __device__ __forceinline__ void mfma1(f32x4& c, const frag& x, const frag& y) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (FMT == 0) {
      c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, c, 0, 0, 0, 0, 0, 0);
    } else {
      const i32x8 x8 = i32x8{x[0], x[1], x[2], x[3], 0, 0, 0, 0}, y8 = i32x8{y[0], y[1], y[2], y[3], 0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x8, y8, c, 4, 4, 0, 0, 0, 0);
    }
#endif
  }
  __device__ __forceinline__ void mfma(const frags& f) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < (FMT == 0 ? 1 : 2); ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI_; ++ni) mfma1(acc[mi][ni], f.b[ks][ni], f.a[ks][mi]);
    __builtin_amdgcn_s_setprio(0);
  }
};
template <int NI_>
using klm_p128_fp8 = klm_p128_mx<0, NI_>;
template <int NI_>
using klm_p128_fp4 = klm_p128_mx<4, NI_>;

// K is the row length in BYTES.
template <int FMT, int SCHED>
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
  klm_stage_dense st;
  st.init(A, B, brow, bcol, (unsigned)K, (unsigned)K, lane, wid, smem);

  klm_op_mx<FMT> op;
  op.init(lane, wr, wc);
  klm_ring256_mainloop<SCHED>(op, st, smem, K >> 7, wr);
  // the asm MFMAs are invisible to the hazard recognizer: let the last ones leave the pipe before the epilogue reads acc
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  const int fr = lane & 15, fq = lane >> 4;
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
          store_c4(C, (size_t)m * N + n, op.acc[ah][bh][mi][ni]);
        }
}

// variant 0: schedule 0 (round 1's); variant 1: one half-tile per phase (needs an even number of K tiles >= 4,
// else it runs schedule 0).
template <int FMT>
static int launch_mx(const void* A, const void* B, void* C, int M, int N, int Kbytes, int variant, void* stream) {
  if (!klm_stream_on_current_device((hipStream_t)stream)) return KLM_ERR_ARG;
  const int grid = (M >> 8) * (N >> 8);
  if (variant == 1 && (Kbytes & 255) == 0 && Kbytes >= 512) {
    if (int e = klm_ensure_dyn_lds<gemm_mx_256<FMT, 1>>(KLM_RING_LDS_BYTES)) return e;
    hipLaunchKernelGGL((gemm_mx_256<FMT, 1>), dim3(grid), dim3(512), KLM_RING_LDS_BYTES, (hipStream_t)stream,
                       (const uint8_t*)A, (const uint8_t*)B, (uint16_t*)C, M, N, Kbytes);
  } else {
    if (int e = klm_ensure_dyn_lds<gemm_mx_256<FMT, 0>>(KLM_RING_LDS_BYTES)) return e;
    hipLaunchKernelGGL((gemm_mx_256<FMT, 0>), dim3(grid), dim3(512), KLM_RING_LDS_BYTES, (hipStream_t)stream,
                       (const uint8_t*)A, (const uint8_t*)B, (uint16_t*)C, M, N, Kbytes);
  }
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}

static int check_mx(const void* A, const void* B, const void* C, int M, int N, int Kbytes, int variant) {
  if (!A || !B || !C) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
  if (variant != 0 && variant != 1) return KLM_ERR_VARIANT;
  if (M <= 0 || N <= 0 || Kbytes <= 0 || (M & 255) || (N & 255) || (Kbytes & 127) || Kbytes < 256) return KLM_ERR_SHAPE;
  return KLM_OK;
}

extern "C" int klm_gemm_fp8(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                            void* stream) {
#if defined(KLM_GEMM_DIAG)
  const bool stream_variant = variant >= 2 && variant <= 6;
#else
  const bool stream_variant = variant == 2 || variant == 4 || variant == 5;      // 3 and 6 are A/B-record streams: diagnostic builds only
#endif
  if (stream_variant) {
    // the hand-scheduled 4-wave stream (gemm_quad_asm.h, generator: body_fp8); 3: another placement, for the A/B record;
    // 4: the persistent form (one workgroup per CU, generator: body_p)
    if (!A || !B || !C) return KLM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
    if (!klm_quad_asm_shape_ok(M, N, K)) return KLM_ERR_SHAPE;
    if (!klm_stream_on_current_device((hipStream_t)stream)) return KLM_ERR_ARG;
    if (int e = variant == 2   ? klm_launch_quad_asm<30>(A, B, C, M, N, K, (hipStream_t)stream)
                : variant == 4 ? klm_launch_quad_asm_p<30>(A, B, C, M, N, K, (hipStream_t)stream)
#if defined(KLM_GEMM_DIAG)
                : variant == 3 ? klm_launch_quad_asm<31>(A, B, C, M, N, K, (hipStream_t)stream)
                : variant == 6 ? klm_launch_quad_asm_p<32>(A, B, C, M, N, K, (hipStream_t)stream)    // 6: 4 on the block-scaled MFMA with unit scales (what 4 used before: same bits, slower)
#endif
                               : klm_launch_quad_asm_p<31>(A, B, C, M, N, K, (hipStream_t)stream)) return e;   // 5: 4 with C streamed (sc1 nt)
    return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
  }
  if (variant == 14 || variant == 16 || variant == 17) {
    // 128 x 128 / 256 x 128 / 128 x 256 tiles on eight waves, LDS-DMA ring (gemm_p128.h): shapes of too few 256 x 256 tiles
    if (!A || !B || !C) return KLM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
    if (!klm_stream_on_current_device((hipStream_t)stream)) return KLM_ERR_ARG;
    if (int e = klm_launch_p128_variant<klm_p128_fp8>(variant, A, B, C, M, N, K, (hipStream_t)stream)) return e;
    return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
  }
  if (int e = check_mx(A, B, C, M, N, K, variant)) return e;
  return launch_mx<0>(A, B, C, M, N, K, variant, stream);
}

// The hand-scheduled MX streams on operands whose rows are `ld_bytes` apart (A and B alike). kind 1: fp8, 2: fp4;
// Kbytes = bytes of a row's K extent (fp8: K, fp4: K / 2).
extern "C" int klm_gemm_mx_pitched(int kind, const void* A, const void* B, void* C, int M, int N, int Kbytes, int ld_bytes,
                                   int persistent, void* stream) {
  if (!A || !B || !C) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
  if (kind != 1 && kind != 2) return KLM_ERR_VARIANT;
  if (!klm_quad_asm_shape_ok(M, N, Kbytes, ld_bytes)) return KLM_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (!klm_stream_on_current_device(s)) return KLM_ERR_ARG;
  int e;
  // persistent: 0 one workgroup per tile, 1 persistent, 2 persistent with C streamed past the caches (sc1 nt)
  if (kind == 1)
    e = persistent == 2   ? klm_launch_quad_asm_p<31>(A, B, C, M, N, Kbytes, s, ld_bytes)
        : persistent == 1 ? klm_launch_quad_asm_p<30>(A, B, C, M, N, Kbytes, s, ld_bytes)
                          : klm_launch_quad_asm<30>(A, B, C, M, N, Kbytes, s, ld_bytes);
  else
    e = persistent == 2   ? klm_launch_quad_asm_p<21>(A, B, C, M, N, Kbytes, s, ld_bytes)
        : persistent == 1 ? klm_launch_quad_asm_p<20>(A, B, C, M, N, Kbytes, s, ld_bytes)
                          : klm_launch_quad_asm<20>(A, B, C, M, N, Kbytes, s, ld_bytes);
  if (e) return e;
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}

// K counts fp4 ELEMENTS; rows are K/2 bytes, two elements per byte.
extern "C" int klm_gemm_fp4(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                            void* stream) {
  if (K <= 0 || (K & 1)) return KLM_ERR_SHAPE;
#if defined(KLM_GEMM_DIAG)
  const bool stream_variant = variant >= 2 && variant <= 5;
#else
  const bool stream_variant = variant >= 2 && variant <= 4;                      // 5 is an A/B-record stream: diagnostic builds only
#endif
  if (stream_variant) {
    // the hand-scheduled 4-wave stream (gemm_quad_asm.h): same fragments and K tile as bf16, the MX MFMA with unit scales;
    // 3: its persistent form; 4: 3 with C streamed (sc1 nt)
    if (!A || !B || !C) return KLM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
    if (!klm_quad_asm_shape_ok(M, N, K >> 1)) return KLM_ERR_SHAPE;
    if (!klm_stream_on_current_device((hipStream_t)stream)) return KLM_ERR_ARG;
    if (int e = variant == 2   ? klm_launch_quad_asm<20>(A, B, C, M, N, K >> 1, (hipStream_t)stream)
                : variant == 3 ? klm_launch_quad_asm_p<20>(A, B, C, M, N, K >> 1, (hipStream_t)stream)
#if defined(KLM_GEMM_DIAG)
                : variant == 5 ? klm_launch_quad_asm_p<22>(A, B, C, M, N, K >> 1, (hipStream_t)stream)   // 5: 3 on the block-scaled MFMA with unit scales (same bits, slower)
#endif
                               : klm_launch_quad_asm_p<21>(A, B, C, M, N, K >> 1, (hipStream_t)stream)) return e;   // 4: 3 with C streamed (sc1 nt)
    return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
  }
  if (variant == 14 || variant == 16 || variant == 17) {
    if (!A || !B || !C) return KLM_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
    if (!klm_stream_on_current_device((hipStream_t)stream)) return KLM_ERR_ARG;
    if (int e = klm_launch_p128_variant<klm_p128_fp4>(variant, A, B, C, M, N, K >> 1, (hipStream_t)stream)) return e;
    return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
  }
  if (int e = check_mx(A, B, C, M, N, K >> 1, variant)) return e;
  return launch_mx<4>(A, B, C, M, N, K >> 1, variant, stream);
}
