// The hand-scheduled 4-wave 256 x 256 kernels: kernel template + launcher around the generated instruction streams
// (gemm_quad_asm.inc, from scripts/gen_gemm_quad_asm.py). Shared by the bf16 and fp4 burners.
#pragma once
#include "gemm_quad128.h"
#include "gemm_quad_asm.inc"

// ------------------------------------------------------------------------------------
// The same geometry (4 waves of 128 x 128, gemm_quad128.h's LDS image and swizzle) with the WHOLE instruction stream
// written out: scripts/gen_gemm_quad_asm.py generates gemm_quad_asm.inc (schedule and register map are documented
// there). HIP computes the descriptors and per-lane offsets; one asm statement does the rest, stores included.
// ------------------------------------------------------------------------------------

// BODY: 0 the bf16 kernel (schedule "stagger"); 1..3 earlier placements of the same instructions (spread, one_barrier,
// per_operand), kept for the A/B record; 10..13 diagnostic streams without the loop's DMA / fragment reads / K advance
// (DMA from L2 only) / all of it (results meaningless: they price each kind of instruction); 20 the fp4 kernel; 30 the fp8 kernel.
// Kb: the operands' row length in BYTES (bf16: 2 K, fp4: K / 2); a K tile is 128 bytes of a row in every format.
// ld: the row PITCH of A and B in bytes (>= Kb, a multiple of 16): dense operands have ld = Kb; a burner that owns its operands
// pads the pitch away from the strides the memory channels camp on (docs/KERNELS.md, "row pitch").
template <int BODY>
__global__ __launch_bounds__(256) void gemm_quad_asm(const void* __restrict__ A, const void* __restrict__ B,
                                                     uint16_t* __restrict__ C, int M, int N, int Kb, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  int tm, tn;
  tile_of_block<8>(blockIdx.x, M >> 8, N >> 8, tm, tn);
  const int brow = tm << 8, bcol = tn << 8;
  const unsigned lda = (unsigned)ld, ldc = (unsigned)N * 2;
  // descriptors start at this WAVE's first staged row (it stages rows [64 wid, 64 wid + 64) of both operands) and at
  // its 128 x 128 part of C, so every offset inside the asm is small and wave-independent
  const uint64_t pa = reinterpret_cast<uint64_t>(A) + (uint64_t)(brow + wid * 64) * lda;
  const uint64_t pb = reinterpret_cast<uint64_t>(B) + (uint64_t)(bcol + wid * 64) * lda;
  const uint64_t pc = reinterpret_cast<uint64_t>(C) + (uint64_t)(brow + wr * 128) * ldc + (uint64_t)(bcol + wc * 128) * 2;
  auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  const unsigned ra0 = uni((unsigned)pa), ra1 = uni((unsigned)(pa >> 32) & 0xffffu), ra2 = uni(64u * lda);
  const unsigned rb0 = uni((unsigned)pb), rb1 = uni((unsigned)(pb >> 32) & 0xffffu);
  const unsigned rc0 = uni((unsigned)pc), rc1 = uni((unsigned)(pc >> 32) & 0xffffu), rc2 = uni(128u * ldc);
  const unsigned r3 = 0x00020000u;
  const unsigned ldsbase = (unsigned)(uintptr_t)(KLM_AS3 char*)smem;
  const unsigned lw = uni(ldsbase + (unsigned)wid * 8192u);
  const unsigned passes = uni((unsigned)((Kb >> 7) - 2) >> 1);
  // per lane: staging source offset (row lane >> 3 of an 8-row piece, swizzled 16-byte chunk) ...
  const unsigned srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned va = srow * lda + schunk * 16;
  // ... fragment read addresses (row lane & 15 of a 16-row tile, chunk k-step * 4 + (lane >> 4), swizzled by row & 7) ...
  const unsigned fr = lane & 15, fq = lane >> 4, key = lane & 7;
  const unsigned ch0 = ((fq ^ key) << 4), ch1 = (((4 + fq) ^ key) << 4);
  const unsigned a0 = ldsbase + ((unsigned)wr * 128 + fr) * 128 + ch0, a1 = ldsbase + ((unsigned)wr * 128 + fr) * 128 + ch1;
  const unsigned b0 = ldsbase + KLM_QUAD_B_OFFSET + ((unsigned)wc * 128 + fr) * 128 + ch0;
  const unsigned b1 = ldsbase + KLM_QUAD_B_OFFSET + ((unsigned)wc * 128 + fr) * 128 + ch1;
  // ... and where its 4 consecutive columns of an accumulator tile go in C
  // (after the epilogue's lane exchange: 8 consecutive columns of n tile (pair + (fq & 1)), its half fq >> 1)
  const unsigned vc = fr * ldc + (fq & 1) * 32 + (fq >> 1) * 16;
#define KLM_QUAD_ASM(BODY_TEXT)                                                                                                \
  asm volatile(BODY_TEXT                                                                                                       \
               :                                                                                                               \
               : [ra0] "s"(ra0), [ra1] "s"(ra1), [ra2] "s"(ra2), [ra3] "s"(r3), [rb0] "s"(rb0), [rb1] "s"(rb1), [rb2] "s"(ra2), \
                 [rb3] "s"(r3), [rc0] "s"(rc0), [rc1] "s"(rc1), [rc2] "s"(rc2), [rc3] "s"(r3), [lda] "s"(lda), [ldc] "s"(ldc), \
                 [passes] "s"(passes), [lw] "s"(lw), [wid] "s"(wid), [va] "v"(va), [vc] "v"(vc), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0),     \
                 [b1] "v"(b1)                                                                                                  \
               : KLM_QUAD_ASM_CLOBBERS)
  if constexpr (BODY == 0) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY);
  else if constexpr (BODY == 1) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_SPREAD);
  else if constexpr (BODY == 2) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_ONEBAR);
  else if constexpr (BODY == 3) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_PEROP);
  else if constexpr (BODY == 10) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_NODMA);
  else if constexpr (BODY == 11) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_NOREAD);
  else if constexpr (BODY == 12) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_L2ONLY);
  else if constexpr (BODY == 13) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_MFMAONLY);
  else if constexpr (BODY == 20) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_FP4);
  else if constexpr (BODY == 30) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_FP8);
  else KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_FP8B);
#undef KLM_QUAD_ASM
}

template <int BODY>
static int klm_launch_quad_asm(const void* A, const void* B, void* C, int M, int N, int Kb, hipStream_t s, int ld = 0) {
  if (int e = klm_ensure_dyn_lds<gemm_quad_asm<BODY>>(KLM_QUAD_LDS_BYTES)) return e;
  hipLaunchKernelGGL(gemm_quad_asm<BODY>, dim3((M >> 8) * (N >> 8)), dim3(256), KLM_QUAD_LDS_BYTES, s, A, B, (uint16_t*)C, M, N, Kb,
                     ld ? ld : Kb);
  return KLM_OK;
}

// ------------------------------------------------------------------------------------
// Persistent form: `grid` workgroups (one per CU) each walk output tiles bid, bid + grid, .. -- the tile -> XCD / round
// map of the kernel above under in-order dispatch -- as ONE stream of K tiles: the last pass of an output tile already
// fetches the next tile's first two K tiles, so the epilogue's stores (32 MB per round on the chip) never sit in front of
// a cold prologue, and launch / argument loads / set-up are paid once per CU (generator: body_p).
// The HIP part writes this workgroup's tile table behind the two LDS stages: {A rows, B rows, C} addresses per tile.
// Not for a GPU shared with another stream's kernels (a collective overlapping the GEMM): tiles are assigned to workgroups
// statically, so a workgroup that finds its CU taken starts late and finishes late; models/linear_step.py uses the
// one-workgroup-per-tile kernel under data parallelism.
// BODY: 0 bf16, 20 fp4, 30 fp8; + 1: C stored with `sc1 nt`; + 2 (fp4, fp8): on the block-scaled MFMA with unit scales (the A/B record); 10 / 11: bf16 without its stores / without its
// epilogue (diagnostic: C is not written).
// ------------------------------------------------------------------------------------
#define KLM_QUAD_P_MAX_TILES 512     // per workgroup: 16 KiB of table
template <int BODY>
__global__ __launch_bounds__(256) void gemm_quad_asm_p(const void* __restrict__ A, const void* __restrict__ B,
                                                       uint16_t* __restrict__ C, int M, int N, int Kb, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const unsigned lda = (unsigned)ld, ldc = (unsigned)N * 2;
  const int tiles_m = M >> 8, tiles_n = N >> 8, ntiles = tiles_m * tiles_n, grid = (int)gridDim.x;
  const int mine = (ntiles - (int)blockIdx.x + grid - 1) / grid;          // >= 1: the launcher keeps grid <= ntiles
  uint4* table = reinterpret_cast<uint4*>(smem + KLM_QUAD_LDS_BYTES);
  for (int i = tid; i < mine; i += 256) {
    int tm, tn;
    tile_of_block<8>((int)blockIdx.x + i * grid, tiles_m, tiles_n, tm, tn);
    const uint64_t pa = reinterpret_cast<uint64_t>(A) + (uint64_t)(tm << 8) * lda;
    const uint64_t pb = reinterpret_cast<uint64_t>(B) + (uint64_t)(tn << 8) * lda;
    const uint64_t pc = reinterpret_cast<uint64_t>(C) + (uint64_t)(tm << 8) * ldc + (uint64_t)(tn << 8) * 2;
    table[2 * i] = make_uint4((unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, (unsigned)pb, (unsigned)(pb >> 32) & 0xffffu);
    table[2 * i + 1] = make_uint4((unsigned)pc, (unsigned)(pc >> 32) & 0xffffu, 0u, 0u);
  }
  __syncthreads();
  auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  // this wave's part of a tile: it stages rows [64 wid, 64 wid + 64) of both operands and owns a 128 x 128 part of C
  const unsigned dab = uni((unsigned)wid * 64u * lda), dc = uni((unsigned)wr * 128u * ldc + (unsigned)wc * 256u);
  const unsigned ra2 = uni(64u * lda), rc2 = uni(128u * ldc), r3 = 0x00020000u;
  const unsigned ldsbase = (unsigned)(uintptr_t)(KLM_AS3 char*)smem;
  const unsigned lw = uni(ldsbase + (unsigned)wid * 8192u);
  const unsigned passes = uni((unsigned)(Kb >> 7) >> 1);
  const unsigned ntl = uni((unsigned)mine);
  const unsigned tbl = ldsbase + KLM_QUAD_LDS_BYTES;
  const unsigned srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned va = srow * lda + schunk * 16;
  const unsigned fr = lane & 15, fq = lane >> 4, key = lane & 7;
  const unsigned ch0 = ((fq ^ key) << 4), ch1 = (((4 + fq) ^ key) << 4);
  const unsigned a0 = ldsbase + ((unsigned)wr * 128 + fr) * 128 + ch0, a1 = ldsbase + ((unsigned)wr * 128 + fr) * 128 + ch1;
  const unsigned b0 = ldsbase + KLM_QUAD_B_OFFSET + ((unsigned)wc * 128 + fr) * 128 + ch0;
  const unsigned b1 = ldsbase + KLM_QUAD_B_OFFSET + ((unsigned)wc * 128 + fr) * 128 + ch1;
  const unsigned vc = fr * ldc + (fq & 1) * 32 + (fq >> 1) * 16;
#define KLM_QUAD_ASM_P(BODY_TEXT)                                                                                              \
  asm volatile(BODY_TEXT                                                                                                       \
               :                                                                                                               \
               : [ra2] "s"(ra2), [ra3] "s"(r3), [rc2] "s"(rc2), [lda] "s"(lda), [ldc] "s"(ldc), [passes] "s"(passes),           \
                 [lw] "s"(lw), [wid] "s"(wid), [dab] "s"(dab), [dc] "s"(dc), [ntl] "s"(ntl), [tbl] "v"(tbl), [va] "v"(va),        \
                 [vc] "v"(vc), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1)                                            \
               : KLM_QUAD_ASM_CLOBBERS)
  if constexpr (BODY == 0) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P);
  else if constexpr (BODY == 10) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_NOSTORE);
  else if constexpr (BODY == 11) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_NOEPI);
  else if constexpr (BODY == 1) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_ST);
  else if constexpr (BODY == 20) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP4);
  else if constexpr (BODY == 21) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP4_ST);
  else if constexpr (BODY == 22) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP4_SC);
  else if constexpr (BODY == 30) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP8);
  else if constexpr (BODY == 31) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP8_ST);
  else KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP8_SC);
#undef KLM_QUAD_ASM_P
}

// tests: a fixed workgroup count (0 = one per CU), so that a small problem walks several tiles per workgroup on any device
inline int& klm_quad_p_grid_override() {
  static int g = 0;
  return g;
}

// one workgroup per CU of the current device (queried once per device)
static int klm_quad_p_grid(int ntiles) {
  if (const int g = klm_quad_p_grid_override(); g > 0) {
    const int need = (ntiles + KLM_QUAD_P_MAX_TILES - 1) / KLM_QUAD_P_MAX_TILES;
    return g > ntiles ? ntiles : (g < need ? need : g);
  }
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  int n = __atomic_load_n(&cus[dev], __ATOMIC_ACQUIRE);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
    __atomic_store_n(&cus[dev], n, __ATOMIC_RELEASE);
  }
  int grid = ntiles < n ? ntiles : n;
  const int need = (ntiles + KLM_QUAD_P_MAX_TILES - 1) / KLM_QUAD_P_MAX_TILES;
  return grid < need ? need : grid;
}

template <int BODY>
static int klm_launch_quad_asm_p(const void* A, const void* B, void* C, int M, int N, int Kb, hipStream_t s, int ld = 0) {
  const int ntiles = (M >> 8) * (N >> 8), grid = klm_quad_p_grid(ntiles);
  if (grid <= 0) return KLM_ERR_HIP;
  const int lds = KLM_QUAD_LDS_BYTES + 32 * KLM_QUAD_P_MAX_TILES;
  if (int e = klm_ensure_dyn_lds<gemm_quad_asm_p<BODY>>(lds)) return e;
  const int per = (ntiles + grid - 1) / grid;
  hipLaunchKernelGGL(gemm_quad_asm_p<BODY>, dim3(grid), dim3(256), KLM_QUAD_LDS_BYTES + 32 * per, s, A, B, (uint16_t*)C, M, N, Kb,
                     ld ? ld : Kb);
  return KLM_OK;
}

// shapes the hand-scheduled kernels take: whole 256 x 256 tiles, an even number of K tiles >= 4, 32-bit descriptor ranges
static inline bool klm_quad_asm_shape_ok(int M, int N, int Kb, int ld = 0) {
  if (!ld) ld = Kb;
  return M > 0 && N > 0 && !(M & 255) && !(N & 255) && !(Kb & 255) && Kb >= 512 && ld >= Kb && !(ld & 15) &&
         (size_t)ld * 256 <= 0x7fffffffull && (size_t)N * 2 * 128 <= 0x7fffffffull;
}


