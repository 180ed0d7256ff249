// The hand-scheduled 4-wave 256 x 256 kernels: kernel template + launcher around the generated instruction streams
// (gemm_quad_asm.inc, from scripts/gen_gemm_quad_asm.py). Shared by the bf16 and fp4 burners.
#pragma once
#include "gemm_quad128.h"
#include "gemm_quad_asm.inc"
#if defined(KLM_GEMM_DIAG)
#include "gemm_quad_asm_diag.inc"   // scripts/gen_gemm_quad_asm.py --diag: A/B and diagnostic streams, never in the shipped library
#endif

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
  else if constexpr (BODY == 20) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_FP4);
  else if constexpr (BODY == 30) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_FP8);
#if defined(KLM_GEMM_DIAG)
  else if constexpr (BODY == 1) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_SPREAD);
  else if constexpr (BODY == 2) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_ONEBAR);
  else if constexpr (BODY == 3) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_PEROP);
  else if constexpr (BODY == 10) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_NODMA);
  else if constexpr (BODY == 11) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_NOREAD);
  else if constexpr (BODY == 12) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_L2ONLY);
  else if constexpr (BODY == 13) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_MFMAONLY);
  else if constexpr (BODY == 31) KLM_QUAD_ASM(KLM_QUAD_ASM_BODY_FP8B);
#endif
  else static_assert(BODY == 0, "this stream is a diagnostic one: build with -DKLM_GEMM_DIAG");
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
// Persistent form: `grid` workgroups (one per CU) each walk a list of UNITS as ONE stream of K tiles (generator: body_p).
// A unit is a K range of one output tile:
//   * a whole tile -- tiles bid, bid + grid, .. of the first R = ntiles / grid full rounds: the tile -> XCD / round map of
//     the kernel above under in-order dispatch. The last pass of a unit already fetches the next unit's first two K tiles,
//     so the epilogue's stores (32 MB per round on the chip) never sit in front of a cold prologue, and launch / argument
//     loads / set-up are paid once per CU;
//   * one K slice of a tile of the last, PARTIAL round (stream-K style): when ntiles is no multiple of the grid, the r
//     remaining tiles are cut into S K slices each, one per workgroup, so the round costs 1 / S of a full one instead of a
//     whole one. A slice stores its fp32 accumulators to a workspace slot; quad_partial_reduce adds the S slots of a tile
//     and writes bf16 C (KlmQuadPlan, klm_quad_plan: split only where the saved time exceeds the partial traffic);
//   * one of two segments of a tile whose K loop starts in its middle (plan.rot): dense operands whose row pitch the
//     memory channels camp on (ops.camping_pitch). The workgroups of XCD x (bid & 7 under round-robin dispatch) start at
//     K pass (x * rot) % P, run to the end (no epilogue), wrap to pass 0 and continue into the same accumulators: at any
//     one time the eight XCDs read eight different K offsets, i.e. different channels, while the workgroups of one XCD --
//     the ones that share A and B panels through its L2 -- stay in step. (Staggering every workgroup was measured first:
//     it breaks that sharing and LOSES 10 - 25 %, profiles/r3_prof/gemm_units_first.json.) Same products, another
//     summation order than the plain stream.
// The HIP part writes this workgroup's unit table behind the two LDS stages: {A rows, B rows, destination, passes, flags}.
// Not for a GPU shared with another stream's kernels (a collective overlapping the GEMM): units are assigned to workgroups
// statically, so a workgroup that finds its CU taken starts late and finishes late; models/linear_step.py uses the
// one-workgroup-per-tile kernel under data parallelism.
// BODY: 0 bf16, 20 fp4, 30 fp8; + 1: C stored with `sc1 nt`; under KLM_GEMM_DIAG also + 2 (fp4, fp8): on the block-scaled MFMA
// with unit scales (the A/B record); 10 / 11: bf16 without its stores / without its epilogue (C is not written).
// ------------------------------------------------------------------------------------
#define KLM_QUAD_P_MAX_TILES 512     // units per workgroup: 16 KiB of table
#define KLM_QUAD_SLOT_BYTES 262144   // one unit's fp32 partial sums: 256 x 256 x 4
#define KLM_QUAD_F_ACC 1u
#define KLM_QUAD_F_NOEPI 2u
#define KLM_QUAD_F_PARTIAL 4u

struct KlmQuadPlan {
  int grid;      // workgroups
  int rounds;    // R: full rounds of whole tiles (tile bid + i * grid, i < R)
  int rest;      // r: tiles of the last, partial round (0: none)
  int split;     // S: K slices per tile of the partial round (0 / 1: the partial round runs whole tiles on the first r workgroups)
  int rot;       // K-loop rotation step in passes (0: every tile starts at K = 0)
  int map;       // 0: tile_of_block (each XCD walks its own contiguous chunk of the tile list); 1: super-blocks (below)
};

// map 1, for 256 workgroups and tile counts that are multiples of 16 both ways: round w of the chip is ONE 16 x 16 super-block
// of output tiles (super-blocks row-major, so consecutive rounds share their A panels), inside it XCD x owns an 8 x 4 patch
// (patches 2 down x 4 across), inside the patch workgroup j = bid >> 3 takes tile (j & 7, j >> 3). What the chip touches in
// one round is 16 tile rows of A and 16 tile rows of B -- 200 MB at K = 12288 bf16, inside the 256 MB Infinity Cache -- where
// map 0's eight far-apart chunks touch up to 500 MB.
__device__ __forceinline__ void klm_quad_superblock_tile(int bid, int w, int tiles_n, int& tm, int& tn) {
  const int xcd = bid & 7, j = bid >> 3;
  const int nsc = tiles_n >> 4;
  const int sr = w / nsc, sc = w - sr * nsc;
  tm = (sr << 4) + ((xcd & 1) << 3) + (j & 7);
  tn = (sc << 4) + ((xcd >> 1) << 2) + (j >> 3);
}

// how many units workgroup `bid` walks under `p` (host and device)
__host__ __device__ static inline int klm_quad_units_of(const KlmQuadPlan& p, int bid, int passes) {
  int whole = p.rounds + ((p.split <= 1 && bid < p.rest) ? 1 : 0);
  if (p.rot) {
    // a rotated tile is two units unless its start falls on pass 0
    const int k0 = (int)(((long long)(bid & 7) * p.rot) % passes);
    whole *= (k0 ? 2 : 1);
  }
  return whole + ((p.split > 1 && bid < p.rest * p.split) ? 1 : 0);
}

template <int BODY>
__global__ __launch_bounds__(256) void gemm_quad_asm_p(const void* __restrict__ A, const void* __restrict__ B,
                                                       uint16_t* __restrict__ C, int M, int N, int Kb, int ld,
                                                       KlmQuadPlan plan, char* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const unsigned lda = (unsigned)ld, ldc = (unsigned)N * 2;
  const int tiles_m = M >> 8, tiles_n = N >> 8, grid = plan.grid, bid = (int)blockIdx.x;
  const int P = (Kb >> 7) >> 1;                                            // passes of two K tiles per whole tile
  const int whole = plan.rounds + ((plan.split <= 1 && bid < plan.rest) ? 1 : 0);
  const int k0 = plan.rot ? (int)(((long long)(bid & 7) * plan.rot) % P) : 0;   // this workgroup's K-loop start, in passes
  const int per_tile = k0 ? 2 : 1;
  const int mine = klm_quad_units_of(plan, bid, P);                        // >= 1: the launcher sizes the grid so
  uint4* table = reinterpret_cast<uint4*>(smem + KLM_QUAD_LDS_BYTES);
  for (int i = tid; i < mine; i += 256) {
    int tile, p0, np;
    unsigned flags = 0;
    uint64_t dst;
    int tm, tn;
    if (i < whole * per_tile) {
      const int w = i / per_tile, seg = i - w * per_tile;
      tile = bid + w * grid;
      if (plan.map == 1) klm_quad_superblock_tile(bid, w, tiles_n, tm, tn);
      else tile_of_block<8>(tile, tiles_m, tiles_n, tm, tn);
      if (per_tile == 1) {
        p0 = 0, np = P;
      } else if (seg == 0) {
        p0 = k0, np = P - k0, flags = KLM_QUAD_F_NOEPI;
      } else {
        p0 = 0, np = k0, flags = KLM_QUAD_F_ACC;
      }
      dst = reinterpret_cast<uint64_t>(C) + (uint64_t)(tm << 8) * ldc + (uint64_t)(tn << 8) * 2;
    } else {
      // this workgroup's K slice of a tile of the partial round: slices as even as the pass count allows
      const int j = bid / plan.split, sl = bid - j * plan.split;
      tile = plan.rounds * grid + j;
      tile_of_block<8>(tile, tiles_m, tiles_n, tm, tn);
      const int base = P / plan.split, rem = P - base * plan.split;
      np = base + (sl < rem ? 1 : 0);
      p0 = sl * base + (sl < rem ? sl : rem);
      flags = KLM_QUAD_F_PARTIAL;
      dst = reinterpret_cast<uint64_t>(ws) + (uint64_t)(j * plan.split + sl) * KLM_QUAD_SLOT_BYTES;
    }
    const uint64_t pa = reinterpret_cast<uint64_t>(A) + (uint64_t)(tm << 8) * lda + (uint64_t)p0 * 256u;
    const uint64_t pb = reinterpret_cast<uint64_t>(B) + (uint64_t)(tn << 8) * lda + (uint64_t)p0 * 256u;
    table[2 * i] = make_uint4((unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, (unsigned)pb, (unsigned)(pb >> 32) & 0xffffu);
    table[2 * i + 1] = make_uint4((unsigned)dst, (unsigned)(dst >> 32) & 0xffffu, (unsigned)np, flags);
  }
  __syncthreads();
  auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  // this wave's part of a tile: it stages rows [64 wid, 64 wid + 64) of both operands and owns a 128 x 128 part of C
  // (or its 64 KiB of a workspace slot)
  const unsigned dab = uni((unsigned)wid * 64u * lda), dc = uni((unsigned)wr * 128u * ldc + (unsigned)wc * 256u);
  const unsigned dw = uni((unsigned)wid * 65536u);
  const unsigned ra2 = uni(64u * lda), rc2 = uni(128u * ldc), r3 = 0x00020000u;
  const unsigned ldsbase = (unsigned)(uintptr_t)(KLM_AS3 char*)smem;
  const unsigned lw = uni(ldsbase + (unsigned)wid * 8192u);
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
  const unsigned vw = (unsigned)lane * 16u;
#define KLM_QUAD_ASM_P(BODY_TEXT)                                                                                              \
  asm volatile(BODY_TEXT                                                                                                       \
               :                                                                                                               \
               : [ra2] "s"(ra2), [ra3] "s"(r3), [rc2] "s"(rc2), [lda] "s"(lda), [ldc] "s"(ldc),                                 \
                 [lw] "s"(lw), [wid] "s"(wid), [dab] "s"(dab), [dc] "s"(dc), [dw] "s"(dw), [ntl] "s"(ntl), [tbl] "v"(tbl),       \
                 [va] "v"(va), [vc] "v"(vc), [vw] "v"(vw), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1)                 \
               : KLM_QUAD_ASM_CLOBBERS)
  if constexpr (BODY == 0) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P);
  else if constexpr (BODY == 1) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_ST);
  else if constexpr (BODY == 20) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP4);
  else if constexpr (BODY == 21) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP4_ST);
  else if constexpr (BODY == 30) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP8);
  else if constexpr (BODY == 31) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP8_ST);
#if defined(KLM_GEMM_DIAG)
  else if constexpr (BODY == 10) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_NOSTORE);
  else if constexpr (BODY == 11) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_NOEPI);
  else if constexpr (BODY == 22) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP4_SC);
  else if constexpr (BODY == 32) KLM_QUAD_ASM_P(KLM_QUAD_ASM_BODY_P_FP8_SC);
#endif
  else static_assert(BODY == 0, "this stream is a diagnostic one: build with -DKLM_GEMM_DIAG");
#undef KLM_QUAD_ASM_P
}

// The S workspace slots of every tile of the partial round -> bf16 C. A slot is register-major (generator: epilogue32_lines):
// float4 q of a slot belongs to wave q >> 12, accumulator tile t = (q >> 6) & 63 = 8 mi + ni, lane q & 63, and holds row
// wr * 128 + mi * 16 + (lane & 15), columns wc * 128 + ni * 16 + (lane >> 4) * 4 .. + 3 of the 256 x 256 output tile.
// One block per 256 float4 (4 accumulator tiles of one wave): 4 KiB contiguous per slot and block; the S slots are summed
// in slice order in fp32.
static __global__ __launch_bounds__(256) void quad_partial_reduce(const float4* __restrict__ ws, uint16_t* __restrict__ C, int M, int N,
                                                           int first_tile, int split) {
  const int j = (int)blockIdx.x >> 6;                               // tile of the partial round
  const int q = (((int)blockIdx.x & 63) << 8) + (int)threadIdx.x;   // float4 within a slot
  const float4* src = ws + (size_t)j * split * (KLM_QUAD_SLOT_BYTES / 16) + q;
  float4 acc = src[0];
  for (int s = 1; s < split; ++s) {
    const float4 v = src[(size_t)s * (KLM_QUAD_SLOT_BYTES / 16)];
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  int tm, tn;
  tile_of_block<8>(first_tile + j, M >> 8, N >> 8, tm, tn);
  const int wid = q >> 12, t = (q >> 6) & 63, lane = q & 63;
  const int row = (tm << 8) + (wid >> 1) * 128 + (t >> 3) * 16 + (lane & 15);
  const int col = (tn << 8) + (wid & 1) * 128 + (t & 7) * 16 + (lane >> 4) * 4;
  store_c4(C, (size_t)row * N + col, f32x4{acc.x, acc.y, acc.z, acc.w});
}

// tests: a fixed workgroup count (0 = one per CU), so that a small problem walks several tiles per workgroup on any device
inline int& klm_quad_p_grid_override() {
  static int g = 0;
  return g;
}

// workgroups the persistent kernels run on: one per CU of the current device (queried once per device), or the tests' override
static int klm_quad_p_cus() {
  if (const int g = klm_quad_p_grid_override(); g > 0) return g;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  int n = __atomic_load_n(&cus[dev], __ATOMIC_ACQUIRE);
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
    __atomic_store_n(&cus[dev], n, __ATOMIC_RELEASE);
  }
  return n;
}

// What a split of the partial round saves and costs, in microseconds (measured on MI355X, profiles/r3_prof/splitk_model.json):
// a pass of two K tiles takes a workgroup ~2.2 us; the partial sums cost their traffic twice (written by the slices, read by
// the reduce kernel, mostly through the Infinity Cache) plus one more kernel launch. A deliberately cautious rule: the measured
// net gains (6144^3: 39 us, 1024 x 4096 x 4096: 22 us) are 0.75 - 0.85 of what it predicts.
#define KLM_SPLIT_PASS_US 2.2
#define KLM_SPLIT_MB_US 0.30          // per MB of partial sums, write + read
#define KLM_SPLIT_LAUNCH_US 8.0

// The plan for an M x N x Kb problem on `cus` workgroups. mode: 0 never split or rotate; 1 split the partial round where the
// model says it pays; 2 split whenever the shape allows (tests); `rot`: rotate the K loop (camping pitches).
static inline KlmQuadPlan klm_quad_plan(int M, int N, int Kb, int cus, int mode_flags, int rot, size_t ws_bytes) {
  const int mode = mode_flags & 3;
  const int ntiles = (M >> 8) * (N >> 8), P = (Kb >> 7) >> 1;
  const int need = (ntiles + KLM_QUAD_P_MAX_TILES - 1) / KLM_QUAD_P_MAX_TILES;     // the table holds 512 units per workgroup
  if (cus < need) cus = need;
  KlmQuadPlan p{};
  p.rounds = ntiles / cus;
  p.rest = ntiles - p.rounds * cus;
  p.grid = p.rounds ? cus : p.rest;                 // fewer tiles than CUs: one workgroup per tile
  p.split = 0;
  p.rot = 0;
  if (mode > 0 && p.rest > 0) {
    int S = cus / p.rest;
    if (S > 8) S = 8;
    if (S > P / 2) S = P / 2;                       // every slice at least two passes
    while (S > 1 && (size_t)p.rest * S * KLM_QUAD_SLOT_BYTES > ws_bytes) --S;
    if (S > 1) {
      const double saved = KLM_SPLIT_PASS_US * (P - (P + S - 1) / S);
      const double cost = KLM_SPLIT_LAUNCH_US + KLM_SPLIT_MB_US * (double)p.rest * S * (KLM_QUAD_SLOT_BYTES / 1048576.0);
      if (mode == 2 || saved > cost) {
        p.split = S;
        if (p.rounds == 0) p.grid = p.rest * S;     // nothing but slices: one workgroup each
      }
    }
  }
  if (rot > 0 && P >= 4 && p.split <= 1) p.rot = rot % P ? rot % P : 1;
  // mode bit 2: the super-block tile order, where the shape allows it
  p.map = ((mode_flags & 4) && cus == 256 && p.rest == 0 && p.rounds >= 2 && !((M >> 8) & 15) && !((N >> 8) & 15)) ? 1 : 0;
  return p;
}

template <int BODY>
static int klm_launch_quad_asm_p(const void* A, const void* B, void* C, int M, int N, int Kb, hipStream_t s, int ld = 0,
                                 int mode = 0, int rot = 0, void* ws = nullptr, size_t ws_bytes = 0) {
  const int cus = klm_quad_p_cus(), P = (Kb >> 7) >> 1;
  if (cus <= 0) return KLM_ERR_HIP;
  const KlmQuadPlan plan = klm_quad_plan(M, N, Kb, cus, ws ? mode : (mode & 4), rot, ws_bytes);
  if constexpr (BODY == 0 || BODY == 20 || BODY == 30) {
    // At most one whole tile per workgroup and nothing to split: every workgroup stores its 128 KiB of C at the same moment, at
    // the very end, and nothing follows that could use C from the caches. Stored with `sc1 nt` (BODY + 1; the same bits) the
    // round ends earlier: fp8 4096 x 4096 x 512 / 1024 / 4096: 14.8 -> 13.5, 19.0 -> 17.7, 47.3 -> 46.5 us; 3584^3 36.8 -> 35.2;
    // bf16 3072^3 54.2 -> 52.5 (scripts/exp/gemm_one_round_variants.py). In a multi-round problem the same stores cost up to
    // 12 % at small K (docs/KERNELS.md), so only here.
    if ((mode & 3) > 0 && plan.split <= 1 && plan.rot == 0 && plan.rounds + (plan.rest ? 1 : 0) == 1)
      return klm_launch_quad_asm_p<BODY + 1>(A, B, C, M, N, Kb, s, ld, 0, 0, nullptr, 0);
  }
  const int lds = KLM_QUAD_LDS_BYTES + 32 * KLM_QUAD_P_MAX_TILES;
  if (int e = klm_ensure_dyn_lds<gemm_quad_asm_p<BODY>>(lds)) return e;
  int per = 0;
  for (int b = 0; b < plan.grid; ++b) {
    const int u = klm_quad_units_of(plan, b, P);
    if (u < 1) return KLM_ERR_SHAPE;                // every workgroup of the grid has work, by construction
    per = u > per ? u : per;
  }
  if (per > KLM_QUAD_P_MAX_TILES) return KLM_ERR_SHAPE;
  hipLaunchKernelGGL(gemm_quad_asm_p<BODY>, dim3(plan.grid), dim3(256), KLM_QUAD_LDS_BYTES + 32 * per, s, A, B, (uint16_t*)C, M, N, Kb,
                     ld ? ld : Kb, plan, (char*)ws);
  if (plan.split > 1)
    hipLaunchKernelGGL(quad_partial_reduce, dim3(plan.rest * 64), dim3(256), 0, s, (const float4*)ws, (uint16_t*)C, M, N,
                       plan.rounds * plan.grid, plan.split);
  return KLM_OK;
}

// shapes the hand-scheduled kernels take: whole 256 x 256 tiles, an even number of K tiles >= 4, 32-bit descriptor ranges
static inline bool klm_quad_asm_shape_ok(int M, int N, int Kb, int ld = 0) {
  if (!ld) ld = Kb;
  return M > 0 && N > 0 && !(M & 255) && !(N & 255) && !(Kb & 255) && Kb >= 512 && ld >= Kb && !(ld & 15) &&
         (size_t)ld * 256 <= 0x7fffffffull && (size_t)N * 2 * 128 <= 0x7fffffffull;
}


