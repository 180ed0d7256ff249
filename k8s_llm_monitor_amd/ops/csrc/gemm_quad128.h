// 256 x 256 tile on FOUR waves (2 x 2), 128 x 128 of output per wave, one wave per SIMD (gfx950, CDNA4).
//
// No reference counterpart (the reference has no native code).
//
// Why a second pipeline beside gemm_ring256.h: rocprofv3 (profiles/r2_prof/pmc_gemm_ab_v1_v2_vendor.json) shows the
// 8-wave ring kernel MFMA-busy 0.80 of SIMD time with 25.2 M LDS reads per 8192^3 GEMM, and the vendor kernel
// (4 waves of 128 x 128) at 0.88 with 16.8 M: a 128 x 128 wave tile reads (8 + 8) fragments per 64 MFMAs where a
// 128 x 64 one reads (8 + 4) per 32, a third less LDS traffic and issue per FLOP, and under the power cap the
// chip's clock follows energy per FLOP. The price: 256 accumulator registers per lane, so one wave per SIMD and
// no partner wave to hide anything behind: the wave's own instruction stream interleaves LDS reads and
// LDS-DMA with its MFMAs (pinned with sched_group_barrier), fragments are double-buffered in registers, and there
// is ONE barrier per K tile.
//
// LDS: 2 stages x 64 KiB; a stage = A [256 rows][128 B] | B [256 rows][128 B] (a K tile is 128 bytes per row in
// every kernel: 64 bf16, 128 fp8, 256 fp4). Rows are lane-linear 8-row pieces (what LDS-DMA writes); the
// bank-conflict swizzle (16-byte chunk ^ row & 7) is applied to the source address and to the ds_read address.
// Wave w stages rows [64 w, 64 w + 64) of both operands: 16 DMA instructions per K tile.
//
// Per K tile t (stage t & 1) the work is two segments of 64 MFMA-equivalents, each with its own fragment set:
//   S0: MFMAs on set 0 (read earlier)            || ds_read tile t, part 1 -> set 1
//   -- s_waitcnt vmcnt(0) (tile t+1 landed: issued a whole tile ago), lgkmcnt(0), s_barrier --
//   S1: MFMAs on set 1                           || ds_read tile t+1, part 0 -> set 0  || LDS-DMA tile t+2 -> stage t & 1
// After the barrier every wave has finished reading tile t, so its stage can be refilled; the DMA then has the rest
// of S1 and all of S0(t+1) (~1.5 k cycles) to land. What a "part" is belongs to the operand policy:
//   OP  read1<SET, PART, G>(stage_base): fragment read G of 16 (ds_read_b128) into fragment set SET
//       mfma4<SET, PART, G>():           MFMAs 4G .. 4G+3 of the segment's 64
// bf16: part = k step of 32 (8 A + 8 B fragments, 64 x mfma_16x16x32).
#pragma once
#include "gemm_ring256.h"

#define KLM_QUAD_STAGE_BYTES 65536
#define KLM_QUAD_B_OFFSET 32768
#define KLM_QUAD_LDS_BYTES 131072

// LLVM SchedGroupMask bits
#define KLM_SG_MFMA 0x8
#define KLM_SG_VMEM 0x10
#define KLM_SG_DSREAD 0x100

struct klm_stage_quad {
  __amdgpu_buffer_rsrc_t ra, rb;
  unsigned va, vb, sa, sb, lda, ldb;
  char* lw;   // this wave's 8 KiB window of the A image (B image at + KLM_QUAD_B_OFFSET)

  __device__ __forceinline__ void init(const void* A, const void* B, int brow, int bcol, unsigned lda_, unsigned ldb_,
                                       int lane, int wid, char* smem) {
    lda = lda_;
    ldb = ldb_;
    const char* a0 = reinterpret_cast<const char*>(A) + (size_t)brow * lda;
    const char* b0 = reinterpret_cast<const char*>(B) + (size_t)bcol * ldb;
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a0), 0, (int)(256u * lda), 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b0), 0, (int)(256u * ldb), 0x00020000);
    const unsigned srow = lane >> 3, schunk = (lane & 7) ^ srow;
    va = srow * lda + schunk * 16;
    vb = srow * ldb + schunk * 16;
    sa = (unsigned)(wid * 64) * lda;
    sb = (unsigned)(wid * 64) * ldb;
    lw = smem + wid * 8192;
  }
  // piece P of 16: 0..7 = this wave's eight 8-row pieces of A, 8..15 = of B
  template <int P>
  __device__ __forceinline__ void issue(int stage, int tile) const {
    if constexpr (P < 8) {
      blds16(ra, va, sa + (unsigned)P * 8 * lda + (unsigned)tile * 128, lw + stage * KLM_QUAD_STAGE_BYTES + P * 1024);
    } else {
      blds16(rb, vb, sb + (unsigned)(P - 8) * 8 * ldb + (unsigned)tile * 128,
             lw + stage * KLM_QUAD_STAGE_BYTES + KLM_QUAD_B_OFFSET + (P - 8) * 1024);
    }
  }
  __device__ __forceinline__ void issue_all(int stage, int tile) const {
    issue<0>(stage, tile); issue<1>(stage, tile); issue<2>(stage, tile); issue<3>(stage, tile);
    issue<4>(stage, tile); issue<5>(stage, tile); issue<6>(stage, tile); issue<7>(stage, tile);
    issue<8>(stage, tile); issue<9>(stage, tile); issue<10>(stage, tile); issue<11>(stage, tile);
    issue<12>(stage, tile); issue<13>(stage, tile); issue<14>(stage, tile); issue<15>(stage, tile);
  }
};

// A segment = 16 groups of 4 MFMAs; after each group come at most one LDS-DMA and one or two ds_reads, in exactly this
// order. The MFMAs are inline asm with the accumulator tied in the AGPR class (`+a`): written as builtins, hipcc
// (ROCm 7.2) leaves destination and accumulator source untied, needs more than the 256 AGPRs the 64 accumulator tiles
// fill exactly, and shuffles accumulators through VGPRs (252 v_accvgpr_read + 252 v_accvgpr_write per two K tiles).
// Inline asm is opaque to the scheduler's instruction classes, so the order is pinned with sched_barrier(0) after
// every group instead of sched_group_barrier masks; the waits for fragment registers are still the compiler's
// (register dependencies of the asm operands), counted per use.
//
// Where things go (measured: two ds_read_b128 or two DMA back to back in one MFMA gap cost 12 %, and a barrier right
// after the last ds_read exposes that read's latency on every wave):
//   S0 group g:  read R[g] of this tile's part 1 -> set 1;  g < 2 also: the two stragglers of the previous S1
//                (R[14], R[15] of this tile's part 0 -> set 0, needed from group 12 on; DMA pieces 14, 15)
//   S1 groups 0, 1: MFMAs only, then  vmcnt(0) / lgkmcnt(0) / s_barrier  (the reads issued in S0 landed two groups ago)
//   S1 group g >= 2:  DMA piece g-2 of tile t+2 -> this stage;  read R[g-2] of tile t+1's part 0 -> set 0
template <int G, bool READ1, bool LEFT_R, bool LEFT_D, int S, class OP>
__device__ __forceinline__ void klm_quad_s0(OP& op, const klm_stage_quad& st, const char* cur, int t) {
  if constexpr (G < 16) {
    op.template mfma4<0, 0, G>();
    if constexpr (G < 2 && LEFT_D) st.template issue<14 + G>(S ^ 1, t + 1);
    if constexpr (READ1) op.template read1<1, 1, G>(cur);
    if constexpr (G < 2 && LEFT_R) op.template read1<0, 0, 14 + G>(cur);
    __builtin_amdgcn_sched_barrier(0);
    klm_quad_s0<G + 1, READ1, LEFT_R, LEFT_D, S>(op, st, cur, t);
  }
}

template <int G, bool NEXT, bool DMA, int S, class OP>
__device__ __forceinline__ void klm_quad_s1(OP& op, const klm_stage_quad& st, const char* nxt, int t) {
  if constexpr (G < 16) {
    op.template mfma4<1, 1, G>();
    if constexpr (G == 1) {
      KLM_VMCNT(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      KLM_BAR();
    }
    if constexpr (G >= 2 && DMA) st.template issue<G - 2>(S, t + 2);
    if constexpr (G >= 2 && NEXT) op.template read1<0, 0, G - 2>(nxt);
    __builtin_amdgcn_sched_barrier(0);
    klm_quad_s1<G + 1, NEXT, DMA, S>(op, st, nxt, t);
  }
}

// One K tile in stage S. NEXT: tile t+1 exists; DMA: tile t+2 exists; LEFT_D: the previous tile started a DMA batch.
template <int S, bool NEXT, bool DMA, bool LEFT_D, class OP>
__device__ __forceinline__ void klm_quad_tile(OP& op, const klm_stage_quad& st, const char* smem, int t) {
  const char* cur = smem + S * KLM_QUAD_STAGE_BYTES;
  const char* nxt = smem + (S ^ 1) * KLM_QUAD_STAGE_BYTES;
  klm_quad_s0<0, true, true, LEFT_D, S>(op, st, cur, t);
  klm_quad_s1<0, NEXT, DMA, S>(op, st, nxt, t);
}

// nt = K tiles, even and >= 4
template <class OP>
__device__ __forceinline__ void klm_quad_mainloop(OP& op, const klm_stage_quad& st, const char* smem, int nt) {
  st.issue_all(0, 0);
  st.issue_all(1, 1);
  KLM_VMCNT(16);   // tile 0 landed (tile 1's 16 pieces may still be in flight)
  KLM_BAR();
  op.template read_all<0, 0>(smem);
  __builtin_amdgcn_sched_barrier(0);
  // tile 0's S0 repeats two reads and two DMA pieces the prologue already did (same data): keeps the loop body uniform
  int t = 0;
  for (; t < nt - 2; t += 2) {
    klm_quad_tile<0, true, true, true>(op, st, smem, t);
    klm_quad_tile<1, true, true, true>(op, st, smem, t + 1);
  }
  klm_quad_tile<0, true, false, true>(op, st, smem, t);       // tile nt-2: stragglers of tile nt-1's batch, no new batch
  klm_quad_tile<1, false, false, false>(op, st, smem, t + 1); // tile nt-1
  // the asm MFMAs are invisible to the hazard recognizer: let the last ones drain before the epilogue reads a[]
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}
