// 128 x 128 output tile on EIGHT waves (2 x 4, 64 x 32 of output each: two waves per SIMD), a K tile of 128 BYTES per row,
// four-stage LDS-DMA ring: the kernel for shapes of fewer 256 x 256 tiles than the chip has CUs (2048^3 = 64 of them: three
// quarters of the chip idle on the 256 x 256 streams; as 256 tiles of 128 x 128 every CU works). A 128 x 128 tile moves twice
// the bytes per FLOP from L2 to LDS (32 KiB per K tile and 512 MFMA cycles = the 64 B/clk a CU can take), so this kernel cannot
// reach the streams' efficiency and does not try to hand-schedule: two waves per SIMD hide each other's LDS and barrier
// latency, and the MFMAs of K tile t-1 run while the fragments of tile t are read (two fragment sets), so after the ONE
// barrier per K tile each wave has MFMAs to issue.
//   LDS: 4 stages x (A [128 rows][128 B] | B [128 rows][128 B]) = 128 KiB, rows in lane-linear 8-row pieces, chunk ^ (row & 7).
//   per K tile t, every wave:  vmcnt(own pieces of t landed) . lgkmcnt(0) . barrier . 4 x LDS-DMA of tile t+3 -> the stage
//   tile t-1 lived in (every wave's reads of it have completed) . the fragment reads of tile t -> set t & 1 . the MFMAs of
//   tile t-1 from the other set.
// The operand policy OP says what 128 bytes of a row hold (bf16: 64 elements, two 16x16x32 MFMAs per 16 x 16 tile; fp8: 128,
// one 16x16x128; fp4: 256, two 16x16x128) -- klm_p128_bf16 in gemm_bf16.hip, klm_p128_mx<FMT> in gemm_mx.hip. Policies use
// MFMA BUILTINS: the odd / even tails make the compiler move accumulators, which only its hazard recognizer can make safe.
// Measured (profiles/r3_prof/gemm_small.json): bf16 2048^3 807 TFLOP/s (hipBLASLt 769, the 256 x 256 stream 465).
#pragma once
#include "gemm_ring256.h"

#define KLM_P128_STAGE 32768
#define KLM_P128_LDS (4 * KLM_P128_STAGE)

// Kb: bytes of a row's K extent (a multiple of 128); rows are dense.
template <class OP>
__global__ __launch_bounds__(512) void gemm_p128(const char* __restrict__ A, const char* __restrict__ B, uint16_t* __restrict__ C,
                                                 int M, int N, int Kb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  int tm, tn;
  tile_of_block<8>(blockIdx.x, M >> 7, N >> 7, tm, tn);
  const int brow = tm << 7, bcol = tn << 7;
  const unsigned ld = (unsigned)Kb;
  // staging: wave `wid` brings rows [16 wid, 16 wid + 16) of both operand tiles, two 8-row pieces each
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A + (size_t)brow * ld), 0, (int)(128u * ld), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B + (size_t)bcol * ld), 0, (int)(128u * ld), 0x00020000);
  const unsigned srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned voff = srow * ld + schunk * 16;
  const unsigned soff = (unsigned)(16 * wid) * ld;
  char* const lw = smem + wid * 2048;
  auto issue = [&](int tile) {
    char* l = lw + (tile & 3) * KLM_P128_STAGE;
    const unsigned so = soff + (unsigned)tile * 128u;
    blds16(ra, voff, so, l);
    blds16(ra, voff, so + 8 * ld, l + 1024);
    blds16(rb, voff, so, l + 16384);
    blds16(rb, voff, so + 8 * ld, l + 16384 + 1024);
  };
  OP op;
  op.init(lane, wr, wc);
  const int nt = Kb >> 7;
  typename OP::frags f0, f1;
  issue(0);
  if (nt > 1) issue(1);
  if (nt > 2) issue(2);
  // one K tile: `cur` receives tile t's fragments while the MFMAs of tile t-1 run on `prev`
  auto step = [&](int t, typename OP::frags& cur, const typename OP::frags& prev) {
    if (t + 2 < nt) KLM_VMCNT(8);          // this wave's pieces of tile t have landed (t+1, t+2 may be in flight)
    else if (t + 1 < nt) KLM_VMCNT(4);
    else KLM_VMCNT(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its reads of tile t-1 have completed
    KLM_BAR();                                            // true for every wave: tile t is whole, tile t-1's stage is free
    if (t + 3 < nt) issue(t + 3);
    op.read(cur, smem + (t & 3) * KLM_P128_STAGE);
    if (t > 0) op.mfma(prev);
  };
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    step(t, f0, f1);
    step(t + 1, f1, f0);
  }
  if (t < nt) {            // odd number of K tiles
    step(t, f0, f1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    op.mfma(f0);
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    op.mfma(f1);
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int m = brow + wr * 64 + mi * 16 + fr;
      const int n = bcol + wc * 32 + ni * 16 + fq * 4;
      store_c4(C, (size_t)m * N + n, op.acc[mi][ni]);
    }
}

// Host side: shapes are checked here (whole 128 x 128 tiles, whole K tiles, a row block within a buffer descriptor's 2 GiB).
template <class OP>
static int klm_launch_p128(const void* A, const void* B, void* C, int M, int N, int Kb, hipStream_t s) {
  if (M <= 0 || N <= 0 || Kb <= 0 || (M & 127) || (N & 127) || (Kb & 127) || (size_t)Kb * 128 > 0x7fffffffull) return KLM_ERR_SHAPE;
  if ((size_t)(M >> 7) * (size_t)(N >> 7) > 0x7fffffffull) return KLM_ERR_SHAPE;
  if (int e = klm_ensure_dyn_lds<gemm_p128<OP>>(KLM_P128_LDS)) return e;
  hipLaunchKernelGGL((gemm_p128<OP>), dim3((M >> 7) * (N >> 7)), dim3(512), KLM_P128_LDS, s, (const char*)A, (const char*)B, (uint16_t*)C, M, N, Kb);
  return KLM_OK;
}
