// TM x TN output tiles (128 x 128, 256 x 128, 128 x 256) on EIGHT waves (two per SIMD), a K tile of 128 BYTES per row, an
// LDS-DMA ring of as many stages as fit (four of 32 KiB / three of 48 KiB): the kernels for shapes of too few 256 x 256 tiles
// to fill the chip (2048^3 = 64 of them: three quarters of the chip idle on the 256 x 256 streams; as 256 tiles of 128 x 128
// every CU works; 2560^3 = 100, as 200 tiles of 256 x 128 one round at 78 %). A 128 x 128 tile moves twice the bytes per FLOP
// from L2 to LDS (32 KiB per K tile and 512 MFMA cycles = the 64 B/clk a CU can take; 256 x 128: 48 B/clk), so these kernels
// cannot reach the streams' efficiency and do not try to hand-schedule: two waves per SIMD hide each other's LDS and barrier
// latency, and the MFMAs of K tile t-1 run while the fragments of tile t are read (two fragment sets), so after the ONE
// barrier per K tile each wave has MFMAs to issue.
//   waves: (TM / 64) x (8 / (TM / 64)); each owns 64 rows x TN / WC columns = 4 x NI accumulator tiles (NI = 2 or 4).
//   LDS: S stages x (A [TM rows][128 B] | B [TN rows][128 B]), rows in lane-linear 8-row pieces, chunk ^ (row & 7).
//   per K tile t, every wave:  vmcnt(own pieces of t landed) . lgkmcnt(0) . barrier . LDS-DMA of tile t+S-1 -> the stage
//   tile t-1 lived in (every wave's reads of it have completed) . the fragment reads of tile t -> set t & 1 . the MFMAs of
//   tile t-1 from the other set.
// The operand policy OP says what 128 bytes of a row hold (bf16: 64 elements, two 16x16x32 MFMAs per 16 x 16 tile; fp8: 128,
// one 16x16x128; fp4: 256, two 16x16x128) -- klm_p128_bf16<NI> in gemm_bf16.hip, klm_p128_mx<FMT, NI> in gemm_mx.hip. Policies
// use MFMA BUILTINS: the odd / even tails make the compiler move accumulators, which only its hazard recognizer can make safe.
// Measured (profiles/r3_prof/gemm_small.json): bf16 2048^3 807 TFLOP/s (hipBLASLt 769, the 256 x 256 stream 465).
#pragma once
#include "gemm_ring256.h"

template <int N>
__device__ __forceinline__ void klm_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TM, int TN>
struct klm_ptile {
  static constexpr int WR = TM / 64, WC = 8 / WR, NI = TN / WC / 16;
  static constexpr int STAGE = (TM + TN) * 128;
  static constexpr int STAGES = (4 * STAGE <= 131072) ? 4 : 3;          // 128 KiB / 144 KiB of the CU's 160
  static constexpr int LDS = STAGES * STAGE;
  static constexpr int PA = TM / 64, PB = TN / 64, PW = PA + PB;        // 8-row pieces a wave stages per K tile
  static_assert((TM == 128 || TM == 256) && (TN == 128 || TN == 256) && TM + TN <= 384, "tile shapes: 128x128, 256x128, 128x256");
};

// Kb: bytes of a row's K extent (a multiple of 128); rows are dense.
template <class OP, int TM, int TN>
__global__ __launch_bounds__(512) void gemm_p128(const char* __restrict__ A, const char* __restrict__ B, uint16_t* __restrict__ C,
                                                 int M, int N, int Kb) {
  typedef klm_ptile<TM, TN> T;
  static_assert(OP::NI == T::NI, "operand policy and tile shape disagree");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid / T::WC, wc = wid % T::WC;
  int tm, tn;
  tile_of_block<8>(blockIdx.x, M / TM, N / TN, tm, tn);
  const int brow = tm * TM, bcol = tn * TN;
  const unsigned ld = (unsigned)Kb;
  // staging: wave `wid` brings rows [TM / 8 * wid, + TM / 8) of the A tile and [TN / 8 * wid, + TN / 8) of the B tile, 8-row pieces
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A + (size_t)brow * ld), 0, (int)((unsigned)TM * ld), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B + (size_t)bcol * ld), 0, (int)((unsigned)TN * ld), 0x00020000);
  const unsigned srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned voff = srow * ld + schunk * 16;
  const unsigned soff_a = (unsigned)(TM / 8 * wid) * ld, soff_b = (unsigned)(TN / 8 * wid) * ld;
  char* const lwa = smem + wid * (TM / 8) * 128;
  char* const lwb = smem + TM * 128 + wid * (TN / 8) * 128;
  auto issue = [&](int tile) {
    const int st = (tile % T::STAGES) * T::STAGE;
    const unsigned k = (unsigned)tile * 128u;
#pragma unroll
    for (int p = 0; p < T::PA; ++p) blds16(ra, voff, soff_a + k + (unsigned)(8 * p) * ld, lwa + st + p * 1024);
#pragma unroll
    for (int p = 0; p < T::PB; ++p) blds16(rb, voff, soff_b + k + (unsigned)(8 * p) * ld, lwb + st + p * 1024);
  };
  OP op;
  op.init(lane, wr, wc, TM * 128);
  const int nt = Kb >> 7;
  typename OP::frags f0, f1;
#pragma unroll
  for (int i = 0; i < T::STAGES - 1; ++i)
    if (i < nt) issue(i);
  // one K tile: `cur` receives tile t's fragments while the MFMAs of tile t-1 run on `prev`
  auto step = [&](int t, typename OP::frags& cur, const typename OP::frags& prev) {
    // this wave's pieces of tile t have landed (up to STAGES - 2 later tiles may be in flight)
    if constexpr (T::STAGES == 4) {
      if (t + 2 < nt) klm_vmcnt<2 * T::PW>();
      else if (t + 1 < nt) klm_vmcnt<T::PW>();
      else klm_vmcnt<0>();
    } else {
      if (t + 1 < nt) klm_vmcnt<T::PW>();
      else klm_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its reads of tile t-1 have completed
    KLM_BAR();                                            // true for every wave: tile t is whole, tile t-1's stage is free
    if (t + T::STAGES - 1 < nt) issue(t + T::STAGES - 1);
    op.read(cur, smem + (t % T::STAGES) * T::STAGE);
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
    for (int ni = 0; ni < T::NI; ++ni) {
      const int m = brow + wr * 64 + mi * 16 + fr;
      const int n = bcol + wc * (16 * T::NI) + ni * 16 + fq * 4;
      store_c4(C, (size_t)m * N + n, op.acc[mi][ni]);
    }
}

// Host side: shapes are checked here (whole tiles, whole K tiles, a row block within a buffer descriptor's 2 GiB).
template <class OP, int TM, int TN>
static int klm_launch_p128(const void* A, const void* B, void* C, int M, int N, int Kb, hipStream_t s) {
  typedef klm_ptile<TM, TN> T;
  if (M <= 0 || N <= 0 || Kb <= 0 || (M % TM) || (N % TN) || (Kb & 127) || (size_t)Kb * 256 > 0x7fffffffull) return KLM_ERR_SHAPE;
  if ((size_t)(M / TM) * (size_t)(N / TN) > 0x7fffffffull) return KLM_ERR_SHAPE;
  if (int e = klm_ensure_dyn_lds<gemm_p128<OP, TM, TN>>(T::LDS)) return e;
  hipLaunchKernelGGL((gemm_p128<OP, TM, TN>), dim3((M / TM) * (N / TN)), dim3(512), T::LDS, s, (const char*)A, (const char*)B, (uint16_t*)C, M, N, Kb);
  return KLM_OK;
}

// variant 14: 128 x 128, 16: 256 x 128 (A rows x B rows), 17: 128 x 256
template <template <int> class OPT>
static int klm_launch_p128_variant(int variant, const void* A, const void* B, void* C, int M, int N, int Kb, hipStream_t s) {
  switch (variant) {
    case 14: return klm_launch_p128<OPT<2>, 128, 128>(A, B, C, M, N, Kb, s);
    case 16: return klm_launch_p128<OPT<4>, 256, 128>(A, B, C, M, N, Kb, s);
    case 17: return klm_launch_p128<OPT<4>, 128, 256>(A, B, C, M, N, Kb, s);
  }
  return KLM_ERR_VARIANT;
}
