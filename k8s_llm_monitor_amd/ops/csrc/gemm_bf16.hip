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
//  * gemm_bf16_256_t: 256x256x64 tile, 8 waves (2x4), on the LDS-DMA ring pipeline of
//    gemm_ring256.h (shared with the fp8 / fp4 kernels of gemm_mx.hip): 128 KiB of LDS as 2 stages
//    x 4 half-tile slots, counted `s_waitcnt vmcnt(N)` keeps three half-tiles in flight across raw
//    `s_barrier`s, the two wave groups run one barrier apart. This file supplies the bf16 operand
//    policy (fragment reads + `v_mfma_f32_16x16x32_bf16` clusters) and the epilogues.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_p128.h"
#include "gemm_quad128.h"
#include "gemm_quad_asm.h"
#include "gemm_ring256.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

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


// The 128 x 128 eight-wave ring kernel for shapes of few tiles (variant 14) is gemm_p128.h, shared with the MX formats.
template <int NI_>
struct klm_p128_bf16 {
  static constexpr int NI = NI_;
  struct frags {
    bf16x8 a[2][4], b[2][NI_];   // [k step][tile]
  };
  f32x4 acc[4][NI_];
  int a_off[2], b_off[2];
  // fragment reads: row = base + tile * 16 + (lane & 15); 16-byte chunk = ks * 4 + (lane >> 4), swizzled by row & 7
  __device__ __forceinline__ void init(int lane, int wr, int wc, int b_base) {
    const int fr = lane & 15, fq = lane >> 4, key = lane & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((ks * 4 + fq) ^ key) << 4;
      a_off[ks] = (wr * 64 + fr) * 128 + ch;
      b_off[ks] = b_base + (wc * 16 * NI_ + fr) * 128 + ch;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NI_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void read(frags& f, const char* st) const {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ni = 0; ni < NI_; ++ni) f.b[ks][ni] = *reinterpret_cast<const bf16x8*>(st + b_off[ks] + ni * 2048);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) f.a[ks][mi] = *reinterpret_cast<const bf16x8*>(st + a_off[ks] + mi * 2048);
    }
  }
  __device__ __forceinline__ void mfma(const frags& f) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI_; ++ni)
          // X = B rows (n), Y = A rows (m): D[n][m]; a lane holds 4 consecutive n of one m
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[ks][ni], f.a[ks][mi], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
};

// ------------------------------------------------------------------------------------
// 256x256x64, 8 waves: bf16 operand policy for the ring pipeline (gemm_ring256.h).
// ------------------------------------------------------------------------------------
struct klm_op_bf16 {
  f32x4 acc[2][2][4][2];   // [a half][b half][m tile][n tile]
  bf16x8 a[2][4];          // [k step][m tile]
  bf16x8 b[2][2][2];       // [register set][k step][n tile]
  int a_off0, a_off1, b_off0, b_off1;

  // fragment reads: row = base + t*16 + (lane&15); 16-byte chunk = ks*4 + (lane>>4), swizzled by row & 7
  __device__ __forceinline__ void init(int lane, int wr, int wc) {
    const int fr = lane & 15, fq = lane >> 4;
    const int sw0 = ((fq ^ (lane & 7)) << 4);
    a_off0 = (wr * 64 + fr) * 128 + sw0;
    a_off1 = a_off0 ^ 64;
    b_off0 = (wc * 32 + fr) * 128 + sw0;
    b_off1 = b_off0 ^ 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[i][j][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void read_a(const char* slot) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      a[0][mi] = *reinterpret_cast<const bf16x8*>(slot + a_off0 + mi * 2048);
      a[1][mi] = *reinterpret_cast<const bf16x8*>(slot + a_off1 + mi * 2048);
    }
  }
  template <int SET>
  __device__ __forceinline__ void read_b(const char* slot) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      b[SET][0][ni] = *reinterpret_cast<const bf16x8*>(slot + b_off0 + ni * 2048);
      b[SET][1][ni] = *reinterpret_cast<const bf16x8*>(slot + b_off1 + ni * 2048);
    }
  }
  template <int AH, int BH, int SET>
  __device__ __forceinline__ void mfma() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          // X = B rows (n), Y = A rows (m): D[n][m]; lane holds 4 consecutive n of one m
          acc[AH][BH][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[SET][ks][ni], a[ks][mi], acc[AH][BH][mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
};

// SPLIT staging (decode at batch 129 - 256): A may have any number of rows <= 256 * tiles (rows past M re-read row
// M-1 and are not stored), so each lane keeps its four clamped A rows; both operands carry a row stride and a K offset.
struct klm_stage_clamped {
  const char* pas[2][2];
  const char* pb;
  size_t ldb;
  char* lw;
  template <int H>
  __device__ __forceinline__ void issue_a(int stage, int tile) const {
    char* l = lw + stage * KLM_STAGE_BYTES + (H ? KLM_SLOT_A1 : KLM_SLOT_A0);
    glds16(pas[H][0] + (size_t)tile * 128, l);
    glds16(pas[H][1] + (size_t)tile * 128, l + 1024);
  }
  template <int H>
  __device__ __forceinline__ void issue_b(int stage, int tile) const {
    const char* g = pb + (size_t)H * 32 * ldb + (size_t)tile * 128;
    char* l = lw + stage * KLM_STAGE_BYTES + (H ? KLM_SLOT_B1 : KLM_SLOT_B0);
    glds16(g, l);
    glds16(g + 8 * ldb, l + 1024);
  }
};

// SPLIT (ops/llm.py `_mid_partials`): the grid's y index picks a K range of `Kc`, the result goes out as fp32 partial
// sums P[split][M][N] (what the mid-batch epilogues and the RoPE kernel consume). SWIGLU: one K range, SwiGLU in the
// tile epilogue. SCHED: the pipeline schedule (gemm_ring256.h); the serving instantiations stay on schedule 0.
template <bool SPLIT, bool SWIGLU = false, int SCHED = 0>
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

  // ---- staging addresses. Wave `wid` fills slot rows [wid*16, wid*16+16): 2 pieces of 8 rows.
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const int arow = brow + (wid >> 2) * 128 + (wid & 3) * 16 + srow;
  const int bro = bcol + (wid >> 1) * 64 + (wid & 1) * 16 + srow;
  char* const lw = smem + wid * 2048;  // + stage*65536 + slot + piece*1024

  klm_op_bf16 op;
  op.init(lane, wr, wc);
  const int nt = (SPLIT ? Kc : K) >> 6;
  if constexpr (SPLIT) {
    klm_stage_clamped st;
    const int kbeg = blockIdx.y * Kc;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int r = arow + h * 64 + j * 8;
        r = r < M ? r : M - 1;
        st.pas[h][j] = reinterpret_cast<const char*>(A + (size_t)r * lda + schunk * 8 + kbeg);
      }
    st.pb = reinterpret_cast<const char*>(B + (size_t)bro * ldb + schunk * 8 + kbeg);
    st.ldb = (size_t)ldb * 2;
    st.lw = lw;
    klm_ring256_mainloop<SCHED>(op, st, smem, nt, wr);
  } else {
    klm_stage_dense st;
    st.init(A, B, brow, bcol, (unsigned)K * 2, (unsigned)K * 2, lane, wid, smem);
    klm_ring256_mainloop<SCHED>(op, st, smem, nt, wr);
  }

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
          if (SWIGLU) {
            // gate / up rows of B are interleaved by 8: in a 16-column tile lanes fq 0,1 hold gate, fq 2,3 the matching up
            // columns (lane ^ 32). One K range only: the activation goes out in bf16, no partial sums, no epilogue launch.
            f32x4 v = op.acc[ah][bh][mi][ni], o;
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
            if (m < M) *reinterpret_cast<f32x4*>(P + (size_t)m * N + n) = op.acc[ah][bh][mi][ni];
          } else {
            store_c4(C, (size_t)m * N + n, op.acc[ah][bh][mi][ni]);
          }
        }
}

// ------------------------------------------------------------------------------------
// 256x256x64 on 4 waves of 128x128 (gemm_quad128.h): bf16 operand policy. A part is a k step of 32.
// ------------------------------------------------------------------------------------
struct klm_op_quad_bf16 {
  f32x4 acc[8][8];                 // [m tile][n tile]: 256 accumulator registers
  bf16x8 fa[2][8], fb[2][8];       // [fragment set][tile]
  int a_off[2], b_off[2];          // per k step

  __device__ __forceinline__ void init(int lane, int wr, int wc) {
    const int fr = lane & 15, fq = lane >> 4, key = lane & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((ks * 4 + fq) ^ key) << 4;
      a_off[ks] = (wr * 128 + fr) * 128 + ch;
      b_off[ks] = KLM_QUAD_B_OFFSET + (wc * 128 + fr) * 128 + ch;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // fragment read R of 16: A0, B0..B7, A1..A7 (all sixteen have landed before the k-step that uses them starts, whatever its order)
  template <int SET, int PART, int R>
  __device__ __forceinline__ void read1(const char* stage) {
    if constexpr (R == 0) fa[SET][0] = *reinterpret_cast<const bf16x8*>(stage + a_off[PART]);
    else if constexpr (R <= 8) fb[SET][R - 1] = *reinterpret_cast<const bf16x8*>(stage + b_off[PART] + (R - 1) * 2048);
    else fa[SET][R - 8] = *reinterpret_cast<const bf16x8*>(stage + a_off[PART] + (R - 8) * 2048);
  }
  template <int SET, int PART>
  __device__ __forceinline__ void read_all(const char* stage) {
    read1<SET, PART, 0>(stage); read1<SET, PART, 1>(stage); read1<SET, PART, 2>(stage); read1<SET, PART, 3>(stage);
    read1<SET, PART, 4>(stage); read1<SET, PART, 5>(stage); read1<SET, PART, 6>(stage); read1<SET, PART, 7>(stage);
    read1<SET, PART, 8>(stage); read1<SET, PART, 9>(stage); read1<SET, PART, 10>(stage); read1<SET, PART, 11>(stage);
    read1<SET, PART, 12>(stage); read1<SET, PART, 13>(stage); read1<SET, PART, 14>(stage); read1<SET, PART, 15>(stage);
  }
  // MFMAs 4G .. 4G+3 of the 64 (m tile major): X = B rows (n), Y = A rows (m): D[n][m], a lane holds 4 consecutive n of one m
  template <int SET, int PART, int G>
  __device__ __forceinline__ void mfma4() {
    constexpr int mi = G >> 1, n0 = (G & 1) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mi][n0 + j]) : "v"(fb[SET][n0 + j]), "v"(fa[SET][mi]));
  }
};

__global__ __launch_bounds__(256) void gemm_bf16_quad(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                      uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  int tm, tn;
  tile_of_block<8>(blockIdx.x, M >> 8, N >> 8, tm, tn);
  const int brow = tm << 8, bcol = tn << 8;
  klm_stage_quad st;
  st.init(A, B, brow, bcol, (unsigned)K * 2, (unsigned)K * 2, lane, wid, smem);
  klm_op_quad_bf16 op;
  op.init(lane, wr, wc);
  klm_quad_mainloop(op, st, smem, K >> 6);
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      const int m = brow + wr * 128 + mi * 16 + fr;
      const int n = bcol + wc * 128 + ni * 16 + fq * 4;
      store_c4(C, (size_t)m * N + n, op.acc[mi][ni]);
    }
}

// tests only: workgroup count of the persistent kernels (0 = one per CU)
extern "C" int klm_gemm_persistent_grid(int grid) {
  const int before = klm_quad_p_grid_override();
  if (grid >= 0) klm_quad_p_grid_override() = grid;
  return before;
}

// 1 when the A/B-record and diagnostic streams are compiled in (-DKLM_GEMM_DIAG: never the shipped library)
extern "C" int klm_gemm_has_diag(void) {
#if defined(KLM_GEMM_DIAG)
  return 1;
#else
  return 0;
#endif
}

// The plan the persistent streams run an M x N problem with rows of `Kbytes` under: out = {grid, rounds, rest, split, rot}.
// cus: workgroups available (0: the current device's CU count, or the tests' override). Pure arithmetic apart from that.
#define KLM_SUPERBLOCK_MIN_BYTES ((size_t)384 << 20)

extern "C" int klm_gemm_stream_plan(int M, int N, int Kbytes, int cus, int mode, int rot, size_t ws_bytes, int* out) {
  if (!out) return KLM_ERR_ARG;
  if (!klm_quad_asm_shape_ok(M, N, Kbytes)) return KLM_ERR_SHAPE;
  if (cus <= 0) cus = klm_quad_p_cus();
  if (cus <= 0) return KLM_ERR_HIP;
  if (mode == 1 && ((size_t)M + (size_t)N) * (size_t)Kbytes >= KLM_SUPERBLOCK_MIN_BYTES) mode |= 4;      // as klm_gemm_stream does
  const KlmQuadPlan p = klm_quad_plan(M, N, Kbytes, cus, mode, rot, ws_bytes);
  out[0] = p.grid, out[1] = p.rounds, out[2] = p.rest, out[3] = p.split, out[4] = p.rot, out[5] = p.map;
  return KLM_OK;
}

// The persistent hand-scheduled stream, every option: dtype 0 bf16 / 1 fp8 (e4m3) / 2 fp4 (e2m1 pairs); rows of A and B hold
// `Kbytes` bytes of K and are `ld_bytes` apart (0: dense); store 1: C written with `sc1 nt`; mode 0: whole tiles only, 1: the
// last partial round is cut into K slices where that pays (needs `ws`: rest * split * 256 KiB of scratch on the device, at most
// CUs * 256 KiB), 2: wherever the shape allows (tests), + 4: the super-block tile order where the shape allows it (KlmQuadPlan.map); rot > 0: every workgroup starts its K loop `rot` passes after its
// neighbour's (dense operands on a row pitch the memory channels camp on).
extern "C" int klm_gemm_stream(int dtype, const void* A, const void* B, void* C, int M, int N, int Kbytes, int ld_bytes, int store,
                               int mode, int rot, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)ws) & 15) return KLM_ERR_ALIGN;
  if (dtype < 0 || dtype > 2 || store < 0 || store > 1 || mode < 0 || mode > 6 || (mode & 3) == 3 || rot < 0) return KLM_ERR_VARIANT;
  if (!klm_quad_asm_shape_ok(M, N, Kbytes, ld_bytes)) return KLM_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (!klm_stream_on_current_device(s)) return KLM_ERR_ARG;
  // the automatic plan (mode 1) also picks the tile order: super-blocks once the operands are well past the 256 MB Infinity
  // Cache (12288^3 bf16: +5 - 10 %, 16384^2 x 8192: +5 %; below that the chunked order is 0.5 - 3 % faster:
  // profiles/r3_prof/gemm_tile_order.json)
  if (mode == 1 && ((size_t)M + (size_t)N) * (size_t)(ld_bytes ? ld_bytes : Kbytes) >= KLM_SUPERBLOCK_MIN_BYTES) mode |= 4;
  int e;
  switch (dtype * 2 + store) {
    case 0: e = klm_launch_quad_asm_p<0>(A, B, C, M, N, Kbytes, s, ld_bytes, mode, rot, ws, ws_bytes); break;
    case 1: e = klm_launch_quad_asm_p<1>(A, B, C, M, N, Kbytes, s, ld_bytes, mode, rot, ws, ws_bytes); break;
    case 2: e = klm_launch_quad_asm_p<30>(A, B, C, M, N, Kbytes, s, ld_bytes, mode, rot, ws, ws_bytes); break;
    case 3: e = klm_launch_quad_asm_p<31>(A, B, C, M, N, Kbytes, s, ld_bytes, mode, rot, ws, ws_bytes); break;
    case 4: e = klm_launch_quad_asm_p<20>(A, B, C, M, N, Kbytes, s, ld_bytes, mode, rot, ws, ws_bytes); break;
    default: e = klm_launch_quad_asm_p<21>(A, B, C, M, N, Kbytes, s, ld_bytes, mode, rot, ws, ws_bytes); break;
  }
  if (e) return e;
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}

// ------------------------------------------------------------------------------------
// Which kernel for a shape: the cost model of ops/__init__.py (`pick_tile_kernel`: the Python side is the reference, fitted to
// profiles/r3_prof/gemm_small.json, gemm_mid.json, gemm_units_first.json; tests/test_ops_layout.py holds the two against each
// other over a grid of shapes), so that native callers -- klm_burner -- dispatch like `variant=None` does.
// Times in microseconds per BYTE of a row's K extent: {quarter-full chip, full chip}.
// ------------------------------------------------------------------------------------
namespace klm_pick {
constexpr double kStream[2] = {0.0073, 0.0106}, kP128[2] = {0.0020, 0.0039}, kP256[2] = {0.0056, 0.0070};
constexpr double kP128SecondRound = 0.0032, kStreamFixed = 8.0, kRingFixed = 6.5, kSplitTimeMB = 0.13;

static inline double per_kb(const double* pair, double occ, int power) {
  double x = (occ - 0.25) / 0.75;
  x = x < 0 ? 0 : (x > 1 ? 1 : x);
  return pair[0] + (pair[1] - pair[0]) * (power == 2 ? x * x : x);
}
static inline bool stream_ok(int M, int N, int Kb) { return M > 0 && N > 0 && !(M & 255) && !(N & 255) && !(Kb & 255) && Kb >= 512; }
static inline bool p128_ok(int M, int N, int Kb) { return M > 0 && N > 0 && !(M & 127) && !(N & 127) && !(Kb & 127) && Kb >= 128; }
static inline bool p256_ok(int M, int N, int Kb) { return p128_ok(M, N, Kb) && (!(M & 255) || !(N & 255)); }

static double stream_us(int M, int N, int Kb, int cus) {
  const KlmQuadPlan p = klm_quad_plan(M, N, Kb, cus, 1, 0, (size_t)cus * KLM_QUAD_SLOT_BYTES);
  double t = kStreamFixed + (double)p.rounds * Kb * kStream[1];
  if (p.rest) {
    if (p.split > 1)
      t += Kb * per_kb(kStream, (double)p.rest * p.split / cus, 2) / p.split + KLM_SPLIT_LAUNCH_US + kSplitTimeMB * p.rest * p.split * 0.25;
    else
      t += Kb * per_kb(kStream, (double)p.rest / cus, 2);
  }
  return t;
}
static double p128_us(int M, int N, int Kb, int cus) {
  const long tiles = (long)(M >> 7) * (N >> 7);
  const long rounds = tiles / cus, rest = tiles % cus;
  double per = per_kb(kP128, (double)rest / cus, 1);
  if (rounds && per < kP128SecondRound) per = kP128SecondRound;
  return kRingFixed + (double)rounds * Kb * kP128[1] + (rest ? Kb * per : 0.0);
}
static double p256_us(int M, int N, int Kb, int cus) {
  const long tiles = ((long)M * N) / (256 * 128);
  const long rounds = tiles / cus, rest = tiles % cus;
  return kRingFixed + (double)rounds * Kb * kP256[1] + (rest ? Kb * per_kb(kP256, (double)rest / cus, 1) : 0.0);
}
// 0: none of the three takes the shape; 1: the persistent 256 x 256 stream; 2: the ring kernel on 128 x 128 tiles; 3: on 256 x 128
static int kind(int M, int N, int Kb, int cus) {
  int best = 0;
  double t = 0;
  if (stream_ok(M, N, Kb)) best = 1, t = stream_us(M, N, Kb, cus);
  if (p256_ok(M, N, Kb)) {
    const double x = p256_us(M, N, Kb, cus);
    if (!best || x < t) best = 3, t = x;
  }
  if (p128_ok(M, N, Kb)) {
    const double x = p128_us(M, N, Kb, cus);
    if (!best || x < t) best = 2, t = x;
  }
  return best;
}
}  // namespace klm_pick

extern "C" int klm_gemm_pick(int M, int N, int Kbytes, int cus) {
  if (cus <= 0) cus = klm_quad_p_cus();
  if (cus <= 0) return KLM_ERR_HIP;
  return klm_pick::kind(M, N, Kbytes, cus);
}

extern "C" int klm_gemm_fp8(const void* A, const void* B, void* C, int M, int N, int K, int variant, void* stream);
extern "C" int klm_gemm_fp4(const void* A, const void* B, void* C, int M, int N, int K, int variant, void* stream);
extern "C" int klm_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int variant, void* stream);

// `variant=None` for native callers: dense operands, dtype 0 bf16 / 1 fp8 / 2 fp4 (rows of `Kbytes` bytes), `ws` the scratch of
// klm_gemm_stream (may be null: then no round is split). Returns the status; *picked (may be null) = the klm_gemm_<dtype> variant
// that ran (bf16 10 / fp8 7 / fp4 6 stand for the plan-driven stream).
extern "C" int klm_gemm_auto(int dtype, const void* A, const void* B, void* C, int M, int N, int Kbytes, void* ws, size_t ws_bytes,
                             void* stream, int* picked) {
  if (dtype < 0 || dtype > 2) return KLM_ERR_VARIANT;
  const int cus = klm_quad_p_cus();
  if (cus <= 0) return KLM_ERR_HIP;
  const int k = klm_pick::kind(M, N, Kbytes, cus);
  const int elems = dtype == 0 ? Kbytes / 2 : dtype == 1 ? Kbytes : Kbytes * 2;
  auto ring = [&](int v) {
    if (picked) *picked = v;
    return dtype == 0 ? klm_gemm_bf16(A, B, C, M, N, elems, v, stream) : dtype == 1 ? klm_gemm_fp8(A, B, C, M, N, elems, v, stream)
                                                                                    : klm_gemm_fp4(A, B, C, M, N, elems, v, stream);
  };
  // whole 256 x 256 tiles with an odd number of K tiles, filling half the chip: the 256 x 256 ring kernel's schedule 0
  const bool ring256 = !klm_pick::stream_ok(M, N, Kbytes) && !(M & 255) && !(N & 255) && Kbytes >= 256 && (M >> 8) * (N >> 8) >= cus / 2;
  if (k == 1) {
    if (picked) *picked = dtype == 0 ? 10 : dtype == 1 ? 7 : 6;
    return klm_gemm_stream(dtype, A, B, C, M, N, Kbytes, 0, 0, 1, 0, ws, ws ? ws_bytes : 0, stream);
  }
  if ((k == 2 || k == 3) && ring256) return ring(1);
  if (k == 2) return ring(14);
  if (k == 3) return ring(dtype == 0 ? (!(M & 255) ? 16 : 17) : (!(N & 255) ? 17 : 16));
  return ring(dtype == 0 ? 0 : 1);
}

// The hand-scheduled stream on operands whose rows are `ld_bytes` apart (>= 2 K, a multiple of 16; A and B alike).
extern "C" int klm_gemm_bf16_pitched(const void* A, const void* B, void* C, int M, int N, int K, int ld_bytes, int persistent,
                                     void* stream) {
  if (!A || !B || !C) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
  if (K <= 0 || K > (1 << 29) || !klm_quad_asm_shape_ok(M, N, K * 2, ld_bytes)) return KLM_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (!klm_stream_on_current_device(s)) return KLM_ERR_ARG;
  // persistent: 0 one workgroup per tile, 1 persistent, 2 persistent with C streamed past the caches (sc1 nt)
  if (int e = persistent == 2   ? klm_launch_quad_asm_p<1>(A, B, C, M, N, K * 2, s, ld_bytes)
              : persistent == 1 ? klm_launch_quad_asm_p<0>(A, B, C, M, N, K * 2, s, ld_bytes)
                                : klm_launch_quad_asm<0>(A, B, C, M, N, K * 2, s, ld_bytes))
    return e;
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}

extern "C" int klm_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                             void* stream) {
  if (!A || !B || !C) return KLM_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return KLM_ERR_ALIGN;
  if (M <= 0 || N <= 0 || K <= 0) return KLM_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (!klm_stream_on_current_device(s)) return KLM_ERR_ARG;
  if (variant == 0) {
    if ((M & 127) || (N & 127) || (K & 63)) return KLM_ERR_SHAPE;
    const int grid = (M >> 7) * (N >> 7);
    hipLaunchKernelGGL(gemm_bf16_128, dim3(grid), dim3(256), 32768, s, (const uint16_t*)A,
                       (const uint16_t*)B, (uint16_t*)C, M, N, K);
  } else if (variant == 14 || variant == 16 || variant == 17) {
    // 128 x 128 / 256 x 128 / 128 x 256 tiles on eight waves, LDS-DMA ring: shapes of too few 256 x 256 tiles to fill the chip
    if ((K & 63) || K > 0x3fffffff) return KLM_ERR_SHAPE;
    if (int e = klm_launch_p128_variant<klm_p128_bf16>(variant, A, B, C, M, N, K * 2, s)) return e;
  } else if (variant == 3) {
    // 4 waves of 128x128; an even number of K tiles >= 4
    if ((M & 255) || (N & 255) || (K & 127) || K < 256) return KLM_ERR_SHAPE;
    if (int e = klm_ensure_dyn_lds<gemm_bf16_quad>(KLM_QUAD_LDS_BYTES)) return e;
    hipLaunchKernelGGL(gemm_bf16_quad, dim3((M >> 8) * (N >> 8)), dim3(256), KLM_QUAD_LDS_BYTES, s, (const uint16_t*)A,
                       (const uint16_t*)B, (uint16_t*)C, M, N, K);
  } else if (variant == 9) {
    // persistent, C stored with `sc1 nt` (streamed past the caches)
    if (!klm_quad_asm_shape_ok(M, N, K * 2)) return KLM_ERR_SHAPE;
    if (int e = klm_launch_quad_asm_p<1>(A, B, C, M, N, K * 2, s)) return e;
  } else if (variant == 8) {
    // the persistent form of stream 4: one workgroup per CU, the next output tile's operands fetched under the epilogue
    if (!klm_quad_asm_shape_ok(M, N, K * 2)) return KLM_ERR_SHAPE;
    if (int e = klm_launch_quad_asm_p<0>(A, B, C, M, N, K * 2, s)) return e;
  } else if (variant == 4) {
    // the hand-scheduled stream (gemm_quad_asm.h), one workgroup per tile
    if (!klm_quad_asm_shape_ok(M, N, K * 2)) return KLM_ERR_SHAPE;
    if (int e = klm_launch_quad_asm<0>(A, B, C, M, N, K * 2, s)) return e;
#if defined(KLM_GEMM_DIAG)
  } else if (variant == 48 || variant == 49) {
    // diagnostic: stream 8 without the stores / without the epilogue -- C is NOT written
    if (!klm_quad_asm_shape_ok(M, N, K * 2)) return KLM_ERR_SHAPE;
    if (int e = variant == 48 ? klm_launch_quad_asm_p<10>(A, B, C, M, N, K * 2, s) : klm_launch_quad_asm_p<11>(A, B, C, M, N, K * 2, s)) return e;
  } else if ((variant >= 5 && variant <= 7) || (variant >= 40 && variant <= 43)) {
    // 5..7 earlier placements of stream 4 (the A/B record); 40..43 diagnostic, results meaningless
    if (!klm_quad_asm_shape_ok(M, N, K * 2)) return KLM_ERR_SHAPE;
    int e = KLM_OK;
    switch (variant) {
      case 5: e = klm_launch_quad_asm<1>(A, B, C, M, N, K * 2, s); break;
      case 6: e = klm_launch_quad_asm<2>(A, B, C, M, N, K * 2, s); break;
      case 7: e = klm_launch_quad_asm<3>(A, B, C, M, N, K * 2, s); break;
      case 40: e = klm_launch_quad_asm<10>(A, B, C, M, N, K * 2, s); break;
      case 41: e = klm_launch_quad_asm<11>(A, B, C, M, N, K * 2, s); break;
      case 42: e = klm_launch_quad_asm<12>(A, B, C, M, N, K * 2, s); break;
      default: e = klm_launch_quad_asm<13>(A, B, C, M, N, K * 2, s); break;
    }
    if (e) return e;
#endif
  } else if (variant == 1 || variant == 2) {
    if ((M & 255) || (N & 255) || (K & 63) || K < 128) return KLM_ERR_SHAPE;
    const int grid = (M >> 8) * (N >> 8);
    // variant 2 = the one-half-tile-per-phase schedule; it needs an even number of K tiles >= 4
    if (variant == 2 && (K & 127) == 0 && K >= 256) {
      if (int e = klm_ensure_dyn_lds<gemm_bf16_256_t<false, false, 1>>(KLM_RING_LDS_BYTES)) return e;
      hipLaunchKernelGGL((gemm_bf16_256_t<false, false, 1>), dim3(grid), dim3(512), KLM_RING_LDS_BYTES, s, (const uint16_t*)A,
                         (const uint16_t*)B, (void*)C, M, N, K, K, K, K);
    } else {
      if (int e = klm_ensure_dyn_lds<gemm_bf16_256_t<false, false, 0>>(KLM_RING_LDS_BYTES)) return e;
      hipLaunchKernelGGL((gemm_bf16_256_t<false, false, 0>), dim3(grid), dim3(512), KLM_RING_LDS_BYTES, s, (const uint16_t*)A,
                         (const uint16_t*)B, (void*)C, M, N, K, K, K, K);
    }
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
  if (!klm_stream_on_current_device((hipStream_t)stream)) return KLM_ERR_ARG;
  if (int e = klm_ensure_dyn_lds<gemm_bf16_256_t<true>>(KLM_RING_LDS_BYTES)) return e;
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
  if (!klm_stream_on_current_device((hipStream_t)stream)) return KLM_ERR_ARG;
  if (int e = klm_ensure_dyn_lds<gemm_bf16_256_t<true, true>>(KLM_RING_LDS_BYTES)) return e;
  const int tiles = ((M + 255) >> 8) * (N >> 8);
  hipLaunchKernelGGL((gemm_bf16_256_t<true, true>), dim3(tiles, 1), dim3(512), 131072, (hipStream_t)stream, (const uint16_t*)A,
                     (const uint16_t*)B, act, M, N, K, lda, ldb, K);
  return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP;
}
