// C ABI of the MI355X (gfx950) validation-load kernels.
//
// These kernels have no counterpart in the reference (it has no native code, SURVEY.md
// section 2.3); they implement BASELINE.json north_star item 4: a known, steady GPU load
// that dashboard readings can be checked against.
//
// Every launcher is asynchronous on `stream`, performs no allocation and no host
// synchronisation (so it can be captured into a hipGraph), validates its shapes on the
// host and returns a negative klm_status instead of launching when they do not match
// what the kernel's grid assumes.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  KLM_OK = 0,
  KLM_ERR_SHAPE = -1,      // M/N/K not a multiple of the tile, or <= 0
  KLM_ERR_ALIGN = -2,      // pointer not 16-byte aligned
  KLM_ERR_VARIANT = -3,    // unknown kernel variant
  KLM_ERR_HIP = -4,        // the HIP runtime reported an error at launch
  KLM_ERR_ARG = -5,
} klm_status;

// bf16 GEMM, C[M][N] = A[M][K] * B[N][K]^T, fp32 accumulate, bf16 output.
//   variant 0: 128x128x64 tile, 4 waves, one LDS buffer, two barriers per K step
//   variant 1: 256x256x64 tile, 8 waves, LDS-DMA ring with counted vmcnt (default)
// Requirements: variant 0: M,N % 128 == 0, K % 64 == 0; variant 1: M,N % 256 == 0,
// K % 64 == 0 and K >= 128. All pointers 16-byte aligned.
int klm_gemm_bf16(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                  void* stream);

// fp8 (OCP e4m3fn) GEMM on the MX block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4)
// with unit scales, C[M][N] (bf16) = A[M][K] * B[N][K]^T.
// Requirements: M,N % 256 == 0, K % 128 == 0 and K >= 256.
int klm_gemm_fp8(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                 void* stream);

// fp4 (e2m1, two elements per byte) GEMM on the same MX MFMA at 4x the bf16 rate, unit scales.
// A is [M][K/2] bytes, B is [N][K/2] bytes, K counts ELEMENTS. Which nibble of a byte is the
// "first" element does not matter to the product as long as A and B are packed alike.
// Requirements: M,N % 256 == 0, K % 256 == 0 and K >= 512.
int klm_gemm_fp4(const void* A, const void* B, void* C, int M, int N, int K, int variant,
                 void* stream);

// HBM helpers. `bytes` must be a multiple of 16 and pointers 16-byte aligned.
int klm_hbm_fill(void* dst, size_t bytes, uint32_t pattern, void* stream);   // write-only
int klm_hbm_copy(void* dst, const void* src, size_t bytes, void* stream);    // read+write
int klm_hbm_read(const void* src, size_t bytes, void* sink8, void* stream);  // read-only
// Tuning variants. flags: bit 0 nontemporal loads, bit 1 nontemporal stores, bit 2 one
// contiguous chunk per block; grid = number of 256-thread blocks (0 = the default cap).
int klm_hbm_copy_ex(void* dst, const void* src, size_t bytes, int flags, int grid, void* stream);
int klm_hbm_fill_ex(void* dst, size_t bytes, uint32_t pattern, int flags, int grid, void* stream);
// Sums 32-bit words of [src, src+bytes) into *sum64 (device memory, must be zeroed by the
// caller on the same stream). Used to check that a filled region holds its pattern.
int klm_hbm_checksum(const void* src, size_t bytes, void* sum64, void* stream);

// A GPU busy-wait of `cycles` s_memtime ticks on one wave per CU: a load with ~0 HBM
// traffic and ~0 MFMA work, used to tell "GFX activity" apart from useful work.
int klm_spin(uint64_t cycles, int blocks, void* stream);

const char* klm_version(void);

#ifdef __cplusplus
}
#endif
