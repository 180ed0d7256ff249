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

// The hand-scheduled streams on ROW-PITCHED operands: rows of A and B are `ld_bytes` apart (>= the K extent in bytes, a multiple
// of 16, the same for both). Shapes as for the streams (M, N % 256 == 0; K extent a multiple of 256 bytes, >= 512).
// A burner that owns its operands pads the pitch away from strides the memory channels camp on (docs/KERNELS.md).
int klm_gemm_bf16_pitched(const void* A, const void* B, void* C, int M, int N, int K, int ld_bytes, int persistent, void* stream);
// kind 1: fp8 (Kbytes = K), 2: fp4 (Kbytes = K / 2)
int klm_gemm_mx_pitched(int kind, const void* A, const void* B, void* C, int M, int N, int Kbytes, int ld_bytes, int persistent,
                        void* stream);

// Persistent GEMM variants (bf16 variant 8, fp8 variant 4, fp4 variant 3): one workgroup per CU walks its output tiles.
// Tests: set a fixed workgroup count (0 = one per CU; < 0 = only read); returns the previous setting.
int klm_gemm_persistent_grid(int grid);
int klm_gemm_has_diag(void);    // 1: built with -DKLM_GEMM_DIAG (A/B-record and diagnostic streams compiled in)

// The persistent stream with every option (gemm_bf16.hip): dtype 0 bf16 / 1 fp8 / 2 fp4; mode 1 cuts the last partial round of
// tiles into K slices (stream-K style; fp32 partial sums in `ws`, added up by a second kernel) where that pays, 2 wherever it
// can; rot > 0 staggers the workgroups' K loops (dense operands on a camping row pitch). klm_gemm_stream_plan: the plan alone.
int klm_gemm_stream(int dtype, const void* A, const void* B, void* C, int M, int N, int Kbytes, int ld_bytes, int store, int mode,
                    int rot, void* ws, size_t ws_bytes, void* stream);
int klm_gemm_stream_plan(int M, int N, int Kbytes, int cus, int mode, int rot, size_t ws_bytes, int* out6);
// The default dispatch for native callers (the cost model of ops/__init__.py, mirrored): klm_gemm_pick -> 0 none / 1 the persistent
// 256 x 256 stream / 2 the ring kernel on 128 x 128 tiles / 3 on 256 x 128 tiles (cus 0: this device's); klm_gemm_auto runs it
// (dtype 0 bf16 / 1 fp8 / 2 fp4, dense rows of Kbytes bytes; ws: the scratch of klm_gemm_stream or null).
int klm_gemm_pick(int M, int N, int Kbytes, int cus);
int klm_gemm_auto(int dtype, const void* A, const void* B, void* C, int M, int N, int Kbytes, void* ws, size_t ws_bytes, void* stream,
                  int* picked);

// HBM helpers. `bytes` must be a multiple of 16 and pointers 16-byte aligned.
int klm_hbm_fill(void* dst, size_t bytes, uint32_t pattern, void* stream);   // write-only
int klm_hbm_copy(void* dst, const void* src, size_t bytes, void* stream);    // read+write
int klm_hbm_read(const void* src, size_t bytes, void* sink8, void* stream);  // read-only
// Tuning variants. flags: bit 0 nontemporal loads, bit 1 nontemporal stores, bit 2 one
// contiguous chunk per block; grid = number of 256-thread blocks (0 = the default cap).
int klm_hbm_copy_ex(void* dst, const void* src, size_t bytes, int flags, int grid, void* stream);
int klm_hbm_fill_ex(void* dst, size_t bytes, uint32_t pattern, int flags, int grid, void* stream);
// flags bits 3-4 (copy, fill, read): one-shot tiles instead of a loop, 8: 4 accesses of 16 bytes per thread, 16: 1, 24: 2.
int klm_hbm_read_ex(const void* src, size_t bytes, void* sink8, int flags, int grid, void* stream);
// Sums 32-bit words of [src, src+bytes) into *sum64 (device memory, must be zeroed by the
// caller on the same stream). Used to check that a filled region holds its pattern.
int klm_hbm_checksum(const void* src, size_t bytes, void* sum64, void* stream);
// `rows` rows of `row_bytes` from row pitch `src_pitch` to row pitch `dst_pitch` (bytes; everything a multiple of 16)
int klm_hbm_repitch(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, int rows, size_t row_bytes, void* stream);

// A GPU busy-wait of `cycles` s_memtime ticks on one wave per CU: a load with ~0 HBM
// traffic and ~0 MFMA work, used to tell "GFX activity" apart from useful work.
int klm_spin(uint64_t cycles, int blocks, void* stream);

// ---- LLM serving kernels (llm_*.hip): bf16, head_dim 128, KV blocks of 16 tokens --------------
// K cache [num_blocks][n_kv][16][128], V cache [num_blocks][n_kv][128][16] (transposed), both
// bf16 and zero-initialised by the caller (stale slots are read and multiplied by zero).
// kv_fp8 != 0 in the three calls below: the caches hold OCP e4m3 bytes (unit scale) in the same
// layouts; q, the in-place k and every output stay bf16. Prefill then always runs variant 0.
//
// out = rmsnorm(x) * weight; with `residual` != NULL: residual <- x + residual first, and the
// norm is taken of the new residual. H % 8 == 0, H <= 8192.
int klm_rmsnorm(const void* x, void* residual, const void* weight, void* out, int T, int H, float eps,
                void* stream);
// out[T][8j+c] = silu(gate_up[T][16j+c]) * gate_up[T][16j+8+c] (gate/up interleaved by 8); F % 8 == 0.
int klm_silu_mul(const void* gate_up, void* out, int T, int F, void* stream);
// RoPE (rotate-half) on the q and k heads of qkv[T][n_q + 2 n_kv][128] in place, with
// cos_sin[max_pos][128] fp32 (64 cos | 64 sin); k and v of token t go to cache slot slots[t]
// (= block * 16 + offset; a negative slot skips the write).
int klm_rope_cache(void* qkv, const float* cos_sin, const int32_t* positions, const int32_t* slots,
                   void* k_cache, void* v_cache, int T, int n_q, int n_kv, int max_pos, int64_t num_slots,
                   int kv_fp8, void* stream);
// K-split variant of the 256 x 256 bf16 GEMM for decode at batch 129 - 256: fp32 partial sums part[ksplit][M][N]
// (N % 256 == 0, K % (ksplit * 64) == 0; ksplit 0 = chosen by klm_gemm_bf16_splitk_plan).
int klm_gemm_bf16_splitk_plan(int M, int N, int K, int* ksplit);
int klm_gemm_bf16_splitk(const void* A, int lda, const void* B, int ldb, float* part, int M, int N, int K, int ksplit,
                         void* stream);
// The same GEMM without a K split and with SwiGLU in the tile epilogue: act[M][N/2] bf16 (B's gate / up rows interleaved by 8).
int klm_gemm_bf16_swiglu(const void* A, int lda, const void* B, int ldb, void* act, int M, int N, int K, void* stream);
// Decode attention: q (row b at q + b*q_stride elements, n_q heads of 128), block_tables[B][max_blocks],
// ctx_lens[B] -> out[B][n_q][128]. q_stride lets q be a view of the fused qkv activation.
// num_splits > 1 splits each context over that many blocks (flash decoding); then
// tmp_o[B][n_q][num_splits][128] and tmp_ml[B][n_q][num_splits][2] (fp32) are scratch.
int klm_paged_attn_decode(const void* q, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                          const int32_t* ctx_lens, void* out, float* tmp_o, float* tmp_ml, int B, int q_stride,
                          int n_q, int n_kv, int max_blocks, int num_blocks_total, int num_splits, float scale,
                          int kv_fp8, void* stream);
// Causal prefill over the paged cache (the keys of the queries are already in it).
// seq_info[n_seqs][3] = {q_start, q_len, ctx_len}; tiles[n_tiles][2] = {sequence, first query
// token of the tile}; a tile covers q_tiles * 64 / (n_q / n_kv) query tokens (q_tiles = 16-row
// query tiles per wave: 1, 2 or 4). variant 0: every wave loads K/V fragments from global memory
// itself (short prompts); 1: the block stages K/V once through a 4-stage LDS ring (long prompts).
// RoPE (q and the newest key) and the cache write of the newest key / value inside the decode attention kernel;
// qkv is the un-rotated activation [B][q_stride >= (n_q + 2 n_kv) * 128], the caches are bf16.
int klm_paged_attn_decode_rope(const void* qkv, void* k_cache, void* v_cache, const int32_t* block_tables,
                               const int32_t* ctx_lens, const int32_t* slots, const float* cos_sin, void* out, float* tmp_o,
                               float* tmp_ml, int B, int q_stride, int n_q, int n_kv, int max_blocks, int num_blocks_total,
                               int num_splits, float scale, int max_pos, void* stream);
int klm_paged_attn_prefill(const void* q, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                           const int32_t* seq_info, const int32_t* tiles, void* out, int n_tiles, int n_seqs,
                           int q_stride, int n_q, int n_kv, int max_blocks, int num_blocks_total, float scale,
                           int q_tiles, int variant, int kv_fp8, void* stream);

// Skinny linear layer for decode (M <= 16): out[M][N] = x[M][K] . W[N][K]^T, W streamed once.
// rms != 0: rows are scaled by rsqrt(mean(x^2) + eps) (RMSNorm whose gain is folded into W).
// epilogue 0: none; 1: out = residual + product (out may alias residual); 2: SwiGLU over gate/up
// rows interleaved by 8, out is [M][N/2]. N % 16 == 0, K % 128 == 0; x rows are x_stride elements
// apart. Tuning: waves = waves per block that split K (1, 2, 4, 8, 16; 0 = default), row_tiles =
// 16-row tiles of W per block (1 or 4; 0 = chosen from M and N).
// tiled != 0: W is stored in the kernel's streaming order instead of row-major: [N/16][K/32] tiles
// of 1 KiB, each holding, for lane l = 0..63, the 8 elements W[16 nt + l%16][32 ks + 8 (l/16) .. +8]
// (fp8: [N/16][K/64] tiles, 16 elements per lane); see ops.llm.tile_weights.
// w_stride: elements between consecutive rows of a row-major W (0 = K; a multiple of 8, of 16 for fp8). Rows a
// power of two apart (K = 16384: 32 KiB) make the 16 rows of a tile collide in the memory system: 64 elements of
// padding per row are worth 12 % there (docs/KERNELS.md).
int klm_skinny_linear(const void* x, int x_stride, const void* W, void* out, const void* residual, int M, int N,
                      int K, int rms, int epilogue, float eps, int waves, int row_tiles, int tiled, int w_stride,
                      void* stream);

// The same with OCP e4m3 weights W8[N][K] (bytes) and one fp32 scale per output row (W8A16): the
// weights are converted to bf16 in registers, x stays bf16. wscale has N entries, 16-byte aligned.
int klm_skinny_linear_w8(const void* x, int x_stride, const void* W8, const float* wscale, void* out,
                         const void* residual, int M, int N, int K, int rms, int epilogue, float eps, int waves,
                         int row_tiles, int tiled, int w_stride, void* stream);

// Linear layers of decode at batch 17 - 128 (llm_mid.hip): out = x[M][K] . W[N][K]^T with W streamed from HBM into
// MFMA operands and x staged through LDS; fp32 partial sums part[ksplit][M][N], then an epilogue kernel that sums the
// K splits and applies kind 0: plain (out[M][N] bf16), 1: SwiGLU on gate/up columns interleaved by 8 (out[M][N/2]),
// 2: residual[M][N] += sum, out = rmsnorm(residual) with unit gain. M <= 128, N % 64 == 0, K % 128 == 0.
// klm_mid_linear_plan fills in the zeros among ksplit / waves (4, 8) / row_tiles (1, 2) the way the launcher would.
int klm_mid_linear_plan(int M, int N, int K, int* ksplit, int* waves, int* row_tiles);
int klm_mid_linear(const void* x, int x_stride, const void* W, int w_stride, float* part, int M, int N, int K, int ksplit,
                   int waves, int row_tiles, void* stream);
int klm_mid_linear_w8(const void* x, int x_stride, const void* W8, const float* wscale, float* part, int M, int N, int K,
                      int ksplit, int waves, int row_tiles, void* stream);     // e4m3 weights, tiled order, fp32 row scales
// gate_up without a K split: SwiGLU inside the main kernel, act[M][N/2] bf16 out, no epilogue launch (wscale: e4m3 tiled W).
int klm_mid_linear_swiglu(const void* x, int x_stride, const void* W, int w_stride, const float* wscale, void* act, int M, int N,
                          int K, int waves, int row_tiles, void* stream);
// The qkv projection's epilogue: sums the K splits, applies RoPE, fills `qkv` [T][(n_q + 2 n_kv) * 128] and the caches
// (klm_rope_cache fed by partials: one launch instead of two).
int klm_rope_cache_from_partials(const float* part, int ksplit, void* qkv, const float* cos_sin, const int32_t* positions,
                                 const int32_t* slots, void* k_cache, void* v_cache, int T, int n_q, int n_kv, int max_pos,
                                 int64_t num_slots, int kv_fp8, void* stream);
int klm_mid_epilogue(const float* part, int ksplit, int M, int N, int kind, void* out, void* residual, float eps, void* stream);

const char* klm_version(void);
const char* klm_src_hash(void);   /* hash of the sources this library was built from (build.py), "unhashed" otherwise */
int klm_hbm_launch_plan(size_t bytes, int u, unsigned* first_tiles, unsigned* last_tiles);   /* launches a buffer of `bytes` takes */

#ifdef __cplusplus
}
#endif
