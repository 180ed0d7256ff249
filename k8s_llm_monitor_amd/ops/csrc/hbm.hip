// HBM load kernels for MI355X (gfx950): fill (write-only), copy (read+write), read-only
// stream, checksum and a traffic-free spin. SURVEY.md 2.4 N3 (HBM fill / exhaustion
// injector) and N4 (HBM bandwidth burner); no reference counterpart.
//
// 256 threads per block (4 waves of 64). Block -> data mapping and cache hints were picked by sweeps on MI355X
// (scripts/hbm_sweep.py, profiles/r1_prof/hbm_sweep_*.log; scripts/exp/hbm_one_shot.py, profiles/r2_prof/hbm_one_shot.json).
// What ships is the ONE-SHOT form with ONE 16-byte access per thread: no loop, a workgroup touches 4 KiB and retires, so
// the order workgroups are dispatched in IS the address order and every memory channel sees the same steady stream:
//   fill  plain stores                          6.9 TB/s   (looping forms, 4 stores in flight per lane: 5.9 - 6.15)
//   copy  nontemporal loads and stores          6.4 TB/s   read + write   (16 KiB tile per block: 5.65; torch copy_: 4.6)
//   read  nontemporal loads                     7.1 TB/s   (grid-stride loop over 2048 blocks: 6.1)
// (the first version, a 2048-block grid-stride loop with nontemporal stores, wrote 4.6-4.7).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "klm_ops.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

static constexpr int kBlock = 256;
static constexpr int kMaxGrid = 256 * 8;
static constexpr int kUnroll = 4;

static inline int grid_for(size_t n16) {
  size_t per_block = (size_t)kBlock * kUnroll;
  size_t g = (n16 + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > (size_t)kMaxGrid) g = kMaxGrid;
  return (int)g;
}

// Read-only stream. The XOR of everything read is written once per block only when it
// equals a value it (practically) never has, which keeps the loads alive without a store
// stream of its own.
__global__ __launch_bounds__(kBlock) void hbm_read_kernel(const u32x4* __restrict__ src, size_t n16,
                                                          uint32_t* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
    u32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) acc ^= v[u];
  }
  for (; i < n16; i += stride) acc ^= src[i];
  const uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (x == 0xA5C3F00Du) sink[0] = x;
}

__global__ __launch_bounds__(kBlock) void hbm_checksum_kernel(const u32x4* __restrict__ src,
                                                              size_t n16,
                                                              unsigned long long* __restrict__ sum) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) {
    const u32x4 v = src[i];
    s += (unsigned long long)v[0] + v[1] + v[2] + v[3];
  }
  // wave reduction (64 lanes), then one atomic per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(sum, s);
}

// Busy without memory traffic or matrix work: every wave spins on the shader clock until
// `cycles` ticks have passed. The exit condition depends only on the wave's own clock
// reads, so every wave reaches it.
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long cycles) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}

// Tuning variants of the copy / fill streams (scripts/hbm_sweep.py picks the defaults):
// flags bit 0: nontemporal loads, bit 1: nontemporal stores, bit 2: each block walks one
// contiguous chunk instead of the grid-strided interleave. `grid` = blocks (0 = default).
template <bool NTL, bool NTS, bool CHUNK>
__global__ __launch_bounds__(kBlock) void hbm_copy_ex_kernel(u32x4* __restrict__ dst,
                                                             const u32x4* __restrict__ src, size_t n16) {
  size_t i, end, stride;
  if (CHUNK) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    i = (size_t)blockIdx.x * per + threadIdx.x;
    end = (size_t)(blockIdx.x + 1) * per < n16 ? (size_t)(blockIdx.x + 1) * per : n16;
    stride = kBlock;
  } else {
    i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    end = n16;
    stride = (size_t)gridDim.x * kBlock;
  }
  for (; i + (kUnroll - 1) * stride < end; i += kUnroll * stride) {
    u32x4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = NTL ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if (NTS) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < end; i += stride) dst[i] = src[i];
}

template <bool NTS, bool CHUNK>
__global__ __launch_bounds__(kBlock) void hbm_fill_ex_kernel(u32x4* __restrict__ dst, size_t n16, uint32_t pattern) {
  const u32x4 v = {pattern, pattern, pattern, pattern};
  size_t i, end, stride;
  if (CHUNK) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    i = (size_t)blockIdx.x * per + threadIdx.x;
    end = (size_t)(blockIdx.x + 1) * per < n16 ? (size_t)(blockIdx.x + 1) * per : n16;
    stride = kBlock;
  } else {
    i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    end = n16;
    stride = (size_t)gridDim.x * kBlock;
  }
  for (; i + (kUnroll - 1) * stride < end; i += kUnroll * stride) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if (NTS) __builtin_nontemporal_store(v, dst + i + u * stride);
      else dst[i + u * stride] = v;
    }
  }
  for (; i < end; i += stride) dst[i] = v;
}

// One-shot form: no loop, one tile of kBlock x U x 16 bytes per block (U = 4: 16 KiB; U = 1: one 16-byte access per thread,
// the shape of torch's vectorized fill, which writes 6.9 TB/s where every looping form stays at 6.1), grid = tiles.
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(kBlock) void hbm_copy_tile_kernel(u32x4* __restrict__ dst,
                                                               const u32x4* __restrict__ src, size_t n16) {
  const size_t base = (size_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  if (base + (U - 1) * kBlock < n16) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(src + base + u * kBlock) : src[base + u * kBlock];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTS) __builtin_nontemporal_store(v[u], dst + base + u * kBlock);
      else dst[base + u * kBlock] = v[u];
    }
  } else {
    for (size_t i = base; i < n16; i += kBlock) dst[i] = src[i];
  }
}

template <bool NTS, int U>
__global__ __launch_bounds__(kBlock) void hbm_fill_tile_kernel(u32x4* __restrict__ dst, size_t n16, uint32_t pattern) {
  const u32x4 v = {pattern, pattern, pattern, pattern};
  const size_t base = (size_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  if (base + (U - 1) * kBlock < n16) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTS) __builtin_nontemporal_store(v, dst + base + u * kBlock);
      else dst[base + u * kBlock] = v;
    }
  } else {
    for (size_t i = base; i < n16; i += kBlock) dst[i] = v;
  }
}

// one-shot read: every thread loads U x 16 bytes; the store never happens (see hbm_read_kernel)
template <bool NTL, int U>
__global__ __launch_bounds__(kBlock) void hbm_read_tile_kernel(const u32x4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
  const size_t base = (size_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  if (base + (U - 1) * kBlock < n16) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= NTL ? __builtin_nontemporal_load(src + base + u * kBlock) : src[base + u * kBlock];
  } else {
    for (size_t i = base; i < n16; i += kBlock) acc ^= src[i];
  }
  const uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (x == 0xA5C3F00Du) sink[0] = x;
}

static inline size_t tiles_for(size_t n16, int u = kUnroll) { return (n16 + (size_t)kBlock * u - 1) / ((size_t)kBlock * u); }

// HIP refuses a launch whose gridDim.x * blockDim.x passes 2^32 - 1, so a one-shot grid holds at most this many blocks:
// 16 777 215 tiles of 4 KiB = just under 64 GiB per launch at one access per thread. The card has 288 GB, so the default
// entry points walk a larger buffer in several launches over sub-ranges; the tuning (_ex) entry points refuse it.
static const size_t kMaxTiles = 0xFFFFFFFFull / kBlock;

// sub-ranges [off, off + n) in 16-byte elements, each at most kMaxTiles tiles of kBlock * u elements: fn(off, n, tiles)
template <typename F>
static inline void for_each_launch(size_t n16, int u, F fn) {
  const size_t per = kMaxTiles * (size_t)kBlock * (size_t)u;
  for (size_t off = 0; off < n16; off += per) {
    const size_t n = n16 - off < per ? n16 - off : per;
    fn(off, n, tiles_for(n, u));
  }
}

// how a buffer of `bytes` is launched at `u` accesses per thread: number of launches; *first_tiles / *last_tiles = their grids
extern "C" int klm_hbm_launch_plan(size_t bytes, int u, unsigned* first_tiles, unsigned* last_tiles) {
  if (u < 1 || u > 4 || bytes == 0 || (bytes & 15)) return KLM_ERR_ARG;
  int launches = 0;
  for_each_launch(bytes >> 4, u, [&](size_t, size_t, size_t tiles) {
    if (launches == 0 && first_tiles) *first_tiles = (unsigned)tiles;
    if (last_tiles) *last_tiles = (unsigned)tiles;
    ++launches;
  });
  return launches;
}

static inline int check(const void* a, const void* b, size_t bytes) {
  if (!a) return KLM_ERR_ARG;
  if (((uintptr_t)a | (uintptr_t)b) & 15) return KLM_ERR_ALIGN;
  if (bytes == 0 || (bytes & 15)) return KLM_ERR_SHAPE;
  return KLM_OK;
}

static inline int done() { return hipGetLastError() == hipSuccess ? KLM_OK : KLM_ERR_HIP; }

extern "C" int klm_hbm_fill(void* dst, size_t bytes, uint32_t pattern, void* stream) {
  if (int e = check(dst, nullptr, bytes)) return e;
  // one 16-byte store per thread: 6.9 TB/s (the chunked loop over 16384 blocks wrote 6.15)
  for_each_launch(bytes >> 4, 1, [&](size_t off, size_t n, size_t tiles) {
    hipLaunchKernelGGL((hbm_fill_tile_kernel<false, 1>), dim3((unsigned)tiles), dim3(kBlock), 0, (hipStream_t)stream,
                       (u32x4*)dst + off, n, pattern);
  });
  return done();
}

extern "C" int klm_hbm_copy(void* dst, const void* src, size_t bytes, void* stream) {
  if (!src) return KLM_ERR_ARG;
  if (int e = check(dst, src, bytes)) return e;
  // one 16-byte load + store per thread, both nontemporal: 6.44 TB/s (16 KiB tiles: 5.65; the strided loop: 4.74)
  for_each_launch(bytes >> 4, 1, [&](size_t off, size_t n, size_t tiles) {
    hipLaunchKernelGGL((hbm_copy_tile_kernel<true, true, 1>), dim3((unsigned)tiles), dim3(kBlock), 0, (hipStream_t)stream,
                       (u32x4*)dst + off, (const u32x4*)src + off, n);
  });
  return done();
}

extern "C" int klm_hbm_copy_ex(void* dst, const void* src, size_t bytes, int flags, int grid, void* stream) {
  if (!src) return KLM_ERR_ARG;
  if (int e = check(dst, src, bytes)) return e;
  if (grid < 0 || grid > 1 << 20 || flags < 0 || flags > 31) return KLM_ERR_ARG;
  const size_t n16 = bytes >> 4;
  const dim3 g(grid ? grid : grid_for(n16)), b(kBlock);
  hipStream_t s = (hipStream_t)stream;
  u32x4* d = (u32x4*)dst;
  const u32x4* r = (const u32x4*)src;
  if (flags & 24) {      // one-shot tiles: 8 -> 4 accesses per thread, 16 -> 1, 24 -> 2
    const int u = (flags & 24) == 8 ? 4 : (flags & 24) == 16 ? 1 : 2;
    const size_t tiles = tiles_for(n16, u);
    if (tiles > kMaxTiles) return KLM_ERR_SHAPE;      // past HIP's grid limit: refused, explicitly (the default entry points split)
    const dim3 gt((unsigned)tiles);
#define KLM_COPY_TILE(U)                                                                                  \
    switch (flags & 3) {                                                                                  \
      case 0: hipLaunchKernelGGL((hbm_copy_tile_kernel<false, false, U>), gt, b, 0, s, d, r, n16); break; \
      case 1: hipLaunchKernelGGL((hbm_copy_tile_kernel<true, false, U>), gt, b, 0, s, d, r, n16); break;  \
      case 2: hipLaunchKernelGGL((hbm_copy_tile_kernel<false, true, U>), gt, b, 0, s, d, r, n16); break;  \
      default: hipLaunchKernelGGL((hbm_copy_tile_kernel<true, true, U>), gt, b, 0, s, d, r, n16); break;  \
    }
    if (u == 4) { KLM_COPY_TILE(4) } else if (u == 1) { KLM_COPY_TILE(1) } else { KLM_COPY_TILE(2) }
#undef KLM_COPY_TILE
    return done();
  }
  switch (flags) {
    case 0: hipLaunchKernelGGL((hbm_copy_ex_kernel<false, false, false>), g, b, 0, s, d, r, n16); break;
    case 1: hipLaunchKernelGGL((hbm_copy_ex_kernel<true, false, false>), g, b, 0, s, d, r, n16); break;
    case 2: hipLaunchKernelGGL((hbm_copy_ex_kernel<false, true, false>), g, b, 0, s, d, r, n16); break;
    case 3: hipLaunchKernelGGL((hbm_copy_ex_kernel<true, true, false>), g, b, 0, s, d, r, n16); break;
    case 4: hipLaunchKernelGGL((hbm_copy_ex_kernel<false, false, true>), g, b, 0, s, d, r, n16); break;
    case 5: hipLaunchKernelGGL((hbm_copy_ex_kernel<true, false, true>), g, b, 0, s, d, r, n16); break;
    case 6: hipLaunchKernelGGL((hbm_copy_ex_kernel<false, true, true>), g, b, 0, s, d, r, n16); break;
    default: hipLaunchKernelGGL((hbm_copy_ex_kernel<true, true, true>), g, b, 0, s, d, r, n16); break;
  }
  return done();
}

extern "C" int klm_hbm_fill_ex(void* dst, size_t bytes, uint32_t pattern, int flags, int grid, void* stream) {
  if (int e = check(dst, nullptr, bytes)) return e;
  if (grid < 0 || grid > 1 << 20 || flags < 0 || flags > 31) return KLM_ERR_ARG;
  const size_t n16 = bytes >> 4;
  const dim3 g(grid ? grid : grid_for(n16)), b(kBlock);
  hipStream_t s = (hipStream_t)stream;
  u32x4* d = (u32x4*)dst;
  if (flags & 24) {      // one-shot tiles: 8 -> 4 stores per thread, 16 -> 1, 24 -> 2
    const int u = (flags & 24) == 8 ? 4 : (flags & 24) == 16 ? 1 : 2;
    const size_t tiles = tiles_for(n16, u);
    if (tiles > kMaxTiles) return KLM_ERR_SHAPE;      // past HIP's grid limit: refused, explicitly (the default entry points split)
    const dim3 gt((unsigned)tiles);
    const bool nts = flags & 2;
    if (u == 4) { if (nts) hipLaunchKernelGGL((hbm_fill_tile_kernel<true, 4>), gt, b, 0, s, d, n16, pattern); else hipLaunchKernelGGL((hbm_fill_tile_kernel<false, 4>), gt, b, 0, s, d, n16, pattern); }
    else if (u == 1) { if (nts) hipLaunchKernelGGL((hbm_fill_tile_kernel<true, 1>), gt, b, 0, s, d, n16, pattern); else hipLaunchKernelGGL((hbm_fill_tile_kernel<false, 1>), gt, b, 0, s, d, n16, pattern); }
    else { if (nts) hipLaunchKernelGGL((hbm_fill_tile_kernel<true, 2>), gt, b, 0, s, d, n16, pattern); else hipLaunchKernelGGL((hbm_fill_tile_kernel<false, 2>), gt, b, 0, s, d, n16, pattern); }
    return done();
  }
  const bool nts = flags & 2, chunk = flags & 4;
  if (!nts && !chunk) hipLaunchKernelGGL((hbm_fill_ex_kernel<false, false>), g, b, 0, s, d, n16, pattern);
  else if (nts && !chunk) hipLaunchKernelGGL((hbm_fill_ex_kernel<true, false>), g, b, 0, s, d, n16, pattern);
  else if (!nts && chunk) hipLaunchKernelGGL((hbm_fill_ex_kernel<false, true>), g, b, 0, s, d, n16, pattern);
  else hipLaunchKernelGGL((hbm_fill_ex_kernel<true, true>), g, b, 0, s, d, n16, pattern);
  return done();
}

extern "C" int klm_hbm_read(const void* src, size_t bytes, void* sink8, void* stream) {
  if (!sink8) return KLM_ERR_ARG;
  if (int e = check(src, nullptr, bytes)) return e;
  // one nontemporal 16-byte load per thread: 7.07 TB/s (the grid-stride loop: 6.09)
  for_each_launch(bytes >> 4, 1, [&](size_t off, size_t n, size_t tiles) {
    hipLaunchKernelGGL((hbm_read_tile_kernel<true, 1>), dim3((unsigned)tiles), dim3(kBlock), 0, (hipStream_t)stream,
                       (const u32x4*)src + off, n, (uint32_t*)sink8);
  });
  return done();
}

// Tuning variants of the read stream: flags bit 0 nontemporal loads; bits 3-4 one-shot tiles (8: 4 loads per thread, 16: 1, 24: 2),
// else the grid-stride loop over `grid` blocks (0 = default).
extern "C" int klm_hbm_read_ex(const void* src, size_t bytes, void* sink8, int flags, int grid, void* stream) {
  if (!sink8) return KLM_ERR_ARG;
  if (int e = check(src, nullptr, bytes)) return e;
  if (grid < 0 || grid > 1 << 20 || flags < 0 || flags > 31) return KLM_ERR_ARG;
  const size_t n16 = bytes >> 4;
  hipStream_t s = (hipStream_t)stream;
  const u32x4* r = (const u32x4*)src;
  uint32_t* k = (uint32_t*)sink8;
  if (!(flags & 24)) {
    hipLaunchKernelGGL(hbm_read_kernel, dim3(grid ? grid : grid_for(n16)), dim3(kBlock), 0, s, r, n16, k);
    return done();
  }
  const int u = (flags & 24) == 8 ? 4 : (flags & 24) == 16 ? 1 : 2;
  const size_t tiles = tiles_for(n16, u);
  if (tiles > kMaxTiles) return KLM_ERR_SHAPE;      // past HIP's grid limit: refused, explicitly (the default entry points split)
  const dim3 gt((unsigned)tiles), b(kBlock);
  const bool ntl = flags & 1;
  if (u == 4) { if (ntl) hipLaunchKernelGGL((hbm_read_tile_kernel<true, 4>), gt, b, 0, s, r, n16, k); else hipLaunchKernelGGL((hbm_read_tile_kernel<false, 4>), gt, b, 0, s, r, n16, k); }
  else if (u == 1) { if (ntl) hipLaunchKernelGGL((hbm_read_tile_kernel<true, 1>), gt, b, 0, s, r, n16, k); else hipLaunchKernelGGL((hbm_read_tile_kernel<false, 1>), gt, b, 0, s, r, n16, k); }
  else { if (ntl) hipLaunchKernelGGL((hbm_read_tile_kernel<true, 2>), gt, b, 0, s, r, n16, k); else hipLaunchKernelGGL((hbm_read_tile_kernel<false, 2>), gt, b, 0, s, r, n16, k); }
  return done();
}

// Rows of `row_bytes` from a matrix with row pitch `src_pitch` into one with row pitch `dst_pitch` (all multiples of 16 bytes):
// the layout change behind ops.gemm_bf16's answer to row pitches the memory channels camp on (a dense 24 KiB pitch costs a bf16
// GEMM 15 - 20 %; the same rows 256 bytes further apart do not). One 16-byte nontemporal load + store per thread and row.
__global__ __launch_bounds__(kBlock) void hbm_repitch_kernel(u32x4* __restrict__ dst, size_t dst_pitch16, const u32x4* __restrict__ src,
                                                            size_t src_pitch16, int rows, int row16) {
  const int c = (int)blockIdx.x * kBlock + (int)threadIdx.x;
  if (c >= row16) return;
  for (int r = (int)blockIdx.y; r < rows; r += (int)gridDim.y)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + (size_t)r * src_pitch16 + c), dst + (size_t)r * dst_pitch16 + c);
}

extern "C" int klm_hbm_repitch(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, int rows, size_t row_bytes, void* stream) {
  if (!dst || !src) return KLM_ERR_ARG;
  if (((uintptr_t)dst | (uintptr_t)src) & 15) return KLM_ERR_ALIGN;
  if (rows <= 0 || row_bytes == 0 || ((row_bytes | dst_pitch | src_pitch) & 15) || dst_pitch < row_bytes || src_pitch < row_bytes ||
      row_bytes > ((size_t)1 << 34))
    return KLM_ERR_SHAPE;
  const int row16 = (int)(row_bytes >> 4);
  const dim3 grid((unsigned)((row16 + kBlock - 1) / kBlock), (unsigned)(rows < 65535 ? rows : 65535));
  hipLaunchKernelGGL(hbm_repitch_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, (u32x4*)dst, dst_pitch >> 4, (const u32x4*)src,
                     src_pitch >> 4, rows, row16);
  return done();
}

extern "C" int klm_hbm_checksum(const void* src, size_t bytes, void* sum64, void* stream) {
  if (!sum64 || ((uintptr_t)sum64 & 7)) return KLM_ERR_ARG;
  if (int e = check(src, nullptr, bytes)) return e;
  const size_t n16 = bytes >> 4;
  hipLaunchKernelGGL(hbm_checksum_kernel, dim3(grid_for(n16)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const u32x4*)src, n16, (unsigned long long*)sum64);
  return done();
}

extern "C" int klm_spin(uint64_t cycles, int blocks, void* stream) {
  if (blocks <= 0 || blocks > 65536) return KLM_ERR_ARG;
  // bound the spin: at most ~2 s at 2.4 GHz (s_memtime ticks are shader cycles)
  if (cycles > 5000000000ull) return KLM_ERR_ARG;
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream,
                     (unsigned long long)cycles);
  return done();
}

// KLM_SRC_HASH: the build's hash over this target's sources and headers (k8s_llm_monitor_amd/build.py passes it; a build by
// other means says so). build.py finds the marker in the binary to decide staleness; smoke() compares it with the tree.
#ifndef KLM_SRC_HASH
#define KLM_SRC_HASH "unhashed"
#endif
extern "C" __attribute__((used)) const char klm_src_hash_marker[] = "KLM_SRC_HASH=" KLM_SRC_HASH;
extern "C" const char* klm_version(void) { return "klm-ops 0.2 gfx950 src:" KLM_SRC_HASH; }
extern "C" const char* klm_src_hash(void) { return KLM_SRC_HASH; }
