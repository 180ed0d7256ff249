// LLM serving kernels for MI355X (gfx950): the linear layers of decode at batch 17 - 128.
// No reference counterpart (vLLM is a README bullet in the reference, README.md:73).
//
// At these batch sizes a projection is still a weight-streaming problem (64 - 256 FLOP per weight byte, the MFMA
// pipe needs < 20 % of its rate to keep up with HBM), but the library GEMMs run it at 1.3 - 4.5 TB/s on the
// Llama-3-8B shapes (scripts/exp/tunableop_mid_batch.py), and the skinny kernel's trick - x fragments straight
// from L2 - stops working because the x traffic per block (M x K x 2 bytes) outgrows its weight traffic.
//
//   out[M][N] = x[M][K] . W[N][K]^T,  16 < M <= 128
//
// mid_linear_kernel: a block of NW waves owns NW x RT x 16 rows of W and one K range (split-K over blockIdx.y, so
// that small N still fills 256 CUs). W goes from HBM straight into MFMA operand registers, as in the skinny kernel
// (16-byte loads, 16 rows x 64 bytes per wave instruction, the next 128-wide K chunk in flight while this one is
// multiplied). The x chunk [M][128] is staged ONCE per block through LDS by producer waves (double-buffered, 16-byte
// slots XOR-swizzled by row so that the 16 rows a fragment read touches fall into 16 different bank groups) and
// every consumer wave reads its M/16 fragments from there: per KiB of weights, M/16 (RT = 1) or M/32 (RT = 2) KiB
// of LDS reads.
// Partial sums go out in fp32, [ksplit][M][N].
//
// mid_epilogue_kernel: sums the K splits and finishes the layer step that would otherwise be a kernel of its own:
//   NONE     out = bf16(sum)
//   SWIGLU   gate / up columns interleaved by 8 (see llm_elementwise.hip): out[M][N/2] = silu(g) * u
//   RES_RMS  residual <- bf16(residual + sum); out <- rmsnorm(residual)   (unit gain: gains are folded into W)
#include "llm_common.h"

using namespace klm_llm;

namespace {

enum { MID_NONE = 0, MID_SWIGLU = 1, MID_RES_RMS = 2 };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // plain vector: register arrays of it stay out of scratch
constexpr int KC = 128;                 // K elements per chunk
constexpr int SLOTS = KC / 8;           // 16-byte slots per x row in a chunk

// Waves 0 .. NW-1 stream weights and multiply; waves NW .. NW+NP-1 only stage x. The split matters because the
// memory counter (vmcnt) retires in issue order: a wave that waits for its x loads also waits for every weight
// load it issued before them, which caps the weight prefetch at one chunk (the first version of this kernel:
// 3.4 - 4.0 TB/s). With x on waves of its own, a consumer keeps D chunks of weights in flight in a register ring
// (16 KiB per wave) and its only waits are for the oldest ring entry.
//
// W8: the weights are e4m3 bytes in the skinny kernel's tiled order (ops.llm.tile_weights: the 1 KiB a wave
// instruction loads - 16 rows x 64 k-values - is contiguous, tiles ordered [row tile][k / 64]) with one fp32 scale per
// row. A lane's 16 bytes become two bf16 fragments in registers (k-values 0..7 and 8..15 of its 16), multiplied with
// the x slots at the same k offsets; the scale multiplies the fp32 partial sums. Half the bytes, contiguous runs.
template <int MT, int RT, int NW, bool W8, bool SWIGLU>
__global__ __launch_bounds__((NW + (MT + 1) / 2) * 64) void mid_linear_kernel(const uint16_t* __restrict__ x, int x_stride,
                                                                              const void* __restrict__ Wv, int w_stride,
                                                                              const float* __restrict__ wscale,
                                                                              float* __restrict__ part, int M, int N, int K, int kper) {
  constexpr int ROWS = MT * 16;                       // x rows held in LDS (rows >= M are zero)
  constexpr int NP = (MT + 1) / 2;                    // producer waves: 8 loads of 16 bytes per lane and chunk
  constexpr int XL = ROWS * SLOTS / (NP * 64);
  constexpr int PD = 2;                               // x chunks in flight per producer (registers)
  constexpr int D = RT == 1 ? 4 : 2;                  // weight chunks in flight per consumer (registers)
  static_assert(XL * NP * 64 == ROWS * SLOTS, "x chunk must divide over the producers");
  __shared__ u32x4 xs[2][ROWS * SLOTS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = blockIdx.y * kper;
  const int nchunks = kper / KC;
  // Blocks walk their K range in different rotations (a sum does not care about the order): with every block
  // starting at chunk 0, all waves of the grid read the same 256-byte column of rows that are a power of two
  // apart at the same time and pile onto a few memory channels (4.8 TB/s; rotated: see docs/KERNELS.md).
  const int start = (blockIdx.x * 5 + blockIdx.y * 3) % nchunks;
  auto phys = [&](int j) { const int p = j + start; return (p >= nchunks ? p - nchunks : p) * KC; };

  if (w >= NW) {
    // ---------------- producer: x[M][kb + c*128 ..] -> xs[c & 1], slot i = row i / 16, logical slot i % 16 at (ls ^ row % 16)
    const int pt = tid - NW * 64;
    const int ls = pt % SLOTS, row0 = pt / SLOTS;
    constexpr int RSTEP = NP * 64 / SLOTS;
    const uint16_t* xsrc = x + kb + ls * 8;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    u32x4 pr[PD][XL];
    auto fetch = [&](int chunk, u32x4* dst) {
#pragma unroll
      for (int j = 0; j < XL; ++j) {
        const int row = row0 + j * RSTEP;
        dst[j] = row < M ? *reinterpret_cast<const u32x4*>(xsrc + (size_t)row * x_stride + phys(chunk)) : zero4;
      }
    };
    auto put = [&](int buf, const u32x4* src) {
#pragma unroll
      for (int j = 0; j < XL; ++j) {
        const int row = row0 + j * RSTEP;
        xs[buf][row * SLOTS + (ls ^ (row & 15))] = src[j];
      }
    };
#pragma unroll
    for (int p = 0; p < PD; ++p)
      if (p < nchunks) fetch(p, pr[p]);
    put(0, pr[0]);
    if (PD < nchunks) fetch(PD, pr[0]);
    // chunk c + 1 sits in pr[(c + 1) % PD]; it is written while the consumers multiply chunk c
    for (int c0 = 0; c0 < nchunks; c0 += 2 * PD) {
#pragma unroll
      for (int u = 0; u < 2 * PD; ++u) {
        const int c = c0 + u;
        if (c < nchunks) {
          __syncthreads();
          if (c + 1 < nchunks) {
            put((u + 1) & 1, pr[(u + 1) % PD]);
            if (c + 1 + PD < nchunks) fetch(c + 1 + PD, pr[(u + 1) % PD]);
          }
        }
      }
    }
    return;
  }

  // ---------------- consumers
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = (blockIdx.x * NW + w) * RT * 16;
  // bf16: a lane's 16 bytes of k step s sit at row n0 + fr, k = chunk + s*32 + fq*8. fp8: tile (row tile, k / 64) is
  // 1 KiB, lane l at byte 16 l; a 128-wide chunk is two tiles. `cstep` = distance between the chunk's loads.
  constexpr int LOADS = W8 ? 2 : 4;                   // 16-byte weight loads per row tile and chunk
  const char* wp[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
    wp[r] = W8 ? static_cast<const char*>(Wv) + ((size_t)((n0 >> 4) + r) * (K >> 6) + (kb >> 6)) * 1024 + lane * 16
               : static_cast<const char*>(Wv) + ((size_t)(n0 + r * 16 + fr) * w_stride + kb + fq * 8) * 2;
  constexpr int cstep = W8 ? 1024 : 64;               // bytes
  auto wofs = [&](int j) { return (size_t)phys(j) * (W8 ? 16 : 2); };     // chunk -> byte offset (fp8 tiled: 2 KiB per chunk)

  f32x4 acc[MT][RT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The launcher guarantees nchunks % D == 0, and every refill is issued unconditionally (past the end it re-reads
  // the last chunk into a ring entry nobody will use): with no branch around a load the compiler can count the
  // loads in flight and waits for exactly the oldest ring entry, not for all of them.
  u32x4 ring[D][RT][LOADS];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int s = 0; s < LOADS; ++s) ring[d][r][s] = *reinterpret_cast<const u32x4*>(wp[r] + wofs(d) + s * cstep);

  for (int c0 = 0; c0 < nchunks; c0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int c = c0 + u;
      __syncthreads();                                // x chunk c is in xs[c & 1]  (D is even: c & 1 == u & 1)
      const u32x4* buf = xs[u & 1];
      // k step s of the chunk (32 k-values per MFMA): bf16 lanes hold k = s*32 + fq*8 .. +8 = x slot s*4 + fq;
      // fp8 lanes hold k = (s/2)*64 + fq*16 + (s%2)*8 .. +8 = x slot (s/2)*8 + fq*2 + s%2
      auto slot = [&](int s) { return W8 ? (s >> 1) * 8 + fq * 2 + (s & 1) : s * 4 + fq; };
      // the x fragments of k step s + 1 are read while the MFMAs of step s run (the scheduler barriers keep the
      // compiler from folding the reads back into read-wait-multiply pairs)
      // (with 128 rows the second fragment buffer would not fit the register budget of 12 waves per CU: there the
      // reads of a step go out together, right before its MFMAs)
      constexpr int XB = MT <= 4 ? 2 : 1;
      u32x4 xf[XB][MT];
      if (XB == 2) {
#pragma unroll
        for (int t = 0; t < MT; ++t) xf[0][t] = buf[(t * 16 + fr) * SLOTS + (slot(0) ^ fr)];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int sn = XB == 2 ? s + 1 : s;           // the step whose fragments are fetched now
        if (sn < 4) {
#pragma unroll
          for (int t = 0; t < MT; ++t) xf[sn % XB][t] = buf[(t * 16 + fr) * SLOTS + (slot(sn) ^ fr)];
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 wf[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          if (W8) {
            const u32x4 raw = ring[u][r][s >> 1];
            wf[r] = (s & 1) ? fp8x8_to_bf16(raw[2], raw[3]) : fp8x8_to_bf16(raw[0], raw[1]);
          } else {
            wf[r] = __builtin_bit_cast(bf16x8, ring[u][r][s]);
          }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < RT; ++r)
            acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[r], __builtin_bit_cast(bf16x8, xf[s % XB][t]), acc[t][r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const size_t nxt = wofs(min(c + D, nchunks - 1));
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int s = 0; s < LOADS; ++s) ring[u][r][s] = *reinterpret_cast<const u32x4*>(wp[r] + nxt + s * cstep);
    }
  }

  if (W8) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(wscale + n0 + r * 16 + fq * 4);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][r] *= sc;
    }
  }

  if (SWIGLU) {
    // no K split (gridDim.y == 1): the sums are final, and a 16-row tile of gate_up holds 8 gate and the 8 matching up
    // columns. Lanes fq 0,1 carry gate columns fq*4+i, lanes fq 2,3 the up columns: one cross-lane exchange pairs them and
    // the activation goes out as bf16 [M][N/2], with no partial sums and no epilogue launch.
    uint16_t* act = reinterpret_cast<uint16_t*>(part);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int row = t * 16 + fr;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        f32x4 u;
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = __shfl_xor(acc[t][r][i], 32);
        if (row < M && fq < 2) {
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (__bf16)(acc[t][r][i] / (1.f + __expf(-acc[t][r][i])) * u[i]);
          *reinterpret_cast<bf16x4*>(act + (size_t)row * (N >> 1) + ((n0 + r * 16) >> 1) + fq * 4) = o;
        }
      }
    }
    return;
  }
  // D[w row fq*4+i][x row fr]: four consecutive output columns of one x row per lane
  float* pbase = part + (size_t)blockIdx.y * M * N;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int row = t * 16 + fr;
    if (row < M) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
        *reinterpret_cast<f32x4*>(pbase + (size_t)row * N + n0 + r * 16 + fq * 4) = acc[t][r];
    }
  }
}

__device__ __forceinline__ void add8(float* v, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
  v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
}

// NONE: one thread per 8 output columns. SWIGLU: one thread per 8 outputs = 16 input columns.
template <int KIND>
__global__ __launch_bounds__(256) void mid_epilogue_kernel(const float* __restrict__ part, int ks, int M, int N,
                                                           uint16_t* __restrict__ out) {
  const size_t vec_per_row = (size_t)N / (KIND == MID_SWIGLU ? 16 : 8);
  const size_t total = (size_t)M * vec_per_row, split = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / vec_per_row, j = i % vec_per_row;
    if (KIND == MID_SWIGLU) {
      float g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
      const float* p = part + row * N + j * 16;
#pragma unroll 4
      for (int k = 0; k < ks; ++k) {
        add8(g, p + k * split);
        add8(u, p + k * split + 8);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = g[e] / (1.f + __expf(-g[e])) * u[e];
      *reinterpret_cast<uint4*>(out + row * (N / 2) + j * 8) = pack8(o);
    } else {
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const float* p = part + row * N + j * 8;
#pragma unroll 4
      for (int k = 0; k < ks; ++k) add8(v, p + k * split);
      *reinterpret_cast<uint4*>(out + row * N + j * 8) = pack8(v);
    }
  }
}

// RES_RMS: one block per row; a thread keeps its (up to 8) vectors of 8 in registers between the two passes.
__global__ __launch_bounds__(256) void mid_res_rms_kernel(const float* __restrict__ part, int ks, int M, int N,
                                                          uint16_t* __restrict__ res, uint16_t* __restrict__ out, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const size_t split = (size_t)M * N, base = (size_t)row * N;
  float v[8][8];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int e = (c * 256 + tid) * 8;
    if (e < N) {
      float rv[8];
      unpack8(*reinterpret_cast<const uint4*>(res + base + e), rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
#pragma unroll 4
      for (int k = 0; k < ks; ++k) add8(v[c], part + k * split + base + e);      // independent loads: several splits in flight
#pragma unroll
      for (int j = 0; j < 8; ++j) v[c][j] = bf2f(f2bf(v[c][j] + rv[j]));
      *reinterpret_cast<uint4*>(res + base + e) = pack8(v[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[c][j] * v[c][j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  ss = red[0] + red[1] + red[2] + red[3];
  const float inv = rsqrtf(ss / (float)N + eps);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int e = (c * 256 + tid) * 8;
    if (e < N) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = v[c][j] * inv;
      *reinterpret_cast<uint4*>(out + base + e) = pack8(o);
    }
  }
}

struct MidArgs {
  const uint16_t* x;
  int x_stride;
  const void* W;
  int w_stride;
  const float* wscale;
  float* part;
  int M, N, K, kper;
};

template <int MT, int RT, bool W8, bool SW>
void launch_nw(int nw, dim3 grid, hipStream_t s, const MidArgs& a) {
#define KLM_MID_GO(NWV)                                                                                                  \
  hipLaunchKernelGGL((mid_linear_kernel<MT, RT, NWV, W8, SW>), grid, dim3((NWV + (MT + 1) / 2) * 64), 0, s, a.x, a.x_stride, a.W, \
                     a.w_stride, a.wscale, a.part, a.M, a.N, a.K, a.kper)
  if (nw == 8) KLM_MID_GO(8);
  else if (nw == 7) KLM_MID_GO(7);
  else KLM_MID_GO(4);
#undef KLM_MID_GO
}

template <int MT, bool W8, bool SW>
void launch_rt(int rt, int nw, dim3 grid, hipStream_t s, const MidArgs& a) {
  if (rt == 2) launch_nw<MT, 2, W8, SW>(nw, grid, s, a);
  else launch_nw<MT, 1, W8, SW>(nw, grid, s, a);
}

template <bool W8, bool SW>
void launch_mt(int rt, int nw, dim3 grid, hipStream_t s, const MidArgs& a) {
  if (a.M <= 32) launch_rt<2, W8, SW>(rt, nw, grid, s, a);
  else if (a.M <= 64) launch_rt<4, W8, SW>(rt, nw, grid, s, a);
  else launch_rt<8, W8, SW>(rt, nw, grid, s, a);
}

}  // namespace

extern "C" {

// The configuration the launcher would pick: rows of W per block and K splits. ksplit / waves / row_tiles = 0: chosen here.
int klm_mid_linear_plan(int M, int N, int K, int* ksplit, int* waves, int* row_tiles) {
  if (M <= 0 || M > 128 || N <= 0 || N % 64 != 0 || K <= 0 || K % KC != 0) return KLM_ERR_SHAPE;
  if (*waves == 7 && N % 112 != 0) return KLM_ERR_SHAPE;
  int rt = *row_tiles, nw = *waves, ks = *ksplit;
  // Measured on MI355X (scripts/gpu_mid_bench.py, docs/KERNELS.md): about one block per CU is best, so 7-wave
  // blocks where they make the row blocks a multiple of 256 (N = 28672: 256 blocks of 112 rows), 32-row tiles
  // (half the LDS reads per weight byte) unless the batch is small and 16-row tiles already give 256 blocks, and
  // the K split that brings the grid closest to 256 blocks.
  if (nw != 4 && nw != 7 && nw != 8) nw = (N % 112 == 0 && N / 112 >= 224) ? 7 : 4;
  if (N % (nw * 16) != 0) nw = 4;
  if (rt != 1 && rt != 2) {
    rt = N % (nw * 32) == 0 ? 2 : 1;
    if (rt == 2 && nw == 7 && M <= 64 && N / (nw * 16) <= 256) rt = 1;   // 256 blocks without a K split (level with 2 x 128 at 33 - 64 rows)
    if (rt == 2 && (long long)N * K <= (1LL << 24)) rt = 1;    // small matrices (8B wo): the same 256 blocks with half the K splits
  }
  if (N % (nw * rt * 16) != 0) rt = 1;
  if (N % (nw * rt * 16) != 0) return KLM_ERR_SHAPE;
  // a block's K range is a whole number of weight-ring turns (4 chunks of 128 with 16-row tiles, 2 with 32-row tiles)
  const int depth = rt == 1 ? 4 : 2;
  if (K % (KC * depth) != 0) return KLM_ERR_SHAPE;
  const int nblk = N / (nw * rt * 16), turns = K / (KC * depth);
  if (ks <= 0) {
    ks = 1;
    for (int cand = 2; cand <= turns; ++cand) {
      if (turns % cand != 0) continue;
      // closest to 256 blocks on a log scale: cand * nblk vs ks * nblk
      const double a = (double)cand * nblk / 256.0, b = (double)ks * nblk / 256.0;
      const double da = a > 1 ? a : 1 / a, db = b > 1 ? b : 1 / b;
      if (da < db) ks = cand;
    }
  }
  if (ks > turns) ks = turns;
  while (ks > 1 && turns % ks != 0) --ks;
  *ksplit = ks;
  *waves = nw;
  *row_tiles = rt;
  return KLM_OK;
}

static int mid_launch(const void* x, int x_stride, const void* W, int w_stride, const float* wscale, bool w8, float* part, int M,
                      int N, int K, int ksplit, int waves, int row_tiles, void* stream, bool swiglu = false) {
  int ks = ksplit, nw = waves, rt = row_tiles;
  const int rc = klm_mid_linear_plan(M, N, K, &ks, &nw, &rt);
  if (rc != KLM_OK) return rc;
  if (ksplit > 0 && ks != ksplit) return KLM_ERR_SHAPE;          // the caller sized `part` for its own split
  if (!x || !W || !part || (w8 && !wscale)) return KLM_ERR_ARG;
  if (x_stride % 8 != 0 || x_stride < K) return KLM_ERR_SHAPE;
  if (w8 ? (w_stride != K) : (w_stride % 8 != 0 || w_stride < K)) return KLM_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(W) || !aligned16(part) || !aligned16(wscale)) return KLM_ERR_ALIGN;
  const dim3 grid(N / (nw * rt * 16), ks);
  const MidArgs a{static_cast<const uint16_t*>(x), x_stride, W, w_stride, wscale, part, M, N, K, K / ks};
  auto s = static_cast<hipStream_t>(stream);
  if (swiglu) {
    if (ks != 1) return KLM_ERR_SHAPE;                           // the fused activation needs the whole K in one block
    if (w8) launch_mt<true, true>(rt, nw, grid, s, a);
    else launch_mt<false, true>(rt, nw, grid, s, a);
  } else if (w8) {
    launch_mt<true, false>(rt, nw, grid, s, a);
  } else {
    launch_mt<false, false>(rt, nw, grid, s, a);
  }
  return launch_status();
}

// part: fp32 [ksplit][M][N] (written in full). x rows x_stride apart, W rows w_stride apart (elements, multiples of 8).
int klm_mid_linear(const void* x, int x_stride, const void* W, int w_stride, float* part, int M, int N, int K, int ksplit,
                   int waves, int row_tiles, void* stream) {
  return mid_launch(x, x_stride, W, w_stride, nullptr, false, part, M, N, K, ksplit, waves, row_tiles, stream);
}

// The same with e4m3 weights in the tiled order of ops.llm.tile_weights and one fp32 scale per row (W8A16).
int klm_mid_linear_w8(const void* x, int x_stride, const void* W8, const float* wscale, float* part, int M, int N, int K,
                      int ksplit, int waves, int row_tiles, void* stream) {
  return mid_launch(x, x_stride, W8, K, wscale, true, part, M, N, K, ksplit, waves, row_tiles, stream);
}

// gate_up with ksplit == 1: SwiGLU applied in the kernel, act[M][N/2] bf16 written directly (no partials, no epilogue).
// W is bf16 (wscale == nullptr) or tiled e4m3 with row scales.
int klm_mid_linear_swiglu(const void* x, int x_stride, const void* W, int w_stride, const float* wscale, void* act, int M, int N,
                          int K, int waves, int row_tiles, void* stream) {
  if (!aligned16(act)) return KLM_ERR_ALIGN;
  return mid_launch(x, x_stride, W, wscale ? K : w_stride, wscale, wscale != nullptr, static_cast<float*>(act), M, N, K, 1, waves,
                    row_tiles, stream, true);
}

// kind 0: out[M][N] = sum of the splits. 1: SwiGLU, out[M][N/2]. 2: residual[M][N] += sum, out[M][N] = rmsnorm(residual).
int klm_mid_epilogue(const float* part, int ksplit, int M, int N, int kind, void* out, void* residual, float eps, void* stream) {
  if (M <= 0 || N <= 0 || ksplit <= 0 || N % 16 != 0) return KLM_ERR_SHAPE;
  if (kind < 0 || kind > 2) return KLM_ERR_VARIANT;
  if (!part || !out || ((kind == MID_RES_RMS) != (residual != nullptr))) return KLM_ERR_ARG;
  if (kind == MID_RES_RMS && N > 16384) return KLM_ERR_SHAPE;
  if (!aligned16(part) || !aligned16(out) || !aligned16(residual)) return KLM_ERR_ALIGN;
  auto s = static_cast<hipStream_t>(stream);
  auto oo = static_cast<uint16_t*>(out);
  if (kind == MID_RES_RMS) {
    hipLaunchKernelGGL(mid_res_rms_kernel, dim3(M), dim3(256), 0, s, part, ksplit, M, N, static_cast<uint16_t*>(residual), oo, eps);
  } else {
    const size_t total = (size_t)M * N / (kind == MID_SWIGLU ? 16 : 8);
    const int blocks = (int)((total + 255) / 256);
    if (kind == MID_SWIGLU)
      hipLaunchKernelGGL(mid_epilogue_kernel<MID_SWIGLU>, dim3(blocks), dim3(256), 0, s, part, ksplit, M, N, oo);
    else
      hipLaunchKernelGGL(mid_epilogue_kernel<MID_NONE>, dim3(blocks), dim3(256), 0, s, part, ksplit, M, N, oo);
  }
  return launch_status();
}

}  // extern "C"
