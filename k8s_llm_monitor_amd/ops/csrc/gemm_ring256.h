// The 256 x 256 LDS-DMA ring pipeline shared by the bf16, fp8 and fp4 GEMM burner kernels (gfx950, CDNA4).
//
// No reference counterpart (the reference has no native code).
//
// One workgroup = 8 waves (2 x 4) = one 256 x 256 output tile; a K tile is 128 BYTES per operand row in every
// kernel (64 bf16, 128 fp8, 256 fp4), so staging, slots, swizzle and barriers are byte-identical and live here;
// a kernel supplies two policies:
//   STAGE  issue_a<H>(stage, tile) / issue_b<H>(stage, tile): two LDS-DMA instructions per wave that bring
//          half-tile H of K tile `tile` into its slot of ring stage `stage`;
//   OP     read_a(slot), read_b<SET>(slot): fragment reads (ds_read_b128) into operand registers;
//          mfma<AH, BH, SET>(): one quadrant's MFMA cluster, acc[AH][BH] += a x b[SET].
//
// LDS: 128 KiB = 2 stages x 4 half-tile slots of 16 KiB:  A0 (+0)  B0 (+16 K)  B1 (+32 K)  A1 (+48 K).
//   A-half h = rows {wr*128 + h*64 + r}, r < 64   (slot row = wr*64 + r)
//   B-half h = cols {wc*64  + h*32 + r}, r < 32   (slot row = wc*32 + r)
// so that every wave needs the half-tiles in the same order. A wave computes its 128 x 64 output as four
// quadrants, one per phase, in the order (a0,b0) (a0,b1) (a1,b1) (a1,b0): each phase needs ONE new operand half.
// The two wave groups (waves 0-3 / 4-7 = the two waves of each SIMD) run one barrier apart, so one group reads
// LDS and issues DMA while its SIMD partner issues MFMAs. Images are lane-linear (what LDS-DMA writes); the
// bank-conflict swizzle (16-byte chunk index XOR row & 7 inside each 128-byte row) is applied to the per-lane
// SOURCE address by STAGE and again on the ds_read address by OP.
//
// Two schedules:
//   SCHED 0  per K tile: phase 0 reads A0 and B0 (12 ds_read_b128) and issues 4 DMA, phase 1 reads B1 / issues 2,
//            phase 2 reads A1 / issues 2, phase 3 reads and issues nothing (round 1's kernel).
//   SCHED 1  "one half-tile per phase": half-tiles are numbered in need order h(4t..4t+3) = B0 A0 B1 A1 of tile t;
//            phase q reads h(q+1), issues h(q+5) and waits (counted) for h(q+2): every phase carries 1 half-tile
//            of reads (8 / 4 / 8 / 4 ds_read_b128), 2 DMA instructions and one `s_waitcnt vmcnt(6)`, and every
//            half-tile gets the same three phases of flight time. B0 of the NEXT tile is read in phase 3 into the
//            operand registers the (a1,b1) quadrant has just released, so the b0 / b1 register sets swap roles
//            every tile (the loop is unrolled by two tiles; no extra registers). Needs an even number of K tiles >= 4.
// Measured and dropped (docs/KERNELS.md): two phases of 32 MFMAs per K tile (half the barriers): 0.98x of schedule 1.
//
// Hazards (same rules for both schedules; cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait
// that retires it"): a half-tile is read one phase after the phase whose counted wait (placed before that phase's
// first barrier) retires it, and a slot is rewritten four phases after its last read.
#pragma once
#include "klm_launch.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "klm_ops.h"

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define KLM_AS1 __attribute__((address_space(1)))
#define KLM_AS3 __attribute__((address_space(3)))

// One wave-instruction: 64 lanes x 16 B from per-lane global addresses into 1 KiB of LDS starting at the
// wave-uniform address `lds` (+ lane*16, added by the hardware).
__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const KLM_AS1 void*)g, (KLM_AS3 void*)lds, 16, 0, 0);
}

#define KLM_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define KLM_BAR() __builtin_amdgcn_s_barrier()

#define KLM_SLOT_A0 0
#define KLM_SLOT_B0 16384
#define KLM_SLOT_B1 32768
#define KLM_SLOT_A1 49152
#define KLM_STAGE_BYTES 65536
#define KLM_RING_LDS_BYTES 131072

// Block id -> output tile. (1) XCD-aware: blocks b and b+8 share an XCD (round-robin dispatch), so give each of
// the 8 residue classes one contiguous chunk of tile ids (bijective for any grid size). (2) Inside the chunk walk
// tiles in groups of GM tile rows, column by column, so that the blocks resident on an XCD at one time form a
// compact patch that shares A and B panels through that XCD's L2.
template <int GM>
__device__ __forceinline__ void tile_of_block(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int per_group = GM * tiles_n;
  const int g = id / per_group;
  const int first_m = g * GM;
  const int gsize = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int in_g = id - g * per_group;
  tm = first_m + in_g % gsize;
  tn = in_g / gsize;
}

__device__ __forceinline__ void store_c4(uint16_t* C, size_t idx, const f32x4& v) {
  bf16x4 o;
  o[0] = (__bf16)v[0];
  o[1] = (__bf16)v[1];
  o[2] = (__bf16)v[2];
  o[3] = (__bf16)v[3];
  *reinterpret_cast<bf16x4*>(C + idx) = o;
}

// One wave-instruction of the same kind through a buffer descriptor: `buffer_load_dwordx4 v, s[rsrc], s_off offen lds`.
// The per-lane part of the address is ONE 32-bit VGPR that never changes; everything that moves (K tile, half,
// 8-row piece, the wave's rows) is scalar arithmetic on `soff`, so a DMA costs no vector instruction and no
// 64-bit address registers (global_load_lds needed a v_lshl_add_u64 or two and a VGPR pair per instruction).
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (KLM_AS3 void*)lds, 16, (int)voff, (int)soff, 0, 0);
}

// Dense row-major operands. The descriptors start at the tile's first A row / B row (so 32-bit offsets hold for any
// matrix size) and cover its 256 rows; `va` / `vb` = this lane's (row lane>>3, swizzled chunk) offset inside an 8-row
// piece; `sa` / `sb` = this wave's first slot row, in bytes. Strides are in BYTES, a K tile is 128 bytes.
struct klm_stage_dense {
  __amdgpu_buffer_rsrc_t ra, rb;
  unsigned va, vb, sa, sb;
  unsigned lda, ldb;   // row strides, bytes
  char* lw;            // this wave's 2 KiB window inside a slot

  // A, B: the operands' first byte; brow / bcol: the tile's first row of A / B; lda / ldb in bytes
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
    sa = (unsigned)((wid >> 2) * 128 + (wid & 3) * 16) * lda;   // wave `wid` fills slot rows [wid*16, wid*16+16)
    sb = (unsigned)((wid >> 1) * 64 + (wid & 1) * 16) * ldb;
    lw = smem + wid * 2048;
  }
  template <int H>
  __device__ __forceinline__ void issue_a(int stage, int tile) const {
    const unsigned so = sa + (unsigned)H * 64 * lda + (unsigned)tile * 128;
    char* l = lw + stage * KLM_STAGE_BYTES + (H ? KLM_SLOT_A1 : KLM_SLOT_A0);
    blds16(ra, va, so, l);
    blds16(ra, va, so + 8 * lda, l + 1024);
  }
  template <int H>
  __device__ __forceinline__ void issue_b(int stage, int tile) const {
    const unsigned so = sb + (unsigned)H * 32 * ldb + (unsigned)tile * 128;
    char* l = lw + stage * KLM_STAGE_BYTES + (H ? KLM_SLOT_B1 : KLM_SLOT_B0);
    blds16(rb, vb, so, l);
    blds16(rb, vb, so + 8 * ldb, l + 1024);
  }
};

// The main loop. `nt` = K tiles (>= 2; SCHED 1: even and >= 4), `wr` = the wave's row group (0 / 1).
template <int SCHED, class OP, class STAGE>
__device__ __forceinline__ void klm_ring256_mainloop(OP& op, const STAGE& st, const char* smem, int nt, int wr) {
  if constexpr (SCHED == 0) {
    // ---- prologue: K tile 0 into stage 0 in need order A0,B0 | B1 | A1
    st.template issue_a<0>(0, 0);
    st.template issue_b<0>(0, 0);
    st.template issue_b<1>(0, 0);
    st.template issue_a<1>(0, 0);
    KLM_VMCNT(4);
    KLM_BAR();
    if (wr == 1) KLM_BAR();  // stagger: group 1 runs one barrier behind group 0
    for (int t = 0; t < nt - 1; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      const char* sb = smem + cur * KLM_STAGE_BYTES;
      // phase 0
      op.read_a(sb + KLM_SLOT_A0);
      op.template read_b<0>(sb + KLM_SLOT_B0);
      st.template issue_a<0>(nxt, t + 1);
      st.template issue_b<0>(nxt, t + 1);
      KLM_VMCNT(6);
      KLM_BAR();
      op.template mfma<0, 0, 0>();
      KLM_BAR();
      // phase 1
      op.template read_b<1>(sb + KLM_SLOT_B1);
      st.template issue_b<1>(nxt, t + 1);
      KLM_VMCNT(6);
      KLM_BAR();
      op.template mfma<0, 1, 1>();
      KLM_BAR();
      // phase 2
      op.read_a(sb + KLM_SLOT_A1);
      st.template issue_a<1>(nxt, t + 1);
      KLM_BAR();
      op.template mfma<1, 1, 1>();
      KLM_BAR();
      // phase 3
      KLM_VMCNT(4);
      KLM_BAR();
      op.template mfma<1, 0, 0>();
      KLM_BAR();
    }
    {  // last K tile: nothing left to issue
      const char* sb = smem + ((nt - 1) & 1) * KLM_STAGE_BYTES;
      op.read_a(sb + KLM_SLOT_A0);
      op.template read_b<0>(sb + KLM_SLOT_B0);
      KLM_VMCNT(2);
      KLM_BAR();
      op.template mfma<0, 0, 0>();
      KLM_BAR();
      op.template read_b<1>(sb + KLM_SLOT_B1);
      KLM_VMCNT(0);
      KLM_BAR();
      op.template mfma<0, 1, 1>();
      KLM_BAR();
      op.read_a(sb + KLM_SLOT_A1);
      KLM_BAR();
      op.template mfma<1, 1, 1>();
      KLM_BAR();
      KLM_BAR();
      op.template mfma<1, 0, 0>();
      KLM_BAR();
    }
    if (wr == 0) KLM_BAR();  // matches group 1's extra barrier in the prologue
  } else if constexpr (SCHED == 1) {
    const char* const s0 = smem;
    const char* const s1 = smem + KLM_STAGE_BYTES;
    // ---- prologue: h0..h4 = B0(0) A0(0) B1(0) A1(0) B0(1); B0(0) goes to register set 0 ("phase -1" read)
    st.template issue_b<0>(0, 0);
    st.template issue_a<0>(0, 0);
    st.template issue_b<1>(0, 0);
    st.template issue_a<1>(0, 0);
    st.template issue_b<0>(1, 1);
    KLM_VMCNT(6);            // h0, h1 landed
    KLM_BAR();
    if (wr == 1) KLM_BAR();
    op.template read_b<0>(s0 + KLM_SLOT_B0);

    // One K tile living in stage S with b0 in register set X (b1 in the other); ISSUE: 1 = all four phases issue
    // for tiles t+1 / t+2, 2 = the last-but-one tile (its phase 3 has nothing left to issue), 0 = the last tile.
#define KLM_TILE(S, X, ISSUE, t)                                                              \
    do {                                                                                      \
      const char* sc_ = (S) ? s1 : s0;                                                        \
      const char* sn_ = (S) ? s0 : s1;                                                        \
      /* phase 0: read A0(t), issue A0(t+1), wait B1(t) */                                    \
      op.read_a(sc_ + KLM_SLOT_A0);                                                           \
      if (ISSUE) { st.template issue_a<0>((S) ^ 1, (t) + 1); KLM_VMCNT(6); } else { KLM_VMCNT(2); } \
      KLM_BAR();                                                                              \
      op.template mfma<0, 0, (X)>();                                                          \
      KLM_BAR();                                                                              \
      /* phase 1: read B1(t), issue B1(t+1), wait A1(t) */                                    \
      op.template read_b<(X) ^ 1>(sc_ + KLM_SLOT_B1);                                         \
      if (ISSUE) { st.template issue_b<1>((S) ^ 1, (t) + 1); KLM_VMCNT(6); } else { KLM_VMCNT(0); } \
      KLM_BAR();                                                                              \
      op.template mfma<0, 1, (X) ^ 1>();                                                      \
      KLM_BAR();                                                                              \
      /* phase 2: read A1(t), issue A1(t+1), wait B0(t+1) */                                  \
      op.read_a(sc_ + KLM_SLOT_A1);                                                           \
      if (ISSUE) { st.template issue_a<1>((S) ^ 1, (t) + 1); KLM_VMCNT(6); }                  \
      KLM_BAR();                                                                              \
      op.template mfma<1, 1, (X) ^ 1>();                                                      \
      KLM_BAR();                                                                              \
      /* phase 3: read B0(t+1) into the set (a1,b1) released, issue B0(t+2), wait A0(t+1) */  \
      if (ISSUE) op.template read_b<(X) ^ 1>(sn_ + KLM_SLOT_B0);                              \
      if ((ISSUE) == 1) { st.template issue_b<0>((S), (t) + 2); KLM_VMCNT(6); }               \
      else if ((ISSUE) == 2) { KLM_VMCNT(4); }                                                \
      KLM_BAR();                                                                              \
      op.template mfma<1, 0, (X)>();                                                          \
      KLM_BAR();                                                                              \
    } while (0)

    int t = 0;
    for (; t < nt - 2; t += 2) {
      KLM_TILE(0, 0, 1, t);
      KLM_TILE(1, 1, 1, t + 1);
    }
    KLM_TILE(0, 0, 2, t);        // tile nt-2
    KLM_TILE(1, 1, 0, t + 1);    // tile nt-1
#undef KLM_TILE
    if (wr == 0) KLM_BAR();
  }
}

// A stream of another device would take the launch to a GPU whose attribute was never set (and whose pointers the
// caller did not mean): refuse instead of failing inside the runtime. The null stream is the current device's.
static inline bool klm_stream_on_current_device(hipStream_t s) {
  if (!s) return true;
  int cur = -1, sd = -1;
  if (hipGetDevice(&cur) != hipSuccess) return false;
  if (hipStreamGetDevice(s, &sd) != hipSuccess) return true;   // cannot tell: let the launch decide
  return cur == sd;
}
