#!/usr/bin/env python3
"""Generates ops/csrc/gemm_quad_asm.inc: the hand-scheduled gfx950 instruction streams of the 4-wave 256 x 256 GEMM kernels
(`gemm_quad_asm<BODY>` in ops/csrc/gemm_quad_asm.h; bf16, fp4 and fp8) as inline-asm bodies.

Why generated text and not HIP source: with one wave per SIMD there is no partner wave to hide anything behind, so the
order of the wave's own instruction stream IS the schedule. Written as HIP (gemm_quad128.h: builtins or per-MFMA asm
statements pinned with sched_barrier) the compiler still decides the waits and the interleave, and that kernel reached
0.75 MFMA-busy; the vendor's hand-written kernel of the same geometry reaches 0.86 - 0.88. Here every instruction's place
is fixed. The shipped bf16 / fp4 schedule is "chain" (documented at tile_chain below: a tile's two k-steps back to back on its
accumulator); the one before it, "stagger", which "chain" took its DMA placement from (see SCHEDULES for the others and
docs/KERNELS.md for what each measured):

  per K tile (stage S, 128 MFMAs; which accumulator tile MFMA i works on: MFMA_ORDER / tile_of_step below), gap = "after MFMA i":
    0, 2, .., 30       the 16 ds_read_b128 of this tile's k-step 1 -> fragment set 1 (A[0], B[0..7], A[1..7])
    36                 s_waitcnt lgkmcnt(0); s_barrier      every wave is done reading stage S
    39 + 5 p + w       LDS-DMA piece p (0..7 A, 8..15 B) of tile t+2 -> stage S, for wave w: one per FIVE MFMAs (closer
                       costs up to 13 %; further apart gains nothing), one gap later per wave (the lockstep waves do not
                       hand their DMA to the texture-address unit in the same cycle)
    85                 s_waitcnt vmcnt(#pieces issued so far); s_barrier     tile t+1 landed, for every wave
    86, 88, ..         the 16 fragments of tile t+1's k-step 0 -> set 0, skipping the DMA gaps
    126                s_waitcnt lgkmcnt(0)
  An M0 write sits one gap before the DMA that uses it (SALU write of M0 -> LDS-DMA needs a wait state).
  Epilogue: fp32 -> bf16, v_permlane16_swap_b32 pairs n tiles so every store is 16 bytes per lane (epilogue_lines).
  fp8 has its own register plan and tile (tile_fp8); fp8 and fp4 use the plain v_mfma_f32_16x16x128_f8f6f4 (MX_SCALED).
  Bodies come in two shapes: one output tile per workgroup (body, body_fp8) and the persistent form (body_p: one workgroup
  per CU walks its tiles as one stream of K tiles; the default kernels). BODIES lists every macro the .inc defines,
  experiments and diagnostic streams included.

Registers (fixed, all on the asm statement's clobber list):
  a[0:255]        accumulators, tile (mi, ni) at a[4 (8 mi + ni)]
  v[40:71]        A fragments set 0 (k-step 0), v[72:103] B set 0, v[104:135] A set 1, v[136:167] B set 1   (bf16 / fp4)
  v[168:175]      ds_read bases: stage 0 {a k0, a k1, b k0, b k1}, stage 1 the same + 64 KiB; v176 fp4 unit scales
  s[64:67] A descriptor, s[68:71] B, s[72:75] C; s[81:87] piece offsets p * 8 * lda; s88 loop count;
  s[90:93] LDS-DMA bases {A stage 0, B stage 0, A stage 1, B stage 1}; s[94:95] scratch
"""
import os
import sys

A0, B0, A1, B1 = 40, 72, 104, 136
RB = 168


def acc(mi, ni):
    b = 4 * (8 * mi + ni)
    return f"a[{b}:{b + 3}]"


def frag(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


NMAJOR = False      # True: consecutive MFMAs share the B fragment (the instruction's FIRST source) instead of the A fragment
DTYPE = "bf16"      # or "fp4": same 16-byte fragments and K tile of 128 bytes per row, the f8f6f4 MFMA
SCALE = 176         # fp4: VGPR holding four E8M0 scales of 2^0
# fp8 / fp4: the plain v_mfma_f32_16x16x128_f8f6f4 (one instruction word; scales are 2^0 by definition) -- or, True, the
# block-scaled v_mfma_scale_* with unit scales in v[SCALE] (two words: same bits, fp8 -1.6 %, fp4 -3.3 %; scripts/exp/mx_noscale.py)
MX_SCALED = False
STORE_CACHE = ""    # cache-policy bits on the epilogue's stores (" nt": streaming)
DMA_CACHE = ""      # cache-policy bits on the LDS-DMA (" sc0", " sc1", " nt", ..): experiment bodies only


# The ORDER of a k-step's 64 MFMAs over the 8 x 8 accumulator tiles. On a card that sits on its power cap the rate is energy per
# FLOP, and what an MFMA costs depends on which of its sources the one before it used (scripts/exp/gen_mfma_order_exp.py,
# register-only loops on the cap: tile-row-major 2100 TFLOP/s, the same as a snake 2135, no shared source 2035, and 2182 when
# every step changes exactly ONE source and the two sources take turns):
#   "rows"    m-tile-major: 8 consecutive MFMAs share the A fragment, at a row change both sources change (rounds 2 - 3)
#   "zigzag"  two tile rows at a time: (r0,n) (r1,n) (r1,n') (r0,n') (r0,n'') .., the next pair of rows walks back; every step
#             changes one source, never both
MFMA_ORDER = os.environ.get("KLM_GEN_MFMA_ORDER", "zigzag")


def tile_of_step(j, order=None):
    """(m tile, n tile) of MFMA j < 64 of a k-step."""
    order = order or MFMA_ORDER
    if order == "rows":
        return j // 8, j % 8
    p, q = j >> 4, j & 15
    col, k = q >> 1, q & 1
    n = 7 - col if p & 1 else col
    r = k ^ (col & 1) ^ (p & 1)
    return 2 * p + r, n


def mfma(i, czero=False):
    """MFMA i of a K tile: k-step i // 64, accumulator tile by MFMA_ORDER (NMAJOR: the roles of m and n swapped).
    czero: an output tile's first K tile -- k-step 0 adds to the constant 0 instead of the accumulator (no clearing pass)."""
    ks = i // 64
    mi, ni = tile_of_step(i % 64)
    if NMAJOR:
        mi, ni = ni, mi
    a, b = (A0, B0) if ks == 0 else (A1, B1)
    c = "0" if (czero and ks == 0) else acc(mi, ni)
    if DTYPE == "fp4":
        if not MX_SCALED:
            return f"v_mfma_f32_16x16x128_f8f6f4 {acc(mi, ni)}, {frag(b, ni)}, {frag(a, mi)}, {c} cbsz:4 blgp:4"
        return (f"v_mfma_scale_f32_16x16x128_f8f6f4 {acc(mi, ni)}, {frag(b, ni)}, {frag(a, mi)}, {c}, v{SCALE}, v{SCALE} "
                f"op_sel_hi:[0,0,0] cbsz:4 blgp:4")
    return f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {frag(b, ni)}, {frag(a, mi)}, {c}"


def read(setno, stage, j):
    """Fragment read j of 16 in need order A[0], B[0..7], A[1..7]; `stage` = the LDS stage read from."""
    abase, bbase = (A0, B0) if setno == 0 else (A1, B1)
    ra = RB + 4 * stage + setno          # a, k-step = setno
    rb = RB + 4 * stage + 2 + setno
    if NMAJOR:                           # need order B[0], A[0..7], B[1..7]
        abase, bbase, ra, rb = bbase, abase, rb, ra
    if j == 0:
        return f"ds_read_b128 {frag(abase, 0)}, v{ra}"
    if j <= 8:
        return f"ds_read_b128 {frag(bbase, j - 1)}, v{rb} offset:{(j - 1) * 2048}"
    return f"ds_read_b128 {frag(abase, j - 8)}, v{ra} offset:{(j - 8) * 2048}"


def dma(stage, p):
    """LDS-DMA piece p of 16 (0..7 A, 8..15 B) of the NEXT tile to issue -> `stage`, as (m0 update, load [+ descriptor
    bump after the operand's last piece]). The M0 write needs one instruction between it and the DMA that uses it
    (SALU write of M0 -> LDS-DMA: 1 wait state), so the caller puts the two parts in different MFMA gaps."""
    op, q = ("A", p) if p < 8 else ("B", p - 8)
    rs = "s[64:67]" if op == "A" else "s[68:71]"
    pre = [f"s_mov_b32 m0, s{90 + 2 * stage + (0 if op == 'A' else 1)}"] if q == 0 else ["s_add_u32 m0, m0, 0x400"]
    so = "0" if q == 0 else f"s{80 + q}"
    main = [f"buffer_load_dwordx4 %[va], {rs}, {so} offen{DMA_CACHE} lds"]
    if q == 7:
        lo = 64 if op == "A" else 68
        main += [f"s_add_u32 s{lo}, s{lo}, 0x80", f"s_addc_u32 s{lo + 1}, s{lo + 1}, 0"]
    return pre, main


def read_b(setno, stage, j):
    return read(setno, stage, 1 + j)                      # B[j]


def read_a(setno, stage, j):
    return read(setno, stage, 0 if j == 0 else 8 + j)     # A[j]


def tile(stage, mode, sched="one_barrier", ablate=(), wave=0, czero=False):
    """mode: 'full' (issues tile t+2), 'tail2' (tile nt-2: no DMA), 'last' (tile nt-1: no DMA, no next reads).
    sched: where the pieces go (see SCHEDULES). ablate: diagnostic builds that drop 'dma', 'read' or 'barrier'
    instructions from the loop (wrong results, right timing: what each kind of instruction costs)."""
    if sched == "chain":
        assert not ablate
        return tile_chain(stage, mode, wave, czero)
    out = [f"; ---- K tile in stage {stage}, {mode}, {sched}"]
    after = {i: [] for i in range(128)}
    P = SCHEDULES[sched]
    if callable(P):
        P = P(wave)
    for j, i in enumerate(P["b1"]):
        after[i].append(read_b(1, stage, j))
    for j, i in enumerate(P["a1"]):
        after[i].append(read_a(1, stage, j))
    for i in P["free_barriers"]:
        after[i] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    n_before = sum(1 for i in P["dma"] if i <= P["landed"])
    if mode == "full":
        for p, i in zip(P["dma_order"], P["dma"]):
            pre, main = dma(stage, p)
            after[i - 1] += pre          # the gap before: an MFMA sits between the M0 write and its use
            after[i] += main
        after[P["landed"]] += (["s_waitcnt lgkmcnt(0)"] if P.get("landed_lgkm") else []) + [f"s_waitcnt vmcnt({n_before})", "s_barrier"]
    elif mode == "tail2":
        after[P["landed"]] += (["s_waitcnt lgkmcnt(0)"] if P.get("landed_lgkm") else []) + ["s_waitcnt vmcnt(0)", "s_barrier"]
    elif P.get("landed_lgkm"):
        after[P["landed"]] += ["s_waitcnt lgkmcnt(0)"]
    if mode != "last":
        for j, i in enumerate(P["next"]):
            after[i].append(read(0, stage ^ 1, j))
        after[126].append("s_waitcnt lgkmcnt(0)")
    drop = []
    if "dma" in ablate:
        drop += ["buffer_load", "s_mov_b32 m0", "s_add_u32 m0", "s_add_u32 s64", "s_addc_u32 s65", "s_add_u32 s68", "s_addc_u32 s69", "s_waitcnt vmcnt"]
    if "bump" in ablate:       # DMA stays, but every tile re-fetches K tile 0/1 (L2-resident): no HBM / fabric traffic
        drop += ["s_add_u32 s64", "s_addc_u32 s65", "s_add_u32 s68", "s_addc_u32 s69"]
    if "read" in ablate:
        drop += ["ds_read", "s_waitcnt lgkmcnt"]
    if "barrier" in ablate:
        drop += ["s_barrier"]
    for i in range(128):
        out.append(mfma(i, czero))
        out += [x for x in after[i] if not any(x.startswith(d) for d in drop)]
    return out


# ----------------------------------------------------------------------------------------------------------------
# Schedule "chain" (bf16 / fp4, the shipped one since the last session of round 3). Energy again: back-to-back MFMAs on the SAME
# accumulator cost less than anything else measured -- the register-only loops of scripts/exp/gen_mfma_order_exp.py run at 2105
# TFLOP/s in the zigzag order with k-step 0 of all 64 tiles before k-step 1 (schedule "stagger"), at 2170 - 2187 when each tile
# gets its two k-steps back to back, although then BOTH sources change at every step: the accumulator is forwarded instead of
# being read from the register file again. Each accumulator still adds k-step 0 before k-step 1, so the bits do not change.
# Chaining needs both k-steps' fragments of a tile at once, which turns the register plan around:
#   v[40:103]     A fragments, ONE set: (m tile, k-step) at v[40 + 4 (2 m + k)]; a pair of tile rows (the zigzag's unit, 32 MFMAs)
#                 is refilled with the next K tile's fragments as soon as its MFMAs are issued
#   v[104:167] / v[168:231]   B fragments of the K tile in stage 0 / stage 1, (n tile, k-step) at base + 4 (2 n + k): all 16 are
#                 needed by every row pair, so B is double-buffered
#   v[232:239] read bases, v240 unit scales, v241.. the persistent kernels' table registers (= the fp8 kernel's numbers)
# per K tile t in stage S, gap = "after MFMA i" of 128:
#   0 .. 3          A rows 6, 7 of tile t itself (their registers were busy to the end of tile t-1)
#   9               s_waitcnt lgkmcnt(0); s_barrier       every wave has read ALL of stage S (27 gaps earlier than "stagger"; at gap 6,
#                   three MFMAs behind the last read, the wait came before an LDS read returns: 0.2 - 0.7 % slower, chain_free_gap.json)
#   12 + 5 p + w    LDS-DMA piece p of tile t+2 -> stage S, wave w one gap later each (as "stagger"); M0 one gap before
#   56              s_waitcnt vmcnt(pieces issued so far); s_barrier      tile t+1 landed (94 gaps after its last piece)
#   57 ..           B[0..7] x 2 k-steps of tile t+1 -> the other B buffer, then A rows 0 - 1 and 2 - 3 of tile t+1 (their row pairs are
#                   done at 31 / 63), one read per gap, DMA gaps skipped; A rows 4 - 5 from gap 95 on (their pair is done at 95)
#   126             s_waitcnt lgkmcnt(0)
CA = 40
CB = (104, 168)
CRB, CSCALE, TBL_CHAIN = 232, 240, 241
CHAIN_FREE, CHAIN_DMA0, CHAIN_DMA_STEP, CHAIN_LANDED = (int(x) for x in os.environ.get("KLM_GEN_CHAIN", "9,12,5,56").split(","))   # env: experiments


CHAIN_SLEEP = int(os.environ.get("KLM_GEN_CHAIN_SLEEP", "0"))      # experiments only: s_sleep N (64 N cycles) at gap 110 of every K tile


def cfrag_a(m, k):
    b = CA + 4 * (2 * m + k)
    return f"v[{b}:{b + 3}]"


def cfrag_b(stage, n, k):
    b = CB[stage] + 4 * (2 * n + k)
    return f"v[{b}:{b + 3}]"


def cread(operand, idx, k, stage, buf=None):
    """ds_read_b128 of fragment (operand, tile idx, k-step k) from LDS stage `stage`; B goes to buffer `buf` (default: its stage's)."""
    if operand == "a":
        return f"ds_read_b128 {cfrag_a(idx, k)}, v{CRB + 4 * stage + k}" + (f" offset:{idx * 2048}" if idx else "")
    return f"ds_read_b128 {cfrag_b(stage if buf is None else buf, idx, k)}, v{CRB + 4 * stage + 2 + k}" + (f" offset:{idx * 2048}" if idx else "")


def mfma_chain(i, stage, czero=False):
    """MFMA i of 128: tile tile_of_step(i // 2, "zigzag"), k-step i % 2 -- a tile's two k-steps back to back."""
    mi, ni = tile_of_step(i // 2, "zigzag")
    ks = i % 2
    c = "0" if (czero and ks == 0) else acc(mi, ni)
    if DTYPE == "fp4":
        if not MX_SCALED:
            return f"v_mfma_f32_16x16x128_f8f6f4 {acc(mi, ni)}, {cfrag_b(stage, ni, ks)}, {cfrag_a(mi, ks)}, {c} cbsz:4 blgp:4"
        return (f"v_mfma_scale_f32_16x16x128_f8f6f4 {acc(mi, ni)}, {cfrag_b(stage, ni, ks)}, {cfrag_a(mi, ks)}, {c}, v{CSCALE}, v{CSCALE} "
                f"op_sel_hi:[0,0,0] cbsz:4 blgp:4")
    return f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {cfrag_b(stage, ni, ks)}, {cfrag_a(mi, ks)}, {c}"


def chain_first_reads(stage=0):
    """What a K tile needs in registers before its first MFMA: all of B and A rows 0 .. 5 (rows 6, 7 are read by the tile itself)."""
    return [cread("b", n, k, stage) for n in range(8) for k in (0, 1)] + [cread("a", m, k, stage) for m in range(6) for k in (0, 1)]


def tile_chain(stage, mode, wave=0, czero=False):
    out = [f"; ---- K tile in stage {stage}, {mode}, chain, wave {wave}"]
    after = {i: [] for i in range(128)}
    for j, (m, k) in enumerate(((6, 0), (6, 1), (7, 0), (7, 1))):
        after[j].append(cread("a", m, k, stage))
    after[CHAIN_FREE] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    dma_gaps = [CHAIN_DMA0 + CHAIN_DMA_STEP * p + wave for p in range(16)]
    if mode == "full":
        for p, i in enumerate(dma_gaps):
            pre, main = dma(stage, p)
            after[i - 1] += pre
            after[i] += main
        n_before = sum(1 for i in dma_gaps if i <= CHAIN_LANDED)
        after[CHAIN_LANDED] += [f"s_waitcnt vmcnt({n_before})", "s_barrier"]
    elif mode == "tail2":
        after[CHAIN_LANDED] += ["s_waitcnt vmcnt(0)", "s_barrier"]
    if mode != "last":
        nxt = stage ^ 1
        reads = [(cread("b", n, k, nxt), 0) for n in range(8) for k in (0, 1)]
        # A rows of pair p may be overwritten once MFMA 32 p + 31 is issued
        reads += [(cread("a", m, k, nxt), 32 * (m // 2) + 31) for m in range(6) for k in (0, 1)]
        g = CHAIN_LANDED + 1
        busy = set(dma_gaps) if mode == "full" else set()
        for text, not_before in reads:
            g = max(g, not_before)
            while g in busy:
                g += 1
            after[g].append(text)
            g += 1
        assert g <= 126, g
        after[126].append("s_waitcnt lgkmcnt(0)")
    if CHAIN_SLEEP:
        # EXPERIMENT (scripts/exp/idle_cycles_on_the_cap.py): idle cycles in every K tile -- on a power-capped card, does the clock pay them back?
        after[110].append(f"s_sleep {CHAIN_SLEEP}")
    for i in range(128):
        out.append(mfma_chain(i, stage, czero))
        out += after[i]
    return out


# gap (= "after MFMA i") of every loop instruction, per schedule. dma_order: pieces 0..7 = A, 8..15 = B.
SCHEDULES = {
    # one barrier frees the whole stage; DMA from 41 on; the wait for tile t+1 at 63
    "one_barrier": {"b1": [2 * (1 + j) for j in range(8)], "a1": [0] + [2 * (9 + j) for j in range(7)], "free_barriers": [40],
                    "dma_order": list(range(16)), "dma": [41 + 2 * p for p in range(12)] + [65, 67, 69, 71], "landed": 63,
                    "next": [64 + 2 * j for j in range(16)]},
    # an operand's half of stage S is refilled as soon as every wave has read that operand's k-step 1 fragments
    # (one barrier each), the wait for tile t+1 comes late (its DMA gets 160 - 200 MFMA slots to land)
    "per_operand": {"b1": [2 * j for j in range(8)], "a1": [22 + 2 * j for j in range(8)], "free_barriers": [20, 46],
                    "dma_order": [8 + q for q in range(8)] + list(range(8)), "dma": [21 + 2 * q for q in range(8)] + [47 + 2 * q for q in range(8)],
                    "landed": 87, "next": [88 + 2 * j for j in range(16)]},
    # as one_barrier, but the 16 DMA pieces spread over the rest of the tile, one per five MFMAs
    "spread": {"b1": [2 * (1 + j) for j in range(8)], "a1": [0] + [2 * (9 + j) for j in range(7)], "free_barriers": [36],
               "dma_order": list(range(16)), "dma": [39 + 5 * p for p in range(16)], "landed": 85,
               "next": [86 + 2 * j + (1 if (86 + 2 * j - 39) % 5 == 0 else 0) for j in range(16)]},
}


def setup_lines(RB, SCALE):
    o = []
    o += ["; ==== setup", "s_mov_b32 s64, %[ra0]", "s_mov_b32 s65, %[ra1]", "s_mov_b32 s66, %[ra2]", "s_mov_b32 s67, %[ra3]",
          "s_mov_b32 s68, %[rb0]", "s_mov_b32 s69, %[rb1]", "s_mov_b32 s70, %[rb2]", "s_mov_b32 s71, %[rb3]",
          "s_mov_b32 s72, %[rc0]", "s_mov_b32 s73, %[rc1]", "s_mov_b32 s74, %[rc2]", "s_mov_b32 s75, %[rc3]",
          "s_lshl_b32 s81, %[lda], 3", "s_lshl_b32 s82, %[lda], 4", "s_add_u32 s83, s82, s81", "s_lshl_b32 s84, %[lda], 5",
          "s_add_u32 s85, s84, s81", "s_add_u32 s86, s84, s82", "s_add_u32 s87, s86, s81",
          "s_mov_b32 s88, %[passes]",
          "s_mov_b32 s90, %[lw]", "s_add_u32 s91, s90, 0x8000", "s_add_u32 s92, s90, 0x10000", "s_add_u32 s93, s90, 0x18000",
          f"v_mov_b32 v{RB}, %[a0]", f"v_mov_b32 v{RB + 1}, %[a1]", f"v_mov_b32 v{RB + 2}, %[b0]", f"v_mov_b32 v{RB + 3}, %[b1]",
          f"v_add_u32 v{RB + 4}, 0x10000, v{RB}", f"v_add_u32 v{RB + 5}, 0x10000, v{RB + 1}",
          f"v_add_u32 v{RB + 6}, 0x10000, v{RB + 2}", f"v_add_u32 v{RB + 7}, 0x10000, v{RB + 3}",
          f"v_mov_b32 v{SCALE}, 0x7f7f7f7f"]
    return o


def epilogue_lines(tmp=A0):
    """(`tmp`: first of 64 scratch registers.)
    fp32 -> bf16 and out, 16 bytes per lane and PAIR of n tiles. A lane holds 4 consecutive columns of one row per tile
    (8 bytes); v_permlane16_swap_b32 exchanges the odd 16-lane rows of tile ni's registers with the even rows of tile
    ni+1's, after which lanes with (lane >> 4) even hold 8 consecutive columns of tile ni and the odd ones of tile ni+1:
    one buffer_store_dwordx4 per pair, 64 contiguous bytes per output row and instruction. The store tail is bound by
    the NUMBER of store instructions (64 x dwordx2 per lane took ~10 us per round of output tiles: a fifth of an
    8192 x 8192 x 1024 GEMM), so halving it is worth more than the extra VALU."""
    o = ["; ==== epilogue", "s_nop 15", "s_nop 15", "s_mov_b32 s94, 0", "s_lshl_b32 s95, %[ldc], 4"]
    t = 0
    for mi in range(8):
        for ni in range(0, 8, 2):
            r = tmp + 8 * (t % 8)
            x, y = 4 * (8 * mi + ni), 4 * (8 * mi + ni + 1)
            o += [f"v_accvgpr_read_b32 v{r + k}, a{x + k}" for k in range(4)]
            o += [f"v_accvgpr_read_b32 v{r + 4 + k}, a{y + k}" for k in range(4)]
            o += [f"v_cvt_pk_bf16_f32 v{r}, v{r}, v{r + 1}", f"v_cvt_pk_bf16_f32 v{r + 1}, v{r + 2}, v{r + 3}",
                  f"v_cvt_pk_bf16_f32 v{r + 2}, v{r + 4}, v{r + 5}", f"v_cvt_pk_bf16_f32 v{r + 3}, v{r + 6}, v{r + 7}",
                  "s_nop 1",                                   # VALU write -> permlane swap read: 2 wait states
                  f"v_permlane16_swap_b32 v{r}, v{r + 2}", f"v_permlane16_swap_b32 v{r + 1}, v{r + 3}",
                  f"buffer_store_dwordx4 v[{r}:{r + 3}], %[vc], s[72:75], s94 offen offset:{ni * 32}{STORE_CACHE}"]
            t += 1
        o.append("s_add_u32 s94, s94, s95")
    # no wait for the stores: the wave may end with them in flight (the memory system owns them), and the CU gets its LDS
    # back for the next workgroup that much earlier
    return o


def body(sched="one_barrier", ablate=(), nmajor=False, dtype="bf16"):
    global NMAJOR, DTYPE, DMA_CACHE
    NMAJOR, DTYPE = nmajor, dtype
    DMA_CACHE = "".join(" " + x[6:] for x in ablate if x.startswith("cache:"))
    chain = sched == "chain"
    o = setup_lines(CRB, CSCALE) if chain else setup_lines(RB, SCALE)
    o += ["; ==== prologue: tiles 0 and 1 on their way, accumulators cleared meanwhile"]
    for st in (0, 1):
        for p in range(16):
            pre, main = dma(st, p)
            o += pre + ["s_nop 0"] + main
    for i in range(256):
        o.append(f"v_accvgpr_write_b32 a{i}, 0")
    o += ["s_waitcnt vmcnt(16)", "s_barrier"]
    if chain:
        o += chain_first_reads(0)
    else:
        for j in range(16):
            o.append(read(0, 0, j))
    o += ["s_waitcnt lgkmcnt(0)"]
    if sched in PER_WAVE:
        # four copies of the loop, one per wave of the workgroup (they differ only in where the DMA pieces sit)
        for w in (1, 2, 3):
            o += [f"s_cmp_eq_u32 %[wid], {w}", f"s_cbranch_scc1 Lklm%=_w{w}"]
        for w in range(4):
            o += [f"; ==== main loop of wave {w}: two K tiles per pass", f"Lklm%=_w{w}:"]
            o += tile(0, "full", sched, ablate, w)
            o += tile(1, "full", sched, ablate, w)
            o += ["s_sub_u32 s88, s88, 1", "s_cmp_lg_u32 s88, 0", f"s_cbranch_scc1 Lklm%=_w{w}"]
            o += tile(0, "tail2", sched, ablate, w)
            o += tile(1, "last", sched, ablate, w)
            if w < 3:
                o += ["s_branch Lklm%=_epilogue"]
        o += ["Lklm%=_epilogue:"]
    else:
        o += ["; ==== main loop: two K tiles per pass", "1:"]
        o += tile(0, "full", sched, ablate)
        o += tile(1, "full", sched, ablate)
        o += ["s_sub_u32 s88, s88, 1", "s_cmp_lg_u32 s88, 0", "s_cbranch_scc1 1b"]
        o += tile(0, "tail2", sched, ablate)
        o += tile(1, "last", sched, ablate)
    o += epilogue_lines()
    return o


# ----------------------------------------------------------------------------------------------------------------
# Persistent kernels: one workgroup per CU walks ITS list of UNITS and treats their K tiles as ONE stream. A unit is a
# K range of one output tile: a whole tile (tile b, b + grid, ..: the same tile -> XCD / round map as one workgroup per tile
# under in-order dispatch), or -- for the last, partial round of a tile count that is no multiple of the CU count -- one K
# slice of a tile, whose fp32 partial sums go to a workspace that a small kernel adds up afterwards (stream-K style: 2.25
# rounds cost 2.25 rounds, not 3), or one of two segments of a tile whose K loop starts in the middle (row pitches the
# memory channels camp on: neighbouring workgroups then read different K offsets at any one time).
# The LDS-DMA of the last pass of a unit fetches K tiles 0 and 1 of the NEXT unit (the descriptors are switched just before
# that pass), and its fragment reads load the next unit's first fragments, so the epilogue -- accumulators out, 32 MB of
# stores per round on the chip -- runs with the next operands already in LDS and registers instead of in front of a cold
# prologue whose loads queue behind those stores; workgroup launch, argument loads and set-up are paid once.
#   table (LDS, written by the HIP part): per unit {A rows lo, hi, B rows lo, hi, C lo, hi, passes, flags} (32 bytes)
#   flags: bit F_ACC the unit adds to the accumulators as they are (it is not its tile's first unit); bit F_NOEPI no
#   epilogue (the tile's sum continues in the next unit); bit F_PARTIAL the epilogue stores the fp32 accumulators to the
#   workspace slot "C" points at (register-major: [wave][acc tile][lane][16 bytes], every store 1 KiB contiguous)
#   v[TBL]      LDS address of the next table entry;  v[TBL+3 : TBL+10]  the entry just read
#   s[96:97] next A, s[98:99] next B, s[76:77] next C (this wave's rows: + %[dab] / %[dab] / %[dc] or %[dw]; s100 up are reserved)
#   s78 units left (this one included), s79 the next unit's passes | flags << 16, s80 num_records of the next unit's A / B
#   descriptors: 0 when there is no next unit (those loads then fetch nothing); s88 passes left, s89 flags of this unit
F_ACC, F_NOEPI, F_PARTIAL = 0, 1, 2
PARTIAL_WAVE_BYTES = 0x10000       # one wave's 128 x 128 fp32 accumulators
TBL_BF16, TBL_FP8 = 177, 241


def fetch_next(tbl):
    e = tbl + 3
    return ["; ---- next unit's bases, pass count and flags from the table",
            f"ds_read_b128 v[{e}:{e + 3}], v{tbl}", f"ds_read_b128 v[{e + 4}:{e + 7}], v{tbl} offset:16",
            f"v_add_u32 v{tbl}, 32, v{tbl}", "s_waitcnt lgkmcnt(0)"] + \
           [f"v_readfirstlane_b32 s{r}, v{e + k}" for k, r in enumerate((96, 97, 98, 99, 76, 77, 79, 94))] + \
           ["s_lshl_b32 s94, s94, 16", "s_or_b32 s79, s79, s94",
            "s_add_u32 s96, s96, %[dab]", "s_addc_u32 s97, s97, 0", "s_add_u32 s98, s98, %[dab]", "s_addc_u32 s99, s99, 0",
            # this wave's part of the destination: its rows and columns of C, or its 64 KiB of a workspace slot
            f"s_bitcmp1_b32 s79, {16 + F_PARTIAL}", "s_cselect_b32 s94, %[dw], %[dc]",
            "s_add_u32 s76, s76, s94", "s_addc_u32 s77, s77, 0"]


# the unit whose bases sit in s[96:99] / s[76:77] / s79 becomes the current one (the A / B descriptors are switched by the
# last pass of the unit before it, or by the set-up)
ADOPT = ["s_mov_b32 s72, s76", "s_mov_b32 s73, s77", "s_and_b32 s88, s79, 0xffff", "s_lshr_b32 s89, s79, 16",
         f"s_bitcmp1_b32 s89, {F_PARTIAL}", f"s_cselect_b32 s74, {PARTIAL_WAVE_BYTES:#x}, %[rc2]"]


def setup_lines_p(RB, SCALE, tbl):
    o = ["; ==== setup (persistent)", "s_mov_b32 s66, %[ra2]", "s_mov_b32 s67, %[ra3]", "s_mov_b32 s70, %[ra2]", "s_mov_b32 s71, %[ra3]",
         "s_mov_b32 s75, %[ra3]",
         "s_lshl_b32 s81, %[lda], 3", "s_lshl_b32 s82, %[lda], 4", "s_add_u32 s83, s82, s81", "s_lshl_b32 s84, %[lda], 5",
         "s_add_u32 s85, s84, s81", "s_add_u32 s86, s84, s82", "s_add_u32 s87, s86, s81",
         "s_mov_b32 s78, %[ntl]",
         "s_mov_b32 s90, %[lw]", "s_add_u32 s91, s90, 0x8000", "s_add_u32 s92, s90, 0x10000", "s_add_u32 s93, s90, 0x18000",
         f"v_mov_b32 v{RB}, %[a0]", f"v_mov_b32 v{RB + 1}, %[a1]", f"v_mov_b32 v{RB + 2}, %[b0]", f"v_mov_b32 v{RB + 3}, %[b1]",
         f"v_add_u32 v{RB + 4}, 0x10000, v{RB}", f"v_add_u32 v{RB + 5}, 0x10000, v{RB + 1}",
         f"v_add_u32 v{RB + 6}, 0x10000, v{RB + 2}", f"v_add_u32 v{RB + 7}, 0x10000, v{RB + 3}",
         f"v_mov_b32 v{SCALE}, 0x7f7f7f7f", f"v_mov_b32 v{tbl}, %[tbl]"]
    o += fetch_next(tbl)
    o += ["s_mov_b32 s64, s96", "s_mov_b32 s65, s97", "s_mov_b32 s68, s98", "s_mov_b32 s69, s99"] + ADOPT
    o += next_or_none(tbl, "first")
    return o


def next_or_none(tbl, tag):
    """s78 = units left including the current one: fetch the next one's bases, or make its loads empty."""
    return ["s_mov_b32 s80, %[ra2]", "s_cmp_lg_u32 s78, 1", f"s_cbranch_scc1 Lklm%=_fetch_{tag}", "s_mov_b32 s80, 0",
            f"s_branch Lklm%=_fetched_{tag}", f"Lklm%=_fetch_{tag}:"] + fetch_next(tbl) + [f"Lklm%=_fetched_{tag}:"]


def epilogue32_lines():
    """A K slice's partial sums: the 256 fp32 accumulators of every lane straight from the AGPRs to this wave's 64 KiB of the
    unit's workspace slot, register-major ([acc tile][lane][16 bytes]: each store instruction writes 1 KiB contiguous). The
    kernel that adds the slices up knows the (tile, lane) -> (row, column) map (gemm_quad_asm.h, quad_partial_reduce)."""
    o = ["; ==== fp32 partial epilogue", "s_nop 15", "s_nop 15", "s_mov_b32 s94, 0"]
    for t in range(64):
        o.append(f"buffer_store_dwordx4 a[{4 * t}:{4 * t + 3}], %[vw], s[72:75], s94 offen offset:{(t % 4) * 1024}")
        if t % 4 == 3 and t != 63:
            o.append("s_add_u32 s94, s94, 0x1000")
    return o


SWITCH = ["; ---- last pass of this unit: its DMA fetches the next unit's K tiles 0 and 1",
          "s_mov_b32 s64, s96", "s_mov_b32 s65, s97", "s_mov_b32 s66, s80", "s_mov_b32 s68, s98", "s_mov_b32 s69, s99", "s_mov_b32 s70, s80"]


def body_p(dtype="bf16", sched="chain", ablate=(), **kw):
    global NMAJOR, DTYPE, DMA_CACHE, MX_SCALED
    NMAJOR, DTYPE, DMA_CACHE = False, dtype, ""
    MX_SCALED = "scaled" in ablate
    fp8 = dtype == "fp8"
    chain = sched == "chain" and not fp8
    rb, scale, tbl = (FRB, FSCALE, TBL_FP8) if fp8 else (CRB, CSCALE, TBL_CHAIN) if chain else (RB, SCALE, TBL_BF16)
    o = setup_lines_p(rb, scale, tbl)
    o += ["; ==== prologue: K tiles 0 and 1 of the first output tile on their way"]
    for st in (0, 1):
        for p in range(16):
            pre, main = dma(st, p)
            o += pre + ["s_nop 0"] + main
    o += ["s_waitcnt vmcnt(16)", "s_barrier"]
    if fp8:
        for j in range(8):
            o += [read_fp8("a", j, 0, 0, 0), read_fp8("a", j, 1, 0, 0)]
        for j in range(6 if MFMA_ORDER == "zigzag" else 7):
            o += [read_fp8("b", j, 0, 0), read_fp8("b", j, 1, 0)]
    elif chain:
        o += chain_first_reads(0)
    else:
        for j in range(16):
            o.append(read(0, 0, j))
    o += ["s_waitcnt lgkmcnt(0)", "Lklm%=_tile:"]
    waves = list(range(4)) if (not fp8 and sched in PER_WAVE) else [0]
    for w in waves[1:]:
        o += [f"s_cmp_eq_u32 %[wid], {w}", f"s_cbranch_scc1 Lklm%=_w{w}"]

    def one_pass(w, first):
        if fp8:
            return tile_fp8(0, "full", czero=first, **kw) + tile_fp8(1, "full", **kw)
        return tile(0, "full", sched, (), w, czero=first) + tile(1, "full", sched, (), w)

    for w in waves:
        # the first pass of an output tile starts its accumulators from the constant 0: nothing is ever cleared
        # (a unit that continues its tile's sum, F_ACC, goes straight to the loop)
        o += [f"; ==== wave {w}, first pass of an output tile (s88 = passes left)", f"Lklm%=_w{w}:",
              f"s_bitcmp1_b32 s89, {F_ACC}", f"s_cbranch_scc1 Lklm%=_l{w}",
              "s_cmp_lg_u32 s88, 1", f"s_cbranch_scc1 Lklm%=_f{w}"] + SWITCH + [f"Lklm%=_f{w}:"]
        o += one_pass(w, True)
        o += ["s_sub_u32 s88, s88, 1", "s_cmp_eq_u32 s88, 0", "s_cbranch_scc1 Lklm%=_epilogue"]
        o += [f"; ==== wave {w}, the other passes: two K tiles each", f"Lklm%=_l{w}:",
              "s_cmp_lg_u32 s88, 1", f"s_cbranch_scc1 Lklm%=_g{w}"] + SWITCH + [f"Lklm%=_g{w}:"]
        o += one_pass(w, False)
        o += ["s_sub_u32 s88, s88, 1", "s_cmp_lg_u32 s88, 0", f"s_cbranch_scc1 Lklm%=_l{w}"]
        if w != waves[-1]:
            o += ["s_branch Lklm%=_epilogue"]
    o += ["Lklm%=_epilogue:", f"s_bitcmp1_b32 s89, {F_NOEPI}", "s_cbranch_scc1 Lklm%=_next",
          f"s_bitcmp1_b32 s89, {F_PARTIAL}", "s_cbranch_scc1 Lklm%=_epi32"]
    # scratch = fragment set 1 (bf16 / fp4) / the odd K tiles' A set (fp8): dead after the last MFMA, while set 0 already
    # holds the next output tile's first fragments
    global STORE_CACHE
    STORE_CACHE = "".join(" " + x[8:] for x in ablate if x.startswith("stcache:"))
    # (schedule "chain": the B buffer of stage 1 -- a unit ends on a stage-1 tile, and the other buffer holds the next unit's B)
    epi = epilogue_lines(tmp=CB[1] if chain else A1)
    STORE_CACHE = ""
    if "store" in ablate:          # diagnostic: what the stores cost (C is never written)
        epi = [l for l in epi if not l.startswith("buffer_store")]
    if "epilogue" in ablate:       # diagnostic: what the whole epilogue costs
        epi = []
    o += epi
    o += ["s_branch Lklm%=_next", "Lklm%=_epi32:"] + epilogue32_lines()
    o += ["; ---- next unit", "Lklm%=_next:", "s_sub_u32 s78, s78, 1", "s_cmp_eq_u32 s78, 0", "s_cbranch_scc1 Lklm%=_done",
          "s_mov_b32 s66, %[ra2]", "s_mov_b32 s70, %[ra2]"] + ADOPT
    o += next_or_none(tbl, "loop")
    MX_SCALED = False
    o += ["s_branch Lklm%=_tile", "Lklm%=_done:",
          "; the last pass's (empty) LDS-DMA must not land in LDS the CU has handed to another workgroup",
          "s_waitcnt vmcnt(0)"]
    return o


def clobbers():
    c = ["memory", "scc", "vcc", "m0"]
    c += [f"s{i}" for i in range(64, 76)] + [f"s{i}" for i in range(76, 100)]
    c += [f"v{i}" for i in range(40, 252)]
    c += [f"a{i}" for i in range(256)]
    return c


def _stagger(wave, first=39, step=5, landed=85):
    """'spread', with wave w's DMA pieces one gap later per wave: the four waves of a workgroup run in lockstep (the
    barriers keep them there), so without this all four hand their DMA to the CU's one texture-address unit in the same
    cycle and three of them wait in its queue with their MFMA issue blocked behind the DMA."""
    dma_gaps = [first + step * p + wave for p in range(16)]
    nxt, g = [], landed + 1
    while len(nxt) < 16:
        if g not in dma_gaps:
            nxt.append(g)
            g += 2
        else:
            g += 1
    return {"b1": [2 * (1 + j) for j in range(8)], "a1": [0] + [2 * (9 + j) for j in range(7)], "free_barriers": [36],
            "dma_order": list(range(16)), "dma": dma_gaps, "landed": landed, "next": nxt}


SCHEDULES["stagger"] = _stagger
PER_WAVE = {"stagger", "chain"}


# ----------------------------------------------------------------------------------------------------------------
# fp8 (OCP e4m3, MX MFMA 16x16x128 with unit scales): same LDS image, DMA and read addresses as bf16 -- an fp8 fragment
# is 32 bytes per lane = the two 16-byte chunks bf16 reads as k-step 0 and k-step 1 -- but a K tile is ONE k-step of 64
# MFMAs of 32 cycles, and a fragment takes 8 registers, so two whole fragment sets do not fit beside 256 accumulators:
#   v[40:103] / v[104:167]   A fragments of the even / odd K tiles (8 x 8 registers each)
#   v[168:231]               B fragments, ONE set: slot ni is refilled with the next tile's B[ni] once this tile's 8
#                            MFMAs on it are issued (the MFMAs walk the tile columns: one by one under order "rows", in pairs under "zigzag")
#   v[232:239] read bases (as bf16), v240 the unit scales
# per K tile t (stage S = t & 1), gaps = "after MFMA i", i < 64:
#   0, 1        B[7](t) (its slot was busy until the previous tile's last MFMA); order "zigzag" walks columns 6 and 7 together,
#               so B[6](t) follows at gaps 2, 3 instead of being read after MFMA 55 of the tile before
#   6           s_waitcnt lgkmcnt(0); s_barrier     every wave holds all of tile t in registers: stage S is free
#   7 .. 44     the 16 LDS-DMA pieces of tile t+2 -> stage S, every 2.5 gaps (= 80 cycles, what bf16 'spread' settled on)
#   46          s_waitcnt vmcnt(16); s_barrier      tile t+1 landed for every wave (>= 60 gaps = 1.9 k cycles after its last piece)
#   47 .. 61    A[0..7](t+1) -> the other A set, B[0..5](t+1) -> their slots (two reads per gap), B[6] after MFMA 55
#   62          s_waitcnt lgkmcnt(0)
FA = (40, 104)
FB = 168
FRB = 232
FSCALE = 240


def f8(base, i):
    return f"v[{base + 8 * i}:{base + 8 * i + 7}]"


def mfma_fp8(i, aset, czero=False):
    # the order of bf16 with the roles of m and n swapped ("rows": n-tile-major; "zigzag": two tile COLUMNS at a time), because
    # the B slots are refilled column by column
    ni, mi = tile_of_step(i)
    if not MX_SCALED:
        return f"v_mfma_f32_16x16x128_f8f6f4 {acc(mi, ni)}, {f8(FB, ni)}, {f8(FA[aset], mi)}, {'0' if czero else acc(mi, ni)}"
    return (f"v_mfma_scale_f32_16x16x128_f8f6f4 {acc(mi, ni)}, {f8(FB, ni)}, {f8(FA[aset], mi)}, {'0' if czero else acc(mi, ni)}, "
            f"v{FSCALE}, v{FSCALE} op_sel_hi:[0,0,0]")


def read_fp8(operand, idx, half, stage, aset=0):
    """One ds_read_b128: half 0 = chunk (lane >> 4), half 1 = chunk 4 + (lane >> 4) of tile `idx` of operand 'a' / 'b'."""
    base = (FA[aset] if operand == "a" else FB) + 8 * idx + 4 * half
    reg = FRB + 4 * stage + (0 if operand == "a" else 2) + half
    off = f" offset:{idx * 2048}" if idx else ""
    return f"ds_read_b128 v[{base}:{base + 3}], v{reg}{off}"


def tile_fp8(t, mode, landed=46, step2=5, czero=False):
    """`step2`: DMA spacing in half gaps (5 = every 2.5 gaps); `landed`: the gap of the wait for tile t+1."""
    stage, aset = t & 1, t & 1
    out = [f"; ---- fp8 K tile in stage {stage}, {mode}"]
    after = {i: [] for i in range(64)}
    zz = MFMA_ORDER == "zigzag"
    after[0].append(read_fp8("b", 7, 0, stage))
    after[1].append(read_fp8("b", 7, 1, stage))
    if zz:
        # columns 6 and 7 are walked together (MFMAs 48 .. 63), so B[6]'s slot is busy to the tile's end as well: the tile
        # reads its own B[6] next to its B[7]
        after[2].append(read_fp8("b", 6, 0, stage))
        after[3].append(read_fp8("b", 6, 1, stage))
    after[6] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    if mode == "full":
        for p in range(16):
            i = 7 + (step2 * p) // 2
            pre, main = dma(stage, p)
            after[i - 1] += pre
            after[i] += main
        n_before = sum(1 for p in range(16) if 7 + (step2 * p) // 2 <= landed)
        after[landed] = after[landed] + [f"s_waitcnt vmcnt({n_before})", "s_barrier"]
    elif mode == "tail2":
        after[landed] += ["s_waitcnt vmcnt(0)", "s_barrier"]
    if mode != "last":
        # A[0..7](t+1) and B[0..4](t+1) (26 reads) over the gaps up to 59, then B[5] (slot free after MFMA 47) and B[6] (after 55)
        nxt = [("a", j, h) for j in range(8) for h in (0, 1)] + [("b", j, h) for j in range(5) for h in (0, 1)]
        span = 60 - (landed + 1)
        for k, (op, j, h) in enumerate(nxt):
            after[landed + 1 + (k * span) // len(nxt)].append(read_fp8(op, j, h, stage ^ 1, aset ^ 1))
        for op, j, h in (("b", 5, 0), ("b", 5, 1)):
            after[60].append(read_fp8(op, j, h, stage ^ 1))
        if not zz:
            for op, j, h in (("b", 6, 0), ("b", 6, 1)):
                after[61].append(read_fp8(op, j, h, stage ^ 1))
        # the next tile starts on A[*] and B[0] (zigzag: B[0], B[1]); B[5], B[6] (the last reads, in order) may still be in flight
        after[63].append("s_waitcnt lgkmcnt(%d)" % (2 if zz else 4))
    for i in range(64):
        out.append(mfma_fp8(i, aset, czero))
        out += after[i]
    return out


def body_fp8(**kw):
    global DMA_CACHE
    DMA_CACHE = ""
    o = setup_lines(FRB, FSCALE)
    o += ["; ==== prologue: tiles 0 and 1 on their way, accumulators cleared meanwhile"]
    for st in (0, 1):
        for p in range(16):
            pre, main = dma(st, p)
            o += pre + ["s_nop 0"] + main
    for i in range(256):
        o.append(f"v_accvgpr_write_b32 a{i}, 0")
    o += ["s_waitcnt vmcnt(16)", "s_barrier"]
    for j in range(8):
        o += [read_fp8("a", j, 0, 0, 0), read_fp8("a", j, 1, 0, 0)]
    for j in range(6 if MFMA_ORDER == "zigzag" else 7):      # B[7] (zigzag: B[6] too) is read by the tile itself, like every tile's
        o += [read_fp8("b", j, 0, 0), read_fp8("b", j, 1, 0)]
    o += ["s_waitcnt lgkmcnt(0)", "; ==== main loop: two K tiles per pass", "1:"]
    o += tile_fp8(0, "full", **kw) + tile_fp8(1, "full", **kw)
    o += ["s_sub_u32 s88, s88, 1", "s_cmp_lg_u32 s88, 0", "s_cbranch_scc1 1b"]
    o += tile_fp8(0, "tail2", **kw) + tile_fp8(1, "last", **kw)
    o += epilogue_lines()
    return o


# macro suffix -> (schedule, ablations, dtype). SHIPPED: the bodies libklm_ops.so and klm_burner are built from. DIAGNOSTIC: earlier
# placements kept for the A/B record and streams with parts of the loop removed (they keep a GEMM's timing, not its result);
# generated only with --diag and compiled only under -DKLM_GEMM_DIAG (scripts/exp/build_exp.sh), never into the shipped library.
# KLM_GEN_SCHEDULE=stagger (with KLM_GEN_MFMA_ORDER=rows|zigzag) regenerates the bf16 / fp4 streams of earlier states for A/B builds
BF16_SCHEDULE = os.environ.get("KLM_GEN_SCHEDULE", "chain")
SHIPPED = {"": (BF16_SCHEDULE, (), "bf16"), "_FP4": (BF16_SCHEDULE, (), "fp4"), "_FP8": ("fp8", (), "fp8"),
           # _ST: C stored with `sc1 nt` (written through, not kept): for shapes whose operands just fit the 256 MB Infinity Cache
           "_P_ST": ("persistent", ("stcache:sc1", "stcache:nt"), "bf16"), "_P_FP4_ST": ("persistent", ("stcache:sc1", "stcache:nt"), "fp4"),
           "_P_FP8_ST": ("persistent", ("stcache:sc1", "stcache:nt"), "fp8"),
           "_P": ("persistent", (), "bf16"), "_P_FP4": ("persistent", (), "fp4"), "_P_FP8": ("persistent", (), "fp8")}
DIAGNOSTIC = {"_STAGGER": ("stagger", (), "bf16"), "_SPREAD": ("spread", (), "bf16"), "_ONEBAR": ("one_barrier", (), "bf16"), "_PEROP": ("per_operand", (), "bf16"),
              "_NODMA": ("spread", ("dma",), "bf16"), "_NOREAD": ("spread", ("read",), "bf16"), "_L2ONLY": ("spread", ("bump",), "bf16"),
              "_MFMAONLY": ("spread", ("dma", "read", "barrier"), "bf16"),
              "_FP8B": ("fp8", (("landed", 46), ("step2", 6)), "fp8"),
              # ("cache:sc0" / "cache:sc1" / "cache:nt" in the ablation tuple put cache-policy bits on the LDS-DMA; measured: sc0 / sc1 no
              #  change, nt -10 %, profiles/r2_prof/dma_cache_bits.json -- the bodies are no longer generated)
              "_P_NOSTORE": ("persistent", ("store",), "bf16"), "_P_NOEPI": ("persistent", ("epilogue",), "bf16"),
              "_P_FP4_SC": ("persistent", ("scaled",), "fp4"), "_P_FP8_SC": ("persistent", ("scaled",), "fp8")}
BODIES = dict(SHIPPED, **DIAGNOSTIC)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    diag = "--diag" in sys.argv[1:]
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "k8s_llm_monitor_amd", "ops", "csrc")
    out = args[0] if args else os.path.join(here, "gemm_quad_asm_diag.inc" if diag else "gemm_quad_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by scripts/gen_gemm_quad_asm.py%s -- do not edit; the schedules are documented there.\n" % (" --diag" if diag else ""))
        for suffix, (sched, ablate, dtype) in (DIAGNOSTIC if diag else SHIPPED).items():
            if sched == "persistent":
                lines = body_p(dtype, ablate=ablate) if diag else body_p(dtype, sched=BF16_SCHEDULE, ablate=ablate)
            else:
                lines = body_fp8(**dict(ablate)) if dtype == "fp8" else body(sched, ablate, dtype=dtype)
            f.write(f"#define KLM_QUAD_ASM_BODY{suffix} \\\n")
            for ln in lines:
                f.write('  "%s\\n\\t" \\\n' % ln.replace('"', '\\"'))
            f.write('  ""\n')
            n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
            print(f"KLM_QUAD_ASM_BODY{suffix}: {len(lines)} lines, {n_mfma} MFMAs, {sum(1 for l in lines if l.startswith('buffer_load'))} DMA, "
                  f"{sum(1 for l in lines if l.startswith('ds_read'))} reads, {sum(1 for l in lines if l.startswith('s_barrier'))} barriers")
        if not diag:
            f.write("#define KLM_QUAD_ASM_CLOBBERS " + ", ".join('"%s"' % c for c in clobbers()) + "\n")


if __name__ == "__main__":
    main()
