#!/usr/bin/env python3
"""Generates ops/csrc/gemm_quad_asm.inc: the hand-scheduled gfx950 instruction stream of the 4-wave 256 x 256 bf16 GEMM
(`gemm_bf16_quad_asm` in gemm_bf16.hip) as one inline-asm body.

Why generated text and not HIP source: with one wave per SIMD there is no partner wave to hide anything behind, so the
order of the wave's own instruction stream IS the schedule. Written as HIP (gemm_quad128.h: builtins or per-MFMA asm
statements pinned with sched_barrier) the compiler still decides the waits and the interleave, and that kernel reached
0.75 MFMA-busy; the vendor's hand-written kernel of the same geometry reaches 0.88 (profiles/r2_prof). Here every
instruction's place is fixed:

  per K tile (stage S, 128 MFMAs, m-tile-major so consecutive MFMAs share the A fragment):
    after MFMA 0,2,..,30     the 16 ds_read_b128 of this tile's k-step 1 -> fragment set 1
    after MFMA 40            s_waitcnt lgkmcnt(0); s_barrier            (#1: every wave is done reading stage S)
    after MFMA 41,43,..,63   LDS-DMA pieces 0..11 of tile t+2 -> stage S
    after MFMA 63            s_waitcnt vmcnt(12); s_barrier             (#2: tile t+1 landed, for every wave)
    after MFMA 64,66,..,94   the 16 ds_read_b128 of tile t+1's k-step 0 -> fragment set 0
    after MFMA 65,..,71      LDS-DMA pieces 12..15
    after MFMA 126           s_waitcnt lgkmcnt(0)
  never more than one memory instruction in an MFMA gap.

Registers (fixed, all on the asm statement's clobber list):
  a[0:255]        accumulators, tile (mi, ni) at a[4 (8 mi + ni)]
  v[40:71]        A fragments set 0 (k-step 0), v[72:103] B set 0, v[104:135] A set 1, v[136:167] B set 1
  v[168:175]      ds_read bases: stage 0 {a k0, a k1, b k0, b k1}, stage 1 the same + 64 KiB
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


def mfma(i):
    """MFMA i of a K tile: k-step i // 64, m tile (i % 64) // 8, n tile i % 8."""
    ks, mi, ni = i // 64, (i % 64) // 8, i % 8
    a, b = (A0, B0) if ks == 0 else (A1, B1)
    return f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {frag(b, ni)}, {frag(a, mi)}, {acc(mi, ni)}"


def read(setno, stage, j):
    """Fragment read j of 16 in need order A[0], B[0..7], A[1..7]; `stage` = the LDS stage read from."""
    abase, bbase = (A0, B0) if setno == 0 else (A1, B1)
    ra = RB + 4 * stage + setno          # a, k-step = setno
    rb = RB + 4 * stage + 2 + setno
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
    main = [f"buffer_load_dwordx4 %[va], {rs}, {so} offen lds"]
    if q == 7:
        lo = 64 if op == "A" else 68
        main += [f"s_add_u32 s{lo}, s{lo}, 0x80", f"s_addc_u32 s{lo + 1}, s{lo + 1}, 0"]
    return pre, main


def tile(stage, mode):
    """mode: 'full' (issues tile t+2), 'tail2' (tile nt-2: no DMA), 'last' (tile nt-1: no DMA, no next reads)."""
    out = [f"; ---- K tile in stage {stage}, {mode}"]
    after = {i: [] for i in range(128)}
    for j in range(16):
        after[2 * j].append(read(1, stage, j))
    after[40] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    if mode == "full":
        for p in range(16):
            i = 41 + 2 * p if p < 12 else 65 + 2 * (p - 12)
            pre, main = dma(stage, p)
            after[i - 1] += pre          # the gap before: an MFMA sits between the M0 write and its use
            after[i] += main
        after[63] += ["s_waitcnt vmcnt(12)", "s_barrier"]
    elif mode == "tail2":
        after[63] += ["s_waitcnt vmcnt(0)", "s_barrier"]
    if mode != "last":
        for j in range(16):
            after[64 + 2 * j].append(read(0, stage ^ 1, j))
        after[126].append("s_waitcnt lgkmcnt(0)")
    for i in range(128):
        out.append(mfma(i))
        out += after[i]
    return out


def body():
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
          f"v_add_u32 v{RB + 6}, 0x10000, v{RB + 2}", f"v_add_u32 v{RB + 7}, 0x10000, v{RB + 3}"]
    o += ["; ==== prologue: tiles 0 and 1 on their way, accumulators cleared meanwhile"]
    for st in (0, 1):
        for p in range(16):
            pre, main = dma(st, p)
            o += pre + ["s_nop 0"] + main
    for i in range(256):
        o.append(f"v_accvgpr_write_b32 a{i}, 0")
    o += ["s_waitcnt vmcnt(16)", "s_barrier"]
    for j in range(16):
        o.append(read(0, 0, j))
    o += ["s_waitcnt lgkmcnt(0)"]
    o += ["; ==== main loop: two K tiles per pass", "1:"]
    o += tile(0, "full")
    o += tile(1, "full")
    o += ["s_sub_u32 s88, s88, 1", "s_cmp_lg_u32 s88, 0", "s_cbranch_scc1 1b"]
    o += tile(0, "tail2")
    o += tile(1, "last")
    o += ["; ==== epilogue: fp32 -> bf16, 8 bytes per lane and tile", "s_nop 15", "s_nop 15", "s_mov_b32 s94, 0", "s_lshl_b32 s95, %[ldc], 4"]
    t = 0
    for mi in range(8):
        for ni in range(8):
            r = A0 + 8 * (t % 8)
            b = 4 * (8 * mi + ni)
            o += [f"v_accvgpr_read_b32 v{r}, a{b}", f"v_accvgpr_read_b32 v{r + 1}, a{b + 1}",
                  f"v_accvgpr_read_b32 v{r + 2}, a{b + 2}", f"v_accvgpr_read_b32 v{r + 3}, a{b + 3}",
                  f"v_cvt_pk_bf16_f32 v{r}, v{r}, v{r + 1}", f"v_cvt_pk_bf16_f32 v{r + 1}, v{r + 2}, v{r + 3}",
                  f"buffer_store_dwordx2 v[{r}:{r + 1}], %[vc], s[72:75], s94 offen offset:{ni * 32}"]
            t += 1
        o.append("s_add_u32 s94, s94, s95")
    o += ["s_waitcnt vmcnt(0)"]
    return o


def clobbers():
    c = ["memory", "scc", "vcc", "m0"]
    c += [f"s{i}" for i in range(64, 76)] + [f"s{i}" for i in range(80, 96)]
    c += [f"v{i}" for i in range(40, 176)]
    c += [f"a{i}" for i in range(256)]
    return c


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "k8s_llm_monitor_amd", "ops", "csrc", "gemm_quad_asm.inc")
    lines = body()
    with open(out, "w") as f:
        f.write("// GENERATED by scripts/gen_gemm_quad_asm.py -- do not edit; the schedule is documented there.\n")
        f.write("#define KLM_QUAD_ASM_BODY \\\n")
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln.replace('"', '\\"'))
        f.write('  ""\n')
        f.write("#define KLM_QUAD_ASM_CLOBBERS " + ", ".join('"%s"' % c for c in clobbers()) + "\n")
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    print(f"{out}: {len(lines)} instructions/labels, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
