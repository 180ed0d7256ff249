"""The generated instruction streams of the hand-scheduled GEMM kernels (CPU): the checked-in .inc is what the
generator writes, and every schedule keeps the invariants the hardware needs (counts, hazards, wait counts)."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_gemm_quad_asm", os.path.join(ROOT, "scripts", "gen_gemm_quad_asm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_checked_in_inc_is_the_generators_output(gen, tmp_path, capsys):
    import sys
    out = tmp_path / "gemm_quad_asm.inc"
    argv, sys.argv = sys.argv, ["gen", str(out)]
    try:
        gen.main()
    finally:
        sys.argv = argv
    have = open(os.path.join(ROOT, "k8s_llm_monitor_amd", "ops", "csrc", "gemm_quad_asm.inc")).read()
    assert out.read_text() == have, "run scripts/gen_gemm_quad_asm.py and rebuild"
    for suffix in gen.SHIPPED:
        assert f"#define KLM_QUAD_ASM_BODY{suffix} " in have
    # the A/B-record and diagnostic streams are not part of the shipped file: --diag writes them elsewhere
    for suffix in gen.DIAGNOSTIC:
        assert f"#define KLM_QUAD_ASM_BODY{suffix} " not in have
    assert set(gen.SHIPPED) & set(gen.DIAGNOSTIC) == set() and len(have.splitlines()) < 30000


def _kinds(lines):
    return {"mfma": sum(l.startswith("v_mfma") for l in lines), "dma": sum(l.startswith("buffer_load") for l in lines),
            "read": sum(l.startswith("ds_read") for l in lines), "barrier": sum(l.startswith("s_barrier") for l in lines)}


@pytest.mark.parametrize("sched", ["one_barrier", "per_operand", "spread", "stagger"])
def test_bf16_tiles_hold_every_instruction_once_and_respect_the_hazards(gen, sched):
    for wave in range(4):
        full = gen.tile(0, "full", sched, (), wave)
        k = _kinds(full)
        assert k == {"mfma": 128, "dma": 16, "read": 32, "barrier": k["barrier"]} and k["barrier"] in (2, 3)
        # every accumulator tile gets exactly two MFMAs (k-step 0 and 1), each fragment register range is read into once per set
        accs = [re.match(r"v_mfma\S+ (a\[\d+:\d+\])", l).group(1) for l in full if l.startswith("v_mfma")]
        assert len(set(accs)) == 64 and all(accs.count(a) == 2 for a in set(accs))
        dsts = [re.match(r"ds_read_b128 (v\[\d+:\d+\])", l).group(1) for l in full if l.startswith("ds_read")]
        assert len(set(dsts)) == 32
        # an M0 write is never directly followed by the DMA that uses it, and no MFMA gap carries two memory instructions
        for a, b in zip(full, full[1:]):
            assert not (("m0" in a.split()[1] if a.startswith("s_") and len(a.split()) > 1 else False) and b.startswith("buffer_load")), (a, b)
        gap = 0
        for l in full:
            if l.startswith("v_mfma"):
                gap = 0
            elif l.startswith(("ds_read", "buffer_load")):
                gap += 1
                assert gap <= 1, (sched, wave, l)
        # the wait for tile t+1 leaves exactly the pieces of tile t+2 issued before it in flight
        idx = next(i for i, l in enumerate(full) if l.startswith("s_waitcnt vmcnt"))
        assert int(re.search(r"vmcnt\((\d+)\)", full[idx]).group(1)) == sum(l.startswith("buffer_load") for l in full[:idx])
        # the tail tiles issue no DMA; the last one reads nothing ahead
        assert _kinds(gen.tile(0, "tail2", sched, (), wave))["dma"] == 0
        last = _kinds(gen.tile(1, "last", sched, (), wave))
        assert last["dma"] == 0 and last["read"] == 16


def test_fp8_tile_refills_a_b_slot_only_after_its_last_mfma(gen):
    full = gen.tile_fp8(0, "full")
    assert _kinds(full) == {"mfma": 64, "dma": 16, "read": 32, "barrier": 2}
    last_use, first_use, pos = {}, {}, {}
    n = -1
    for l in full:
        if l.startswith("v_mfma"):
            n += 1
            b = int(re.match(r"v_mfma\S+ a\[\d+:\d+\], v\[(\d+):", l).group(1))
            last_use[b] = n
            first_use.setdefault(b, n)
        elif l.startswith("ds_read"):
            lo = int(re.match(r"ds_read_b128 v\[(\d+):", l).group(1))
            pos[lo] = n
    # the tile's own late fragments: B[7] at gaps 0, 1 and -- order "zigzag", which walks columns 6 and 7 together -- B[6] at gaps 2, 3;
    # both land before the wait at gap 6 and are first used long after it
    own = {gen.FB + 56: 1} if gen.MFMA_ORDER != "zigzag" else {gen.FB + 56: 1, gen.FB + 48: 3}
    wait6 = [i for i, l in enumerate(full) if l == "s_waitcnt lgkmcnt(0)"][0]
    assert sum(l.startswith("v_mfma") for l in full[:wait6]) == 7
    for lo, at in pos.items():
        slot = lo - (lo - gen.FB) % 8 if gen.FB <= lo < gen.FB + 64 else None
        if slot in own and at <= own[slot]:
            assert first_use[slot] > 6, (lo, at, first_use[slot])
        elif slot is not None:
            assert at >= last_use[slot], (lo, at, last_use[slot])             # a next-tile B fragment: its slot's MFMAs are all issued
    assert sum(1 for lo, at in pos.items() if gen.FB <= lo < gen.FB + 64 and at <= 3) == 2 * len(own)
    # the wait for tile t+1 comes after all 16 pieces of tile t+2 went out
    idx = next(i for i, l in enumerate(full) if l.startswith("s_waitcnt vmcnt"))
    assert "vmcnt(16)" in full[idx] and sum(l.startswith("buffer_load") for l in full[:idx]) == 16


def test_mfma_order_changes_one_source_per_step(gen):
    """Order "zigzag" (the shipped one): every MFMA of a k-step shares exactly one source fragment with the one before it, the
    two sources take turns, every accumulator tile is visited once; "rows" is the tile-row-major order of rounds 2 - 3."""
    z = [gen.tile_of_step(j, "zigzag") for j in range(64)]
    assert len(set(z)) == 64 and z[0] == (0, 0)
    assert all((a[0] != b[0]) + (a[1] != b[1]) == 1 for a, b in zip(z, z[1:]))
    changed = ["m" if a[0] != b[0] else "n" for a, b in zip(z, z[1:])]
    for p in range(4):                                                        # m, n, m, n, .. inside a pair of rows; m from pair to pair
        assert changed[16 * p: 16 * p + 15] == ["m", "n"] * 7 + ["m"]
    assert [changed[16 * p + 15] for p in range(3)] == ["m"] * 3
    assert [gen.tile_of_step(j, "rows") for j in range(64)] == [(j // 8, j % 8) for j in range(64)]
    assert gen.MFMA_ORDER == "zigzag"
    # the bf16 K tile follows it in both k-steps; fp8 with the roles of m and n swapped (its B slots are refilled by column)
    t = [l for l in gen.tile(0, "full", "stagger", (), 0) if l.startswith("v_mfma")]
    accs = [int(re.match(r"v_mfma\S+ a\[(\d+):", l).group(1)) // 4 for l in t]
    assert accs[:64] == [8 * m + n for m, n in z] and accs[64:] == accs[:64]
    # ... and the shipped schedule ("chain") visits the tiles in the same order, both k-steps of a tile back to back
    t = [l for l in gen.tile(0, "full", "chain", (), 0) if l.startswith("v_mfma")]
    accs = [int(re.match(r"v_mfma\S+ a\[(\d+):", l).group(1)) // 4 for l in t]
    assert accs[0::2] == [8 * m + n for m, n in z] and accs[1::2] == accs[0::2]
    assert gen.SHIPPED[""][0] == "chain" and gen.SHIPPED["_FP4"][0] == "chain"
    t8 = [l for l in gen.tile_fp8(0, "full") if l.startswith("v_mfma")]
    assert [int(re.match(r"v_mfma\S+ a\[(\d+):", l).group(1)) // 4 for l in t8] == [8 * n + m for m, n in z]


def test_chain_tile_keeps_every_fragment_alive_until_its_last_mfma(gen):
    """Schedule "chain" (bf16 / fp4): a tile's two k-steps run back to back on one accumulator (k-step 0 first: the bits of the
    other schedules), A is one register set refilled row pair by row pair, B is double-buffered by stage."""
    for wave in range(4):
        for stage in (0, 1):
            full = gen.tile_chain(stage, "full", wave)
            assert _kinds(full) == {"mfma": 128, "dma": 16, "read": 32, "barrier": 2}
            mf = [l for l in full if l.startswith("v_mfma")]
            accs = [re.match(r"v_mfma\S+ (a\[\d+:\d+\])", l).group(1) for l in mf]
            assert all(accs[2 * j] == accs[2 * j + 1] for j in range(64)) and len(set(accs)) == 64
            src = [tuple(int(x) for x in re.match(r"v_mfma\S+ a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", l).groups()) for l in mf]
            # k-step 0 before k-step 1 on every accumulator: the even MFMA reads the k = 0 fragments
            assert all((b - gen.CB[stage]) % 8 == 0 and (a - gen.CA) % 8 == 0 for b, a in src[0::2])
            assert all((b - gen.CB[stage]) % 8 == 4 and (a - gen.CA) % 8 == 4 for b, a in src[1::2])
            # walk the stream: a register is written by a read only after its last MFMA of THIS tile, and read by an MFMA only
            # after the wait that covers the read
            n, last_use, first_use = -1, {}, {}
            for l in mf:
                n += 1
                for r in src[n]:
                    last_use[r] = n
                    first_use.setdefault(r, n)
            n, pending = -1, {}
            own = set()
            for l in full:
                if l.startswith("v_mfma"):
                    n += 1
                    for r in src[n]:
                        assert r not in pending, (wave, stage, n, r)          # never reads a fragment whose load has not been waited for
                elif l.startswith("ds_read"):
                    lo = int(re.match(r"ds_read_b128 v\[(\d+):", l).group(1))
                    if n <= 3:
                        own.add(lo)                                             # rows 6, 7 of this tile
                        assert first_use[lo] >= 96
                    else:
                        # the next tile's fragments: A into the one A set (its row pair is done), B into the OTHER stage's buffer
                        assert (gen.CA <= lo < gen.CA + 48 and n >= last_use[lo]) or gen.CB[stage ^ 1] <= lo < gen.CB[stage ^ 1] + 64, (lo, n)
                    pending[lo] = n
                elif l.startswith("s_waitcnt lgkmcnt(0)"):
                    pending.clear()
            assert own == {gen.CA + 48 + 4 * i for i in range(4)} and not pending
            # the wait for tile t+1 leaves exactly the pieces of tile t+2 issued before it in flight; no gap holds two memory instructions
            idx = next(i for i, l in enumerate(full) if l.startswith("s_waitcnt vmcnt"))
            assert int(re.search(r"vmcnt\((\d+)\)", full[idx]).group(1)) == sum(l.startswith("buffer_load") for l in full[:idx])
            gap = 0
            for a, b in zip(full, full[1:]):
                assert not (a.startswith("s_") and len(a.split()) > 1 and "m0" in a.split()[1] and b.startswith("buffer_load")), (a, b)
            for l in full:
                if l.startswith("v_mfma"):
                    gap = 0
                elif l.startswith(("ds_read", "buffer_load")):
                    gap += 1
                    assert gap <= 1, (wave, l)
        assert _kinds(gen.tile_chain(0, "tail2", wave))["dma"] == 0 and _kinds(gen.tile_chain(0, "tail2", wave))["read"] == 32
        last = _kinds(gen.tile_chain(1, "last", wave))
        assert last == {"mfma": 128, "dma": 0, "read": 4, "barrier": 1}
    # what must be in registers before a tile starts is exactly what the tile before it (or the prologue) loads ahead
    first = {int(re.match(r"ds_read_b128 v\[(\d+):", l).group(1)) for l in gen.chain_first_reads(0)}
    assert first == {gen.CB[0] + 4 * i for i in range(16)} | {gen.CA + 4 * i for i in range(12)}
    ahead = {int(re.match(r"ds_read_b128 v\[(\d+):", l).group(1)) for l in gen.tile_chain(1, "full", 0) if l.startswith("ds_read")} - {gen.CA + 48 + 4 * i for i in range(4)}
    assert ahead == first
    # the first K tile of an output tile starts every accumulator from the constant 0 on its k-step 0 MFMA
    z = [l for l in gen.tile_chain(0, "full", 0, czero=True) if l.startswith("v_mfma")]
    assert all(l.endswith(", 0") for l in z[0::2]) and not any(l.endswith(", 0") for l in z[1::2])


def test_epilogue_pairs_tiles_and_pads_the_permlane_hazard(gen):
    e = gen.epilogue_lines()
    assert sum(l.startswith("buffer_store_dwordx4") for l in e) == 32 and sum(l.startswith("v_permlane16_swap") for l in e) == 64
    assert sum(l.startswith("v_accvgpr_read") for l in e) == 256
    for i, l in enumerate(e):
        if l.startswith("v_permlane16_swap") and not e[i - 1].startswith("v_permlane16_swap"):
            assert e[i - 1] == "s_nop 1"                                       # VALU write -> swap: two wait states


@pytest.mark.parametrize("dtype", ["bf16", "fp4", "fp8"])
def test_persistent_body_is_one_stream_over_output_tiles(gen, dtype):
    L = gen.body_p(dtype)
    waves = 1 if dtype == "fp8" else 4
    per_pass = 128 if dtype == "fp8" else 256
    # per wave: a first pass whose first K tile starts the accumulators from the constant 0, and a loop pass
    mf = [l for l in L if l.startswith("v_mfma")]
    assert len(mf) == waves * 2 * per_pass
    assert sum(l.split(", ")[3].startswith("0") for l in mf) == waves * 64
    assert not any(l.startswith("v_accvgpr_write") for l in L)                       # nothing is ever cleared
    # every pass may be the last of its output tile: the descriptor switch sits in front of each, skipped unless s88 == 1
    sw = [i for i, l in enumerate(L) if l.startswith("; ---- last pass of this unit")]
    assert len(sw) == 2 * waves
    for i in sw:
        assert L[i - 2] == "s_cmp_lg_u32 s88, 1" and L[i - 1].startswith("s_cbranch_scc1")
        assert L[i + 1: i + 7] == gen.SWITCH[1:]
    # all wave copies run the same number of barriers per pass (they meet at every one)
    if waves == 4:
        marks = [i for i, l in enumerate(L) if l.startswith("; ==== wave")] + [L.index("Lklm%=_epilogue:")]
        counts = [sum(l == "s_barrier" for l in L[a:b]) for a, b in zip(marks, marks[1:])]
        assert len(set(counts)) == 1, counts
    # the epilogue's scratch is dead by then: fp8 the odd K tiles' A set; bf16 / fp4 (schedule "chain") the B buffer of stage 1 (a unit
    # ends on a stage-1 tile; the A set and the other B buffer already hold the next output tile's first fragments)
    e = L[L.index("Lklm%=_epilogue:"):]
    regs = {int(m) for l in e if l.startswith("v_accvgpr_read") for m in re.findall(r"v_accvgpr_read_b32 v(\d+)", l)}
    lo = gen.A1 if dtype == "fp8" else gen.CB[1]
    assert regs and min(regs) >= lo and max(regs) < lo + 64
    # a table entry is fetched only while another output tile follows; the kernel ends with its (empty) DMA drained
    assert L[-1] == "s_waitcnt vmcnt(0)" and L[-3] == "Lklm%=_done:"
    for tag in ("first", "loop"):
        i = L.index(f"Lklm%=_fetch_{tag}:")
        assert L[i - 4: i] == ["s_cmp_lg_u32 s78, 1", f"s_cbranch_scc1 Lklm%=_fetch_{tag}", "s_mov_b32 s80, 0", f"s_branch Lklm%=_fetched_{tag}"]
    # units (round 3): a unit that continues its tile's sum skips the constant-0 pass; the end of a unit takes no epilogue, the
    # bf16 epilogue or the fp32 partial one by its flags; pass count and flags come from the table entry, never from an operand
    for w in range(waves):
        i = L.index(f"Lklm%=_w{w}:")
        assert L[i + 1: i + 3] == [f"s_bitcmp1_b32 s89, {gen.F_ACC}", f"s_cbranch_scc1 Lklm%=_l{w}"]
    i = L.index("Lklm%=_epilogue:")
    assert L[i + 1: i + 5] == [f"s_bitcmp1_b32 s89, {gen.F_NOEPI}", "s_cbranch_scc1 Lklm%=_next", f"s_bitcmp1_b32 s89, {gen.F_PARTIAL}", "s_cbranch_scc1 Lklm%=_epi32"]
    p32 = L[L.index("Lklm%=_epi32:"): L.index("Lklm%=_next:")]
    st = [l for l in p32 if l.startswith("buffer_store_dwordx4")]
    assert len(st) == 64 and [re.match(r"buffer_store_dwordx4 a\[(\d+):", l).group(1) for l in st] == [str(4 * t) for t in range(64)]
    assert all("%[vw], s[72:75], s94 offen" in l for l in st) and sum(l == "s_add_u32 s94, s94, 0x1000" for l in p32) == 15    # 64 KiB per wave, 1 KiB per store
    assert p32[2:4] == ["s_nop 15", "s_nop 15"]                                      # MFMA result -> VMEM read of the AGPR
    assert not any("%[passes]" in l for l in L) and sum(l == "s_and_b32 s88, s79, 0xffff" for l in L) == 2
    for adopt_at in (L.index("Lklm%=_next:"),):
        tail = L[adopt_at:]
        assert gen.ADOPT[0] in tail and tail.index(gen.ADOPT[0]) < tail.index("s_mov_b32 s80, %[ra2]")
    # no register above the architectural limits, none of the reserved scalar registers
    for l in L:
        for m in re.findall(r"\bs(\d+)\b", l):
            assert int(m) < 100, l
        for m in re.findall(r"\bv\[?(\d+)", l):
            assert int(m) < 256, l


def test_mfma_order_harness_orders_are_permutations():
    """scripts/exp/gen_mfma_order_exp.py (the register-only loops behind MFMA_ORDER and schedule "chain"): every order visits each
    of the 64 accumulator tiles once, the zigzags change exactly one source per step, the two-set bodies give every accumulator
    both k-steps, and the generated file holds one kernel per variant."""
    spec = importlib.util.spec_from_file_location("gen_mfma_order_exp", os.path.join(ROOT, "scripts", "exp", "gen_mfma_order_exp.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for kind in ("m", "m_snake", "n", "n_snake", "diag", "block2", "zigzag2", "zigzag2n"):
        o = m.order16(kind)
        assert sorted(o) == [(a, b) for a in range(8) for b in range(8)], kind
    for kind in ("zigzag2", "zigzag2n", "m_snake", "n_snake"):
        o = m.order16(kind)
        assert all((a[0] != b[0]) + (a[1] != b[1]) == 1 for a, b in zip(o, o[1:])), kind
    assert all((a[0] != b[0]) and (a[1] != b[1]) for a, b in zip(m.order16("diag"), m.order16("diag")[1:]))
    for kind in ("ksteps", "dependent", "double_snake", "pairs"):
        body = m.body16_two_sets(kind)
        accs = [re.match(r"v_mfma\S+ (a\[\d+:\d+\])", l).group(1) for l in body]
        assert len(body) == 256 and all(accs.count(a) == 4 for a in set(accs)) and len(set(accs)) == 64     # two halves (A, -A) x two k-steps
    dep = m.body16_two_sets("dependent")
    assert all(dep[i].split(",")[0] == dep[i + 1].split(",")[0] for i in range(0, 256, 2))                 # back to back on one accumulator
    names = [n for n, _ in m.VARIANTS]
    assert len(names) == len(set(names)) and all(len(l) in (32, 64, 128, 256) for _, l in m.VARIANTS)
