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
    for suffix in gen.BODIES:
        assert f"#define KLM_QUAD_ASM_BODY{suffix} " in have


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
    last_use, pos = {}, {}
    n = -1
    for l in full:
        if l.startswith("v_mfma"):
            n += 1
            b = int(re.match(r"v_mfma\S+ a\[\d+:\d+\], v\[(\d+):", l).group(1))
            last_use[b] = n
        elif l.startswith("ds_read"):
            lo = int(re.match(r"ds_read_b128 v\[(\d+):", l).group(1))
            pos[lo] = n
    for lo, at in pos.items():
        slot = lo - (lo - gen.FB) % 8 if gen.FB <= lo < gen.FB + 64 else None
        if slot is not None and not (slot == gen.FB + 56 and at <= 1):        # B[7] of THIS tile is read at gaps 0, 1
            assert at >= last_use[slot], (lo, at, last_use[slot])             # a next-tile B fragment: its slot's MFMAs are all issued
    # the wait for tile t+1 comes after all 16 pieces of tile t+2 went out
    idx = next(i for i, l in enumerate(full) if l.startswith("s_waitcnt vmcnt"))
    assert "vmcnt(16)" in full[idx] and sum(l.startswith("buffer_load") for l in full[:idx]) == 16


def test_epilogue_pairs_tiles_and_pads_the_permlane_hazard(gen):
    e = gen.epilogue_lines()
    assert sum(l.startswith("buffer_store_dwordx4") for l in e) == 32 and sum(l.startswith("v_permlane16_swap") for l in e) == 64
    assert sum(l.startswith("v_accvgpr_read") for l in e) == 256
    for i, l in enumerate(e):
        if l.startswith("v_permlane16_swap") and not e[i - 1].startswith("v_permlane16_swap"):
            assert e[i - 1] == "s_nop 1"                                       # VALU write -> swap: two wait states
