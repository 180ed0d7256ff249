"""Operand layout helpers of the GEMM wrappers (CPU): which row pitches get padded, what a padded operand looks like, and
that the wrappers refuse what the kernels cannot take before anything is launched."""
import os

import pytest
import torch

from k8s_llm_monitor_amd import ops


def test_camping_pitch_is_a_multiple_of_2k_that_is_no_power_of_two():
    bad = [6144, 12288, 20480, 24576, 28672, 40960, 49152, 10240]
    good = [512, 2048, 4096, 8192, 16384, 32768, 65536, 24064, 24832, 12032]
    assert all(ops.camping_pitch(n) for n in bad) and not any(ops.camping_pitch(n) for n in good)


@pytest.mark.parametrize("dtype,cols,padded", [(torch.bfloat16, 12288, True), (torch.bfloat16, 8192, False), (torch.uint8, 12288, True),
                                               (torch.float8_e4m3fn, 6144, True), (torch.bfloat16, 12032, False), (torch.bfloat16, 1024, False)])
def test_padded_operand_pads_only_camping_pitches(dtype, cols, padded):
    t = ops.padded_operand(256, cols, dtype, "cpu")
    es = torch.empty((), dtype=dtype).element_size()
    assert t.shape == (256, cols) and t.dtype == dtype and t.stride(1) == 1
    if padded:
        assert not t.is_contiguous() and t.stride(0) * es == cols * es + 256 and not ops.camping_pitch(t.stride(0) * es)
    else:
        assert t.is_contiguous()


def test_pitch_of_a_pair_of_operands():
    a = ops.padded_operand(256, 12288, torch.bfloat16, "cpu")
    b = ops.padded_operand(512, 12288, torch.bfloat16, "cpu")
    assert ops._pitch_bytes(a, b) == 12288 * 2 + 256
    assert ops._pitch_bytes(torch.empty(256, 64), torch.empty(256, 64)) == 0
    with pytest.raises(ops.KlmOpsError, match="one row pitch"):
        ops._pitch_bytes(a, torch.empty(512, 12288, dtype=torch.bfloat16))
    with pytest.raises(ops.KlmOpsError, match="GPU only"):
        ops.gemm_bf16(a, b)


def test_build_staleness_is_a_content_hash_embedded_in_the_binary(tmp_path):
    """mtimes prove nothing after a snapshot copy or a checkout. The build passes a hash of the target's
    sources to the compiler, the binary carries it, `stale()` compares it with the tree, `smoke()` asserts it."""
    import shutil
    import time
    from k8s_llm_monitor_amd import build
    if not shutil.which("g++"):
        import pytest
        pytest.skip("g++ not installed")
    src = tmp_path / "tool.cpp"
    hdr = tmp_path / "tool.h"
    hdr.write_text("#define ANSWER 42\n")
    src.write_text('#include <cstdio>\n#include "tool.h"\nextern "C" __attribute__((used)) const char klm_src_hash_marker[] = "KLM_SRC_HASH=" KLM_SRC_HASH;\n'
                   'int main() { printf("%d %s\\n", ANSWER, klm_src_hash_marker); return 0; }\n')
    t = build.Target("tool", tmp_path / "tool", [src], "cxx-exe", deps=[hdr])
    assert t.stale() and build.embedded_hash(t.out) is None
    assert build.build_target(t, verbose=False) is True
    h1 = t.source_hash()
    assert not t.stale() and build.embedded_hash(t.out) == h1 and build.build_target(t, verbose=False) is False
    # touching the files (new mtimes, same content) does not make it stale; an OLDER source with new content does
    now = time.time()
    os.utime(src, (now + 100, now + 100))
    assert not t.stale()
    hdr.write_text("#define ANSWER 43\n")
    os.utime(hdr, (now - 10000, now - 10000))
    assert t.stale() and t.source_hash() != h1
    build.build_target(t, verbose=False)
    assert not t.stale() and build.embedded_hash(t.out) == t.source_hash()
    # flags are part of the hash: the sanitizer preset of the same sources is another artefact
    assert build.sanitized(t, tmp_path / "asan").source_hash() != t.source_hash()


def test_shipped_library_carries_the_tree_hash_and_the_hbm_launch_plan():
    from k8s_llm_monitor_amd import build, ops
    if build.hipcc() is None:
        import pytest
        pytest.skip("hipcc not installed")
    build.build_all(verbose=False)
    ops.assert_current()
    assert ops.src_hash() == ops.tree_hash() and ops.src_hash() in ops.version() and len(ops.src_hash()) == 16
    assert all(not t.stale() for t in build.targets())
    # HIP's grid limit (2^32 - 1 threads) is 64 GiB at one 16-byte access per thread; the 288 GB card needs more
    GIB = 1 << 30
    assert ops.hbm_launch_plan(4 * GIB) == (1, 4 * GIB // 4096, 4 * GIB // 4096)
    assert ops.hbm_launch_plan(64 * GIB - 4096) == (1, 16777215, 16777215)
    n, first, last = ops.hbm_launch_plan(64 * GIB)
    assert (n, first, last) == (2, 16777215, 1)
    n, first, last = ops.hbm_launch_plan(200 * GIB)
    assert n == 4 and first == 16777215 and (n - 1) * first + last == 200 * GIB // 4096
    assert first * 256 <= 0xFFFFFFFF < (first + 1) * 256


def test_stream_plan_splits_the_partial_round_only_where_it_pays():
    """The persistent streams' plan (klm_quad_plan): R full rounds of whole tiles, then the r remaining tiles, cut into S K
    slices when the time a split saves exceeds its partial-sum traffic (6144^3 on 256 CUs: 2.25 rounds instead of 3)."""
    from k8s_llm_monitor_amd import build, ops
    if build.hipcc() is None:
        pytest.skip("hipcc not installed")
    build.build_all(only=["klm_ops"], verbose=False)
    plan = lambda M, N, K, **kw: ops.gemm_stream_plan(M, N, 2 * K, cus=256, **kw)
    assert plan(8192, 8192, 8192) == {"grid": 256, "rounds": 4, "rest": 0, "split": 0, "rot": 0, "map": 0}
    assert plan(6144, 6144, 6144) == {"grid": 256, "rounds": 2, "rest": 64, "split": 4, "rot": 0, "map": 0}
    assert plan(6144, 6144, 1536)["split"] == 0 and plan(6144, 6144, 1536, mode=2)["split"] == 4       # K too short to pay for the partial sums
    assert plan(2048, 2048, 2048) == {"grid": 64, "rounds": 0, "rest": 64, "split": 0, "rot": 0, "map": 0}       # 64 MB of partial sums cost more than they save
    assert plan(2048, 2048, 2048, mode=2) == {"grid": 256, "rounds": 0, "rest": 64, "split": 4, "rot": 0, "map": 0}
    assert plan(4096, 6144, 4096)["split"] == 2 and plan(4096, 6144, 4096, mode=0)["split"] == 0
    assert plan(6144, 6144, 6144, ws_bytes=64 * 2 * ops.SLOT_BYTES)["split"] == 2 and plan(6144, 6144, 6144, ws_bytes=0)["split"] == 0   # bounded by the scratch
    assert plan(5120, 5120, 5120)["split"] == 0                                                        # 144 left of 256: nothing to split into
    # + 4: the super-block tile order, for 256 workgroups and tile counts that are multiples of 16 both ways, from two rounds on
    assert plan(8192, 8192, 8192, mode=5)["map"] == 1 and plan(12288, 12288, 4096, mode=5)["map"] == 1 and plan(8192, 8192, 8192, mode=1)["map"] == 0
    # the automatic plan takes it once the operands are well past the Infinity Cache
    assert plan(12288, 12288, 12288)["map"] == 1 and plan(16384, 16384, 8192)["map"] == 1 and plan(12288, 12288, 6144)["map"] == 0 and plan(12288, 12288, 12288, mode=0)["map"] == 0
    assert plan(4096, 4096, 4096, mode=5)["map"] == 0 and plan(6144, 6144, 6144, mode=5)["map"] == 0 and plan(8192, 12288, 4096, mode=4)["map"] == 1
    assert plan(8192, 8192, 8192, mode=0, rot=16)["rot"] == 16 and plan(8192, 8192, 8192, mode=0, rot=64)["rot"] == 1
    # every slice gets at least two passes, at most 8 slices, and the slices' passes add up (the kernel's arithmetic, mirrored)
    for M, N, K in ((256, 256, 512), (256, 512, 1024), (768, 768, 1024), (1536, 1536, 12288)):
        p = plan(M, N, K, mode=2)
        P = 2 * K // 256
        if p["split"]:
            S = p["split"]
            assert 2 <= S <= 8 and P // S >= 2
            assert sum(P // S + (1 if s < P % S else 0) for s in range(S)) == P
    assert ops.pick_bf16_variant(8192, 8192, 8192) == 10 and ops.pick_bf16_variant(1024, 1024, 1024) == 14 and ops.pick_bf16_variant(3072, 3072, 192) == 1


def test_default_bf16_kernel_follows_the_cost_model_and_the_measurements():
    """`pick_bf16_variant` prices the persistent 256 x 256 stream against the eight-wave 128 x 128 ring kernel; its plan
    arithmetic is the library's (klm_quad_plan), and its choices are the measured winners of profiles/r3_prof/gemm_small.json."""
    from k8s_llm_monitor_amd import build, ops
    if build.hipcc() is None:
        pytest.skip("hipcc not installed")
    build.build_all(only=["klm_ops"], verbose=False)
    for M, N, K in ((2048, 2048, 2048), (2048, 2048, 8192), (1024, 4096, 4096), (4096, 6144, 4096), (6144, 6144, 6144), (5120, 5120, 5120),
                    (256, 256, 512), (768, 768, 1024), (8192, 8192, 8192), (2560, 2560, 2560)):
        p = ops.gemm_stream_plan(M, N, 2 * K, cus=256)
        assert ops.stream_plan_py(M, N, 2 * K) == (p["rounds"], p["rest"], p["split"]), (M, N, K)
    pick = ops.pick_bf16_variant
    # fewer 256 x 256 tiles than CUs and a short K: the 128 x 128 ring kernel (2048^3: 768 TFLOP/s against the stream's 453)
    assert [pick(n, n, n) for n in (512, 1024, 1536, 2048)] == [14] * 4
    # between the two: 256 x 128 tiles of the same ring kernel (2560^3: 200 of them, one round at 78 %: 43.4 -> 38.8 us)
    assert [pick(n, n, n) for n in (2304, 2560, 2816)] == [16] * 3 and pick(2048, 4096, 4096) == 16 and pick(2176, 2560, 2560) == 17
    assert ops.pick_mx_variant(2560, 2560, 2560, 0, 7) == 17 and ops.pick_mx_variant(2560, 2176, 2560, 0, 7) == 16
    assert pick(1024, 4096, 4096) == 14 and pick(4096, 1024, 1024) == 14 and pick(128, 384, 64) == 14 and pick(384, 128, 64) == 14
    # from 144 tiles on, and on a long K (the split fills the chip), the stream
    assert [pick(n, n, n) for n in (3072, 3584, 4096, 5120, 6144, 8192, 12288)] == [10] * 7
    assert pick(2048, 2048, 8192) == 10 and pick(1024, 1024, 16384) == 10 and pick(2048, 6144, 4096) == 10 and pick(8192, 8192, 1024) == 10
    assert pick(1024, 6144, 4096) == 10 and pick(1536, 4096, 4096) == 10          # 96 tiles, K = 4096: the split stream (54 us) beats both ring forms (56, 63)
    assert pick(2048, 2048, 2048, pitch=4352) == 10                 # pitched operands: only the streams take them
    assert pick(100, 128, 64) == 0 and pick(4096, 4096, 192) == 1 and pick(1024, 1024, 192) == 14
    # fp8 / fp4 go by the bytes of a row: the same rule, their own stream variants
    assert ops.pick_mx_variant(2048, 2048, 2048, 0, 7) == 14 and ops.pick_mx_variant(8192, 8192, 8192, 0, 7) == 7 and ops.pick_mx_variant(4096, 4096, 2048, 0, 6) == 6
    assert ops.pick_mx_variant(2048, 2048, 2048, 4352, 7) == 7 and ops.pick_mx_variant(256, 256, 384, 0, 7) == 14 and ops.pick_mx_variant(4096, 4096, 384, 0, 7) == 1
    assert ops.pick_mx_variant(384, 128, 128, 0, 6) == 14
    # the library carries the same model for native callers (klm_gemm_pick / klm_gemm_auto, the C++ burner): equal over a grid
    import random
    rnd = random.Random(7)
    grid = [(128 * rnd.randint(1, 100), 128 * rnd.randint(1, 100), 128 * rnd.randint(1, 192)) for _ in range(600)]
    grid += [(n, n, kb) for n in range(256, 12289, 256) for kb in (n // 2, n, 2 * n)]
    diff = [(M, N, Kb, ops.pick_tile_kernel(M, N, Kb), ops.native_pick(M, N, Kb)) for M, N, Kb in grid if ops.pick_tile_kernel(M, N, Kb) != ops.native_pick(M, N, Kb)]
    assert not diff, diff[:5]
    # the modelled times are the measured ones to ~15 %
    for (M, N, K), us in (((2048, 2048, 2048), 22.4), ((1024, 1024, 1024), 10.6), ((2048, 2048, 8192), 70.2), ((2560, 2560, 2560), 43.4), ((2304, 2304, 2304), 39.2)):
        assert abs(ops.p128_time_us(M, N, 2 * K) - us) / us < 0.15
    for (M, N, K), us in (((2560, 2560, 2560), 38.8), ((2048, 4096, 4096), 62.7), ((3072, 3072, 3072), 86.8), ((8192, 8192, 8192), 952.0)):
        assert abs(ops.p256_time_us(M, N, 2 * K) - us) / us < 0.15
    for (M, N, K), us in (((4096, 4096, 4096), 94.8), ((3072, 3072, 3072), 57.5), ((8192, 8192, 8192), 687.0)):
        assert abs(ops.stream_time_us(M, N, 2 * K) - us) / us < 0.15


def test_repack_rule_and_camping_pitch_are_pure_and_selective():
    from k8s_llm_monitor_amd import ops
    assert ops.repack_worthwhile(12288, 12288, 24576) and ops.repack_worthwhile(8192, 8192, 40960) and ops.repack_worthwhile(6144, 6144, 24576)
    assert not ops.repack_worthwhile(8192, 8192, 16384)          # a power of two: the channel hash handles it
    assert not ops.repack_worthwhile(8192, 8192, 20480)          # camps mildly (7 %): a copy costs more
    assert not ops.repack_worthwhile(4096, 4096, 24576)          # too few rows to pay for the copy
    assert not ops.repack_worthwhile(12288, 12288, 24576, dtype_bytes=1)
    assert ops.camping_pitch(24576) and ops.camping_pitch(20480) and not ops.camping_pitch(16384)
