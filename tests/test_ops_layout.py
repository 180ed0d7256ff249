"""Operand layout helpers of the GEMM wrappers (CPU): which row pitches get padded, what a padded operand looks like, and
that the wrappers refuse what the kernels cannot take before anything is launched."""
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
