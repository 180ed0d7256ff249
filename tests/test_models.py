"""The flagship validation step (models/linear_step.py): operand roles of fwd / dgrad / wgrad."""
import pytest
import torch

from k8s_llm_monitor_amd.models import LinearLayerStep


def cpu_gemm(a, b, out):
    out.copy_(a.float() @ b.float().t())


def test_three_gemms_are_the_linear_layer_training_step():
    s = LinearLayerStep(torch, None, None, torch.device("cpu"), world=1, size=64, gemm=cpu_gemm)
    s.step()
    x, w, dy = s.x.float(), s.w.float(), s.dy.float()
    torch.testing.assert_close(s.y.float(), (x @ w.t()).to(torch.bfloat16).float())            # Y = X W^T
    torch.testing.assert_close(s.dx.float(), (dy @ w).to(torch.bfloat16).float())              # dX = dY W
    torch.testing.assert_close(s.c[0].float(), (dy.t() @ x).to(torch.bfloat16).float())        # dW = dY^T X
    assert s.flops == 6 * 64 ** 3 and s.i == 1 and s.allreduce_bytes == 0
    s.step()
    assert s.i == 2 and torch.equal(s.c[1], s.c[0])                                            # double-buffered dW


@pytest.mark.gpu
def test_step_on_the_mfma_kernel_matches_fp32():
    from k8s_llm_monitor_amd import ops
    dev = torch.device("cuda:0")
    s = LinearLayerStep(torch, None, ops, dev, world=1, size=1024)
    s.step()
    torch.cuda.synchronize()
    x, w, dy = s.x.float(), s.w.float(), s.dy.float()
    for got, ref in ((s.y, x @ w.t()), (s.dx, dy @ w), (s.c[0], dy.t() @ x)):
        torch.testing.assert_close(got.float(), ref, rtol=1.6e-2, atol=0.35)
