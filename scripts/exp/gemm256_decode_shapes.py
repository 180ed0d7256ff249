"""The 256 x 256 burner GEMM (no K split) and the 128 x 128 one on the Llama-3-8B decode shapes at 256 rows, next to
hipBLASLt: is a split-K variant of our kernel a candidate for decode at batch 129 - 256?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd import ops  # noqa: E402

shapes = [("qkv", 6144, 4096), ("wo", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]


def t(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = 256
for name, N, K in shapes:
    ncopy = max(2, int(1.5e9 // (N * K * 2)))
    ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    i = [0]

    def lib():
        i[0] = (i[0] + 1) % ncopy
        torch.matmul(x, ws[i[0]].t(), out=out)

    def k256():
        i[0] = (i[0] + 1) % ncopy
        ops.gemm_bf16(x, ws[i[0]], out=out, variant=1)

    def k128():
        i[0] = (i[0] + 1) % ncopy
        ops.gemm_bf16(x, ws[i[0]], out=out, variant=0)

    # a K split by hand: the 256 kernel on K/2 (what each half of a 2-way split would cost, without the partial traffic)
    xs, wh = x[:, : K // 2].contiguous(), [w[:, : K // 2].contiguous() for w in ws[:2]]

    def k256_half():
        i[0] = (i[0] + 1) % 2
        ops.gemm_bf16(xs, wh[i[0]], out=out, variant=1)

    print(f"{name:8s} N={N:6d} K={K:6d} blocks256={N // 256:4d}  hipBLASLt {t(lib):6.1f} us | 256x256 {t(k256):6.1f} | "
          f"128x128 ({(N // 128) * 2} blocks) {t(k128):6.1f} | 256x256 on K/2 {t(k256_half):6.1f}", flush=True)
    del ws, wh
