"""K-split sweep of the tiled GEMM + its epilogue on the Llama-3-8B shapes at 256 and 192 rows (planner input)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.ops import llm  # noqa: E402

shapes = [("qkv", 6144, 4096, "none"), ("wo", 4096, 4096, "res_rms"), ("gate_up", 28672, 4096, "swiglu"), ("down", 4096, 14336, "res_rms")]


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M in (256, 192):
    for name, N, K, kind in shapes:
        ncopy = max(2, int(1.2e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if kind == "res_rms" else None
        line = f"M={M} {name:8s} plan ks={llm.wide_plan(M, N, K)}:"
        for ks in (1, 2, 4, 8, 16, 32):
            if K % (ks * 64) or K // ks < 128:
                continue
            i = [0]

            def f():
                i[0] = (i[0] + 1) % ncopy
                llm.mid_linear(x, ws[i[0]], kind=kind, residual=res, ksplit=ks, wide=True)
            line += f"  ks{ks}: {t(f):6.1f}"
        print(line, flush=True)
        del ws
