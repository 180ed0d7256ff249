"""hipBLASLt on the Llama-3-8B decode GEMM shapes at a given number of rows (what the engine's library path pays)."""
import sys

import torch

shapes = [("qkv", 6144, 4096), ("wo", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]
Ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [160, 192, 256]


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


for M in Ms:
    tot = 0.0
    line = f"M={M:3d}"
    for name, N, K in shapes:
        ncopy = max(2, int(1.5e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        i = [0]

        def f():
            i[0] = (i[0] + 1) % ncopy
            torch.matmul(x, ws[i[0]].t(), out=out)
        us = t(f)
        tot += us
        line += f" | {name} {us:6.1f}us {N * K * 2 / us / 1e6:4.2f} TB/s"
        del ws
    print(line + f" | sum {tot:6.1f}us", flush=True)
