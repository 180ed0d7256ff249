"""How fast can the LIBRARY be on the decode GEMMs at batch 32 - 128 (Llama-3-8B shapes)? Default heuristic pick
against PyTorch TunableOp's search over every rocBLAS / hipBLASLt solution. Sets the bar for a hand-written kernel."""
import os
import sys
import time

import torch

shapes = [("qkv", 6144, 4096), ("wo", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]
Ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64]


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


res = {}
for tuned in (False, True):
    if tuned:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(30)
        torch.cuda.tunable.set_max_tuning_iterations(20)
        torch.cuda.tunable.set_filename("/tmp/tunableop.csv")
    for M in Ms:
        for name, N, K in shapes:
            ncopy = max(2, int(1.5e9 // (N * K * 2)))
            ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            i = [0]

            def f():
                i[0] = (i[0] + 1) % ncopy
                torch.matmul(x, ws[i[0]].t(), out=out)
            t0 = time.time()
            us = t(f)
            res[(tuned, M, name)] = us
            print(f"{'tuned  ' if tuned else 'default'} M={M:3d} {name:8s} {us:7.1f} us {N * K * 2 / us / 1e6:5.2f} TB/s  (took {time.time() - t0:.1f} s)", flush=True)
            del ws
