"""Is the row-major skinny kernel at K = 16384 held back by the power-of-two row stride (16 rows of a tile 32 KiB apart)?
Same shapes with K = 17408 (34 KiB rows): if the rate per byte jumps, padding the row stride is the fix."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.ops import llm  # noqa: E402


def t(fn, iters=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N in (2304, 13312, 16384):
    for K in (16384, 17408, 8192, 8704, 4096, 4352):
        ncopy = max(2, int(1.5e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
        x = torch.randn(1, K, device="cuda", dtype=torch.bfloat16)
        i = [0]
        line = f"N={N:6d} K={K:6d}"
        for waves in (4, 8, 16):
            def f():
                i[0] = (i[0] + 1) % ncopy
                llm.skinny_linear(x, ws[i[0]], rms=True, waves=waves, row_tiles=1)
            a = t(f)
            line += f" | w{waves} {a:6.1f}us {N * K * 2 / a / 1e6:4.2f} TB/s"
        print(line, flush=True)
        del ws
