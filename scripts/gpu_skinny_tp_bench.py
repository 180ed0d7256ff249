"""Skinny linear kernel on the shapes ONE rank of llama3-405b / tp8 streams per decode step: waves x row tiles sweep,
row-major and tiled weights, against torch.matmul (hipBLASLt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.ops import llm  # noqa: E402


def t(fn, iters=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


SETS = {
    "405b-tp8": [("qkv", 2304, 16384, True, "none"), ("wo", 16384, 2048, False, "residual"),
                 ("gate_up", 13312, 16384, True, "swiglu"), ("down", 16384, 6656, False, "residual"),
                 ("lm_head", 16032, 16384, True, "none")],
    "8b": [("qkv", 6144, 4096, True, "none"), ("wo", 4096, 4096, False, "residual"),
           ("gate_up", 28672, 4096, True, "swiglu"), ("down", 4096, 14336, False, "residual"),
           ("lm_head", 128256, 4096, True, "none")],
    "70b": [("qkv", 10240, 8192, True, "none"), ("wo", 8192, 8192, False, "residual"),
            ("gate_up", 57344, 8192, True, "swiglu"), ("down", 8192, 28672, False, "residual")],
}
Ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 8]
shapes = SETS[sys.argv[2] if len(sys.argv) > 2 else "405b-tp8"]
only_rowmajor = len(sys.argv) > 3 and sys.argv[3] == "R"
for M in Ms:
    for name, N, K, rms, epi in shapes:
        ncopy = max(2, int(1.5e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
        wts = [llm.tile_weights(w) for w in ws]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == "residual" else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        i = [0]
        gb = N * K * 2 / 1e9
        line = f"M={M:2d} {name:8s} N={N:6d} K={K:5d} {gb * 1e3:6.1f} MB"
        for tiled, arr in (((False, ws),) if only_rowmajor else ((False, ws), (True, wts))):
            for waves, rt in ((0, 0), (4, 1), (8, 1), (16, 1), (4, 4), (8, 4)):
                def ours():
                    i[0] = (i[0] + 1) % ncopy
                    llm.skinny_linear(x, arr[i[0]], rms=rms, epilogue=epi, residual=res, out=res, waves=waves, row_tiles=rt, tiled=tiled)
                try:
                    a = t(ours)
                    line += f" | {'T' if tiled else 'R'} w{waves}r{rt} {a:6.1f}us {gb / a * 1e3:4.2f}"
                except Exception as e:
                    line += f" | {'T' if tiled else 'R'} w{waves}r{rt} n/a"

        def lib():
            i[0] = (i[0] + 1) % ncopy
            torch.matmul(x, ws[i[0]].t(), out=out)
        b = t(lib)
        print(line + f" | hipBLASLt {b:6.1f}us {gb / b * 1e3:4.2f} TB/s", flush=True)
        del ws, wts
