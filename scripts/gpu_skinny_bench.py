"""Skinny linear kernel vs torch.matmul (hipBLASLt) on the Llama-3-8B decode shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.ops import llm  # noqa: E402


def t(fn, iters=50):
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


shapes = [("qkv", 6144, 4096, True, "none"), ("wo", 4096, 4096, False, "residual"),
          ("gate_up", 28672, 4096, True, "swiglu"), ("down", 4096, 14336, False, "residual"),
          ("lm_head", 128256, 4096, True, "none")]
Ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 16]
for M in Ms:
    for name, N, K, rms, epi in shapes:
        # several weight copies so that the 256 MB of L2 / MALL does not serve the stream
        ncopy = max(2, int(1.5e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == "residual" else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        i = [0]
        gb = N * K * 2 / 1e9
        line = f"M={M:2d} {name:8s} N={N:6d} K={K:5d}"
        for waves, rt in ((0, 0),):
            def ours():
                i[0] = (i[0] + 1) % ncopy
                llm.skinny_linear(x, ws[i[0]], rms=rms, epilogue=epi, residual=res, out=res, waves=waves, row_tiles=rt)
            a = t(ours)
            line += f" | w{waves}r{rt}: {a:6.1f} us {gb / a * 1e3:4.2f}"

        wts = [llm.tile_weights(wt) for wt in ws]
        for waves, rt in ((0, 0), (16, 1), (4, 1)):
            def ourst():
                i[0] = (i[0] + 1) % ncopy
                llm.skinny_linear(x, wts[i[0]], rms=rms, epilogue=epi, residual=res, out=res, waves=waves,
                                  row_tiles=rt, tiled=True)
            a = t(ourst)
            line += f" | tiled w{waves}r{rt}: {a:6.1f} us {gb / a * 1e3:4.2f}"
        del wts
        w8s = [llm.quantize_fp8_rows(wt) for wt in ws]
        w8t = [(llm.tile_weights(a8), sc) for a8, sc in w8s]
        for name8, arr, tl in (("fp8", w8s, False), ("fp8t", w8t, True)):
            for waves, rt in ((0, 0), (4, 1)):
                def ours8():
                    i[0] = (i[0] + 1) % ncopy
                    llm.skinny_linear(x, arr[i[0]][0], rms=rms, epilogue=epi, residual=res, out=res, waves=waves,
                                      row_tiles=rt, weight_scale=arr[i[0]][1], tiled=tl)
                a = t(ours8)
                line += f" | {name8} w{waves}r{rt}: {a:6.1f} us {gb / 2 / a * 1e3:4.2f}"

        def lib():
            i[0] = (i[0] + 1) % ncopy
            torch.matmul(x, ws[i[0]].t(), out=out)
        b = t(lib)
        print(line + f" | hipBLASLt {b:6.1f} us {gb / b * 1e3:4.2f} TB/s", flush=True)
        del ws
