"""Mid-batch linear kernel (llm_mid.hip) against torch.matmul (hipBLASLt) on the Llama-3-8B decode shapes:
sweep of K splits x waves x row tiles. Times include the epilogue kernel (plain sum)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.ops import llm  # noqa: E402


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


SETS = {"8b": [("qkv", 6144, 4096), ("wo", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)],
        "70b": [("qkv", 10240, 8192), ("wo", 8192, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)]}
Ms = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 64, 128]
shapes = SETS[sys.argv[2] if len(sys.argv) > 2 else "8b"]
full = len(sys.argv) > 3 and sys.argv[3] == "sweep"
fp8 = len(sys.argv) > 4 and sys.argv[4] == "fp8"       # e4m3 tiled weight copies (half the bytes; TB/s counts fp8 bytes)
for M in Ms:
    for name, N, K in shapes:
        ncopy = max(2, int(1.5e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(ncopy)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        i = [0]
        gb = N * K * (1 if fp8 else 2) / 1e9
        if fp8:
            q = [llm.quantize_fp8_rows(w) for w in ws]
            w8 = [(llm.tile_weights(a), sc) for a, sc in q]
            del q
        line = f"M={M:3d} {name:8s}"
        best = None
        cfgs = [(0, 0, 0)]
        if full:
            cfgs += [(ks, nw, rt) for nw in (4, 7, 8) for rt in (1, 2) for ks in (1, 2, 4, 8, 16) if N % (nw * rt * 16) == 0 and (K // (128 * (4 if rt == 1 else 2))) % ks == 0]
        for ks, nw, rt in cfgs:
            def ours():
                i[0] = (i[0] + 1) % ncopy
                if fp8:
                    llm.mid_linear(x, w8[i[0]][0], weight_scale=w8[i[0]][1], out=out, ksplit=ks, waves=nw, row_tiles=rt)
                else:
                    llm.mid_linear(x, ws[i[0]], out=out, ksplit=ks, waves=nw, row_tiles=rt)
            a = t(ours)
            if (ks, nw, rt) == (0, 0, 0):
                line += f" | default{llm.mid_plan(M, N, K)} {a:6.1f}us {gb / a * 1e3:4.2f}"
            elif best is None or a < best[0]:
                best = (a, ks, nw, rt)
        if best:
            line += f" | best ks{best[1]} w{best[2]} r{best[3]} {best[0]:6.1f}us {gb / best[0] * 1e3:4.2f}"

        def lib():
            i[0] = (i[0] + 1) % ncopy
            torch.matmul(x, ws[i[0]].t(), out=out)
        b = t(lib)
        print(line + f" | hipBLASLt {b:6.1f}us {gb / b * 1e3:4.2f} TB/s", flush=True)
        del ws
