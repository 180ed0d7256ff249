"""The persistent forms of the hand-scheduled GEMM streams (bf16 variant 8, fp8 4, fp4 3): bit-equal to the one-tile-per-
workgroup streams on shapes that give a workgroup 1, 2, 3 and 4 output tiles (even and uneven), then timing at 8192^3
and along K next to those streams and the vendor kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(11)
out = {"equal": {}, "us": {}}


def operands(M, N, K):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    return a, b


def check(M, N, K, grid):
    ops.gemm_persistent_grid(grid)
    a, b = operands(M, N, K)
    r = {}
    r["bf16"] = bool(torch.equal(ops.gemm_bf16(a, b, variant=8), ops.gemm_bf16(a, b, variant=4)))
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    if K % 256 == 0 and K >= 512:
        r["fp8"] = bool(torch.equal(ops.gemm_fp8(a8, b8, variant=4), ops.gemm_fp8(a8, b8, variant=2)))
        a4 = torch.randint(0, 256, (M, K), device=dev, generator=g, dtype=torch.uint8)
        b4 = torch.randint(0, 256, (N, K), device=dev, generator=g, dtype=torch.uint8)
        r["fp4"] = bool(torch.equal(ops.gemm_fp4(a4, b4, variant=3), ops.gemm_fp4(a4, b4, variant=2)))
    torch.cuda.synchronize()
    ops.gemm_persistent_grid(0)
    out["equal"][f"{M}x{N}x{K}/grid{grid}"] = r
    print(M, N, K, "grid", grid, r, flush=True)
    return all(r.values())


ok = True
for shape, grid in (((256, 256, 512), 0), ((512, 512, 512), 1), ((512, 512, 256), 3), ((768, 512, 1024), 4), ((1024, 1024, 512), 5),
                    ((4352, 4352, 512), 0), ((8192, 8192, 1024), 0), ((2048, 8192, 2048), 0)):
    ok = check(*shape, grid) and ok
if not ok:
    json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/persistent_check.json", "w"), indent=1)
    sys.exit("MISMATCH")

n = 8192
one = torch.ones((), device=dev)
for K in (1024, 2048, 8192):
    a, b = operands(n, n, K)
    c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    a4 = torch.randint(0, 256, (n, K), device=dev, generator=g, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (n, K), device=dev, generator=g, dtype=torch.uint8)
    bt, b8t = b.t(), b8.t()
    cands = {"bf16_asm": lambda: ops.gemm_bf16(a, b, out=c, variant=4), "bf16_persistent": lambda: ops.gemm_bf16(a, b, out=c, variant=8),
             "bf16_hipblaslt": lambda: torch.matmul(a, bt, out=c),
             "fp8_asm": lambda: ops.gemm_fp8(a8, b8, out=c, variant=2), "fp8_persistent": lambda: ops.gemm_fp8(a8, b8, out=c, variant=4),
             "fp8_scaled_mm": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16),
             "fp4_asm": lambda: ops.gemm_fp4(a4, b4, out=c, variant=2), "fp4_persistent": lambda: ops.gemm_fp4(a4, b4, out=c, variant=3)}
    runs = {k: [] for k in cands}
    for r in range(6):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(e0.elapsed_time(e1) / 20 * 1e3)
    out["us"][str(K)] = {k: sorted(v)[len(v) // 2] for k, v in runs.items()}
    print("K", K, {k: round(v, 1) for k, v in out["us"][str(K)].items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/persistent_check.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
