"""fp8 / fp4 persistent streams on the block-scaled MFMA with unit scales (two instruction words per MFMA) vs the plain
v_mfma_f32_16x16x128_f8f6f4 (one word): same bits? faster?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(9)
out = {}
one = torch.ones((), device=dev)
for n, K in ((8192, 8192), (8192, 2048), (4096, 4096)):
    a8 = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.float8_e4m3fn)
    b8 = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.float8_e4m3fn)
    a4 = torch.randint(0, 256, (n, K // 2), device=dev, generator=g, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (n, K // 2), device=dev, generator=g, dtype=torch.uint8)
    c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    eq = {"fp8": bool(torch.equal(ops.gemm_fp8(a8, b8, variant=6), ops.gemm_fp8(a8, b8, variant=4)) and torch.equal(ops.gemm_fp8(a8, b8, variant=2), ops.gemm_fp8(a8, b8, variant=1))),
          "fp4": bool(torch.equal(ops.gemm_fp4(a4, b4, variant=5), ops.gemm_fp4(a4, b4, variant=3)) and torch.equal(ops.gemm_fp4(a4, b4, variant=2), ops.gemm_fp4(a4, b4, variant=1)))}
    print(n, K, "equal to the ring kernels:", eq, flush=True)
    b8t = b8.t()
    cands = {"fp8_scaled": lambda: ops.gemm_fp8(a8, b8, out=c, variant=6), "fp8_plain": lambda: ops.gemm_fp8(a8, b8, out=c, variant=4),
             "fp8_scaled_mm": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16),
             "fp4_scaled": lambda: ops.gemm_fp4(a4, b4, out=c, variant=5), "fp4_plain": lambda: ops.gemm_fp4(a4, b4, out=c, variant=3)}
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
                runs[name].append(2.0 * n * n * K * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    out[f"{n}x{n}x{K}"] = {"equal": eq, "tflops": {k: sorted(v)[len(v) // 2] for k, v in runs.items()}}
    print(n, K, {k: round(v) for k, v in out[f"{n}x{n}x{K}"]["tflops"].items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mx_noscale.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
