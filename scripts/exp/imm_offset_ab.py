"""(Historic: the bodies were removed again after this measurement -- same bits, no gain; profiles/r2_prof/imm_offset_ab.json.)
LDS-DMA pieces addressed through the instruction's immediate offset (M0 written for 2 of 8 pieces per operand instead of 8):
same bits? faster? bf16 variant 11 vs 8, fp4 variant 6 vs 3."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(6)
out = {}
for n, K in ((8192, 8192), (8192, 2048), (1024, 1536)):
    a = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a4 = torch.randint(0, 256, (n, K // 2), device=dev, generator=g, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (n, K // 2), device=dev, generator=g, dtype=torch.uint8)
    c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    eq = {"bf16": bool(torch.equal(ops.gemm_bf16(a, b, variant=11), ops.gemm_bf16(a, b, variant=8))),
          "fp4": bool(torch.equal(ops.gemm_fp4(a4, b4, variant=6), ops.gemm_fp4(a4, b4, variant=3)))}
    print(n, K, "equal:", eq, flush=True)
    if not all(eq.values()):
        out[f"{n}x{K}"] = {"equal": eq}
        break
    cands = {"bf16": lambda: ops.gemm_bf16(a, b, out=c, variant=8), "bf16_imm": lambda: ops.gemm_bf16(a, b, out=c, variant=11),
             "fp4": lambda: ops.gemm_fp4(a4, b4, out=c, variant=3), "fp4_imm": lambda: ops.gemm_fp4(a4, b4, out=c, variant=6)}
    runs = {k: [] for k in cands}
    for r in range(7):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(2.0 * n * n * K * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    out[f"{n}x{K}"] = {"equal": eq, "tflops": {k: sorted(v)[len(v) // 2] for k, v in runs.items()}}
    print(n, K, {k: round(v, 1) for k, v in out[f"{n}x{K}"]["tflops"].items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/imm_offset_ab.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
