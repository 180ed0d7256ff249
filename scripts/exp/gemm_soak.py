"""Soak of the default (persistent) GEMM streams: each dtype runs back to back for `seconds`, every 200th result is
compared bit for bit with the first one (the kernels are deterministic), the monitor's sampler runs beside it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
seconds = float(os.environ.get("KLM_SOAK_S", "40"))
g = torch.Generator(device=dev); g.manual_seed(21)
n = 8192
out = {}
a = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
arms = {"bf16": (ops.gemm_bf16, a, b), "fp8": (ops.gemm_fp8, a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)),
        "fp4": (ops.gemm_fp4, torch.randint(0, 256, (n, n // 2), device=dev, generator=g, dtype=torch.uint8),
                torch.randint(0, 256, (n, n // 2), device=dev, generator=g, dtype=torch.uint8)),
        "bf16_12288_padded": (ops.gemm_bf16, None, None)}
pa = ops.padded_operand(n, 12288, torch.bfloat16, dev); pa.copy_((torch.rand(n, 12288, device=dev, generator=g) * 2 - 1).to(torch.bfloat16))
pb = ops.padded_operand(n, 12288, torch.bfloat16, dev); pb.copy_((torch.rand(n, 12288, device=dev, generator=g) * 2 - 1).to(torch.bfloat16))
arms["bf16_12288_padded"] = (ops.gemm_bf16, pa, pb)
for name, (fn, x, y) in arms.items():
    ref = fn(x, y).clone()
    c = torch.empty_like(ref)
    t0 = time.perf_counter(); launches = checks = bad = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            fn(x, y, out=c)
        launches += 200
        checks += 1
        if not torch.equal(c, ref):
            bad += 1
        print(f"{name}: {launches} launches, {time.perf_counter() - t0:.0f} s, mismatches {bad}", flush=True) if checks % 20 == 0 else None
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    K = x.shape[1] * (2 if name == "fp4" else 1)
    out[name] = {"launches": launches, "seconds": round(dt, 1), "checks": checks, "mismatches": bad,
                 "tflops_wall": round(2.0 * n * n * K * launches / dt / 1e12, 1)}
    print(name, out[name], flush=True)
    assert bad == 0
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_soak.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
