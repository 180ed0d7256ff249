"""Soak of round 3's kernels: every arm runs back to back for KLM_SOAK_S seconds, every 100th result is compared bit for bit
with the first one (all kernels are deterministic). Arms: the eight-wave ring kernel on its three tile shapes in the three
formats (odd and even K tile counts, one and several rounds of workgroups), the persistent stream with a split last round
(partial sums + reduce kernel), the one-round stream with streamed C, the super-block tile order, the repacked path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
seconds = float(os.environ.get("KLM_SOAK_S", "6"))
g = torch.Generator(device=dev); g.manual_seed(33)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/gemm_soak_r3.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)


def operands(fmt, M, N, K):
    if fmt == "fp4":
        return (torch.randint(0, 256, (M, K // 2), device=dev, generator=g, dtype=torch.uint8), torch.randint(0, 256, (N, K // 2), device=dev, generator=g, dtype=torch.uint8))
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    return (a, b) if fmt == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))


FN = {"bf16": ops.gemm_bf16, "fp8": ops.gemm_fp8, "fp4": ops.gemm_fp4}
arms = []
for fmt in ("bf16", "fp8", "fp4"):
    kmul = {"bf16": 1, "fp8": 2, "fp4": 4}[fmt]
    arms += [(f"{fmt} v14 2048x2048 x 2048B rows", fmt, (2048, 2048, 1024 * kmul), 14), (f"{fmt} v14 odd K tiles, 1.5 rounds", fmt, (2560, 2560, 1088 * kmul), 14),
             (f"{fmt} v16 2560^2, odd K tiles", fmt, (2560, 2560, 1344 * kmul), 16), (f"{fmt} v17 2304x2816", fmt, (2304, 2816, 1024 * kmul), 17),
             (f"{fmt} v16 several rounds", fmt, (4096, 4096, 512 * kmul), 16)]
arms += [("bf16 default 6144^3 (split round + reduce)", "bf16", (6144, 6144, 6144), None), ("fp8 default 6144^3", "fp8", (6144, 6144, 6144), None),
         ("fp4 default 4096x6144x8192 (split)", "fp4", (4096, 6144, 8192), None),
         ("bf16 default 4096^3 (one round, streamed C)", "bf16", (4096, 4096, 4096), None), ("fp8 default 3072^3 (one round)", "fp8", (3072, 3072, 3072), None),
         ("bf16 v15 super-block order 8192^2 x 2048", "bf16", (8192, 8192, 2048), 15), ("bf16 default 6144x6144x12288 (repack)", "bf16", (6144, 6144, 12288), None)]
out = {}
for name, fmt, (M, N, K), v in arms:
    x, y = operands(fmt, M, N, K)
    fn = FN[fmt]
    ref = fn(x, y, variant=v).clone()
    c = torch.empty_like(ref)
    t0 = time.perf_counter(); launches = checks = bad = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(100):
            fn(x, y, out=c, variant=v)
        launches += 100
        checks += 1
        if not torch.equal(c, ref):
            bad += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[name] = {"shape": [M, N, K], "variant": v, "launches": launches, "seconds": round(dt, 1), "checks": checks, "mismatches": bad,
                 "tflops_wall": round(2.0 * M * N * K * launches / dt / 1e12, 1)}
    print(name, out[name], flush=True)
    json.dump(out, open(_o, "w"), indent=1)
    del x, y, ref, c
bad = {k: v["mismatches"] for k, v in out.items() if v["mismatches"]}
print("MISMATCHES" if bad else "all arms deterministic", bad)
sys.exit(3 if bad else 0)
