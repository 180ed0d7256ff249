"""One round of the chip (4096 x 4096 = 256 tiles of 256 x 256) at growing K: time = fixed + slope * K. The fixed part
(launch, cold prologue, epilogue, tail) of this repo's streams next to the vendor library's."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/gemm_fixed_cost.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
one = torch.ones((), device=dev)
M = N = 4096
out = {"rows": {}, "fit": {}}
KS = (512, 1024, 2048, 3072, 4096, 6144, 8192)


def us(fn, iters=200):
    vals = []
    for r in range(5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        if r:
            vals.append(e0.elapsed_time(e1) * 1e3 / iters)
    return round(sorted(vals)[1], 2)


for K in KS:
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bt, b8t = b.t(), b8.t()
    row = {"bf16_tile_per_wg": us(lambda: ops.gemm_bf16(a, b, out=c, variant=4)), "bf16_persistent": us(lambda: ops.gemm_bf16(a, b, out=c, variant=8)),
           "hipblaslt": us(lambda: torch.matmul(a, bt, out=c)),
           "fp8_tile_per_wg": us(lambda: ops.gemm_fp8(a8, b8, out=c, variant=2)), "fp8_persistent": us(lambda: ops.gemm_fp8(a8, b8, out=c, variant=4)),
           "scaled_mm": us(lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)),
           "empty_launch_pair": us(lambda: ops.spin(1, 256, dev))}
    out["rows"][str(K)] = row
    print(K, json.dumps(row), flush=True)
for name in next(iter(out["rows"].values())):
    xs = [float(k) for k in out["rows"]]
    ys = [out["rows"][k][name] for k in out["rows"]]
    n = len(xs); mx = sum(xs) / n; my = sum(ys) / n
    slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    out["fit"][name] = {"fixed_us": round(my - slope * mx, 2), "us_per_1024_K": round(slope * 1024, 2)}
print(json.dumps(out["fit"], indent=1))
json.dump(out, open(_o, "w"), indent=1)
