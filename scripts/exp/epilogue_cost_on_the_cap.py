"""What the epilogue and its stores cost the persistent bf16 chain stream IN TIME on the power cap (the diagnostic streams 48 / 49:
the same kernel without its stores / without its whole epilogue, C not written), at 8192 x 8192 x K, alternating with the normal
stream, medians. Needs a -DKLM_GEMM_DIAG build at _native/lib/exp/libklm_ops_diag.so (scripts/exp/README.md)."""
import json, os, sys
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from k8s_llm_monitor_amd import ops
ops._LIB_PATH = Path(ROOT) / "k8s_llm_monitor_amd/_native/lib/exp/libklm_ops_diag.so"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
n = 8192
out = {}
for K in (2048, 8192):
    a = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    cands = {"persistent": lambda: ops.gemm_bf16(a, b, out=c, variant=8), "no_stores": lambda: ops.gemm_bf16(a, b, out=c, variant=48),
             "no_epilogue": lambda: ops.gemm_bf16(a, b, out=c, variant=49)}
    runs = {k: [] for k in cands}
    iters = max(10, int(8e13 / (2.0 * n * n * K)))
    for r in range(7):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(e0.elapsed_time(e1) / iters * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in runs.items()}
    out[str(K)] = {"us": {k: round(v, 1) for k, v in med.items()}, "stores_cost_pct": round((med["persistent"] / med["no_stores"] - 1) * 100, 2),
                   "epilogue_cost_pct": round((med["persistent"] / med["no_epilogue"] - 1) * 100, 2)}
    print(K, out[str(K)], flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/exp/epilogue_cost_on_the_cap.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
