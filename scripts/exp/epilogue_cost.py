"""What an output tile's epilogue costs the persistent bf16 stream: the kernel, without its stores, without its epilogue
(diagnostic variants 48 / 49: C is not written), at 8192 x 8192 x K. Microseconds per GEMM; fit t = a + b K."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
n = 8192
out = {}
ref_ok = {}
for K in (1024, 4096, 8192, 16384):
    a = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    bt = b.t()
    cands = {"persistent": lambda: ops.gemm_bf16(a, b, out=c, variant=8), "no_stores": lambda: ops.gemm_bf16(a, b, out=c, variant=48),
             "no_epilogue": lambda: ops.gemm_bf16(a, b, out=c, variant=49), "st_nt": lambda: ops.gemm_bf16(a, b, out=c, variant=50),
             "st_sc1_nt": lambda: ops.gemm_bf16(a, b, out=c, variant=51), "st_sc0_sc1_nt": lambda: ops.gemm_bf16(a, b, out=c, variant=52), "per_tile": lambda: ops.gemm_bf16(a, b, out=c, variant=4),
             "hipblaslt": lambda: torch.matmul(a, bt, out=c), "fill_c": lambda: ops.hbm_fill(c)}
    want = ops.gemm_bf16(a, b, variant=4)
    ref_ok[str(K)] = {v: bool(torch.equal(ops.gemm_bf16(a, b, variant=v), want)) for v in (8, 50, 51, 52)}
    print("equal:", ref_ok[str(K)], flush=True)
    runs = {k: [] for k in cands}
    for r in range(5):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(e0.elapsed_time(e1) / 20 * 1e3)
    out[str(K)] = {k: sorted(v)[len(v) // 2] for k, v in runs.items()}
    print(K, {k: round(v, 1) for k, v in out[str(K)].items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/epilogue_cost.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
