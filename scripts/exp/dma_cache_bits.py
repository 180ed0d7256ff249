"""(Historic: needs the _SC0 / _SC1 / _NT / _SC1NT bodies, which the generator no longer emits -- re-add them to BODIES and the
dispatchers (variants 44..47) to rerun.) Cache-policy bits on the GEMM's LDS-DMA (sc0 / sc1 / nt / sc1 nt): every operand byte is used by exactly one wave of
a CU, so the vector L1 holds nothing worth keeping. Right results in every arm; interleaved timing at 8192^3."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
n = 8192
g = torch.Generator(device=dev); g.manual_seed(7)
a = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
bt = b.t()
ref = ops.gemm_bf16(a, b, variant=4)
arms = {"plain": 4, "sc0": 44, "sc1": 45, "nt": 46, "sc1_nt": 47, "persistent": 8}
ok = {k: bool(torch.equal(ops.gemm_bf16(a, b, variant=v), ref)) for k, v in arms.items()}
print("equal:", ok, flush=True)
cands = {k: (lambda v=v: ops.gemm_bf16(a, b, out=c, variant=v)) for k, v in arms.items()}
cands["hipblaslt"] = lambda: torch.matmul(a, bt, out=c)
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
            runs[name].append(2.0 * n ** 3 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
out = {"equal": ok, "tflops": {k: {"median": sorted(v)[len(v) // 2], "min": min(v), "max": max(v)} for k, v in runs.items()}}
for k, v in out["tflops"].items():
    print(k, {x: round(y, 1) for x, y in v.items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/dma_cache_bits.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
