"""8192 x 8192 x K for K around 12288: is the drop at 12288^3 a property of the row stride (24 KiB)?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
out = {}
M = N = int(os.environ.get("KLM_MN", "8192"))
for K in (8192, 10240, 11264, 12032, 12288, 12544, 13312, 14336, 16384, 20480, 24576):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bt = b.t()
    cands = {"v4": lambda: ops.gemm_bf16(a, b, out=c, variant=4), "v8": lambda: ops.gemm_bf16(a, b, out=c, variant=8),
             "v2_ring": lambda: ops.gemm_bf16(a, b, out=c, variant=2), "hipblaslt": lambda: torch.matmul(a, bt, out=c)}
    flop = 2.0 * M * N * K
    runs = {k: [] for k in cands}
    for r in range(4):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(flop * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    out[str(K)] = {k: sorted(v)[len(v) // 2] for k, v in runs.items()}
    print(M, K, {k: round(v) for k, v in out[str(K)].items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_stride_sweep.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
