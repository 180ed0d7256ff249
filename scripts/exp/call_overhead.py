"""Host-side cost of one `ops.gemm_*` call next to `torch.matmul`: a tiny problem launched 5000 times without a sync (the GPU
is never the limit), then a cProfile of the same loop."""
import cProfile, io, json, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
a = torch.randn(256, 256, device=dev, dtype=torch.bfloat16); b = torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
c = torch.empty(256, 256, device=dev, dtype=torch.bfloat16)
bt = b.t()
big = torch.randn(4096, 512, device=dev, dtype=torch.bfloat16)
out = {}
def loop(fn, n=5000):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return round((t1 - t0) / n * 1e6, 2)
cb = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
for name, fn in (("torch.matmul", lambda: torch.matmul(a, bt, out=c)), ("gemm_bf16 default (ring kernel)", lambda: ops.gemm_bf16(a, b, out=c)),
                 ("gemm_bf16 variant=14", lambda: ops.gemm_bf16(a, b, out=c, variant=14)), ("gemm_fp8 default", lambda: ops.gemm_fp8(a8, b8, out=c)),
                 ("gemm_bf16 default (stream, 4096x4096x512)", lambda: ops.gemm_bf16(big, big, out=cb))):
    out[name] = loop(fn)
    print(name, out[name], "us per call", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    ops.gemm_bf16(big, big, out=cb)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/call_overhead.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
