"""torch's fill_ reads 6.9 TB/s where klm_hbm_fill reads 6.1: which kernel is it, with what launch shape? (run under
rocprofv3 --kernel-trace for the names and grids; alone for the rates)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("KLM_FILL_GIB", "4")) << 30
y = torch.empty(n, dtype=torch.uint8, device=dev)
def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return n / (e0.elapsed_time(e1) / iters) / 1e9
res = {}
y32, yf, yb = y.view(torch.int32), y.view(torch.float32), y.view(torch.bfloat16)
arms = {"torch_fill_u8": lambda: y.fill_(7), "torch_fill_i32": lambda: y32.fill_(7), "torch_fill_f32": lambda: yf.fill_(1.5), "torch_fill_bf16": lambda: yb.fill_(1.5),
        "torch_zero_u8": lambda: y.zero_(), "klm_fill": lambda: ops.hbm_fill(y, 0x5A5A5A5A),
        "klm_fill_f0_g32768": lambda: ops.hbm_fill_ex(y, 0x5A5A5A5A, 0, 32768), "klm_fill_f4_g8192": lambda: ops.hbm_fill_ex(y, 0x5A5A5A5A, 4, 8192),
        "klm_fill_tile": lambda: ops.hbm_fill_ex(y, 0x5A5A5A5A, 8, 0)}
for r in range(2):
    for k, fn in arms.items():
        res[k] = bench(fn)
for k, v in res.items():
    print(f"{k}: {v:.2f} TB/s", flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fill_vs_torch.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(res, open(_o, "w"), indent=1)
