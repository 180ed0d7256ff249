"""One-shot HBM kernels (no loop; 1 / 2 / 4 accesses of 16 bytes per thread) against the shipped forms and torch, 4 GiB buffers."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("KLM_FILL_GIB", "4")) << 30
x = torch.empty(n, dtype=torch.uint8, device=dev); y = torch.empty(n, dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
ops.hbm_fill(x, 0x01020304)
def bench(fn, nbytes, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return nbytes / (e0.elapsed_time(e1) / iters) / 1e9
arms = {"torch_fill": (lambda: y.fill_(7), n), "klm_fill": (lambda: ops.hbm_fill(y, 0x5A5A5A5A), n),
        "torch_copy": (lambda: y.copy_(x), 2 * n), "klm_copy": (lambda: ops.hbm_copy(y, x), 2 * n), "klm_read": (lambda: ops.hbm_read(x, sink), n)}
for u, bits in ((1, 16), (2, 24), (4, 8)):
    for nt in (0, 2):
        arms[f"fill_u{u}_{'nt' if nt else 'plain'}"] = (lambda f=bits | nt: ops.hbm_fill_ex(y, 0x5A5A5A5A, f, 0), n)
    for nt in (0, 1, 2, 3):
        arms[f"copy_u{u}_ld{'nt' if nt & 1 else 'pl'}_st{'nt' if nt & 2 else 'pl'}"] = (lambda f=bits | nt: ops.hbm_copy_ex(y, x, f, 0), 2 * n)
    for nt in (0, 1):
        arms[f"read_u{u}_{'nt' if nt else 'plain'}"] = (lambda f=bits | nt: ops.hbm_read_ex(x, sink, f, 0), n)
res = {}
for r in range(2):
    for k, (fn, nb) in arms.items():
        res[k] = bench(fn, nb)
for k, v in res.items():
    print(f"{k}: {v:.2f} TB/s", flush=True)
ops.hbm_copy_ex(y, x, 16 | 3, 0); torch.cuda.synchronize()
print("copy_u1 equal:", bool(torch.equal(x[: 1 << 28], y[: 1 << 28]) and torch.equal(x[-(1 << 28):], y[-(1 << 28):])))
ops.hbm_fill_ex(y, 0x11223344, 16, 0); torch.cuda.synchronize()
print("fill_u1 ok:", bool((y.view(torch.int32)[:: 1 << 20] == 0x11223344).all()) and int(y.view(torch.int32)[-1]) == 0x11223344)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/hbm_one_shot.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(res, open(_o, "w"), indent=1)
