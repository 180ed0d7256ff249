"""(Historic: needs `klm_gemm_persistent_map` and the `map` argument of gemm_quad_asm_p, which were removed again after this
measurement -- no difference, profiles/r2_prof/tile_map_ab.json.) Tile -> (workgroup, round) map of the persistent kernels: the shipped one (every XCD walks its own contiguous chunk in 8 x 4
patches) against one 16 x 16 patch per round for the whole chip (XCDs share panels through the Infinity Cache), on a repeated
8192^3 GEMM and on the training step (six operand matrices)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
from k8s_llm_monitor_amd.models.linear_step import LinearLayerStep
dev = torch.device("cuda:0")
lib = ops.load()
n = 8192
g = torch.Generator(device=dev); g.manual_seed(4)
a = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
bt = b.t()
ref, ref8 = ops.gemm_bf16(a, b, variant=4), ops.gemm_fp8(a8, b8, variant=2)
lib.klm_gemm_persistent_map(1)
eq = {"bf16": bool(torch.equal(ops.gemm_bf16(a, b, variant=8), ref)), "fp8": bool(torch.equal(ops.gemm_fp8(a8, b8, variant=4), ref8))}
lib.klm_gemm_persistent_map(0)
print("map 1 equal:", eq, flush=True)
assert all(eq.values())
step = LinearLayerStep(torch, None, ops, dev, 1, n)
step_v = LinearLayerStep(torch, None, ops, dev, 1, n, gemm=lambda x, w, out: torch.matmul(x, w.t(), out=out))
def timed(fn, flop, iters):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12
arms = {"bf16_map0": (0, lambda: ops.gemm_bf16(a, b, out=c, variant=8), 2.0 * n ** 3, 20), "bf16_map1": (1, lambda: ops.gemm_bf16(a, b, out=c, variant=8), 2.0 * n ** 3, 20),
        "bf16_hipblaslt": (0, lambda: torch.matmul(a, bt, out=c), 2.0 * n ** 3, 20),
        "fp8_map0": (0, lambda: ops.gemm_fp8(a8, b8, out=c, variant=4), 2.0 * n ** 3, 20), "fp8_map1": (1, lambda: ops.gemm_fp8(a8, b8, out=c, variant=4), 2.0 * n ** 3, 20),
        "step_map0": (0, step.step, step.flops, 10), "step_map1": (1, step.step, step.flops, 10), "step_hipblaslt": (0, step_v.step, step.flops, 10)}
runs = {k: [] for k in arms}
for r in range(6):
    for name, (m, fn, flop, it) in arms.items():
        lib.klm_gemm_persistent_map(m)
        v = timed(fn, flop, it)
        if r:
            runs[name].append(v)
lib.klm_gemm_persistent_map(0)
out = {"equal": eq, "tflops": {k: sorted(v)[len(v) // 2] for k, v in runs.items()}}
for k, v in out["tflops"].items():
    print(k, round(v, 1), flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tile_map_ab.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
