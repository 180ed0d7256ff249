"""The bench's training step (fwd, dgrad, wgrad on six different operand matrices) with C written back vs streamed."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
from k8s_llm_monitor_amd.models.linear_step import LinearLayerStep
dev = torch.device("cuda:0")
out = {}
arms = {"persistent_writeback": 8, "persistent_streamed": 9, "per_tile": 4}
steps = {k: LinearLayerStep(torch, None, ops, dev, 1, 8192, gemm=(lambda a, b, out, v=v: ops.gemm_bf16(a, b, out=out, variant=v))) for k, v in arms.items()}
steps["hipblaslt"] = LinearLayerStep(torch, None, ops, dev, 1, 8192, gemm=lambda a, b, out: torch.matmul(a, b.t(), out=out))
runs = {k: [] for k in steps}
for r in range(6):
    for name, st in steps.items():
        for _ in range(3):
            st.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            st.step()
        e1.record(); e1.synchronize()
        if r:
            runs[name].append(st.flops * 30 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
out = {k: {"median": sorted(v)[len(v) // 2], "min": min(v), "max": max(v)} for k, v in runs.items()}
for k, v in out.items():
    print(k, {a: round(b, 1) for a, b in v.items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_stream_c.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
