"""Llama-3-8B (random weights) on one MI355X: decode step sweep, prefill, and an end-to-end
open-loop serving run. Writes gpurun_out/llm_bench.json."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402
from k8s_llm_monitor_amd.serving.loadclient import OpenLoopLoad  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama3-8b"
kv_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
decode_weights = sys.argv[3] if len(sys.argv) > 3 else "bf16"
kv_dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
out = {"model": model}
t0 = time.time()
eng = Engine(EngineConfig(model=model, max_num_seqs=256, max_model_len=4096, kv_blocks=kv_blocks, decode_weights=decode_weights, kv_cache_dtype=kv_dtype))
torch.cuda.synchronize()
out["init_s"] = time.time() - t0
out["engine"] = eng.describe()
print("init", out["init_s"], flush=True)
t0 = time.time()
eng.warmup()
out["graph_capture_s"] = time.time() - t0
print("graphs", out["graph_capture_s"], flush=True)
rows = []
for ctx in (128, 1024, 2048):
    for b in (1, 8, 32, 64, 128, 256):
        if b * ctx > eng.kv_capacity_tokens:
            continue
        r = sb.decode_step_ms(eng, b, ctx)
        rows.append(r)
        print(json.dumps(r), flush=True)
out["decode"] = rows
# the same step without the hipGraph: what capture buys
eng.cfg.use_graphs = False
out["decode_eager"] = [sb.decode_step_ms(eng, b, 1024, iters=5) for b in (1, 32)]
eng.cfg.use_graphs = True
print(json.dumps(out["decode_eager"]), flush=True)
pre = []
for n, L in ((1, 128), (1, 1024), (1, 4096), (8, 1024)):
    r = sb.prefill_ms(eng, n, L)
    pre.append(r)
    print(json.dumps(r), flush=True)
out["prefill"] = pre
# end to end: open-loop load for 20 s
eng.start()
load = OpenLoopLoad(eng, rate_rps=20, prompt_len=512, output_len=128, seed=0).start(20.0)
load.join(40)
ok = load.wait_all(120)
eng.stop()
out["serving"] = {"drained": ok, "requests": len(load.requests), "truth": eng.metrics.truth(),
                  "engine": eng.describe()}
print(json.dumps(out["serving"]["truth"]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/llm_bench%s.json" % (("" if model == "llama3-8b" else "_" + model) + ("" if decode_weights == "bf16" else "_" + decode_weights) + ("" if kv_dtype == "bf16" else "_kv" + kv_dtype)), "w"), indent=1, default=str)
