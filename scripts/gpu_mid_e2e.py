"""Decode step of Llama-3-8B at batch 24 - 128 with the mid-batch kernel on and off (same engine, same graphs path)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama3-8b"
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [24, 32, 48, 64, 96, 128]
eng = Engine(EngineConfig(model=name, max_num_seqs=128, max_model_len=4096, kv_blocks=20000))
eng.buckets = sorted(set(eng.buckets) | set(batches))
rows = []
for lim, tag in ((0, "library"), (None, "mid")):
    eng.model.mid_limit = lim
    eng._graphs.clear()
    for b in batches:
        r = sb.decode_step_ms(eng, b, 1024, iters=10)
        rows.append({"path": tag if eng.model.mid(b) or lim == 0 else "library", "batch": b, "ms": round(r["ms"], 3),
                     "tokens_per_s": round(r["tokens_per_s"], 1)})
        print(json.dumps(rows[-1]), flush=True)
pre = []
for lim, tag in ((0, "library"), (None, "mid")):
    eng.model.mid_limit = lim
    for n, L in ((1, 64), (1, 128), (2, 48)):
        r = sb.prefill_ms(eng, n, L, iters=5)
        pre.append({"path": tag, "n_seqs": n, "prompt_len": L, "ms": round(r["ms"], 3)})
        print(json.dumps(pre[-1]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"model": name, "ctx": 1024, "decode": rows, "prefill": pre}, open(f"gpurun_out/mid_e2e_{name}.json", "w"), indent=1)
