"""Decode step of Llama-3-8B (or another preset) with fp8 decode weights at batch 1 - 128 against bf16 weights."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama3-8b"
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 16, 24, 32, 48, 64, 96, 128]
rows = []
for weights in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("bf16", "fp8")):
    eng = Engine(EngineConfig(model=name, max_num_seqs=128, max_model_len=4096, kv_blocks=20000, decode_weights=weights))
    for b in batches:
        r = sb.decode_step_ms(eng, b, 1024, iters=10)
        rows.append({"weights": weights, "batch": b, "ms": round(r["ms"], 3), "tokens_per_s": round(r["tokens_per_s"], 1),
                     "TBps": round(r["TBps"], 2)})
        print(json.dumps(rows[-1]), flush=True)
    del eng
    import gc
    gc.collect()
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"model": name, "ctx": 1024, "decode": rows}, open(f"gpurun_out/mid_fp8_e2e_{name}.json", "w"), indent=1)
