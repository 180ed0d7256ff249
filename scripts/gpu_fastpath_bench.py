"""Llama-3-8B decode with both opt-in fp8 switches (decode weights and KV cache) against the bf16 defaults."""
import gc
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama3-8b"
rows = []
for weights, kv in (("bf16", "bf16"), ("fp8", "fp8")):
    eng = Engine(EngineConfig(model=name, max_num_seqs=256, max_model_len=4096, kv_blocks=20000, decode_weights=weights,
                              kv_cache_dtype=kv))
    for b in (1, 16, 32, 64, 128, 256):
        r = sb.decode_step_ms(eng, b, 1024, iters=10)
        rows.append({"weights": weights, "kv": kv, "batch": b, "ms": round(r["ms"], 3), "tokens_per_s": round(r["tokens_per_s"], 1)})
        print(json.dumps(rows[-1]), flush=True)
    del eng
    gc.collect()
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"model": name, "ctx": 1024, "decode": rows}, open(f"gpurun_out/fastpath_{name}.json", "w"), indent=1)
