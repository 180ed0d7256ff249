"""Open-loop load on the Llama-3-8B engine with and without mixed batches (running sequences decoding inside prefill
launches): inter-token latency, TTFT and throughput from the engine's exact ground truth."""
import gc
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_llm_monitor_amd.models.llama import PRESETS, LlamaModel  # noqa: E402
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving.loadclient import OpenLoopLoad  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mixed_batch_ab.json"
model = LlamaModel(PRESETS["llama3-8b"], "cuda", seed=0)
rows = []
for rate, plen, olen in ((20, 512, 128), (40, 512, 128), (12, 2048, 128)):
    for mixed in (False, True):
        eng = Engine(EngineConfig(model="llama3-8b", max_num_seqs=128, max_model_len=4096, kv_blocks=16000,
                                  max_num_batched_tokens=2048, mixed_batches=mixed), model=model)
        eng.warmup()
        eng.start()
        try:
            load = OpenLoopLoad(eng, rate_rps=rate, prompt_len=plen, output_len=olen, seed=4).start(10.0)
            load.join(30)
            ok = load.wait_all(120)
            t = eng.metrics.truth()
        finally:
            eng.stop()
        ms = lambda v: None if v is None else round(v * 1e3, 2)   # noqa: E731
        row = {"rate_rps": rate, "prompt_len": plen, "output_len": olen, "mixed_batches": mixed, "all_finished": ok,
               "requests": len(load.requests), "itl_p50_ms": ms(t["itl_p50"]), "itl_p99_ms": ms(t["itl_p99"]),
               "ttft_p50_ms": ms(t["ttft_p50"]), "ttft_p99_ms": ms(t["ttft_p99"]), "e2e_p50_ms": ms(t["e2e_p50"]),
               "tokens_per_s": round(t["tokens_per_s"] or 0, 1), "launches": dict(eng.step_counts),
               "launch_seconds": {k: round(v, 3) for k, v in eng.step_seconds.items()}}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del eng
        gc.collect()
        torch.cuda.empty_cache()
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump({"model": "llama3-8b (random weights, bf16)", "rows": rows}, open(out_path, "w"), indent=1)
