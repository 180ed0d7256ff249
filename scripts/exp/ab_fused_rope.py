"""A/B in one process: RoPE + cache write inside the decode attention kernel against the separate rope_cache launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

eng = Engine(EngineConfig(model=sys.argv[1] if len(sys.argv) > 1 else "llama3-8b", max_num_seqs=16, max_model_len=4096, kv_blocks=4000))
for rep in range(2):
    for fused in (True, False):
        eng.model.fused_rope = fused
        eng._graphs.clear()
        row = {"fused_rope": fused}
        for b in (1, 4, 16):
            row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=30)["ms"], 4)
        print(json.dumps(row), flush=True)
