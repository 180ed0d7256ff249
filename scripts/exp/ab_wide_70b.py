"""The K-split tiled GEMM path on the Llama-3-70B shapes at 129 - 256 rows against hipBLASLt (policy input)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

eng = Engine(EngineConfig(model="llama3-70b", max_num_seqs=256, max_model_len=4096, kv_blocks=int(os.environ.get("KLM_KV_BLOCKS", "18000"))))
eng.buckets = [40, 64, 80, 96, 128, 136, 256]
for rep in range(1):
    for name, limit, wide_from in (("tiled gemm", 256, 1), ("mid kernel (<= 128)", 256, None), ("library", 0, None)):
        eng.model.mid_limit, eng.model.wide_from = limit, wide_from
        eng._graphs.clear()
        row = {"path": name}
        for b in eng.buckets:
            row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=6)["ms"], 3)
        print(json.dumps(row), flush=True)
