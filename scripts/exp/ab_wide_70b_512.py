"""Does the K-split tiled GEMM path keep its lead on the Llama-3-70B shapes past 256 rows (two and more row tiles)?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.ops import llm  # noqa: E402
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

llm.WIDE_MAX_ROWS = 1024
eng = Engine(EngineConfig(model=sys.argv[1] if len(sys.argv) > 1 else "llama3-70b", max_num_seqs=1024, max_model_len=1024, kv_blocks=18000))
eng.buckets = [256, 320, 384, 512, 768, 1024]
for name, limit in (("tiled gemm", 1024), ("library", 0)):
    eng.model.mid_limit = limit
    eng._graphs.clear()
    row = {"path": name}
    for b in eng.buckets:
        row[b] = round(sb.decode_step_ms(eng, b, 192, iters=5)["ms"], 3)
    print(json.dumps(row), flush=True)
