"""A/B in one process: decode at batch 129 - 256 on the K-split tiled GEMM + mid-batch epilogues against hipBLASLt."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

eng = Engine(EngineConfig(model="llama3-8b", max_num_seqs=256, max_model_len=4096, kv_blocks=20000))
eng.buckets = [136, 160, 192, 224, 240, 256]            # time row counts that only prompt chunks / mixed batches see, too
for rep in range(2):
    for wide in (True, False):
        eng.model.mid_limit = 256 if wide else 0          # every row count on either path, whatever the policy says
        eng._graphs.clear()
        row = {"k_split_gemm": wide}
        for b in eng.buckets:
            row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=15)["ms"], 3)
        print(json.dumps(row), flush=True)
