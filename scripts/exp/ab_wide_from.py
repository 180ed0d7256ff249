"""Should the tiled GEMM (qkv / wo / w_down; gate_up on the library) replace the mid-batch kernel below 129 rows on 8B shapes?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

eng = Engine(EngineConfig(model=sys.argv[1] if len(sys.argv) > 1 else "llama3-8b", max_num_seqs=256, max_model_len=4096,
                          kv_blocks=int(os.environ.get("KLM_KV_BLOCKS", "20000"))))
eng.buckets = [64, 80, 96, 112, 128]
for rep in range(2):
    for wide_from in (None, 65):
        eng.model.wide_from = wide_from
        eng._graphs.clear()
        row = {"tiled_from_rows": wide_from}
        for b in eng.buckets:
            row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=int(os.environ.get("KLM_ITERS", "15")))["ms"], 3)
        print(json.dumps(row), flush=True)
