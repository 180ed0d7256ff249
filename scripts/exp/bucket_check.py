"""Decode step of Llama-3-8B at batch sizes between the old buckets (now 160 and 224 have graphs of their own)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

eng = Engine(EngineConfig(model="llama3-8b", max_num_seqs=256, max_model_len=4096, kv_blocks=20000))
t0 = time.time()
free0 = torch.cuda.mem_get_info()[0]
eng.warmup()
print(json.dumps({"buckets": eng.buckets, "warmup_s": round(time.time() - t0, 2), "graph_pool_GB": round((free0 - torch.cuda.mem_get_info()[0]) / 1e9, 2)}))
print(json.dumps({b: round(sb.decode_step_ms(eng, b, 1024, iters=15)["ms"], 3) for b in (129, 150, 160, 170, 192, 210, 224, 256)}))
