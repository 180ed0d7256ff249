"""Llama-3-8B at 224 - 256 rows: which projections should the K-split tiled GEMM take, which stay on hipBLASLt?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

eng = Engine(EngineConfig(model=sys.argv[1] if len(sys.argv) > 1 else "llama3-8b", max_num_seqs=256, max_model_len=4096, kv_blocks=int(os.environ.get("KLM_KV_BLOCKS", "20000"))))
eng.buckets = [192, 224, 256]
cases = [("policy", None, None), ("all tiled", 256, ()), ("wo on library", 256, ("wo",)), ("gate_up on library", 256, ("w_gate_up",)),
         ("wo + gate_up on library", 256, ("wo", "w_gate_up")), ("wo + gate_up + w_down on library", 256, ("wo", "w_gate_up", "w_down")),
         ("all library", 0, ())]
for name, limit, lib in cases:
    eng.model.mid_limit, eng.model.lib_proj = limit, (None if lib is None else frozenset(lib))
    eng._graphs.clear()
    row = {"case": name}
    for b in eng.buckets:
        row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=int(os.environ.get("KLM_ITERS", "15")))["ms"], 3)
    print(json.dumps(row), flush=True)
