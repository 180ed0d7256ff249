"""One rank of llama3-405b / tp8 (peers absent, as scripts/gpu_tp_shard_bench.py): decode step at 48 - 256 rows on the
K-split tiled GEMM, on the mid-batch kernel and on hipBLASLt (policy input)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_tp_shard_bench import AbsentPeers  # noqa: E402
from k8s_llm_monitor_amd.models.llama import PRESETS, LlamaModel  # noqa: E402
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = LlamaModel(PRESETS["llama3-405b"], dev, tp=AbsentPeers(8, dev))
eng = Engine(EngineConfig(model="llama3-405b", device="cuda:0", max_num_seqs=256, max_model_len=4096, kv_blocks=18000), model=model)
MIX = len(sys.argv) > 1 and sys.argv[1] == "mix"
if MIX:          # which projections of the shard should the tiled GEMM take?
    eng.buckets = [136, 192, 256]
    cases = [("all tiled", 256, 1, ()), ("gate_up on library", 256, 1, ("w_gate_up",)), ("wo on library", 256, 1, ("wo",)),
             ("qkv-side only tiled (wo + gate_up + w_down on library)", 256, 1, ("wo", "w_gate_up", "w_down")),
             ("w_down on library", 256, 1, ("w_down",)), ("all library", 0, None, ())]
else:
    eng.buckets = [48, 64, 96, 128, 136, 192, 256]
    cases = [("tiled gemm", 256, 1, ()), ("mid kernel (<= 128)", 256, None, ()), ("library", 0, None, ())]
for name, limit, wide_from, lib in cases:
    model.mid_limit, model.wide_from, model.lib_proj = limit, wide_from, frozenset(lib)
    eng._graphs.clear()
    row = {"path": name}
    for b in eng.buckets:
        row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=6)["ms"], 3)
    print(json.dumps(row), flush=True)
