"""Decode / prefill steps of the Llama engine for rocprofv3 (kernel trace):
    rocprofv3 --kernel-trace --stats -d out -- python3 scripts/prof_llm.py decode 32 1024 20"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

mode, batch, ctx, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
model = sys.argv[5] if len(sys.argv) > 5 else "llama3-8b"
eng = Engine(EngineConfig(model=model, max_num_seqs=256, max_model_len=4096, kv_blocks=40000, use_graphs=False))
if mode == "decode":
    print(sb.decode_step_ms(eng, batch, ctx, iters=iters, warmup=2))
else:
    print(sb.prefill_ms(eng, batch, ctx, iters=iters, warmup=1))
torch.cuda.synchronize()
