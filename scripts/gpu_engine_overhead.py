"""How much of a served decode step is the engine's own Python (scheduling, host buffers, metrics)? Offline run:
N requests submitted at once, stepped to completion; wall time per decode step against the bare graph replay."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine(EngineConfig(model="llama3-8b", max_num_seqs=n_req, max_model_len=2048, kv_blocks=20000))
eng.warmup()
bare = sb.decode_step_ms(eng, n_req, 192, iters=20)["ms"]
g = torch.Generator().manual_seed(0)
reqs = [eng.submit(torch.randint(0, 128000, (128,), generator=g).tolist(), 128) for _ in range(n_req)]
t0 = time.perf_counter()
steps = 0
while not eng.idle():
    eng.step()
    steps += 1
wall = time.perf_counter() - t0
d = eng.step_counts["decode"]
out = {"requests": n_req, "wall_s": round(wall, 3), "tokens_per_s": round(n_req * 128 / wall, 1), "steps": steps,
       "decode_steps": d, "decode_ms_per_step_served": round(eng.step_seconds["decode"] / d * 1e3, 3),
       "decode_ms_bare_graph_replay_ctx192": round(bare, 3), "prefill_s": round(eng.step_seconds["prefill"], 3)}
print(json.dumps(out))
