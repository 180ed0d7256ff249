"""A/B in one process: the fused qkv split-sum + RoPE kernel against mid_linear + rope_cache (decode step, Llama-3-8B)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.ops import llm  # noqa: E402
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

fused = llm.mid_qkv_rope


def unfused(x, w, cos_sin, pos, slots, kc, vc, n_q, n_kv, weight_scale=None):
    return llm.rope_cache(llm.mid_linear(x, w, weight_scale=weight_scale), cos_sin, pos, slots, kc, vc, n_q, n_kv)


for weights in ("bf16", "fp8"):
    eng = Engine(EngineConfig(model="llama3-8b", max_num_seqs=128, max_model_len=4096, kv_blocks=20000, decode_weights=weights))
    for rep in range(2):
        for name, fn in (("fused", fused), ("unfused", unfused)):
            llm.mid_qkv_rope = fn
            eng._graphs.clear()
            row = {"weights": weights, "variant": name}
            for b in (32, 64, 128):
                row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=20)["ms"], 3)
            print(json.dumps(row), flush=True)
    del eng
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
