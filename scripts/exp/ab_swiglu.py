"""A/B in one process: SwiGLU inside the mid-batch main kernel (no K split) against partial sums + epilogue kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.ops import llm  # noqa: E402
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402

real = llm.mid_linear


def split2(x, w, kind="none", **kw):
    if kind == "swiglu" and x.shape[0] > 32:
        kw = dict(kw, ksplit=2, row_tiles=2)         # the previous plan for 33 - 64 rows: 2 x 128 blocks + epilogue
    elif kind == "swiglu":
        return real(x, w, kind="none", **kw).new_empty(0) if False else _epilogue(x, w, **kw)
    return real(x, w, kind=kind, **kw)


def _epilogue(x, w, **kw):
    from k8s_llm_monitor_amd.ops import ref
    return llm.silu_mul(real(x, w, **kw))           # closest stand-in for "ks = 1 + separate epilogue launch"


for weights in ("bf16", "fp8"):
    eng = Engine(EngineConfig(model="llama3-8b", max_num_seqs=128, max_model_len=4096, kv_blocks=20000, decode_weights=weights))
    for rep in range(2):
        for name, fn in (("fused", real), ("separate", split2)):
            llm.mid_linear = fn
            eng._graphs.clear()
            row = {"weights": weights, "variant": name}
            for b in (24, 32, 48, 64):
                row[b] = round(sb.decode_step_ms(eng, b, 1024, iters=20)["ms"], 3)
            print(json.dumps(row), flush=True)
    llm.mid_linear = real
    del eng
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()
