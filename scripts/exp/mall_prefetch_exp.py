"""Does a projection run faster when its weights were read just before (Infinity Cache / L2 hit)? Times the skinny kernel on
wo / qkv-sized weights (a) after 1 GB of unrelated reads, (b) after the same flush followed by one read of the weights."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.ops import llm  # noqa: E402

dev = "cuda"
junk = torch.empty(1 << 29, dtype=torch.bfloat16, device=dev).normal_()
x = torch.randn(1, 4096, device=dev, dtype=torch.bfloat16)


def timed(w, touch, reps=20):
    ts = []
    for _ in range(reps):
        junk.sum()                                   # 1 GiB through every cache level
        if touch:
            w.view(torch.int16).sum()                # one read of the weights
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        llm.skinny_linear(x, w)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 2)


for name, n in (("wo 4096x4096 (33.5 MB)", 4096), ("qkv 6144x4096 (50 MB)", 6144), ("16384x4096 (134 MB)", 16384), ("28672x4096 (235 MB)", 28672)):
    w = torch.randn(n, 4096, device=dev, dtype=torch.bfloat16) * 0.02
    llm.skinny_linear(x, w)
    print(json.dumps({"weights": name, "cold_us": timed(w, False), "after_one_read_us": timed(w, True)}), flush=True)
