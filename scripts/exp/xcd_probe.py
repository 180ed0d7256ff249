"""Per-XCD busy share (amdsmi xcp_stats accumulators through the collector) under loads of known duty and shape."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from k8s_llm_monitor_amd.collectors.gpu import open_gpu_source  # noqa: E402
from k8s_llm_monitor_amd.loadgen.burner import Burner  # noqa: E402

src = open_gpu_source("amdsmi")
src.sample()
for kind, duty, size in (("gemm_bf16", 50.0, 8192), ("gemm_bf16", 100.0, 8192), ("gemm_bf16", 100.0, 1024), ("hbm_read", 100.0, 8192)):
    b = Burner(kind, size=size, duty=duty, device=0)
    th = threading.Thread(target=b.run, args=(5.0,))
    th.start()
    time.sleep(1.5)
    src.sample()
    time.sleep(3.0)
    g = src.sample()[0]
    th.join()
    print(json.dumps({"kind": kind, "duty": duty, "size": size, "gfxActivityAvg": round(g["gfxActivityAvg"], 2),
                      "xcdActivityAvg": [None if v is None else round(v, 2) for v in (g["xcdActivityAvg"] or [])],
                      "xcdActivity": g["xcdActivity"], "pcie": g["pcie"]}), flush=True)
    del b
    time.sleep(1.0)
