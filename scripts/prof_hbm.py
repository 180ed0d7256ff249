"""rocprofv3 workload: the HBM kernels on 2 GiB buffers (copy = 4 GiB of traffic), KLM_PROF_ITERS launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
n = 2 << 30
it = int(os.environ.get("KLM_PROF_ITERS", "8"))
x = torch.empty(n, dtype=torch.uint8, device=dev); y = torch.empty(n, dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
ops.hbm_fill(x, 0x5A5A5A5A)
for fn in (lambda: ops.hbm_copy(y, x), lambda: ops.hbm_read(x, sink), lambda: ops.hbm_fill(y, 0x12345678), lambda: y.copy_(x)):
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
print("done")
