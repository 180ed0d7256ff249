"""A short, fixed workload for rocprofv3 passes: each hot kernel a few times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
n = 8192
a = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16)
a8 = a.to(torch.float8_e4m3fn); b8 = b.to(torch.float8_e4m3fn)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
x = torch.empty(2 << 30, dtype=torch.uint8, device=dev); y = torch.empty_like(x); sink = torch.zeros(4, dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("KLM_PROF_ITERS", "5"))):
    ops.gemm_bf16(a, b, out=c, variant=1)
    ops.gemm_bf16(a, b, out=c, variant=0)
    ops.gemm_fp8(a8, b8, out=c)
    ops.hbm_copy(y, x); ops.hbm_read(x, sink); ops.hbm_fill(y, 1)
torch.cuda.synchronize()
print("done")
