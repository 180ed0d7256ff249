"""rocprofv3 PMC workload: the fp4 default kernel at 8192 x 8192 x K for K in KLM_PROF_K (one K per run: the counters are summed
per kernel name). MFMA-busy share as a function of K: the K loop's efficiency is fixed, the per-tile epilogue is not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
K = int(os.environ.get("KLM_PROF_K", "8192"))
n = 8192
a4 = torch.randint(0, 256, (n, K // 2), device=dev, dtype=torch.uint8); b4 = torch.randint(0, 256, (n, K // 2), device=dev, dtype=torch.uint8)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
for _ in range(8):
    ops.gemm_fp4(a4, b4, out=c)
torch.cuda.synchronize()
print("done")
