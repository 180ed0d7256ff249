"""Fixed workload for rocprofv3 PMC passes over round 3's default kernels: `variant=None` of the three formats at 8192^3 (four
full rounds of the persistent stream), 6144^3 (two rounds + a partial one cut into K slices + the reduce kernel) and 2048^3 (the
eight-wave 128 x 128 ring kernel), next to the vendor kernels on the same operands. KLM_PROF_ITERS launches of each, back to back; KLM_PROF_N: the sizes
(one per run when the counters are to be read per shape: rocprofv3 sums a kernel's dispatches by name)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
it = int(os.environ.get("KLM_PROF_ITERS", "6"))
one = torch.ones((), device=dev)
for n in [int(x) for x in os.environ.get("KLM_PROF_N", "8192,6144,2048").split(",")]:
    a = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16)
    a8 = a.to(torch.float8_e4m3fn); b8 = b.to(torch.float8_e4m3fn)
    a4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8); b4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8)
    c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    bt, b8t = b.t(), b8.t()
    for fn in (lambda: ops.gemm_bf16(a, b, out=c), lambda: torch.matmul(a, bt, out=c),
               lambda: ops.gemm_fp8(a8, b8, out=c), lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16),
               lambda: ops.gemm_fp4(a4, b4, out=c)):
        for _ in range(it):
            fn()
        torch.cuda.synchronize()
    del a, b, a8, b8, a4, b4, c
print("done")
