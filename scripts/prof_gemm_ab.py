"""Fixed workload for rocprofv3 PMC passes: the ring kernels (schedule 1) and the hand-scheduled 4-wave streams next to the vendor kernels, 8192^3,
uniform [-1, 1) operands. KLM_PROF_ITERS launches of each, back to back per kernel (so DVFS settles per kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("KLM_PROF_N", "8192"))
it = int(os.environ.get("KLM_PROF_ITERS", "6"))
a = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16)
a8 = a.to(torch.float8_e4m3fn); b8 = b.to(torch.float8_e4m3fn)
a4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8); b4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
one = torch.ones((), device=dev)
bt, b8t = b.t(), b8.t()
for fn in (lambda: ops.gemm_bf16(a, b, out=c, variant=2), lambda: ops.gemm_bf16(a, b, out=c, variant=4), lambda: torch.matmul(a, bt, out=c),
           lambda: ops.gemm_fp8(a8, b8, out=c, variant=1), lambda: ops.gemm_fp8(a8, b8, out=c, variant=2),
           lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16),
           lambda: ops.gemm_fp4(a4, b4, out=c, variant=1), lambda: ops.gemm_fp4(a4, b4, out=c, variant=2)):
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
print("done")
