"""rocprofv3 workload: chosen GEMM arms only (KLM_ARMS=bf16_v3,...), 8192^3, uniform [-1, 1) operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("KLM_PROF_N", "8192")); it = int(os.environ.get("KLM_PROF_ITERS", "6"))
a = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(n, n, device=dev) * 2 - 1).to(torch.bfloat16)
a8 = a.to(torch.float8_e4m3fn); b8 = b.to(torch.float8_e4m3fn)
a4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8); b4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16); one = torch.ones((), device=dev); bt, b8t = b.t(), b8.t()
arms = {"bf16_v1": lambda: ops.gemm_bf16(a, b, out=c, variant=1), "bf16_v2": lambda: ops.gemm_bf16(a, b, out=c, variant=2),
        "bf16_v3": lambda: ops.gemm_bf16(a, b, out=c, variant=3), "bf16_v4": lambda: ops.gemm_bf16(a, b, out=c, variant=4),
        "bf16_v5": lambda: ops.gemm_bf16(a, b, out=c, variant=5), "bf16_v43": lambda: ops.gemm_bf16(a, b, out=c, variant=43), "bf16_hipblaslt": lambda: torch.matmul(a, bt, out=c),
        "fp8_v0": lambda: ops.gemm_fp8(a8, b8, out=c, variant=0), "fp8_v1": lambda: ops.gemm_fp8(a8, b8, out=c, variant=1),
        "fp8_v2": lambda: ops.gemm_fp8(a8, b8, out=c, variant=2), "fp8_v4": lambda: ops.gemm_fp8(a8, b8, out=c, variant=4),
        "bf16_v8": lambda: ops.gemm_bf16(a, b, out=c, variant=8), "fp8_v6": lambda: ops.gemm_fp8(a8, b8, out=c, variant=6),
        "fp4_v3": lambda: ops.gemm_fp4(a4, b4, out=c, variant=3), "fp4_v5": lambda: ops.gemm_fp4(a4, b4, out=c, variant=5),
        "fp4_v1": lambda: ops.gemm_fp4(a4, b4, out=c, variant=1),
        "fp8_scaled_mm": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)}
for k in os.environ.get("KLM_ARMS", "bf16_v2,bf16_v3,bf16_hipblaslt").split(","):
    for _ in range(it):
        arms[k]()
    torch.cuda.synchronize()
print("done")
