"""Per-output-tile cost (prologue + epilogue) of the GEMM kernels: 8192 x 8192 x 1024, where it is a fifth of the time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
M = N = 8192
for K in (1024, 8192):
    g = torch.Generator(device=dev); g.manual_seed(1)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    one = torch.ones((), device=dev); bt, b8t = b.t(), b8.t()
    ref = ops.gemm_bf16(a, b, variant=2)
    assert torch.equal(ops.gemm_bf16(a, b, variant=4), ref)
    cands = {"bf16_ring": lambda: ops.gemm_bf16(a, b, out=c, variant=2), "bf16_asm": lambda: ops.gemm_bf16(a, b, out=c, variant=4),
             "bf16_vendor": lambda: torch.matmul(a, bt, out=c), "fp8_asm": lambda: ops.gemm_fp8(a8, b8, out=c, variant=2),
             "fp8_vendor": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)}
    runs = {k: [] for k in cands}
    it = 200 if K == 1024 else 30
    for r in range(6):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(e0.elapsed_time(e1) * 1e3 / it)
    print(K, {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items()}, flush=True)
