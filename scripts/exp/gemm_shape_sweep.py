"""Square and rectangular shapes: every bf16 kernel next to hipBLASLt (which kernel should `variant=None` pick where?), and
the fp8 default next to torch._scaled_mm. Interleaved rounds, medians, TFLOP/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
shapes = [(1024, 1024, 1024), (2048, 2048, 2048), (3072, 3072, 3072), (4096, 4096, 4096), (6144, 6144, 6144), (8192, 8192, 8192),
          (12288, 12288, 12288), (16384, 16384, 8192), (8192, 8192, 512), (2048, 8192, 8192), (8192, 28672, 4096), (1024, 16384, 16384),
          (4096, 4096, 16384), (5120, 5120, 5120)]
one = torch.ones((), device=dev)
out = {}
for M, N, K in shapes:
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    bt, b8t = b.t(), b8.t()
    cands = {"v0_128": lambda: ops.gemm_bf16(a, b, out=c, variant=0), "v2_ring": lambda: ops.gemm_bf16(a, b, out=c, variant=2),
             "v4_stream": lambda: ops.gemm_bf16(a, b, out=c, variant=4), "v8_persistent": lambda: ops.gemm_bf16(a, b, out=c, variant=8),
             "default": lambda: ops.gemm_bf16(a, b, out=c),
             "hipblaslt": lambda: torch.matmul(a, bt, out=c),
             "fp8_default": lambda: ops.gemm_fp8(a8, b8, out=c), "fp8_ring": lambda: ops.gemm_fp8(a8, b8, out=c, variant=1),
             "fp8_scaled_mm": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)}
    flop = 2.0 * M * N * K
    iters = max(5, min(50, int(2e13 / flop)))
    runs = {k: [] for k in cands}
    for r in range(4):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    out[f"{M}x{N}x{K}"] = {k: sorted(v)[len(v) // 2] for k, v in runs.items()}
    print(f"{M}x{N}x{K}", {k: round(v) for k, v in out[f"{M}x{N}x{K}"].items()}, flush=True)
    del a, b, c, a8, b8
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_shape_sweep.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
