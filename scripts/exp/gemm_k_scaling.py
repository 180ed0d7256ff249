"""Where does the time of the 8192 x 8192 x K GEMM go: per-K-tile (main loop) or per-output-tile (prologue, epilogue,
lock-step bursts)? Times K = 1024 .. 16384 for the ring kernels and the vendor kernels, interleaved rounds, and fits
t = a + b K per kernel: b -> main-loop rate, a -> cost per round of output tiles."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
M = N = 8192
Ks = [1024, 2048, 4096, 8192, 16384]
res = {}
g = torch.Generator(device=dev); g.manual_seed(1)
for K in Ks:
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    one = torch.ones((), device=dev)
    bt, b8t = b.t(), b8.t()
    cands = {"bf16_klm": lambda: ops.gemm_bf16(a, b, out=c, variant=2), "bf16_quad": lambda: ops.gemm_bf16(a, b, out=c, variant=3), "bf16_asm": lambda: ops.gemm_bf16(a, b, out=c, variant=4),
             "bf16_vendor": lambda: torch.matmul(a, bt, out=c),
             "fp8_klm": lambda: ops.gemm_fp8(a8, b8, out=c, variant=1), "fp8_asm": lambda: ops.gemm_fp8(a8, b8, out=c, variant=2), "fp8_vendor": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)}
    iters = max(10, int(40 * 8192 / K))
    runs = {k: [] for k in cands}
    for r in range(6):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(e0.elapsed_time(e1) * 1e3 / iters)      # us
    res[K] = {k: sorted(v)[len(v) // 2] for k, v in runs.items()}
    print(K, {k: round(v, 1) for k, v in res[K].items()}, flush=True)
fit = {}
for name in res[Ks[0]]:
    xs, ys = Ks, [res[K][name] for K in Ks]
    n = len(xs); mx, my = sum(xs) / n, sum(ys) / n
    b_ = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    a_ = my - b_ * mx
    fit[name] = {"intercept_us": round(a_, 2), "us_per_1024_k": round(b_ * 1024, 3), "mainloop_TFLOPs": round(2.0 * M * N * 1024 / (b_ * 1024 * 1e-6) / 1e12, 1),
                 "tflops_at_8192": round(2.0 * M * N * 8192 / res[8192][name] / 1e6, 1)}
    print(name, fit[name], flush=True)
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_k_scaling.json"
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/out.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump({"M": M, "N": N, "us": res, "fit": fit}, open(out, "w"), indent=1)
