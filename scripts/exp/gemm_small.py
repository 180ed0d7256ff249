"""Shapes of fewer 256 x 256 tiles than CUs: the pipelined 128 x 128 kernel (variant 14) next to the old 128 x 128 kernel (0),
the persistent stream (10) and hipBLASLt. Exact small-integer check first; then interleaved rounds, medians, TFLOP/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(9)
out = {"exact": {}, "timing": {}}
for M, N, K in ((128, 128, 64), (128, 256, 128), (256, 384, 192), (640, 384, 320), (1024, 1024, 1024), (2048, 1920, 4160)):
    a = torch.randint(-3, 4, (M, K), device=dev, generator=g).to(torch.bfloat16)
    b = torch.randint(-3, 4, (N, K), device=dev, generator=g).to(torch.bfloat16)
    b[:, 0] = (torch.arange(N, device=dev) % 5).to(torch.bfloat16)
    want = (a.float() @ b.float().t()).to(torch.bfloat16)
    c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.gemm_bf16(a, b, out=c, variant=14)
    torch.cuda.synchronize()
    out["exact"][f"{M}x{N}x{K}"] = bool(torch.equal(c, want))
    print(M, N, K, out["exact"][f"{M}x{N}x{K}"], flush=True)
ok = all(out["exact"].values())
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/gemm_small.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
if not ok:
    print("MISMATCH")
    sys.exit(3)
for M, N, K in ((512, 512, 512), (1024, 1024, 1024), (1536, 1536, 1536), (2048, 2048, 2048), (2560, 2560, 2560), (3072, 3072, 3072), (3584, 3584, 3584), (4096, 4096, 4096),
                (2048, 2048, 8192), (1024, 4096, 4096), (4096, 1024, 1024), (2048, 6144, 4096), (5120, 5120, 5120)):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bt = b.t()
    cands = {"v0_128": lambda: ops.gemm_bf16(a, b, out=c, variant=0), "v14_128p": lambda: ops.gemm_bf16(a, b, out=c, variant=14),
             "default": lambda: ops.gemm_bf16(a, b, out=c), "hipblaslt": lambda: torch.matmul(a, bt, out=c)}
    if M % 256 == 0 and N % 256 == 0 and K % 128 == 0 and K >= 256:
        cands["v10_stream"] = lambda: ops.gemm_bf16(a, b, out=c, variant=10)
        cands["v11_split"] = lambda: ops.gemm_bf16(a, b, out=c, variant=11)
    flop = 2.0 * M * N * K
    iters = max(10, min(200, int(4e12 / flop)))
    runs = {k: [] for k in cands}
    for r in range(4):
        for k, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[k].append(flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items()}
    med["tiles_256"] = (M // 256) * (N // 256)
    med["picked"] = ops.pick_bf16_variant(M, N, K)
    med["model_us"] = {"p128": round(ops.p128_time_us(M, N, 2 * K), 1), "stream": round(ops.stream_time_us(M, N, 2 * K), 1) if "v10_stream" in cands else None}
    med["measured_us"] = {k: round(flop / v / 1e6, 1) for k, v in med.items() if k in runs}
    med["default_over_hipblaslt"] = round(med["default"] / med["hipblaslt"], 3)
    med["best_klm_over_hipblaslt"] = round(max(v for k, v in med.items() if k.startswith("v")) / med["hipblaslt"], 3)
    out["timing"][f"{M}x{N}x{K}"] = med
    print(f"{M}x{N}x{K}", med, flush=True)
    json.dump(out, open(_o, "w"), indent=1)
