"""The 256 x 128 / 128 x 256 forms of the eight-wave ring kernel (variants 16 / 17) on shapes between the 128 x 128 kernel's and the
256 x 256 stream's ranges: exactness first (all three formats, odd and even K tile counts), then microseconds next to variants
14, 10 and the vendor kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(21)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/gemm_mid.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
out = {"exact": {}, "us": {}}
one = torch.ones((), device=dev)
for M, N, K in ((256, 256, 128), (256, 256, 256), (512, 768, 384), (768, 512, 640), (1024, 1280, 1024), (2048, 1792, 4224)):
    a = torch.randint(-3, 4, (M, K), device=dev, generator=g).float()
    b = torch.randint(-3, 4, (N, K), device=dev, generator=g).float()
    b[:, 0] = (torch.arange(N, device=dev) % 5).float()
    a[:, 1] = (torch.arange(M, device=dev) % 7).float() - 3
    want = (a @ b.t()).to(torch.bfloat16)
    a4 = torch.randint(0, 256, (M, K), device=dev, generator=g, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (N, K), device=dev, generator=g, dtype=torch.uint8)
    want4 = (ops.fp4_decode(a4).double() @ ops.fp4_decode(b4).double().t()).float().to(torch.bfloat16)
    row = {}
    for v in (16, 17):
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        row[f"bf16_v{v}"] = bool(torch.equal(ops.gemm_bf16(a.to(torch.bfloat16), b.to(torch.bfloat16), out=c, variant=v), want))
        c.fill_(float("nan"))
        row[f"fp8_v{v}"] = bool(torch.equal(ops.gemm_fp8(a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn), out=c, variant=v), want))
        c.fill_(float("nan"))
        row[f"fp4_v{v}"] = bool(torch.equal(ops.gemm_fp4(a4, b4, out=c, variant=v), want4))
    torch.cuda.synchronize()
    out["exact"][f"{M}x{N}x{K}"] = row
    print(M, N, K, row, flush=True)
json.dump(out, open(_o, "w"), indent=1)
if not all(v for r in out["exact"].values() for v in r.values()):
    print("MISMATCH"); sys.exit(3)

for M, N, K in ((2560, 2560, 2560), (3072, 3072, 3072), (2048, 2048, 2048), (2304, 2304, 2304), (2816, 2816, 2816), (3328, 3328, 3328), (4096, 4096, 4096),
                (1024, 6144, 4096), (1536, 4096, 4096), (2048, 4096, 4096), (512, 28672, 4096), (1024, 4096, 14336), (2048, 6144, 4096), (8192, 8192, 8192)):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bt, b8t = b.t(), b8.t()
    cands = {"hipblaslt": lambda: torch.matmul(a, bt, out=c), "bf16_default": lambda: ops.gemm_bf16(a, b, out=c), "bf16_v14": lambda: ops.gemm_bf16(a, b, out=c, variant=14),
             "bf16_v17": lambda: ops.gemm_bf16(a, b, out=c, variant=17),
             "scaled_mm": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16), "fp8_default": lambda: ops.gemm_fp8(a8, b8, out=c),
             "fp8_v14": lambda: ops.gemm_fp8(a8, b8, out=c, variant=14), "fp8_v17": lambda: ops.gemm_fp8(a8, b8, out=c, variant=17)}
    if M % 256 == 0:
        cands["bf16_v16"] = lambda: ops.gemm_bf16(a, b, out=c, variant=16)
        cands["fp8_v16"] = lambda: ops.gemm_fp8(a8, b8, out=c, variant=16)
    if ops.stream_shape_ok(M, N, 2 * K):
        cands["bf16_v10"] = lambda: ops.gemm_bf16(a, b, out=c, variant=10)
    if ops.stream_shape_ok(M, N, K):
        cands["fp8_v7"] = lambda: ops.gemm_fp8(a8, b8, out=c, variant=7)
    iters = max(10, min(100, int(2e12 / (2.0 * M * N * K))))
    runs = {k: [] for k in cands}
    for r in range(5):
        for k, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[k].append(e0.elapsed_time(e1) * 1e3 / iters)
    med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items()}
    med["tiles_256"] = (M // 256) * (N // 256)
    best = min((v, k) for k, v in med.items() if k.startswith("bf16_v"))
    best8 = min((v, k) for k, v in med.items() if k.startswith("fp8_v"))
    med["best_bf16"] = f"{best[1]} {round(med['hipblaslt'] / best[0], 3)}x hipBLASLt"
    med["best_fp8"] = f"{best8[1]} {round(med['scaled_mm'] / best8[0], 3)}x _scaled_mm"
    out["us"][f"{M}x{N}x{K}"] = med
    print(f"{M}x{N}x{K}", json.dumps(med), flush=True)
    json.dump(out, open(_o, "w"), indent=1)
