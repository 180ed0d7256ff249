"""Which row pitch do the persistent streams like? The same problem on operands whose rows are K bytes + pad apart, for a
list of pads (0 = dense), bf16 and fp8: TFLOP/s per pad, next to the vendor library on the dense operands."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/gemm_pitch_pad.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
PADS = (0, 64, 128, 256, 512, 768, 1024, 1280, 2048, 2304, 4096, 4352)
out = {}
one = torch.ones((), device=dev)


def pitched(t, pad_bytes):
    if not pad_bytes:
        return t
    rows, k = t.shape
    pe = pad_bytes // t.element_size()
    buf = torch.empty(rows, k + pe, dtype=t.dtype, device=dev)
    v = buf[:, :k]
    v.copy_(t)
    return v


def bench(fn, flop, iters):
    vals = []
    for r in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); e1.synchronize()
        if r:
            vals.append(flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return round(max(vals), 1)


for fmt, (M, N, K) in (("bf16", (12288, 12288, 12288)), ("bf16", (8192, 8192, 10240)), ("bf16", (8192, 8192, 8192)), ("fp8", (12288, 12288, 12288)),
                       ("fp8", (8192, 8192, 8192)), ("fp8", (4096, 4096, 4096)), ("bf16", (4096, 4096, 4096))):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    if fmt == "fp8":
        a, b = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    flop = 2.0 * M * N * K
    iters = max(5, min(100, int(1e13 / flop)))
    row = {}
    bt = b.t()
    row["vendor"] = bench((lambda: torch.matmul(a, bt, out=c)) if fmt == "bf16" else (lambda: torch._scaled_mm(a, bt, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)), flop, iters)
    for pad in PADS:
        pa, pb = pitched(a, pad), pitched(b, pad)
        if fmt == "bf16":
            fn = lambda: ops.gemm_bf16(pa, pb, out=c, variant=13)      # the plain persistent stream: no split, no repack
        else:
            fn = (lambda: ops.gemm_fp8(pa, pb, out=c, variant=4))
        row[f"pad_{pad}"] = bench(fn, flop, iters)
        del pa, pb
    out[f"{fmt}_{M}x{N}x{K}"] = row
    print(fmt, M, N, K, json.dumps(row), flush=True)
    json.dump(out, open(_o, "w"), indent=1)
    del a, b, c
