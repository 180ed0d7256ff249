"""Row pitch: the hand-scheduled streams on dense operands vs the same operands in a padded allocation (pitch + 256 / 512 /
1024 bytes), at the K extents whose dense pitch the memory channels camp on. Same bits, TFLOP/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
M = N = 8192
out = {}


def padded(t, pad_bytes):
    p = torch.empty((t.shape[0], t.shape[1] + pad_bytes // t.element_size()), dtype=t.dtype, device=dev)[:, :t.shape[1]]
    p.copy_(t)
    return p


for K in [int(x) for x in os.environ.get("KLM_KS", "8192,10240,12288,14336,20480,24576").split(",")]:
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    bt = b.t()
    ref, ref8 = ops.gemm_bf16(a, b, variant=8), ops.gemm_fp8(a8, b8, variant=4)
    cands = {"bf16_dense": lambda: ops.gemm_bf16(a, b, out=c, variant=8), "hipblaslt": lambda: torch.matmul(a, bt, out=c),
             "fp8_dense": lambda: ops.gemm_fp8(a8, b8, out=c, variant=4)}
    eq = {}
    for pad in (256, 512, 1024):
        ap, bp, a8p, b8p = padded(a, pad), padded(b, pad), padded(a8, pad), padded(b8, pad)
        eq[f"bf16+{pad}"] = bool(torch.equal(ops.gemm_bf16(ap, bp, variant=8), ref) and torch.equal(ops.gemm_bf16(ap, bp, variant=4), ref))
        eq[f"fp8+{pad}"] = bool(torch.equal(ops.gemm_fp8(a8p, b8p, variant=4), ref8) and torch.equal(ops.gemm_fp8(a8p, b8p, variant=2), ref8))
        cands[f"bf16+{pad}"] = lambda ap=ap, bp=bp: ops.gemm_bf16(ap, bp, out=c, variant=8)
        cands[f"fp8+{pad}"] = lambda a8p=a8p, b8p=b8p: ops.gemm_fp8(a8p, b8p, out=c, variant=4)
    flop = 2.0 * M * N * K
    runs = {k: [] for k in cands}
    for r in range(4):
        for name, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[name].append(flop * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    out[str(K)] = {"equal": eq, "tflops": {k: sorted(v)[len(v) // 2] for k, v in runs.items()}}
    print(K, "equal:", all(eq.values()), {k: round(v) for k, v in out[str(K)]["tflops"].items()}, flush=True)
    assert all(eq.values()), eq
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_pitch_check.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
