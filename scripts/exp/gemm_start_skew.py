"""One-round problems: do the one-workgroup-per-tile streams gain when the eight XCDs do not start in lockstep? XCD x sleeps
x * skew * 128 cycles before its first load (klm_gemm_start_skew). Interleaved with the vendor kernels, medians, microseconds.
OUTCOME (profiles/r3_prof/gemm_start_skew.json): no; every skew only adds its delay (4096^3 fp8 50.9 -> 51.2 / 51.5 / 51.9 us at
1 / 2 / 4 units). The hook (a sleep loop in front of gemm_quad_asm and the exported setter) was removed again; this script needs it
back to run."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
lib = ops.load()
lib.klm_gemm_start_skew.argtypes = [ctypes.c_int]; lib.klm_gemm_start_skew.restype = ctypes.c_int
g = torch.Generator(device=dev); g.manual_seed(5)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/gemm_start_skew.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
one = torch.ones((), device=dev)
out = {}
SKEWS = (0, 1, 2, 4, 8, 16, 32)
for M, N, K in ((4096, 4096, 4096), (4096, 4096, 2048), (3072, 3072, 3072), (4096, 4096, 8192)):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bt, b8t = b.t(), b8.t()
    want = ops.gemm_bf16(a, b, variant=4).clone()
    cands = {"hipblaslt": (0, lambda: torch.matmul(a, bt, out=c)), "scaled_mm": (0, lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16))}
    for s in SKEWS:
        cands[f"bf16_skew{s}"] = (s, lambda: ops.gemm_bf16(a, b, out=c, variant=4))
        cands[f"fp8_skew{s}"] = (s, lambda: ops.gemm_fp8(a8, b8, out=c, variant=2))
    runs = {k: [] for k in cands}
    for r in range(6):
        for k, (s, fn) in cands.items():
            lib.klm_gemm_start_skew(s)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[k].append(e0.elapsed_time(e1) * 1e3 / 50)
    lib.klm_gemm_start_skew(8)
    same = bool(torch.equal(ops.gemm_bf16(a, b, variant=4), want))
    lib.klm_gemm_start_skew(0)
    med = {k: round(sorted(v)[len(v) // 2], 2) for k, v in runs.items()}
    med["same_bits_with_skew"] = same
    out[f"{M}x{N}x{K}"] = med
    print(M, N, K, json.dumps(med), flush=True)
    json.dump(out, open(_o, "w"), indent=1)
