"""Dense bf16 operands on a camping row pitch: the GEMM as it is, the GEMM on operands re-pitched first (copy included),
the copy alone, and hipBLASLt; interleaved rounds, medians."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(2)
out = {}
for M, N, K in ((12288, 12288, 12288), (8192, 8192, 12288), (8192, 8192, 10240), (16384, 16384, 6144), (8192, 8192, 20480), (6144, 6144, 12288)):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bt = b.t()
    def repack_gemm():
        pa, pb = ops.repitched(a, "a"), ops.repitched(b, "b")
        ops.gemm_bf16(pa, pb, out=c)
    cands = {"dense_v13": lambda: ops.gemm_bf16(a, b, out=c, variant=13), "default": lambda: ops.gemm_bf16(a, b, out=c),
             "repack_then_gemm": repack_gemm, "repack_only": lambda: (ops.repitched(a, "a"), ops.repitched(b, "b")),
             "hipblaslt": lambda: torch.matmul(a, bt, out=c)}
    same = bool(torch.equal(ops.gemm_bf16(a, b, variant=13), (repack_gemm(), c.clone())[1])) if ops.gemm_stream_plan(M, N, 2 * K)["split"] <= 1 else None
    flop = 2.0 * M * N * K
    runs = {k: [] for k in cands}
    for r in range(4):
        for k, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[k].append(e0.elapsed_time(e1) / 10)
    ms = {k: sorted(v)[len(v) // 2] for k, v in runs.items()}
    row = {k + "_tflops": round(flop / (v * 1e-3) / 1e12, 1) for k, v in ms.items() if k != "repack_only"}
    row["repack_ms"] = round(ms["repack_only"], 4)
    row["repack_TBps"] = round(2 * 2 * (M + N) * K / (ms["repack_only"] * 1e-3) / 1e12, 2)
    row["same_bits_as_dense"] = same
    row["auto_repack"] = ops.repack_worthwhile(M, N, 2 * K)
    row["default_over_hipblaslt"] = round(ms["hipblaslt"] / ms["default"], 4)
    out[f"{M}x{N}x{K}"] = row
    print(f"{M}x{N}x{K}", row, flush=True)
    del a, b, c
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_repack.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
