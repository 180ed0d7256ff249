"""Where does storing C past the caches (`sc1 nt`) pay? Persistent streams with write-back C vs streamed C over M = N and
the operands' row bytes, bf16 / fp8 / fp4 (rows padded where the pitch camps). Ratio > 1: streaming is faster."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
out = []
for MN in (4096, 8192, 12288, 16384):
    for row in (2048, 4096, 8192, 12288, 16384, 24576, 32768):
        if MN * row * 2 > (3 << 30):
            continue
        raw = torch.randint(0, 256, (2, MN, row), device=dev, generator=g, dtype=torch.uint8)
        c = torch.empty(MN, MN, device=dev, dtype=torch.bfloat16)
        rec = {"MN": MN, "row_bytes": row, "operands_MB": 2 * MN * row / 1e6, "c_MB": MN * MN * 2 / 1e6}
        for kind in ("bf16", "fp8", "fp4"):
            if kind == "bf16":
                f = ((raw.float() / 128.0) - 1.0).to(torch.bfloat16)            # [2, MN, row] bf16 -> K = row / 2 columns used
                src = f[:, :, : row // 2]
                fn = ops.gemm_bf16; wb, st = 8, 9
            elif kind == "fp8":
                src = ((raw.float() / 128.0) - 1.0).to(torch.float8_e4m3fn)
                fn = ops.gemm_fp8; wb, st = 4, 5
            else:
                src = raw
                fn = ops.gemm_fp4; wb, st = 3, 4
            a = ops.padded_operand(MN, src.shape[2], src.dtype, dev); a.copy_(src[0])
            b = ops.padded_operand(MN, src.shape[2], src.dtype, dev); b.copy_(src[1])
            del src
            assert torch.equal(fn(a, b, variant=wb), fn(a, b, variant=st))
            runs = {wb: [], st: []}
            iters = max(3, min(20, int(4e9 / (MN * MN * row))))
            for r in range(5):
                for v in (wb, st):
                    fn(a, b, out=c, variant=v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        fn(a, b, out=c, variant=v)
                    e1.record(); e1.synchronize()
                    if r:
                        runs[v].append(e0.elapsed_time(e1) / iters * 1e3)
            t_wb, t_st = sorted(runs[wb])[2], sorted(runs[st])[2]
            rec[kind] = {"us_writeback": t_wb, "us_streamed": t_st, "ratio": t_wb / t_st}
            del a, b
        out.append(rec)
        print(MN, row, f"ops {rec['operands_MB']:.0f} MB C {rec['c_MB']:.0f} MB", {k: round(rec[k]["ratio"], 3) for k in ("bf16", "fp8", "fp4")},
              {k: round(rec[k]["us_writeback"], 1) for k in ("bf16", "fp8", "fp4")}, flush=True)
        del raw, c
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stream_c_sweep.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
