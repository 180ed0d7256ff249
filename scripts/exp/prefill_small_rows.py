"""Llama-3-8B projections at FEW token rows (256 .. 2048) on this repo's kernels (the cost model picks the 128 x 128 ring kernel
or the stream) against hipBLASLt, then whole prefills with the serving model's own-GEMM threshold lowered to 256 rows.
Decides `LlamaModel.OWN_GEMM_MIN_ROWS`."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
from k8s_llm_monitor_amd.serving import Engine, EngineConfig
from k8s_llm_monitor_amd.serving import bench as sb

dev = torch.device("cuda:0")
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/prefill_small_rows.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
out = {"shapes": {}, "prefill": {}}
g = torch.Generator(device=dev); g.manual_seed(11)
for T in (256, 512, 768, 1024, 1536, 2048):
    for name, (N, K) in (("wqkv", (6144, 4096)), ("wo", (4096, 4096)), ("w_gate_up", (28672, 4096)), ("w_down", (4096, 14336))):
        a = (torch.rand(T, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        c = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
        bt = b.t()
        cands = {"hipblaslt": lambda: torch.matmul(a, bt, out=c), "klm": lambda: ops.gemm_bf16(a, b, out=c)}
        if ops.stream_shape_ok(T, N, 2 * K):
            cands["klm_stream"] = lambda: ops.gemm_bf16(a, b, out=c, variant=10)
        cands["klm_p128"] = lambda: ops.gemm_bf16(a, b, out=c, variant=14)
        runs = {k: [] for k in cands}
        for r in range(5):
            for k, fn in cands.items():
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record(); e1.synchronize()
                if r:
                    runs[k].append(e0.elapsed_time(e1) * 1e3 / 30)
        med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items()}
        med["picked"] = ops.pick_bf16_variant(T, N, K)
        med["klm_over_hipblaslt_speed"] = round(med["hipblaslt"] / med["klm"], 3)
        out["shapes"][f"{T} {name} {N}x{K}"] = med
        print(T, name, med, flush=True)
        del a, b, c
json.dump(out, open(_o, "w"), indent=1)
eng = Engine(EngineConfig(model="llama3-8b", device=str(dev), max_num_seqs=8, max_model_len=4096, kv_blocks=2048))
type(eng.model).OWN_GEMM_MIN_ROWS = 256
for n_seqs, plen in ((1, 256), (1, 512), (1, 1024), (2, 768), (1, 2048), (4, 1024)):
    rows = {"lib": [], "own": []}
    for r in range(4):
        for mode in ("lib", "own"):
            eng.model.own_gemm = mode == "own"
            x = sb.prefill_ms(eng, n_seqs, plen, iters=3, warmup=1)
            if r:
                rows[mode].append(x["tokens_per_s"])
    med = {k: sorted(v)[len(v) // 2] for k, v in rows.items()}
    out["prefill"][f"{n_seqs}x{plen}"] = {"tokens_per_s_hipblaslt": round(med["lib"], 1), "tokens_per_s_klm": round(med["own"], 1),
                                          "klm_over_hipblaslt": round(med["own"] / med["lib"], 4)}
    print(n_seqs, plen, out["prefill"][f"{n_seqs}x{plen}"], flush=True)
json.dump(out, open(_o, "w"), indent=1)
