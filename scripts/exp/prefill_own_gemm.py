"""Llama-3-8B prefill with the library path's projections on hipBLASLt (torch.matmul) and on this repo's persistent stream
(VERDICT r2 next-round 8): the four projection shapes one by one, then whole prefills at 4 x 1024 and 1 x 4096 tokens,
interleaved rounds, medians."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
from k8s_llm_monitor_amd.serving import Engine, EngineConfig
from k8s_llm_monitor_amd.serving import bench as sb

dev = torch.device("cuda:0")
out = {"shapes": {}, "prefill": {}}
g = torch.Generator(device=dev); g.manual_seed(11)
T = 4096
for name, (N, K) in (("wqkv", (6144, 4096)), ("wo", (4096, 4096)), ("w_gate_up", (28672, 4096)), ("w_down", (4096, 14336)), ("lm_head_4rows", (128256, 4096))):
    M = 256 if name.startswith("lm_head") else T
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    pa, pb = ops.padded_operand(M, K, torch.bfloat16, dev), ops.padded_operand(N, K, torch.bfloat16, dev)
    pa.copy_(a); pb.copy_(b)
    bt = b.t()
    cands = {"hipblaslt": lambda: torch.matmul(a, bt, out=c), "klm_default": lambda: ops.gemm_bf16(a, b, out=c),
             "klm_plain": lambda: ops.gemm_bf16(a, b, out=c, variant=8), "klm_padded_operands": lambda: ops.gemm_bf16(pa, pb, out=c)}
    flop = 2.0 * M * N * K
    runs = {k: [] for k in cands}
    for r in range(4):
        for k, fn in cands.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            if r:
                runs[k].append(flop * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items()}
    med["plan"] = ops.gemm_stream_plan(M, N, 2 * K, mode=1)
    med["camping_pitch"] = ops.camping_pitch(2 * K)
    out["shapes"][f"{name} {M}x{N}x{K}"] = med
    print(name, med, flush=True)
    del a, b, c, pa, pb

eng = Engine(EngineConfig(model="llama3-8b", device=str(dev), max_num_seqs=8, max_model_len=4096, kv_blocks=2048))
for n_seqs, plen in ((4, 1024), (1, 4096), (2, 1024)):
    rows = {"lib": [], "own": []}
    for r in range(4):
        for mode in ("lib", "own"):
            eng.model.own_gemm = mode == "own"
            x = sb.prefill_ms(eng, n_seqs, plen, iters=3, warmup=1)
            if r:
                rows[mode].append(x["tokens_per_s"])
    med = {k: sorted(v)[len(v) // 2] for k, v in rows.items()}
    out["prefill"][f"{n_seqs}x{plen}"] = {"tokens_per_s_hipblaslt": round(med["lib"], 1), "tokens_per_s_klm": round(med["own"], 1),
                                          "klm_over_hipblaslt": round(med["own"] / med["lib"], 4), "backend": eng.model.prefill_gemm_backend(n_seqs * plen)}
    print(n_seqs, plen, out["prefill"][f"{n_seqs}x{plen}"], flush=True)
# logits of the two paths agree to bf16 rounding
eng.model.own_gemm = True
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prefill_own_gemm.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
