"""Schedule "chain": where should the stage-free wait sit? The tile reads its own A rows 6 - 7 at gaps 0 - 3 and waits for them
(and frees the LDS stage) at gap CHAIN_FREE; at 6 the wait comes 3 MFMAs after the last read, earlier than an LDS read returns.
Builds with (free, first DMA gap, step, landed) = 6,9,5,56 (the tree at the time) / 9,12,5,56 / 12,15,5,58 in one process, alternating.
The two extra libraries: `KLM_GEN_CHAIN=9,12,5,56 python scripts/gen_gemm_quad_asm.py <copy of ops/csrc>/gemm_quad_asm.inc`, then hipcc
-shared gemm_bf16.hip gemm_mx.hip hbm.hip of that copy into _native/lib/exp/libklm_ops_free9.so (free12 alike). Outcome: 9,12,5,56
became the default (profiles/r3_prof/chain_free_gap.json)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda:0")
L = os.path.join(ROOT, "k8s_llm_monitor_amd/_native/lib")
# KLM_CHAIN_LIBS: names of other builds under _native/lib/exp (a later sweep: the landed gap and the DMA spacing)
libs = {"free6_tree": ctypes.CDLL(L + "/libklm_ops.so")}
for _n in os.environ.get("KLM_CHAIN_LIBS", "free9,free12").split(","):
    libs[_n] = ctypes.CDLL(L + f"/exp/libklm_ops_{_n}.so")
vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
for l in libs.values():
    l.klm_gemm_stream.argtypes, l.klm_gemm_stream.restype = [i, vp, vp, vp, i, i, i, i, i, i, i, vp, sz, vp], i
g = torch.Generator(device=dev); g.manual_seed(11)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/exp/chain_free_gap.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
out = {}
s = torch.cuda.current_stream().cuda_stream
for (M, N, K) in ((8192, 8192, 8192), (4096, 4096, 4096), (8192, 8192, 1024), (2048, 2048, 8192)):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a4 = torch.randint(0, 256, (M, K // 2), device=dev, dtype=torch.uint8, generator=g)
    b4 = torch.randint(0, 256, (N, K // 2), device=dev, dtype=torch.uint8, generator=g)
    c = {k: torch.empty(M, N, device=dev, dtype=torch.bfloat16) for k in libs}
    for name, code, x, y, kb in (("bf16", 0, a, b, 2 * K), ("fp4", 2, a4, b4, K // 2)):
        calls = {k: (lambda k=k: libs[k].klm_gemm_stream(code, x.data_ptr(), y.data_ptr(), c[k].data_ptr(), M, N, kb, 0, 0, 1, 0, None, 0, s)) for k in libs}
        rc = {k: f() for k, f in calls.items()}
        torch.cuda.synchronize()
        same = all(torch.equal(c["free6_tree"], c[k]) for k in libs)
        flop = 2.0 * M * N * K
        iters = max(10, int(8e13 / flop))
        runs = {k: [] for k in calls}
        for r in range(7):
            for k, f in calls.items():
                f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    f()
                e1.record(); e1.synchronize()
                if r:
                    runs[k].append(flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items()}
        out[f"{name} {M}x{N}x{K}"] = {"rc": rc, "same_bits": same, "tflops_median": med}
        print(name, M, N, K, out[f"{name} {M}x{N}x{K}"], flush=True)
        json.dump(out, open(_o, "w"), indent=1)
