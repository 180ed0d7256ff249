"""A/B of the relaxed first wait after an epilogue (tile_chain, after_stores): the tree's build against the build before it
(_native/lib/exp/libklm_ops_strict.so), persistent bf16 / fp4 streams over K (the store stall is a fixed cost per round of tiles,
so it weighs most at small K), both libraries in one process, alternating, medians; same bits required."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda:0")
L = os.path.join(ROOT, "k8s_llm_monitor_amd/_native/lib")
libs = {"strict": ctypes.CDLL(L + "/exp/libklm_ops_strict.so"), "relaxed_tree": ctypes.CDLL(L + "/libklm_ops.so")}
vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
for l in libs.values():
    l.klm_gemm_stream.argtypes, l.klm_gemm_stream.restype = [i, vp, vp, vp, i, i, i, i, i, i, i, vp, sz, vp], i
g = torch.Generator(device=dev); g.manual_seed(11)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/exp/relaxed_wait_ab.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
ws = torch.empty(256 * 262144, dtype=torch.uint8, device=dev)
out = {}
s = torch.cuda.current_stream().cuda_stream
for (M, N, K) in ((8192, 8192, 512), (8192, 8192, 1024), (8192, 8192, 2048), (8192, 8192, 8192), (6144, 6144, 6144), (12288, 12288, 4096)):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a4 = torch.randint(0, 256, (M, K // 2), device=dev, dtype=torch.uint8, generator=g)
    b4 = torch.randint(0, 256, (N, K // 2), device=dev, dtype=torch.uint8, generator=g)
    c = {k: torch.empty(M, N, device=dev, dtype=torch.bfloat16) for k in libs}
    for name, code, x, y, kb in (("bf16", 0, a, b, 2 * K), ("fp4", 2, a4, b4, K // 2)):
        if kb < 512:
            continue
        calls = {k: (lambda k=k: libs[k].klm_gemm_stream(code, x.data_ptr(), y.data_ptr(), c[k].data_ptr(), M, N, kb, 0, 0, 1, 0, ws.data_ptr(), ws.numel(), s)) for k in libs}
        rc = {k: f() for k, f in calls.items()}
        torch.cuda.synchronize()
        same = bool(torch.equal(c["strict"], c["relaxed_tree"]))
        flop = 2.0 * M * N * K
        iters = max(10, int(6e13 / flop))
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
                    runs[k].append(e0.elapsed_time(e1) / iters * 1e3)
        med = {k: round(sorted(v)[len(v) // 2], 2) for k, v in runs.items()}
        out[f"{name} {M}x{N}x{K}"] = {"rc": rc, "same_bits": same, "us_median": med, "relaxed_over_strict_time": round(med["relaxed_tree"] / med["strict"], 4)}
        print(name, M, N, K, out[f"{name} {M}x{N}x{K}"], flush=True)
        json.dump(out, open(_o, "w"), indent=1)
