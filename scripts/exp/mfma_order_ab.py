"""A/B of the MFMA order inside a k-step on the real GEMM streams: two builds of libklm_ops.so in ONE process (ctypes), the
same operands, alternating launches (the card sits on its power cap: each side runs in the state the other left), medians.
  lib A: _native/lib/exp/libklm_ops_rows.so   (tile-row-major order, rounds 2 - 3)
  lib B: _native/lib/libklm_ops.so            (the tree's generator output)
Both must give the same bits: the order of the MFMAs across accumulators changes, no accumulator's own summation order does."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
BASE = os.environ.get("KLM_AB_BASE", "rows")      # rows: tile-row-major (rounds 2 - 3); zigzag: the zigzag order on schedule "stagger"
libs = {"rows": ctypes.CDLL(os.path.join(ROOT, f"k8s_llm_monitor_amd/_native/lib/exp/libklm_ops_{BASE}.so")),
        "tree": ctypes.CDLL(os.path.join(ROOT, "k8s_llm_monitor_amd/_native/lib/libklm_ops.so"))}
vp, i = ctypes.c_void_p, ctypes.c_int
sz = ctypes.c_size_t
for l in libs.values():
    for f in (l.klm_gemm_bf16, l.klm_gemm_fp8, l.klm_gemm_fp4):
        f.argtypes, f.restype = [vp, vp, vp, i, i, i, i, vp], i
    l.klm_gemm_stream.argtypes, l.klm_gemm_stream.restype = [i, vp, vp, vp, i, i, i, i, i, i, i, vp, sz, vp], i
g = torch.Generator(device=dev); g.manual_seed(11)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/exp/mfma_order_ab.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
out = {"base_library": BASE}
one = torch.ones((), device=dev)
for (M, N, K) in ((8192, 8192, 8192), (4096, 4096, 4096), (8192, 8192, 2048)):
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    a4 = torch.randint(0, 256, (M, K // 2), device=dev, dtype=torch.uint8, generator=g)
    b4 = torch.randint(0, 256, (N, K // 2), device=dev, dtype=torch.uint8, generator=g)
    c = {k: torch.empty(M, N, device=dev, dtype=torch.bfloat16) for k in libs}
    bt, b8t = b.t(), b8.t()
    s = torch.cuda.current_stream().cuda_stream
    # (function, A, B, variant) or ("stream", A, B, dtype, row bytes): the persistent streams, as variant=None runs them at these shapes
    cases = {"bf16_persistent": ("stream", a, b, 0, 2 * K), "bf16_per_tile_v4": ("klm_gemm_bf16", a, b, 4),
             "fp4_persistent": ("stream", a4, b4, 2, K // 2), "fp4_per_tile_v2": ("klm_gemm_fp4", a4, b4, 2),
             "fp8_persistent": ("stream", a8, b8, 1, K), "fp8_per_tile_v2": ("klm_gemm_fp8", a8, b8, 2)}
    row = {}
    for name, case in cases.items():
        fn, x, y = case[:3]
        variant = case[3]
        if fn == "stream":
            dt, kb = case[3], case[4]
            calls = {k: (lambda k=k: libs[k].klm_gemm_stream(dt, x.data_ptr(), y.data_ptr(), c[k].data_ptr(), M, N, kb, 0, 0, 1, 0, None, 0, s)) for k in libs}
        else:
            calls = {k: (lambda k=k: getattr(libs[k], fn)(x.data_ptr(), y.data_ptr(), c[k].data_ptr(), M, N, K, variant, s)) for k in libs}
        rc = {k: f() for k, f in calls.items()}
        torch.cuda.synchronize()
        same = bool(torch.equal(c["rows"], c["tree"]))
        flop = 2.0 * M * N * K
        iters = max(10, int(8e13 / flop))
        runs = {k: [] for k in calls}
        if "bf16" in name:
            calls["hipblaslt"] = lambda: torch.matmul(a, bt, out=c["rows"])
            runs["hipblaslt"] = []
        if "fp8" in name:
            calls["scaled_mm"] = lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
            runs["scaled_mm"] = []
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
        row[name] = {"variant": variant, "rc": rc, "same_bits": same, "tflops_median": med, "tree_over_rows": round(med["tree"] / med["rows"], 4)}
        for v in ("hipblaslt", "scaled_mm"):
            if v in med:
                row[name]["tree_over_" + v] = round(med["tree"] / med[v], 4)
        print(M, N, K, name, row[name], flush=True)
    out[f"{M}x{N}x{K}"] = row
    json.dump(out, open(_o, "w"), indent=1)
