"""Soak of the GEMM streams on the zigzag MFMA order: every arm runs back to back for KLM_SOAK_S seconds; every 50th result is
compared bit for bit with the FIRST one (determinism) and the first one with what the tile-row-major build of the same stream
gives on the same operands (_native/lib/exp/libklm_ops_rows.so: no accumulator's summation order changed, so the bits must not)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda:0")
seconds = float(os.environ.get("KLM_SOAK_S", "8"))
libs = {"rows": ctypes.CDLL(os.path.join(ROOT, "k8s_llm_monitor_amd/_native/lib/exp/libklm_ops_%s.so" % os.environ.get("KLM_AB_BASE", "rows"))),
        "tree": ctypes.CDLL(os.path.join(ROOT, "k8s_llm_monitor_amd/_native/lib/libklm_ops.so"))}
vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
for l in libs.values():
    for f in (l.klm_gemm_bf16, l.klm_gemm_fp8, l.klm_gemm_fp4):
        f.argtypes, f.restype = [vp, vp, vp, i, i, i, i, vp], i
    l.klm_gemm_stream.argtypes, l.klm_gemm_stream.restype = [i, vp, vp, vp, i, i, i, i, i, i, i, vp, sz, vp], i
g = torch.Generator(device=dev); g.manual_seed(77)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/exp/mfma_order_soak.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
ws = torch.empty(256 * 262144, dtype=torch.uint8, device=dev)


def operands(fmt, M, N, K):
    if fmt == "fp4":
        return (torch.randint(0, 256, (M, K // 2), device=dev, generator=g, dtype=torch.uint8), torch.randint(0, 256, (N, K // 2), device=dev, generator=g, dtype=torch.uint8))
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    return (a, b) if fmt == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))


CODE = {"bf16": 0, "fp8": 1, "fp4": 2}
KB = {"bf16": lambda K: 2 * K, "fp8": lambda K: K, "fp4": lambda K: K // 2}
arms = []
for fmt in ("bf16", "fp8", "fp4"):
    arms += [(f"{fmt} persistent stream 8192^3", fmt, (8192, 8192, 8192), "stream"),
             (f"{fmt} persistent stream 6144^3 (split last round + reduce)", fmt, (6144, 6144, 6144), "stream"),
             (f"{fmt} one workgroup per tile 4096^3", fmt, (4096, 4096, 4096), "tile"),
             (f"{fmt} persistent stream 2304 x 2816, 2560-byte rows (odd pass count)", fmt, (2304, 2816, {"bf16": 1280, "fp8": 2560, "fp4": 5120}[fmt]), "stream")]
only = os.environ.get("KLM_SOAK_ONLY", "")
arms = [a for a in arms if only in a[0]]
out = {}
s = torch.cuda.current_stream().cuda_stream
for name, fmt, (M, N, K), kind in arms:
    x, y = operands(fmt, M, N, K)

    def run(lib, c):
        if kind == "stream":
            return lib.klm_gemm_stream(CODE[fmt], x.data_ptr(), y.data_ptr(), c.data_ptr(), M, N, KB[fmt](K), 0, 0, 1, 0, ws.data_ptr(), ws.numel(), s)
        fn = {"bf16": lib.klm_gemm_bf16, "fp8": lib.klm_gemm_fp8, "fp4": lib.klm_gemm_fp4}[fmt]
        return fn(x.data_ptr(), y.data_ptr(), c.data_ptr(), M, N, K, {"bf16": 4, "fp8": 2, "fp4": 2}[fmt], s)
    ref_rows = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rc = (run(libs["rows"], ref_rows), run(libs["tree"], ref))
    torch.cuda.synchronize()
    assert rc == (0, 0), (name, rc)
    same_as_rows = bool(torch.equal(ref, ref_rows))
    c = torch.empty_like(ref)
    n = checks = bad = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(50):
            run(libs["tree"], c)
        n += 50
        checks += 1
        if not torch.equal(c, ref):
            bad += 1
        c.zero_()
    out[name] = {"rc": rc, "launches": n, "checks": checks, "mismatches": bad, "same_bits_as_row_major_build": same_as_rows,
                 "tflops": round(2.0 * M * N * K * n / (time.time() - t0) / 1e12, 1)}
    print(name, out[name], flush=True)
    json.dump(out, open(_o, "w"), indent=1)
    del x, y, ref, ref_rows, c
tot = {"arms": len(out), "launches": sum(v["launches"] for v in out.values()), "mismatches": sum(v["mismatches"] for v in out.values()),
       "all_same_bits_as_row_major_build": all(v["same_bits_as_row_major_build"] for v in out.values())}
out["total"] = tot
print(tot)
json.dump(out, open(_o, "w"), indent=1)
