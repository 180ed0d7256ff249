"""A/B of the ring pipeline schedules (gemm_ring256.h) against each other and the vendor library, one process,
interleaved rounds, uniform [-1, 1) operands. Before timing: every variant is checked against an fp32 reference at a
small shape, against the vendor result at the timed shape, and for run-to-run bit-identity (a race in the
LDS-DMA / barrier protocol shows up as a run that differs from the first).

  python scripts/exp/ab_sched.py [N] [rounds] -> JSON on stdout (progress on stderr)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
dev = torch.device("cuda:0")
log = lambda *a: print(*a, file=sys.stderr, flush=True)


def rnd(m, k):
    return (torch.rand(m, k, device=dev) * 2 - 1)


def check_small():
    out = {}
    for (M, Nn, K) in ((512, 768, 1024), (256, 256, 256), (1024, 512, 4096 + 128), (256, 512, 4096 + 256)):
        a32, b32 = rnd(M, K), rnd(Nn, K)
        a, b = a32.bfloat16(), b32.bfloat16()
        ref = a.float() @ b.float().t()
        for v in (1, 2, 3):
            if v == 3 and (K % 128 or K < 256):
                continue
            c = ops.gemm_bf16(a, b, variant=v).float()
            err = (c - ref).abs().max().item() / ref.abs().max().item()
            out[f"bf16_v{v}_{M}x{Nn}x{K}"] = err
            assert err < 1e-2, (v, M, Nn, K, err)
        a8, b8 = a32.to(torch.float8_e4m3fn), b32.to(torch.float8_e4m3fn)
        ref8 = a8.float() @ b8.float().t()
        for v in (0, 1):
            c = ops.gemm_fp8(a8, b8, variant=v).float()
            err = (c - ref8).abs().max().item() / ref8.abs().max().item()
            out[f"fp8_v{v}_{M}x{Nn}x{K}"] = err
            assert err < 1e-2, (v, M, Nn, K, err)
        if K < 512 or K % 256:
            continue                                      # fp4 rows are K/2 bytes: at least two 128-byte K tiles
        a4 = torch.randint(0, 256, (M, K // 2), device=dev, dtype=torch.uint8)
        b4 = torch.randint(0, 256, (Nn, K // 2), device=dev, dtype=torch.uint8)
        ref4 = (ops.fp4_decode(a4).double() @ ops.fp4_decode(b4).double().t()).float()
        for v in (0, 1):
            c = ops.gemm_fp4(a4, b4, variant=v).float()
            # bf16 output rounding only: the products and fp32 sums are exact
            assert torch.equal(c, ref4.bfloat16().float()), (v, M, Nn, K)
            out[f"fp4_v{v}_{M}x{Nn}x{K}"] = 0.0
    return out


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(max(3, iters // 10)):
        fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    res = {"N": N, "device": torch.cuda.get_device_name(0), "small_checks": check_small()}
    log("small checks ok")
    a32, b32 = rnd(N, N), rnd(N, N)
    a, b = a32.bfloat16(), b32.bfloat16()
    a8, b8 = a32.to(torch.float8_e4m3fn), b32.to(torch.float8_e4m3fn)
    a4 = torch.randint(0, 256, (N, N // 2), device=dev, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (N, N // 2), device=dev, dtype=torch.uint8)
    del a32, b32
    c = torch.empty(N, N, device=dev, dtype=torch.bfloat16)
    one = torch.ones((), device=dev)
    bt, b8t = b.t(), b8.t()
    # ---- big-shape agreement and race screen
    ref = torch.matmul(a, bt)
    big = {}
    for name, fn in (("bf16_v1", lambda: ops.gemm_bf16(a, b, out=c, variant=1)), ("bf16_v2", lambda: ops.gemm_bf16(a, b, out=c, variant=2)),
                     ("bf16_v3", lambda: ops.gemm_bf16(a, b, out=c, variant=3)),
                     ("fp8_v0", lambda: ops.gemm_fp8(a8, b8, out=c, variant=0)), ("fp8_v1", lambda: ops.gemm_fp8(a8, b8, out=c, variant=1)),
                     ("fp4_v0", lambda: ops.gemm_fp4(a4, b4, out=c, variant=0)), ("fp4_v1", lambda: ops.gemm_fp4(a4, b4, out=c, variant=1))):
        fn()
        first = c.clone()
        same = True
        for _ in range(12):
            fn()
            same = same and bool(torch.equal(c, first))
        big[name] = {"bit_identical_13_runs": same}
        if name.startswith("bf16"):
            big[name]["max_rel_vs_hipblaslt"] = ((first.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
        del first
    big["fp8_v1_equals_v0"] = bool(torch.equal(ops.gemm_fp8(a8, b8, variant=0), ops.gemm_fp8(a8, b8, variant=1)))
    big["fp4_v1_equals_v0"] = bool(torch.equal(ops.gemm_fp4(a4, b4, variant=0), ops.gemm_fp4(a4, b4, variant=1)))
    big["bf16_v2_equals_v1"] = bool(torch.equal(ops.gemm_bf16(a, b, variant=1), ops.gemm_bf16(a, b, variant=2)))
    big["bf16_v3_equals_v1"] = bool(torch.equal(ops.gemm_bf16(a, b, variant=1), ops.gemm_bf16(a, b, variant=3)))
    res["big_checks"] = big
    log("big checks", json.dumps(big))
    assert all(v["bit_identical_13_runs"] for k, v in big.items() if isinstance(v, dict)), big
    assert big["fp8_v1_equals_v0"] and big["fp4_v1_equals_v0"] and big["bf16_v2_equals_v1"] and big["bf16_v3_equals_v1"], big

    arms = {
        "bf16_v1": (lambda: ops.gemm_bf16(a, b, out=c, variant=1), 2.0 * N ** 3),
        "bf16_v2": (lambda: ops.gemm_bf16(a, b, out=c, variant=2), 2.0 * N ** 3),
        "bf16_v3": (lambda: ops.gemm_bf16(a, b, out=c, variant=3), 2.0 * N ** 3),
        "bf16_hipblaslt": (lambda: torch.matmul(a, bt, out=c), 2.0 * N ** 3),
        "fp8_v0": (lambda: ops.gemm_fp8(a8, b8, out=c, variant=0), 2.0 * N ** 3),
        "fp8_v1": (lambda: ops.gemm_fp8(a8, b8, out=c, variant=1), 2.0 * N ** 3),
        "fp8_scaled_mm": (lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16), 2.0 * N ** 3),
        "fp4_v0": (lambda: ops.gemm_fp4(a4, b4, out=c, variant=0), 2.0 * N ** 3),
        "fp4_v1": (lambda: ops.gemm_fp4(a4, b4, out=c, variant=1), 2.0 * N ** 3),
    }
    only = os.environ.get("KLM_ARMS")
    if only:
        arms = {k: v for k, v in arms.items() if k in only.split(",")}
    for fn, _ in arms.values():            # warm everything (clocks, hipBLASLt heuristics)
        timed(fn, 20)
    runs = {k: [] for k in arms}
    t0 = time.time()
    for r in range(ROUNDS):
        for k, (fn, flop) in arms.items():
            runs[k].append(flop / timed(fn, 60) / 1e12)
        log(f"round {r} ({time.time() - t0:.0f}s):", {k: round(v[-1], 1) for k, v in runs.items()})
    res["tflops"] = {k: {"median": sorted(v)[len(v) // 2], "min": min(v), "max": max(v), "runs": [round(x, 1) for x in v]} for k, v in runs.items()}
    med = lambda k: res["tflops"][k]["median"] if k in res["tflops"] else None
    res["ratios"] = {"bf16_v3_over_hipblaslt": med("bf16_v3") / med("bf16_hipblaslt") if med("bf16_v3") and med("bf16_hipblaslt") else None,
                     "bf16_v3_over_v2": med("bf16_v3") / med("bf16_v2") if med("bf16_v3") and med("bf16_v2") else None,
                     "bf16_v2_over_v1": med("bf16_v2") / med("bf16_v1") if med("bf16_v2") and med("bf16_v1") else None,
                     "bf16_v2_over_hipblaslt": med("bf16_v2") / med("bf16_hipblaslt") if med("bf16_v2") and med("bf16_hipblaslt") else None,
                     "bf16_v1_over_hipblaslt": med("bf16_v1") / med("bf16_hipblaslt") if med("bf16_v1") and med("bf16_hipblaslt") else None,
                     "fp8_v1_over_v0": med("fp8_v1") / med("fp8_v0") if med("fp8_v1") and med("fp8_v0") else None,
                     "fp8_v1_over_scaled_mm": med("fp8_v1") / med("fp8_scaled_mm") if med("fp8_v1") and med("fp8_scaled_mm") else None,
                     "fp4_v1_over_v0": med("fp4_v1") / med("fp4_v0") if med("fp4_v1") and med("fp4_v0") else None}
    print(json.dumps(res))


main()
