"""When does storing C past the caches (`sc1 nt`) pay in a MULTI-round GEMM? Found by accident at fp8 8192 x 8192 x 12288: +8 %
(A + B = 201 MB just fit the 256 MB Infinity Cache; 128 MB of C written through it evict them). Sweep the operand bytes for
square-ish outputs in both formats: persistent stream with store 0 / 1, alternating, medians. Output: ratio store1 / store0.
--sets N: cycle through N different operand / output sets (a repeated GEMM on ONE set keeps its operands in the Infinity Cache
from launch to launch, which no real caller does; 3 sets of >= 134 MB operands defeat that, like the training step's six matrices)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
_o = next((x for x in sys.argv[1:] if not x.startswith("--")), "gpurun_out/exp/stream_c_mall_window.json")
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
SETS = int(sys.argv[sys.argv.index("--sets") + 1]) if "--sets" in sys.argv else 1
ONLY = "--window" in sys.argv
out = []
shapes = []
for MN in (4096, 6144, 8192, 12288):
    for kb in (2048, 4096, 6144, 8192, 10240, 12288, 16384, 20480, 24576):     # row bytes
        ab = 2 * MN * kb
        if (48e6 <= ab <= 420e6 and not ONLY) or (ONLY and MN >= 8192 and 130e6 <= ab <= 310e6):
            shapes.append((MN, kb))
for dt, code in (("fp8", 1), ("bf16", 0)):
    for MN, kb in shapes:
        K = kb if dt == "fp8" else kb // 2
        M = N = MN
        sets = []
        for _ in range(SETS):
            a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
            b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
            if dt == "fp8":
                a, b = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
            # operands on a pitch that does not camp, so that only the store policy differs
            pa, pb = ops.padded_operand(M, K, a.dtype, dev), ops.padded_operand(N, K, b.dtype, dev)
            pa.copy_(a); pb.copy_(b)
            sets.append((pa, pb, torch.empty(M, N, device=dev, dtype=torch.bfloat16)))
            del a, b
        pa = sets[0][0]
        pitch = pa.stride(0) * pa.element_size()
        flop = 2.0 * M * N * K
        iters = max(6, int(2e13 / flop))
        iters -= iters % SETS
        cnt = [0]

        def call(s):
            x, y, z = sets[cnt[0] % SETS]
            cnt[0] += 1
            ops._stream(code, x, y, z, M, N, kb, pitch if pitch != kb else 0, store=s, mode=1, rot=0, what=dt)
        fns = {s: (lambda s=s: call(s)) for s in (0, 1)}
        runs = {0: [], 1: []}
        for r in range(6):
            for s, fn in fns.items():
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record(); e1.synchronize()
                if r:
                    runs[s].append(flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        med = {s: sorted(v)[len(v) // 2] for s, v in runs.items()}
        plan = ops.gemm_stream_plan(M, N, kb)
        row = {"dtype": dt, "sets": SETS, "M": M, "N": N, "row_bytes": kb, "pitch": pitch, "ab_MB": round(2 * MN * kb / 1e6, 1), "c_MB": round(M * N * 2 / 1e6, 1),
               "rounds": plan["rounds"], "rest": plan["rest"], "tflops_store0": round(med[0], 1), "tflops_store1": round(med[1], 1), "store1_over_store0": round(med[1] / med[0], 4)}
        out.append(row)
        print(row, flush=True)
        json.dump(out, open(_o, "w"), indent=1)
        del sets, pa
        torch.cuda.empty_cache()
