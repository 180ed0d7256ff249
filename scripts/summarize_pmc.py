"""Condense rocprofv3 --pmc CSVs into per-kernel raw counters + derived shares (our kernels only)."""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_llm_monitor_amd.validation import rocprof
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    per = rocprof.read_counters(files[0])
    for k, c in per.items():
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        names = sys.argv[2].split(",") if len(sys.argv) > 2 else ("gemm_bf16", "gemm_fp8", "gemm_mx", "gemm_quad_asm", "Cijk", "hbm_")
        if not any(s in short for s in names):
            continue
        n = c.get("dispatches", 1) or 1
        e = out.setdefault(short, {})
        e[os.path.basename(d)] = {"per_dispatch": {x: round(v / n, 1) for x, v in c.items() if x != "dispatches"},
                                  "dispatches": n, "derived": rocprof.derive(c)}
print(json.dumps(out, indent=1))
