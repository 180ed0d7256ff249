"""Sampling overhead on the GPU side: per-kernel burner durations from rocprofv3 --kernel-trace, monitor on vs off."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_llm_monitor_amd.validation import rocprof
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/overhead_rocprof"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
res = rocprof.overhead_kernel_trace(out, repeats=repeats, seconds=5.0)
print(json.dumps(res), flush=True)
with open(os.path.join(out, "overhead_rocprof.json"), "w") as f:
    json.dump(res, f, indent=1)
