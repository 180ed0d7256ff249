"""Burner duty as rocprofv3's kernel trace sees it vs the burner's own event-based figure."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_llm_monitor_amd.validation import rocprof
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rocprof_duty"
rows = rocprof.cross_check_duty(out, duties=(25.0, 50.0, 75.0), seconds=4.0)
for r in rows:
    print(json.dumps(r), flush=True)
with open(os.path.join(out, "duty_crosscheck.json"), "w") as f:
    json.dump(rows, f, indent=1)
