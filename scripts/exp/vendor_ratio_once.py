"""One box's answer: the default GEMM and HBM kernels next to the vendor's, same process (validation.benchlib.vendor_comparison)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from k8s_llm_monitor_amd import ops
from k8s_llm_monitor_amd.validation.benchlib import vendor_comparison
dev = torch.device("cuda:0")
out = vendor_comparison(torch, ops, dev, 8192, rounds=7, iters=20)
out["device"] = torch.cuda.get_device_name(dev)
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    out["uuid_tail"] = str(amdsmi.amdsmi_get_gpu_device_uuid(h))[-8:]
except Exception as e:
    out["uuid_tail"] = f"n/a ({type(e).__name__})"
print(json.dumps(out))
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/vendor_ratio.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
