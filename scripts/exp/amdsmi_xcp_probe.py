"""What does amdsmi report per XCD (xcp_stats) and for PCIe on this box? Prints the relevant gpu_metrics fields."""
import json

import amdsmi

amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
m = amdsmi.amdsmi_get_gpu_metrics_info(h)
keys = sorted(m.keys())
print("keys:", keys)
for k in keys:
    if "xcp" in k or "pcie" in k or "jpeg" in k or "vcn" in k or "num_partition" in k:
        v = m[k]
        print(k, "=>", json.dumps(v, default=str)[:1500])
try:
    print("pcie_info:", json.dumps(amdsmi.amdsmi_get_pcie_info(h), default=str)[:1200])
except Exception as e:
    print("pcie_info failed:", e)
amdsmi.amdsmi_shut_down()
