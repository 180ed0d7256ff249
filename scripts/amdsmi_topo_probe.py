"""Dump what amdsmi's xGMI / link / topology calls return on this box (every call guarded).

Pins the field names and units that collectors/gpu.py::XgmiTopology builds on. On the one-GPU
gpurun box the hive's other seven GPUs are not visible to the container: the per-link counters are
live (other tenants' traffic) but peers cannot be named, and this dump records exactly that."""
import json
import sys
import time

import amdsmi


def j(x):
    try:
        json.dumps(x)
        return x
    except TypeError:
        if isinstance(x, dict):
            return {str(k): j(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [j(v) for v in x]
        return repr(x)


def call(out, name, fn, *a):
    t0 = time.perf_counter()
    try:
        out[name] = j(fn(*a))
    except Exception as e:
        out[name] = "ERR " + type(e).__name__ + ": " + str(e)[:300]
    out[name + "__ms"] = round((time.perf_counter() - t0) * 1e3, 3)


def main():
    res = {}
    amdsmi.amdsmi_init()
    call(res, "lib_version", amdsmi.amdsmi_get_lib_version)
    hs = list(amdsmi.amdsmi_get_processor_handles())
    res["n"] = len(hs)
    LT = amdsmi.AmdSmiLinkType
    res["link_type_enum"] = {m.name: int(m.value) for m in LT}
    for i, h in enumerate(hs):
        o = res[f"gpu{i}"] = {}
        call(o, "bdf", amdsmi.amdsmi_get_gpu_device_bdf, h)
        call(o, "xgmi_info", amdsmi.amdsmi_get_xgmi_info, h)
        call(o, "xgmi_error_status", amdsmi.amdsmi_gpu_xgmi_error_status, h)
        call(o, "xgmi_link_status", amdsmi.amdsmi_get_gpu_xgmi_link_status, h)
        call(o, "xgmi_plpd", amdsmi.amdsmi_get_xgmi_plpd, h)
        call(o, "numa_affinity", amdsmi.amdsmi_get_gpu_topo_numa_affinity, h)
        call(o, "numa_node", amdsmi.amdsmi_topo_get_numa_node_number, h)
        call(o, "link_metrics_a", amdsmi.amdsmi_get_link_metrics, h)
        for m in LT:
            call(o, f"nearest_{m.name}", amdsmi.amdsmi_get_link_topology_nearest, h, m)
        for k, h2 in enumerate(hs):
            if k == i:
                continue
            call(o, f"link_type_to_{k}", amdsmi.amdsmi_topo_get_link_type, h, h2)
            call(o, f"link_weight_to_{k}", amdsmi.amdsmi_topo_get_link_weight, h, h2)
            call(o, f"p2p_status_to_{k}", amdsmi.amdsmi_topo_get_p2p_status, h, h2)
            call(o, f"p2p_accessible_to_{k}", amdsmi.amdsmi_is_P2P_accessible, h, h2)
        # counter cadence and unit: two readings 1.5 s apart next to the metrics table's own copy
        m0 = amdsmi.amdsmi_get_gpu_metrics_info(h)
        time.sleep(1.5)
        call(o, "link_metrics_b", amdsmi.amdsmi_get_link_metrics, h)
        m1 = amdsmi.amdsmi_get_gpu_metrics_info(h)
        o["metrics_xgmi"] = {k: [j(m0.get(k)), j(m1.get(k))] for k in
                             ("xgmi_read_data_acc", "xgmi_write_data_acc", "xgmi_link_status", "xgmi_link_width", "xgmi_link_speed")}
    amdsmi.amdsmi_shut_down()
    json.dump(res, sys.stdout, indent=1)


main()
