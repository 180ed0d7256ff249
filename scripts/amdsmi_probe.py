"""Dump what the amdsmi Python bindings return on this box (every call guarded)."""
import json, sys, time
import amdsmi
from amdsmi import AmdSmiTemperatureType as TT, AmdSmiTemperatureMetric as TM, AmdSmiClkType as CT, AmdSmiMemoryType as MT

def j(x):
    try:
        json.dumps(x); return x
    except TypeError:
        if isinstance(x, dict): return {str(k): j(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)): return [j(v) for v in x]
        return repr(x)

def call(out, name, fn, *a):
    t0 = time.perf_counter()
    try:
        out[name] = j(fn(*a))
    except Exception as e:
        out[name] = "ERR " + type(e).__name__ + ": " + str(e)[:200]
    out[name + "__ms"] = round((time.perf_counter() - t0) * 1e3, 3)

def main():
    res = {}
    amdsmi.amdsmi_init()
    call(res, "lib_version", amdsmi.amdsmi_get_lib_version)
    hs = amdsmi.amdsmi_get_processor_handles()
    res["n"] = len(hs)
    for i, h in enumerate(hs[:2]):
        o = res[f"gpu{i}"] = {}
        call(o, "asic", amdsmi.amdsmi_get_gpu_asic_info, h)
        call(o, "board", amdsmi.amdsmi_get_gpu_board_info, h)
        call(o, "bdf", amdsmi.amdsmi_get_gpu_device_bdf, h)
        call(o, "uuid", amdsmi.amdsmi_get_gpu_device_uuid, h)
        call(o, "activity", amdsmi.amdsmi_get_gpu_activity, h)
        call(o, "vram_usage", amdsmi.amdsmi_get_gpu_vram_usage, h)
        call(o, "mem_used_vram", amdsmi.amdsmi_get_gpu_memory_usage, h, MT.VRAM)
        call(o, "mem_total_vram", amdsmi.amdsmi_get_gpu_memory_total, h, MT.VRAM)
        call(o, "power", amdsmi.amdsmi_get_power_info, h)
        call(o, "power_cap", amdsmi.amdsmi_get_power_cap_info, h)
        for nm, t in (("edge", TT.EDGE), ("hotspot", TT.HOTSPOT), ("vram", TT.VRAM)):
            call(o, "temp_" + nm, amdsmi.amdsmi_get_temp_metric, h, t, TM.CURRENT)
        call(o, "temp_hotspot_crit", amdsmi.amdsmi_get_temp_metric, h, TT.HOTSPOT, TM.CRITICAL)
        call(o, "clock_gfx", amdsmi.amdsmi_get_clock_info, h, CT.GFX)
        call(o, "clock_mem", amdsmi.amdsmi_get_clock_info, h, CT.MEM)
        call(o, "ecc_total", amdsmi.amdsmi_get_gpu_total_ecc_count, h)
        call(o, "violation", amdsmi.amdsmi_get_violation_status, h)
        call(o, "link_metrics", amdsmi.amdsmi_get_link_metrics, h)
        call(o, "xgmi_info", amdsmi.amdsmi_get_xgmi_info, h)
        call(o, "xgmi_link_status", amdsmi.amdsmi_get_gpu_xgmi_link_status, h)
        call(o, "metrics_info", amdsmi.amdsmi_get_gpu_metrics_info, h)
        call(o, "process_list", amdsmi.amdsmi_get_gpu_process_list, h)
        call(o, "pcie", amdsmi.amdsmi_get_pcie_info, h)
        call(o, "vram_info", amdsmi.amdsmi_get_gpu_vram_info, h)
        call(o, "busy_percent", amdsmi.amdsmi_get_gpu_busy_percent, h)
        call(o, "utilization_count", amdsmi.amdsmi_get_utilization_count, h, [amdsmi.AmdSmiUtilizationCounterType.COARSE_GRAIN_GFX_ACTIVITY, amdsmi.AmdSmiUtilizationCounterType.COARSE_GRAIN_MEM_ACTIVITY])
        call(o, "energy", amdsmi.amdsmi_get_energy_count, h)
    amdsmi.amdsmi_shut_down()
    json.dump(res, sys.stdout, indent=1)

main()
