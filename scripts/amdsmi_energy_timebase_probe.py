"""How often does the metrics table move, and which clock goes with its energy accumulator? Samples the raw table at
20 Hz under the burner and prints host time, firmware_timestamp, system_clock_counter, energy_accumulator, power."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amdsmi
from k8s_llm_monitor_amd.loadgen.burner import Burner
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
res = amdsmi.amdsmi_get_energy_count(h)
print("energy_count:", res)
b = Burner("gemm_bf16", size=8192, duty=100.0)
b.start(60.0)
time.sleep(4.0)
rows = []
t_end = time.perf_counter() + 6.0
while time.perf_counter() < t_end:
    t = time.perf_counter()
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    rows.append({"t": t, "fw": m.get("firmware_timestamp"), "sys": m.get("system_clock_counter"), "e": m.get("energy_accumulator"),
                 "p": m.get("current_socket_power"), "acc": m.get("accumulation_counter")})
    time.sleep(0.05)
e2 = amdsmi.amdsmi_get_energy_count(h)
b.stop()
r0, r1 = rows[0], rows[-1]
changes = sum(1 for a, c in zip(rows, rows[1:]) if a["e"] != c["e"])
out = {"samples": len(rows), "table_changes": changes, "host_dt_s": r1["t"] - r0["t"],
       "fw_dt_ticks": r1["fw"] - r0["fw"], "fw_ticks_per_host_s": (r1["fw"] - r0["fw"]) / (r1["t"] - r0["t"]),
       "sys_dt": None if r0["sys"] is None else r1["sys"] - r0["sys"],
       "acc_dt": None if r0["acc"] is None else r1["acc"] - r0["acc"],
       "energy_counts": r1["e"] - r0["e"], "resolution_uj": res.get("counter_resolution"),
       "power_by_host_clock_w": (r1["e"] - r0["e"]) * res.get("counter_resolution") * 1e-6 / (r1["t"] - r0["t"]),
       "power_by_fw_clock_10ns_w": (r1["e"] - r0["e"]) * res.get("counter_resolution") * 1e-6 / ((r1["fw"] - r0["fw"]) * 1e-8),
       "gauge_mean_w": sum(r["p"] for r in rows) / len(rows), "energy_count_end": e2,
       "first_rows": [{k: (round(v - r0[k], 4) if k == "t" else (v - r0[k] if isinstance(v, int) and k != "p" else v)) for k, v in r.items()} for r in rows[:30]]}
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/energy_timebase.json", "w"), indent=1)
