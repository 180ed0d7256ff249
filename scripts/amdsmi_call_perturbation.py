"""Which telemetry call disturbs a running kernel? The bf16 burner runs with a HIP event pair around EVERY kernel
while one amdsmi call at a time is repeated at a fixed rate from a second thread; per call: the kernels' mean, p99
and max duration next to an undisturbed leg. (The rocprofv3 overhead leg found one ~7 ms kernel per monitor start;
this names the call.)"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import amdsmi
from k8s_llm_monitor_amd import ops

dev = torch.device("cuda:0")
N = 8192
a = (torch.rand(N, N, device=dev) * 2 - 1).to(torch.bfloat16)
b = (torch.rand(N, N, device=dev) * 2 - 1).to(torch.bfloat16)
c = torch.empty(N, N, device=dev, dtype=torch.bfloat16)
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]

CALLS = {
    "none": None,
    "gpu_metrics_info": lambda: amdsmi.amdsmi_get_gpu_metrics_info(h),
    "vram_usage": lambda: amdsmi.amdsmi_get_gpu_vram_usage(h),
    "total_ecc_count": lambda: amdsmi.amdsmi_get_gpu_total_ecc_count(h),
    "xgmi_error_status": lambda: amdsmi.amdsmi_gpu_xgmi_error_status(h),
    "energy_count": lambda: amdsmi.amdsmi_get_energy_count(h),
    "power_cap_info": lambda: amdsmi.amdsmi_get_power_cap_info(h),
    "process_list": lambda: amdsmi.amdsmi_get_gpu_process_list(h),
    "link_metrics": lambda: amdsmi.amdsmi_get_link_metrics(h),
    "init_shutdown_second_handle": "reinit",
}


def kernels(n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record()
        ops.gemm_bf16(a, b, out=c)
        e1.record()
    torch.cuda.synchronize()
    d = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return {"kernels": n, "mean_us": round(sum(d) / n, 2), "median_us": round(d[n // 2], 2), "p99_us": round(d[int(n * 0.99)], 2),
            "max_us": round(d[-1], 2), "over_1ms": sum(1 for x in d if x > 1000.0)}


kernels(400)    # clocks
rows = {}
hz = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
for name, fn in CALLS.items():
    stop = threading.Event()
    stat = {"calls": 0, "errors": 0, "ms": 0.0, "error": None}

    def loop():
        while not stop.wait(1.0 / hz):
            t0 = time.perf_counter()
            try:
                if fn == "reinit":
                    # what a second monitor process starting up does (the library is reference counted in-process,
                    # so this is the cheap in-process form: enumerate handles and static info again)
                    for hh in amdsmi.amdsmi_get_processor_handles():
                        amdsmi.amdsmi_get_gpu_asic_info(hh); amdsmi.amdsmi_get_gpu_board_info(hh)
                else:
                    fn()
            except Exception as e:
                stat["errors"] += 1
                stat["error"] = f"{type(e).__name__}: {str(e)[:80]}"
            stat["calls"] += 1
            stat["ms"] += (time.perf_counter() - t0) * 1e3

    th = None
    if fn is not None:
        th = threading.Thread(target=loop, daemon=True)
        th.start()
    r = kernels(2500)
    stop.set()
    if th:
        th.join()
    r.update(call_hz=hz if fn else 0, calls=stat["calls"], call_errors=stat["errors"], call_error=stat["error"],
             call_mean_ms=round(stat["ms"] / stat["calls"], 3) if stat["calls"] else None)
    rows[name] = r
    print(name, json.dumps(r), flush=True)
base = rows["none"]["mean_us"]
for k, r in rows.items():
    r["mean_increase_pct"] = round((r["mean_us"] - base) / base * 100.0, 3)
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/amdsmi_call_perturbation.json"
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump({"kernel": "gemm_bf16 8192^3", "call_rate_hz": hz, "rows": rows}, open(out, "w"), indent=1)
