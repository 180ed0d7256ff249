"""On a power-capped card, do idle cycles cost time? The chain stream with `s_sleep N` (64 N cycles) in every K tile of ~2150 cycles
(KLM_GEN_CHAIN_SLEEP=2 / 4 / 8: +6 / +12 / +24 % cycles per GEMM, no extra work) next to the unmodified one, bf16 and fp4 8192^3,
6 s per arm under the amdsmi energy accumulator. If the card is ENERGY-limited, the clock rises and the time barely moves (until
the clock reaches its 2.4 GHz maximum) -- which is the argument why a higher MFMA-busy share (hiding the epilogue) would not by
itself make the fp4 stream faster. Libraries: _native/lib/exp/libklm_ops_sleep{2,4,8}.so (generator env + hipcc, as chain_free_gap.py)."""
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import amdsmi
from k8s_llm_monitor_amd import ops

seconds, settle, rounds = 6.0, 2.5, 2
dev = torch.device("cuda:0")
n = 8192
g = torch.Generator(device=dev); g.manual_seed(7)
a = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
bt = b.t()
a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
one = torch.ones((), device=dev)
a4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8, generator=g)
b4 = torch.randint(0, 256, (n, n // 2), device=dev, dtype=torch.uint8, generator=g)
L = os.path.join(ROOT, "k8s_llm_monitor_amd/_native/lib")
libs = {"sleep0": ctypes.CDLL(L + "/libklm_ops.so")}
for k in (2, 4, 8):
    libs[f"sleep{k}"] = ctypes.CDLL(L + f"/exp/libklm_ops_sleep{k}.so")
vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
for l in libs.values():
    l.klm_gemm_stream.argtypes, l.klm_gemm_stream.restype = [i, vp, vp, vp, i, i, i, i, i, i, i, vp, sz, vp], i
st = torch.cuda.current_stream().cuda_stream


def stream(lib, code, x, y, kb):
    return lambda: libs[lib].klm_gemm_stream(code, x.data_ptr(), y.data_ptr(), c.data_ptr(), n, n, kb, 0, 0, 1, 0, None, 0, st)


arms = {}
for k in libs:
    arms[f"bf16_{k}"] = stream(k, 0, a, b, 2 * n)
for k in libs:
    arms[f"fp4_{k}"] = stream(k, 2, a4, b4, n // 2)

amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
res_uj = float(amdsmi.amdsmi_get_energy_count(h).get("counter_resolution") or 15.3)


def flat(x):
    out = []
    for r in (x if isinstance(x, (list, tuple)) else [x]):
        out += [v for v in (r if isinstance(r, (list, tuple)) else [r]) if isinstance(v, (int, float)) and not isinstance(v, bool) and 0 < v < 65535]
    return out


def watch(rows, stop):
    while not stop.is_set():
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        rows.append({"t": time.perf_counter(), "fw": m.get("firmware_timestamp"), "e": m.get("energy_accumulator"),
                     "p": m.get("current_socket_power"), "clk": m.get("current_gfxclk"), "clks": flat(m.get("current_gfxclks")),
                     "avg_clk": m.get("average_gfxclk_frequency"), "temp": m.get("temperature_hotspot"),
                     "uclk": m.get("current_uclk"), "ppt": m.get("ppt_residency_acc"), "acc": m.get("accumulation_counter")})
        stop.wait(0.2)


def run(name, fn):
    fn(); torch.cuda.synchronize()
    rows, stop = [], threading.Event()
    th = threading.Thread(target=watch, args=(rows, stop)); th.start()
    t0 = time.perf_counter()
    done, marks = 0, []
    while time.perf_counter() - t0 < seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); e1.synchronize()
        marks.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 20.0))
    stop.set(); th.join()
    late = [ms for t, ms in marks if t >= settle]
    rows = [r for r in rows if r["t"] - t0 >= settle]
    tables = [r for i, r in enumerate(rows) if i == 0 or r["fw"] != rows[i - 1]["fw"]]     # one row per refresh of the table
    r0, r1 = tables[0], tables[-1]
    dt = (r1["fw"] - r0["fw"]) * 1e-8
    clks = [sum(r["clks"]) / len(r["clks"]) for r in tables if r["clks"]]
    ms = sum(late) / len(late)
    return {"ms": ms, "tflops": 2.0 * n ** 3 / (ms * 1e-3) / 1e12, "tables": len(tables),
            "power_energy_w": (r1["e"] - r0["e"]) * res_uj * 1e-6 / dt if dt > 0 else None,
            "power_gauge_w": sum(r["p"] for r in tables) / len(tables),
            "gfxclk_mhz": sum(r["clk"] for r in tables) / len(tables),
            "gfxclk_xcd_mean_mhz": sum(clks) / len(clks) if clks else None,
            "gfxclk_xcd_last": tables[-1]["clks"], "avg_clk": tables[-1]["avg_clk"], "uclk": tables[-1]["uclk"],
            "temp_c": tables[-1]["temp"],
            "ppt_residency_share": ((r1["ppt"] - r0["ppt"]) / (r1["acc"] - r0["acc"])) if isinstance(r1["ppt"], int) and isinstance(r1["acc"], int) and r1["acc"] != r0["acc"] else None}


out = {"n": n, "seconds": seconds, "settle": settle, "runs": {k: [] for k in arms}}
for r in range(rounds):
    for name, fn in arms.items():
        rec = run(name, fn)
        out["runs"][name].append(rec)
        print(r, name, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in rec.items()}, flush=True)
out["mean"] = {k: {f: sum(x[f] for x in v) / len(v) for f in ("tflops", "power_energy_w", "power_gauge_w", "gfxclk_mhz", "gfxclk_xcd_mean_mhz", "temp_c") if all(isinstance(x[f], (int, float)) for x in v)}
               for k, v in out["runs"].items()}
# cycles per GEMM implied by the reported clock
for k, v in out["mean"].items():
    clk = v.get("gfxclk_xcd_mean_mhz") or v.get("gfxclk_mhz")
    if clk:
        v["implied_cycles_per_gemm"] = clk * 1e6 * (2.0 * n ** 3 / (v["tflops"] * 1e12))
    # energy per GEMM: the power over the measured window times one GEMM's share of it
    if v.get("power_energy_w"):
        v["joule_per_gemm"] = v["power_energy_w"] * (2.0 * n ** 3 / (v["tflops"] * 1e12))
    print(k, {f: round(x, 4 if f == "joule_per_gemm" else 1) for f, x in v.items()}, flush=True)
_o = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/exp/idle_cycles_on_the_cap.json"
os.makedirs(os.path.dirname(_o) or ".", exist_ok=True)
json.dump(out, open(_o, "w"), indent=1)
