"""End-to-end soak on one MI355X (BASELINE.json config 2): the real server on amdsmi, a 50 % duty
MFMA burner and an HBM hold that crosses the 70 % tier, everything observed through HTTP only."""
import json, os, sys, time, urllib.request
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd.config import load_config
from k8s_llm_monitor_amd.loadgen.burner import Burner, HbmHold
from k8s_llm_monitor_amd.server.http import MonitorServer
from k8s_llm_monitor_amd.server.sampler import Monitor
from k8s_llm_monitor_amd.collectors.vllm import parse_exposition

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/soak.json"
cfg = load_config(env={}, overrides={"server": {"port": 0, "host": "127.0.0.1"}, "periods": {"realtime": 1.0, "history_step": 2.0},
                                     "sources": {"gpu": "amdsmi", "kube": "none", "vllm": "none", "history": "local"}})
mon = Monitor(cfg).start(); srv = MonitorServer(mon, cfg).start()
get = lambda p: json.loads(urllib.request.urlopen(srv.url + p, timeout=5).read())
# a dashboard tab would hold /api/events open: count what is pushed to it (one `realtime` event per sampler tick)
import socket, threading
pushed = {"bytes": 0, "buf": b""}
def sse():
    c = socket.create_connection(("127.0.0.1", srv.port), timeout=60)
    c.sendall(b"GET /api/events HTTP/1.1\r\nHost: x\r\n\r\n")
    try:
        while True:
            chunk = c.recv(65536)
            if not chunk:
                break
            pushed["bytes"] += len(chunk); pushed["buf"] += chunk
    except OSError:
        pass
threading.Thread(target=sse, daemon=True).start()
timeline = []
def poll(tag):
    g = get("/api/gpu/metrics")[0]; a = get("/api/alerts")
    timeline.append({"t": round(time.time() - t0, 2), "phase": tag, "gfx": g["gfxActivityAvg"], "hbmPercent": round(g["hbmPercent"], 2),
                     "powerW": g["powerW"], "temp": g["temperature"], "clock": g["clockGfxMHz"],
                     "alerts": {k: [x["key"] for x in v] for k, v in a.items() if v}})
t0 = time.time()
for _ in range(3): time.sleep(1.0); poll("idle")
b = Burner("gemm_bf16", size=8192, duty=50.0); b.start(120)
for _ in range(8): time.sleep(1.0); poll("burner50")
with HbmHold() as hold:
    hold.fill_to_percent(72.0)
    for _ in range(6): time.sleep(1.0); poll("burner50+hbm72")
burn = b.stop()
for _ in range(4): time.sleep(1.0); poll("released")
hist = get("/api/history"); stats = get("/api/monitor/stats")
metrics = urllib.request.urlopen(srv.url + "/metrics", timeout=5).read().decode()
doc = {"device": torch.cuda.get_device_name(0), "burner": burn, "timeline": timeline,
       "history_points": {k: len(v["data"]) for k, v in hist.items() if isinstance(v, dict) and "data" in v},
       "sampler": stats["tasks"]["realtime"], "collectors_ms": stats["collectors"], "metrics_series": len(parse_exposition(metrics)),
       "alert_events": get("/api/alerts/history")}
events = {k: pushed["buf"].count(b"event: " + k.encode()) for k in ("realtime", "alerts", "pods", "vllm")}
doc["sse"] = {"events": events, "bytes": pushed["bytes"], "seconds": round(time.time() - t0, 1)}
srv.stop(); mon.stop()
mid = [r for r in timeline if r["phase"] == "burner50"][3:]
doc["checks"] = {"gfx_mean_during_burner": sum(r["gfx"] for r in mid) / len(mid), "duty_achieved": burn["duty_achieved"],
                 "hbm_alert_seen": any("gpu/0/hbm" in sum(r["alerts"].values(), []) for r in timeline if r["phase"] == "burner50+hbm72"),
                 "hbm_alert_cleared": not any("gpu/0/hbm" in sum(r["alerts"].values(), []) for r in timeline[-2:]),
                 "sse_realtime_events": events["realtime"], "sse_alert_events": events["alerts"]}
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps(doc["checks"])); print(json.dumps(timeline[-1])); print("saved", out_path)
assert abs(doc["checks"]["gfx_mean_during_burner"] - burn["duty_achieved"]) < 3.0 and doc["checks"]["hbm_alert_seen"] and doc["checks"]["hbm_alert_cleared"]
assert events["realtime"] >= 15 and events["alerts"] >= 2, events          # a tick per second for ~21 s; first paint + the HBM alert
