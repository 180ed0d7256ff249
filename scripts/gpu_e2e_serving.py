"""BASELINE.json config 4 end to end on one MI355X, as separate processes and through HTTP only:

  process 1  `serve-llm`  Llama-3-8B architecture (random weights) with a deliberately small KV cache and
                           an open-loop load that overruns it (queueing, preemption)
  process 2  `serve`       the monitor: amdsmi for the GPU, `sources.vllm = direct` pointed at process 1

This script only polls the monitor's public API and records what a dashboard user would see: the vLLM
panel, the KV-cache / waiting-queue / TTFT alerts going up and coming down, and the GPU card next to them."""
import json
import os
import subprocess
import sys
import time
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/e2e_serving.json"
LLM, MON = 18021, 18022
env = dict(os.environ, PYTHONPATH=ROOT, PYTHONUNBUFFERED="1")


def get(port, path, timeout=5):
    return json.loads(urllib.request.urlopen(f"http://127.0.0.1:{port}{path}", timeout=timeout).read())


def wait_up(port, path, seconds):
    t_end = time.time() + seconds
    while time.time() < t_end:
        try:
            urllib.request.urlopen(f"http://127.0.0.1:{port}{path}", timeout=2).read()
            return True
        except Exception:
            time.sleep(0.5)
    return False


llm = subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", "serve-llm", "--model", "llama3-8b", "--host", "127.0.0.1",
                        "--port", str(LLM), "--kv-blocks", "1200", "--max-num-seqs", "64", "--load-rps", "40",
                        "--prompt-len", "512", "--output-len", "160", "--duration", "24"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
mon = None
doc = {}
try:
    assert wait_up(LLM, "/health", 120), "serve-llm did not come up"
    menv = dict(env, KLM_SERVER_PORT=str(MON), KLM_SERVER_HOST="127.0.0.1", KLM_SOURCES_GPU="amdsmi", KLM_SOURCES_KUBE="none",
                KLM_SOURCES_HISTORY="local", KLM_SOURCES_VLLM="direct", KLM_SOURCES_VLLM_ENDPOINTS=f"http://127.0.0.1:{LLM}",
                KLM_PERIODS_REALTIME="1", KLM_PERIODS_VLLM="1", KLM_PERIODS_ALERT_HOLD="0")
    mon = subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", "serve"], cwd=ROOT, env=menv,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert wait_up(MON, "/healthz", 60), "the monitor did not come up"
    timeline, t0 = [], time.time()
    while time.time() - t0 < 34:
        time.sleep(1.0)
        v, a, g = get(MON, "/api/vllm/metrics"), get(MON, "/api/alerts"), get(MON, "/api/gpu/metrics")[0]
        timeline.append({"t": round(time.time() - t0, 1), "available": v.get("available"), "running": v.get("running"),
                         "waiting": v.get("waiting"), "kv": v.get("kvCachePercent"), "tok_s": v.get("tokensPerSec"),
                         "ttftP50": v.get("ttftP50"), "ttftP99": v.get("ttftP99"), "preempt_s": v.get("preemptionsPerSec"),
                         "gfx": g.get("gfxActivityAvg"), "mem": g.get("memActivityAvg"), "hbmPercent": round(g["hbmPercent"], 2),
                         "powerW": g.get("powerW"),
                         "alerts": {k: [x["key"] for x in al] for k, al in a.items() if al}})
    try:
        truth = get(LLM, "/stats")["truth"]
    except Exception:
        truth = None                                    # the engine has exited after its --duration
    doc = {"timeline": timeline, "engine_truth_at_end": truth, "alert_events": get(MON, "/api/alerts/history")}
finally:
    for p in (mon, llm):
        if p is not None:
            p.terminate()
    tails = {}
    for name, p in (("monitor", mon), ("serve_llm", llm)):
        if p is not None:
            try:
                tails[name] = p.communicate(timeout=30)[0][-1500:]
            except Exception:
                p.kill()
    doc["process_output_tail"] = tails

seen = sorted({k for r in doc.get("timeline", []) for ks in r["alerts"].values() for k in ks})
peak = max((r for r in doc["timeline"] if r["kv"] is not None), key=lambda r: r["kv"])
last = doc["timeline"][-1]
doc["checks"] = {"alert_keys_seen": seen, "kv_peak_percent": peak["kv"], "waiting_peak": max((r["waiting"] or 0) for r in doc["timeline"]),
                 "tok_s_peak": max((r["tok_s"] or 0) for r in doc["timeline"]),
                 "vllm_alerts_fired": any(k.startswith("vllm/") for k in seen),
                 "vllm_alerts_cleared_at_end": not any(k.startswith("vllm/kv") or k.startswith("vllm/waiting")
                                                       for ks in last["alerts"].values() for k in ks),
                 "outage_alert_after_engine_exit": "vllm/down" in last["alerts"].get("critical", [])}
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps(doc["checks"]))
for r in doc["timeline"][::3]:
    print(json.dumps(r))
assert doc["checks"]["vllm_alerts_fired"], "the overloaded engine raised no vLLM alert"
assert doc["checks"]["outage_alert_after_engine_exit"], "the vanished engine raised no outage alert"
