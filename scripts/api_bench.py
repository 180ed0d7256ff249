"""API serving benchmark: this monitor vs the reference server, same machine, same client.

The reference publishes no numbers, but its server can be run: it is two
interpreted files. This script starts `node /root/reference/monitor_server.js` (port 8888,
fixed in its source) and this repo's server in demo mode, gives both the same fake Prometheus
on :9090, and measures per endpoint the latency (p50 / p99) and the throughput with C
concurrent keep-alive-less clients, like browsers polling.

Caveats that favour the reference: on this host `nvidia-smi` and `kubectl` do not exist, so its
collectors fail fast (a spawn error) instead of running; with the real tools every
/api/alerts request would pay three process spawns.
"""
import json, os, statistics, subprocess, sys, threading, time, urllib.request
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ENDPOINTS = ["/api/host/metrics", "/api/gpu/metrics", "/api/k8s/pods", "/api/alerts", "/api/history", "/"]


def hammer(base, path, clients, seconds):
    lat, errs = [], [0]
    stop = time.time() + seconds
    lock = threading.Lock()

    def work():
        mine = []
        while time.time() < stop:
            t0 = time.perf_counter()
            try:
                with urllib.request.urlopen(base + path, timeout=10) as r:
                    r.read()
                mine.append(time.perf_counter() - t0)
            except Exception:
                errs[0] += 1
        with lock:
            lat.extend(mine)

    th = [threading.Thread(target=work) for _ in range(clients)]
    t0 = time.time()
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.time() - t0
    lat.sort()
    if not lat:
        return {"rps": 0.0, "errors": errs[0]}
    return {"rps": round(len(lat) / dt, 1), "p50_ms": round(lat[len(lat) // 2] * 1e3, 2),
            "p99_ms": round(lat[min(len(lat) - 1, int(len(lat) * 0.99))] * 1e3, 2), "errors": errs[0]}


def wait_up(url, t=10):
    end = time.time() + t
    while time.time() < end:
        try:
            urllib.request.urlopen(url, timeout=1).read(); return True
        except Exception:
            time.sleep(0.1)
    return False


def main():
    clients = int(os.environ.get("CLIENTS", "8")); seconds = float(os.environ.get("SECONDS", "3"))
    from k8s_llm_monitor_amd.fakes.prometheus import FakePrometheus
    prom = FakePrometheus(port=9090).start()
    out = {"clients": clients, "seconds": seconds, "endpoints": {}}
    ref = None
    if os.path.exists("/root/reference/monitor_server.js") and subprocess.run(["which", "node"], capture_output=True).returncode == 0:
        ref = subprocess.Popen(["node", "/root/reference/monitor_server.js"], cwd="/root/reference", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert wait_up("http://127.0.0.1:8888/api/host/metrics"), "reference server did not start"
    # this repo's server in its own process too, so that the Python client's GIL is not shared with it
    env = dict(os.environ, KLM_SOURCES_GPU="fake", KLM_SOURCES_KUBE="fake", KLM_SOURCES_VLLM="fake", KLM_SOURCES_HISTORY="prometheus",
               KLM_PROMETHEUS_URL=prom.url, KLM_SERVER_PORT="18888", KLM_SERVER_HOST="127.0.0.1")
    ours = subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", "serve"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                            cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert wait_up("http://127.0.0.1:18888/healthz", 20), "this repo's server did not start"

    class srv:
        url = "http://127.0.0.1:18888"
    try:
        for ep in ENDPOINTS:
            row = {}
            if ref:
                row["reference"] = hammer("http://127.0.0.1:8888", ep, clients, seconds)
            row["this_repo"] = hammer(srv.url, ep, clients, seconds)
            if ref and row["reference"].get("rps"):
                row["speedup_rps"] = round(row["this_repo"]["rps"] / row["reference"]["rps"], 2)
            out["endpoints"][ep] = row
            print(ep, json.dumps(row), flush=True)
    finally:
        ours.terminate(); ours.wait(timeout=5)
        prom.stop()
        if ref:
            ref.terminate(); ref.wait(timeout=5)
    dst = sys.argv[1] if len(sys.argv) > 1 else "profiles/r1_api_bench.json"
    json.dump(out, open(dst, "w"), indent=1)
    print("saved", dst)


if __name__ == "__main__":
    main()
