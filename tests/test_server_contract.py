"""HTTP contract of SURVEY.md 2.5 against a live server on an ephemeral port with fake
sources (BASELINE.json config 1), plus the routing behaviour of monitor_server.js:243-299."""
import json
import re
import time
import urllib.error
import urllib.request

import pytest

from k8s_llm_monitor_amd.config import demo_config
from k8s_llm_monitor_amd.parallel import probe
from k8s_llm_monitor_amd.server.http import MonitorServer
from k8s_llm_monitor_amd.server.sampler import Monitor


@pytest.fixture(scope="module")
def server(tmp_path_factory):
    state = tmp_path_factory.mktemp("klm")
    cfg = demo_config(periods={"realtime": 0.1, "pods": 0.1, "vllm": 0.1, "history_step": 1.0},
                      sources={"probe_results": str(state / "probe.json")})
    mon = Monitor(cfg).start()
    srv = MonitorServer(mon, cfg).start()
    time.sleep(0.5)
    yield srv
    srv.stop()
    mon.stop()


def get(srv, path, method="GET"):
    req = urllib.request.Request(srv.url + path, method=method)
    try:
        with urllib.request.urlopen(req, timeout=5) as r:
            return r.status, dict(r.headers), r.read()
    except urllib.error.HTTPError as e:
        return e.code, dict(e.headers), e.read()


def get_json(srv, path):
    st, hd, body = get(srv, path)
    assert st == 200, (path, st, body)
    assert hd["Content-Type"] == "application/json; charset=utf-8"        # monitor_server.js:246
    assert hd["Access-Control-Allow-Origin"] == "*"                         # :244
    assert hd["Access-Control-Allow-Methods"] == "GET, OPTIONS"             # :245
    return json.loads(body)


def test_host_metrics_shape(server):
    h = get_json(server, "/api/host/metrics")
    assert set(h) == {"cpu", "memory", "disk"}
    assert isinstance(h["cpu"]["load_1min"], float) and re.fullmatch(r"\d+\.\d", h["cpu"]["percent"])
    for sec in ("memory", "disk"):
        assert isinstance(h[sec]["total"], int) and isinstance(h[sec]["used"], int) and re.fullmatch(r"\d+\.\d", h[sec]["percent"])


def test_gpu_metrics_shape(server):
    g = get_json(server, "/api/gpu/metrics")
    assert isinstance(g, list) and len(g) == 8
    for x in g:
        assert {"name", "utilization", "memoryUsed", "memoryTotal", "temperature"} <= set(x)
        assert all(isinstance(x[k], int) for k in ("utilization", "memoryUsed", "memoryTotal", "temperature"))
        assert x["memoryTotal"] == 294896 and {"index", "hbmPercent", "powerW", "xgmi", "ecc", "throttle"} <= set(x)


def test_pods_shape(server):
    pods = get_json(server, "/api/k8s/pods")
    assert len(pods) == 11
    for p in pods:
        assert {"namespace", "name", "status", "restarts", "age"} <= set(p)
        assert isinstance(p["restarts"], int) and re.fullmatch(r"(\d+d \d+h|\d+h \d+m|\d+m)", p["age"])


def test_history_shape(server):
    h = get_json(server, "/api/history")
    for k in ("cpu", "memory", "disk", "gpuUtil", "gpuMem", "gpuTemp"):
        assert set(h[k]) == {"labels", "data"} and len(h[k]["labels"]) == len(h[k]["data"])
        assert all(isinstance(v, str) for v in h[k]["data"]) and all(re.fullmatch(r"\d\d:\d\d", l) for l in h[k]["labels"])
    assert h["source"] == "local" and len(h["gpuUtil"]["data"]) >= 1


def test_alerts_shape_and_idempotence(server):
    a = get_json(server, "/api/alerts")
    assert set(a) == {"minor", "serious", "critical"} and all(isinstance(a[t], list) for t in a)
    # reading alerts does not consume them: two clients see the same thing
    b = get_json(server, "/api/alerts")
    assert {x["key"] for t in a.values() for x in t} == {x["key"] for t in b.values() for x in t}
    before = server.monitor._alert_state.evaluations
    for _ in range(5):
        get_json(server, "/api/alerts")
    time.sleep(0.05)
    assert server.monitor._alert_state.evaluations - before < 5          # handlers never evaluate


def test_static_routes(server):
    for p in ("/", "/monitor.html"):
        st, hd, body = get(server, p)
        assert st == 200 and hd["Content-Type"] == "text/html; charset=utf-8" and b"<canvas" in body
    st, hd, body = get(server, "/logo.svg")
    assert st == 200 and hd["Content-Type"] == "image/svg+xml" and body.startswith(b"<svg")


def test_query_string_is_ignored(server):
    assert get(server, "/api/gpu/metrics?x=1&y=2")[0] == 200               # :250


def test_options_404_and_methods(server):
    st, hd, body = get(server, "/api/alerts", "OPTIONS")
    assert st == 200 and body == b"" and hd["Access-Control-Allow-Origin"] == "*"      # :248
    st, hd, body = get(server, "/api/nope")
    assert st == 404 and body == b"Not Found"                               # :290
    assert get(server, "/api/alerts", "POST")[0] == 405


def test_traversal_route_is_gone(server):
    for p in ("/danyichun", "/danyichun/../../etc/passwd", "/danyichun%2f..%2f..%2fetc%2fpasswd", "/../../etc/passwd"):
        st, _, body = get(server, p)
        assert st == 404 and b"root:" not in body
    assert get(server, "/healthz")[0] == 200                                 # and the server is still alive


def test_handler_exception_is_500_json(server, monkeypatch):
    monkeypatch.setattr(server.monitor, "history", lambda: 1 / 0)
    st, _, body = get(server, "/api/history")
    assert st == 500 and "error" in json.loads(body)                         # :292-294


def test_additions(server):
    v = get_json(server, "/api/vllm/metrics")
    assert v["available"] and {"ttftP50", "ttftP99", "tokensPerSec", "kvCachePercent", "running", "waiting"} <= set(v)
    x = get_json(server, "/api/xgmi")
    assert len(x["gpus"]) == 8 and len(x["gpus"][0]["links"]) == 7 and x["linkPeakGBps"] == 153.0
    s = get_json(server, "/api/summary")
    assert {"host", "gpus", "pods", "vllm", "alerts", "ages"} <= set(s)
    st = get_json(server, "/api/monitor/stats")
    assert st["tasks"]["realtime"]["calls"] >= 2 and st["sources"]["gpu"].startswith("fake(")
    assert get_json(server, "/api/config")["server"]["cors_origin"] == "*"
    status, hd, body = get(server, "/metrics")
    assert status == 200 and hd["Content-Type"].startswith("text/plain") and b'klm_gpu_hbm_used_bytes{gpu="7"' in body
    from k8s_llm_monitor_amd.collectors.vllm import parse_exposition
    assert len(parse_exposition(body.decode())) > 100                        # our own exposition parses


def test_rccl_probe_endpoint(server):
    assert get_json(server, "/api/rccl/probe")["curves"] == []
    pts = [probe.point(8, b, b / 300e9) for b in probe.sweep_sizes()]
    probe.save_results(probe.results_doc([probe.curve(8, pts)]), server.cfg.sources.probe_results)
    d = get_json(server, "/api/rccl/probe")
    assert d["curves"][0]["n_gpus"] == 8 and d["scaling"][0]["ceiling_mesh_GBps"] == pytest.approx(1071.0)


def test_sse_stream_delivers_alerts(server):
    import socket
    s = socket.create_connection(("127.0.0.1", server.port), timeout=5)
    s.sendall(b"GET /api/events HTTP/1.1\r\nHost: x\r\n\r\n")
    buf = b""
    while b"event: alerts" not in buf:
        buf += s.recv(4096)
    assert b"text/event-stream" in buf and b'"minor"' in buf
    s.close()


def test_missing_page_is_fatal_like_the_reference(tmp_path):
    cfg = demo_config(server={"static_dir": str(tmp_path)})
    mon = Monitor(cfg)
    with pytest.raises(FileNotFoundError):
        MonitorServer(mon, cfg)                                              # readFileSync at :241
    mon.stop()
