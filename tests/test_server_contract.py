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
    assert get(server, "/api/config")[0] == 404                              # off unless server.expose_config
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


def test_sse_stream_pushes_every_panel(server):
    """First paint on connect, then one `realtime` / `pods` / `vllm` event per sampler tick: the page needs no polling."""
    import socket
    s = socket.create_connection(("127.0.0.1", server.port), timeout=5)
    s.sendall(b"GET /api/events HTTP/1.1\r\nHost: x\r\n\r\n")

    def events(buf):
        out = []
        for block in buf.split(b"\n\n"):
            m = re.match(rb"event: (\w+)\ndata: (.*)\Z", block, re.S)
            if m:
                out.append((m.group(1).decode(), json.loads(m.group(2))))
        return out

    buf = b""
    while len(events(buf.split(b"\r\n\r\n", 1)[-1])) < 4:
        buf += s.recv(65536)
    first = dict(events(buf.split(b"\r\n\r\n", 1)[1]))
    assert set(first) == {"alerts", "realtime", "pods", "vllm"}
    # the same documents as the polled endpoints (compared by shape: the sampler may tick between the two reads)
    polled = get_json(server, "/api/gpu/metrics")
    assert [g["index"] for g in first["realtime"]["gpus"]] == [g["index"] for g in polled] and len(polled) == 8
    assert set(first["realtime"]["gpus"][0]) == set(polled[0]) and set(first["realtime"]["host"]) == {"cpu", "memory", "disk"}
    assert [p["name"] for p in first["pods"]] == [p["name"] for p in get_json(server, "/api/k8s/pods")]
    assert set(first["alerts"]) == {"minor", "serious", "critical"} and "available" in first["vllm"]
    mon = server.monitor
    n0 = len(events(buf.split(b"\r\n\r\n", 1)[1]))
    mon.tick_all()
    deadline = time.time() + 5
    while time.time() < deadline:
        buf += s.recv(65536)
        ev = events(buf.split(b"\r\n\r\n", 1)[1])[n0:]
        if {"realtime", "pods", "vllm"} <= {k for k, _ in ev}:
            break
    kinds = [k for k, _ in ev]
    assert {"realtime", "pods", "vllm"} <= set(kinds)
    last = [d for k, d in ev if k == "realtime"][-1]
    assert last["host"]["memory"]["total"] > 0 and len(last["gpus"]) == 8 and "xcdActivity" in last["gpus"][0]
    s.close()


def test_missing_page_is_fatal_like_the_reference(tmp_path):
    cfg = demo_config(server={"static_dir": str(tmp_path)})
    mon = Monitor(cfg)
    with pytest.raises(FileNotFoundError):
        MonitorServer(mon, cfg)                                              # readFileSync at :241
    mon.stop()


def _raw(server, payload: bytes, want: int = 1, timeout=5.0) -> bytes:
    import socket
    s = socket.create_connection(("127.0.0.1", server.port), timeout=timeout)
    s.sendall(payload)
    buf = b""
    try:
        while buf.count(b"HTTP/1.1 ") < want or not _complete(buf):
            chunk = s.recv(65536)
            if not chunk:
                break
            buf += chunk
    finally:
        s.close()
    return buf


def _complete(buf: bytes) -> bool:
    """Every response in `buf` has its full body."""
    rest = buf
    while rest:
        head, sep, tail = rest.partition(b"\r\n\r\n")
        if not sep:
            return False
        m = re.search(rb"content-length: (\d+)", head.lower())
        n = int(m.group(1)) if m else 0
        if b"HEAD-MARK" in head:
            n = 0
        if len(tail) < n:
            return False
        rest = tail[n:]
    return True


def test_keep_alive_serves_several_requests_on_one_connection(server):
    req = b"GET /api/host/metrics HTTP/1.1\r\nHost: x\r\n\r\n" + b"GET /healthz?x=1 HTTP/1.1\r\nHost: x\r\n\r\n" + \
          b"GET /nope HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n"
    buf = _raw(server, req, want=3)
    assert buf.count(b"HTTP/1.1 200 OK") == 2 and buf.count(b"HTTP/1.1 404 Not Found") == 1
    assert buf.count(b"Connection: keep-alive") == 2 and buf.endswith(b"Not Found")


def test_head_has_headers_only_and_post_body_is_drained(server):
    import socket
    s = socket.create_connection(("127.0.0.1", server.port), timeout=5)
    s.sendall(b"HEAD /api/gpu/metrics HTTP/1.1\r\nHost: x\r\n\r\n")
    head = b""
    while b"\r\n\r\n" not in head:
        head += s.recv(4096)
    assert head.endswith(b"\r\n\r\n") and int(re.search(rb"Content-Length: (\d+)", head).group(1)) > 1000
    s.sendall(b"POST /api/alerts HTTP/1.1\r\nHost: x\r\nContent-Length: 5\r\n\r\nhelloGET /healthz HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n")
    buf = b""
    while True:
        c = s.recv(4096)
        if not c:
            break
        buf += c
    s.close()
    assert buf.startswith(b"HTTP/1.1 405") and b"HTTP/1.1 200 OK" in buf and b'"ok":true' in buf


def test_oversized_header_and_garbage_do_not_kill_the_server(server):
    assert b" 431 " in _raw(server, b"GET / HTTP/1.1\r\nX: " + b"a" * 40000 + b"\r\n\r\n")
    assert b" 400 " in _raw(server, b"\r\n\r\n")
    assert get(server, "/healthz")[0] == 200


def test_bodies_are_cached_per_snapshot_version(server):
    m = server.monitor
    for t in m.tasks.values():
        t.stop()
    time.sleep(0.15)
    a = get(server, "/api/gpu/metrics")[2]
    v = m.version
    assert server._cache["/api/gpu/metrics"][0] == v and get(server, "/api/gpu/metrics")[2] == a
    m.tick_realtime()
    assert m.version > v
    b = get(server, "/api/gpu/metrics")[2]
    assert server._cache["/api/gpu/metrics"][0] == m.version and json.loads(b)[0]["timestamp"] != json.loads(a)[0]["timestamp"]
    for t in m.tasks.values():
        t._stop.clear(); t.start()


def test_config_endpoint_redacts_the_webhook():
    from k8s_llm_monitor_amd.server.http import redacted_config
    cfg = demo_config(notify={"webhook_url": "https://hooks.example.com:8443/services/T000/B000/secret-token?x=1"})
    d = redacted_config(cfg)
    assert d["notify"]["webhook_url"] == "https://hooks.example.com:8443/..." and "secret-token" not in json.dumps(d)


def test_config_endpoint_is_opt_in_and_redacted():
    cfg = demo_config(server={"expose_config": True}, notify={"webhook_url": "https://hooks.example.com/T000/B000/secret"})
    mon = Monitor(cfg)
    srv = MonitorServer(mon, cfg).start()
    try:
        d = get_json(srv, "/api/config")
        assert d["server"]["cors_origin"] == "*" and d["notify"]["webhook_url"] == "https://hooks.example.com/..."
        assert "kube_token_file" not in d["sources"]
    finally:
        srv.stop()
        mon.stop()


def test_slow_prometheus_does_not_starve_other_routes():
    """/api/history held a lock across six network calls, so concurrent requests queued inside the
    executor and stalled the other blocking routes. Now one refresh runs and everybody else gets the previous result."""
    import threading
    from k8s_llm_monitor_amd.fakes.prometheus import FakePrometheus
    prom = FakePrometheus().start()
    cfg = demo_config(sources={"history": "prometheus"}, prometheus={"url": prom.url, "timeout": 3.0},
                      periods={"history_step": 1.0, "history_window": "1m"})  # result ttl 0.5 s
    mon = Monitor(cfg)
    mon.tick_all()
    srv = MonitorServer(mon, cfg).start()
    try:
        first = get_json(srv, "/api/history")
        assert first["source"] == "prometheus" and len(first["cpu"]["data"]) > 10
        prom.delay_s = 0.4                                                 # six queries, an 8-way pool: a refresh now takes >= 0.4 s
        time.sleep(0.6)                                                    # the cached result is stale when the burst arrives
        out, lat = [], []

        def hit(path):
            t0 = time.perf_counter()
            out.append((path, get(srv, path)[0]))
            lat.append((path, time.perf_counter() - t0))

        th = [threading.Thread(target=hit, args=("/api/history",)) for _ in range(12)]
        [t.start() for t in th]
        time.sleep(0.05)
        t0 = time.perf_counter()
        st, _, body = get(srv, "/api/cluster")
        dt = time.perf_counter() - t0
        [t.join() for t in th]
        assert st == 200 and dt < 0.1, dt                                  # not queued behind the history refreshes
        assert all(code == 200 for _, code in out)
        fast = [d for _, d in lat if d < 0.2]
        assert len(fast) >= 10, sorted(d for _, d in lat)                  # at most a refresher or two paid for Prometheus
    finally:
        srv.stop()
        mon.stop()
        prom.stop()
