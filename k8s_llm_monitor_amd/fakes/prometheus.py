"""A tiny Prometheus HTTP API stub: ``/api/v1/query`` and ``/api/v1/query_range``.

Serves synthetic series for any PromQL string through a ``resolver(query, t) -> float|None``
so the production client (collectors.prometheus) and history shaping run against real HTTP
on a host without Prometheus. SURVEY.md section 4 "FakePrometheus"; the reference has no fakes.
"""
from __future__ import annotations

import json
import math
import threading
import urllib.parse
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Callable, List, Optional

Resolver = Callable[[str, float], Optional[float]]


def default_resolver(query: str, t: float) -> Optional[float]:
    """Plausible numbers keyed on substrings of the reference's / this repo's queries."""
    w = math.sin(t / 300.0) * 0.5 + 0.5
    q = query
    if "node_cpu_seconds_total" in q:
        return 5.0 + 20.0 * w
    if "node_memory_MemAvailable_bytes" in q:
        return 40.0 + 5.0 * w
    if "node_filesystem_avail_bytes" in q:
        return 45.3
    if "gfx_activity" in q or "DCGM_FI_DEV_GPU_UTIL" in q:
        return 60.0 + 30.0 * w
    if "used_vram" in q or "DCGM_FI_DEV_FB_USED" in q:
        return 170000.0 + 8000.0 * w
    if "junction_temperature" in q or "DCGM_FI_DEV_GPU_TEMP" in q:
        return 66.0 + 9.0 * w
    if "time_to_first_token" in q:
        return 0.8 if "0.99" in q else 0.12
    if "generation_tokens" in q:
        return 5000.0 + 600.0 * w
    if "prompt_tokens" in q:
        return 21000.0
    if "cache_usage_perc" in q:
        return 55.0 + 10.0 * w
    if "num_requests_running" in q:
        return 48.0
    if "num_requests_waiting" in q:
        return 2.0
    if q.strip() == "1":
        return 1.0
    return None


def _step_seconds(s: str) -> float:
    try:
        return float(s)
    except ValueError:
        from ..utils.timefmt import parse_duration
        return float(parse_duration(s, default=30))


class FakePrometheus:
    def __init__(self, resolver: Resolver = default_resolver, host: str = "127.0.0.1", port: int = 0):
        self.resolver = resolver
        self.requests: List[str] = []
        self.delay_s = 0.0
        self.garbage = False          # answer with a non-JSON body
        outer = self

        class H(BaseHTTPRequestHandler):
            def log_message(self, *a):   # quiet
                pass

            def do_GET(self):
                import time as _t
                u = urllib.parse.urlsplit(self.path)
                qs = urllib.parse.parse_qs(u.query)
                outer.requests.append(self.path)
                if outer.delay_s:
                    _t.sleep(outer.delay_s)
                if outer.garbage:
                    body = b"<html>bad gateway</html>"
                    self.send_response(502)
                elif u.path == "/api/v1/query":
                    q = qs.get("query", [""])[0]
                    t = float(qs.get("time", [_t.time()])[0])
                    v = outer.resolver(q, t)
                    res = [] if v is None else [{"metric": {}, "value": [t, repr(float(v))]}]
                    body = json.dumps({"status": "success", "data": {"resultType": "vector", "result": res}}).encode()
                    self.send_response(200)
                elif u.path == "/api/v1/query_range":
                    q = qs.get("query", [""])[0]
                    try:
                        start, end = float(qs["start"][0]), float(qs["end"][0])
                        step = _step_seconds(qs["step"][0])
                        vals = []
                        t = start
                        while t <= end + 1e-9:
                            v = outer.resolver(q, t)
                            if v is not None:
                                vals.append([t, repr(float(v))])
                            t += step
                        res = [{"metric": {}, "values": vals}] if vals else []
                        body = json.dumps({"status": "success", "data": {"resultType": "matrix", "result": res}}).encode()
                        self.send_response(200)
                    except (KeyError, ValueError) as e:
                        body = json.dumps({"status": "error", "errorType": "bad_data", "error": str(e)}).encode()
                        self.send_response(400)
                else:
                    body = b"404 page not found"
                    self.send_response(404)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

        self._srv = ThreadingHTTPServer((host, port), H)
        self._srv.daemon_threads = True
        self.url = "http://%s:%d" % (host, self._srv.server_address[1])
        self._th: Optional[threading.Thread] = None

    def start(self) -> "FakePrometheus":
        self._th = threading.Thread(target=self._srv.serve_forever, name="fake-prometheus", daemon=True)
        self._th.start()
        return self

    def stop(self) -> None:
        self._srv.shutdown()
        self._srv.server_close()
        if self._th:
            self._th.join(timeout=2)

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()
