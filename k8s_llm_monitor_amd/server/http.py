"""HTTP API and static dashboard.

Reference: the ``http.createServer`` handler (monitor_server.js:243-299). Kept:
  * every response carries ``Access-Control-Allow-Origin: *`` and
    ``Access-Control-Allow-Methods: GET, OPTIONS`` (:244-245); API bodies are
    ``application/json; charset=utf-8`` (:246); ``OPTIONS`` answers 200 with an empty body
    (:248); the query string is ignored for routing (:250);
  * routes ``/`` and ``/monitor.html`` (the page, read once at start-up, :241/:252-256),
    ``/logo.svg`` (:259-263), ``/api/host/metrics``, ``/api/gpu/metrics``, ``/api/k8s/pods``,
    ``/api/history``, ``/api/alerts`` (:273-288) with the shapes of SURVEY.md 2.5;
  * unknown paths: 404 with the body ``Not Found`` (:290); a handler exception: 500
    ``{"error": msg}`` (:292-294); listens on 0.0.0.0:8888 by default (:297).

Dropped: the ``/danyichun*`` route, which joins the raw URL onto the directory outside the
``try`` block (path traversal plus a crash on a missing file, :266-270).

Added (additive only): ``/api/vllm/metrics``, ``/api/xgmi``, ``/api/rccl/probe``,
``/api/alerts/history``, ``/api/cluster`` (this node plus its configured peers), ``/api/summary`` (everything the page needs in one request),
``/api/events`` (server-sent events, so an alert reaches an open page without waiting for
its next poll), ``/api/monitor/stats``, ``/api/validation``, ``/metrics`` (Prometheus text),
``/healthz``. Handlers read the sampler's snapshot and never run a collector.
"""
from __future__ import annotations

import json
import os
import queue
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from pathlib import Path
from typing import Any, Callable, Dict, Optional, Tuple

from ..config import Config
from ..parallel import probe as rccl_probe
from . import exporter
from .sampler import Monitor

WEB_DIR = Path(__file__).resolve().parent.parent / "web"
JSON_CT = "application/json; charset=utf-8"


def _dumps(obj: Any) -> bytes:
    # JSON.stringify turns NaN/Infinity into null; allow_nan=False would raise instead
    return json.dumps(_clean(obj), ensure_ascii=False, separators=(",", ":")).encode("utf-8")


def _clean(o: Any) -> Any:
    if isinstance(o, float):
        return None if (o != o or o in (float("inf"), float("-inf"))) else o
    if isinstance(o, dict):
        return {k: _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    return o


def xgmi_matrix(gpus) -> Dict[str, Any]:
    """Per-GPU, per-link table for the xGMI panel."""
    rows = []
    for g in gpus:
        x = g.get("xgmi") or {}
        rows.append({"gpu": g.get("index"), "bdf": g.get("bdf"), "rxGBps": x.get("rxGBps"), "txGBps": x.get("txGBps"),
                     "linkSpeedGbps": x.get("linkSpeedGbps"), "linkWidth": x.get("linkWidth"),
                     "links": x.get("links") or []})
    total = sum(((r["rxGBps"] or 0.0) + (r["txGBps"] or 0.0)) for r in rows)
    return {"gpus": rows, "totalGBps": total, "linkPeakGBps": rccl_probe.XGMI_LINK_GBPS}


class MonitorServer:
    def __init__(self, monitor: Monitor, cfg: Optional[Config] = None):
        self.monitor = monitor
        self.cfg = cfg or monitor.cfg
        static = Path(self.cfg.server.static_dir) if self.cfg.server.static_dir else WEB_DIR
        self.static_dir = static
        # like the reference, the page is read once at start-up and a missing file is fatal (:241)
        self.html = (static / "monitor.html").read_bytes()
        self._sse: "set[queue.Queue]" = set()
        self._sse_lock = threading.Lock()
        monitor.listeners.append(self._on_event)
        self.httpd = ThreadingHTTPServer((self.cfg.server.host, self.cfg.server.port), self._handler_class())
        self.httpd.daemon_threads = True
        self.port = self.httpd.server_address[1]
        self._th: Optional[threading.Thread] = None
        self.requests = 0

    # ---- server-sent events ------------------------------------------------------------
    def _on_event(self, kind: str, payload: Any) -> None:
        if kind != "alerts":
            return
        msg = b"event: alerts\ndata: " + _dumps(self.monitor.alerts_public()) + b"\n\n"
        with self._sse_lock:
            for q in list(self._sse):
                try:
                    q.put_nowait(msg)
                except queue.Full:
                    pass

    # ---- routing -------------------------------------------------------------------------
    def route(self, path: str) -> Optional[Tuple[int, str, bytes]]:
        """(status, content type, body) for a GET, None for an unknown path."""
        m = self.monitor
        if path in ("/", "/monitor.html"):
            return 200, "text/html; charset=utf-8", self.html
        if path == "/logo.svg":
            return 200, "image/svg+xml", (self.static_dir / "logo.svg").read_bytes()
        if path == "/api/host/metrics":
            return 200, JSON_CT, _dumps(m.get("host"))
        if path == "/api/gpu/metrics":
            return 200, JSON_CT, _dumps(m.get("gpus"))
        if path == "/api/k8s/pods":
            return 200, JSON_CT, _dumps(m.get("pods"))
        if path == "/api/history":
            return 200, JSON_CT, _dumps(m.history())
        if path == "/api/alerts":
            return 200, JSON_CT, _dumps(m.alerts_public())
        if path == "/api/alerts/history":
            return 200, JSON_CT, _dumps(m.book.history[-200:])
        if path == "/api/vllm/metrics":
            return 200, JSON_CT, _dumps(m.get("vllm"))
        if path == "/api/xgmi":
            return 200, JSON_CT, _dumps(xgmi_matrix(m.get("gpus")))
        if path == "/api/rccl/probe":
            doc = rccl_probe.load_results(self.cfg.sources.probe_results)
            return 200, JSON_CT, _dumps(doc or {"curves": [], "scaling": [], "note": "no probe result yet"})
        if path == "/api/cluster":
            return 200, JSON_CT, _dumps(m.cluster())
        if path == "/api/validation":
            return 200, JSON_CT, _dumps(load_validation_report())
        if path == "/api/summary":
            s = m.snapshot()
            return 200, JSON_CT, _dumps({"host": s["host"], "gpus": s["gpus"], "pods": s["pods"], "vllm": s["vllm"],
                                         "alerts": m.alerts_public(), "time": time.time(),
                                         "ages": {k[2:]: round(time.time() - s[k], 3) for k in s if k.startswith("t_") and s[k]}})
        if path == "/api/monitor/stats":
            st = m.stats()
            st["http_requests"] = self.requests
            return 200, JSON_CT, _dumps(st)
        if path == "/api/config":
            return 200, JSON_CT, _dumps(self.cfg.to_dict())
        if path == "/metrics":
            return 200, "text/plain; version=0.0.4; charset=utf-8", exporter.render(m.snapshot(), m.stats()).encode("utf-8")
        if path == "/healthz":
            age = time.time() - (m.get("t_host") or 0.0)
            ok = age < 3 * self.cfg.periods.realtime + 5
            return (200 if ok else 503), JSON_CT, _dumps({"ok": ok, "last_sample_age_s": round(age, 3)})
        return None

    def _handler_class(self):
        outer = self
        cfg = self.cfg

        class Handler(BaseHTTPRequestHandler):
            protocol_version = "HTTP/1.1"
            server_version = "klm-monitor"

            def log_message(self, fmt, *args):
                if cfg.server.access_log:
                    BaseHTTPRequestHandler.log_message(self, fmt, *args)

            def _send(self, status: int, ctype: str, body: bytes) -> None:
                self.send_response(status)
                self.send_header("Access-Control-Allow-Origin", cfg.server.cors_origin)
                self.send_header("Access-Control-Allow-Methods", "GET, OPTIONS")
                self.send_header("Content-Type", ctype)
                self.send_header("Content-Length", str(len(body)))
                self.send_header("Cache-Control", "no-store")
                self.end_headers()
                if self.command != "HEAD":
                    self.wfile.write(body)

            def do_OPTIONS(self):
                outer.requests += 1
                self._send(200, JSON_CT, b"")

            def do_HEAD(self):
                self.do_GET()

            def do_GET(self):
                outer.requests += 1
                path = self.path.split("?", 1)[0]
                if path == "/api/events":
                    return self._events()
                try:
                    r = outer.route(path)
                    if r is None:
                        # the reference keeps the JSON content type on its plain-text 404 (:290)
                        return self._send(404, JSON_CT, b"Not Found")
                    self._send(*r)
                except (BrokenPipeError, ConnectionResetError):
                    pass
                except Exception as e:
                    self._send(500, JSON_CT, _dumps({"error": str(e)}))

            def do_POST(self):
                outer.requests += 1
                self._send(405, JSON_CT, _dumps({"error": "GET and OPTIONS only"}))

            do_PUT = do_DELETE = do_PATCH = do_POST

            def _events(self):
                q: "queue.Queue[bytes]" = queue.Queue(maxsize=64)
                with outer._sse_lock:
                    outer._sse.add(q)
                try:
                    self.send_response(200)
                    self.send_header("Access-Control-Allow-Origin", cfg.server.cors_origin)
                    self.send_header("Content-Type", "text/event-stream; charset=utf-8")
                    self.send_header("Cache-Control", "no-store")
                    self.send_header("Connection", "close")
                    self.end_headers()
                    self.close_connection = True
                    self.wfile.write(b"event: alerts\ndata: " + _dumps(outer.monitor.alerts_public()) + b"\n\n")
                    self.wfile.flush()
                    while not outer._closing:
                        try:
                            msg = q.get(timeout=10.0)
                        except queue.Empty:
                            msg = b": keep-alive\n\n"
                        self.wfile.write(msg)
                        self.wfile.flush()
                except (BrokenPipeError, ConnectionResetError, OSError):
                    pass
                finally:
                    with outer._sse_lock:
                        outer._sse.discard(q)

        return Handler

    _closing = False

    # ---- lifecycle -------------------------------------------------------------------------
    def start(self) -> "MonitorServer":
        self._th = threading.Thread(target=self.httpd.serve_forever, name="klm-http", daemon=True)
        self._th.start()
        return self

    def serve_forever(self) -> None:
        self.httpd.serve_forever()

    def stop(self) -> None:
        self._closing = True
        self.httpd.shutdown()
        self.httpd.server_close()
        if self._th:
            self._th.join(timeout=2)

    @property
    def url(self) -> str:
        host = self.cfg.server.host
        if host in ("0.0.0.0", "::", ""):
            host = "127.0.0.1"
        return f"http://{host}:{self.port}"


def validation_report_path() -> str:
    d = os.environ.get("KLM_STATE_DIR") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "klm")
    return os.path.join(d, "validation.json")


def load_validation_report() -> Dict[str, Any]:
    try:
        with open(validation_report_path(), "r", encoding="utf-8") as f:
            d = json.load(f)
        return d if isinstance(d, dict) else {}
    except (OSError, ValueError):
        return {}
