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
``/healthz``, and ``/api/config`` (only with ``server.expose_config``; credentials redacted). Handlers read the sampler's snapshot and never run a collector.
"""
from __future__ import annotations

import json
import os
import threading
import time
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

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
    """Per-GPU, per-link table for the xGMI panel, plus the GPU x GPU traffic matrix the topology calls make
    possible: ``matrix[i][j]`` = {tx, rx, state} of the link from GPU i to GPU j (None where amdsmi names no
    link between them). A down link or a link carrying far less than its siblings shows up as one cell."""
    rows = []
    index_of = {g.get("index"): n for n, g in enumerate(gpus)}
    matrix = [[None] * len(gpus) for _ in gpus]
    notes = []
    for n, g in enumerate(gpus):
        x = g.get("xgmi") or {}
        rows.append({"gpu": g.get("index"), "bdf": g.get("bdf"), "rxGBps": x.get("rxGBps"), "txGBps": x.get("txGBps"),
                     "linkSpeedGbps": x.get("linkSpeedGbps"), "linkWidth": x.get("linkWidth"),
                     "hiveId": x.get("hiveId"), "nodeId": x.get("nodeId"), "peersKnown": x.get("peersKnown"),
                     "errorStatus": x.get("errorStatus"), "topologyNote": x.get("topologyNote"),
                     "links": x.get("links") or []})
        if x.get("topologyNote"):
            notes.append(f"GPU{g.get('index')}: {x['topologyNote']}")
        for l in x.get("links") or []:
            j = index_of.get(l.get("peerIndex"))
            if j is not None and j != n:
                matrix[n][j] = {"link": l.get("link"), "state": l.get("state"), "txGBps": l.get("txGBps"), "rxGBps": l.get("rxGBps"),
                                "source": l.get("peerSource")}
    total = sum(((r["rxGBps"] or 0.0) + (r["txGBps"] or 0.0)) for r in rows)
    # imbalance over the named, up links that carry traffic: busiest / mean (1.0 = even; a ring reads >> 1)
    tx = [c["txGBps"] for row in matrix for c in row if c and c.get("state") == "up" and c.get("txGBps") is not None]
    mean = sum(tx) / len(tx) if tx else 0.0
    return {"gpus": rows, "totalGBps": total, "linkPeakGBps": rccl_probe.XGMI_LINK_GBPS,
            "indices": [g.get("index") for g in gpus], "matrix": matrix,
            "peersKnown": any(r.get("peersKnown") for r in rows), "notes": notes,
            "linkImbalance": (max(tx) / mean) if mean > 1e-9 else None}


# Paths whose body depends only on the sampler's snapshot: serialised once per snapshot
# version, then served as cached bytes (the reference re-runs its collectors per request).
_SNAPSHOT_PATHS = {"/api/host/metrics", "/api/gpu/metrics", "/api/k8s/pods", "/api/alerts", "/api/vllm/metrics",
                   "/api/xgmi", "/api/alerts/history", "/metrics"}
_REASONS = {200: b"OK", 404: b"Not Found", 405: b"Method Not Allowed", 400: b"Bad Request", 500: b"Internal Server Error",
            503: b"Service Unavailable", 431: b"Request Header Fields Too Large"}
_PUSHED = ("alerts", "realtime", "pods", "vllm")      # sampler events forwarded to /api/events
_BLOCKING_PATHS = {"/api/history", "/api/cluster", "/api/rccl/probe", "/api/validation"}     # may do I/O: run off the loop


class MonitorServer:
    """asyncio HTTP/1.1 server (keep-alive, GET / HEAD / OPTIONS) on its own thread.

    One event loop serves every connection; handlers only copy cached bytes, so a request
    costs microseconds and a slow client cannot hold a thread. The few routes that may touch
    the network or the disk (history through Prometheus, the cluster view, result files) run in
    the loop's executor. ``/api/events`` is a server-sent-events stream fed from the sampler."""

    MAX_HEADER = 16384

    def __init__(self, monitor: Monitor, cfg: Optional[Config] = None):
        import asyncio
        import socket
        self.monitor = monitor
        self.cfg = cfg or monitor.cfg
        static = Path(self.cfg.server.static_dir) if self.cfg.server.static_dir else WEB_DIR
        self.static_dir = static
        # like the reference, the page is read once at start-up and a missing file is fatal (:241)
        self.html = (static / "monitor.html").read_bytes()
        # the page's alert poll period is configuration (reference: 10 s, monitor.html:608; default here 4 s)
        self.html = self.html.replace(b"/*ALERT_POLL_MS*/4000", b"%d" % int(self.cfg.periods.page_alert_poll * 1000))
        self._cache: Dict[str, Tuple[int, Tuple[int, str, bytes]]] = {}
        self._head_cors = ("Access-Control-Allow-Origin: %s\r\nAccess-Control-Allow-Methods: GET, OPTIONS\r\n"
                           % self.cfg.server.cors_origin).encode("latin-1")
        self._sse: "set" = set()
        self._closing = False
        self.requests = 0
        self._loop = asyncio.new_event_loop()
        self._sock = socket.socket(socket.AF_INET6 if ":" in self.cfg.server.host else socket.AF_INET, socket.SOCK_STREAM)
        self._sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self._sock.bind((self.cfg.server.host, self.cfg.server.port))
        self._sock.listen(512)
        self._sock.setblocking(False)
        self.port = self._sock.getsockname()[1]
        self._server = None
        self._th: Optional[threading.Thread] = None
        self._ready = threading.Event()
        monitor.listeners.append(self._on_event)

    # ---- server-sent events ------------------------------------------------------------
    def _on_event(self, kind: str, payload: Any) -> None:
        if kind not in _PUSHED or not self._sse or self._closing:      # nobody listening: nothing is serialised
            return
        doc = self.monitor.alerts_public() if kind == "alerts" else payload
        msg = b"event: " + kind.encode("ascii") + b"\ndata: " + _dumps(doc) + b"\n\n"
        try:
            self._loop.call_soon_threadsafe(self._broadcast, msg)
        except RuntimeError:
            pass

    def _broadcast(self, msg: bytes) -> None:
        for w in list(self._sse):
            try:
                if w.transport.get_write_buffer_size() < 1 << 20:     # a stalled client is skipped, not buffered
                    w.write(msg)
            except Exception:
                self._sse.discard(w)

    # ---- routing -------------------------------------------------------------------------
    def route(self, path: str) -> Optional[Tuple[int, str, bytes]]:
        """(status, content type, body) for a GET, None for an unknown path."""
        if path in _SNAPSHOT_PATHS and not (path == "/api/alerts" and self.cfg.compat.alerts_per_request):
            ver = self.monitor.version
            hit = self._cache.get(path)
            if hit is not None and hit[0] == ver:
                return hit[1]
            r = self._route(path)
            if r is not None:
                self._cache[path] = (ver, r)
            return r
        return self._route(path)

    def _route(self, path: str) -> Optional[Tuple[int, str, bytes]]:
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
        if path == "/api/config" and self.cfg.server.expose_config:
            # off by default: CORS is "*" like the reference's, so any page the operator's browser visits could read it
            return 200, JSON_CT, _dumps(redacted_config(self.cfg))
        if path == "/metrics":
            return 200, "text/plain; version=0.0.4; charset=utf-8", exporter.render(m.snapshot(), m.stats()).encode("utf-8")
        if path == "/healthz":
            age = time.time() - (m.get("t_host") or 0.0)
            ok = age < 3 * self.cfg.periods.realtime + 5
            return (200 if ok else 503), JSON_CT, _dumps({"ok": ok, "last_sample_age_s": round(age, 3)})
        return None

    # ---- wire ----------------------------------------------------------------------------
    def _response(self, status: int, ctype: str, body: bytes, head_only: bool, keep: bool) -> bytes:
        h = (b"HTTP/1.1 %d %s\r\n" % (status, _REASONS.get(status, b"OK")) + self._head_cors +
             b"Content-Type: " + ctype.encode("latin-1") + b"\r\nContent-Length: %d\r\nCache-Control: no-store\r\n" % len(body) +
             (b"Connection: keep-alive\r\n\r\n" if keep else b"Connection: close\r\n\r\n"))
        return h if head_only else h + body

    async def _handle(self, reader, writer) -> None:
        import asyncio
        loop = self._loop
        try:
            while not self._closing:
                try:
                    head = await asyncio.wait_for(reader.readuntil(b"\r\n\r\n"), timeout=30.0)
                except asyncio.LimitOverrunError:
                    writer.write(self._response(431, JSON_CT, b"", False, False))
                    break
                except (asyncio.IncompleteReadError, asyncio.TimeoutError, ConnectionError):
                    break
                self.requests += 1
                line, _, rest = head.partition(b"\r\n")
                parts = line.split()
                if len(parts) < 2:
                    writer.write(self._response(400, JSON_CT, _dumps({"error": "bad request line"}), False, False))
                    break
                method, target = parts[0].upper(), parts[1].decode("latin-1")
                low = rest.lower()
                keep = (len(parts) > 2 and parts[2] == b"HTTP/1.1" and b"connection: close" not in low) or b"connection: keep-alive" in low
                clen = 0
                i = low.find(b"content-length:")
                if i >= 0:
                    try:
                        clen = int(low[i + 15: low.find(b"\r\n", i)].strip() or 0)
                    except ValueError:
                        clen = 0
                if clen:                                   # a body we never use: drain it so keep-alive stays in sync
                    if clen > 1 << 20:
                        keep = False
                    else:
                        await reader.readexactly(clen)
                path = target.split("?", 1)[0]             # the query string is ignored (monitor_server.js:250)
                if self.cfg.server.access_log:
                    print("[http] %s %s" % (method.decode("latin-1"), target), flush=True)
                if method == b"OPTIONS":
                    writer.write(self._response(200, JSON_CT, b"", False, keep))
                elif method not in (b"GET", b"HEAD"):
                    writer.write(self._response(405, JSON_CT, _dumps({"error": "GET and OPTIONS only"}), False, keep))
                elif path == "/api/events" and method == b"GET":
                    await self._events(writer)
                    return
                else:
                    try:
                        if path in _BLOCKING_PATHS:
                            r = await loop.run_in_executor(None, self.route, path)
                        else:
                            r = self.route(path)
                        if r is None:
                            # the reference keeps the JSON content type on its plain-text 404 (:290)
                            r = (404, JSON_CT, b"Not Found")
                    except Exception as e:
                        r = (500, JSON_CT, _dumps({"error": str(e)}))
                    writer.write(self._response(r[0], r[1], r[2], method == b"HEAD", keep))
                await writer.drain()
                if not keep:
                    break
        except (ConnectionError, OSError):
            pass
        finally:
            try:
                writer.close()
            except Exception:
                pass

    async def _events(self, writer) -> None:
        import asyncio
        writer.write(b"HTTP/1.1 200 OK\r\n" + self._head_cors + b"Content-Type: text/event-stream; charset=utf-8\r\n"
                     b"Cache-Control: no-store\r\nConnection: close\r\n\r\n")
        m = self.monitor                                   # first paint: the current state of every pushed panel
        for kind, doc in (("alerts", m.alerts_public()), ("realtime", {"host": m.get("host"), "gpus": m.get("gpus")}),
                          ("pods", m.get("pods")), ("vllm", m.get("vllm"))):
            writer.write(b"event: " + kind.encode("ascii") + b"\ndata: " + _dumps(doc) + b"\n\n")
        self._sse.add(writer)
        try:
            while not self._closing and not writer.is_closing():
                await writer.drain()
                await asyncio.sleep(10.0)
                writer.write(b": keep-alive\n\n")
        except (ConnectionError, OSError, asyncio.CancelledError):
            pass
        finally:
            self._sse.discard(writer)
            try:
                writer.close()
            except Exception:
                pass

    # ---- lifecycle -------------------------------------------------------------------------
    def _run(self) -> None:
        import asyncio
        asyncio.set_event_loop(self._loop)

        async def boot():
            self._server = await asyncio.start_server(self._handle, sock=self._sock, limit=self.MAX_HEADER)
            self._ready.set()

        self._loop.run_until_complete(boot())
        try:
            self._loop.run_forever()
        finally:
            try:
                pending = [t for t in asyncio.all_tasks(self._loop) if not t.done()]
                for t in pending:
                    t.cancel()
                if pending:
                    self._loop.run_until_complete(asyncio.gather(*pending, return_exceptions=True))
                self._loop.run_until_complete(self._loop.shutdown_asyncgens())
            finally:
                self._loop.close()

    def start(self) -> "MonitorServer":
        self._th = threading.Thread(target=self._run, name="klm-http", daemon=True)
        self._th.start()
        self._ready.wait(timeout=5)
        return self

    def serve_forever(self) -> None:
        self.start()
        while self._th is not None and self._th.is_alive():
            self._th.join(timeout=0.5)

    def stop(self) -> None:
        if self._closing:
            return
        self._closing = True

        def shutdown():
            if self._server is not None:
                self._server.close()
            for w in list(self._sse):
                try:
                    w.close()
                except Exception:
                    pass
            self._loop.stop()

        try:
            self._loop.call_soon_threadsafe(shutdown)
        except RuntimeError:
            pass
        if self._th:
            self._th.join(timeout=3)
        try:
            self._sock.close()
        except OSError:
            pass

    @property
    def url(self) -> str:
        host = self.cfg.server.host
        if host in ("0.0.0.0", "::", ""):
            host = "127.0.0.1"
        return f"http://{host}:{self.port}"


def redacted_config(cfg: Config) -> Dict[str, Any]:
    """The running configuration without anything that can carry a credential: the webhook URL is cut to its
    origin and the service-account token path is dropped."""
    d = cfg.to_dict()
    d.get("sources", {}).pop("kube_token_file", None)
    url = d.get("notify", {}).get("webhook_url") or ""
    if url:
        from urllib.parse import urlsplit
        u = urlsplit(url)
        d["notify"]["webhook_url"] = f"{u.scheme}://{u.hostname or ''}{':%d' % u.port if u.port else ''}/..."
    return d


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
