"""Data-parallel serving on one node: one engine process per GPU behind a small router.

The MI355X-first way to scale this workload is replicas, not tensor parallelism: 288 GB per GPU
holds the 8B and the 70B model whole, so no collective sits on the token path and eight GPUs are
eight independent engines. ``ReplicaSet`` starts ``serve-llm`` once per GPU (each child sees one
device through ``HIP_VISIBLE_DEVICES``), ``Router`` spreads ``POST /v1/completions`` over them
(least outstanding requests) and answers ``GET /metrics`` with the replicas' vLLM series, each
carrying a ``replica`` label, so one scrape target shows the whole node. The monitor can equally
be pointed at the replicas themselves (``sources.vllm_endpoints`` takes a list).

Standard library only. No reference counterpart (vLLM is a README bullet, reference README.md:73).
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import threading
import time
import urllib.error
import urllib.request
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Dict, List, Optional, Sequence

_SERIES = re.compile(r"^([a-zA-Z_:][a-zA-Z0-9_:]*)(\{[^}]*\})?(\s+.*)$")


def label_exposition(text: str, replica: str) -> List[str]:
    """Prometheus text -> its sample lines with ``replica="<replica>"`` added (comments dropped)."""
    out = []
    for line in text.splitlines():
        if not line or line[0] == "#":
            continue
        m = _SERIES.match(line)
        if not m:
            continue
        name, labels, rest = m.group(1), m.group(2), m.group(3)
        lab = f'replica="{replica}"'
        labels = "{" + lab + "}" if not labels else labels[:-1] + ("," if len(labels) > 2 else "") + lab + "}"
        out.append(f"{name}{labels}{rest}")
    return out


class Router:
    def __init__(self, backends: Sequence[str], host: str = "127.0.0.1", port: int = 0, timeout: float = 600.0):
        self.backends = [b.rstrip("/") for b in backends]
        self.timeout = timeout
        self.outstanding = [0] * len(self.backends)
        self.sent = [0] * len(self.backends)
        self.failed = [0] * len(self.backends)
        self._lock = threading.Lock()
        outer = self

        class H(BaseHTTPRequestHandler):
            protocol_version = "HTTP/1.1"

            def log_message(self, *a):
                pass

            def _send(self, code: int, body: bytes, ctype: str = "application/json") -> None:
                self.send_response(code)
                self.send_header("Content-Type", ctype)
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

            def do_GET(self):
                path = self.path.split("?")[0]
                if path == "/metrics":
                    self._send(200, outer.metrics().encode(), "text/plain; version=0.0.4; charset=utf-8")
                elif path == "/health":
                    up = outer.healthy()
                    self._send(200 if any(up) else 503, json.dumps({"replicas_up": sum(up), "replicas": len(up)}).encode())
                elif path == "/stats":
                    self._send(200, json.dumps(outer.stats()).encode())
                elif path == "/v1/models":
                    code, body = outer.forward_get(0, "/v1/models")
                    self._send(code, body)
                else:
                    self._send(404, b'{"error": "not found"}')

            def do_POST(self):
                if self.path.split("?")[0] != "/v1/completions":
                    self._send(404, b'{"error": "not found"}')
                    return
                try:
                    n = int(self.headers.get("Content-Length") or 0)
                except ValueError:
                    n = -1
                if n < 0 or n > (8 << 20):                   # the same bound the replicas apply
                    self._send(400, b'{"error": "bad or oversized Content-Length"}')
                    return
                body = self.rfile.read(n) if n else b"{}"
                code, out = outer.dispatch(body)
                self._send(code, out)

        self._srv = ThreadingHTTPServer((host, port), H)
        self._srv.daemon_threads = True
        self.host, self.port = host, self._srv.server_address[1]
        self.url = f"http://{host}:{self.port}"
        self._th: Optional[threading.Thread] = None

    # ---- request path
    def pick(self, exclude: Sequence[int] = ()) -> Optional[int]:
        with self._lock:
            cand = [i for i in range(len(self.backends)) if i not in exclude]
            if not cand:
                return None
            i = min(cand, key=lambda j: (self.outstanding[j], self.sent[j]))
            self.outstanding[i] += 1
            self.sent[i] += 1
            return i

    def dispatch(self, body: bytes):
        """Send to the least loaded replica; a replica that cannot be reached is skipped (the request has
        not started anywhere, so trying the next one is safe)."""
        tried: List[int] = []
        while True:
            i = self.pick(tried)
            if i is None:
                return 503, json.dumps({"error": "no replica reachable", "tried": len(tried)}).encode()
            try:
                req = urllib.request.Request(self.backends[i] + "/v1/completions", data=body,
                                             headers={"Content-Type": "application/json"})
                with urllib.request.urlopen(req, timeout=self.timeout) as r:
                    out = json.loads(r.read())
                if not isinstance(out, dict):
                    raise ValueError("replica answered with something that is not a JSON object")
                out["replica"] = i
                return 200, json.dumps(out).encode()
            except urllib.error.HTTPError as e:
                if e.code != 503:                        # the replica judged the request: its verdict stands
                    return e.code, e.read() or b"{}"
                with self._lock:                         # 503: the replica is stopping or failed the launch: try another
                    self.failed[i] += 1
                tried.append(i)
            except (urllib.error.URLError, OSError, ValueError):     # unreachable, or a body that is not JSON: a replica failure
                with self._lock:
                    self.failed[i] += 1
                tried.append(i)
            finally:
                with self._lock:
                    self.outstanding[i] -= 1

    def forward_get(self, i: int, path: str, timeout: float = 3.0):
        try:
            with urllib.request.urlopen(self.backends[i] + path, timeout=timeout) as r:
                return 200, r.read()
        except urllib.error.HTTPError as e:
            return e.code, e.read()
        except (urllib.error.URLError, OSError) as e:
            return 502, json.dumps({"error": str(e)}).encode()

    # ---- aggregate views
    def healthy(self) -> List[bool]:
        return [self.forward_get(i, "/health", 1.5)[0] == 200 for i in range(len(self.backends))]

    def metrics(self) -> str:
        lines: List[str] = []
        for i in range(len(self.backends)):
            code, body = self.forward_get(i, "/metrics")
            if code == 200:
                lines.extend(label_exposition(body.decode("utf-8", "replace"), str(i)))
        with self._lock:
            for i in range(len(self.backends)):
                lines.append(f'klm_router_requests_total{{replica="{i}"}} {self.sent[i]}')
                lines.append(f'klm_router_unreachable_total{{replica="{i}"}} {self.failed[i]}')
                lines.append(f'klm_router_outstanding{{replica="{i}"}} {self.outstanding[i]}')
        return "\n".join(lines) + "\n"

    def stats(self) -> Dict[str, object]:
        per = []
        for i in range(len(self.backends)):
            code, body = self.forward_get(i, "/stats")
            per.append(json.loads(body) if code == 200 else {"error": code})
        with self._lock:
            return {"backends": self.backends, "sent": list(self.sent), "failed": list(self.failed),
                    "outstanding": list(self.outstanding), "replicas": per}

    # ---- lifecycle
    def start(self) -> "Router":
        self._th = threading.Thread(target=self._srv.serve_forever, name="klm-llm-router", daemon=True)
        self._th.start()
        return self

    def stop(self) -> None:
        self._srv.shutdown()
        self._srv.server_close()
        if self._th:
            self._th.join(timeout=5)

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()


class ReplicaSet:
    """``serve-llm`` once per device, each child pinned to one GPU before anything touches HIP."""

    def __init__(self, devices: Sequence[int], base_port: int, args: Sequence[str], host: str = "127.0.0.1",
                 device_kind: str = "cuda", env: Optional[Dict[str, str]] = None):
        self.devices, self.base_port, self.args, self.host = list(devices), base_port, list(args), host
        self.device_kind = device_kind
        self.env = dict(os.environ if env is None else env)
        self.procs: List[subprocess.Popen] = []

    @property
    def urls(self) -> List[str]:
        return [f"http://{self.host}:{self.base_port + i}" for i in range(len(self.devices))]

    def start(self) -> "ReplicaSet":
        for i, d in enumerate(self.devices):
            env = dict(self.env, PYTHONUNBUFFERED="1", KLM_CHILD_OF=str(os.getpid()))   # the replica dies with this process
            if self.device_kind == "cuda":
                env["HIP_VISIBLE_DEVICES"] = str(d)          # the child sees exactly one GPU, as cuda:0
            cmd = [sys.executable, "-m", "k8s_llm_monitor_amd", "serve-llm", "--host", self.host, "--port",
                   str(self.base_port + i), "--device", "cuda:0" if self.device_kind == "cuda" else "cpu", *self.args]
            self.procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        return self

    def wait_ready(self, timeout: float = 180.0) -> bool:
        t_end = time.time() + timeout
        pending = set(range(len(self.procs)))
        while pending and time.time() < t_end:
            for i in list(pending):
                if self.procs[i].poll() is not None:
                    return False                              # a replica died during start-up
                try:
                    urllib.request.urlopen(self.urls[i] + "/health", timeout=1.0).read()
                    pending.discard(i)
                except (urllib.error.URLError, OSError):
                    pass
            time.sleep(0.3)
        return not pending

    def stop(self) -> None:
        for p in self.procs:
            if p.poll() is None:
                p.terminate()
        for p in self.procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        self.procs = []
