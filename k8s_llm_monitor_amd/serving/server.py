"""HTTP front of the serving engine: what a vLLM pod shows to the monitor and to clients.

    GET  /metrics            Prometheus text with vLLM's metric names (scraped by collectors/vllm.py)
    GET  /health             200 once the engine thread runs
    GET  /v1/models          the served model
    POST /v1/completions     {"prompt": [token ids] | "prompt_len": n, "max_tokens": m} -> token ids
    GET  /stats[?window=S]   exact ground truth (quantiles, rates) + engine description
    GET  /trace              the recent model launches as a Chrome trace (open in Perfetto)

There is no tokenizer (no network to fetch one): prompts are token ids, or a length from which
random ids are drawn. Standard library only.
"""
from __future__ import annotations

import json
import random
import threading
import urllib.parse
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Optional

from .engine import Engine


class LlmServer:
    def __init__(self, engine: Engine, host: str = "127.0.0.1", port: int = 0, request_timeout: float = 600.0):
        self.engine = engine
        self.request_timeout = request_timeout
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

            def _json(self, code: int, obj) -> None:
                self._send(code, json.dumps(obj).encode())

            def do_GET(self):
                u = urllib.parse.urlsplit(self.path)
                if u.path == "/metrics":
                    self._send(200, outer.engine.metrics.render().encode(), "text/plain; version=0.0.4; charset=utf-8")
                elif u.path == "/health":
                    self._send(200, b"", "text/plain")
                elif u.path == "/v1/models":
                    self._json(200, {"object": "list", "data": [{"id": outer.engine.mcfg.name, "object": "model",
                                                                 "max_model_len": outer.engine.max_len}]})
                elif u.path == "/stats":
                    q = urllib.parse.parse_qs(u.query)
                    try:
                        w: Optional[float] = float(q["window"][0]) if "window" in q else None
                    except ValueError:
                        w = None
                    self._json(200, {"truth": outer.engine.metrics.truth(w), "engine": outer.engine.describe()})
                elif u.path == "/trace":
                    self._json(200, outer.engine.trace())
                else:
                    self._json(404, {"error": "not found"})

            def do_POST(self):
                if urllib.parse.urlsplit(self.path).path != "/v1/completions":
                    self._json(404, {"error": "not found"})
                    return
                try:
                    n = int(self.headers.get("Content-Length") or 0)
                    if n > (8 << 20):
                        raise ValueError("body too large")
                    if n < 0:
                        raise ValueError("bad Content-Length")
                    req = json.loads(self.rfile.read(n) or b"{}")
                    if not isinstance(req, dict):
                        raise ValueError("the body must be a JSON object")
                    prompt = req.get("prompt")
                    if isinstance(prompt, list) and prompt and all(isinstance(t, int) for t in prompt):
                        ids = prompt
                    else:
                        plen = int(req.get("prompt_len") or 0)
                        if plen < 1:
                            raise ValueError("give `prompt` (token ids) or `prompt_len`")
                        if plen + 1 > outer.engine.max_len:      # before any ids are made: a huge number must cost nothing
                            raise ValueError(f"prompt_len {plen} does not fit max_model_len {outer.engine.max_len}")
                        rng = random.Random(req.get("seed"))
                        ids = [rng.randrange(outer.engine.mcfg.vocab) for _ in range(plen)]
                    if len(ids) + 1 > outer.engine.max_len:
                        raise ValueError(f"prompt of {len(ids)} tokens does not fit max_model_len {outer.engine.max_len}")
                    max_tokens = int(req.get("max_tokens") or 16)
                    if not 1 <= max_tokens <= outer.engine.max_len:
                        raise ValueError(f"max_tokens must be within 1..{outer.engine.max_len}")
                except (ValueError, TypeError, json.JSONDecodeError) as e:
                    self._json(400, {"error": str(e)})
                    return
                r = outer.engine.submit(ids, max_tokens)
                if not r.done.wait(outer.request_timeout):
                    self._json(504, {"error": "timed out", "id": r.rid})
                    return
                if r.error:                      # a bad prompt is the caller's fault, anything else the server's
                    self._json(400 if r.error.startswith("prompt") else 503, {"error": r.error, "id": r.rid})
                    return
                self._json(200, {"id": f"cmpl-{r.rid}", "object": "text_completion", "model": outer.engine.mcfg.name,
                                 "choices": [{"index": 0, "token_ids": r.out, "finish_reason": "length"}],
                                 "usage": {"prompt_tokens": len(r.prompt), "completion_tokens": len(r.out),
                                           "total_tokens": r.n_tokens},
                                 "timing": {"ttft_s": r.ttft, "e2e_s": (r.t_done or 0) - r.arrival,
                                            "preempted": r.preempted}})

        self._srv = ThreadingHTTPServer((host, port), H)
        self._srv.daemon_threads = True
        self.host, self.port = host, self._srv.server_address[1]
        self.url = f"http://{host}:{self.port}"
        self._th: Optional[threading.Thread] = None

    def start(self) -> "LlmServer":
        self._th = threading.Thread(target=self._srv.serve_forever, name="klm-llm-http", daemon=True)
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
