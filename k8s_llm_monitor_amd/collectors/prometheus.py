"""Prometheus HTTP API client (instant and range queries).

Reference: ``httpGet`` (monitor_server.js:14-25), ``queryPrometheus`` (:27-36, dead code
there; used here by the vLLM scraper), ``queryRangePrometheus`` (:38-52), ``parseDuration``
(:54-63). Same contracts: an unparsable body is ``None``, an instant query returns
``float(result[0].value[1])`` or ``None``, a range query returns
``[{"time": [ms...], "value": [float...]}]`` per series and ``[]`` on any failure.

Fixed on the way: a timeout on every request, https allowed.
"""
from __future__ import annotations

import json
import math
import time
import urllib.error
import urllib.parse
import urllib.request
from typing import Any, Callable, Dict, List, Optional

from ..utils.timefmt import parse_duration


def _float(s: Any) -> float:
    """JavaScript parseFloat on Prometheus sample values ("NaN", "+Inf" included)."""
    try:
        return float(s)
    except (TypeError, ValueError):
        return float("nan")


class PrometheusClient:
    def __init__(self, url: str = "http://localhost:9090", timeout: float = 3.0,
                 now: Callable[[], float] = time.time, log: Optional[Callable[[str], None]] = None):
        self.url = url.rstrip("/")
        self.timeout = timeout
        self._now = now
        self._log = log or (lambda s: None)
        self.requests = 0
        self.failures = 0
        self.last_error: Optional[str] = None

    # monitor_server.js:14-25
    def http_get(self, url: str) -> Any:
        """Parsed JSON body, None when the body is not JSON. Raises on socket errors."""
        self.requests += 1
        try:
            with urllib.request.urlopen(url, timeout=self.timeout) as resp:
                body = resp.read()
        except urllib.error.HTTPError as e:
            body = e.read()        # Prometheus answers 4xx with a JSON error document
        try:
            return json.loads(body.decode("utf-8"))
        except (ValueError, UnicodeDecodeError):
            return None

    # monitor_server.js:27-36
    def query(self, q: str) -> Optional[float]:
        try:
            data = self.http_get(f"{self.url}/api/v1/query?query={urllib.parse.quote(q, safe='')}")
            if data and data.get("status") == "success" and len(data["data"]["result"]) > 0:
                return _float(data["data"]["result"][0]["value"][1])
        except Exception as e:
            self.failures += 1
            self.last_error = f"{type(e).__name__}: {e}"
            self._log(f"Prom query error: {q} {e}")
        return None

    def query_vector(self, q: str) -> List[Dict[str, Any]]:
        """Every series of an instant vector as ``{"labels": {...}, "value": float}``."""
        try:
            data = self.http_get(f"{self.url}/api/v1/query?query={urllib.parse.quote(q, safe='')}")
            if data and data.get("status") == "success":
                return [{"labels": r.get("metric") or {}, "value": _float(r["value"][1])}
                        for r in data["data"]["result"]]
        except Exception as e:
            self.failures += 1
            self.last_error = f"{type(e).__name__}: {e}"
            self._log(f"Prom query error: {q} {e}")
        return []

    # monitor_server.js:38-52
    def query_range(self, q: str, duration: str = "30m", step: str = "30s") -> List[Dict[str, List[float]]]:
        try:
            end = math.floor(self._now())
            start = end - parse_duration(duration)
            url = (f"{self.url}/api/v1/query_range?query={urllib.parse.quote(q, safe='')}"
                   f"&start={start}&end={end}&step={step}")
            data = self.http_get(url)
            if data and data.get("status") == "success":
                return [{"time": [v[0] * 1000 for v in r["values"]],
                         "value": [_float(v[1]) for v in r["values"]]}
                        for r in data["data"]["result"]]
        except Exception as e:
            self.failures += 1
            self.last_error = f"{type(e).__name__}: {e}"
            self._log(f"Prom range error: {q} {e}")
        return []

    def reachable(self) -> bool:
        try:
            data = self.http_get(f"{self.url}/api/v1/query?query=1")
            return bool(data and data.get("status") == "success")
        except Exception:
            return False
