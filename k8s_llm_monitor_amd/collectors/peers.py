"""Cluster view: one monitor per GPU node (DaemonSet), any of them can aggregate the others.

No reference counterpart: the reference watches exactly one host (monitor_server.js:66-95).
An MI355X inference cluster is several 8-GPU nodes, so a monitor can be given the URLs of
its peers (``sources.peers``); it polls their ``/api/summary`` off the request path and serves
``GET /api/cluster``: per node the host figures, per-GPU HBM / activity / temperature
digest and alert counts, plus cluster totals. Peers are fetched concurrently with a timeout;
a dead peer is reported as ``ok: false`` with its error, never an exception.
"""
from __future__ import annotations

import json
import socket
import time
import urllib.request
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, Dict, Mapping, Optional, Sequence


def digest(summary: Mapping[str, Any]) -> Dict[str, Any]:
    """The part of a node's ``/api/summary`` the cluster table needs."""
    gpus = summary.get("gpus") or []
    alerts = summary.get("alerts") or {}
    host = summary.get("host") or {}

    def pf(x):
        try:
            return float(x)
        except (TypeError, ValueError):
            return None

    used = sum(g.get("memoryUsed") or 0 for g in gpus)
    total = sum(g.get("memoryTotal") or 0 for g in gpus)
    utils = [g.get("gfxActivityAvg") if g.get("gfxActivityAvg") is not None else g.get("utilization") for g in gpus]
    utils = [u for u in utils if u is not None]
    temps = [g.get("temperature") for g in gpus if g.get("temperature") is not None]
    power = [g.get("powerW") for g in gpus if g.get("powerW") is not None]
    v = summary.get("vllm") or {}
    return {"gpus": len(gpus), "hbmUsedMiB": used, "hbmTotalMiB": total, "hbmPercent": used / total * 100.0 if total else None,
            "gfxActivityMean": sum(utils) / len(utils) if utils else None, "temperatureMax": max(temps) if temps else None,
            "powerW": sum(power) if power else None,
            "cpuPercent": pf((host.get("cpu") or {}).get("percent")), "memoryPercent": pf((host.get("memory") or {}).get("percent")),
            "diskPercent": pf((host.get("disk") or {}).get("percent")),
            "pods": len(summary.get("pods") or []),
            "alerts": {t: len(alerts.get(t) or []) for t in ("minor", "serious", "critical")},
            "vllmTokensPerSec": v.get("tokensPerSec") if v.get("available") else None,
            "perGpu": [{"index": g.get("index"), "hbmPercent": g.get("hbmPercent"), "util": g.get("utilization"),
                        "temperature": g.get("temperature"), "powerW": g.get("powerW")} for g in gpus]}


class PeerPoller:
    def __init__(self, peers: Sequence[str], timeout: float = 2.0, fetch: Optional[Callable[[str], Mapping[str, Any]]] = None,
                 clock: Callable[[], float] = time.time):
        self.peers = [p.rstrip("/") for p in peers if p]
        self.timeout = timeout
        self._fetch = fetch or self._http
        self._clock = clock
        self.nodes: Dict[str, Dict[str, Any]] = {}
        self.last_duration_s = 0.0

    def _http(self, url: str) -> Mapping[str, Any]:
        with urllib.request.urlopen(url + "/api/summary", timeout=self.timeout) as r:
            return json.loads(r.read().decode("utf-8"))

    def _one(self, url: str) -> Dict[str, Any]:
        t0 = time.perf_counter()
        try:
            s = self._fetch(url)
            d = digest(s)
            d.update(url=url, ok=True, error=None, sampleAge=(s.get("ages") or {}).get("gpu"))
        except Exception as e:
            prev = self.nodes.get(url) or {}
            d = {"url": url, "ok": False, "error": f"{type(e).__name__}: {e}", "gpus": prev.get("gpus", 0),
                 "alerts": {"minor": 0, "serious": 0, "critical": 0}}
        d["fetch_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        d["t"] = self._clock()
        return d

    def poll(self) -> None:
        t0 = time.perf_counter()
        if self.peers:
            with ThreadPoolExecutor(max_workers=min(16, len(self.peers))) as ex:
                for d in ex.map(self._one, self.peers):
                    self.nodes[d["url"]] = d
        self.last_duration_s = time.perf_counter() - t0

    def cluster(self, local_summary: Mapping[str, Any]) -> Dict[str, Any]:
        me = digest(local_summary)
        me.update(url="local", ok=True, error=None, node=socket.gethostname(), t=self._clock())
        nodes = [me] + [self.nodes[p] for p in self.peers if p in self.nodes]
        ok = [n for n in nodes if n.get("ok")]
        tot_used = sum(n.get("hbmUsedMiB") or 0 for n in ok)
        tot = sum(n.get("hbmTotalMiB") or 0 for n in ok)
        return {"nodes": nodes,
                "totals": {"nodes": len(nodes), "nodesDown": len(nodes) - len(ok), "gpus": sum(n.get("gpus") or 0 for n in ok),
                           "hbmUsedMiB": tot_used, "hbmTotalMiB": tot, "hbmPercent": tot_used / tot * 100.0 if tot else None,
                           "powerW": sum(n.get("powerW") or 0 for n in ok),
                           "alerts": {t: sum((n.get("alerts") or {}).get(t, 0) for n in nodes) for t in ("minor", "serious", "critical")}}}
