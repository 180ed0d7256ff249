"""vLLM-ROCm serving metrics: TTFT, tokens/s, KV-cache usage, running / waiting requests.

No reference counterpart: vLLM appears in the reference only as a README bullet
(README.md:73). BASELINE.json north_star item 3 asks for these metrics
"scraped from Prometheus"; two sources give the same sample shape:

  * ``VllmPrometheusSource`` - instant PromQL queries through the same client that serves
    ``/api/history`` (collectors.prometheus, reference :27-36);
  * ``VllmDirectSource``     - scrapes the ``/metrics`` text exposition of each vLLM pod
    itself and derives rates and histogram quantiles from counter deltas between two
    scrapes, so the panel works without a Prometheus server.

Sample shape (``GET /api/vllm/metrics``)::

    {"available": true, "source": "...", "endpoints": 2, "models": ["meta-llama/..."],
     "ttftP50": 0.12, "ttftP99": 0.80,            # seconds
     "tokensPerSec": 5321.0, "promptTokensPerSec": 20111.0,
     "kvCachePercent": 63.2, "running": 48, "waiting": 3,
     "e2eP50": 4.1, "itlP50": 0.018, "preemptionsPerSec": 0.0}
"""
from __future__ import annotations

import math
import re
import time
import urllib.request
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

from .counters import RateTracker

Sample = Tuple[str, Dict[str, str], float]

_LINE = re.compile(r"^([a-zA-Z_:][a-zA-Z0-9_:]*)(\{.*\})?\s+(\S+)(?:\s+\S+)?$")
_LABEL = re.compile(r'([a-zA-Z_][a-zA-Z0-9_]*)="((?:[^"\\]|\\.)*)"')

# vLLM renamed a few series over time; the first name that exists wins
KV_NAMES = ("vllm:kv_cache_usage_perc", "vllm:gpu_cache_usage_perc")
RUNNING = "vllm:num_requests_running"
WAITING = "vllm:num_requests_waiting"
GEN_TOKENS = ("vllm:generation_tokens_total", "vllm:generation_tokens")
PROMPT_TOKENS = ("vllm:prompt_tokens_total", "vllm:prompt_tokens")
PREEMPT = ("vllm:num_preemptions_total", "vllm:num_preemptions")
TTFT = "vllm:time_to_first_token_seconds"
E2E = "vllm:e2e_request_latency_seconds"
ITL = ("vllm:inter_token_latency_seconds", "vllm:time_per_output_token_seconds")


def parse_exposition(text: str) -> List[Sample]:
    """Prometheus text format 0.0.4 -> [(name, labels, value)]. Comment lines and lines that do
    not parse are skipped."""
    out: List[Sample] = []
    for line in text.splitlines():
        line = line.strip()
        if not line or line[0] == "#":
            continue
        m = _LINE.match(line)
        if not m:
            continue
        name, lab, val = m.group(1), m.group(2), m.group(3)
        try:
            v = float(val.replace("+Inf", "inf").replace("-Inf", "-inf"))
        except ValueError:
            continue
        labels = {k: v2.replace('\\"', '"').replace("\\\\", "\\").replace("\\n", "\n")
                  for k, v2 in _LABEL.findall(lab)} if lab else {}
        out.append((name, labels, v))
    return out


def histogram_quantile(q: float, buckets: Sequence[Tuple[float, float]]) -> Optional[float]:
    """PromQL ``histogram_quantile`` on cumulative ``(le, count)`` buckets (linear interpolation
    inside the bucket, the highest finite bound when the quantile falls in +Inf)."""
    b = sorted(buckets)
    if not b or not math.isinf(b[-1][0]):
        return None
    total = b[-1][1]
    if total <= 0 or not (0 <= q <= 1):
        return None
    rank = q * total
    prev_le, prev_c = 0.0, 0.0
    for le, c in b:
        if c >= rank:
            if math.isinf(le):
                return prev_le if len(b) > 1 else None
            if c == prev_c:
                return le
            return prev_le + (le - prev_le) * (rank - prev_c) / (c - prev_c)
        prev_le, prev_c = le, c
    return None


def _sum(samples: Iterable[Sample], names: Sequence[str]) -> Optional[float]:
    for n in names:
        vals = [v for (name, _, v) in samples if name == n]
        if vals:
            return sum(vals)
    return None


def _max(samples: Iterable[Sample], names: Sequence[str]) -> Optional[float]:
    for n in names:
        vals = [v for (name, _, v) in samples if name == n]
        if vals:
            return max(vals)
    return None


def _buckets(samples: Iterable[Sample], names: Sequence[str]) -> Dict[float, float]:
    """Bucket counts summed over every series of the first histogram that exists."""
    for n in names:
        out: Dict[float, float] = {}
        for (name, labels, v) in samples:
            if name == n + "_bucket" and "le" in labels:
                try:
                    le = float(labels["le"].replace("+Inf", "inf"))
                except ValueError:
                    continue
                out[le] = out.get(le, 0.0) + v
        if out:
            return out
    return {}


class VllmSource:
    kind = "none"

    def __init__(self):
        self.last_error: Optional[str] = None
        self.last_duration_s = 0.0

    def _sample(self) -> Dict[str, Any]:
        return {"available": False}

    def sample(self) -> Dict[str, Any]:
        t0 = time.perf_counter()
        self.last_error = None
        try:
            out = self._sample()
        except Exception as e:
            out = {"available": False}
            self.last_error = f"{type(e).__name__}: {e}"
        out.setdefault("source", self.describe())
        self.last_duration_s = time.perf_counter() - t0
        return out

    def describe(self) -> str:
        return self.kind

    def close(self) -> None:
        pass


def _http_text(url: str, timeout: float) -> str:
    with urllib.request.urlopen(url, timeout=timeout) as r:
        return r.read().decode("utf-8", "replace")


class VllmDirectSource(VllmSource):
    """Scrape ``/metrics`` of each endpoint; rates and quantiles come from deltas between
    this scrape and the previous one (first scrape: gauges only)."""
    kind = "direct"

    def __init__(self, endpoints: Sequence[str], timeout: float = 2.0,
                 fetch: Optional[Callable[[str], str]] = None, clock: Callable[[], float] = time.monotonic):
        super().__init__()
        self.endpoints = [e if e.rstrip("/").endswith("/metrics") else e.rstrip("/") + "/metrics" for e in endpoints]
        self.timeout = timeout
        self._fetch = fetch or (lambda u: _http_text(u, self.timeout))
        self._clock = clock
        self._rates = RateTracker(bits=63)
        self._prev_buckets: Dict[str, Dict[float, float]] = {}

    def _replica_rows(self, per_ep: List[Tuple[str, Optional[List[Sample]]]], t: float) -> List[Dict[str, Any]]:
        """One row per serving replica: per endpoint, and per ``replica`` label inside an endpoint
        (the router of serving/router.py merges its replicas into one exposition). The totals hide a
        saturated replica next to idle ones; these rows do not."""
        rows: List[Dict[str, Any]] = []
        for ep, smp in per_ep:
            if smp is None:
                rows.append({"endpoint": ep, "up": False})
                continue
            reps = sorted({l["replica"] for (_, l, _) in smp if "replica" in l}) or [None]
            for rep in reps:
                mine = [x for x in smp if rep is None or x[1].get("replica") == rep]
                kv = _max(mine, KV_NAMES)
                gen = _sum(mine, GEN_TOKENS)
                key = f"gen|{ep}|{rep}"
                rows.append({"endpoint": ep, "replica": rep, "up": True,
                             "running": _sum(mine, (RUNNING,)), "waiting": _sum(mine, (WAITING,)),
                             "kvCachePercent": None if kv is None else kv * 100.0,
                             "tokensPerSec": self._rates.update(key, t, None if gen is None else int(gen))})
        return rows

    def _sample(self) -> Dict[str, Any]:
        samples: List[Sample] = []
        per_ep: List[Tuple[str, Optional[List[Sample]]]] = []
        ok = 0
        errs = []
        for ep in self.endpoints:
            try:
                smp = parse_exposition(self._fetch(ep))
                samples.extend(smp)
                per_ep.append((ep, smp))
                ok += 1
            except Exception as e:
                errs.append(f"{ep}: {type(e).__name__}")
                per_ep.append((ep, None))
        if ok == 0:
            if errs:
                self.last_error = "; ".join(errs)
            return {"available": False, "endpoints": 0}
        t = self._clock()
        out: Dict[str, Any] = {"available": True, "endpoints": ok}
        rows = self._replica_rows(per_ep, t)
        if len(rows) > 1:
            out["replicas"] = rows
        out["models"] = sorted({l["model_name"] for (_, l, _) in samples if "model_name" in l})
        kv = _max(samples, KV_NAMES)
        out["kvCachePercent"] = None if kv is None else kv * 100.0
        out["running"] = _sum(samples, (RUNNING,))
        out["waiting"] = _sum(samples, (WAITING,))

        def rate(key: str, names: Sequence[str]) -> Optional[float]:
            v = _sum(samples, names)
            return self._rates.update(key, t, None if v is None else int(v))

        out["tokensPerSec"] = rate("gen", GEN_TOKENS)
        out["promptTokensPerSec"] = rate("prompt", PROMPT_TOKENS)
        out["preemptionsPerSec"] = rate("preempt", PREEMPT)

        def quantiles(key: str, names: Sequence[str], qs: Sequence[float]) -> List[Optional[float]]:
            cur = _buckets(samples, names)
            prev = self._prev_buckets.get(key)
            self._prev_buckets[key] = cur
            if not cur:
                return [None] * len(qs)
            if prev and set(prev) == set(cur):
                d = {le: cur[le] - prev[le] for le in cur}
                if any(v < 0 for v in d.values()):      # a restarted server
                    d = cur
                elif max(d.values()) <= 0:
                    return [None] * len(qs)             # no request finished in this window
            else:
                d = cur                                   # first scrape: lifetime distribution
            b = list(d.items())
            return [histogram_quantile(q, b) for q in qs]

        out["ttftP50"], out["ttftP99"] = quantiles("ttft", (TTFT,), (0.5, 0.99))
        out["e2eP50"], = quantiles("e2e", (E2E,), (0.5,))
        out["itlP50"], = quantiles("itl", ITL, (0.5,))
        return out

    def describe(self) -> str:
        return f"direct({len(self.endpoints)} endpoint(s))"


class VllmPrometheusSource(VllmSource):
    """The same figures as PromQL instant queries (query strings come from the config)."""
    kind = "prometheus"

    def __init__(self, client, pcfg):
        super().__init__()
        self.client, self.p = client, pcfg

    def _sample(self) -> Dict[str, Any]:
        q, p = self.client.query, self.p
        running = q(p.q_vllm_running)
        kv = q(p.q_vllm_kv_cache)
        if running is None and kv is None:
            return {"available": False}

        def clean(x: Optional[float]) -> Optional[float]:
            return None if x is None or math.isnan(x) or math.isinf(x) else x

        return {"available": True, "endpoints": None, "models": [],
                "ttftP50": clean(q(p.q_vllm_ttft_p50)), "ttftP99": clean(q(p.q_vllm_ttft_p99)),
                "tokensPerSec": clean(q(p.q_vllm_tokens_per_s)),
                "promptTokensPerSec": clean(q(p.q_vllm_prompt_tokens_per_s)),
                "kvCachePercent": clean(kv), "running": clean(running), "waiting": clean(q(p.q_vllm_waiting))}

    def describe(self) -> str:
        return f"prometheus({self.client.url})"


def open_vllm_source(kind: str = "auto", cfg=None, prom=None) -> VllmSource:
    """auto: direct scraping when endpoints are configured, else Prometheus when it answers,
    else none."""
    src = getattr(cfg, "sources", None)
    endpoints = list(getattr(src, "vllm_endpoints", []) or [])
    kind = (kind or "auto").lower()
    if kind == "none":
        return VllmSource()
    if kind == "fake":
        from ..fakes.vllm import FakeVllm
        fv = FakeVllm(scenario=getattr(src, "fake_scenario", "") or "")
        s = VllmDirectSource(["fake://vllm-0/metrics"], fetch=fv.fetch, clock=fv.clock)
        s.kind = "fake"
        s.fake = fv   # type: ignore[attr-defined]
        return s
    if kind == "direct":
        return VllmDirectSource(endpoints)
    if kind == "prometheus":
        return VllmPrometheusSource(prom, cfg.prometheus)
    if kind == "auto":
        if endpoints:
            return VllmDirectSource(endpoints)
        if prom is not None and prom.reachable():
            return VllmPrometheusSource(prom, cfg.prometheus)
        return VllmSource()
    raise ValueError(f"unknown vllm source {kind!r}")
