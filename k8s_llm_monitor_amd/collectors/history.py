"""History for the line charts: ``GET /api/history``.

Reference: ``getHistoryMetrics`` (monitor_server.js:117-154): six PromQL range queries over
30 min at a 30 s step, awaited one after another, only ``results[0]`` of each used (:139),
shaped by ``processResult`` (:137-144) into ``{"labels": ["HH:MM"...], "data": ["12.3"...]}``
with string-typed values (``toFixed(1)``) and ``{"labels": [], "data": []}`` for an empty
series (:138). Keys: cpu, memory, disk, gpuUtil, gpuMem, gpuTemp (:146-153).

Two sources behind one shape:
  * ``PrometheusHistory`` - the same six queries (DCGM names replaced by AMD
    device-metrics-exporter names from the config), issued concurrently;
  * ``LocalHistory``      - the sampler's own ring buffer, so the charts work on a host
    without Prometheus. Values are means over each step bucket.

Extra keys (additive): gpuMemActivity, gpuPower, xgmi, per-GPU series under ``perGpu``,
and the vLLM series.
"""
from __future__ import annotations

import math
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Tuple

from ..utils.ringbuffer import HistoryStore
from ..utils.timefmt import fixed1, hhmm, parse_duration

REFERENCE_KEYS = ("cpu", "memory", "disk", "gpuUtil", "gpuMem", "gpuTemp")
EXTRA_KEYS = ("gpuMemActivity", "gpuPower", "gpuHbmPercent", "xgmi",
              "vllmTokensPerSec", "vllmTtftP50", "vllmTtftP99", "vllmKvCache", "vllmRunning", "vllmWaiting")


def empty_series() -> Dict[str, list]:
    return {"labels": [], "data": []}


def process_result(results: Optional[Sequence[Mapping[str, Sequence[float]]]]) -> Dict[str, list]:
    """monitor_server.js:137-144."""
    if not results:
        return empty_series()
    r = results[0]
    return {"labels": [hhmm(t / 1000.0) for t in r["time"]],
            "data": [fixed1(v) for v in r["value"]]}


def shape_pairs(pairs: Sequence[Tuple[float, float]]) -> Dict[str, list]:
    return {"labels": [hhmm(t) for t, _ in pairs], "data": [fixed1(v) for _, v in pairs]}


class PrometheusHistory:
    kind = "prometheus"

    def __init__(self, client, queries: Mapping[str, str], window: str = "30m", step: str = "30s",
                 parallel: bool = True):
        self.client = client
        self.queries = dict(queries)
        self.window, self.step = window, step
        self.parallel = parallel
        self.last_duration_s = 0.0

    def payload(self) -> Dict[str, Dict[str, list]]:
        t0 = time.perf_counter()
        keys = list(self.queries)

        def one(k: str):
            return self.client.query_range(self.queries[k], self.window, self.step)

        if self.parallel and len(keys) > 1:
            with ThreadPoolExecutor(max_workers=min(8, len(keys))) as ex:
                results = list(ex.map(one, keys))
        else:
            results = [one(k) for k in keys]
        out = {k: process_result(r) for k, r in zip(keys, results)}
        for k in REFERENCE_KEYS:
            out.setdefault(k, empty_series())
        self.last_duration_s = time.perf_counter() - t0
        return out


def bucket_mean(pairs: Sequence[Tuple[float, float]], step: float, end: float, window: float) -> List[Tuple[float, float]]:
    """Mean of the samples in each ``step``-wide bucket of [end-window, end]; empty buckets
    are skipped (like gaps in a Prometheus range result)."""
    if step <= 0:
        return list(pairs)
    start = end - window
    acc: Dict[int, List[float]] = {}
    for t, v in pairs:
        if t < start or t > end + 1e-9:
            continue
        b = int((t - start) // step)
        a = acc.get(b)
        if a is None:
            acc[b] = [v, 1.0]
        else:
            a[0] += v
            a[1] += 1.0
    return [(start + b * step, s / n) for b, (s, n) in sorted(acc.items())]


class LocalHistory:
    """Chart series from the sampler's ring buffer."""
    kind = "local"

    def __init__(self, store: HistoryStore, window: str = "30m", step_s: float = 30.0,
                 now: Callable[[], float] = time.time):
        self.store = store
        self.window_s = float(parse_duration(window))
        self.step_s = float(step_s)
        self._now = now
        self.last_duration_s = 0.0

    def series(self, name: str, now: Optional[float] = None) -> List[Tuple[float, float]]:
        now = self._now() if now is None else now
        pairs = self.store.window(name, since=now - self.window_s)
        return bucket_mean(pairs, self.step_s, now, self.window_s)

    def payload(self) -> Dict[str, Any]:
        t0 = time.perf_counter()
        now = self._now()
        out: Dict[str, Any] = {}
        for k in REFERENCE_KEYS + EXTRA_KEYS:
            s = self.series(k, now)
            if s or k in REFERENCE_KEYS:
                out[k] = shape_pairs(s)
        per: Dict[str, Dict[str, Dict[str, list]]] = {}
        for name in self.store.names():
            if not name.startswith("gpu"):
                continue
            # per-GPU series are named gpu<i>.<metric>
            head, _, metric = name.partition(".")
            if not metric or not head[3:].isdigit():
                continue
            per.setdefault(head[3:], {})[metric] = shape_pairs(self.series(name, now))
        if per:
            out["perGpu"] = per
        self.last_duration_s = time.perf_counter() - t0
        return out


def history_values(host: Mapping[str, Any], gpus: Sequence[Mapping[str, Any]],
                   vllm: Optional[Mapping[str, Any]] = None) -> Dict[str, Optional[float]]:
    """One sampler tick -> values recorded into the ring buffer. GPU aggregates are means
    over GPUs, like ``avg(...)`` in the reference's PromQL (:128-134)."""
    def pf(x):
        try:
            return float(x)
        except (TypeError, ValueError):
            return None

    def mean(xs):
        xs = [x for x in xs if isinstance(x, (int, float)) and not (isinstance(x, float) and math.isnan(x))]
        return sum(xs) / len(xs) if xs else None

    cpu = host.get("cpu") or {}
    v: Dict[str, Optional[float]] = {
        # the history chart shows true utilisation (node_cpu in the reference, :119), not load
        "cpu": pf(cpu.get("util_percent")) if cpu.get("util_percent") is not None else pf(cpu.get("percent")),
        "memory": pf((host.get("memory") or {}).get("percent")),
        "disk": pf((host.get("disk") or {}).get("percent")),
    }
    if gpus:
        v["gpuUtil"] = mean([g.get("gfxActivityAvg") if g.get("gfxActivityAvg") is not None else g.get("utilization") for g in gpus])
        v["gpuMem"] = mean([g.get("memoryUsed") for g in gpus])
        v["gpuTemp"] = mean([g.get("temperature") for g in gpus])
        v["gpuMemActivity"] = mean([g.get("memActivityAvg") if g.get("memActivityAvg") is not None else g.get("memActivity") for g in gpus])
        v["gpuPower"] = mean([g.get("powerAvgW") if g.get("powerAvgW") is not None else g.get("powerW") for g in gpus])
        v["gpuHbmPercent"] = mean([g.get("hbmPercent") for g in gpus])
        xs = [((g.get("xgmi") or {}).get("rxGBps") or 0.0) + ((g.get("xgmi") or {}).get("txGBps") or 0.0)
              for g in gpus if (g.get("xgmi") or {}).get("rxGBps") is not None]
        v["xgmi"] = sum(xs) if xs else None
        for g in gpus:
            i = g.get("index", 0)
            v[f"gpu{i}.util"] = g.get("gfxActivityAvg") if g.get("gfxActivityAvg") is not None else g.get("utilization")
            v[f"gpu{i}.hbm"] = g.get("hbmPercent")
            v[f"gpu{i}.temp"] = g.get("temperature")
            v[f"gpu{i}.power"] = g.get("powerW")
    if vllm:
        v["vllmTokensPerSec"] = vllm.get("tokensPerSec")
        v["vllmTtftP50"] = vllm.get("ttftP50")
        v["vllmTtftP99"] = vllm.get("ttftP99")
        v["vllmKvCache"] = vllm.get("kvCachePercent")
        v["vllmRunning"] = vllm.get("running")
        v["vllmWaiting"] = vllm.get("waiting")
    return v


def prometheus_queries(cfg) -> Dict[str, str]:
    p = cfg.prometheus
    return {"cpu": p.q_cpu, "memory": p.q_memory, "disk": p.q_disk,
            "gpuUtil": p.q_gpu_util, "gpuMem": p.q_gpu_mem, "gpuTemp": p.q_gpu_temp}
