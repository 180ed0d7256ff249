"""vLLM-compatible Prometheus metrics of the serving engine, plus exact ground truth.

The metric names, label (``model_name``) and histogram bucket bounds are the ones a vLLM server
exposes on ``/metrics``, so ``collectors/vllm.py`` (and a real Prometheus with the PromQL of
``config.py``) reads this engine exactly as it reads a vLLM-ROCm pod. Unlike vLLM the engine also
keeps every observation, so a validation run can compare what the dashboard shows (histogram
quantiles, counter rates between two scrapes) with the exact values (BASELINE.json config 4).
"""
from __future__ import annotations

import bisect
import math
import threading
import time
from collections import deque
from typing import Deque, Dict, List, Optional, Sequence, Tuple

TTFT_BUCKETS = (0.001, 0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.1, 0.25, 0.5, 0.75, 1.0, 2.5, 5.0, 7.5, 10.0,
                20.0, 40.0, 80.0, 160.0, 640.0, 2560.0)
ITL_BUCKETS = (0.001, 0.0025, 0.005, 0.0075, 0.01, 0.025, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.75, 1.0,
               2.5, 5.0, 7.5, 10.0, 20.0, 40.0, 80.0)
E2E_BUCKETS = (0.3, 0.5, 0.8, 1.0, 1.5, 2.0, 2.5, 5.0, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0, 120.0, 240.0,
               480.0, 960.0, 1920.0, 7680.0)


class Histogram:
    def __init__(self, bounds: Sequence[float]):
        self.bounds = tuple(bounds)
        self.counts = [0] * (len(self.bounds) + 1)     # last = +Inf
        self.total = 0
        self.sum = 0.0

    def observe(self, v: float) -> None:
        self.counts[bisect.bisect_left(self.bounds, v)] += 1
        self.total += 1
        self.sum += v

    def cumulative(self) -> List[Tuple[str, int]]:
        out, acc = [], 0
        for b, c in zip(self.bounds, self.counts):
            acc += c
            out.append((repr(float(b)), acc))
        out.append(("+Inf", self.total))
        return out


def exact_quantile(values: Sequence[float], q: float) -> Optional[float]:
    if not values:
        return None
    v = sorted(values)
    pos = q * (len(v) - 1)
    lo, hi = int(math.floor(pos)), int(math.ceil(pos))
    return v[lo] + (v[hi] - v[lo]) * (pos - lo)


class EngineMetrics:
    """Thread-safe: the engine thread writes, HTTP handler threads render."""

    def __init__(self, model_name: str, keep: int = 200000, clock=time.monotonic):
        self.model_name = model_name
        self._clock = clock
        self._lock = threading.Lock()
        self.running = 0
        self.waiting = 0
        self.kv_usage = 0.0                  # 0..1
        self.prompt_tokens = 0
        self.generation_tokens = 0
        self.preemptions = 0
        self.finished = 0
        self.rejected = 0
        self.steps = 0
        self.ttft = Histogram(TTFT_BUCKETS)
        self.itl = Histogram(ITL_BUCKETS)
        self.e2e = Histogram(E2E_BUCKETS)
        # ground truth: (time, value)
        self.ttft_log: Deque[Tuple[float, float]] = deque(maxlen=keep)
        self.itl_log: Deque[Tuple[float, float]] = deque(maxlen=keep)
        self.e2e_log: Deque[Tuple[float, float]] = deque(maxlen=keep)
        self.token_log: Deque[Tuple[float, int, int]] = deque(maxlen=keep)    # (time, generated, prompt)
        self.kv_log: Deque[Tuple[float, float, int, int]] = deque(maxlen=keep)  # (time, usage, running, waiting)

    # ---- writers (engine thread)
    def on_step(self, generated: int, prompt: int, running: int, waiting: int, kv_usage: float) -> None:
        t = self._clock()
        with self._lock:
            self.steps += 1
            self.generation_tokens += generated
            self.prompt_tokens += prompt
            self.running, self.waiting, self.kv_usage = running, waiting, kv_usage
            self.token_log.append((t, generated, prompt))
            self.kv_log.append((t, kv_usage, running, waiting))

    def on_first_token(self, ttft: float) -> None:
        with self._lock:
            self.ttft.observe(ttft)
            self.ttft_log.append((self._clock(), ttft))

    def on_token_gap(self, gap: float) -> None:
        with self._lock:
            self.itl.observe(gap)
            self.itl_log.append((self._clock(), gap))

    def on_finish(self, e2e: float) -> None:
        with self._lock:
            self.finished += 1
            self.e2e.observe(e2e)
            self.e2e_log.append((self._clock(), e2e))

    def on_preempt(self) -> None:
        with self._lock:
            self.preemptions += 1

    def on_reject(self) -> None:
        with self._lock:
            self.rejected += 1

    def set_queue(self, running: int, waiting: int, kv_usage: float) -> None:
        with self._lock:
            self.running, self.waiting, self.kv_usage = running, waiting, kv_usage

    # ---- readers
    def render(self) -> str:
        lab = '{model_name="%s"}' % self.model_name.replace('"', "'")
        L: List[str] = []

        def one(name: str, typ: str, help_: str, value) -> None:
            L.append(f"# HELP {name} {help_}")
            L.append(f"# TYPE {name} {typ}")
            L.append(f"{name}{lab} {float(value)!r}")

        def hist(name: str, help_: str, h: Histogram) -> None:
            L.append(f"# HELP {name} {help_}")
            L.append(f"# TYPE {name} histogram")
            for le, c in h.cumulative():
                L.append('%s_bucket{model_name="%s",le="%s"} %d' % (name, self.model_name, le, c))
            L.append(f"{name}_count{lab} {h.total}")
            L.append(f"{name}_sum{lab} {h.sum!r}")

        with self._lock:
            one("vllm:num_requests_running", "gauge", "Number of requests currently running on GPU.", self.running)
            one("vllm:num_requests_waiting", "gauge", "Number of requests waiting to be processed.", self.waiting)
            one("vllm:kv_cache_usage_perc", "gauge", "KV-cache usage. 1 means 100 percent usage.", self.kv_usage)
            one("vllm:gpu_cache_usage_perc", "gauge", "GPU KV-cache usage (legacy name).", self.kv_usage)
            one("vllm:prompt_tokens_total", "counter", "Number of prefill tokens processed.", self.prompt_tokens)
            one("vllm:generation_tokens_total", "counter", "Number of generation tokens processed.", self.generation_tokens)
            one("vllm:num_preemptions_total", "counter", "Cumulative number of preemptions from the engine.", self.preemptions)
            L.append("# HELP vllm:request_success_total Count of successfully processed requests.")
            L.append("# TYPE vllm:request_success_total counter")
            L.append('vllm:request_success_total{model_name="%s",finished_reason="length"} %r'
                     % (self.model_name, float(self.finished)))
            hist("vllm:time_to_first_token_seconds", "Histogram of time to first token in seconds.", self.ttft)
            hist("vllm:inter_token_latency_seconds", "Histogram of inter-token latency in seconds.", self.itl)
            hist("vllm:time_per_output_token_seconds", "Histogram of time per output token in seconds.", self.itl)
            hist("vllm:e2e_request_latency_seconds", "Histogram of end to end request latency in seconds.", self.e2e)
        return "\n".join(L) + "\n"

    def truth(self, window_s: Optional[float] = None, now: Optional[float] = None) -> Dict[str, object]:
        """Exact figures over the last `window_s` seconds (everything kept when None)."""
        now = self._clock() if now is None else now
        t0 = -math.inf if window_s is None else now - window_s
        with self._lock:
            ttft = [v for (t, v) in self.ttft_log if t >= t0]
            itl = [v for (t, v) in self.itl_log if t >= t0]
            e2e = [v for (t, v) in self.e2e_log if t >= t0]
            tok = [(t, g, p) for (t, g, p) in self.token_log if t >= t0]
            kv = [(t, u, r, w) for (t, u, r, w) in self.kv_log if t >= t0]
            totals = {"prompt_tokens": self.prompt_tokens, "generation_tokens": self.generation_tokens,
                      "finished": self.finished, "preemptions": self.preemptions, "rejected": self.rejected,
                      "steps": self.steps, "running": self.running, "waiting": self.waiting,
                      "kv_usage": self.kv_usage}
        span = (window_s if window_s is not None else (tok[-1][0] - tok[0][0] if len(tok) > 1 else 0.0)) or 0.0
        gen = sum(g for (_, g, _) in tok)
        prm = sum(p for (_, _, p) in tok)
        return {
            "window_s": span, "requests_first_token": len(ttft), "requests_finished": len(e2e),
            "ttft_p50": exact_quantile(ttft, 0.5), "ttft_p99": exact_quantile(ttft, 0.99),
            "ttft_mean": (sum(ttft) / len(ttft)) if ttft else None,
            "itl_p50": exact_quantile(itl, 0.5), "itl_p99": exact_quantile(itl, 0.99),
            "e2e_p50": exact_quantile(e2e, 0.5),
            "generation_tokens": gen, "prompt_tokens": prm,
            "tokens_per_s": gen / span if span > 0 else None,
            "prompt_tokens_per_s": prm / span if span > 0 else None,
            "kv_usage_mean": (sum(u for (_, u, _, _) in kv) / len(kv)) if kv else None,
            "kv_usage_max": max((u for (_, u, _, _) in kv), default=None),
            "running_mean": (sum(r for (_, _, r, _) in kv) / len(kv)) if kv else None,
            "totals": totals,
        }
