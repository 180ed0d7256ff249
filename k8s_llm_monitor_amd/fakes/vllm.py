"""A scripted vLLM-ROCm server: produces the ``/metrics`` text a real pod would expose.

The production parser and delta arithmetic (collectors.vllm.VllmDirectSource) run on this
text unchanged. BASELINE.json config 4 (K3s + vLLM-ROCm Llama-3-8B pod) without the
cluster; no reference counterpart.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, List

TTFT_BOUNDS = [0.001, 0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.1, 0.25, 0.5, 0.75, 1.0, 2.5, 5.0, 7.5, 10.0,
               20.0, 40.0, 80.0, 160.0, 640.0, 2560.0]
E2E_BOUNDS = [0.3, 0.5, 0.8, 1.0, 1.5, 2.0, 2.5, 5.0, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0, 120.0]
ITL_BOUNDS = [0.01, 0.025, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.75, 1.0, 2.5, 5.0]


@dataclass
class VllmState:
    tokens_per_s: float = 5200.0
    prompt_tokens_per_s: float = 21000.0
    req_per_s: float = 24.0
    ttft_median_s: float = 0.12
    ttft_sigma: float = 0.6          # log-normal shape
    e2e_median_s: float = 4.0
    itl_median_s: float = 0.018
    kv_usage: float = 0.55           # 0..1
    running: int = 48
    waiting: int = 0
    preempt_per_s: float = 0.0
    up: bool = True


Scenario = Callable[[float], VllmState]


def _steady(t: float) -> VllmState:
    w = math.sin(t / 23.0) * 0.5 + 0.5
    return VllmState(tokens_per_s=4800 + 900 * w, kv_usage=0.5 + 0.1 * w, running=int(40 + 16 * w))


def _kv_pressure(t: float) -> VllmState:
    """KV cache climbs through 70 / 85 / 95 % while the queue and TTFT grow."""
    s = _steady(t)
    s.kv_usage = 0.65 if t < 10 else 0.72 if t < 20 else 0.88 if t < 30 else 0.97
    s.waiting = 0 if t < 10 else 6 if t < 20 else 20 if t < 30 else 80
    s.ttft_median_s = 0.12 if t < 10 else 0.6 if t < 20 else 2.0 if t < 30 else 8.0
    s.preempt_per_s = 0.0 if t < 30 else 1.5
    return s


def _down(t: float) -> VllmState:
    s = _steady(t)
    s.up = t < 10
    return s


SCENARIOS: Dict[str, Scenario] = {"": _steady, "steady": _steady, "kv_pressure": _kv_pressure, "vllm_down": _down}


def _lognormal_cdf(x: float, median: float, sigma: float) -> float:
    if x <= 0:
        return 0.0
    return 0.5 * (1.0 + math.erf(math.log(x / median) / (sigma * math.sqrt(2.0))))


class FakeVllm:
    def __init__(self, scenario: Any = "", clock: Callable[[], float] = time.monotonic,
                 model: str = "meta-llama/Meta-Llama-3-8B-Instruct"):
        self.scenario: Scenario = scenario if callable(scenario) else SCENARIOS.get(scenario, _steady)
        self.clock = clock
        self.model = model
        self.t0 = clock()
        self._last = None
        self._gen = self._prompt = self._preempt = 0.0
        self._ttft = [0.0] * (len(TTFT_BOUNDS) + 1)
        self._e2e = [0.0] * (len(E2E_BOUNDS) + 1)
        self._itl = [0.0] * (len(ITL_BOUNDS) + 1)
        self._sums = {"ttft": 0.0, "e2e": 0.0, "itl": 0.0}

    def set_scenario(self, scenario: Any) -> None:
        self.scenario = scenario if callable(scenario) else SCENARIOS[scenario]
        self.t0 = self.clock()

    @staticmethod
    def _observe(hist: List[float], bounds: List[float], n: float, median: float, sigma: float) -> None:
        for i, b in enumerate(bounds):
            hist[i] += n * _lognormal_cdf(b, median, sigma)
        hist[-1] += n

    def _advance(self) -> VllmState:
        now = self.clock()
        s = self.scenario(now - self.t0)
        dt = 0.0 if self._last is None else max(0.0, now - self._last)
        self._last = now
        self._gen += s.tokens_per_s * dt
        self._prompt += s.prompt_tokens_per_s * dt
        self._preempt += s.preempt_per_s * dt
        n = s.req_per_s * dt
        self._observe(self._ttft, TTFT_BOUNDS, n, s.ttft_median_s, s.ttft_sigma)
        self._observe(self._e2e, E2E_BOUNDS, n, s.e2e_median_s, 0.5)
        self._observe(self._itl, ITL_BOUNDS, s.tokens_per_s * dt, s.itl_median_s, 0.4)
        self._sums["ttft"] += n * s.ttft_median_s
        self._sums["e2e"] += n * s.e2e_median_s
        self._sums["itl"] += s.tokens_per_s * dt * s.itl_median_s
        return s

    def fetch(self, url: str = "") -> str:
        s = self._advance()
        if not s.up:
            raise ConnectionRefusedError(url)
        lab = '{model_name="%s"}' % self.model
        L: List[str] = []

        def gauge(name: str, help_: str, v: float) -> None:
            L.extend([f"# HELP {name} {help_}", f"# TYPE {name} gauge", f"{name}{lab} {v}"])

        def counter(name: str, help_: str, v: float) -> None:
            L.extend([f"# HELP {name} {help_}", f"# TYPE {name} counter", f"{name}{lab} {v}"])

        def hist(name: str, help_: str, bounds: List[float], h: List[float], total: float) -> None:
            L.extend([f"# HELP {name} {help_}", f"# TYPE {name} histogram"])
            for b, c in zip(bounds, h):
                L.append('%s_bucket{le="%s",model_name="%s"} %s' % (name, b, self.model, c))
            L.append('%s_bucket{le="+Inf",model_name="%s"} %s' % (name, self.model, h[-1]))
            L.append(f"{name}_count{lab} {h[-1]}")
            L.append(f"{name}_sum{lab} {total}")

        gauge("vllm:num_requests_running", "Number of requests currently running on GPU.", float(s.running))
        gauge("vllm:num_requests_waiting", "Number of requests waiting to be processed.", float(s.waiting))
        gauge("vllm:gpu_cache_usage_perc", "GPU KV-cache usage. 1 means 100 percent usage.", s.kv_usage)
        counter("vllm:num_preemptions_total", "Cumulative number of preemption from the engine.", self._preempt)
        counter("vllm:prompt_tokens_total", "Number of prefill tokens processed.", self._prompt)
        counter("vllm:generation_tokens_total", "Number of generation tokens processed.", self._gen)
        hist("vllm:time_to_first_token_seconds", "Histogram of time to first token in seconds.", TTFT_BOUNDS, self._ttft, self._sums["ttft"])
        hist("vllm:e2e_request_latency_seconds", "Histogram of end to end request latency in seconds.", E2E_BOUNDS, self._e2e, self._sums["e2e"])
        hist("vllm:time_per_output_token_seconds", "Histogram of time per output token in seconds.", ITL_BOUNDS, self._itl, self._sums["itl"])
        return "\n".join(L) + "\n"
