"""Open-loop request generator for the serving engine: Poisson arrivals, random token prompts.

Open loop (arrival times do not depend on completions) is what makes queueing visible: above
the engine's capacity the waiting count and TTFT grow, which is the condition the monitor's
vLLM alerts watch for. Requests go straight into ``Engine.submit`` (no HTTP hop), so the client
costs nothing measurable next to a model step.
"""
from __future__ import annotations

import random
import threading
import time
from typing import Callable, List, Optional

from .engine import Engine, Request


class OpenLoopLoad:
    def __init__(self, engine: Engine, rate_rps: float, prompt_len: int = 512, output_len: int = 128,
                 jitter: float = 0.25, seed: int = 0, clock: Callable[[], float] = time.monotonic):
        self.engine, self.rate = engine, float(rate_rps)
        self.prompt_len, self.output_len, self.jitter = int(prompt_len), int(output_len), float(jitter)
        self.rng = random.Random(seed)
        self.clock = clock
        self.requests: List[Request] = []
        self._stop = threading.Event()
        self._th: Optional[threading.Thread] = None

    def _length(self, mean: int) -> int:
        lo, hi = max(1, int(mean * (1 - self.jitter))), max(1, int(mean * (1 + self.jitter)))
        return self.rng.randint(lo, hi)

    def one(self) -> Request:
        n = self._length(self.prompt_len)
        vocab = self.engine.mcfg.vocab
        r = self.engine.submit([self.rng.randrange(vocab) for _ in range(n)], self._length(self.output_len))
        self.requests.append(r)
        return r

    def burst(self, n: int) -> List[Request]:
        return [self.one() for _ in range(n)]

    def _loop(self, duration: float) -> None:
        t_end = self.clock() + duration
        nxt = self.clock()
        while not self._stop.is_set():
            now = self.clock()
            if now >= t_end:
                break
            if now >= nxt:
                self.one()
                nxt += self.rng.expovariate(self.rate) if self.rate > 0 else duration
            else:
                self._stop.wait(min(nxt - now, 0.05))

    def start(self, duration: float) -> "OpenLoopLoad":
        self._stop.clear()
        self._th = threading.Thread(target=self._loop, args=(duration,), name="klm-llm-load", daemon=True)
        self._th.start()
        return self

    def join(self, timeout: Optional[float] = None) -> None:
        if self._th:
            self._th.join(timeout)

    def stop(self) -> None:
        self._stop.set()
        self.join(5)

    def wait_all(self, timeout: float = 600.0) -> bool:
        t_end = time.monotonic() + timeout
        for r in list(self.requests):
            if not r.done.wait(max(0.0, t_end - time.monotonic())):
                return False
        return True
