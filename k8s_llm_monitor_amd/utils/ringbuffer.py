"""Fixed-capacity time series ring buffer with optional on-disk persistence.

The reference keeps no history of its own: every chart point comes from Prometheus
(monitor_server.js:117-154), so the dashboard is empty without one. This ring buffer is
the local history source and the optional checkpoint/resume of
SURVEY.md section 5: ``save``/``load`` write a small JSON file so that a restarted server
still shows the last window.
"""
from __future__ import annotations

import json
import os
import threading
from typing import Dict, Iterable, List, Optional, Tuple


class SeriesRing:
    """(timestamp, value) pairs, oldest dropped first. Thread-safe."""

    def __init__(self, capacity: int):
        if capacity <= 0:
            raise ValueError("capacity must be positive")
        self.capacity = int(capacity)
        self._t: List[float] = [0.0] * self.capacity
        self._v: List[float] = [0.0] * self.capacity
        self._n = 0
        self._head = 0  # next write position
        self._lock = threading.Lock()

    def __len__(self) -> int:
        return self._n

    def append(self, t: float, v: float) -> None:
        with self._lock:
            self._t[self._head] = float(t)
            self._v[self._head] = float(v)
            self._head = (self._head + 1) % self.capacity
            if self._n < self.capacity:
                self._n += 1

    def items(self, since: Optional[float] = None) -> List[Tuple[float, float]]:
        with self._lock:
            start = (self._head - self._n) % self.capacity
            out = []
            for i in range(self._n):
                j = (start + i) % self.capacity
                if since is None or self._t[j] >= since:
                    out.append((self._t[j], self._v[j]))
            return out

    def last(self) -> Optional[Tuple[float, float]]:
        with self._lock:
            if self._n == 0:
                return None
            j = (self._head - 1) % self.capacity
            return (self._t[j], self._v[j])

    def extend(self, pairs: Iterable[Tuple[float, float]]) -> None:
        for t, v in pairs:
            self.append(t, v)


class HistoryStore:
    """Named SeriesRings sampled on a common step."""

    def __init__(self, capacity: int, names: Iterable[str] = ()):
        self.capacity = capacity
        self._series: Dict[str, SeriesRing] = {}
        self._lock = threading.Lock()
        for n in names:
            self.series(n)

    def series(self, name: str) -> SeriesRing:
        with self._lock:
            s = self._series.get(name)
            if s is None:
                s = self._series[name] = SeriesRing(self.capacity)
            return s

    def names(self) -> List[str]:
        with self._lock:
            return sorted(self._series)

    def record(self, t: float, values: Dict[str, Optional[float]]) -> None:
        for k, v in values.items():
            if v is None or v != v:
                continue
            self.series(k).append(t, v)

    def window(self, name: str, since: Optional[float] = None) -> List[Tuple[float, float]]:
        with self._lock:
            s = self._series.get(name)
        return s.items(since) if s else []

    def save(self, path: str) -> None:
        data = {"capacity": self.capacity,
                "series": {n: self.series(n).items() for n in self.names()}}
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(data, f)
        os.replace(tmp, path)

    def load(self, path: str) -> bool:
        """Restore series from `path`; returns False (and changes nothing) if unusable."""
        try:
            with open(path, "r", encoding="utf-8") as f:
                data = json.load(f)
            series = data["series"]
            if not isinstance(series, dict):
                return False
            parsed = {}
            for n, pairs in series.items():
                parsed[str(n)] = [(float(t), float(v)) for t, v in pairs]
        except (OSError, ValueError, KeyError, TypeError):
            return False
        for n, pairs in parsed.items():
            self.series(n).extend(pairs)
        return True
