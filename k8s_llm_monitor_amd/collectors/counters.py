"""Cumulative-counter arithmetic: deltas with wrap handling, rates, residency shares.

amdsmi reports xGMI traffic, energy, GFX/MEM activity and throttle residency as
monotonically increasing accumulators (``xgmi_read_data_acc`` in KiB, ``energy_accumulator``
in counter ticks, ``gfx_activity_acc`` / ``*_residency_acc`` against ``accumulation_counter``).
A dashboard value is a delta over the sample period. No reference
counterpart: the reference only reads instantaneous nvidia-smi values (monitor_server.js:85).
"""
from __future__ import annotations

from typing import Dict, Hashable, Optional, Tuple


def counter_delta(prev: int, cur: int, bits: int = 64) -> Optional[int]:
    """cur - prev for an unsigned `bits`-wide counter that may have wrapped once.

    A counter that went backwards by more than half its range is taken to have been
    reset (driver reload), for which no delta exists: None.
    """
    if prev is None or cur is None:
        return None
    if cur >= prev:
        return cur - prev
    span = 1 << bits
    d = cur + span - prev
    if d < span // 2:
        return d
    return None


class RateTracker:
    """Remembers the last (time, value) per key and turns new readings into rates."""

    def __init__(self, bits: int = 64):
        self.bits = bits
        self._last: Dict[Hashable, Tuple[float, int]] = {}

    def update(self, key: Hashable, t: float, value: Optional[int]) -> Optional[float]:
        """Units of `value` per second since the previous call for `key`; None on the first
        reading, on a reset, or when dt <= 0."""
        if value is None:
            return None
        prev = self._last.get(key)
        self._last[key] = (t, int(value))
        if prev is None:
            return None
        dt = t - prev[0]
        if dt <= 0:
            return None
        d = counter_delta(prev[1], int(value), self.bits)
        if d is None:
            return None
        return d / dt

    def delta(self, key: Hashable, value: Optional[int]) -> Optional[int]:
        """Plain delta (no time base) since the previous call for `key`."""
        if value is None:
            return None
        k = ("d", key)
        prev = self._last.get(k)
        self._last[k] = (0.0, int(value))
        if prev is None:
            return None
        return counter_delta(prev[1], int(value), self.bits)

    def forget(self, key: Hashable) -> None:
        self._last.pop(key, None)
        self._last.pop(("d", key), None)


def share(num_delta: Optional[int], den_delta: Optional[int], scale: float = 100.0) -> Optional[float]:
    """num/den*scale for two accumulator deltas, None when the denominator did not move."""
    if num_delta is None or den_delta is None or den_delta <= 0:
        return None
    return num_delta / den_delta * scale
