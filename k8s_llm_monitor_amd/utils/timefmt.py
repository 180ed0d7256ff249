"""Duration / age / clock-label helpers.

Reference behaviour being reproduced:
  * ``parse_duration``: monitor_server.js:54-63 (regex ``(\\d+)([smhd])``, default 1800 s).
  * ``format_age``: monitor_server.js:106-110 (``"Xd Yh"``, ``"Xh Ym"``, ``"Xm"``; ``"0m"``
    without a start time).
  * ``hhmm``: monitor_server.js:141 (``toLocaleTimeString('zh-CN', {hour:'2-digit',
    minute:'2-digit'})`` gives 24-hour ``HH:MM`` in local time).
"""
from __future__ import annotations

import re
import time
from datetime import datetime, timezone
from typing import Optional

_DUR = re.compile(r"(\d+)([smhd])")
_UNIT = {"s": 1, "m": 60, "h": 3600, "d": 86400}


def parse_duration(s: object, default: int = 1800) -> int:
    """'30m' -> 1800. First match anywhere in the string, like the reference's regex."""
    if not isinstance(s, str):
        return default
    m = _DUR.search(s)
    if not m:
        return default
    return int(m.group(1)) * _UNIT[m.group(2)]


def parse_k8s_time(s: Optional[str]) -> Optional[float]:
    """RFC 3339 timestamp as the Kubernetes API prints it -> epoch seconds, None if unusable."""
    if not s or not isinstance(s, str):
        return None
    try:
        t = s.strip()
        if t.endswith("Z"):
            t = t[:-1] + "+00:00"
        # Python 3.10's fromisoformat wants at most microseconds
        m = re.match(r"(.*\.\d{6})\d+(.*)$", t)
        if m:
            t = m.group(1) + m.group(2)
        dt = datetime.fromisoformat(t)
        if dt.tzinfo is None:
            dt = dt.replace(tzinfo=timezone.utc)
        return dt.timestamp()
    except ValueError:
        return None


def format_age(start_epoch: Optional[float], now: Optional[float] = None) -> str:
    if start_epoch is None:
        return "0m"
    now = time.time() if now is None else now
    s = max(0.0, now - start_epoch)
    d, h, m = int(s // 86400), int((s % 86400) // 3600), int((s % 3600) // 60)
    if d > 0:
        return f"{d}d {h}h"
    if h > 0:
        return f"{h}h {m}m"
    return f"{m}m"


def hhmm(epoch_s: float) -> str:
    return time.strftime("%H:%M", time.localtime(epoch_s))


def fixed1(x: float) -> str:
    """JavaScript ``toFixed(1)`` for the values this API produces (monitor_server.js:76-78, :142).

    JS rounds the exact binary value half away from zero on the decimal expansion; Python's
    ``format`` rounds the exact binary value half to even. They differ only on exact ties of
    the binary value (x.x5 exactly representable, e.g. 0.25), so ties are rounded up here.
    """
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Infinity" if x > 0 else "-Infinity"
    from decimal import ROUND_HALF_UP, Decimal
    q = Decimal(x).quantize(Decimal("0.1"), rounding=ROUND_HALF_UP)
    s = format(q, "f")
    return "0.0" if s == "-0.0" else s
