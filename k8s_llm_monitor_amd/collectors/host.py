"""Host CPU / memory / disk collector.

Reference: ``getHostMetrics`` (monitor_server.js:66-81). Same output shape:

    {"cpu": {"load_1min": f, "percent": "5.3"},
     "memory": {"total": int, "used": int, "percent": "4.1"},
     "disk": {"total": int, "used": int, "percent": "45.3"}}

with string-typed ``percent`` (JS ``toFixed(1)``) and the failure shape
``{"cpu": {}, "memory": {}, "disk": {}}`` (:80).

Differences, all additive or configurable:
  * no ``df`` subprocess: ``os.statvfs('/')`` gives the same two numbers df prints in
    columns 2 and 3 (total = f_blocks*f_frsize, used = (f_blocks-f_bfree)*f_frsize);
  * the divisor of the load-based CPU percent is the real core count, not the literal 8
    (:76); ``compat.cpu_cores = 8`` restores the reference number;
  * true utilisation from ``/proc/stat`` deltas is reported beside it as
    ``cpu.util_percent`` (and becomes ``cpu.percent`` with ``compat.cpu_percent = util``).
"""
from __future__ import annotations

import os
import re
import threading
import time
from typing import Callable, Dict, Optional, Tuple

from ..utils.timefmt import fixed1

_MEMTOTAL = re.compile(r"MemTotal:\s+(\d+)")
_MEMAVAIL = re.compile(r"MemAvailable:\s+(\d+)")

EMPTY = {"cpu": {}, "memory": {}, "disk": {}}


def parse_loadavg(text: str) -> float:
    return float(text.strip().split(" ")[0])


def parse_meminfo(text: str) -> Tuple[int, int]:
    """(total_bytes, available_bytes). Raises on a file without both fields, like the
    reference's ``match(...)[1]`` on null (:70-71)."""
    t, a = _MEMTOTAL.search(text), _MEMAVAIL.search(text)
    if not t or not a:
        raise ValueError("MemTotal/MemAvailable not found")
    return int(t.group(1)) * 1024, int(a.group(1)) * 1024


def parse_df(text: str) -> Tuple[int, int]:
    """(total, used) from ``df -B1 /`` output: line 2, columns 2 and 3 (:73, :78)."""
    parts = re.split(r"\s+", text.strip().split("\n")[1])
    return int(parts[1]), int(parts[2])


def parse_proc_stat_cpu(text: str) -> Tuple[int, int]:
    """(busy_jiffies, total_jiffies) of the aggregate ``cpu`` line."""
    for line in text.splitlines():
        if line.startswith("cpu "):
            f = [int(x) for x in line.split()[1:]]
            idle = f[3] + (f[4] if len(f) > 4 else 0)  # idle + iowait
            total = sum(f[:8]) if len(f) >= 8 else sum(f)  # guest time is already in user
            return total - idle, total
    raise ValueError("no aggregate cpu line")


def _read(path: str) -> str:
    with open(path, "r", encoding="utf-8") as f:
        return f.read()


class HostCollector:
    """Stateful only for the /proc/stat delta; everything else is read per call."""

    def __init__(self, cores: int = 0, cpu_percent: str = "load", root: str = "/",
                 read: Callable[[str], str] = _read,
                 statvfs: Callable[[str], os.statvfs_result] = os.statvfs):
        self.cores = cores if cores > 0 else (os.cpu_count() or 1)
        self.cpu_percent = cpu_percent
        self.root = root
        self._read = read
        self._statvfs = statvfs
        self._prev: Optional[Tuple[int, int]] = None
        self._last_util: Optional[float] = None
        self._lock = threading.Lock()
        self.last_duration_s = 0.0

    def _util(self) -> Optional[float]:
        try:
            busy, total = parse_proc_stat_cpu(self._read("/proc/stat"))
        except (OSError, ValueError, IndexError):
            return self._last_util
        with self._lock:
            prev, self._prev = self._prev, (busy, total)
            if prev is not None and total > prev[1]:
                self._last_util = 100.0 * (busy - prev[0]) / (total - prev[1])
            return self._last_util

    def collect(self) -> Dict[str, dict]:
        t0 = time.perf_counter()
        try:
            load1 = parse_loadavg(self._read("/proc/loadavg"))
            mem_total, mem_avail = parse_meminfo(self._read("/proc/meminfo"))
            st = self._statvfs(self.root)
            disk_total = st.f_blocks * st.f_frsize
            disk_used = (st.f_blocks - st.f_bfree) * st.f_frsize
            util = self._util()
            load_pct = load1 / self.cores * 100.0
            pct = util if (self.cpu_percent == "util" and util is not None) else load_pct
            mem_used = mem_total - mem_avail
            out = {
                "cpu": {"load_1min": load1, "percent": fixed1(pct), "cores": self.cores,
                        "load_percent": fixed1(load_pct)},
                "memory": {"total": mem_total, "used": mem_used,
                           "percent": fixed1(mem_used / mem_total * 100.0)},
                "disk": {"total": disk_total, "used": disk_used,
                         "percent": fixed1(disk_used / disk_total * 100.0)},
            }
            if util is not None:
                out["cpu"]["util_percent"] = fixed1(util)
            return out
        except Exception:  # same contract as the reference's catch-all (:80)
            return {"cpu": {}, "memory": {}, "disk": {}}
        finally:
            self.last_duration_s = time.perf_counter() - t0
