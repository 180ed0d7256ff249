"""Host collector vs. getHostMetrics (monitor_server.js:66-81) on captured /proc text."""
import os
from types import SimpleNamespace

import pytest

from k8s_llm_monitor_amd.collectors.host import (HostCollector, parse_df, parse_loadavg, parse_meminfo,
                                                 parse_proc_stat_cpu)

LOADAVG = "0.42 0.30 0.25 2/1234 56789\n"
MEMINFO = """MemTotal:       65842536 kB
MemFree:        10000000 kB
MemAvailable:   63142992 kB
Buffers:          123456 kB
"""
DF = """Filesystem        1B-blocks        Used    Available Use% Mounted on
/dev/root      105089261568 47605452800 57467031552  46% /
"""
STAT1 = "cpu  100 0 100 800 0 0 0 0 0 0\ncpu0 50 0 50 400 0 0 0 0 0 0\n"
STAT2 = "cpu  150 0 150 900 0 0 0 0 0 0\ncpu0 75 0 75 450 0 0 0 0 0 0\n"


def test_parsers():
    assert parse_loadavg(LOADAVG) == 0.42
    assert parse_meminfo(MEMINFO) == (65842536 * 1024, 63142992 * 1024)
    assert parse_df(DF) == (105089261568, 47605452800)
    assert parse_proc_stat_cpu(STAT1) == (200, 1000)
    with pytest.raises(ValueError):
        parse_meminfo("MemTotal: 5 kB\n")


def make(files, cores=8, mode="load", total=1000, free=547):
    def read(p):
        v = files[p]
        if isinstance(v, list):
            return v.pop(0) if len(v) > 1 else v[0]
        return v
    st = SimpleNamespace(f_blocks=total, f_bfree=free, f_frsize=4096)
    return HostCollector(cores=cores, cpu_percent=mode, read=read, statvfs=lambda p: st)


def test_reference_shape_and_numbers():
    h = make({"/proc/loadavg": LOADAVG, "/proc/meminfo": MEMINFO, "/proc/stat": STAT1}).collect()
    # load 0.42 on the reference's 8 cores -> "5.3" (0.42/8*100 = 5.25 -> toFixed(1), monitor_server.js:76)
    assert h["cpu"]["load_1min"] == 0.42 and h["cpu"]["percent"] == "5.3"
    assert h["memory"]["total"] == 65842536 * 1024
    assert h["memory"]["used"] == (65842536 - 63142992) * 1024
    assert h["memory"]["percent"] == "4.1"
    assert h["disk"] == {"total": 1000 * 4096, "used": 453 * 4096, "percent": "45.3"}
    for sec in ("cpu", "memory", "disk"):
        assert isinstance(h[sec]["percent"], str)            # string-typed like toFixed(1)


def test_core_count_is_configurable_not_hard_coded():
    files = {"/proc/loadavg": "16.00 1 1 1/1 1\n", "/proc/meminfo": MEMINFO, "/proc/stat": STAT1}
    assert make(dict(files), cores=8).collect()["cpu"]["percent"] == "200.0"     # load can exceed 100
    assert make(dict(files), cores=64).collect()["cpu"]["percent"] == "25.0"
    assert HostCollector(cores=0).cores == (os.cpu_count() or 1)


def test_util_mode_uses_proc_stat_deltas():
    c = make({"/proc/loadavg": LOADAVG, "/proc/meminfo": MEMINFO, "/proc/stat": [STAT1, STAT2]}, mode="util")
    first = c.collect()
    assert first["cpu"]["percent"] == "5.3"          # no delta yet: falls back to load
    second = c.collect()
    # busy 200 -> 300, total 1000 -> 1200: 50 %
    assert second["cpu"]["util_percent"] == "50.0" and second["cpu"]["percent"] == "50.0"


def test_failure_shape():
    def boom(p):
        raise OSError(p)
    h = HostCollector(read=boom).collect()
    assert h == {"cpu": {}, "memory": {}, "disk": {}}     # monitor_server.js:80


def test_real_proc_when_present():
    if not os.path.exists("/proc/meminfo"):
        pytest.skip("no /proc")
    h = HostCollector().collect()
    assert 0.0 <= float(h["memory"]["percent"]) <= 100.0 and h["disk"]["total"] > 0
