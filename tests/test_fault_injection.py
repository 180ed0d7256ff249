"""Scripted fault timelines through the whole stack (fake readers -> sampler -> alert engine
-> HTTP), and the threshold-crossing -> alert-visible latency (BASELINE.md target < 10 s).
BASELINE.json config 5; SURVEY.md section 4 "fault injection"."""
import json
import time
import urllib.request

import pytest

from k8s_llm_monitor_amd.collectors.gpu import GpuSource
from k8s_llm_monitor_amd.collectors.vllm import VllmDirectSource
from k8s_llm_monitor_amd.config import demo_config
from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader, GpuState
from k8s_llm_monitor_amd.fakes.kube import FakeKube
from k8s_llm_monitor_amd.fakes.vllm import FakeVllm
from k8s_llm_monitor_amd.server.http import MonitorServer
from k8s_llm_monitor_amd.server.sampler import Monitor, PeriodicTask


def stepped_monitor(clock, gpu="steady", kube="steady", vllm="steady", n_gpus=8, **cfg_over):
    cfg = demo_config(**cfg_over)
    fv = FakeVllm(vllm, clock=clock)
    mon = Monitor(cfg, gpu=GpuSource(FakeGpuReader(n_gpus, gpu, clock=clock)), kube=FakeKube(kube, clock=clock),
                  vllm=VllmDirectSource(["fake://v"], fetch=fv.fetch, clock=clock), clock=clock)
    return mon


def keys(mon, tier):
    return sorted(a["key"] for a in mon.alerts_public()[tier])


def test_hbm_ramp_walks_minor_serious_critical(clock):
    mon = stepped_monitor(clock, gpu="hbm_ramp")
    seen = []
    for _ in range(4):
        mon.tick_all()
        seen.append([t for t in ("minor", "serious", "critical") if "gpu/0/hbm" in keys(mon, t)])
        clock.advance(10)
    assert seen == [[], ["minor"], ["serious"], ["critical"]]       # 69 / 71 / 86 / 96 % of 288 GB


def test_thermal_ramp_and_throttle(clock):
    mon = stepped_monitor(clock, gpu="thermal")
    mon.tick_all()
    assert not any(k.startswith("gpu/1") for t in ("serious", "critical") for k in keys(mon, t))
    clock.advance(45)           # 70 + 22.5 = 92.5 C > 90
    mon.tick_all()
    assert "gpu/1/temp" in keys(mon, "serious")
    clock.advance(30)           # 107.5 C > 100 and thermal throttling
    mon.tick_all()
    clock.advance(5)
    mon.tick_all()
    assert "gpu/1/temp" in keys(mon, "critical") and "gpu/1/throttle_thm" in keys(mon, "serious")


def test_pod_lifecycle_alert_sequence(clock):
    mon = stepped_monitor(clock, kube="pod_lifecycle")
    mon.tick_all()
    assert keys(mon, "serious") == [] and keys(mon, "critical") == []      # 11 Running pods, first sight: silent
    clock.advance(6)
    mon.tick_all()
    assert keys(mon, "serious") == ["pod/llm/vllm-llama3-8b-canary-0/pending"]
    clock.advance(10)
    mon.tick_all()
    assert keys(mon, "serious") == ["pod/llm/vllm-llama3-8b-canary-0/recovered"]
    clock.advance(15)
    mon.tick_all()
    assert "pod/llm/vllm-llama3-8b-canary-0/restart" in keys(mon, "serious")
    assert "pod/llm/vllm-llama3-8b-canary-0/recovered" in keys(mon, "serious")     # edge alert still held (30 s)
    clock.advance(10)
    mon.tick_all()
    assert keys(mon, "critical") == ["pod/llm/vllm-llama3-8b-canary-0/failed"]
    clock.advance(31)
    mon.tick_all()
    assert keys(mon, "serious") == []                                            # held edges expired


def test_ecc_xgmi_gpu_lost_and_kv_pressure(clock):
    mon = stepped_monitor(clock, gpu="ecc")
    mon.tick_all(); clock.advance(16); mon.tick_all()
    assert "gpu/2/ecc_ue" in keys(mon, "critical") and "gpu/2/ecc_ce" in keys(mon, "minor")
    # retired HBM pages (RAS): GPU 5 came up with 3 (history: no alert) and gained a fourth (minor, edge); a page still waiting to be
    # retired is serious for as long as it waits
    assert "gpu/5/bad_pages" in keys(mon, "minor") and "gpu/2/bad_pages" not in keys(mon, "serious") and "gpu/0/bad_pages" not in keys(mon, "minor")
    clock.advance(6); mon.tick_all()
    assert "gpu/2/bad_pages" in keys(mon, "serious")
    g = {x["index"]: x for x in mon.get("gpus")}
    assert g[5]["badPages"] == {"retired": 4, "pending": 0, "unreservable": 0} and g[2]["badPages"]["pending"] == 1 and g[0]["badPages"]["retired"] == 0
    clock.advance(40); mon.tick_all()
    assert "gpu/5/bad_pages" not in keys(mon, "minor") and "gpu/2/bad_pages" in keys(mon, "serious")      # the edge expired, the level did not
    mon = stepped_monitor(clock, gpu="xgmi_down")
    mon.tick_all(); clock.advance(11); mon.tick_all()
    assert any(k.startswith("gpu/3/xgmi/") for k in keys(mon, "critical"))
    mon = stepped_monitor(clock, gpu="gpu_lost")
    mon.tick_all(); clock.advance(11); mon.tick_all()
    assert "gpu/lost" in keys(mon, "critical")
    mon = stepped_monitor(clock, vllm="kv_pressure")
    mon.tick_all(); clock.advance(12); mon.tick_all()
    assert "vllm/kv" in keys(mon, "minor") and "vllm/waiting" in keys(mon, "minor")
    clock.advance(20); mon.tick_all(); clock.advance(1); mon.tick_all()
    assert "vllm/kv" in keys(mon, "critical") and "vllm/waiting" in keys(mon, "critical")


def test_collector_failure_is_contained(clock):
    mon = stepped_monitor(clock)
    mon.kube.fail = True
    mon.tick_all()
    assert mon.get("pods") == [] and mon.stats()["errors"]["kube"]
    assert len(mon.get("gpus")) == 8                                   # the other sources keep working


def test_crossing_to_visible_latency_over_http():
    """Real threads, real HTTP, default-like periods scaled 10x down: the HBM threshold is
    crossed at a known instant and /api/alerts is polled like the page does."""
    t_cross = {}

    def scen(t, i):
        s = GpuState(hbm_frac=0.5)
        if i == 0 and "at" in t_cross and time.monotonic() >= t_cross["at"]:
            s.hbm_frac = 0.97
        return s

    cfg = demo_config(periods={"realtime": 0.5, "pods": 1.0, "vllm": 0.5})
    mon = Monitor(cfg, gpu=GpuSource(FakeGpuReader(8, scen))).start()
    srv = MonitorServer(mon, cfg).start()
    try:
        time.sleep(0.7)
        t_cross["at"] = time.monotonic()
        t0 = time.monotonic()
        latency = None
        while time.monotonic() - t0 < 5.0:
            a = json.loads(urllib.request.urlopen(srv.url + "/api/alerts", timeout=2).read())
            if any(x["key"] == "gpu/0/hbm" for x in a["critical"]):
                latency = time.monotonic() - t0
                break
            time.sleep(0.05)
        assert latency is not None and latency < 0.5 + 0.3       # at most one sampling period + slack
    finally:
        srv.stop()
        mon.stop()


def test_periodic_task_grid_and_error_containment():
    n = {"c": 0}

    def fn():
        n["c"] += 1
        if n["c"] == 2:
            raise RuntimeError("boom")

    t = PeriodicTask("t", fn, 0.05).start()
    time.sleep(0.38)
    t.stop()
    st = t.stats()
    assert 5 <= st["calls"] <= 9 and st["errors"] == 1 and "boom" in st["last_error"]
    assert st["duty_percent"] < 5.0
