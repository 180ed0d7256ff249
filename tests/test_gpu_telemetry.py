"""Counter arithmetic, the GPU normaliser and the scripted fake GPUs (CPU only).

The raw-reading fixture below is the amdsmi output captured on a real MI355X
(profiles/r1_amdsmi_probe_mi355x.json), so the field mapping is checked against what the
hardware actually reports."""
import json
import os

import pytest

from k8s_llm_monitor_amd.collectors.counters import RateTracker, counter_delta, share
from k8s_llm_monitor_amd.collectors.gpu import GpuNormalizer, GpuReader, GpuSource, metrics_to_raw, open_gpu_source
from k8s_llm_monitor_amd.fakes.gpu import HBM_TOTAL_MIB, SCENARIOS, FakeGpuReader, GpuState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counter_delta_wrap_and_reset():
    assert counter_delta(10, 15) == 5
    assert counter_delta((1 << 64) - 5, 10) == 15           # one wrap
    assert counter_delta((1 << 32) - 5, 10, bits=32) == 15
    assert counter_delta(1_000_000, 5) is None              # went backwards: a reset, no delta
    assert counter_delta(None, 5) is None


def test_rate_tracker():
    r = RateTracker()
    assert r.update("k", 0.0, 100) is None
    assert r.update("k", 2.0, 300) == 100.0
    assert r.update("k", 2.0, 400) is None                  # dt == 0
    assert r.update("k", 3.0, 0) is None                    # reset
    assert r.update("k", 4.0, 50) == 50.0
    assert share(5, 10) == 50.0 and share(5, 0) is None and share(None, 3) is None


def test_real_mi355x_metrics_mapping():
    with open(os.path.join(ROOT, "profiles", "r1_amdsmi_probe_mi355x.json")) as f:
        probe = json.load(f)["gpu0"]
    raw = metrics_to_raw(probe["metrics_info"])
    assert raw["temp_hotspot"] == 46 and raw["temp_mem"] == 34 and "temp_edge" not in raw     # edge is N/A on MI355X
    assert raw["power_w"] == 308 and raw["accumulation_counter"] == 23735538
    assert len(raw["xgmi_read_kib"]) == 8 and raw["xgmi_link_status"][0] == "N/A"
    raw.update(index=0, t=1.0, vram_used_mib=probe["vram_usage"]["vram_used"], vram_total_mib=probe["vram_usage"]["vram_total"],
               name=probe["asic"]["market_name"], power_cap_w=1400.0)
    s = GpuNormalizer().normalize(raw)
    assert s["name"] == "AMD Instinct MI355 OAM" and s["memoryTotal"] == 294896 and s["memoryUsed"] == 284
    assert s["temperature"] == 46 and s["utilization"] == 7 and s["powerCapW"] == 1400.0
    assert len(s["xgmi"]["links"]) == 7 and all(l["state"] == "up" for l in s["xgmi"]["links"])   # own slot skipped
    assert s["xgmi"]["rxGBps"] is None                      # no rate from a single reading


def test_normaliser_uses_accumulator_deltas(clock):
    fr = FakeGpuReader(1, scenario=lambda t, i: GpuState(gfx=37.0, umc=12.0, power_w=900.0, xgmi_gbps=100.0, ppt_throttle=0.25), clock=clock)
    src = GpuSource(fr)
    first = src.sample()[0]
    assert first["gfxActivityAvg"] is None and first["utilization"] == 37
    clock.advance(5.0)
    s = src.sample()[0]
    assert s["gfxActivityAvg"] == pytest.approx(37.0, abs=0.01) and s["memActivityAvg"] == pytest.approx(12.0, abs=0.01)
    assert s["powerAvgW"] == pytest.approx(900.0, rel=1e-3)
    assert s["xgmi"]["rxGBps"] == pytest.approx(700.0, rel=1e-3) and s["xgmi"]["links"][0]["txGBps"] == pytest.approx(100.0, rel=1e-3)
    assert s["throttle"]["active"] and s["throttle"]["residencyPercent"]["ppt"] == pytest.approx(25.0, abs=0.01)
    assert s["hbmTotalBytes"] == HBM_TOTAL_MIB << 20 and 0 < s["hbmPercent"] < 100


def test_rates_follow_the_tables_own_clock_and_hold_while_it_is_stale():
    """The driver refreshes the metrics table about once a second (measured on MI355X). A read 0.61 s after the
    previous one that sees a table 1.0 s newer must give delta energy over 1.0 s, and a second read of the same
    table must repeat the period figures instead of reporting zero power and no activity."""
    n = GpuNormalizer()
    res_uj = 15.3
    counts_per_s_at_1390w = 1390.0 / (res_uj * 1e-6)

    def raw(host_t, table_s):
        return {"index": 0, "t": host_t, "vram_used_mib": 1000, "vram_total_mib": 294896, "energy_resolution_uj": res_uj,
                "firmware_timestamp": int(5_000_000_000 + table_s * 1e8), "accumulation_counter": int(1000 * table_s),
                "gfx_activity_acc": int(1000 * table_s * 100), "mem_activity_acc": int(1000 * table_s * 40),
                "energy_acc": int(10_000_000 + counts_per_s_at_1390w * table_s), "power_w": 1386,
                "xgmi_read_kib": [int(table_s * 50e9 / 1024)] * 2, "xgmi_write_kib": [int(table_s * 25e9 / 1024)] * 2,
                "xgmi_link_status": ["U", "U"]}

    assert n.normalize(raw(0.00, 0.0))["powerAvgW"] is None
    s = n.normalize(raw(0.61, 1.0))                          # host says 0.61 s, the table moved 1.0 s
    assert s["powerAvgW"] == pytest.approx(1390.0, rel=1e-4) and s["gfxActivityAvg"] == pytest.approx(100.0)
    assert s["xgmi"]["links"][0]["rxGBps"] == pytest.approx(50.0, rel=1e-3)
    assert s["metricsTableTimeS"] == pytest.approx(51.0) and s["energyJ"] == pytest.approx((10_000_000 + counts_per_s_at_1390w) * res_uj * 1e-6)
    for host_t in (0.66, 0.71):                              # the same table again: figures held, not zeroed
        s = n.normalize(raw(host_t, 1.0))
        assert s["powerAvgW"] == pytest.approx(1390.0, rel=1e-4) and s["gfxActivityAvg"] == pytest.approx(100.0)
        assert s["memActivityAvg"] == pytest.approx(40.0) and s["xgmi"]["links"][1]["txGBps"] == pytest.approx(25.0, rel=1e-3)
        assert s["utilization"] == 100
    s = n.normalize(raw(1.62, 2.0))
    assert s["powerAvgW"] == pytest.approx(1390.0, rel=1e-4)
    # a reader without the firmware clock (the fakes, an older library) still gets host-time rates
    m = GpuNormalizer()
    a, b = raw(0.0, 0.0), raw(2.0, 2.0)
    a.pop("firmware_timestamp"), b.pop("firmware_timestamp")
    m.normalize(a)
    s = m.normalize(b)
    assert s["powerAvgW"] == pytest.approx(1390.0, rel=1e-4) and s["metricsTableTimeS"] is None


def test_reference_keys_and_types(clock):
    s = GpuSource(FakeGpuReader(8, clock=clock)).sample()
    assert len(s) == 8 and [g["index"] for g in s] == list(range(8))
    for g in s:
        for k in ("utilization", "memoryUsed", "memoryTotal", "temperature"):
            assert isinstance(g[k], int)                    # parseInt in the reference (:90)
        assert isinstance(g["name"], str)
        assert g["memoryTotal"] == HBM_TOTAL_MIB            # MiB, like nvidia-smi nounits


def test_source_never_raises():
    class Boom(GpuReader):
        kind = "boom"

        def read(self):
            raise RuntimeError("driver gone")
    src = GpuSource(Boom())
    assert src.sample() == [] and "driver gone" in src.last_error      # [] like monitor_server.js:94
    assert open_gpu_source("none").sample() == []
    assert open_gpu_source("auto").kind in ("amdsmi", "none")
    with pytest.raises(ValueError):
        open_gpu_source("nvml")


@pytest.mark.parametrize("name", sorted(k for k in SCENARIOS if k))
def test_every_scenario_runs(name, clock):
    src = GpuSource(FakeGpuReader(8, name, clock=clock))
    for _ in range(12):
        clock.advance(5)
        assert len(src.sample()) >= 7


def test_hbm_ramp_crosses_the_three_tiers(clock):
    src = GpuSource(FakeGpuReader(8, "hbm_ramp", clock=clock))
    seen = []
    for _ in range(4):
        seen.append(round(src.sample()[0]["hbmPercent"]))
        clock.advance(10)
    assert seen == [69, 71, 86, 96]


def test_xgmi_down_and_gpu_lost(clock):
    src = GpuSource(FakeGpuReader(8, "xgmi_down", clock=clock))
    clock.advance(11)
    g3 = src.sample()[3]
    assert [l["state"] for l in g3["xgmi"]["links"]].count("down") == 1
    src = GpuSource(FakeGpuReader(8, "gpu_lost", clock=clock))
    assert len(src.sample()) == 8
    clock.advance(11)
    assert len(src.sample()) == 7


def test_unknown_scenario():
    with pytest.raises(ValueError):
        FakeGpuReader(1, "nope")


# ------------------------------------------------------------------ per-XCD activity and PCIe health
def test_xcd_activity_and_pcie_errors_from_accumulators():
    from k8s_llm_monitor_amd.collectors.gpu import GpuNormalizer, metrics_to_raw
    m0 = {"accumulation_counter": 1000, "gfx_activity_acc": 50000,
          "xcp_stats.gfx_busy_acc": [[100, 200, 300, 400, 500, 600, 700, 800]] + [["N/A"] * 8] * 7,
          "xcp_stats.gfx_busy_inst": [[10, 20, 30, 40, 50, 60, 70, 0]] + [["N/A"] * 8] * 7,
          "pcie_replay_count_acc": 4, "pcie_l0_to_recov_count_acc": 1, "pcie_nak_sent_count_acc": 0, "pcie_nak_rcvd_count_acc": 0,
          "pcie_bandwidth_inst": 800, "pcie_link_speed": 320, "pcie_link_width": 16}
    r0 = metrics_to_raw(m0)
    assert r0["xcd_busy_acc"] == [100, 200, 300, 400, 500, 600, 700, 800] and r0["xcd_busy_inst"][-1] == 0
    m1 = dict(m0, accumulation_counter=2000, gfx_activity_acc=100000, pcie_replay_count_acc=9)
    m1["xcp_stats.gfx_busy_acc"] = [[100 + 1000 * p for p in (100, 100, 100, 100, 50, 50, 0, 25)][i] + 100 * i for i in range(8)]
    m1["xcp_stats.gfx_busy_acc"] = [m1["xcp_stats.gfx_busy_acc"]] + [["N/A"] * 8] * 7
    n = GpuNormalizer()
    g0 = n.normalize(dict(r0, index=0, t=0.0, vram_total_mib=1000, vram_used_mib=10))
    assert g0["xcdActivityAvg"] is None and g0["xcdActivity"][:2] == [10, 20]
    assert g0["pcie"]["errors"]["replay"] == {"count": 4, "delta": None}
    g1 = n.normalize(dict(metrics_to_raw(m1), index=0, t=1.0, vram_total_mib=1000, vram_used_mib=10))
    assert g1["xcdActivityAvg"] == [100.0, 100.0, 100.0, 100.0, 50.0, 50.0, 0.0, 25.0]
    assert g1["pcie"]["errors"]["replay"] == {"count": 9, "delta": 5} and g1["pcie"]["errors"]["l0Recovery"]["delta"] == 0
    assert g1["pcie"]["linkSpeedGTs"] == 32.0 and g1["pcie"]["linkWidth"] == 16 and g1["pcie"]["GBps"] == pytest.approx(0.1)
    # CPX mode: eight partitions of one XCD each flatten to the same eight entries
    cpx = metrics_to_raw({"xcp_stats.gfx_busy_acc": [[7] + ["N/A"] * 7 for _ in range(8)]})
    assert cpx["xcd_busy_acc"] == [7] * 8


def test_fake_pcie_scenario_raises_the_replay_alert_and_shows_the_idle_xcd():
    from k8s_llm_monitor_amd.collectors.gpu import GpuSource
    from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader
    from k8s_llm_monitor_amd.server.alerts import AlertState, check_alerts
    clock = {"t": 0.0}
    src = GpuSource(FakeGpuReader(n_gpus=8, scenario="pcie", clock=lambda: clock["t"]))
    src.sample()
    st = AlertState()
    seen = []
    for step in range(4):
        clock["t"] += 5.0
        gpus = src.sample()
        alerts, st = check_alerts({"cpu": {"usage": 1}, "memory": {"percent": 1}, "disk": {"percent": 1}}, gpus, [], state=st)
        seen.append([a["key"] for tier in ("minor", "serious", "critical") for a in alerts[tier] if a.get("key", "").endswith("/pcie")])
    assert any(k == ["gpu/2/pcie"] for k in seen), seen
    g5 = [g for g in gpus if g["index"] == 5][0]
    assert g5["xcdActivityAvg"][5] == 0.0 and min(v for i, v in enumerate(g5["xcdActivityAvg"]) if i != 5) > 10.0
    assert "XCD" in open(os.path.join(ROOT, "k8s_llm_monitor_amd", "web", "monitor.html"), encoding="utf-8").read()


def test_partition_modes_travel_from_the_readers_to_the_api_the_exporter_and_the_page(clock):
    """What one "GPU" of the list is a slice of (compute partition SPX / CPX ..., memory partition NPS1 ...): from amdsmi
    (Python reader: asked once where unsupported; C++ sampler: its selftest line) through the normaliser to /api/gpu/metrics,
    /metrics and the dashboard's GPU table."""
    import json
    import subprocess
    from k8s_llm_monitor_amd.build import artifact
    from k8s_llm_monitor_amd.collectors.gpu import GpuNormalizer, _partition_name
    from k8s_llm_monitor_amd.server import exporter
    g = GpuSource(FakeGpuReader(2, clock=clock)).sample()[0]
    assert g["partition"] == {"compute": "SPX", "memory": "NPS1"}
    text = exporter.render({"gpus": [g], "host": {}}, {})
    assert 'klm_gpu_partition_info{' in text and 'compute="SPX"' in text and 'memory="NPS1"' in text
    # the forms amdsmi answers in: plain strings, enum members, "N/A"
    class E:
        name = "AMDSMI_COMPUTE_PARTITION_CPX"
    assert _partition_name("SPX") == "SPX" and _partition_name(E()) == "CPX" and _partition_name("N/A") is None and _partition_name(None) is None
    # a driver that does not say leaves both None and no info series
    raw = dict(FakeGpuReader(1, clock=clock).read()[0])
    raw.pop("compute_partition"); raw.pop("memory_partition")
    g2 = GpuNormalizer().normalize(raw)
    assert g2["partition"] == {"compute": None, "memory": None} and "klm_gpu_partition_info" not in exporter.render({"gpus": [g2], "host": {}}, {})
    exe = artifact("klm_amdsmi_sampler")
    if exe.exists():
        r = subprocess.run([str(exe), "--selftest"], capture_output=True, text=True, timeout=20)
        g0, g1 = json.loads(r.stdout)["gpus"]
        assert g0["compute_partition"] == "SPX" and g0["memory_partition"] == "NPS1" and "compute_partition" not in g1
        assert (g0["bad_pages_retired"], g0["bad_pages_pending"], g0["bad_pages_unreservable"]) == (2, 1, 0) and "bad_pages_retired" not in g1
        assert GpuNormalizer().normalize(g0)["badPages"] == {"retired": 2, "pending": 1, "unreservable": 0}
        assert GpuNormalizer().normalize(g0)["partition"] == {"compute": "SPX", "memory": "NPS1"}
    page = open(os.path.join(os.path.dirname(exporter.__file__), "..", "web", "monitor.html"), encoding="utf-8").read()
    assert "g.partition" in page and "compute partition" in page


def test_bad_page_records_are_counted_by_status_and_exported(clock):
    from k8s_llm_monitor_amd.collectors.gpu import GpuNormalizer, bad_pages_to_raw
    from k8s_llm_monitor_amd.server import exporter

    class St:
        def __init__(self, name):
            self.name = name
    recs = [{"status": St("AMDSMI_MEM_PAGE_STATUS_RESERVED"), "page_address": 1}, {"status": St("AMDSMI_MEM_PAGE_STATUS_PENDING")},
            {"status": "RESERVED"}, {"status": St("AMDSMI_MEM_PAGE_STATUS_UNRESERVABLE")}]
    assert bad_pages_to_raw(recs) == {"bad_pages_retired": 2, "bad_pages_pending": 1, "bad_pages_unreservable": 1}
    assert bad_pages_to_raw([]) == bad_pages_to_raw("No bad pages found.") == {"bad_pages_retired": 0, "bad_pages_pending": 0, "bad_pages_unreservable": 0}
    raw = dict(FakeGpuReader(1, clock=clock).read()[0], **bad_pages_to_raw(recs))
    g = GpuNormalizer().normalize(raw)
    assert g["badPages"] == {"retired": 2, "pending": 1, "unreservable": 1}
    text = exporter.render({"gpus": [g], "host": {}}, {})
    assert 'klm_gpu_bad_pages{' in text and 'status="pending"} 1' in text and 'status="retired"} 2' in text
    raw.pop("bad_pages_retired")                    # a driver that does not say: no field, no series
    g = GpuNormalizer().normalize(raw)
    assert g["badPages"] is None and "klm_gpu_bad_pages" not in exporter.render({"gpus": [g], "host": {}}, {})
