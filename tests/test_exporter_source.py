"""AMD device-metrics-exporter source: the same scenarios through exposition text."""
import pytest

from k8s_llm_monitor_amd.collectors.exporter import DEFAULT_NAMES, ExporterGpuReader
from k8s_llm_monitor_amd.collectors.gpu import GpuSource, open_gpu_source
from k8s_llm_monitor_amd.config import load_config
from k8s_llm_monitor_amd.fakes.exporter import FakeExporter
from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader, GpuState
from k8s_llm_monitor_amd.server.alerts import check_alerts


def test_exporter_samples_match_the_direct_reader(clock):
    scen = lambda t, i: GpuState(hbm_frac=0.5 + 0.05 * i, gfx=40.0 + i, umc=20.0, power_w=800.0, temp=70.0 + i, xgmi_gbps=50.0, ecc_ce=i)
    fx = FakeExporter(FakeGpuReader(4, scen, clock=clock))
    src = GpuSource(ExporterGpuReader("http://exporter:5000", fetch=fx.fetch, clock=clock))
    direct = GpuSource(FakeGpuReader(4, scen, clock=clock))
    src.sample(); direct.sample()
    clock.advance(5.0)
    a, b = src.sample(), direct.sample()
    assert src.last_error is None and len(a) == 4
    for x, y in zip(a, b):
        for k in ("index", "utilization", "memoryUsed", "memoryTotal", "temperature", "uuid", "bdf"):
            assert x[k] == y[k], k
        assert x["powerW"] == y["powerW"] and x["powerCapW"] == 1400.0 and x["clockMemMHz"] == 2000
        assert x["powerAvgW"] == pytest.approx(800.0, rel=1e-3)                # from the energy counter delta
        assert x["xgmi"]["rxGBps"] == pytest.approx(y["xgmi"]["rxGBps"], rel=1e-3) and len(x["xgmi"]["links"]) == 7
        assert x["ecc"]["correctable"] == y["ecc"]["correctable"]
        assert x["gfxActivityAvg"] is None           # the exporter has no activity accumulator: the gauge is used
    assert a[0]["name"].startswith("AMD Instinct")


def test_hbm_alert_through_the_exporter_path(clock):
    fx = FakeExporter(FakeGpuReader(8, "hbm_ramp", clock=clock))
    src = GpuSource(ExporterGpuReader("http://exporter:5000/metrics", fetch=fx.fetch, clock=clock))
    clock.advance(31)
    alerts, _ = check_alerts({"cpu": {}, "memory": {}, "disk": {}}, src.sample(), [])
    assert [x["key"] for x in alerts["critical"]] == ["gpu/0/hbm"]


def test_metric_names_are_configuration(clock):
    cfg = load_config(env={"KLM_SOURCES_GPU": "exporter", "KLM_EXPORTER_URL": "http://node7:5000",
                           "KLM_EXPORTER_M_GFX_ACTIVITY": "amd_gpu_gfx_busy_percent"})
    assert cfg.exporter.names() == {"gfx_activity": "amd_gpu_gfx_busy_percent"}
    src = open_gpu_source("exporter", cfg)
    assert src.kind == "exporter" and src.reader.url == "http://node7:5000/metrics"
    assert src.reader.names["gfx_activity"] == "amd_gpu_gfx_busy_percent" and src.reader.names["temp_hotspot"] == DEFAULT_NAMES["temp_hotspot"]
    fx = FakeExporter(FakeGpuReader(1, clock=clock), names={"gfx_activity": "amd_gpu_gfx_busy_percent"})
    src.reader._fetch = fx.fetch
    assert src.sample()[0]["utilization"] > 0
    assert src.sample() and "DCGM" not in fx.fetch()


def test_unreachable_exporter_gives_empty_list():
    src = GpuSource(ExporterGpuReader("http://127.0.0.1:9", timeout=0.2))
    assert src.sample() == [] and src.last_error
