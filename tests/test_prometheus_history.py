"""Prometheus client (monitor_server.js:14-63) and /api/history shaping (:117-154) against the
HTTP stub, plus the local ring-buffer history."""
import time

import pytest

from k8s_llm_monitor_amd.collectors import history as H
from k8s_llm_monitor_amd.collectors.prometheus import PrometheusClient
from k8s_llm_monitor_amd.config import Config
from k8s_llm_monitor_amd.fakes.prometheus import FakePrometheus
from k8s_llm_monitor_amd.utils.ringbuffer import HistoryStore, SeriesRing
from k8s_llm_monitor_amd.utils.timefmt import hhmm


@pytest.fixture
def prom():
    with FakePrometheus() as p:
        yield p


def test_instant_query(prom):
    c = PrometheusClient(prom.url)
    assert c.query("sum(vllm:num_requests_running)") == 48.0
    assert c.query("no_such_metric") is None              # empty result -> null (:31-35)
    assert c.reachable()


def test_range_query_61_points_ms_timestamps(prom):
    c = PrometheusClient(prom.url, now=lambda: 1_700_000_000.9)
    r = c.query_range('100 - (avg(rate(node_cpu_seconds_total{mode="idle"}[1m])) * 100)', "30m", "30s")
    assert len(r) == 1 and len(r[0]["time"]) == 61 and len(r[0]["value"]) == 61
    assert r[0]["time"][0] == (1_700_000_000 - 1800) * 1000 and r[0]["time"][-1] == 1_700_000_000 * 1000
    # the query string is URL-encoded and start/end/step are passed like the reference (:42)
    assert "start=1699998200&end=1700000000&step=30s" in prom.requests[-1] and "%7Bmode%3D%22idle%22%7D" in prom.requests[-1]


def test_failures_give_empty_results():
    c = PrometheusClient("http://127.0.0.1:9", timeout=0.3)
    assert c.query_range("x") == [] and c.query("x") is None and c.failures == 2 and not c.reachable()


def test_non_json_body_is_null(prom):
    prom.garbage = True
    c = PrometheusClient(prom.url)
    assert c.http_get(prom.url + "/api/v1/query?query=1") is None and c.query_range("x") == []


def test_timeout_is_enforced(prom):
    prom.delay_s = 1.0
    c = PrometheusClient(prom.url, timeout=0.2)
    t0 = time.time()
    assert c.query_range("node_cpu_seconds_total") == []
    assert time.time() - t0 < 0.9           # the reference has no timeout at all


def test_process_result_shape():
    assert H.process_result([]) == {"labels": [], "data": []} and H.process_result(None) == {"labels": [], "data": []}   # :138
    r = H.process_result([{"time": [0, 60000], "value": [5.25, 7.0]}, {"time": [0], "value": [99.0]}])
    assert r == {"labels": [hhmm(0), hhmm(60)], "data": ["5.3", "7.0"]}     # only results[0] (:139), strings (:142)


def test_prometheus_history_payload(prom):
    cfg = Config()
    ph = H.PrometheusHistory(PrometheusClient(prom.url), H.prometheus_queries(cfg))
    out = ph.payload()
    assert set(out) == {"cpu", "memory", "disk", "gpuUtil", "gpuMem", "gpuTemp"}       # :146-153
    assert all(len(out[k]["labels"]) == 61 and len(out[k]["data"]) == 61 for k in out)
    assert out["disk"]["data"][0] == "45.3" and all(isinstance(x, str) for x in out["cpu"]["data"])
    # AMD exporter metric names, never DCGM
    assert not any("DCGM" in r for r in prom.requests) and any("gpu_gfx_activity" in r for r in prom.requests)


def test_ring_wraps_and_filters():
    r = SeriesRing(3)
    for i in range(5):
        r.append(i, i * 10)
    assert r.items() == [(2, 20), (3, 30), (4, 40)] and r.items(since=3) == [(3, 30), (4, 40)] and r.last() == (4, 40)
    with pytest.raises(ValueError):
        SeriesRing(0)


def test_bucket_mean():
    pairs = [(0, 10), (10, 20), (31, 40), (95, 7)]
    assert H.bucket_mean(pairs, 30, 100, 100) == [(0, 15.0), (30, 40.0), (90, 7.0)]


def test_local_history_and_persistence(tmp_path, clock):
    st = HistoryStore(100)
    lh = H.LocalHistory(st, "30m", 30.0, now=clock)
    host = {"cpu": {"percent": "12.0", "util_percent": "9.5"}, "memory": {"percent": "40.0"}, "disk": {"percent": "45.3"}}
    gpus = [{"index": 0, "utilization": 50, "gfxActivityAvg": 52.0, "memoryUsed": 1000, "temperature": 60, "hbmPercent": 10.0,
             "powerW": 500.0, "xgmi": {"rxGBps": 1.0, "txGBps": 2.0}},
            {"index": 1, "utilization": 70, "gfxActivityAvg": None, "memoryUsed": 3000, "temperature": 70, "hbmPercent": 30.0,
             "powerW": 700.0, "xgmi": {"rxGBps": 3.0, "txGBps": 4.0}}]
    for _ in range(3):
        st.record(clock(), H.history_values(host, gpus, {"tokensPerSec": 5000.0, "kvCachePercent": 60.0}))
        clock.advance(30)
    out = lh.payload()
    assert out["cpu"]["data"] == ["9.5"] * 3                 # true utilisation, like node_cpu in the reference
    assert out["gpuUtil"]["data"][0] == "61.0" and out["gpuMem"]["data"][0] == "2000.0" and out["gpuTemp"]["data"][0] == "65.0"
    assert out["xgmi"]["data"][0] == "10.0" and out["vllmTokensPerSec"]["data"][0] == "5000.0"
    assert out["perGpu"]["1"]["hbm"]["data"][0] == "30.0"
    path = str(tmp_path / "hist.json")
    st.save(path)
    st2 = HistoryStore(100)
    assert st2.load(path) and st2.window("cpu") == st.window("cpu")
    assert not HistoryStore(10).load(str(tmp_path / "missing.json"))
    (tmp_path / "bad.json").write_text("{not json")
    assert not HistoryStore(10).load(str(tmp_path / "bad.json"))


def test_local_history_empty_shape():
    out = H.LocalHistory(HistoryStore(10)).payload()
    for k in H.REFERENCE_KEYS:
        assert out[k] == {"labels": [], "data": []}
