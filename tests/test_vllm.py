"""vLLM metrics: exposition parser, histogram quantiles, rate/quantile deltas, PromQL source."""
import math

import pytest

from k8s_llm_monitor_amd.collectors.prometheus import PrometheusClient
from k8s_llm_monitor_amd.collectors.vllm import (VllmDirectSource, VllmPrometheusSource, VllmSource, histogram_quantile,
                                                 open_vllm_source, parse_exposition)
from k8s_llm_monitor_amd.config import Config, demo_config
from k8s_llm_monitor_amd.fakes.prometheus import FakePrometheus
from k8s_llm_monitor_amd.fakes.vllm import FakeVllm, VllmState

TEXT = '''# HELP vllm:num_requests_running Number of requests currently running on GPU.
# TYPE vllm:num_requests_running gauge
vllm:num_requests_running{model_name="meta-llama/Meta-Llama-3-8B-Instruct"} 12.0
vllm:gpu_cache_usage_perc{model_name="m \\"q\\""} 0.5 1700000000000
vllm:time_to_first_token_seconds_bucket{le="0.1",model_name="m"} 10
vllm:time_to_first_token_seconds_bucket{le="+Inf",model_name="m"} 20
garbage line here
vllm:bad{x="1"} notanumber
'''


def test_parse_exposition():
    s = parse_exposition(TEXT)
    assert ("vllm:num_requests_running", {"model_name": "meta-llama/Meta-Llama-3-8B-Instruct"}, 12.0) in s
    assert ("vllm:gpu_cache_usage_perc", {"model_name": 'm "q"'}, 0.5) in s          # timestamp ignored, escapes undone
    assert s[-1][1]["le"] == "+Inf" and s[-1][2] == 20.0 and len(s) == 4


def test_histogram_quantile_matches_promql():
    b = [(0.1, 10.0), (0.5, 30.0), (1.0, 40.0), (math.inf, 40.0)]
    assert histogram_quantile(0.5, b) == pytest.approx(0.1 + 0.4 * (20 - 10) / 20)
    assert histogram_quantile(0.99, b) == pytest.approx(0.5 + 0.5 * (39.6 - 30) / 10)
    assert histogram_quantile(0.5, [(0.1, 0.0), (math.inf, 10.0)]) == 0.1         # falls in +Inf: highest finite bound
    assert histogram_quantile(0.5, [(0.1, 5.0)]) is None                             # no +Inf bucket
    assert histogram_quantile(0.5, [(math.inf, 0.0)]) is None


def test_direct_source_rates_and_quantiles(clock):
    st = VllmState(tokens_per_s=5000.0, prompt_tokens_per_s=20000.0, ttft_median_s=0.2, kv_usage=0.63, running=32, waiting=3)
    fv = FakeVllm(lambda t: st, clock=clock)
    src = VllmDirectSource(["fake://v/metrics"], fetch=fv.fetch, clock=clock)
    first = src.sample()
    assert first["available"] and first["tokensPerSec"] is None and first["kvCachePercent"] == pytest.approx(63.0)
    assert first["running"] == 32 and first["waiting"] == 3 and first["models"] == ["meta-llama/Meta-Llama-3-8B-Instruct"]
    clock.advance(5.0)
    s = src.sample()
    assert s["tokensPerSec"] == pytest.approx(5000.0, rel=1e-3) and s["promptTokensPerSec"] == pytest.approx(20000.0, rel=1e-3)
    assert s["ttftP50"] == pytest.approx(0.2, rel=0.15) and s["ttftP99"] > s["ttftP50"]
    # the window quantile follows a latency change at once (lifetime histograms would lag)
    st.ttft_median_s = 3.0
    clock.advance(5.0)
    assert src.sample()["ttftP50"] == pytest.approx(3.0, rel=0.25)


def test_direct_source_down_and_url_normalisation():
    def refuse(u):
        raise ConnectionRefusedError(u)
    src = VllmDirectSource(["http://10.0.0.1:8000"], fetch=refuse)
    assert src.endpoints == ["http://10.0.0.1:8000/metrics"]
    s = src.sample()
    assert s["available"] is False and "ConnectionRefusedError" in src.last_error
    assert VllmSource().sample()["available"] is False


def test_prometheus_source():
    with FakePrometheus() as p:
        s = VllmPrometheusSource(PrometheusClient(p.url), Config().prometheus).sample()
        assert s["available"] and s["ttftP50"] == 0.12 and s["ttftP99"] == 0.8 and s["running"] == 48.0 and s["waiting"] == 2.0
        assert 55.0 <= s["kvCachePercent"] <= 65.0 and s["tokensPerSec"] >= 5000.0
        assert any("time_to_first_token_seconds_bucket" in r for r in p.requests)
    down = VllmPrometheusSource(PrometheusClient("http://127.0.0.1:9", timeout=0.2), Config().prometheus).sample()
    assert down["available"] is False


def test_open_source_selection():
    assert open_vllm_source("none").kind == "none"
    assert open_vllm_source("fake", demo_config()).kind == "fake"
    cfg = Config()
    cfg.sources.vllm_endpoints = ["http://a:8000", "http://b:8000/metrics"]
    s = open_vllm_source("auto", cfg)
    assert s.kind == "direct" and s.endpoints == ["http://a:8000/metrics", "http://b:8000/metrics"]
    assert open_vllm_source("auto", Config(), PrometheusClient("http://127.0.0.1:9", timeout=0.2)).kind == "none"


def test_a_vllm_endpoint_that_stops_answering_raises_an_outage_alert():
    """Never seen -> silence; seen and gone -> critical; one of two replicas gone -> serious; back -> clear."""
    from k8s_llm_monitor_amd.config import load_config
    from k8s_llm_monitor_amd.server.alerts import AlertState, check_alerts
    cfg = load_config(env={})
    host = {"cpu": {"percent": "1.0"}, "memory": {"percent": "1.0"}, "disk": {"percent": "1"}}
    up2 = {"available": True, "endpoints": 2, "kvCachePercent": 10.0, "waiting": 0, "running": 1}
    up1 = dict(up2, endpoints=1)
    keys = lambda a: {x["key"]: x["tier"] for t in a.values() for x in t}
    a, st = check_alerts(host, [], [], AlertState(), cfg, None)
    assert "vllm/down" not in keys(a) and st.vllm_endpoints == 0          # nothing configured / never up
    a, st = check_alerts(host, [], [], st, cfg, up2)
    assert "vllm/down" not in keys(a) and st.vllm_endpoints == 2
    a, st = check_alerts(host, [], [], st, cfg, up1)
    assert keys(a)["vllm/down"] == "serious" and "1 of 2" in a["serious"][0]["desc"] or "1/2" in a["serious"][0]["desc"]
    a, st = check_alerts(host, [], [], st, cfg, None)
    assert keys(a)["vllm/down"] == "critical"
    a, st = check_alerts(host, [], [], st, cfg, None)
    assert keys(a)["vllm/down"] == "critical"                             # level-triggered while it lasts
    a, st = check_alerts(host, [], [], st, cfg, up2)
    assert "vllm/down" not in keys(a)


def test_direct_source_reports_one_row_per_replica():
    """Two endpoints, one of them a router that merges two replicas: three rows, so a saturated
    replica is visible next to idle ones (the totals and the max alone would hide which one)."""
    from k8s_llm_monitor_amd.collectors.vllm import VllmDirectSource
    t = [0.0]
    gen = {"a": 0, "r0": 0, "r1": 0}

    def fetch(url):
        if "pod-a" in url:
            return (f'vllm:num_requests_running{{model_name="m"}} 2\nvllm:num_requests_waiting{{model_name="m"}} 0\n'
                    f'vllm:kv_cache_usage_perc{{model_name="m"}} 0.10\nvllm:generation_tokens_total{{model_name="m"}} {gen["a"]}\n')
        if "router" in url:
            return "".join(f'vllm:num_requests_running{{model_name="m",replica="{i}"}} {run}\n'
                           f'vllm:num_requests_waiting{{model_name="m",replica="{i}"}} {wait}\n'
                           f'vllm:kv_cache_usage_perc{{model_name="m",replica="{i}"}} {kv}\n'
                           f'vllm:generation_tokens_total{{model_name="m",replica="{i}"}} {gen["r%d" % i]}\n'
                           for i, (run, wait, kv) in enumerate([(30, 200, 0.99), (1, 0, 0.02)]))
        raise OSError("connection refused")

    src = VllmDirectSource(["http://pod-a:8000", "http://router:8000", "http://pod-dead:8000"], fetch=fetch, clock=lambda: t[0])
    src.sample()
    t[0] += 2.0
    gen.update(a=1000, r0=4000, r1=100)
    v = src.sample()
    assert v["endpoints"] == 2 and v["running"] == 33 and v["waiting"] == 200 and v["kvCachePercent"] == pytest.approx(99.0)
    rows = {(r["endpoint"].split("//")[1].split(":")[0], r.get("replica")): r for r in v["replicas"]}
    assert set(rows) == {("pod-a", None), ("router", "0"), ("router", "1"), ("pod-dead", None)}
    assert rows[("pod-dead", None)]["up"] is False
    hot = rows[("router", "0")]
    assert hot["waiting"] == 200 and hot["kvCachePercent"] == pytest.approx(99.0) and hot["tokensPerSec"] == pytest.approx(2000.0)
    assert rows[("router", "1")]["tokensPerSec"] == pytest.approx(50.0) and rows[("pod-a", None)]["tokensPerSec"] == pytest.approx(500.0)
    assert v["tokensPerSec"] == pytest.approx(2550.0)
