"""The validation phases' own arithmetic and plumbing, on CPU (the phases themselves run on a GPU: test_gpu_stack.py)."""
import json
import time
import urllib.request

import pytest

from k8s_llm_monitor_amd.validation import phases, rocprof


def test_mean_ci95_known_values():
    r = phases.mean_ci95([1.0, 2.0, 3.0, 4.0, 5.0])
    assert r["n"] == 5 and abs(r["mean"] - 3.0) < 1e-12
    # sd = sqrt(2.5), t(4) = 2.776: half-width = 2.776 * 1.5811 / sqrt(5)
    assert abs(r["ci95"] - 2.776 * (2.5 ** 0.5) / 5 ** 0.5) < 1e-9
    assert phases.mean_ci95([]) == {"n": 0, "mean": None, "ci95": None}
    assert phases.mean_ci95([7.0])["ci95"] is None
    assert phases.mean_ci95([None, 2.0, 2.0])["ci95"] == 0.0


def test_product_serves_clients_and_counts_samples():
    """The overhead phase's "product" leg: sampler + server + one SSE stream + pollers, here on the fake GPU source."""
    p = phases.Product(period_s=0.2, pollers=2, gpu_source="fake")
    try:
        time.sleep(1.5)
        body = json.loads(urllib.request.urlopen(p.srv.url + "/api/gpu/metrics", timeout=3).read())
        assert isinstance(body, list) and body
    finally:
        st = p.stop()
    assert st["calls"] >= 4 and st["errors"] == 0
    assert st["clients"]["errors"] == 0 and st["clients"]["requests"] >= 2 and st["clients"]["sse_events"] >= 1
    assert st["pollers"] == 2 and st["sse_clients"] == 1


def _phase_results(overhead_mean, overhead_ci, power_err, mono=True, follows=True):
    lo, hi = overhead_mean - overhead_ci, overhead_mean + overhead_ci
    return {"hbm_fidelity": {"max_error_vs_driver_pct": 0.2, "max_error_vs_allocated_pct": 0.0},
            "activity_fidelity": {"max_error_points": 0.5},
            "power_temperature": {"power_error_pct": power_err, "temperature_rise_c": 11.0, "temperature_monotonic": mono,
                                  "temperature_follows_load": follows},
            "overhead": {"overhead_pct": overhead_mean, "overhead_ci95_pct": overhead_ci, "overhead_stress_pct": 0.1,
                         "overhead_gpu_side_pct": 0.01,
                         "overhead": {"mean_pct": overhead_mean, "ci95_pct": overhead_ci, "low_pct": lo, "high_pct": hi,
                                      "below_noise_floor": lo <= 0 <= hi},
                         "sampler_stats": {"on": {"calls": 6}}},
            "alert_latency": {"max_latency_s": 4.2}}


def test_summarize_has_power_temperature_and_judges_the_interval():
    s = phases.summarize(_phase_results(0.05, 0.12, 0.4))
    assert s["pass"] == {"hbm": True, "activity": True, "overhead": True, "latency": True, "power": True, "temperature": True}
    assert s["sampling_overhead_below_noise_floor"] is True and s["sampler_calls_per_window"] == 6
    assert abs(s["sampling_overhead_high_pct"] - 0.17) < 1e-12
    # a mean under 1 % whose interval reaches past it does not pass; neither does a disagreeing power gauge
    s = phases.summarize(_phase_results(0.8, 0.5, 3.5, mono=False))
    assert s["pass"]["overhead"] is False and s["pass"]["power"] is False and s["pass"]["temperature"] is False
    # phases that did not run add no pass key
    s = phases.summarize({"hbm_fidelity": {"max_error_vs_driver_pct": 0.2}})
    assert "power" not in s["pass"] and "temperature" not in s["pass"]


def test_kernel_durations_skip_warmup_and_match_by_name():
    rows = [{"name": "gemm_bf16_256_t<false>", "start": 1000 * i, "end": 1000 * i + 700 + (300 if i < 4 else 0)} for i in range(10)]
    rows.append({"name": "fill", "start": 5, "end": 50})
    d = rocprof.kernel_durations(rows, "gemm_bf16", skip=4)
    assert d == [0.7] * 6
    assert rocprof.kernel_durations(rows, "nothing") == []


# ---- time budget ---------------------------------------------------------------
def test_plan_phases_sheds_the_longest_first_and_keeps_canonical_order():
    cost = {"hbm": 10, "activity": 25, "overhead": 85, "latency": 120, "vllm": 20}
    want = {"vllm", "hbm", "latency", "overhead", "activity"}
    assert phases.plan_phases(want, None, cost) == (["hbm", "activity", "overhead", "latency", "vllm"], [])
    assert phases.plan_phases(want, 260, cost) == (["hbm", "activity", "overhead", "latency", "vllm"], [])
    assert phases.plan_phases(want, 259, cost) == (["hbm", "activity", "overhead", "vllm"], ["latency"])
    assert phases.plan_phases(want, 60, cost) == (["hbm", "activity", "vllm"], ["latency", "overhead"])
    assert phases.plan_phases(want, 5, cost) == ([], ["latency", "overhead", "activity", "vllm", "hbm"])
    assert set(phases.PHASE_COST_S) == set(phases.PHASE_ORDER)
    b = phases.Budget(100.0, t0=time.monotonic() - 50.0, reserve_s=10.0)
    assert 39.0 < b.left() <= 40.0 and b.allows("hbm", 10.0) and not b.allows("overhead", 85.0)
    assert b.shed[0]["phase"] == "overhead" and phases.Budget(None).allows("x", 1e9)


def test_stratified_phases_cover_the_period_and_are_seeded():
    xs = phases.stratified_phases(8, 20261111)
    assert xs == phases.stratified_phases(8, 20261111) != phases.stratified_phases(8, 1)
    assert sorted(int(x * 8) for x in xs) == list(range(8)) and max(xs) - min(xs) > 0.75


class _FakeHold:
    """HbmHold's interface over the fake GPU reader's scripted HBM level."""

    def __init__(self, level, total_bytes):
        self.level, self.total, self.held = level, total_bytes, 0

    def add(self, n):
        self.held += n
        self.level["frac"] += n / self.total
        return n

    def drop(self, n):
        self.held -= n
        self.level["frac"] -= n / self.total
        return n

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.level["frac"] -= self.held / self.total
        self.held = 0


def _fake_monitor(scenario, realtime, page_poll=0.3):
    from k8s_llm_monitor_amd.collectors.gpu import GpuSource
    from k8s_llm_monitor_amd.config import demo_config
    from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader
    from k8s_llm_monitor_amd.server.sampler import Monitor
    cfg = demo_config(periods={"realtime": realtime, "page_alert_poll": page_poll}, sources={"kube": "none", "vllm": "none"})
    return Monitor(cfg, gpu=GpuSource(FakeGpuReader(1, scenario=scenario)))


def test_alert_latency_loop_measures_a_distribution_on_every_path():
    """The crossing loop, the three observers and the summary arithmetic against the fake GPU source (a 0.5 s sampler period
    and a 0.3 s page poll so the test takes seconds): VERDICT r2 weak 4."""
    from k8s_llm_monitor_amd.fakes.gpu import HBM_TOTAL_MIB, GpuState
    level = {"frac": 0.10}
    mon = _fake_monitor(lambda t, i: GpuState(hbm_frac=level["frac"]), realtime=0.5)
    r = phases.alert_latency(None, None, crossings=4, step_gib=4.0, monitor=mon, hold=_FakeHold(level, HBM_TOTAL_MIB << 20), timeout_s=5.0)
    assert r["complete"] and r["crossings"] == 6 and [x["tier"] for x in r["rows"]] == ["minor"] * 4 + ["serious", "critical"]
    assert sorted(int(x["phase_of_period"] * 4) for x in r["rows"][:4]) == [0, 1, 2, 3]          # one crossing per quarter of the period
    for x in r["rows"]:
        assert 0 <= x["server_latency_s"] <= r["bound_sse_s"] + 0.1 and x["sse_latency_s"] is not None and x["poll_latency_s"] is not None
        assert x["sse_latency_s"] <= x["server_latency_s"] + 0.06              # the push does not wait for a poll
        assert x["poll_latency_s"] <= r["bound_poll_s"] + 0.1
    assert r["server_latency_spread_s"] > 0.2 and r["max_latency_s"] == max(r["poll"]["max"], r["sse"]["max"], r["server"]["max"])
    assert r["sample_period_s"] == 0.5 and r["page_poll_s"] == 0.3 and r["bound_poll_s"] == pytest_approx(0.8 + r["tick_max_s"])
    assert abs(level["frac"] - 0.10) < 1e-9                                       # everything released
    s = phases.summarize({"alert_latency": r})
    assert s["pass"] == {"latency": True} and s["alert_latency_crossings"] == 6 and s["sse_latency_max_s"] == r["sse"]["max"]


def test_alert_watcher_runs_the_tick_early_when_a_tier_changes():
    """`periods.alert_watch`: between two realtime ticks (here 30 s apart: none happens during the test) the GPUs are read again
    through the watcher's own normaliser; a tiered signal in another tier than the published snapshot runs the tick at once."""
    import time
    from k8s_llm_monitor_amd.collectors.gpu import GpuSource
    from k8s_llm_monitor_amd.config import demo_config
    from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader, GpuState
    from k8s_llm_monitor_amd.server.sampler import Monitor
    level = {"frac": 0.10, "temp": 50.0}
    scen = lambda t, i: GpuState(hbm_frac=level["frac"], temp=level["temp"])
    cfg = demo_config(periods={"realtime": 30.0, "alert_watch": 0.1}, sources={"kube": "none", "vllm": "none"})
    mon = Monitor(cfg, gpu=GpuSource(FakeGpuReader(1, scenario=scen))).start()
    try:
        assert "watch" in mon.tasks and mon.stats()["alert_watch"]["period_s"] == 0.1
        time.sleep(0.5)
        assert mon.watch_pokes == 0 and mon.tasks["realtime"].calls == 1 and mon.alerts_public() == {"minor": [], "serious": [], "critical": []}
        level["frac"] = 0.72                                    # crosses 70 %: the next watcher read sees another tier
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 3.0 and not mon.alerts_public()["minor"]:
            time.sleep(0.01)
        dt = time.perf_counter() - t0
        assert [a["key"] for a in mon.alerts_public()["minor"]] == ["gpu/0/hbm"] and dt < 0.6 and mon.watch_pokes == 1
        time.sleep(0.4)
        assert mon.watch_pokes == 1                             # the published snapshot now says "minor": nothing more to do
        level["frac"], level["temp"] = 0.10, 96.0               # one clears, another appears: one early tick for both
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 3.0 and mon.alerts_public()["minor"]:
            time.sleep(0.01)
        assert time.perf_counter() - t0 < 0.6 and mon.watch_pokes == 2
        assert any(a["key"] == "gpu/0/temp" for tier in mon.alerts_public().values() for a in tier)
        assert mon.tasks["realtime"].calls == 1                 # the task's own grid was not disturbed
    finally:
        mon.stop()
    # off: by configuration, under the reference preset (alerts are evaluated per request there), for a source without a fast read
    off = Monitor(demo_config(periods={"alert_watch": 0.0}, sources={"kube": "none", "vllm": "none"}))
    ref = Monitor(demo_config(compat={"reference_defaults": True}, sources={"kube": "none", "vllm": "none"}))
    assert "watch" not in off.tasks and "watch" not in ref.tasks and ref.cfg.periods.alert_watch == 0.0
    with pytest.raises(Exception, match="alert_watch"):
        demo_config(periods={"alert_watch": 0.01})


def test_throttle_phase_sees_both_alerts_appear_and_clear():
    """BASELINE.json config 5 on the fake: the "burner" puts the card on its power cap (99 % of 1400 W, ppt residency 100 %,
    clock 2400 -> 1830 MHz); both alerts appear on /api/alerts and /api/events and clear within two periods of the stop."""
    from k8s_llm_monitor_amd.fakes.gpu import GpuState
    on = {"burn": False}

    class FakeBurner:
        def start(self, seconds):
            on["burn"] = True
            return self

        def stop(self):
            on["burn"] = False
            return {"tflops_busy": 1600.0, "duty_achieved": 100.0}

    def scen(t, i):
        return GpuState(power_w=1390.0, ppt_throttle=1.0, clk_gfx=1830.0, gfx=100.0) if on["burn"] else GpuState(power_w=300.0, clk_gfx=2400.0)

    r = phases.throttle_alerts(None, None, None, 0, burn_s=2.0, idle_s=0.5, timeout_s=4.0, monitor=_fake_monitor(scen, realtime=0.25), burner=FakeBurner())
    assert r["power_alert"] and r["throttle_alert"] and r["alerts_before_load"] == [], r
    assert r["ppt_residency_mean_pct"] > 90 and r["clock_dropped"] and r["clock_idle_mhz"] == 2400.0 and r["clock_under_cap_mhz"] == 1830.0
    assert 0 < r["power_alert_latency_s"] < 0.6 and 0 < r["throttle_alert_latency_s"] < 0.9 and r["power_alert_sse_latency_s"] < 0.6
    assert 0 < r["power_alert_clear_s"] <= 0.5 and 0 < r["throttle_alert_clear_s"] <= 0.5 and r["power_alert_sse_clear_s"] is not None
    s = phases.summarize({"throttle_alerts": r})
    assert s["pass"] == {"power_alert": True, "throttle_alert": True} and s["power_alert"] is True
    # an unloaded card may report its PARKED clock (~150 MHz) instead of its boost clock (seen on the MI355X: idle samples 2403
    # and 158 MHz): the reference for "the clock dropped" is the highest reading outside the load, not the idle mean
    calls = {"n": 0}

    def parked(t, i):
        if on["burn"]:
            return GpuState(power_w=1390.0, ppt_throttle=1.0, clk_gfx=1760.0, gfx=100.0)
        calls["n"] += 1
        return GpuState(power_w=260.0, clk_gfx=158.0 if calls["n"] % 2 else 2403.0)

    r = phases.throttle_alerts(None, None, None, 0, burn_s=2.0, idle_s=0.8, timeout_s=4.0, monitor=_fake_monitor(parked, realtime=0.25), burner=FakeBurner())
    assert r["throttle_alert"] and r["clock_dropped"] and r["clock_idle_mhz"] == 2403.0 and 158.0 in r["clock_idle_samples_mhz"], r
    # a card that never reaches its cap fails both flags, and says why
    r = phases.throttle_alerts(None, None, None, 0, burn_s=1.0, idle_s=0.3, timeout_s=1.0, burner=FakeBurner(),
                               monitor=_fake_monitor(lambda t, i: GpuState(power_w=900.0, clk_gfx=2400.0), realtime=0.25))
    assert not r["power_alert"] and not r["throttle_alert"] and r["power_alert_latency_s"] is None and not r["clock_dropped"]


def pytest_approx(x):
    import pytest
    return pytest.approx(x, abs=1e-6)


def test_bench_detail_feeds_api_validation(tmp_path, monkeypatch):
    """bench.py's full result goes to a file; its validation part is what /api/validation serves."""
    from k8s_llm_monitor_amd.server import http
    from k8s_llm_monitor_amd.validation import benchlib
    monkeypatch.setenv("KLM_STATE_DIR", str(tmp_path / "state"))
    xf = {"world": 8, "tx_error_pct": 0.7, "unit_check_tx": {"verdict": "KiB (x1024)"}, "unit_check_rx": {"verdict": "KiB (x1024)"}}
    detail = dict(_phase_results(0.05, 0.12, 0.4), xgmi_fidelity=xf)
    res = {"metric": "m", "value": 1.0, "validation": phases.summarize(detail), "validation_detail": detail}
    p = benchlib.write_detail(res, str(tmp_path / "out" / "bench_detail.json"))
    assert json.load(open(p))["validation"]["pass"]["hbm"] is True
    rep = http.load_validation_report()
    assert rep["source"] == "bench.py" and rep["summary"]["pass"]["xgmi"] is True and rep["summary"]["xgmi_unit_verdict_tx"] == "KiB (x1024)"
    assert rep["xgmi_fidelity"]["world"] == 8 and rep["alert_latency"]["max_latency_s"] == 4.2
    line = benchlib.compact(res, p)
    assert line["xgmi_unit_verdict_tx"] == "KiB (x1024)" and line["hbm_error_pct"] == 0.2 and line["validation_pass"] is True and line["detail"] == p
