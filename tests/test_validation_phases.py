"""The validation phases' own arithmetic and plumbing, on CPU (the phases themselves run on a GPU: test_gpu_stack.py)."""
import json
import time
import urllib.request

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
