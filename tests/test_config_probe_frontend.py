"""Config loader, the RCCL probe's sweep/bandwidth model, and static checks of the page."""
import json
import os
import re
import subprocess

import pytest

from k8s_llm_monitor_amd import config as C
from k8s_llm_monitor_amd.parallel import probe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- config ------------------------------------------------------------------------------
def test_defaults_are_the_reference_literals():
    c = C.load_config(env={})
    assert c.server.port == 8888 and c.server.host == "0.0.0.0" and c.prometheus.url == "http://localhost:9090"   # :10-11, :297
    assert c.thresholds.cpu == [70.0, 85.0, 95.0] and c.thresholds.hbm == [70.0, 85.0, 95.0]                      # :163-182
    assert c.periods.realtime == 5.0 and c.periods.pods == 10.0 and c.periods.history_window == "30m" and c.periods.history_step == 30.0
    assert "DCGM" not in json.dumps(c.to_dict()) and "nvidia" not in json.dumps(c.to_dict()).lower()


def test_file_env_override_precedence(tmp_path):
    p = tmp_path / "c.json"
    p.write_text(json.dumps({"server": {"port": 9000}, "thresholds": {"hbm": [60, 80, 90]}}))
    c = C.load_config(str(p), env={"KLM_SERVER_PORT": "9100", "KLM_SOURCES_GPU": "fake", "KLM_THRESHOLDS_CPU": "50, 60,70",
                                   "KLM_COMPAT_FIRST_SIGHT_ALERTS": "true", "KLM_SOURCES_VLLM_ENDPOINTS": "http://a:8000,http://b:8000"},
                      overrides={"server": {"port": 9200}})
    assert c.server.port == 9200 and c.thresholds.hbm == [60.0, 80.0, 90.0] and c.thresholds.cpu == [50.0, 60.0, 70.0]
    assert c.sources.gpu == "fake" and c.compat.first_sight_alerts is True and c.sources.vllm_endpoints == ["http://a:8000", "http://b:8000"]
    y = tmp_path / "c.yaml"
    y.write_text("server:\n  port: 7000\ncompat:\n  language: en\n")
    assert C.load_config(str(y), env={}).compat.language == "en"


@pytest.mark.parametrize("bad", [{"server": {"prot": 1}}, {"nope": {}}, {"thresholds": {"cpu": [1, 2]}},
                                 {"thresholds": {"cpu": [90, 80, 95]}}, {"periods": {"realtime": 0}},
                                 {"sources": {"gpu": "nvml"}}, {"server": {"port": "abc"}}, {"compat": {"cpu_percent": "x"}}])
def test_bad_config_is_rejected(bad):
    with pytest.raises(C.ConfigError):
        C.load_config(env={}, overrides=bad)


# ---- probe model ---------------------------------------------------------------------------
def test_sweep_and_bus_factor():
    s = probe.sweep_sizes()
    assert s[0] == 1024 and s[-1] == 1 << 30 and len(s) == 11 and all(b == a * 4 for a, b in zip(s, s[1:]))
    assert probe.sweep_sizes(1000, 5000, 2) == [1000, 2000, 4000, 5000]
    assert probe.bus_factor(1) == 0.0 and probe.bus_factor(2) == 1.0 and probe.bus_factor(8) == 1.75
    assert probe.bus_factor(8, "allgather") == 0.875
    with pytest.raises(ValueError):
        probe.sweep_sizes(0, 10)


def test_ceilings_match_baseline_table():
    assert probe.ceilings(1) == {"mesh_GBps": None, "ring_GBps": None}
    assert [probe.ceilings(n)["mesh_GBps"] for n in (2, 4, 8)] == [153.0, 459.0, 1071.0]
    assert probe.ceilings(8)["ring_GBps"] == 153.0


def test_curve_and_single_ring_detection():
    fast = probe.curve(8, [probe.point(8, b, b / 400e9) for b in probe.sweep_sizes()])
    assert fast["peak_busbw_GBps"] == pytest.approx(700.0) and fast["mesh_efficiency"] == pytest.approx(700 / 1071)
    assert not fast["single_ring_suspected"]
    ring = probe.curve(8, [probe.point(8, 1 << 30, (1 << 30) / 85e9)])
    assert ring["single_ring_suspected"]                    # busBW ~149 GB/s at n=8: one ring
    one = probe.curve(1, [probe.point(1, 1 << 20, 1e-6)])
    assert one["peak_busbw_GBps"] == 0.0 and one["ceiling_mesh_GBps"] is None and one["peak_algbw_GBps"] > 0


def test_results_roundtrip_and_merge(tmp_path):
    p = str(tmp_path / "sub" / "probe.json")
    d1 = probe.results_doc([probe.curve(1, [probe.point(1, 1024, 1e-6)]), probe.curve(8, [], verified=False, executor="none")])
    probe.save_results(d1, p)
    assert probe.load_results(p)["scaling"][1] == {"n_gpus": 8, "dtype": "bf16", "peak_busbw_GBps": None, "peak_algbw_GBps": None,
                                                   "ceiling_mesh_GBps": 1071.0, "verified": False, "mode": "allreduce"}
    assert probe.load_results(p)["scaling"][0]["mode"] == "local_copy"
    d2 = probe.merge_results(probe.load_results(p), probe.results_doc([probe.curve(8, [probe.point(8, 1 << 30, 2e-3)])]))
    assert [c["n_gpus"] for c in d2["curves"]] == [1, 8] and d2["curves"][1]["verified"]
    assert probe.load_results(str(tmp_path / "none.json")) is None


def test_native_probe_driver_with_a_stub_runner():
    lines = [json.dumps({"type": "header", "n_gpus": 2, "rccl_version": 22203}),
             "noise", json.dumps(probe.point(2, 1024, 1e-5, check="ok")), json.dumps(probe.point(2, 4096, 2e-5, check="ok"))]

    def runner(cmd, **kw):
        assert cmd[1:5] == ["--gpus", "2", "--dtype", "bf16"]
        return subprocess.CompletedProcess(cmd, 0, "\n".join(lines), "")

    c = probe.run_native(2, exe="/x/klm_rccl_probe", runner=runner)
    assert c["verified"] and len(c["points"]) == 2 and c["header"]["rccl_version"] == 22203

    def failing(cmd, **kw):
        return subprocess.CompletedProcess(cmd, 1, json.dumps({"type": "error", "error": "need 2 devices, have 1"}), "")
    with pytest.raises(RuntimeError, match="need 2 devices"):
        probe.run_native(2, exe="/x", runner=failing)

    # n = 1: only an out-of-place run (a real local copy) is a measurement; an in-place header (an older binary's no-op
    # timing) and a point faster than 8 TB/s both leave the curve unverified
    def one(in_place, seconds):
        out = "\n".join([json.dumps({"type": "header", "n_gpus": 1, "in_place": in_place, "mode": "local_copy"}),
                         json.dumps(dict(probe.point(1, 1 << 28, seconds), check="ok"))])
        return lambda cmd, **kw: subprocess.CompletedProcess(cmd, 0, out, "")
    good = probe.run_native(1, exe="/x", runner=one(False, (1 << 28) / 3.1e12))
    assert good["verified"] and good["mode"] == "local_copy" and good["peak_algbw_GBps"] == pytest.approx(3100.0)
    assert not probe.run_native(1, exe="/x", runner=one(True, (1 << 28) / 3.1e12))["verified"]
    noop = probe.run_native(1, exe="/x", runner=one(False, 0.436e-6))
    assert not noop["verified"] and noop["points"][0]["check"] == "implausible" and noop["failed_bytes"] == [1 << 28]


# ---- frontend ------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def page():
    with open(os.path.join(ROOT, "k8s_llm_monitor_amd", "web", "monitor.html"), encoding="utf-8") as f:
        return f.read()


def test_page_is_self_contained(page):
    assert not re.search(r"<script[^>]+src=", page) and not re.search(r"<link[^>]+href=\"https?:", page)
    assert "cdn." not in page.lower() or "CDN" in page        # only mentioned in the comment
    assert not re.search(r"https?://(?!www\.w3\.org)", re.sub(r"<!--.*?-->", "", page, flags=re.S))


def test_page_has_the_reference_panels_and_periods(page):
    for el in ("cpuValue", "memValue", "diskValue", "gpuValue", "gpuLabel", "cpuChart", "memChart", "diskChart", "gpuChart",
               "podTableBody", "minorCount", "seriousCount", "criticalCount", "alertModal", "alertList", "currentTime", "lastUpdate"):
        assert f'id="{el}"' in page, el
    for api in ("/api/host/metrics", "/api/gpu/metrics", "/api/k8s/pods", "/api/history", "/api/alerts"):
        assert api in page
    assert "setInterval(fetchRealtime, 5000)" in page and "setInterval(fetchHistory, 30000)" in page
    assert "setInterval(fetchPods, 10000)" in page and "setInterval(updateTime, 1000)" in page
    for txt in ("轻微", "严重", "紧急", "暂无告警", "修复建议"):
        assert txt in page


def test_page_escapes_api_text(page):
    script = page[page.index("<script>"):]
    # every innerHTML assignment in the script goes through esc() or is a constant
    for m in re.finditer(r"innerHTML\s*=\s*([^;]+);", script):
        rhs = m.group(1)
        assert "esc(" in rhs or "html" == rhs.strip(), rhs
    assert "tbody.innerHTML +=" not in script and "textContent" in script


def test_page_script_parses_with_node_if_present(page, tmp_path):
    import shutil
    node = shutil.which("node")
    if not node:
        pytest.skip("node not installed")
    js = page[page.index("<script>") + 8: page.rindex("</script>")]
    f = tmp_path / "page.js"
    f.write_text(js)
    r = subprocess.run([node, "--check", str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_xgmi_expected_volume_and_counter_totals():
    from k8s_llm_monitor_amd.validation.phases import xgmi_expected_bytes, xgmi_link_totals
    assert xgmi_expected_bytes(1, 1 << 20, 10) == 0.0
    assert xgmi_expected_bytes(2, 1 << 20, 10) == 10 * (1 << 20)
    assert xgmi_expected_bytes(8, 1 << 28, 40) == 2 * 7 / 8 * (1 << 28) * 40
    t = xgmi_link_totals({"xgmi": {"links": [{"txBytes": 10, "rxBytes": 4, "state": "up"}, {"txBytes": None, "rxBytes": 6, "state": "down"}]}})
    assert t["tx"] == 10 and t["rx"] == 10 and t["links_up"] == 1 and t["per_link_tx"] == [10.0, 0.0]
    assert xgmi_link_totals({})["tx"] == 0.0
