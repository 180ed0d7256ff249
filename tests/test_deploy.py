"""The shipped deployment files parse and agree with the code (config keys, routes, RBAC)."""
import json
import os

import yaml

from k8s_llm_monitor_amd.config import load_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_config_loads():
    c = load_config(os.path.join(ROOT, "deploy", "config.example.yaml"), env={})
    assert c.server.port == 8888 and c.periods.realtime == 5 and c.thresholds.hbm == [70.0, 85.0, 95.0]


def test_daemonset_manifest(tmp_path):
    docs = list(yaml.safe_load_all(open(os.path.join(ROOT, "deploy", "k8s-daemonset.yaml"))))
    kinds = {d["kind"]: d for d in docs}
    assert {"DaemonSet", "Service", "ClusterRole", "ClusterRoleBinding", "ServiceAccount", "ConfigMap"} <= set(kinds)
    rule = kinds["ClusterRole"]["rules"][0]
    assert rule["resources"] == ["pods"] and set(rule["verbs"]) == {"get", "list"}          # read-only, pods only
    c = kinds["DaemonSet"]["spec"]["template"]["spec"]["containers"][0]
    assert c["command"][:4] == ["python3", "-m", "k8s_llm_monitor_amd", "serve"] and c["readinessProbe"]["httpGet"]["path"] == "/healthz"
    assert c["securityContext"]["privileged"] is False
    p = tmp_path / "c.json"
    p.write_text(json.dumps(yaml.safe_load(kinds["ConfigMap"]["data"]["config.yaml"])))
    assert load_config(str(p), env={}).sources.kube == "api"


def test_prometheus_jobs_have_no_nvidia_exporter():
    d = yaml.safe_load(open(os.path.join(ROOT, "deploy", "prometheus-scrape.yaml")))
    jobs = [j["job_name"] for j in d["scrape_configs"]]
    assert "amd-device-metrics-exporter" in jobs and not any("dcgm" in j.lower() or "nvidia" in j.lower() for j in jobs)


def test_llm_loadgen_manifest_runs_the_cli_and_looks_like_a_vllm_pod():
    from k8s_llm_monitor_amd.__main__ import main  # noqa: F401  (the command the pod runs exists)
    from k8s_llm_monitor_amd.collectors.kube import pods_from_list
    docs = list(yaml.safe_load_all(open(os.path.join(ROOT, "deploy", "k8s-llm-loadgen.yaml"))))
    dep = next(d for d in docs if d["kind"] == "Deployment")
    c = dep["spec"]["template"]["spec"]["containers"][0]
    assert c["command"][-1] == "serve-llm" and c["resources"]["limits"]["amd.com/gpu"] == 1
    import argparse
    from k8s_llm_monitor_amd import __main__ as cli
    ap_args = ["serve-llm"] + c["args"]
    # the manifest's arguments parse with the real CLI
    orig = cli._cmd_serve_llm
    seen = {}
    cli._cmd_serve_llm = lambda a: seen.update(vars(a)) or 0
    try:
        assert cli.main(ap_args) == 0
    finally:
        cli._cmd_serve_llm = orig
    assert seen["model"] == "llama3-8b" and seen["port"] == 8000 and seen["load_rps"] == 8.0
    pod = {"metadata": {"name": "klm-vllm-loadgen-abc", "namespace": "monitoring"},
           "spec": dep["spec"]["template"]["spec"], "status": {"phase": "Running"}}
    assert pods_from_list({"items": [pod]})[0]["vllm"] is True
    del argparse


def test_cmake_project_lists_every_native_source():
    """CMakeLists.txt and build.py describe the same targets and sources."""
    from k8s_llm_monitor_amd import build
    text = open(os.path.join(ROOT, "CMakeLists.txt")).read()
    for t in build.targets():
        assert t.name in text, t.name
        for s in t.sources:
            assert s.name in text, f"{s.name} missing from CMakeLists.txt"


def test_grafana_dashboard_and_alert_rules_only_use_metrics_the_monitor_exports():
    """deploy/grafana-dashboard.json and deploy/prometheus-alerts.yaml against a rendered /metrics of the fake node."""
    import json
    import re

    import yaml

    from k8s_llm_monitor_amd.collectors.gpu import GpuSource
    from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader
    from k8s_llm_monitor_amd.server.exporter import render
    clock = {"t": 0.0}
    src = GpuSource(FakeGpuReader(n_gpus=8, scenario="pcie", clock=lambda: clock["t"]))
    src.sample()
    clock["t"] = 5.0
    snap = {"host": {"cpu": {"percent": "5.0", "load_1min": 0.4}, "memory": {"percent": "10.0"}, "disk": {"percent": "20"}},
            "gpus": src.sample(), "pods": [{"status": "Running"}],
            "vllm": {"available": True, "ttftP50": 0.02, "ttftP99": 0.1, "tokensPerSec": 1000.0, "kvCachePercent": 40.0, "running": 3, "waiting": 0},
            "alerts": {"minor": [], "serious": [], "critical": []}}
    text = render(snap, {"tasks": {"gpu": {"calls": 2, "errors": 0, "mean_ms": 1.0, "max_ms": 1.5, "duty_percent": 0.02}}, "uptime_s": 10})
    exported = set(re.findall(r"^(klm_[a-z0-9_]+)", text, re.M))
    assert {"klm_gpu_xcd_activity_percent", "klm_gpu_pcie_replay_total", "klm_gpu_hbm_used_bytes"} <= exported

    dash = json.load(open(os.path.join(ROOT, "deploy", "grafana-dashboard.json")))
    exprs = [t["expr"] for p in dash["panels"] for t in p["targets"]]
    assert len(dash["panels"]) >= 12 and len({p["id"] for p in dash["panels"]}) == len(dash["panels"])
    rules = yaml.safe_load(open(os.path.join(ROOT, "deploy", "prometheus-alerts.yaml")))["groups"][0]["rules"]
    assert {r["labels"]["severity"] for r in rules} == {"minor", "serious", "critical"}
    exprs += [r["expr"] for r in rules]
    used = set(re.findall(r"klm_[a-z0-9_]+", " ".join(exprs)))
    assert used and used <= exported, sorted(used - exported)
    # the three HBM tiers are the reference's 70 / 85 / 95
    assert [r["expr"].split(">")[-1].strip() for r in rules if r["alert"].startswith("GpuHbm") and "Hot" not in r["alert"]] == ["70", "85", "95"]
