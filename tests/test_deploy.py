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
