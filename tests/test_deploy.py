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
