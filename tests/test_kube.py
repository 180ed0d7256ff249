"""Pod mapping vs. getK8sPods (monitor_server.js:97-114) on PodList fixtures."""
import json
import os
import stat
import sys

import pytest

from k8s_llm_monitor_amd.collectors.kube import (KubeApiSource, KubectlSource, KubeSource, open_kube_source, pod_from_item,
                                                 pods_from_list)
from k8s_llm_monitor_amd.fakes.kube import FakeKube, PodSpec, pod_item

NOW = 1_700_000_000.0


def item(**kw):
    return pod_item(PodSpec("kube-system", "coredns-1", **kw), NOW)


def test_reference_fields():
    p = pod_from_item(item(restarts=1, age_s=3 * 3600 + 36 * 60 + 10), NOW)
    assert {k: p[k] for k in ("namespace", "name", "status", "restarts", "age")} == {
        "namespace": "kube-system", "name": "coredns-1", "status": "Running", "restarts": 1, "age": "3h 36m"}


def test_restarts_are_summed_over_containers_and_phase_defaults_to_unknown():
    it = item()
    it["status"]["containerStatuses"] = [{"restartCount": 2}, {"restartCount": 3}, {}]
    del it["status"]["phase"]
    p = pod_from_item(it, NOW)
    assert p["restarts"] == 5 and p["status"] == "Unknown"          # :102, :104


def test_pending_pod_without_start_time():
    p = pod_from_item(item(phase="Pending"), NOW)
    assert p["status"] == "Pending" and p["age"] == "0m" and p["restarts"] == 0      # :105


def test_additions_reason_gpu_request_vllm_detection():
    it = pod_item(PodSpec("llm", "vllm-llama3-8b-0", gpus=8, image="rocm/vllm:rocm7.0", waiting_reason="CrashLoopBackOff", restarts=7), NOW)
    p = pod_from_item(it, NOW)
    assert p["reason"] == "CrashLoopBackOff" and p["gpus"] == 8 and p["vllm"] is True and p["ready"] == "0/1"
    assert p["status"] == "Running"          # the phase hides the crash loop
    assert pod_from_item(item(), NOW)["reason"] is None and pod_from_item(item(), NOW)["vllm"] is False


def test_oomkilled_last_state():
    it = item()
    it["status"]["containerStatuses"][0]["lastState"] = {"terminated": {"reason": "OOMKilled", "exitCode": 137}}
    assert pod_from_item(it, NOW)["reason"] == "OOMKilled"


def test_list_and_failure_shape():
    assert pods_from_list({"items": []}) == []
    with pytest.raises(KeyError):
        pods_from_list({})
    fk = FakeKube()
    assert len(fk.pods()) == 11 and fk.last_error is None
    fk.fail = True
    assert fk.pods() == [] and "kubectl" in fk.last_error      # :113
    assert KubeSource().pods() == []


def test_kubectl_subprocess(tmp_path):
    doc = {"items": [item(restarts=4)]}
    exe = tmp_path / "kubectl"
    exe.write_text("#!%s\nimport sys, json\nassert sys.argv[1:] == ['get','pods','-A','-o','json'], sys.argv\nprint(%r)\n"
                   % (sys.executable, json.dumps(doc)))
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    src = KubectlSource(str(exe))
    pods = src.pods()
    assert len(pods) == 1 and pods[0]["restarts"] == 4 and src.last_error is None
    bad = KubectlSource(str(tmp_path / "missing"))
    assert bad.pods() == [] and bad.last_error


def test_kubectl_timeout(tmp_path):
    exe = tmp_path / "kubectl"
    exe.write_text("#!/bin/sh\nsleep 5\n")
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    src = KubectlSource(str(exe), timeout=0.2)
    assert src.pods() == [] and "Timeout" in src.last_error


def test_api_source_sends_bearer_token(tmp_path):
    import threading
    from http.server import BaseHTTPRequestHandler, HTTPServer
    seen = {}

    class H(BaseHTTPRequestHandler):
        def log_message(self, *a):
            pass

        def do_GET(self):
            seen["path"], seen["auth"] = self.path, self.headers.get("Authorization")
            body = json.dumps({"items": [item()]}).encode()
            self.send_response(200)
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

    srv = HTTPServer(("127.0.0.1", 0), H)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    tok = tmp_path / "token"
    tok.write_text("s3cret\n")
    try:
        src = KubeApiSource(f"http://127.0.0.1:{srv.server_address[1]}", token_file=str(tok))
        assert len(src.pods()) == 1 and seen == {"path": "/api/v1/pods?limit=500&resourceVersion=0", "auth": "Bearer s3cret"}
    finally:
        srv.shutdown()


def test_api_source_walks_pages_and_survives_an_expired_continue_token(tmp_path, monkeypatch):
    """the pod list is read in pages (limit / continue), the first from the watch cache; a continue token
    that expired mid-walk (410 Gone) restarts the walk once; an IPv6 service host is bracketed."""
    import threading
    from http.server import BaseHTTPRequestHandler, HTTPServer
    from urllib.parse import parse_qs, urlsplit
    from k8s_llm_monitor_amd.collectors.kube import api_host_url
    pods = [dict(item(), metadata={"namespace": "llm", "name": f"pod-{i}"}) for i in range(23)]
    state = {"requests": [], "expire_once": False}

    class H(BaseHTTPRequestHandler):
        def log_message(self, *a):
            pass

        def do_GET(self):
            q = {k: v[0] for k, v in parse_qs(urlsplit(self.path).query).items()}
            state["requests"].append(q)
            start = int(q.get("continue", "0"))
            if start and state["expire_once"]:
                state["expire_once"] = False
                self.send_response(410)
                self.send_header("Content-Length", "0")
                self.end_headers()
                return
            limit = int(q.get("limit", "100000"))
            page = pods[start:start + limit]
            meta = {"continue": str(start + limit)} if start + limit < len(pods) else {}
            body = json.dumps({"kind": "PodList", "metadata": meta, "items": page}).encode()
            self.send_response(200)
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

    srv = HTTPServer(("127.0.0.1", 0), H)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    try:
        src = KubeApiSource(f"http://127.0.0.1:{srv.server_address[1]}", page_size=10)
        got = src.pods()
        assert [p["name"] for p in got] == [f"pod-{i}" for i in range(23)] and src.last_pages == 3
        assert state["requests"] == [{"limit": "10", "resourceVersion": "0"}, {"limit": "10", "continue": "10"}, {"limit": "10", "continue": "20"}]
        state["requests"].clear()
        state["expire_once"] = True
        assert len(src.pods()) == 23 and len(state["requests"]) == 5 and src.last_error is None      # page, 410, then a fresh walk of three
        # a page cap bounds the walk (and says the list is cut)
        capped = KubeApiSource(f"http://127.0.0.1:{srv.server_address[1]}", page_size=10, max_pages=2)
        assert len(capped.pods()) == 20 and capped.fetch()["truncated"] is True
    finally:
        srv.shutdown()
    assert api_host_url("10.43.0.1") == "https://10.43.0.1:443" and api_host_url("fd00:10:96::1", "6443") == "https://[fd00:10:96::1]:6443"
    assert api_host_url("[fd00::1]") == "https://[fd00::1]:443"
    tok = tmp_path / "token"
    tok.write_text("t")
    monkeypatch.setenv("KUBERNETES_SERVICE_HOST", "fd00:10:96::1")
    monkeypatch.setenv("KUBERNETES_SERVICE_PORT", "443")

    class Cfg:
        class sources:
            kubectl, kube_api, kube_token_file, kube_ca_file = "kubectl", "", str(tok), ""
    assert open_kube_source("auto", Cfg).url == "https://[fd00:10:96::1]:443/api/v1/pods"


def test_open_auto_without_cluster(monkeypatch):
    monkeypatch.delenv("KUBERNETES_SERVICE_HOST", raising=False)
    monkeypatch.setenv("PATH", "/nonexistent")
    assert open_kube_source("auto").kind == "none"
    assert open_kube_source("fake").kind == "fake"
    with pytest.raises(ValueError):
        open_kube_source("bogus")


def test_lifecycle_scenario(clock):
    fk = FakeKube("pod_lifecycle", clock=clock, wall=lambda: NOW)
    names = lambda: {p["name"]: p for p in fk.pods()}
    assert "vllm-llama3-8b-canary-0" not in names()
    clock.advance(6)
    assert names()["vllm-llama3-8b-canary-0"]["status"] == "Pending"
    clock.advance(10)
    assert names()["vllm-llama3-8b-canary-0"]["status"] == "Running"
    clock.advance(15)
    assert names()["vllm-llama3-8b-canary-0"]["restarts"] == 1
    clock.advance(10)
    assert names()["vllm-llama3-8b-canary-0"]["status"] == "Failed"
