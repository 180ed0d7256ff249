"""Canned and scripted Kubernetes pod lists (a K3s cluster running vLLM-ROCm).

``FakeKube`` returns real ``PodList`` documents (the JSON ``kubectl get pods -A -o json``
prints), so the production mapping code (collectors.kube.pods_from_list) runs on them
unchanged. BASELINE.json configs 1, 4 and 5; the reference has no fakes.

Scenarios are functions ``t_seconds -> list of PodSpec``.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from datetime import datetime, timezone
from typing import Any, Callable, Dict, List, Optional

from ..collectors.kube import KubeSource


@dataclass
class PodSpec:
    namespace: str
    name: str
    phase: str = "Running"
    restarts: int = 0
    age_s: float = 3 * 3600 + 36 * 60
    waiting_reason: Optional[str] = None
    gpus: int = 0
    image: str = "registry.k8s.io/pause:3.9"
    node: str = "mi355x-node-0"
    ready: bool = True


def _iso(epoch: float) -> str:
    return datetime.fromtimestamp(epoch, tz=timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ")


def fake_uid(namespace: str, name: str) -> str:
    """A stable UUID-shaped uid per pod name."""
    import hashlib
    h = hashlib.md5(f"{namespace}/{name}".encode()).hexdigest()
    return f"{h[:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:32]}"


def pod_item(p: PodSpec, now: float) -> Dict[str, Any]:
    state: Dict[str, Any]
    if p.waiting_reason:
        state = {"waiting": {"reason": p.waiting_reason, "message": "back-off restarting failed container"}}
    elif p.phase == "Running":
        state = {"running": {"startedAt": _iso(now - p.age_s)}}
    elif p.phase == "Succeeded":
        state = {"terminated": {"reason": "Completed", "exitCode": 0}}
    elif p.phase == "Failed":
        state = {"terminated": {"reason": "Error", "exitCode": 1}}
    else:
        state = {}
    container: Dict[str, Any] = {"name": "main", "image": p.image}
    if p.gpus:
        container["resources"] = {"limits": {"amd.com/gpu": str(p.gpus)}, "requests": {"amd.com/gpu": str(p.gpus)}}
    item: Dict[str, Any] = {
        "apiVersion": "v1", "kind": "Pod",
        "metadata": {"namespace": p.namespace, "name": p.name, "uid": fake_uid(p.namespace, p.name),
                     "labels": {"app": p.name.rsplit("-", 2)[0]}},
        "spec": {"nodeName": p.node, "containers": [container]},
        "status": {"phase": p.phase},
    }
    if p.phase != "Pending" or p.waiting_reason:
        item["status"]["startTime"] = _iso(now - p.age_s)
        item["status"]["containerStatuses"] = [{
            "name": "main", "ready": p.ready and p.phase == "Running" and not p.waiting_reason,
            "restartCount": p.restarts, "state": state}]
    return item


def base_pods() -> List[PodSpec]:
    """What a small K3s + vLLM-ROCm install looks like."""
    return [
        PodSpec("kube-system", "coredns-6799fbcd5-x7k2p", restarts=1),
        PodSpec("kube-system", "local-path-provisioner-6c86858495-4jq9m"),
        PodSpec("kube-system", "metrics-server-54fd9b65b-tz8wl"),
        PodSpec("kube-system", "traefik-7d5f6474df-9hbn4", age_s=2 * 86400 + 5 * 3600),
        PodSpec("kube-system", "amdgpu-device-plugin-daemonset-k5x8d", image="rocm/k8s-device-plugin:latest"),
        PodSpec("monitoring", "prometheus-0", image="prom/prometheus:v2.53.0"),
        PodSpec("monitoring", "node-exporter-5m4zq", image="prom/node-exporter:v1.8.1"),
        PodSpec("monitoring", "amd-device-metrics-exporter-8qk2c", image="rocm/device-metrics-exporter:v1.3.0"),
        PodSpec("monitoring", "kube-state-metrics-7c6d8b9f4-hw2xn"),
        PodSpec("llm", "vllm-llama3-8b-7f9c6d5b8-2xkzv", gpus=1, image="rocm/vllm:rocm7.0_mi355", age_s=5 * 3600 + 12 * 60),
        PodSpec("llm", "vllm-llama3-70b-tp8-0", gpus=8, image="rocm/vllm:rocm7.0_mi355", age_s=26 * 3600),
    ]


Scenario = Callable[[float], List[PodSpec]]


def _steady(t: float) -> List[PodSpec]:
    return base_pods()


def _pod_lifecycle(t: float) -> List[PodSpec]:
    """A new vLLM replica: absent (t<5) -> Pending (5..15) -> Running (15..30) -> restart (30..40)
    -> Failed (t>=40). Exercises every pod rule of the alert engine."""
    pods = base_pods()
    if t >= 5:
        p = PodSpec("llm", "vllm-llama3-8b-canary-0", gpus=1, image="rocm/vllm:rocm7.0_mi355", age_s=max(0.0, t - 5))
        if t < 15:
            p.phase = "Pending"
        elif t < 30:
            p.phase = "Running"
        elif t < 40:
            p.phase, p.restarts = "Running", 1
        else:
            p.phase, p.restarts = "Failed", 1
        pods.append(p)
    return pods


def _crashloop(t: float) -> List[PodSpec]:
    pods = base_pods()
    pods.append(PodSpec("llm", "vllm-qwen-72b-0", gpus=4, image="rocm/vllm:rocm7.0_mi355", phase="Running",
                        restarts=int(t // 10), waiting_reason="CrashLoopBackOff" if t >= 10 else None, age_s=600 + t))
    return pods


def _pending(t: float) -> List[PodSpec]:
    pods = base_pods()
    pods.append(PodSpec("llm", "vllm-llama3-405b-tp8-0", gpus=8, phase="Pending", image="rocm/vllm:rocm7.0_mi355"))
    return pods


def _empty(t: float) -> List[PodSpec]:
    return []


SCENARIOS: Dict[str, Scenario] = {"": _steady, "steady": _steady, "pod_lifecycle": _pod_lifecycle,
                                  "crashloop": _crashloop, "pending": _pending, "empty": _empty}


class FakeKube(KubeSource):
    kind = "fake"

    def __init__(self, scenario: Any = "", clock: Callable[[], float] = time.monotonic,
                 wall: Callable[[], float] = time.time):
        super().__init__()
        if callable(scenario):
            self.scenario: Scenario = scenario
            self.scenario_name = getattr(scenario, "__name__", "custom")
        else:
            # GPU-only scenario names (hbm_ramp, thermal, ...) mean "steady" for the pods
            self.scenario = SCENARIOS.get(scenario, _steady)
            self.scenario_name = scenario if scenario in SCENARIOS and scenario else "steady"
        self.clock, self.wall = clock, wall
        self.t0 = clock()
        self.fail = False          # set to make fetch() raise, like a missing kubectl

    def set_scenario(self, scenario: Any) -> None:
        self.scenario = scenario if callable(scenario) else SCENARIOS[scenario]
        self.scenario_name = getattr(scenario, "__name__", str(scenario))
        self.t0 = self.clock()

    def fetch(self) -> Dict[str, Any]:
        if self.fail:
            raise FileNotFoundError("kubectl")
        now = self.wall()
        t = self.clock() - self.t0
        return {"apiVersion": "v1", "kind": "List", "items": [pod_item(p, now) for p in self.scenario(t)]}

    def describe(self) -> str:
        return f"fake(scenario={self.scenario_name})"
