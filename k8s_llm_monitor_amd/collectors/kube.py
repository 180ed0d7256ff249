"""Kubernetes pod collector.

Reference: ``getK8sPods`` (monitor_server.js:97-114) shells ``kubectl get pods -A -o json``
on the request path and maps every item to ``{namespace, name, status, restarts, age}``:
``status = phase || 'Unknown'`` (:102), ``restarts`` = sum of
``containerStatuses[].restartCount`` (:104), ``age`` from ``status.startTime`` (:106-110),
and ``[]`` on any failure (:113).

Here the mapping is a pure function (``pods_from_list``) shared by three sources:
  * ``KubectlSource``  - the same subprocess, but with a timeout and off the request path
    (the sampler thread calls it; SURVEY.md 2.8 "blocking collectors");
  * ``KubeApiSource``  - the API server over HTTPS with the in-cluster service-account token;
  * ``fakes.kube.FakeKube`` - canned / scripted pod lists for CPU-only runs and tests.

Added fields (never renamed ones): ``reason`` (container waiting /
terminated reason such as CrashLoopBackOff, which the phase hides),
``ready`` ("1/2"), ``node``, ``gpus`` (``amd.com/gpu`` requested) and ``vllm`` (pod looks like
a vLLM server).
"""
from __future__ import annotations

import json
import ssl
import subprocess
import time
import urllib.error
import urllib.request
from typing import Any, Dict, List, Mapping, Optional, Sequence

from ..utils.timefmt import format_age, parse_k8s_time

GPU_RESOURCE = "amd.com/gpu"


def _container_reason(statuses: Sequence[Mapping[str, Any]]) -> Optional[str]:
    """First non-trivial waiting reason, else a terminated-with-error reason."""
    for cs in statuses:
        w = (cs.get("state") or {}).get("waiting") or {}
        r = w.get("reason")
        if r and r not in ("ContainerCreating", "PodInitializing"):
            return str(r)
    for cs in statuses:
        t = (cs.get("state") or {}).get("terminated") or {}
        r = t.get("reason")
        if r and r != "Completed" and t.get("exitCode", 0) != 0:
            return str(r)
    for cs in statuses:
        t = (cs.get("lastState") or {}).get("terminated") or {}
        if t.get("reason") == "OOMKilled":
            return "OOMKilled"
    return None


def _gpu_request(spec: Mapping[str, Any]) -> int:
    n = 0
    for c in spec.get("containers") or []:
        res = c.get("resources") or {}
        v = (res.get("limits") or {}).get(GPU_RESOURCE) or (res.get("requests") or {}).get(GPU_RESOURCE)
        try:
            n += int(v) if v is not None else 0
        except (TypeError, ValueError):
            pass
    return n


def _looks_like_vllm(item: Mapping[str, Any]) -> bool:
    meta = item.get("metadata") or {}
    labels = meta.get("labels") or {}
    text = " ".join([str(meta.get("name", "")), *(str(v) for v in labels.values())]).lower()
    if "vllm" in text:
        return True
    for c in (item.get("spec") or {}).get("containers") or []:
        if "vllm" in str(c.get("image", "")).lower():
            return True
        if any("vllm" in str(a).lower() for a in (c.get("command") or []) + (c.get("args") or [])):
            return True
    return False


def pod_from_item(item: Mapping[str, Any], now: Optional[float] = None) -> Dict[str, Any]:
    meta = item.get("metadata") or {}
    status = item.get("status") or {}
    spec = item.get("spec") or {}
    cs = status.get("containerStatuses") or []
    restarts = 0
    for c in cs:
        restarts += int(c.get("restartCount") or 0)
    ready = sum(1 for c in cs if c.get("ready"))
    total = len(spec.get("containers") or cs)
    return {
        "namespace": meta.get("namespace"),
        "name": meta.get("name"),
        "status": status.get("phase") or "Unknown",
        "restarts": restarts,
        "age": format_age(parse_k8s_time(status.get("startTime")), now),
        # additions
        "reason": _container_reason(list(status.get("initContainerStatuses") or []) + list(cs)),
        "ready": f"{ready}/{total}",
        "node": spec.get("nodeName"),
        "gpus": _gpu_request(spec),
        "vllm": _looks_like_vllm(item),
        "uid": meta.get("uid"),
    }


def pods_from_list(data: Mapping[str, Any], now: Optional[float] = None) -> List[Dict[str, Any]]:
    """A ``PodList`` object (kubectl -o json / GET /api/v1/pods) -> API pods.

    Raises on a document without ``items``, like ``data.items.map`` on undefined; sources
    catch that and return ``[]`` (:113)."""
    items = data["items"]
    return [pod_from_item(it, now) for it in items]


class KubeSource:
    kind = "none"

    def __init__(self):
        self.last_error: Optional[str] = None
        self.last_duration_s = 0.0

    def fetch(self) -> Mapping[str, Any]:
        return {"items": []}

    def pods(self) -> List[Dict[str, Any]]:
        t0 = time.perf_counter()
        try:
            out = pods_from_list(self.fetch())
            self.last_error = None
        except Exception as e:   # the reference's catch-all (:113)
            out = []
            self.last_error = f"{type(e).__name__}: {e}"
        self.last_duration_s = time.perf_counter() - t0
        return out

    def describe(self) -> str:
        return self.kind

    def close(self) -> None:
        pass


class KubectlSource(KubeSource):
    kind = "kubectl"

    def __init__(self, kubectl: str = "kubectl", timeout: float = 8.0, extra_args: Sequence[str] = ()):
        super().__init__()
        self.argv = [kubectl, "get", "pods", "-A", "-o", "json", *extra_args]
        self.timeout = timeout

    def fetch(self) -> Mapping[str, Any]:
        r = subprocess.run(self.argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=self.timeout, check=True)
        return json.loads(r.stdout.decode("utf-8"))

    def describe(self) -> str:
        return "kubectl(%s)" % self.argv[0]


def api_host_url(host: str, port: str = "443") -> str:
    """``https://host:port`` for the in-cluster service variables; an IPv6 literal (KUBERNETES_SERVICE_HOST on an IPv6
    or dual-stack cluster: ``fd00::1``) is bracketed, as a URL needs it."""
    h = host.strip()
    if ":" in h and not h.startswith("["):
        h = "[" + h + "]"
    return "https://%s:%s" % (h, port)


class KubeApiSource(KubeSource):
    """``GET {api}/api/v1/pods`` with a bearer token (in-cluster service account by default).

    The list is read in PAGES (``limit`` / ``continue``, 500 pods each by default), so a large cluster is never one
    multi-hundred-megabyte response, and from the API server's watch cache (``resourceVersion=0``: any recent version, no
    quorum read from etcd every ten seconds). A continue token that expired between pages (HTTP 410) restarts the walk
    once; the reference has nothing comparable (it shells out to ``kubectl get pods -A -o json``, monitor_server.js:99)."""
    kind = "api"

    def __init__(self, api: str, token_file: str = "", ca_file: str = "", timeout: float = 8.0,
                 insecure: bool = False, page_size: int = 500, max_pages: int = 400):
        super().__init__()
        self.url = api.rstrip("/") + "/api/v1/pods"
        self.token_file, self.ca_file, self.timeout = token_file, ca_file, timeout
        self.page_size, self.max_pages = int(page_size), int(max_pages)
        self.last_pages = 0
        if self.url.startswith("https"):
            if insecure:
                self._ctx: Optional[ssl.SSLContext] = ssl._create_unverified_context()
            else:
                self._ctx = ssl.create_default_context(cafile=ca_file or None)
        else:
            self._ctx = None

    def _token(self) -> str:
        if not self.token_file:
            return ""
        try:
            with open(self.token_file, "r", encoding="utf-8") as f:
                return f.read().strip()   # re-read each time: projected tokens rotate
        except OSError:
            return ""

    def _get(self, url: str) -> Mapping[str, Any]:
        req = urllib.request.Request(url, headers={"Accept": "application/json"})
        tok = self._token()
        if tok:
            req.add_header("Authorization", "Bearer " + tok)
        with urllib.request.urlopen(req, timeout=self.timeout, context=self._ctx) as resp:
            return json.loads(resp.read().decode("utf-8"))

    def fetch(self) -> Mapping[str, Any]:
        import urllib.error
        import urllib.parse
        for attempt in (0, 1):
            items: list = []
            cont, pages = "", 0
            try:
                while True:
                    q = {"limit": str(self.page_size)} if self.page_size > 0 else {}
                    if cont:
                        q["continue"] = cont
                    else:
                        q["resourceVersion"] = "0"                  # the first page may come from the watch cache
                    doc = self._get(self.url + ("?" + urllib.parse.urlencode(q) if q else ""))
                    items.extend(doc.get("items") or [])
                    pages += 1
                    cont = ((doc.get("metadata") or {}).get("continue") or "") if self.page_size > 0 else ""
                    if not cont or pages >= self.max_pages:
                        self.last_pages = pages
                        return {"kind": "PodList", "items": items, "truncated": bool(cont)}
            except urllib.error.HTTPError as e:
                if e.code == 410 and attempt == 0:                 # the continue token expired: one fresh walk
                    continue
                raise
        raise RuntimeError("unreachable")

    def describe(self) -> str:
        return f"api({self.url})"


def open_kube_source(kind: str = "auto", cfg=None) -> KubeSource:
    """auto: the API server when a service-account token is mounted or ``kube_api`` is set,
    else ``kubectl`` when it is on PATH, else none (``[]`` like the reference without kubectl)."""
    import os
    import shutil
    src = getattr(cfg, "sources", None)
    kubectl = getattr(src, "kubectl", "kubectl")
    api = getattr(src, "kube_api", "")
    token = getattr(src, "kube_token_file", "")
    ca = getattr(src, "kube_ca_file", "")
    kind = (kind or "auto").lower()
    if kind == "none":
        return KubeSource()
    if kind == "fake":
        from ..fakes.kube import FakeKube
        return FakeKube(scenario=getattr(src, "fake_scenario", "") or "")
    if kind == "kubectl":
        return KubectlSource(kubectl)
    if kind == "api":
        return KubeApiSource(api or "https://kubernetes.default.svc", token, ca if os.path.exists(ca) else "")
    if kind == "auto":
        if api or (token and os.path.exists(token) and os.environ.get("KUBERNETES_SERVICE_HOST")):
            host = api or api_host_url(os.environ["KUBERNETES_SERVICE_HOST"], os.environ.get("KUBERNETES_SERVICE_PORT", "443"))
            return KubeApiSource(host, token, ca if os.path.exists(ca) else "")
        if shutil.which(kubectl):
            return KubectlSource(kubectl)
        return KubeSource()
    raise ValueError(f"unknown kube source {kind!r}")
