"""Which pod owns the HBM: GPU process list -> Kubernetes pod.

No reference counterpart: the reference shows one memory number for GPU 0
(monitor_server.js:90). On a shared MI355X node the useful question is which vLLM pod holds
the 288 GB, so each GPU sample can carry its processes (pid, name, VRAM bytes) from
``amdsmi_get_gpu_process_list`` and every process is mapped to its pod through the pod UID in
``/proc/<pid>/cgroup`` (works when the monitor shares the host PID namespace, ``hostPID: true``).

cgroup path forms handled:
  cgroupfs driver  /kubepods/burstable/pod0f3a...-...-.../<container id>
  systemd driver   /kubepods.slice/kubepods-burstable.slice/kubepods-burstable-pod0f3a..._...slice/cri-containerd-<id>.scope
"""
from __future__ import annotations

import re
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional

_POD_UID = re.compile(r"pod([0-9a-fA-F]{8}[-_][0-9a-fA-F]{4}[-_][0-9a-fA-F]{4}[-_][0-9a-fA-F]{4}[-_][0-9a-fA-F]{12})")


def pod_uid_from_cgroup(text: str) -> Optional[str]:
    m = _POD_UID.search(text or "")
    return m.group(1).replace("_", "-").lower() if m else None


def _read_cgroup(pid: int) -> str:
    with open(f"/proc/{int(pid)}/cgroup", "r", encoding="utf-8") as f:
        return f.read()


class PodResolver:
    """pid -> pod UID with a small cache (a pid's cgroup does not change while it lives)."""

    def __init__(self, read_cgroup: Callable[[int], str] = _read_cgroup, max_cache: int = 4096):
        self._read = read_cgroup
        self._cache: Dict[int, Optional[str]] = {}
        self._max = max_cache

    def uid_of(self, pid: int) -> Optional[str]:
        if pid in self._cache:
            return self._cache[pid]
        try:
            uid = pod_uid_from_cgroup(self._read(pid))
        except (OSError, ValueError):
            uid = None
        if len(self._cache) >= self._max:
            self._cache.clear()
        self._cache[pid] = uid
        return uid

    def forget_dead(self, live_pids: Iterable[int]) -> None:
        live = set(live_pids)
        for p in [p for p in self._cache if p not in live]:
            del self._cache[p]


def attribute(gpus: List[Dict[str, Any]], pods: Iterable[Mapping[str, Any]], resolver: PodResolver) -> List[Dict[str, Any]]:
    """Fill ``pod`` / ``podUid`` into every entry of ``gpu["processList"]`` and add
    ``gpu["podsHbm"]``: VRAM bytes per pod on that GPU, largest first."""
    by_uid = {str(p.get("uid")).lower(): p for p in pods if p.get("uid")}
    live: List[int] = []
    for g in gpus:
        procs = g.get("processList")
        if not procs:
            continue
        per_pod: Dict[str, int] = {}
        for pr in procs:
            pid = pr.get("pid")
            if not isinstance(pid, int):
                continue
            live.append(pid)
            uid = resolver.uid_of(pid)
            pod = by_uid.get(uid) if uid else None
            pr["podUid"] = uid
            pr["pod"] = f"{pod.get('namespace')}/{pod.get('name')}" if pod else None
            key = pr["pod"] or (f"pod:{uid}" if uid else f"host:{pr.get('name') or pid}")
            per_pod[key] = per_pod.get(key, 0) + int(pr.get("vramBytes") or 0)
        g["podsHbm"] = [{"owner": k, "vramBytes": v} for k, v in sorted(per_pod.items(), key=lambda kv: -kv[1])]
    resolver.forget_dead(live)
    return gpus
