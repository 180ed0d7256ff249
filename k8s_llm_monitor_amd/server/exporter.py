"""``GET /metrics``: the monitor's own state in Prometheus text format.

The reference has no metrics endpoint of its own ("Metrics / logging");
with this one the monitor doubles as a light MI355X exporter (per-GPU HBM, activity, power,
temperature, xGMI traffic, ECC) and exposes its own sampling cost, which is one of the
self-defined baseline figures (sampling overhead).
"""
from __future__ import annotations

from typing import Any, Dict, List, Mapping, Optional


def _esc(v: Any) -> str:
    return str(v).replace("\\", "\\\\").replace('"', '\\"').replace("\n", "\\n")


class _Doc:
    def __init__(self) -> None:
        self.lines: List[str] = []
        self._seen = set()

    def add(self, name: str, help_: str, typ: str, value: Optional[float], labels: Optional[Mapping[str, Any]] = None) -> None:
        if value is None:
            return
        try:
            v = float(value)
        except (TypeError, ValueError):
            return
        if name not in self._seen:
            self._seen.add(name)
            self.lines.append(f"# HELP {name} {help_}")
            self.lines.append(f"# TYPE {name} {typ}")
        lab = ""
        if labels:
            lab = "{" + ",".join(f'{k}="{_esc(x)}"' for k, x in labels.items() if x is not None) + "}"
        self.lines.append(f"{name}{lab} {v!r}")

    def text(self) -> str:
        return "\n".join(self.lines) + "\n"


def render(snapshot: Mapping[str, Any], stats: Mapping[str, Any]) -> str:
    d = _Doc()
    host = snapshot.get("host") or {}

    def pf(x):
        try:
            return float(x)
        except (TypeError, ValueError):
            return None

    d.add("klm_host_cpu_percent", "Host CPU percent as shown on the dashboard.", "gauge", pf((host.get("cpu") or {}).get("percent")))
    d.add("klm_host_load1", "1-minute load average.", "gauge", (host.get("cpu") or {}).get("load_1min"))
    d.add("klm_host_memory_percent", "Host memory used percent.", "gauge", pf((host.get("memory") or {}).get("percent")))
    d.add("klm_host_disk_percent", "Root filesystem used percent.", "gauge", pf((host.get("disk") or {}).get("percent")))
    for g in snapshot.get("gpus") or []:
        L = {"gpu": g.get("index"), "uuid": g.get("uuid"), "name": g.get("name")}
        part = g.get("partition") or {}
        if part.get("compute") or part.get("memory"):
            d.add("klm_gpu_partition_info", "Compute / memory partition mode this device is a slice of (always 1).", "gauge", 1,
                  dict(L, compute=part.get("compute"), memory=part.get("memory")))
        bp = g.get("badPages")
        if bp:
            for st in ("retired", "pending", "unreservable"):
                d.add("klm_gpu_bad_pages", "HBM pages the driver retired (RAS), by status.", "gauge", bp.get(st), dict(L, status=st))
        d.add("klm_gpu_hbm_used_bytes", "HBM in use.", "gauge", g.get("hbmUsedBytes"), L)
        d.add("klm_gpu_hbm_total_bytes", "HBM capacity.", "gauge", g.get("hbmTotalBytes"), L)
        d.add("klm_gpu_gfx_activity_percent", "GFX activity, mean over the last sample period.", "gauge",
              g.get("gfxActivityAvg") if g.get("gfxActivityAvg") is not None else g.get("gfxActivity"), L)
        d.add("klm_gpu_mem_activity_percent", "Memory controller (UMC) activity.", "gauge",
              g.get("memActivityAvg") if g.get("memActivityAvg") is not None else g.get("memActivity"), L)
        d.add("klm_gpu_power_watts", "Socket power.", "gauge", g.get("powerW"), L)
        d.add("klm_gpu_power_cap_watts", "Socket power cap.", "gauge", g.get("powerCapW"), L)
        d.add("klm_gpu_power_average_watts", "Socket power from the energy accumulator over the last sampling period.", "gauge", g.get("powerAvgW"), L)
        d.add("klm_gpu_energy_joules_total", "Socket energy accumulator.", "counter", g.get("energyJ"), L)
        d.add("klm_gpu_junction_temperature_celsius", "Junction (hotspot) temperature.", "gauge", g.get("temperature"), L)
        d.add("klm_gpu_hbm_temperature_celsius", "HBM temperature.", "gauge", g.get("temperatureHbm"), L)
        d.add("klm_gpu_gfx_clock_mhz", "Current GFX clock.", "gauge", g.get("clockGfxMHz"), L)
        ecc = g.get("ecc") or {}
        d.add("klm_gpu_ecc_correctable_total", "Correctable ECC errors.", "counter", ecc.get("correctable"), L)
        d.add("klm_gpu_ecc_uncorrectable_total", "Uncorrectable ECC errors.", "counter", ecc.get("uncorrectable"), L)
        d.add("klm_gpu_throttled", "1 while any throttle reason is active.", "gauge", 1.0 if (g.get("throttle") or {}).get("active") else 0.0, L)
        for x, v in enumerate(g.get("xcdActivityAvg") or g.get("xcdActivity") or []):
            d.add("klm_gpu_xcd_activity_percent", "GFX busy share of one XCD (compute die), mean over the last sample period.",
                  "gauge", v, dict(L, xcd=x))
        pcie = g.get("pcie") or {}
        d.add("klm_gpu_pcie_link_speed_gts", "Current PCIe link speed.", "gauge", pcie.get("linkSpeedGTs"), L)
        d.add("klm_gpu_pcie_link_width", "Current PCIe link width (lanes).", "gauge", pcie.get("linkWidth"), L)
        for name, metric in (("replay", "klm_gpu_pcie_replay_total"), ("l0Recovery", "klm_gpu_pcie_l0_recovery_total"),
                             ("nakSent", "klm_gpu_pcie_nak_sent_total"), ("nakReceived", "klm_gpu_pcie_nak_received_total")):
            d.add(metric, f"PCIe {name} events.", "counter", ((pcie.get("errors") or {}).get(name) or {}).get("count"), L)
        for link in (g.get("xgmi") or {}).get("links") or []:
            LL = dict(L, link=link.get("link"))
            d.add("klm_gpu_xgmi_rx_bytes_total", "xGMI bytes received on the link.", "counter", link.get("rxBytes"), LL)
            d.add("klm_gpu_xgmi_tx_bytes_total", "xGMI bytes sent on the link.", "counter", link.get("txBytes"), LL)
            d.add("klm_gpu_xgmi_link_up", "1 when the xGMI link is up.", "gauge", 1.0 if link.get("state") == "up" else 0.0, LL)
    pods = snapshot.get("pods") or []
    by: Dict[str, int] = {}
    for p in pods:
        by[p.get("status") or "Unknown"] = by.get(p.get("status") or "Unknown", 0) + 1
    for st, n in sorted(by.items()):
        d.add("klm_pods", "Pods by phase.", "gauge", n, {"phase": st})
    v = snapshot.get("vllm") or {}
    if v.get("available"):
        d.add("klm_vllm_ttft_p50_seconds", "vLLM time to first token, median.", "gauge", v.get("ttftP50"))
        d.add("klm_vllm_ttft_p99_seconds", "vLLM time to first token, p99.", "gauge", v.get("ttftP99"))
        d.add("klm_vllm_generation_tokens_per_second", "vLLM generated tokens per second.", "gauge", v.get("tokensPerSec"))
        d.add("klm_vllm_kv_cache_percent", "vLLM KV-cache usage.", "gauge", v.get("kvCachePercent"))
        d.add("klm_vllm_requests_running", "vLLM running requests.", "gauge", v.get("running"))
        d.add("klm_vllm_requests_waiting", "vLLM waiting requests.", "gauge", v.get("waiting"))
    alerts = snapshot.get("alerts") or {}
    for tier in ("minor", "serious", "critical"):
        d.add("klm_alerts", "Active alerts by tier.", "gauge", len(alerts.get(tier) or []), {"tier": tier})
    for name, t in (stats.get("tasks") or {}).items():
        L = {"task": name}
        d.add("klm_sampler_calls_total", "Sampler ticks.", "counter", t.get("calls"), L)
        d.add("klm_sampler_errors_total", "Sampler ticks that raised.", "counter", t.get("errors"), L)
        d.add("klm_sampler_mean_milliseconds", "Mean tick duration.", "gauge", t.get("mean_ms"), L)
        d.add("klm_sampler_max_milliseconds", "Longest tick.", "gauge", t.get("max_ms"), L)
        d.add("klm_sampler_duty_percent", "Share of wall time spent sampling.", "gauge", t.get("duty_percent"), L)
    d.add("klm_uptime_seconds", "Seconds since the monitor started.", "gauge", stats.get("uptime_s"))
    return d.text()
