"""GPU telemetry from AMD device-metrics-exporter (Prometheus text), for a monitor that does
not run on the GPU node itself.

The reference reads NVIDIA's dcgm-exporter through Prometheus for its history charts
(``DCGM_FI_DEV_GPU_UTIL`` / ``_FB_USED`` / ``_GPU_TEMP``, monitor_server.js:128-134). This is
the AMD counterpart (BASELINE.json north_star item 1: "amdsmi / AMD device-metrics-exporter"):
``ExporterGpuReader`` scrapes the exporter's ``/metrics`` endpoint directly and produces the
same RAW readings as the amdsmi reader, so normalisation, alerts and the API are shared.

Metric names are configuration (``Config.exporter``): the defaults follow the exporter's
published field names but could not be checked against a running exporter in the build
environment (no network, no exporter) - treat them as unverified and override
them per deployment. Counters that the exporter publishes as gauges of cumulative values
(energy, ECC, xGMI bytes) go through the same delta arithmetic as amdsmi's accumulators.
"""
from __future__ import annotations

import time
import urllib.request
from typing import Any, Callable, Dict, List, Mapping, Optional

from .gpu import GpuReader
from .vllm import parse_exposition

#: raw-reading key -> default exporter metric name
DEFAULT_NAMES: Dict[str, str] = {
    "gfx_activity": "gpu_gfx_activity",
    "umc_activity": "gpu_umc_activity",
    "vram_used_mib": "gpu_used_vram",
    "vram_total_mib": "gpu_total_vram",
    "temp_hotspot": "gpu_junction_temperature",
    "temp_mem": "gpu_memory_temperature",
    "temp_edge": "gpu_edge_temperature",
    "power_w": "gpu_package_power",
    "power_avg_w": "gpu_average_package_power",
    "power_cap_w": "gpu_power_usage_limit",
    "clk_gfx": "gpu_clock",                     # with label clock_type="GPU_CLOCK_TYPE_SYSTEM"
    "energy_uj": "gpu_energy_consumed",
    "ecc_correctable": "gpu_ecc_correct_total",
    "ecc_uncorrectable": "gpu_ecc_uncorrect_total",
    "xgmi_rx_bytes": "gpu_xgmi_link_rx",        # per link, label link_index
    "xgmi_tx_bytes": "gpu_xgmi_link_tx",
    "pcie_bandwidth": "pcie_bandwidth",
}
GPU_LABELS = ("gpu_id", "gpu", "gpu_index")
LINK_LABELS = ("link_index", "link", "xgmi_link")


def _gpu_of(labels: Mapping[str, str]) -> Optional[int]:
    for k in GPU_LABELS:
        if k in labels:
            try:
                return int(labels[k])
            except ValueError:
                return None
    return None


class ExporterGpuReader(GpuReader):
    kind = "exporter"

    def __init__(self, url: str, names: Optional[Mapping[str, str]] = None, timeout: float = 3.0,
                 fetch: Optional[Callable[[str], str]] = None, clock: Callable[[], float] = time.monotonic):
        self.url = url if url.rstrip("/").endswith("/metrics") else url.rstrip("/") + "/metrics"
        self.names = dict(DEFAULT_NAMES)
        self.names.update({k: v for k, v in (names or {}).items() if v})
        self.timeout = timeout
        self._fetch = fetch or self._http
        self._clock = clock

    def _http(self, url: str) -> str:
        with urllib.request.urlopen(url, timeout=self.timeout) as r:
            return r.read().decode("utf-8", "replace")

    def describe(self) -> str:
        return f"exporter({self.url})"

    def read(self) -> List[Dict[str, Any]]:
        samples = parse_exposition(self._fetch(self.url))
        now = self._clock()
        by_name: Dict[str, str] = {v: k for k, v in self.names.items()}
        gpus: Dict[int, Dict[str, Any]] = {}
        for name, labels, value in samples:
            key = by_name.get(name)
            if key is None:
                continue
            g = _gpu_of(labels)
            if g is None:
                continue
            raw = gpus.setdefault(g, {"index": g, "t": now, "wall": time.time(), "xgmi_read_kib": [None] * 8,
                                      "xgmi_write_kib": [None] * 8, "xgmi_link_status": [None] * 8})
            raw.setdefault("name", labels.get("card_model") or labels.get("card_series") or labels.get("gpu_model"))
            raw.setdefault("uuid", labels.get("gpu_uuid") or labels.get("uuid"))
            raw.setdefault("bdf", labels.get("pci_bus_id") or labels.get("bdf"))
            raw.setdefault("node", labels.get("hostname") or labels.get("node"))
            if key == "clk_gfx":
                ct = labels.get("clock_type", "")
                if "SYSTEM" in ct or "GFX" in ct or not ct:
                    raw["clk_gfx"] = value
                elif "MEMORY" in ct:
                    raw["clk_mem"] = value
            elif key in ("xgmi_rx_bytes", "xgmi_tx_bytes"):
                link = next((labels[k] for k in LINK_LABELS if k in labels), None)
                try:
                    li = int(link)
                except (TypeError, ValueError):
                    continue
                if 0 <= li < 8:
                    raw["xgmi_read_kib" if key == "xgmi_rx_bytes" else "xgmi_write_kib"][li] = int(value // 1024)
                    raw["xgmi_link_status"][li] = 1
            elif key == "energy_uj":
                raw["energy_acc"], raw["energy_resolution_uj"] = int(value), 1.0
            elif key == "power_avg_w":
                raw.setdefault("power_w", value)
            elif key == "pcie_bandwidth":
                raw["pcie_bandwidth_inst"] = value
            else:
                raw[key] = value
        out = []
        for g in sorted(gpus):
            raw = gpus[g]
            if all(v is None for v in raw["xgmi_read_kib"]):
                raw.pop("xgmi_read_kib"), raw.pop("xgmi_write_kib"), raw.pop("xgmi_link_status")
            for k in ("ecc_correctable", "ecc_uncorrectable", "vram_used_mib", "vram_total_mib"):
                if k in raw:
                    raw[k] = int(raw[k])
            out.append(raw)
        return out
