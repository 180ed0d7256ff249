"""device-metrics-exporter ``/metrics`` text rendered from the scripted fake GPUs, so the
exporter reader is exercised on the same scenarios as the amdsmi path (no reference counterpart)."""
from __future__ import annotations

from typing import Dict, List, Optional

from ..collectors.exporter import DEFAULT_NAMES
from .gpu import FakeGpuReader


class FakeExporter:
    def __init__(self, reader: Optional[FakeGpuReader] = None, names: Optional[Dict[str, str]] = None, hostname: str = "mi355x-node-0"):
        self.reader = reader or FakeGpuReader(8)
        self.names = dict(DEFAULT_NAMES)
        self.names.update(names or {})
        self.hostname = hostname

    def fetch(self, url: str = "") -> str:
        n = self.names
        L: List[str] = []
        for raw in self.reader.read():
            lab = 'gpu_id="%d",gpu_uuid="%s",card_model="%s",hostname="%s",pci_bus_id="%s"' % (
                raw["index"], raw["uuid"], raw["name"], self.hostname, raw["bdf"])
            def g(key: str, v, extra: str = "") -> None:
                L.append("%s{%s%s} %s" % (n[key], lab, extra, float(v)))
            g("gfx_activity", raw["gfx_activity"]); g("umc_activity", raw["umc_activity"])
            g("vram_used_mib", raw["vram_used_mib"]); g("vram_total_mib", raw["vram_total_mib"])
            g("temp_hotspot", raw["temp_hotspot"]); g("temp_mem", raw["temp_mem"])
            g("power_w", raw["power_w"]); g("power_cap_w", raw["power_cap_w"])
            g("clk_gfx", raw["clk_gfx"], ',clock_type="GPU_CLOCK_TYPE_SYSTEM"')
            g("clk_gfx", raw["clk_mem"], ',clock_type="GPU_CLOCK_TYPE_MEMORY"')
            g("energy_uj", raw["energy_acc"] * raw["energy_resolution_uj"])
            g("ecc_correctable", raw["ecc_correctable"]); g("ecc_uncorrectable", raw["ecc_uncorrectable"])
            for l, st in enumerate(raw["xgmi_link_status"]):
                if st == 1:
                    g("xgmi_rx_bytes", raw["xgmi_read_kib"][l] * 1024, ',link_index="%d"' % l)
                    g("xgmi_tx_bytes", raw["xgmi_write_kib"][l] * 1024, ',link_index="%d"' % l)
        return "# HELP %s GFX activity\n# TYPE %s gauge\n" % (n["gfx_activity"], n["gfx_activity"]) + "\n".join(L) + "\n"
