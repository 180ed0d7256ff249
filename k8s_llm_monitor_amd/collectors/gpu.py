"""GPU telemetry for AMD Instinct MI355X through amdsmi. No NVIDIA path of any kind.

Reference: ``getGpuMetrics`` (monitor_server.js:83-95) shells out to ``nvidia-smi`` and
returns ``[{name, utilization, memoryUsed, memoryTotal, temperature}]`` with memory in MiB
and ``[]`` on any failure. This module keeps exactly those five keys with the same units
(so the reference frontend would still render it) and adds, per GPU: ``index`` (parsed but
dropped by the reference, :90), HBM bytes and percent, memory-controller activity, power
and cap, HBM temperature, clocks, ECC counts, throttle residency, and per-link xGMI
traffic.

Layers:
  raw reading  - flat dict per GPU straight from amdsmi (``AmdSmiReader``), from the C++
                 sampler's JSON lines (``NativeSamplerReader``) or from a script
                 (``fakes.gpu.FakeGpuReader``). Accumulators are still cumulative here.
  GpuNormalizer - turns raw readings into API samples; owns the delta state.
  GpuSource     - reader + normalizer, what the sampler loop talks to.
"""
from __future__ import annotations

import json
import os
import subprocess
import threading
import time
from typing import Any, Dict, List, Optional, Sequence

from .counters import RateTracker, counter_delta, share

MIB = 1 << 20
XGMI_LINKS = 8
ACT_ACC_BITS = 32

RESIDENCY_KEYS = ("prochot", "ppt", "socket_thm", "vr_thm", "hbm_thm")


def _num(x: Any) -> Optional[float]:
    """amdsmi's Python layer reports unsupported fields as the string 'N/A'."""
    if isinstance(x, bool):
        return float(x)
    if isinstance(x, (int, float)):
        return float(x)
    return None


def _int(x: Any) -> Optional[int]:
    if isinstance(x, int) and not isinstance(x, bool):
        return x                      # 64-bit ids and accumulators must not pass through a float
    v = _num(x)
    return None if v is None else int(v)


# --------------------------------------------------------------------------------------
# normaliser
# --------------------------------------------------------------------------------------
class GpuNormalizer:
    """Raw cumulative readings -> API samples. One instance per source (keeps last readings)."""

    def __init__(self):
        self._rates = RateTracker(bits=64)
        self._prev: Dict[int, Dict[str, Any]] = {}
        self._held: Dict[Any, Any] = {}        # last period figures, shown again while the metrics table has not moved

    def reset(self) -> None:
        self.__init__()

    def normalize(self, raw: Dict[str, Any]) -> Dict[str, Any]:
        idx = int(raw.get("index", 0))
        t = float(raw.get("t", time.monotonic()))
        prev = self._prev.get(idx)
        self._prev[idx] = raw
        # The driver refreshes the metrics table about once a second (measured on MI355X: 6 changes in 6 s at 20 Hz
        # polling, profiles/r2_validation/energy_timebase.json), so its accumulators belong to the table's OWN clock
        # (firmware_timestamp, 10 ns ticks: 100.6 M per host second), not to the host time of the read: against
        # the host clock a rate is off by up to one refresh per period, and a table read twice gives a zero.
        fw = _int(raw.get("firmware_timestamp"))
        tt = fw * 1e-8 if fw else t
        table_same = bool(prev is not None and fw and _int(prev.get("firmware_timestamp")) == fw)

        def held(key, value):
            """A period figure: remembered when it could be formed, repeated while the table is unchanged."""
            k = (key, idx)
            if value is not None:
                self._held[k] = value
                return value
            return self._held.get(k) if table_same else None

        total_mib = _int(raw.get("vram_total_mib")) or 0
        used_mib = _int(raw.get("vram_used_mib")) or 0
        used_b = _int(raw.get("vram_used_bytes"))
        total_b = _int(raw.get("vram_total_bytes"))
        if not used_b:
            used_b = used_mib * MIB
        if not total_b or (total_mib and abs(total_b - total_mib * MIB) > total_mib * MIB // 8):
            # amdsmi's byte-typed total includes carve-outs the MiB-typed one does not; keep
            # both fields of one API call consistent
            total_b = total_mib * MIB
        if not total_mib and total_b:
            total_mib = total_b // MIB
        if not used_mib and used_b:
            used_mib = used_b // MIB

        # activity: accumulator deltas give the exact mean over the sample period; the
        # instantaneous gauge is a firmware moving average and is kept beside it
        gfx_inst = _num(raw.get("gfx_activity"))
        umc_inst = _num(raw.get("umc_activity"))
        gfx_avg = umc_avg = None
        residency: Dict[str, Optional[float]] = {}
        if prev is not None:
            dacc = counter_delta(_int(prev.get("accumulation_counter")), _int(raw.get("accumulation_counter")))
            # gfx/mem activity accumulators are 32-bit in amdsmi_gpu_metrics_t (percent x ticks)
            gfx_avg = share(counter_delta(_int(prev.get("gfx_activity_acc")), _int(raw.get("gfx_activity_acc")), ACT_ACC_BITS), dacc, 1.0)
            umc_avg = share(counter_delta(_int(prev.get("mem_activity_acc")), _int(raw.get("mem_activity_acc")), ACT_ACC_BITS), dacc, 1.0)
            for k in RESIDENCY_KEYS:
                residency[k] = held(("res", k), share(counter_delta(_int(prev.get(f"res_{k}_acc")), _int(raw.get(f"res_{k}_acc"))), dacc, 100.0))
            gfx_avg, umc_avg = held("gfx_avg", gfx_avg), held("umc_avg", umc_avg)
        # per-XCD busy share over the period (same tick base as the whole-GPU accumulator)
        xcd_inst = raw.get("xcd_busy_inst") if isinstance(raw.get("xcd_busy_inst"), list) else None
        xcd_avg: Optional[List[Optional[float]]] = None
        xa, xb = (prev or {}).get("xcd_busy_acc"), raw.get("xcd_busy_acc")
        if prev is not None and isinstance(xa, list) and isinstance(xb, list) and len(xa) == len(xb) and dacc:
            xcd_avg = []
            for a0, a1 in zip(xa, xb):
                v = share(counter_delta(_int(a0), _int(a1)), dacc, 1.0)
                xcd_avg.append(None if v is None else min(100.0, max(0.0, v)))
        xcd_avg = held("xcd_avg", xcd_avg)
        # PCIe link health: cumulative error counters and what they gained since the last reading
        pcie_err = {}
        for key, name in (("pcie_replay_acc", "replay"), ("pcie_l0_recovery_acc", "l0Recovery"),
                          ("pcie_nak_sent_acc", "nakSent"), ("pcie_nak_rcvd_acc", "nakReceived")):
            cur = _int(raw.get(key))
            if cur is None:
                continue
            before = _int((prev or {}).get(key))
            pcie_err[name] = {"count": cur, "delta": None if before is None else counter_delta(before, cur)}
        if gfx_avg is not None:
            gfx_avg = min(100.0, max(0.0, gfx_avg))
        if umc_avg is not None:
            umc_avg = min(100.0, max(0.0, umc_avg))
        util = gfx_avg if gfx_avg is not None else gfx_inst

        # power: instantaneous gauge plus the energy-counter mean over the period
        power_w = _num(raw.get("power_w"))
        power_avg = None
        res_uj = _num(raw.get("energy_resolution_uj"))
        e_rate = self._rates.update(("energy", idx), tt, _int(raw.get("energy_acc")))
        if e_rate is not None and res_uj:
            power_avg = e_rate * res_uj * 1e-6
        power_avg = held("power_avg", power_avg)
        cap_w = _num(raw.get("power_cap_w"))

        # throttle
        active = [k for k in RESIDENCY_KEYS if raw.get(f"active_{k}") is True]
        throttling = [k for k, v in residency.items() if v is not None and v > 0.0]

        # xGMI per-link traffic
        links = []
        rd, wr = raw.get("xgmi_read_kib") or [], raw.get("xgmi_write_kib") or []
        st = raw.get("xgmi_link_status") or []
        peers = raw.get("xgmi_peers") or []
        peer_idx, peer_src = raw.get("xgmi_peer_index") or [], raw.get("xgmi_peer_source") or []
        max_gbps = raw.get("xgmi_link_max_gbps") or []
        rx_total = tx_total = 0.0
        any_rate = False
        for i in range(max(len(rd), len(wr), len(st))):
            r_kib = _int(rd[i]) if i < len(rd) else None
            w_kib = _int(wr[i]) if i < len(wr) else None
            s = st[i] if i < len(st) else None
            state = {1: "up", 0: "down", "U": "up", "D": "down", "X": "disabled"}.get(s, "n/a")
            rx = held(("xr", i), self._rates.update(("xr", idx, i), tt, r_kib))
            tx = held(("xw", i), self._rates.update(("xw", idx, i), tt, w_kib))
            rx_gbps = None if rx is None else rx * 1024 / 1e9
            tx_gbps = None if tx is None else tx * 1024 / 1e9
            if rx_gbps is not None:
                rx_total += rx_gbps
                any_rate = True
            if tx_gbps is not None:
                tx_total += tx_gbps
            if state == "n/a" and r_kib in (None, 0) and w_kib in (None, 0):
                continue  # link slot not populated (the GPU's own position in the hive)
            cap = _num(max_gbps[i]) if i < len(max_gbps) else None
            links.append({"link": i, "state": state, "peer": peers[i] if i < len(peers) else None,
                          "peerIndex": peer_idx[i] if i < len(peer_idx) else None,
                          "peerSource": peer_src[i] if i < len(peer_src) else None,
                          "maxGBps": None if cap is None else cap / 8.0,       # amdsmi: Gb/s, one direction
                          "rxBytes": None if r_kib is None else r_kib * 1024,
                          "txBytes": None if w_kib is None else w_kib * 1024,
                          "rxGBps": rx_gbps, "txGBps": tx_gbps})

        temp = _num(raw.get("temp_hotspot"))
        out = {
            # --- the reference's five keys, same meaning and units (monitor_server.js:90)
            "name": raw.get("name") or "AMD Instinct",
            "utilization": int(round(util)) if util is not None else 0,
            "memoryUsed": int(used_mib),
            "memoryTotal": int(total_mib),
            "temperature": int(round(temp)) if temp is not None else 0,
            # --- MI355X extensions
            "index": idx,
            "uuid": raw.get("uuid"),
            "bdf": raw.get("bdf"),
            "hbmUsedBytes": int(used_b),
            "hbmTotalBytes": int(total_b),
            "hbmPercent": (used_b / total_b * 100.0) if total_b else None,
            "gfxActivity": gfx_inst,
            "gfxActivityAvg": gfx_avg,
            "memActivity": umc_inst,
            "memActivityAvg": umc_avg,
            "powerW": power_w,
            "powerAvgW": power_avg,
            # the accumulator itself, in joules: lets a reader form its own mean over any window (delta E / delta t)
            "metricsTableTimeS": tt if fw else None,     # the table's firmware clock at its last refresh (pairs with energyJ)
            "energyJ": (_int(raw.get("energy_acc")) * res_uj * 1e-6) if (_int(raw.get("energy_acc")) is not None and res_uj) else None,
            "powerCapW": cap_w,
            "powerPercent": (power_w / cap_w * 100.0) if (power_w is not None and cap_w) else None,
            "temperatureHbm": _num(raw.get("temp_mem")),
            "temperatureEdge": _num(raw.get("temp_edge")),
            "clockGfxMHz": _num(raw.get("clk_gfx")),
            "clockMemMHz": _num(raw.get("clk_mem")),
            "ecc": {"correctable": _int(raw.get("ecc_correctable")),
                    "uncorrectable": _int(raw.get("ecc_uncorrectable")),
                    "deferred": _int(raw.get("ecc_deferred"))},
            # retired HBM pages (RAS): taken out of use / waiting for the next driver reload / could not be reserved;
            # None where the driver does not say
            "badPages": (None if raw.get("bad_pages_retired") is None else
                         {"retired": _int(raw.get("bad_pages_retired")), "pending": _int(raw.get("bad_pages_pending")) or 0,
                          "unreservable": _int(raw.get("bad_pages_unreservable")) or 0}),
            "throttle": {"active": bool(active or throttling),
                         "reasons": sorted(set(active) | set(throttling)),
                         "residencyPercent": residency},
            "xgmi": {"links": links,
                     "rxGBps": rx_total if any_rate else None,
                     "txGBps": tx_total if any_rate else None,
                     "linkWidth": _int(raw.get("xgmi_link_width")),
                     "linkSpeedGbps": _int(raw.get("xgmi_link_speed")),
                     "hiveId": None if raw.get("xgmi_hive_id") is None else str(raw.get("xgmi_hive_id")),   # 64-bit: a string for JS
                     "nodeId": _int(raw.get("xgmi_node_id")),
                     "peersKnown": any(l.get("peerIndex") is not None or l.get("peer") for l in links),
                     "nearest": raw.get("xgmi_nearest"),
                     "topologyNote": raw.get("xgmi_topology_note"),
                     "errorStatus": raw.get("xgmi_error_status")},
            # amdsmi's pcie_bandwidth_inst is megabits per second (amd-smi prints it as Mb/s; 700 under launch traffic)
            "pcieGBps": None if _num(raw.get("pcie_bandwidth_inst")) is None else _num(raw.get("pcie_bandwidth_inst")) / 8000.0,
            "pcie": {"GBps": None if _num(raw.get("pcie_bandwidth_inst")) is None else _num(raw.get("pcie_bandwidth_inst")) / 8000.0,
                     "linkSpeedGTs": (_num(raw.get("pcie_link_speed")) or 0) / 10.0 or None,
                     "linkWidth": _int(raw.get("pcie_link_width")),
                     "errors": pcie_err},
            # what this device is a slice of: compute partition mode (SPX: the whole GPU; CPX: one XCD per device, ...) and
            # memory partition mode (NPS1 ...); None where the driver does not say
            "partition": {"compute": raw.get("compute_partition"), "memory": raw.get("memory_partition")},
            "xcdActivity": xcd_inst,
            "xcdActivityAvg": xcd_avg,
            "processes": raw.get("processes"),
            "processList": raw.get("process_list"),
            "peak": raw.get("peak"),
            "timestamp": raw.get("wall", time.time()),
        }
        return out


# --------------------------------------------------------------------------------------
# xGMI topology ("the link-to-peer mapping comes from topology calls")
# --------------------------------------------------------------------------------------
def _valid_bdf(bdf: Any) -> bool:
    """amdsmi names a link's far end by PCI address; a GPU the process cannot see (another container's
    device) comes back as the all-ones address ``ffffffffffff:ff:1f.7`` (profiles/r2_amdsmi_topo_mi355x.json)."""
    return isinstance(bdf, str) and bool(bdf) and not bdf.lower().startswith("ffff") and bdf != "N/A"


def read_xgmi_topology(smi, handles: Sequence[Any], bdfs: Sequence[Optional[str]]) -> List[Dict[str, Any]]:
    """One record per GPU, read once at start-up:

      ``links[slot]`` = {peer_bdf, peer_index, source, bit_rate_gbps, max_gbps, link_type} for every slot of
      ``amdsmi_get_link_metrics`` (same slot order as ``xgmi_read_data_acc`` in the metrics table: the two
      carry identical counters on MI355X), ``hive_id`` / ``node_id`` from ``amdsmi_get_xgmi_info``,
      ``nearest`` = indices from ``amdsmi_get_link_topology_nearest(XGMI)``, ``pair_type[j]`` from
      ``amdsmi_topo_get_link_type``, and ``note`` saying why peers are unnamed when they are.

    Peer naming, in order: the far-end PCI address of the link slot when it is one of the visible GPUs
    (``source = "link_metrics"``); else, inside one hive, the GPU whose ``xgmi_node_id`` equals the slot
    number (the GPU's own slot is the one its own node id names: status ``X``) (``source = "xgmi_node_id"``);
    else None. Every call is guarded: a failing one leaves its field empty and its reason in ``note``."""
    by_bdf = {str(b).lower(): i for i, b in enumerate(bdfs) if b}
    infos: List[Dict[str, Any]] = []
    for h in handles:
        rec: Dict[str, Any] = {"links": [], "hive_id": None, "node_id": None, "nearest": None, "pair_type": {}, "notes": []}
        try:
            xi = smi.amdsmi_get_xgmi_info(h)
            rec["hive_id"], rec["node_id"] = _int(xi.get("xgmi_hive_id")), _int(xi.get("xgmi_node_id"))
        except Exception as e:
            rec["notes"].append(f"xgmi_info: {type(e).__name__}")
        try:
            lm = smi.amdsmi_get_link_metrics(h)
            for slot, l in enumerate(lm.get("links") or []):
                rec["links"].append({"slot": slot, "peer_bdf": l.get("bdf") if _valid_bdf(l.get("bdf")) else None, "peer_index": None,
                                     "source": None, "bit_rate_gbps": _num(l.get("bit_rate")), "max_gbps": _num(l.get("max_bandwidth")),
                                     "link_type": _int(l.get("link_type"))})
        except Exception as e:
            rec["notes"].append(f"link_metrics: {type(e).__name__}")
        infos.append(rec)
    LT = getattr(smi, "AmdSmiLinkType", None)
    for i, (h, rec) in enumerate(zip(handles, infos)):
        for l in rec["links"]:
            j = by_bdf.get(str(l["peer_bdf"]).lower()) if l["peer_bdf"] else None
            if j is not None and j != i:
                l["peer_index"], l["source"] = j, "link_metrics"
        if not any(l["peer_index"] is not None for l in rec["links"]) and len(handles) > 1 and rec["hive_id"] is not None:
            # far ends unnamed by the link table: same-hive GPUs by xgmi node id = slot number
            for l in rec["links"]:
                for j, other in enumerate(infos):
                    if j != i and other["hive_id"] == rec["hive_id"] and other["node_id"] == l["slot"] and rec["node_id"] != l["slot"]:
                        l["peer_index"], l["peer_bdf"], l["source"] = j, bdfs[j], "xgmi_node_id"
        if LT is not None and len(handles) > 1:
            try:
                near = smi.amdsmi_get_link_topology_nearest(h, LT.AMDSMI_LINK_TYPE_XGMI)
                got = {getattr(x, "value", x) for x in (near.get("processor_list") or [])}
                rec["nearest"] = sorted(k for k, h2 in enumerate(handles) if k != i and getattr(h2, "value", h2) in got)
            except Exception as e:
                rec["notes"].append(f"topology_nearest: {type(e).__name__}")
            for j, h2 in enumerate(handles):
                if j == i:
                    continue
                try:
                    t = smi.amdsmi_topo_get_link_type(h, h2)
                    rec["pair_type"][j] = {"hops": _int(t.get("hops")), "type": _int(t.get("type"))} if isinstance(t, dict) else t
                except Exception:
                    pass
        if rec["links"] and not any(l["peer_index"] is not None for l in rec["links"]):
            rec["notes"].append("peers not visible to this process (1 GPU visible of the hive)" if len(handles) == 1
                                else "no link slot names a visible GPU")
        rec["note"] = "; ".join(rec.pop("notes")) or None
    return infos


# --------------------------------------------------------------------------------------
# readers
# --------------------------------------------------------------------------------------
class GpuReader:
    kind = "none"

    def read(self) -> List[Dict[str, Any]]:
        return []

    def close(self) -> None:
        pass

    def describe(self) -> str:
        return self.kind


class AmdSmiReader(GpuReader):
    """In-process reader on the ``amdsmi`` Python bindings: one ``gpu_metrics`` call, one
    VRAM call and one ECC call per GPU per sample (about 0.5 ms per GPU on MI355X)."""
    kind = "amdsmi"

    def __init__(self, processes: bool = False):
        import amdsmi  # raises ImportError / OSError when the library is absent
        self.processes = processes
        self._smi = amdsmi
        amdsmi.amdsmi_init()
        self._handles = list(amdsmi.amdsmi_get_processor_handles())
        if not self._handles:
            amdsmi.amdsmi_shut_down()
            raise RuntimeError("amdsmi found no GPU")
        self._static: List[Dict[str, Any]] = []
        for h in self._handles:
            st: Dict[str, Any] = {}
            for key, fn in (("asic", amdsmi.amdsmi_get_gpu_asic_info),
                            ("uuid", amdsmi.amdsmi_get_gpu_device_uuid),
                            ("bdf", amdsmi.amdsmi_get_gpu_device_bdf),
                            ("board", amdsmi.amdsmi_get_gpu_board_info),
                            ("cap", amdsmi.amdsmi_get_power_cap_info),
                            ("energy", amdsmi.amdsmi_get_energy_count)):
                try:
                    st[key] = fn(h)
                except Exception:
                    st[key] = None
            self._static.append(st)
        try:
            self.topology = read_xgmi_topology(amdsmi, self._handles, [st.get("bdf") for st in self._static])
        except Exception as e:                      # telemetry must come up even when the topology calls do not
            self.topology = [{"links": [], "hive_id": None, "node_id": None, "nearest": None, "pair_type": {},
                              "note": f"topology unavailable: {type(e).__name__}: {e}"} for _ in self._handles]
        self._xgmi_err_supported = [True] * len(self._handles)
        self._partition_supported = [True] * len(self._handles)
        self._bad_pages_supported = [True] * len(self._handles)
        self._slow_every = 12       # ECC and process list change rarely: read every Nth sample
        self._n = 0
        self._slow_cache: Dict[int, Dict[str, Any]] = {}
        self._lock = threading.Lock()

    @staticmethod
    def _name(st) -> Optional[str]:
        """Market name; libdrm's generic fallback ("AMD Radeon Graphics", when amdgpu.ids is not
        installed) is replaced by the board's product name."""
        name = (st.get("asic") or {}).get("market_name")
        board = (st.get("board") or {}).get("product_name")
        if (not name or name in ("N/A", "AMD Radeon Graphics")) and board and board != "N/A":
            return board
        return name

    def _cap_w(self, st) -> Optional[float]:
        cap = (st.get("cap") or {}).get("power_cap")
        v = _num(cap)
        if not v:
            return None
        return v / 1e6 if v > 100000 else v   # reported in microwatts

    supports_fast = True

    def read(self, fast: bool = False) -> List[Dict[str, Any]]:
        """`fast`: the alert watcher's read between two regular samples: the metrics table and VRAM only, the slow-path
        answers (ECC, bad pages, partitions, processes) from the cache, the slow-path cadence untouched."""
        smi = self._smi
        out = []
        with self._lock:
            slow = (not fast) and (self._n % self._slow_every) == 0
            if not fast:
                self._n += 1
            for i, h in enumerate(self._handles):
                st = self._static[i]
                t = time.monotonic()
                raw: Dict[str, Any] = {"index": i, "t": t, "wall": time.time(),
                                       "name": self._name(st),
                                       "uuid": st.get("uuid"), "bdf": st.get("bdf"),
                                       "power_cap_w": self._cap_w(st),
                                       "energy_resolution_uj": _num((st.get("energy") or {}).get("counter_resolution"))}
                try:
                    v = smi.amdsmi_get_gpu_vram_usage(h)
                    raw["vram_used_mib"], raw["vram_total_mib"] = v.get("vram_used"), v.get("vram_total")
                except Exception:
                    pass
                try:
                    m = smi.amdsmi_get_gpu_metrics_info(h)
                except Exception:
                    m = {}
                raw.update(metrics_to_raw(m))
                raw.update(topology_to_raw(self.topology[i]))
                if slow or (i not in self._slow_cache and not fast):
                    sc: Dict[str, Any] = {}
                    if self._xgmi_err_supported[i]:
                        try:
                            est = smi.amdsmi_gpu_xgmi_error_status(h)
                            sc["xgmi_error_status"] = getattr(est, "name", str(est)).replace("AMDSMI_XGMI_STATUS_", "").lower()
                        except Exception as e:
                            # MI355X / amdsmi 26.2 answers NOT_SUPPORTED (profiles/r2_amdsmi_topo_mi355x.json): ask once
                            self._xgmi_err_supported[i] = False
                            sc["xgmi_error_status"] = "not_supported" if "NOT_SUPPORTED" in str(e) else f"error: {type(e).__name__}"
                    else:
                        sc["xgmi_error_status"] = (self._slow_cache.get(i) or {}).get("xgmi_error_status", "not_supported")
                    # compute / memory partition modes (SPX / CPX ..., NPS1 / NPS4 ...): what one "GPU" of this list is a slice of.
                    # They change only through an administrator's call: the slow path, and asked once where unsupported
                    if self._partition_supported[i]:
                        try:
                            sc["compute_partition"] = _partition_name(smi.amdsmi_get_gpu_compute_partition(h))
                            sc["memory_partition"] = _partition_name(smi.amdsmi_get_gpu_memory_partition(h))
                        except Exception:
                            self._partition_supported[i] = False
                    else:
                        for k in ("compute_partition", "memory_partition"):
                            if (self._slow_cache.get(i) or {}).get(k) is not None:
                                sc[k] = self._slow_cache[i][k]
                    try:
                        e = smi.amdsmi_get_gpu_total_ecc_count(h)
                        sc.update(ecc_correctable=e.get("correctable_count"),
                                  ecc_uncorrectable=e.get("uncorrectable_count"),
                                  ecc_deferred=e.get("deferred_count"))
                    except Exception:
                        pass
                    # retired HBM pages (RAS): how many pages the driver took out of use, and how many wait for the next reload
                    if self._bad_pages_supported[i]:
                        try:
                            sc.update(bad_pages_to_raw(smi.amdsmi_get_gpu_bad_page_info(h)))
                        except Exception:
                            self._bad_pages_supported[i] = False
                    else:
                        sc.update({k: v for k, v in (self._slow_cache.get(i) or {}).items() if k.startswith("bad_pages_")})
                    if self.processes:       # off by default: the call is slow and noisy on stderr without /proc access
                        try:
                            pl = smi.amdsmi_get_gpu_process_list(h)
                            sc["processes"] = len(pl)
                            sc["process_list"] = [{"pid": int(p.get("pid")), "name": p.get("name"),
                                                   "vramBytes": int((p.get("memory_usage") or {}).get("vram_mem") or p.get("mem") or 0)}
                                                  for p in pl]
                        except Exception:
                            pass
                    self._slow_cache[i] = sc
                raw.update(self._slow_cache.get(i) or {})
                out.append(raw)
        return out

    def close(self) -> None:
        try:
            self._smi.amdsmi_shut_down()
        except Exception:
            pass


def bad_pages_to_raw(info: Any) -> Dict[str, int]:
    """amdsmi_get_gpu_bad_page_info -> counts per status. The bindings answer a list of records ({"status": <enum or str>, ...})
    or, with nothing retired, an empty list / a sentence ("No bad pages found.")."""
    out = {"bad_pages_retired": 0, "bad_pages_pending": 0, "bad_pages_unreservable": 0}
    if isinstance(info, (list, tuple)):
        for rec in info:
            st = rec.get("status") if isinstance(rec, dict) else None
            name = str(getattr(st, "name", st) or "").upper()
            if "PENDING" in name:
                out["bad_pages_pending"] += 1
            elif "UNRESERVABLE" in name:
                out["bad_pages_unreservable"] += 1
            else:
                out["bad_pages_retired"] += 1
    return out


def _partition_name(v: Any) -> Optional[str]:
    """amdsmi returns the mode as a string ("SPX", "NPS1"), an enum member or "N/A"."""
    name = getattr(v, "name", None) or (v if isinstance(v, str) else None)
    if not name or name in ("N/A", "UNKNOWN", "INVALID"):
        return None
    return str(name).replace("AMDSMI_COMPUTE_PARTITION_", "").replace("AMDSMI_MEMORY_PARTITION_", "")


def topology_to_raw(topo: Dict[str, Any]) -> Dict[str, Any]:
    """One GPU's start-up topology record -> raw-reading keys (per link slot)."""
    links = topo.get("links") or []
    return {"xgmi_peers": [l.get("peer_bdf") for l in links],
            "xgmi_peer_index": [l.get("peer_index") for l in links],
            "xgmi_peer_source": [l.get("source") for l in links],
            "xgmi_link_max_gbps": [l.get("max_gbps") for l in links],
            "xgmi_hive_id": topo.get("hive_id"), "xgmi_node_id": topo.get("node_id"),
            "xgmi_nearest": topo.get("nearest"), "xgmi_topology_note": topo.get("note")}


def metrics_to_raw(m: Dict[str, Any]) -> Dict[str, Any]:
    """Map one ``amdsmi_get_gpu_metrics_info`` dict to raw-reading keys."""
    g = m.get
    raw = {
        "gfx_activity": g("average_gfx_activity"),
        "umc_activity": g("average_umc_activity"),
        "gfx_activity_acc": g("gfx_activity_acc"),
        "mem_activity_acc": g("mem_activity_acc"),
        "accumulation_counter": g("accumulation_counter"),
        "firmware_timestamp": g("firmware_timestamp"),
        "power_w": g("current_socket_power") if _num(g("current_socket_power")) is not None else g("average_socket_power"),
        "energy_acc": g("energy_accumulator"),
        "temp_hotspot": g("temperature_hotspot"),
        "temp_mem": g("temperature_mem"),
        "temp_edge": g("temperature_edge"),
        "clk_gfx": g("current_gfxclk"),
        "clk_mem": g("current_uclk"),
        "res_prochot_acc": g("prochot_residency_acc"),
        "res_ppt_acc": g("ppt_residency_acc"),
        "res_socket_thm_acc": g("socket_thm_residency_acc"),
        "res_vr_thm_acc": g("vr_thm_residency_acc"),
        "res_hbm_thm_acc": g("hbm_thm_residency_acc"),
        "xgmi_read_kib": g("xgmi_read_data_acc"),
        "xgmi_write_kib": g("xgmi_write_data_acc"),
        "xgmi_link_status": g("xgmi_link_status"),
        "xgmi_link_width": g("xgmi_link_width"),
        "xgmi_link_speed": g("xgmi_link_speed"),
        "pcie_bandwidth_inst": g("pcie_bandwidth_inst"),
        "pcie_bandwidth_acc": g("pcie_bandwidth_acc"),
        "pcie_link_speed": g("pcie_link_speed"),               # 0.1 GT/s
        "pcie_link_width": g("pcie_link_width"),
        "pcie_replay_acc": g("pcie_replay_count_acc"),
        "pcie_replay_rollover_acc": g("pcie_replay_rover_count_acc"),
        "pcie_l0_recovery_acc": g("pcie_l0_to_recov_count_acc"),
        "pcie_nak_sent_acc": g("pcie_nak_sent_count_acc"),
        "pcie_nak_rcvd_acc": g("pcie_nak_rcvd_count_acc"),
        # per-XCD busy accumulators: one row per compute partition (one row of 8 in SPX mode, 8 rows of 1 in CPX)
        "xcd_busy_acc": _flatten_ints(g("xcp_stats.gfx_busy_acc")),
        "xcd_busy_inst": _flatten_ints(g("xcp_stats.gfx_busy_inst")),
    }
    return {k: v for k, v in raw.items() if v is not None and v != "N/A"}


def _flatten_ints(rows: Any) -> Optional[List[int]]:
    """amdsmi's per-partition tables ([[...8...], ["N/A", ...], ...]) -> the populated entries in order."""
    if not isinstance(rows, (list, tuple)):
        return None
    out: List[int] = []
    for row in rows:
        for v in (row if isinstance(row, (list, tuple)) else [row]):
            if isinstance(v, bool) or not isinstance(v, (int, float)):
                continue
            out.append(int(v))
    return out or None


class NativeSamplerReader(GpuReader):
    """Reads the JSON lines of the C++ sampler (``klm_amdsmi_sampler``), which polls amdsmi
    at a high internal rate and emits one aggregated line per period (mean and peak)."""
    kind = "native"

    #: raw keys the in-process reader emits and the sampler does not, on purpose: `xgmi_nearest` comes from the Python
    #: bindings' topology helper (amdsmi_get_link_topology_nearest per link type), which only orders peers the link slots
    #: already name; nothing downstream of the normaliser reads it for alerts or panels other than the matrix tooltip
    RAW_KEYS_NOT_EMITTED = frozenset({"xgmi_nearest"})

    def __init__(self, exe: str = "", period: float = 1.0, hz: float = 20.0,
                 argv: Optional[Sequence[str]] = None, processes: bool = False):
        if not exe:
            from ..build import artifact
            exe = str(artifact("klm_amdsmi_sampler"))
        if argv is None and not os.path.exists(exe):
            raise FileNotFoundError(exe)
        self.exe = exe
        cmd = list(argv) if argv is not None else [exe, "--period", str(period), "--hz", str(hz)] + (["--processes"] if processes else [])
        self._p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                   text=True, bufsize=1)
        self._latest: List[Dict[str, Any]] = []
        self._err: Optional[str] = None
        self._got = threading.Event()
        self._th = threading.Thread(target=self._pump, name="klm-native-sampler", daemon=True)
        self._th.start()
        # wait for the first line so that a sampler that cannot start is an error now
        if not self._got.wait(timeout=max(5.0, 3 * period)):
            self.close()
            raise RuntimeError(f"{exe} produced no sample")
        if self._err and not self._latest:
            self.close()
            raise RuntimeError(f"{exe}: {self._err}")

    def _pump(self) -> None:
        assert self._p.stdout is not None
        for line in self._p.stdout:
            line = line.strip()
            if not line:
                continue
            try:
                msg = json.loads(line)
            except ValueError:
                continue
            if "error" in msg:
                self._err = str(msg["error"])
                self._got.set()
                continue
            gpus = msg.get("gpus")
            if isinstance(gpus, list):
                self._latest = gpus
                self._got.set()
        self._got.set()

    def read(self) -> List[Dict[str, Any]]:
        if self._p.poll() is not None and not self._latest:
            return []
        return [dict(g) for g in self._latest]

    def close(self) -> None:
        if self._p.poll() is None:
            self._p.terminate()
            try:
                self._p.wait(timeout=2)
            except subprocess.TimeoutExpired:
                self._p.kill()

    def describe(self) -> str:
        return f"native({os.path.basename(self.exe)})"


# --------------------------------------------------------------------------------------
# source
# --------------------------------------------------------------------------------------
class GpuSource:
    """reader + normaliser. ``sample()`` never raises: a failing reader yields ``[]`` exactly
    like the reference's catch-all (monitor_server.js:94)."""

    def __init__(self, reader: GpuReader):
        self.reader = reader
        self.kind = reader.kind
        self.norm = GpuNormalizer()
        self.last_duration_s = 0.0
        self.last_error: Optional[str] = None
        self.samples = 0

    def sample(self) -> List[Dict[str, Any]]:
        t0 = time.perf_counter()
        try:
            raws = self.reader.read()
            out = [self.norm.normalize(r) for r in raws]
            self.last_error = None
        except Exception as e:
            out = []
            self.last_error = f"{type(e).__name__}: {e}"
        self.last_duration_s = time.perf_counter() - t0
        self.samples += 1
        return out

    def close(self) -> None:
        self.reader.close()

    def describe(self) -> str:
        return self.reader.describe()


def open_gpu_source(kind: str = "auto", cfg=None) -> GpuSource:
    """kind: auto | amdsmi | native | exporter | fake | none. ``auto`` = amdsmi when it finds a GPU,
    else none (an empty list, like a host without nvidia-smi in the reference)."""
    kind = (kind or "auto").lower()
    if kind == "none":
        return GpuSource(GpuReader())
    if kind == "fake":
        from ..fakes.gpu import FakeGpuReader
        n = getattr(getattr(cfg, "sources", None), "fake_gpus", 8) if cfg is not None else 8
        scen = getattr(getattr(cfg, "sources", None), "fake_scenario", "") if cfg is not None else ""
        return GpuSource(FakeGpuReader(n_gpus=n, scenario=scen))
    if kind == "native":
        exe = getattr(getattr(cfg, "sources", None), "native_sampler", "") if cfg is not None else ""
        period = getattr(getattr(cfg, "periods", None), "realtime", 1.0) if cfg is not None else 1.0
        procs = bool(getattr(getattr(cfg, "sources", None), "gpu_processes", False)) if cfg is not None else False
        return GpuSource(NativeSamplerReader(exe=exe, period=min(period, 1.0), processes=procs))
    if kind == "exporter":
        from .exporter import ExporterGpuReader
        ex = getattr(cfg, "exporter", None)
        if ex is None:
            raise ValueError("gpu source 'exporter' needs a config (exporter.url)")
        return GpuSource(ExporterGpuReader(ex.url, ex.names(), ex.timeout))
    procs = bool(getattr(getattr(cfg, "sources", None), "gpu_processes", False)) if cfg is not None else False
    if kind == "amdsmi":
        return GpuSource(AmdSmiReader(processes=procs))
    if kind == "auto":
        try:
            return GpuSource(AmdSmiReader(processes=procs))
        except Exception:
            return GpuSource(GpuReader())
    raise ValueError(f"unknown gpu source {kind!r}")
