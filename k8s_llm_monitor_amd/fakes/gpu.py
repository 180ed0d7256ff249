"""Scripted MI355X telemetry for CPU-only runs, tests and fault injection.

``FakeGpuReader`` produces the same RAW cumulative readings as the amdsmi reader
(accumulators included), so everything downstream of a reader (delta arithmetic,
normalisation, alert engine, API, dashboard) runs unchanged on a machine without a GPU.
This is BASELINE.json config 1 ("mocked GPU/K8s endpoints") and the fault-injection tier
of SURVEY.md section 4; the reference has neither fakes nor tests.

A scenario is a function ``(t_seconds, gpu_index) -> GpuState``; ``SCENARIOS`` names the
shipped ones. Time comes from an injectable clock so tests can step it.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

HBM_TOTAL_MIB = 294896          # what amdsmi reports for one MI355X (288 GiB)
POWER_CAP_W = 1400.0
ENERGY_RES_UJ = 15.3
ACC_HZ = 1000                    # accumulation_counter ticks per second in the fake firmware


@dataclass
class GpuState:
    hbm_frac: float = 0.02
    gfx: float = 0.0             # percent busy
    umc: float = 0.0
    power_w: float = 250.0
    temp: float = 45.0
    temp_mem: float = 38.0
    clk_gfx: float = 2100.0
    ecc_ce: int = 0
    ecc_ue: int = 0
    ppt_throttle: float = 0.0    # share of time (0..1) spent power-throttled
    thm_throttle: float = 0.0
    link_up: List[bool] = field(default_factory=lambda: [True] * 7)
    xgmi_gbps: float = 0.0       # per link, each direction
    present: bool = True
    xcd_idle: bool = False       # XCD (index % 8) of this GPU does no work (a kernel whose grid leaves a die out)
    pcie_replays: int = 0        # cumulative PCIe replay count
    bad_pages: int = 0           # retired HBM pages (RAS)
    bad_pages_pending: int = 0   # pages marked bad, retired at the next driver reload


Scenario = Callable[[float, int], GpuState]


def _steady(t: float, i: int) -> GpuState:
    """A vLLM-style serving load: ~60 % HBM (weights + KV cache), bursty GFX activity."""
    wob = math.sin(t / 17.0 + i) * 0.5 + 0.5
    return GpuState(hbm_frac=0.58 + 0.04 * wob, gfx=55 + 30 * wob, umc=35 + 20 * wob,
                    power_w=700 + 300 * wob, temp=68 + 8 * wob, temp_mem=60 + 6 * wob,
                    xgmi_gbps=4.0 * wob)


def _idle(t: float, i: int) -> GpuState:
    return GpuState()


def _hbm_ramp(t: float, i: int) -> GpuState:
    """GPU 0 climbs through the three HBM tiers: 69 % -> 71 % (t>=10) -> 86 % (t>=20) -> 96 % (t>=30)."""
    s = _steady(t, i)
    if i == 0:
        s.hbm_frac = 0.69 if t < 10 else 0.71 if t < 20 else 0.86 if t < 30 else 0.96
    return s


def _thermal(t: float, i: int) -> GpuState:
    """GPU 1's junction temperature ramps 70 -> 110 C over 80 s and starts thermal throttling at 100 C."""
    s = _steady(t, i)
    if i == 1:
        s.temp = min(110.0, 70.0 + t * 0.5)
        s.thm_throttle = 0.0 if s.temp < 100 else min(1.0, (s.temp - 100) / 10)
        s.clk_gfx = 2100 - 900 * s.thm_throttle
    return s


def _ecc(t: float, i: int) -> GpuState:
    s = _steady(t, i)
    if i == 2:
        s.ecc_ce = int(t // 5)
        s.ecc_ue = 1 if t >= 15 else 0
        # the driver marks the page bad after the uncorrectable error (retired at the next reload); GPU 5 carries old history
        s.bad_pages_pending = 1 if t >= 20 else 0
    if i == 5:
        s.bad_pages = 3 + (1 if t >= 10 else 0)
    return s


def _xgmi_down(t: float, i: int) -> GpuState:
    s = _allreduce(t, i)
    if i == 3 and t >= 10:
        s.link_up[4] = False
    return s


def _allreduce(t: float, i: int) -> GpuState:
    """RCCL all-reduce probe running: every link busy at ~120 GB/s each way."""
    s = _steady(t, i)
    s.xgmi_gbps = 120.0
    s.gfx = 35.0
    return s


def _power(t: float, i: int) -> GpuState:
    s = _steady(t, i)
    s.power_w = POWER_CAP_W * (0.99 if t >= 5 else 0.8)
    s.ppt_throttle = 0.4 if t >= 5 else 0.0
    s.gfx = 100.0
    return s


def _pcie(t: float, i: int) -> GpuState:
    """GPU 2 sits on a marginal PCIe link (replays accumulate); GPU 5 runs a kernel that leaves one XCD idle."""
    s = _steady(t, i)
    if i == 2:
        s.pcie_replays = int(t // 4) * 3
    if i == 5:
        s.xcd_idle = True
    return s


def _gone(t: float, i: int) -> GpuState:
    s = _steady(t, i)
    if i == 7 and t >= 10:
        s.present = False
    return s


SCENARIOS: Dict[str, Scenario] = {
    "": _steady, "steady": _steady, "idle": _idle, "hbm_ramp": _hbm_ramp, "thermal": _thermal,
    "ecc": _ecc, "xgmi_down": _xgmi_down, "allreduce": _allreduce, "power": _power,
    "gpu_lost": _gone, "pcie": _pcie,
}


class FakeGpuReader:
    kind = "fake"
    supports_fast = True

    def __init__(self, n_gpus: int = 8, scenario: Any = "", clock: Callable[[], float] = time.monotonic):
        if callable(scenario):
            self.scenario: Scenario = scenario
            self.scenario_name = getattr(scenario, "__name__", "custom")
        else:
            if scenario not in SCENARIOS:
                raise ValueError(f"unknown fake GPU scenario {scenario!r}; have {sorted(k for k in SCENARIOS if k)}")
            self.scenario = SCENARIOS[scenario]
            self.scenario_name = scenario or "steady"
        self.n = n_gpus
        self.clock = clock
        self.t0 = clock()
        self._last_t: Optional[float] = None
        # cumulative firmware counters per GPU
        self._acc = [dict(acc=0.0, gfx=0.0, mem=0.0, energy=0.0, ppt=0.0, thm=0.0,
                          rd=[0.0] * 8, wr=[0.0] * 8, xcd=[0.0] * 8) for _ in range(n_gpus)]

    def describe(self) -> str:
        return f"fake({self.n}x MI355X, scenario={self.scenario_name})"

    def set_scenario(self, scenario: Any) -> None:
        self.scenario = scenario if callable(scenario) else SCENARIOS[scenario]
        self.scenario_name = getattr(scenario, "__name__", str(scenario))
        self.t0 = self.clock()

    def close(self) -> None:
        pass

    def read(self, fast: bool = False) -> List[Dict[str, Any]]:
        now = self.clock()
        t = now - self.t0
        dt = 0.0 if self._last_t is None else max(0.0, now - self._last_t)
        self._last_t = now
        out = []
        for i in range(self.n):
            s = self.scenario(t, i)
            if not s.present:
                continue
            a = self._acc[i]
            # integrate the state held since the last reading into the accumulators
            ticks = dt * ACC_HZ
            a["acc"] += ticks
            a["gfx"] += s.gfx * ticks
            a["mem"] += s.umc * ticks
            for x in range(8):                       # XCDs share the load evenly; XCD x of GPU x idles in `xcd_idle`
                a["xcd"][x] += (0.0 if (s.xcd_idle and x == i % 8) else s.gfx) * ticks
            a["ppt"] += s.ppt_throttle * ticks
            a["thm"] += s.thm_throttle * ticks
            a["energy"] += s.power_w * dt / (ENERGY_RES_UJ * 1e-6)
            status: List[Any] = ["N/A"] * 8
            for l in range(8):
                if l == i % 8:
                    continue          # own position in the hive: no link
                peer_slot = l if l < i % 8 else l - 1
                up = s.link_up[peer_slot] if peer_slot < len(s.link_up) else True
                status[l] = 1 if up else 0
                if up:
                    kib = s.xgmi_gbps * 1e9 * dt / 1024
                    a["rd"][l] += kib
                    a["wr"][l] += kib
            out.append({
                "index": i, "t": now, "wall": time.time(),
                "name": "AMD Instinct MI355X (fake)",
                "uuid": f"fake-{i:02d}", "bdf": f"0000:{0x10 + i:02x}:00.0",
                "vram_total_mib": HBM_TOTAL_MIB,
                "vram_used_mib": int(HBM_TOTAL_MIB * s.hbm_frac),
                "gfx_activity": round(s.gfx), "umc_activity": round(s.umc),
                "gfx_activity_acc": int(a["gfx"]) & 0xFFFFFFFF, "mem_activity_acc": int(a["mem"]) & 0xFFFFFFFF,
                "accumulation_counter": int(a["acc"]),
                "xcd_busy_acc": [int(v) for v in a["xcd"]], "xcd_busy_inst": [round(s.gfx)] * 8,
                "pcie_bandwidth_inst": 18, "pcie_link_speed": 320, "pcie_link_width": 16,
                "pcie_replay_acc": int(s.pcie_replays), "pcie_l0_recovery_acc": 0,
                "pcie_nak_sent_acc": 0, "pcie_nak_rcvd_acc": 0,
                "power_w": round(s.power_w), "power_cap_w": POWER_CAP_W,
                "energy_acc": int(a["energy"]), "energy_resolution_uj": ENERGY_RES_UJ,
                "temp_hotspot": round(s.temp), "temp_mem": round(s.temp_mem),
                "clk_gfx": s.clk_gfx, "clk_mem": 2000,
                "ecc_correctable": s.ecc_ce, "ecc_uncorrectable": s.ecc_ue, "ecc_deferred": 0,
                "bad_pages_retired": s.bad_pages, "bad_pages_pending": s.bad_pages_pending, "bad_pages_unreservable": 0,
                "res_prochot_acc": 0, "res_ppt_acc": int(a["ppt"]), "res_socket_thm_acc": int(a["thm"]),
                "res_vr_thm_acc": 0, "res_hbm_thm_acc": 0,
                "xgmi_read_kib": [int(x) for x in a["rd"]], "xgmi_write_kib": [int(x) for x in a["wr"]],
                "xgmi_link_status": status, "xgmi_link_width": 16, "xgmi_link_speed": 38,
                # a full mesh as amdsmi reports it on an 8-GPU hive: link slot l of GPU i ends at the GPU whose xgmi node id is l
                "xgmi_peers": [None if (l == i % 8 or l >= self.n) else f"0000:{0x10 + l:02x}:00.0" for l in range(8)],
                "xgmi_peer_index": [None if (l == i % 8 or l >= self.n) else l for l in range(8)],
                "xgmi_peer_source": [None if (l == i % 8 or l >= self.n) else "link_metrics" for l in range(8)],
                "xgmi_link_max_gbps": [608] * 8, "xgmi_hive_id": 0x57b3c0de57b3c0de, "xgmi_node_id": i % 8,
                "xgmi_nearest": [j for j in range(self.n) if j != i], "xgmi_topology_note": None, "xgmi_error_status": "no_errors", "compute_partition": "SPX", "memory_partition": "NPS1",
                "processes": 1,
                "process_list": [{"pid": 40000 + i, "name": "python3 (vllm worker)", "vramBytes": int(HBM_TOTAL_MIB * s.hbm_frac) << 20}],
            })
        return out
