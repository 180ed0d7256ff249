"""rocprofv3 cross-check harness: ground truth for the burner, independent of amdsmi.

No reference counterpart. Two kinds of run, always separate
(a counter run is never combined with the trace domains):

  kernel trace  ``rocprofv3 --kernel-trace --stats``: start/end of every dispatch. The
                union of the burner kernel's intervals over the wall span is the TRUE GPU
                busy fraction, to hold against the burner's event-measured duty and the
                monitor's GFX activity.
  counters      ``rocprofv3 --pmc ...``: raw SQ / GRBM counters per dispatch (ROCm 7.2 has no
                gfx950 derived-metric section, so MfmaUtil-style figures are computed here
                from raw counters): MFMA busy share, MFMA FLOPs from the MOPS counters
                (one MOPS unit = 512 FLOP), LDS bank-conflict share.

A pass holds at most 8 SQ and 2 GRBM counters (more makes rocprofv3 hang), which
``check_pass`` enforces before anything is launched.
"""
from __future__ import annotations

import csv
import glob
import os
import shutil
import subprocess
import sys
from typing import Any, Dict, Iterable, List, Optional, Sequence

SLOTS = {"SQ_": 8, "TCC_": 4, "TCP_": 4, "TA_": 2, "TD_": 2, "GRBM_": 2}
TCC_COST = {"FETCH_SIZE": 3, "WRITE_SIZE": 2}

PASSES: Dict[str, List[str]] = {
    "mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES",
             "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F8", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_ANY",
             "GRBM_GUI_ACTIVE", "GRBM_COUNT"],
    "stall": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT",
              "SQ_INST_CYCLES_VMEM", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE", "GRBM_COUNT"],
    "hbm": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum",
            "GRBM_GUI_ACTIVE", "GRBM_COUNT"],
}
MOPS_FLOP = 512.0


def check_pass(counters: Sequence[str]) -> None:
    """Raise before launching a pass that does not fit the hardware's counter slots."""
    used: Dict[str, int] = {}
    seen = set()
    for c in counters:
        base = c
        for suf in ("_sum", "_avr", "_min", "_max"):
            if base.endswith(suf):
                base = base[: -len(suf)]
        if base in seen:
            continue
        seen.add(base)
        if base in TCC_COST:                      # derived counters that live in the TCC block
            used["TCC_"] = used.get("TCC_", 0) + TCC_COST[base]
            continue
        for pre in SLOTS:
            if base.startswith(pre):
                used[pre] = used.get(pre, 0) + 1
    for pre, n in used.items():
        if n > SLOTS[pre]:
            raise ValueError(f"{n} {pre}* counters in one pass; the hardware holds {SLOTS[pre]}")


def rocprofv3() -> Optional[str]:
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def _run(args: List[str], timeout: float) -> subprocess.CompletedProcess:
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    # hard kill on timeout: a hung counter run ignores SIGTERM
    return subprocess.run(["timeout", "-s", "KILL", str(int(timeout)), *args], env=env, capture_output=True, text=True)


def kernel_trace(cmd: Sequence[str], outdir: str, timeout: float = 180.0) -> Dict[str, Any]:
    """Run `cmd` (program first: `python3 x.py ...`, no env/shell wrappers) under a kernel trace."""
    exe = rocprofv3()
    if not exe:
        raise RuntimeError("rocprofv3 not found")
    os.makedirs(outdir, exist_ok=True)
    r = _run([exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", outdir, "--", *cmd], timeout)
    files = glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True)
    if r.returncode != 0 or not files:
        raise RuntimeError(f"rocprofv3 kernel trace failed rc={r.returncode}: {r.stderr[-500:]}")
    rows = read_kernel_trace(files[0])
    return {"file": files[0], "rows": rows, "stdout": r.stdout}


def read_kernel_trace(path: str) -> List[Dict[str, Any]]:
    out = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            try:
                out.append({"name": row["Kernel_Name"], "start": int(row["Start_Timestamp"]), "end": int(row["End_Timestamp"])})
            except (KeyError, ValueError):
                continue
    return out


def busy_fraction(rows: Iterable[Dict[str, Any]], name_contains: str = "", t0: Optional[int] = None,
                  t1: Optional[int] = None) -> Dict[str, float]:
    """Union of the kernel intervals (ns) over [t0, t1] (default: first start .. last end)."""
    iv = sorted((r["start"], r["end"]) for r in rows if name_contains in r["name"])
    if not iv:
        return {"busy_fraction": 0.0, "busy_ns": 0.0, "span_ns": 0.0, "kernels": 0}
    lo = iv[0][0] if t0 is None else t0
    hi = max(e for _, e in iv) if t1 is None else t1
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        s, e = max(s, lo), min(e, hi)
        if e <= s:
            continue
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    span = max(1, hi - lo)
    return {"busy_fraction": busy / span, "busy_ns": float(busy), "span_ns": float(span), "kernels": len(iv)}


def pmc(cmd: Sequence[str], counters: Sequence[str], outdir: str, timeout: float = 120.0) -> Dict[str, Any]:
    check_pass(counters)
    exe = rocprofv3()
    if not exe:
        raise RuntimeError("rocprofv3 not found")
    os.makedirs(outdir, exist_ok=True)
    r = _run([exe, "--pmc", *counters, "--output-format", "csv", "-d", outdir, "--", *cmd], timeout)
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        raise RuntimeError(f"rocprofv3 --pmc failed rc={r.returncode}: {r.stderr[-500:]}")
    return {"file": files[0], "per_kernel": read_counters(files[0])}


def read_counters(path: str) -> Dict[str, Dict[str, float]]:
    """kernel name -> counter -> value summed over that kernel's dispatches (+ 'dispatches')."""
    out: Dict[str, Dict[str, float]] = {}
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            try:
                k, c, v = row["Kernel_Name"], row["Counter_Name"], float(row["Counter_Value"])
            except (KeyError, ValueError):
                continue
            d = out.setdefault(k, {"dispatches": 0.0})
            d[c] = d.get(c, 0.0) + v
            did = row.get("Dispatch_Id")
            if (k, did) not in seen:
                seen.add((k, did))
                d["dispatches"] += 1
    return out


def derive(c: Dict[str, float], n_cu: int = 256, simd_per_cu: int = 4, n_xcd: int = 8,
           read_req_bytes: float = 128.0) -> Dict[str, Optional[float]]:
    """MfmaUtil-style figures from the raw counters of one kernel (sums over its dispatches;
    byte and FLOP totals are returned per dispatch).

    Calibrated on MI355X against known work (profiles/r1_prof/pmc_summary.json):
      * SQ_INSTS_VALU_MFMA_MOPS_* x 512 is exactly 2*M*N*K for both the bf16 and the MX-fp8 kernel;
      * GRBM_GUI_ACTIVE / GRBM_COUNT are summed over the 8 XCDs (8 x clock x time), while
        SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, so the MFMA-busy share of SIMD
        time is busy / (gui / n_xcd * n_cu * simd_per_cu);
      * a wide (16 B per lane) streaming read shows one TCC_EA0_RDREQ per 128 bytes; writes show
        one 64-byte WRREQ per 64 bytes (a 2 GiB copy reads 16.78 M and writes 33.56 M requests).
    """
    g = c.get
    n = g("dispatches") or 1.0
    out: Dict[str, Optional[float]] = {}
    gui = g("GRBM_GUI_ACTIVE")
    if g("GRBM_COUNT"):
        out["gpu_busy_share"] = gui / g("GRBM_COUNT") if gui is not None else None
    if gui:
        out["gfx_cycles_per_dispatch"] = gui / n_xcd / n
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and gui:
        out["mfma_busy_share"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / n_xcd * n_cu * simd_per_cu)
    mops = (g("SQ_INSTS_VALU_MFMA_MOPS_BF16") or 0.0) + (g("SQ_INSTS_VALU_MFMA_MOPS_F8") or 0.0)
    if mops:
        out["mfma_flop"] = mops * MOPS_FLOP / n
    if g("SQ_WAVE_CYCLES"):
        for k, name in (("SQ_WAIT_ANY", "wait_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"), ("SQ_ACTIVE_INST_ANY", "active_share")):
            if g(k) is not None:
                out[name] = g(k) / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_ACTIVE_INST_ANY"):
        out["lds_conflict_per_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_ANY")
    rd, rd32 = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum")
    wr, wr64 = g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_64B_sum")
    if rd is not None:
        out["hbm_read_bytes"] = ((rd - (rd32 or 0.0)) * read_req_bytes + (rd32 or 0.0) * 32.0) / n
    if wr is not None:
        out["hbm_write_bytes"] = ((wr64 or 0.0) * 64.0 + (wr - (wr64 or 0.0)) * 32.0) / n
    return out


#: substrings of the bf16 burner's kernel names over the rounds: gemm_bf16_128 / gemm_bf16_256_t (ring), gemm_bf16_quad
#: (compiler-scheduled), gemm_quad_asm / gemm_quad_asm_p (the hand-scheduled streams, shared with fp8 / fp4)
BURNER_GEMM_NAMES = ("gemm_bf16", "gemm_quad_asm")


def is_burner_gemm(name: str) -> bool:
    return any(k in name for k in BURNER_GEMM_NAMES)


def burner_cmd(kind: str = "gemm_bf16", duty: float = 50.0, seconds: float = 4.0, size: int = 8192) -> List[str]:
    """The program rocprofv3 runs: python itself, no env/shell hop in front of it."""
    return [sys.executable, "-m", "k8s_llm_monitor_amd", "burner", "--kind", kind, "--duty", str(duty),
            "--seconds", str(seconds), "--size", str(size)]


def cross_check_duty(outdir: str, duties: Sequence[float] = (25.0, 50.0, 75.0), seconds: float = 4.0) -> List[Dict[str, Any]]:
    """Burner duty as rocprofv3 sees it vs what the burner measured itself."""
    import json
    rows = []
    for d in duties:
        res = kernel_trace(burner_cmd(duty=d, seconds=seconds), os.path.join(outdir, f"duty{int(d)}"))
        gemm = [r for r in res["rows"] if is_burner_gemm(r["name"])]
        # skip the calibration launches: measure over the duty-cycled part only
        gemm.sort(key=lambda r: r["start"])
        tail = gemm[13:] if len(gemm) > 40 else gemm
        bf = busy_fraction(tail)
        self_report = {}
        for line in reversed(res["stdout"].splitlines()):
            if line.startswith("{") and "duty_achieved" in line:
                self_report = json.loads(line)
                break
        rows.append({"duty_target": d, "duty_rocprof": bf["busy_fraction"] * 100.0, "kernels": bf["kernels"],
                     "duty_burner_events": self_report.get("duty_achieved")})
    return rows


def kernel_durations(rows: Iterable[Dict[str, Any]], name_contains, skip: int = 16) -> List[float]:
    """Durations (us) of the matching dispatches in start order, without the first `skip` (calibration, cold clocks).
    `name_contains`: a substring, or a predicate on the kernel name."""
    match = name_contains if callable(name_contains) else (lambda n: name_contains in n)
    ks = sorted((r for r in rows if match(r["name"])), key=lambda r: r["start"])
    return [(r["end"] - r["start"]) / 1e3 for r in ks[skip:]]


def overhead_kernel_trace(outdir: str, repeats: int = 3, seconds: float = 5.0, size: int = 8192,
                          monitor_period: float = 1.0) -> Dict[str, Any]:
    """GPU-side ground truth for the sampling overhead: the burner's per-kernel duration in rocprofv3's kernel
    trace with the whole monitor running beside it vs alone, interleaved, as a paired difference per repeat."""
    from .phases import mean_ci95
    legs: Dict[str, List[Dict[str, Any]]] = {"off": [], "on": []}
    for i in range(repeats):
        for leg in ("off", "on"):
            cmd = burner_cmd(duty=100.0, seconds=seconds, size=size) + (["--monitored", "--monitor-period", str(monitor_period)] if leg == "on" else [])
            res = kernel_trace(cmd, os.path.join(outdir, f"{leg}{i}"))
            d = kernel_durations(res["rows"], is_burner_gemm)
            ks = sorted((r for r in res["rows"] if is_burner_gemm(r["name"])), key=lambda r: r["start"])[16:]
            med = sorted(d)[len(d) // 2] if d else 0.0
            # kernels that took more than twice the median: when in the window, and how long
            outliers = [{"at_s": round((r["start"] - ks[0]["start"]) / 1e9, 3), "us": round((r["end"] - r["start"]) / 1e3, 1)}
                        for r in ks if (r["end"] - r["start"]) / 1e3 > 2.0 * med][:20]
            d.sort()
            n = len(d)
            legs[leg].append({"kernels": n, "mean_us": sum(d) / n if n else None, "median_us": d[n // 2] if n else None,
                              "p99_us": d[min(n - 1, int(n * 0.99))] if n else None, "max_us": d[-1] if n else None,
                              "outliers_over_2x_median": outliers, "monitor": "monitor" in res["stdout"]})
    diffs = [(on["mean_us"] - off["mean_us"]) / off["mean_us"] * 100.0 for off, on in zip(legs["off"], legs["on"])
             if off["mean_us"] and on["mean_us"]]
    meds = [(on["median_us"] - off["median_us"]) / off["median_us"] * 100.0 for off, on in zip(legs["off"], legs["on"])
            if off["median_us"] and on["median_us"]]
    return {"kernel": "gemm_bf16 (8192^3 burner)", "window_s": seconds, "monitor_period_s": monitor_period, "legs": legs,
            "kernel_time_increase_pct": mean_ci95(diffs), "kernel_median_increase_pct": mean_ci95(meds)}
