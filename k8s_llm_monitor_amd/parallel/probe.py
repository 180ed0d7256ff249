"""RCCL all-reduce probe over xGMI: sweep model, ceilings, result files, native-probe driver.

No reference counterpart (the reference has no collective call site); this is
BASELINE.json north_star item 4 / config 3 and SURVEY.md 2.4 N5 + 5.1.

Two executors produce the same point records:
  * the C++ probe ``klm_rccl_probe`` (parallel/csrc/rccl_probe.cpp): one process, one RCCL
    communicator per device via ``ncclCommInitAll``, hipEvent timing;
  * ``parallel.dist_probe``: one process per GPU with ``torch.distributed`` (backend "nccl" =
    RCCL), which is what ``bench.py`` runs under ``torch.distributed.run``.

Bandwidth definitions (the nccl-tests convention):
  algBW = bytes / time;  busBW = algBW * 2(n-1)/n  for all-reduce.
Ceilings (arithmetic from 7 xGMI links x ~153 GB/s per GPU, one link per peer):
  full mesh (direct reduce-scatter + all-gather): (n-1) * link;  single ring: link.
"""
from __future__ import annotations

import json
import os
import subprocess
import time
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Sequence

XGMI_LINK_GBPS = 153.0
LOCAL_BOUND_GBPS = 8000.0      # nothing on one MI355X moves more than 8 TB/s (HBM3E peak); a copy's algBW is half its traffic
LINK_SLACK = 1.1               # the per-link figure is itself approximate
DTYPE_BYTES = {"bf16": 2, "fp16": 2, "fp32": 4, "fp8": 1}


def default_results_path() -> str:
    d = os.environ.get("KLM_STATE_DIR") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "klm")
    return os.path.join(d, "rccl_probe.json")


def sweep_sizes(min_bytes: int = 1024, max_bytes: int = 1 << 30, factor: int = 4) -> List[int]:
    """1 KiB -> 1 GiB in x4 steps; always ends on ``max_bytes``."""
    if min_bytes <= 0 or max_bytes < min_bytes or factor < 2:
        raise ValueError("bad sweep")
    out, b = [], min_bytes
    while b < max_bytes:
        out.append(b)
        b *= factor
    out.append(max_bytes)
    return out


def bus_factor(n: int, op: str = "allreduce") -> float:
    if n <= 1:
        return 0.0
    if op == "allreduce":
        return 2.0 * (n - 1) / n
    if op in ("allgather", "reducescatter", "alltoall"):
        return (n - 1) / n
    return 1.0


def ceilings(n: int, link_gbps: float = XGMI_LINK_GBPS) -> Dict[str, Optional[float]]:
    if n <= 1:
        return {"mesh_GBps": None, "ring_GBps": None}
    return {"mesh_GBps": (n - 1) * link_gbps, "ring_GBps": link_gbps}


def bound_GBps(n: int, link_gbps: float = XGMI_LINK_GBPS) -> float:
    """The physical bound a point is held against: the local copy rate at n = 1, the full mesh's bus bandwidth beyond."""
    return (n - 1) * link_gbps * LINK_SLACK if n > 1 else LOCAL_BOUND_GBPS


def plausible(n: int, algbw: float, busbw: float, link_gbps: float = XGMI_LINK_GBPS) -> bool:
    return (busbw if n > 1 else algbw) <= bound_GBps(n, link_gbps)


def judged(p: Mapping[str, Any], link_gbps: float = XGMI_LINK_GBPS) -> Dict[str, Any]:
    """The point with check "implausible" when it is faster than the hardware can be (an in-place all-reduce on a
    one-rank communicator is a no-op and once read as 615 678 GB/s with check "ok")."""
    q = dict(p)
    if q.get("check") != "MISMATCH" and not plausible(int(q.get("n_gpus", 1)), float(q.get("algbw_GBps") or 0.0),
                                                        float(q.get("busbw_GBps") or 0.0), link_gbps):
        q["check"] = "implausible"
    return q


def point(n: int, nbytes: int, seconds: float, dtype: str = "bf16", op: str = "allreduce",
          check: Optional[str] = None) -> Dict[str, Any]:
    alg = nbytes / seconds / 1e9 if seconds > 0 else 0.0
    return judged({"type": "point", "n_gpus": n, "bytes": int(nbytes), "dtype": dtype, "op": op,
                   "time_us": seconds * 1e6, "algbw_GBps": alg, "busbw_GBps": alg * bus_factor(n, op),
                   "check": check})


def curve(n: int, points: Sequence[Mapping[str, Any]], dtype: str = "bf16", verified: bool = True,
          executor: str = "native", link_gbps: float = XGMI_LINK_GBPS) -> Dict[str, Any]:
    pts = sorted((judged(p, link_gbps) for p in points), key=lambda p: p["bytes"])
    bad = [p["bytes"] for p in pts if p.get("check") not in (None, "ok", "skipped")]
    # the peak is taken over the points that passed: an implausible point must not become the headline
    good = [p for p in pts if p.get("check") in (None, "ok", "skipped")] or pts
    peak = max(good, key=lambda p: p["busbw_GBps"] if n > 1 else p["algbw_GBps"]) if good else None
    c = ceilings(n, link_gbps)
    out = {"n_gpus": n, "dtype": dtype, "executor": executor, "verified": bool(verified and not bad), "points": pts,
           "mode": "allreduce" if n > 1 else "local_copy", "bound_GBps": bound_GBps(n, link_gbps),
           "peak_busbw_GBps": peak["busbw_GBps"] if peak else None,
           "peak_algbw_GBps": max((p["algbw_GBps"] for p in good), default=None),
           "peak_bytes": peak["bytes"] if peak else None,
           "ceiling_mesh_GBps": c["mesh_GBps"], "ceiling_ring_GBps": c["ring_GBps"]}
    if bad:
        out["failed_bytes"] = bad
    if peak and c["mesh_GBps"]:
        out["mesh_efficiency"] = peak["busbw_GBps"] / c["mesh_GBps"]
        # a peak near one link's rate at n>2 means RCCL ran a single ring
        out["single_ring_suspected"] = bool(n > 2 and peak["busbw_GBps"] < 1.25 * c["ring_GBps"])
    return out


def brief(c: Mapping[str, Any]) -> Dict[str, Any]:
    """One curve as bench.py's JSON line carries it: peak, ceilings, [[bytes, busBW], ...] (algBW at n = 1)."""
    n = c["n_gpus"]
    key = "busbw_GBps" if n > 1 else "algbw_GBps"
    rnd = lambda v: None if v is None else round(v, 2)
    out = {"peak_busbw_GBps": rnd(c.get("peak_busbw_GBps")), "peak_algbw_GBps": rnd(c.get("peak_algbw_GBps")),
           "peak_bytes": c.get("peak_bytes"), "ceiling_mesh_GBps": c.get("ceiling_mesh_GBps"),
           "ceiling_ring_GBps": c.get("ceiling_ring_GBps"), "verified": c.get("verified"),
           "executor": c.get("executor"), "ranks_seen": c.get("ranks_seen", (c.get("header") or {}).get("n_gpus")),
           "mode": c.get("mode"),
           "curve_key": key, "curve": [[p["bytes"], round(p[key], 4)] for p in c.get("points", [])]}
    for k in ("mesh_efficiency", "single_ring_suspected"):
        if k in c:
            out[k] = c[k] if isinstance(c[k], bool) else round(c[k], 3)
    return out


def results_doc(curves: Iterable[Mapping[str, Any]], note: str = "") -> Dict[str, Any]:
    cs = sorted((dict(c) for c in curves), key=lambda c: (c["n_gpus"], c["dtype"]))
    return {"timestamp": time.time(), "link_GBps": XGMI_LINK_GBPS, "note": note, "curves": cs,
            "scaling": [{"n_gpus": c["n_gpus"], "dtype": c["dtype"], "peak_busbw_GBps": c["peak_busbw_GBps"],
                         "peak_algbw_GBps": c["peak_algbw_GBps"], "ceiling_mesh_GBps": c["ceiling_mesh_GBps"],
                         "verified": c["verified"], "mode": c.get("mode") or ("allreduce" if c["n_gpus"] > 1 else "local_copy")}
                        for c in cs]}


def merge_results(old: Optional[Mapping[str, Any]], new: Mapping[str, Any]) -> Dict[str, Any]:
    """Keep the curves of `old` that `new` does not replace (same n_gpus and dtype)."""
    have = {(c["n_gpus"], c["dtype"]) for c in new.get("curves", [])}
    keep = [c for c in (old or {}).get("curves", []) if (c["n_gpus"], c["dtype"]) not in have]
    return results_doc([*keep, *new.get("curves", [])], note=new.get("note", ""))


def save_results(doc: Mapping[str, Any], path: str = "") -> str:
    path = path or default_results_path()
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "w", encoding="utf-8") as f:
        json.dump(doc, f)
    os.replace(tmp, path)
    return path


def load_results(path: str = "") -> Optional[Dict[str, Any]]:
    path = path or default_results_path()
    try:
        with open(path, "r", encoding="utf-8") as f:
            d = json.load(f)
        return d if isinstance(d, dict) and isinstance(d.get("curves"), list) else None
    except (OSError, ValueError):
        return None


def parse_probe_output(text: str) -> Dict[str, Any]:
    """JSON lines of the native probe -> {"header": {...}, "points": [...], "errors": [...]}."""
    header: Dict[str, Any] = {}
    pts: List[Dict[str, Any]] = []
    errs: List[str] = []
    for line in text.splitlines():
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            m = json.loads(line)
        except ValueError:
            continue
        t = m.get("type")
        if t == "header":
            header = m
        elif t == "point":
            pts.append(m)
        elif t == "error":
            errs.append(str(m.get("error")))
    return {"header": header, "points": pts, "errors": errs}


def run_native(n_gpus: int, dtype: str = "bf16", min_bytes: int = 1024, max_bytes: int = 1 << 30, factor: int = 4,
               warmup: int = 5, iters: int = 20, exe: str = "", timeout: float = 300.0,
               runner: Optional[Callable[..., subprocess.CompletedProcess]] = None) -> Dict[str, Any]:
    """Run ``klm_rccl_probe`` for one GPU count and return its curve."""
    if not exe:
        from ..build import artifact
        exe = str(artifact("klm_rccl_probe"))
    cmd = [exe, "--gpus", str(n_gpus), "--dtype", dtype, "--min", str(min_bytes), "--max", str(max_bytes),
           "--factor", str(factor), "--warmup", str(warmup), "--iters", str(iters)]
    run = runner or subprocess.run
    # several devices in one process share buffers through dmabuf IPC on this driver (see bench.py)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True, env=env)
    parsed = parse_probe_output(r.stdout)
    if r.returncode != 0 or parsed["errors"] or not parsed["points"]:
        raise RuntimeError(f"rccl probe failed (rc={r.returncode}): {parsed['errors'] or r.stderr[-400:]}")
    ok = all(p.get("check") in (None, "ok") for p in parsed["points"])
    if n_gpus == 1 and parsed["header"].get("in_place") is not False:
        ok = False                # a one-rank in-place all-reduce moves nothing: not a measurement (an older binary)
    c = curve(n_gpus, parsed["points"], dtype, verified=ok, executor="native")
    c["header"] = parsed["header"]
    return c


def run_sweep(gpu_counts: Sequence[int] = (1, 2, 4, 8), dtypes: Sequence[str] = ("bf16",), path: str = "",
              available: Optional[int] = None, **kw) -> Dict[str, Any]:
    """Probe every GPU count that fits the machine; counts that do not fit are recorded as
    unverified empty curves so the dashboard shows them as 'not measured'."""
    if available is None:
        try:
            import torch
            available = torch.cuda.device_count()
        except Exception:
            available = 0
    curves = []
    for dt in dtypes:
        for n in gpu_counts:
            if n <= available:
                curves.append(run_native(n, dt, **kw))
            else:
                curves.append(curve(n, [], dt, verified=False, executor="none"))
    doc = merge_results(load_results(path), results_doc(curves, note=f"{available} GPU(s) visible"))
    save_results(doc, path)
    return doc
