"""The timed loop and the validation phases behind ``bench.py``.

No reference counterpart (the reference has no benchmark); the
quantities are the self-defined targets of BASELINE.md: telemetry error, sampling
overhead, alert latency and the RCCL bus bandwidth at N GPUs.
"""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional


def _sync_all(torch, dist, distributed):
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    if gpu:
        torch.cuda.synchronize()


from ..models.linear_step import LinearLayerStep as _Load   # noqa: E402  (kept under its old name for callers)


def timed_steps(torch, dist, load: _Load, world: int, steps: int, warmup: int, distributed=None, min_warm_s: float = 0.0,
                info: Optional[Dict[str, Any]] = None) -> float:
    """Seconds for exactly `steps` steps: barrier + synchronize on both sides, MAX over ranks.

    Warm-up is `warmup` steps AND at least `min_warm_s` seconds of steps: five 2 ms steps leave the card on its start-up
    clocks (the cap's control loop has not settled: 1 552 TFLOP/s timed after 10 ms of warm-up, 1 611 sustained), so the
    number would depend on --steps / --warmup. The extra step count is derived from the measured step time and agreed across
    ranks (MAX), because under data parallelism every step is also a collective all ranks must enter."""
    dd = (world > 1) if distributed is None else bool(distributed)   # a one-rank group still takes the collective path
    t_w = time.perf_counter()
    for _ in range(warmup):
        load.step()
    load.drain()
    _sync_all(torch, dist, dd)
    extra = 0
    if min_warm_s > 0:
        # the first steps of a process carry one-time costs (code objects, RCCL's first collective), so the step time is
        # taken from a small synchronised chunk and the count re-derived until the wall time is met (at most a few rounds)
        chunk = 8
        for _ in range(6):
            t1 = time.perf_counter()
            for i in range(chunk):
                load.step()
                if i % 64 == 63:
                    load.drain()
                    if torch.cuda.is_available():
                        torch.cuda.current_stream().synchronize()   # keep the launch queue bounded
            load.drain()
            _sync_all(torch, dist, dd)
            extra += chunk
            per = (time.perf_counter() - t1) / chunk
            left = min_warm_s - (time.perf_counter() - t_w)
            more = 0 if left <= 0 else min(int(left / max(per, 1e-6)) + 1, 1_000_000)
            if dd:
                # under data parallelism every step is also a collective all ranks must enter: agree on the count
                t = torch.tensor([more], device=load.x.device, dtype=torch.int64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                more = int(t.item())
            if more <= 0:
                break
            chunk = more
    if info is not None:
        info["warmup_steps_run"] = warmup + extra
        info["warmup_seconds"] = round(time.perf_counter() - t_w, 3)
    t0 = time.perf_counter()
    for _ in range(steps):
        load.step()
    load.drain()
    _sync_all(torch, dist, dd)
    dt = time.perf_counter() - t0
    if dd:
        t = torch.tensor([dt], device=load.x.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def _start_sampler(period: float, device_index: int):
    """Start the monitor's real sampler on this rank (None when telemetry is unavailable)."""
    try:
        from ..collectors.gpu import open_gpu_source
        from ..server.sampler import BackgroundPoller
    except Exception:
        return None
    try:
        src = open_gpu_source("auto")
        if src is None or src.kind == "none":
            return None
        p = BackgroundPoller(src, period=period)
        p.start()
        return p
    except Exception:
        return None


def vendor_comparison(torch, ops, dev, size: int, rounds: int = 5, iters: int = 20) -> Dict[str, Any]:
    """The hand-written burner kernels next to the vendor library's on the same operands, in this process:
    hipBLASLt behind ``torch.matmul`` (bf16) and ``torch._scaled_mm`` (e4m3, unit scales). The kernels alternate
    round by round so both see the same clock and temperature state; the figure is the median round."""
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    a = (torch.rand(size, size, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(size, size, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    c = torch.empty(size, size, device=dev, dtype=torch.bfloat16)
    one = torch.ones((), device=dev)
    bt, b8t = b.t(), b8.t()
    cands = {"bf16_klm": lambda: ops.gemm_bf16(a, b, out=c), "bf16_hipblaslt": lambda: torch.matmul(a, bt, out=c),
             "fp8_klm": lambda: ops.gemm_fp8(a8, b8, out=c),
             "fp8_scaled_mm": lambda: torch._scaled_mm(a8, b8t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)}
    flop = 2.0 * size ** 3
    runs: Dict[str, list] = {k: [] for k in cands}
    err: Dict[str, str] = {}
    for r in range(rounds + 1):                   # round 0 warms every kernel and is dropped
        for name, fn in cands.items():
            if name in err:
                continue
            try:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn()
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record()
                e1.synchronize()
                if r:
                    runs[name].append(flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            except Exception as e:                # e.g. a torch build without _scaled_mm for this dtype
                err[name] = f"{type(e).__name__}: {e}"
    med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items() if v}
    out: Dict[str, Any] = {"size": size, "rounds": rounds, "iters_per_round": iters, "tflops_median": med, "unit": "TFLOP/s"}
    if "bf16_klm" in med and "bf16_hipblaslt" in med:
        out["bf16_klm_over_hipblaslt"] = round(med["bf16_klm"] / med["bf16_hipblaslt"], 4)
    if "fp8_klm" in med and "fp8_scaled_mm" in med:
        out["fp8_klm_over_scaled_mm"] = round(med["fp8_klm"] / med["fp8_scaled_mm"], 4)
    if err:
        out["errors"] = err
    del a, b, a8, b8, c
    try:
        out["training_step"] = step_vendor_comparison(torch, ops, dev, size)
    except Exception as e:
        out["training_step"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["hbm"] = hbm_vendor_comparison(torch, ops, dev)
    except Exception as e:
        out["hbm"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["step_energy"] = step_energy_comparison(torch, ops, dev, size)
    except Exception as e:
        out["step_energy"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def step_vendor_comparison(torch, ops, dev, size: int, rounds: int = 3, steps: int = 15) -> Dict[str, Any]:
    """The bench's own step (fwd, dgrad, wgrad of one linear layer: three GEMMs on six different operand matrices, one GPU)
    on the hand-written kernel and on hipBLASLt (``torch.matmul``), alternating, median round."""
    from ..models.linear_step import LinearLayerStep
    arms = {"klm": LinearLayerStep(torch, None, ops, dev, 1, size),
            "hipblaslt": LinearLayerStep(torch, None, ops, dev, 1, size, gemm=lambda a, b, out: torch.matmul(a, b.t(), out=out))}
    runs: Dict[str, list] = {k: [] for k in arms}
    for r in range(rounds + 1):
        for name, st in arms.items():
            st.step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                st.step()
            e1.record()
            e1.synchronize()
            if r:
                runs[name].append(st.flops * steps / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in runs.items()}
    return {"tflops_median": med, "klm_over_hipblaslt": round(med["klm"] / med["hipblaslt"], 4)}


def step_energy_comparison(torch, ops, dev, size: int, seconds: float = 3.5, settle_s: float = 1.0) -> Dict[str, Any]:
    """Joules per training step of the hand-written kernels and of hipBLASLt, from the card's own energy accumulator. The card
    runs this load on its power cap, so energy per step IS the rate; the figure says why one side is faster. Each arm runs
    without a gap for `seconds`; power = energy accumulator over the metrics table's OWN firmware clock between the first table
    refresh after `settle_s` and the last one (both numbers come from the same snapshot, so the ~1 s refresh period of the table
    costs no accuracy); J per step = that power x the step time measured with HIP events over the same window."""
    import threading
    from ..collectors.gpu import open_gpu_source
    from ..models.linear_step import LinearLayerStep
    from . import phases as _ph
    src = open_gpu_source("amdsmi")
    try:
        idx = _ph._smi_index_for(torch, dev, src)
        arms = {"klm": LinearLayerStep(torch, None, ops, dev, 1, size),
                "hipblaslt": LinearLayerStep(torch, None, ops, dev, 1, size, gemm=lambda a, b, out: torch.matmul(a, b.t(), out=out))}
        out: Dict[str, Any] = {"seconds_per_arm": seconds, "unit": "J per step (fwd + dgrad + wgrad)"}
        for name, st in arms.items():
            rows: List[Any] = []
            stop = threading.Event()

            def watch():
                while not stop.is_set():
                    g = _ph._gpu_sample(src, idx)
                    if g and g.get("energyJ") is not None and g.get("metricsTableTimeS") is not None:
                        rows.append((time.perf_counter(), float(g["energyJ"]), float(g["metricsTableTimeS"])))
                    stop.wait(0.15)
            st.step()
            torch.cuda.synchronize()
            th = threading.Thread(target=watch, daemon=True)
            t0 = time.perf_counter()
            th.start()
            marks = []
            while time.perf_counter() - t0 < seconds:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    st.step()
                e1.record()
                e1.synchronize()
                marks.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 10.0))
            stop.set()
            th.join(2.0)
            late = [r for r in rows if r[0] - t0 >= settle_s]
            tables = [r for i, r in enumerate(late) if i == 0 or r[2] != late[i - 1][2]]           # one row per refresh of the table
            ms = [m for t, m in marks if t >= settle_s]
            if len(tables) < 2 or tables[-1][2] <= tables[0][2] or not ms:
                out[name] = {"error": f"{len(tables)} table refreshes in the window"}
                continue
            watts = (tables[-1][1] - tables[0][1]) / (tables[-1][2] - tables[0][2])
            step_ms = sum(ms) / len(ms)
            out[name] = {"power_w": round(watts, 1), "ms_per_step": round(step_ms, 4), "joule_per_step": round(watts * step_ms * 1e-3, 4),
                         "tflops": round(st.flops / (step_ms * 1e-3) / 1e12, 1), "table_refreshes": len(tables)}
        if "joule_per_step" in out.get("klm", {}) and "joule_per_step" in out.get("hipblaslt", {}):
            out["klm_over_hipblaslt_joules"] = round(out["klm"]["joule_per_step"] / out["hipblaslt"]["joule_per_step"], 4)
        return out
    finally:
        try:
            src.close()
        except Exception:
            pass


def hbm_vendor_comparison(torch, ops, dev, gib: int = 2, rounds: int = 3, iters: int = 10) -> Dict[str, Any]:
    """The HBM kernels next to what the framework offers for the same job on the same buffers: ``Tensor.fill_`` (an
    elementwise kernel) and ``Tensor.copy_`` (the runtime's copy kernel). TB/s, median round; copy counts read + write."""
    n = gib << 30
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    y = torch.empty(n, dtype=torch.uint8, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    ops.hbm_fill(x, 0x01020304)
    cands = {"fill_klm": (lambda: ops.hbm_fill(y, 0x5A5A5A5A), n), "fill_torch": (lambda: y.fill_(7), n),
             "copy_klm": (lambda: ops.hbm_copy(y, x), 2 * n), "copy_torch": (lambda: y.copy_(x), 2 * n),
             "read_klm": (lambda: ops.hbm_read(x, sink), n)}
    runs: Dict[str, list] = {k: [] for k in cands}
    for r in range(rounds + 1):
        for name, (fn, nbytes) in cands.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            if r:
                runs[name].append(nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    med = {k: round(sorted(v)[len(v) // 2], 2) for k, v in runs.items()}
    return {"buffer_GiB": gib, "TBps_median": med, "fill_klm_over_torch": round(med["fill_klm"] / med["fill_torch"], 3),
            "copy_klm_over_torch": round(med["copy_klm"] / med["copy_torch"], 3)}


#: what the sections around the timed region cost (seconds, one MI355X), for the time budget
SECTION_COST_S = {"dp_kernel": 4.0, "probe": 10.0, "xgmi": 15.0, "other_kernels": 35.0, "llm_serving": 30.0}


def _rehearsal_validation(spec: str, budget) -> Dict[str, Any]:
    """--cpu-rehearsal only: stand-ins for the validation phases, "name:priced_s:actual_s,...". They sleep instead of
    measuring, so that the budget / shedding / rank-waiting logic of a multi-rank launch can be run without a GPU."""
    from . import phases
    items = [x.split(":") for x in spec.split(",") if x]
    cost = {n: float(c) for n, c, _ in items}
    plan, shed = phases.plan_phases(set(cost), budget.left() if budget is not None else None, cost)
    ran = []
    for n, c, actual in items:
        if n not in plan:
            continue
        if budget is not None and not budget.allows(n, float(c)):
            continue
        time.sleep(float(actual))
        ran.append(n)
    shed = shed + [x["phase"] for x in (budget.shed if budget is not None else [])]
    return {"summary": {"pass": {n: True for n in ran}, "shed": shed, "rehearsal": True}, "ran": ran, "shed": shed}


def run(a, rank: int, world: int, dev, ops) -> Dict[str, Any]:
    import torch
    import torch.distributed as dist
    from . import phases as _ph

    t_run0 = time.perf_counter()
    budget = getattr(a, "budget", None)
    wall: Dict[str, float] = {}
    size = a.size
    gemm = None
    gpu = dev.type == "cuda"
    if not gpu:                                 # --cpu-rehearsal
        gemm = lambda x, w, out: out.copy_(x.float() @ w.float().t())
    dd = world > 1 or bool(getattr(a, "force_dist", False))       # a process group exists (one rank with --force-dist)
    load = _Load(torch, dist, ops, dev, world, size, gemm=gemm, distributed=dd)
    extras: Dict[str, Any] = {}
    shed: List[str] = []

    def allowed(section: str) -> bool:
        """Sections outside the timed region are skipped, never cut short, when the time budget cannot pay for them. Every
        rank holds the same budget clock to within the launch skew; sections with collectives ask rank 0 (see below)."""
        ok = budget is None or budget.allows(section, SECTION_COST_S[section])
        if dd and section in ("probe", "xgmi"):
            # a collective section: all ranks must take the same decision, rank 0's
            t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.broadcast(t, src=0)
            ok = bool(int(t.item()))
        if not ok:
            shed.append(section)
        return ok

    sampler = None
    period = a.sample_period if a.sample_period > 0 else 5.0
    if not a.no_sampler:
        sampler = _start_sampler(period, dev.index or 0)
    extras["sampler"] = (sampler.describe() if sampler is not None else "off")

    extras["flop_per_step"] = load.flops
    extras["gemm_variant"] = load.gemm_variant
    wall["setup"] = round(time.perf_counter() - t_run0, 2)
    t_sec = time.perf_counter()
    winfo: Dict[str, Any] = {}
    min_warm = float(getattr(a, "min_warmup_seconds", 2.0)) if gpu else 0.0
    dt = timed_steps(torch, dist, load, world, a.steps, a.warmup, distributed=dd, min_warm_s=min_warm, info=winfo)
    extras.update(winfo)
    wall["warmup_and_timed_steps"] = round(time.perf_counter() - t_sec, 3)
    wall["timed_steps_only"] = round(dt, 4)
    ms = dt / a.steps * 1e3
    tflops = load.flops * world / (dt / a.steps) / 1e12
    head = _headline(a, world, size, ms, tflops, dev)
    # from here on the number exists: the watchdog (bench.py) prints it if a later, untimed section never returns
    a.provisional = (head, extras)
    a.section = "after_timed_steps"
    if dd:
        algbw = load.allreduce_bytes / (dt / a.steps) / 1e9
        extras["allreduce_bytes_per_step"] = load.allreduce_bytes
        extras["allreduce_algbw_GBps_overlapped"] = round(algbw, 2)
        extras["allreduce_busbw_GBps_overlapped"] = round(algbw * 2 * (world - 1) / world, 2)
        extras["distributed_backend"] = dist.get_backend()

    # the same step on the kernel the N > 1 ranks run (one workgroup per tile, variant 4), still on one GPU: the point a
    # scaling curve's efficiency should be taken against, so that it does not include a kernel change
    a.section = "dp_kernel"
    if gpu and not dd and load.gemm_variant not in (None, 4) and size % 256 == 0 and allowed("dp_kernel"):
        t_sec = time.perf_counter()
        dp = _Load(torch, dist, ops, dev, world, size, distributed=False, dp_kernel=True)
        dt_dp = timed_steps(torch, dist, dp, world, a.steps, a.warmup, distributed=False, min_warm_s=min(1.0, min_warm))
        extras["value_dp_kernel"] = round(dp.flops * world / (dt_dp / a.steps) / 1e12, 2)
        extras["gemm_variant_dp_kernel"] = dp.gemm_variant
        del dp
        wall["dp_kernel_step"] = round(time.perf_counter() - t_sec, 2)
    elif dd and gpu:
        extras["value_dp_kernel"] = round(tflops / world, 2) if world == 1 else None
        extras["gemm_variant_dp_kernel"] = load.gemm_variant

    if sampler is not None:
        extras["sampler_stats"] = sampler.stats()

    a.section = "rccl_probe"
    t_sec = time.perf_counter()
    if not getattr(a, "no_probe", False) and allowed("probe"):
        # the RCCL all-reduce curve at EVERY n in {1, 2, 4, 8} <= world from this one launch (outside the
        # timed region): the first n ranks form their own communicator and sweep it (n = 1: the local copy rate)
        try:
            from ..parallel import dist_probe, probe
            sizes = probe.sweep_sizes(1024, getattr(a, "probe_max_bytes", 1 << 28), 4)
            if dd:
                curves = dist_probe.sweep_subgroups(dev, (1, 2, 4, 8), sizes, "bf16", warmup=3, iters=10,
                                                    ctrl=getattr(a, "ctrl_group", None))
                extras["rccl_ranks_seen"] = {"process_group_world_size": dist.get_world_size(),
                                             "communicators": [c.get("ranks_seen") for c in curves]}
            elif gpu:
                # one rank and no process group: the native probe (a child process) gives the n = 1 point
                curves = [probe.run_native(1, "bf16", max_bytes=sizes[-1], warmup=3, iters=10, timeout=120.0)]
                extras["rccl_ranks_seen"] = {"process_group_world_size": 1,
                                             "communicators": [c.get("header", {}).get("n_gpus") for c in curves]}
            else:
                curves = []
            if rank == 0 and curves:
                extras["rccl_curves"] = {str(c["n_gpus"]): probe.brief(c) for c in curves}
                top = curves[-1]
                extras["rccl_allreduce"] = dict(probe.brief(top), n_gpus=top["n_gpus"])
                try:
                    probe.save_results(probe.merge_results(probe.load_results(), probe.results_doc(curves, note="bench.py")))
                except OSError:
                    pass
        except Exception as e:
            extras["rccl_curves"] = {"error": f"{type(e).__name__}: {e}"}

    a.section = "xgmi_fidelity"
    which = getattr(a, "validation_phases", "all") or "all"
    xf = None
    if dd and not a.no_validation and (which == "all" or "xgmi" in which.split(",")) and allowed("xgmi"):
        # xGMI counter fidelity: only measurable with peers (every rank takes part in the collectives)
        try:
            xf = _ph.xgmi_fidelity(torch, dist, dev, rank, world)
        except Exception as e:
            xf = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            extras["xgmi_fidelity"] = xf

    wall["rccl_probe_and_xgmi_check"] = round(time.perf_counter() - t_sec, 2)
    t_sec = time.perf_counter()
    a.section = "other_kernels"
    if not a.no_validation and rank == 0 and gpu and which == "all" and allowed("other_kernels"):
        # the other validation-load kernels, 1 s each (outside the timed region)
        try:
            from ..loadgen.burner import Burner
            k = {}
            for kind in ("gemm_fp8", "gemm_fp4", "hbm_read", "hbm_fill", "hbm_copy"):
                b = Burner(kind, size=size, duty=100.0, device=dev.index or 0)
                o = b.run(1.0)
                k[kind] = round(o.get("tflops_busy") or o.get("TBps_busy") or 0.0, 2)
                del b
            torch.cuda.empty_cache()
            extras["other_kernels"] = dict(k, units={"gemm_*": "TFLOP/s", "hbm_*": "TB/s"})
            extras["vendor_comparison"] = vendor_comparison(torch, ops, dev, size)
        except Exception as e:
            extras["other_kernels"] = {"error": f"{type(e).__name__}: {e}"}
    a.section = "llm_serving"
    if not a.no_validation and rank == 0 and gpu and which == "all" and allowed("llm_serving"):
        extras["llm_serving"] = llm_serving_rates(torch, dev)

    wall["other_kernels_and_llm_serving"] = round(time.perf_counter() - t_sec, 2)
    t_sec = time.perf_counter()
    a.section = "validation_phases"
    local_phases = ",".join(p for p in which.split(",") if p != "xgmi")
    v: Dict[str, Any] = {}
    spec = getattr(a, "rehearsal_phases", "")
    if spec and not gpu and rank == 0:
        v = _rehearsal_validation(spec, budget)
    elif not a.no_validation and local_phases:
        try:
            t0 = time.time()
            v = _ph.run_all(torch, dist, ops, dev, rank, world, load, sampler, a, quick=True, which=local_phases, budget=budget)
            if v:
                extras["validation_seconds"] = round(time.time() - t0, 1)
        except Exception as e:  # the validation phases never invalidate the timed number
            extras["validation_error"] = f"{type(e).__name__}: {e}"
    if rank == 0 and (v or xf):
        if xf:
            # the unit verdict of the xGMI counters (KiB or KB per count) belongs to the report /api/validation serves: the
            # first run with peers pins it there
            v = dict(v)
            v["xgmi_fidelity"] = xf
            v["summary"] = _ph.summarize(v) if "rehearsal" not in (v.get("summary") or {}) else v["summary"]
        extras["validation"] = v.get("summary")
        extras["validation_detail"] = {k: x for k, x in v.items() if k != "summary"}
        shed += [p for p in (v.get("shed") or [])]

    if sampler is not None:
        sampler.stop()
    wall["validation_phases"] = round(time.perf_counter() - t_sec, 2)
    wall["total"] = round(time.perf_counter() - t_run0, 2)
    # only `timed_steps_only` enters `value`; everything else is reported, untimed work around it
    extras["wall_seconds"] = wall
    extras["time_budget_s"] = None if budget is None else budget.seconds
    extras["shed"] = shed

    a.section = "done"
    return {**head, **extras}


def _headline(a, world: int, size: int, ms: float, tflops: float, dev) -> Dict[str, Any]:
    """The contract's keys of the result line (everything else is `extras`)."""
    return {
        "metric": "monitored validation-load throughput: bf16 MFMA burner training step (fwd+dgrad+wgrad GEMMs, DP all-reduce over RCCL) under telemetry sampling",
        "value": round(tflops, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (uniform [-1,1) operands, random init)" + ("; CPU REHEARSAL, not a benchmark" if dev.type != "cuda" else ""),
        "config": {
            "model": "klm bf16 GEMM burner, one %dx%d linear layer: fwd+dgrad+wgrad (hand-written gfx950 MFMA kernel) + RCCL all-reduce of dW" % (size, size),
            "global_batch": world,
            "seq_len": size,
            "parallelism": f"dp{world}" if world > 1 else "single",
        },
    }


class Watchdog:
    """Last line of defence for the output contract. The time budget sheds sections BEFORE they start; it cannot help when a
    section that was started never returns (a collective on a communicator one rank never joined, a wedged child). At
    `deadline_s` after the run's start every rank's watchdog thread ends its process; rank 0 first prints the compact line
    from what is known by then -- the timed number (complete since `a.provisional` was set) plus `watchdog`: which section
    was running -- so that the driver records the measurement instead of a timeout. Before the timed steps have finished
    there is no number to print and the exit status says so (3)."""

    def __init__(self, a, rank: int, deadline_s: float, t0: float, out=None, exit_fn=None):
        import threading
        self.a, self.rank, self.deadline_s, self.t0 = a, rank, deadline_s, t0
        self.out, self.exit_fn = out, exit_fn
        self._cancel = threading.Event()
        self._th = threading.Thread(target=self._run, name="klm-bench-watchdog", daemon=True)

    def start(self) -> "Watchdog":
        self._th.start()
        return self

    def cancel(self) -> None:
        self._cancel.set()

    def line(self) -> Optional[Dict[str, Any]]:
        prov = getattr(self.a, "provisional", None)
        if prov is None:
            return None
        head, extras = prov
        res = {**head, **dict(extras)}
        res.setdefault("shed", [])
        res.setdefault("wall_seconds", {"total": round(time.monotonic() - self.t0, 2), "incomplete": True})
        res["watchdog"] = {"fired_after_s": round(time.monotonic() - self.t0, 1), "section": getattr(self.a, "section", "?")}
        return res

    def _run(self) -> None:
        import json
        import os
        import sys
        # rank 0 prints at the deadline and leaves last: a rank waiting for it at the closing (gloo) barrier would see its exit
        # as a broken connection, so the others go 2 s after the deadline and rank 0 4 s after it
        wait = self.deadline_s - (time.monotonic() - self.t0) + (0.0 if self.rank == 0 else 2.0)
        if self._cancel.wait(max(0.0, wait)):
            return
        code = 0
        if self.rank == 0:
            res = self.line()
            if res is None:
                print(f"[bench] watchdog: no timed result after {self.deadline_s:.0f} s (section {getattr(self.a, 'section', 'start')})",
                      file=sys.stderr, flush=True)
                code = 3
            else:
                path = ""
                try:
                    path = write_detail(res, detail_path(self.a))
                except Exception:
                    path = ""
                print(f"[bench] watchdog: section {res['watchdog']['section']} did not return; printing the timed result",
                      file=sys.stderr, flush=True)
                line = compact(res, path)
                line["watchdog"] = res["watchdog"]
                (self.out or sys.stdout).write(json.dumps(line) + "\n")
                (self.out or sys.stdout).flush()
            if self.exit_fn is None:
                time.sleep(4.0)
        elif getattr(self.a, "provisional", None) is None:
            code = 3
        (self.exit_fn or os._exit)(code)


def llm_serving_rates(torch, dev) -> Dict[str, Any]:
    """The serving load generator (Llama-3-8B architecture, random weights): decode and prefill rates."""
    import gc
    try:
        from ..serving import Engine, EngineConfig
        from ..serving import bench as sb
        eng = Engine(EngineConfig(model="llama3-8b", device=str(dev), max_num_seqs=256, max_model_len=4096,
                                  kv_blocks=20000))
        eng.warmup([1, 16, 64, 192, 256])
        dec = [sb.decode_step_ms(eng, b, 1024, iters=10) for b in (1, 16, 64, 192, 256)]   # skinny / skinny / mid / tiled GEMM paths
        pre = sb.prefill_ms(eng, 4, 1024, iters=2)
        out: Dict[str, Any] = {
            "model": "llama3-8b (random weights, bf16)", "ctx": 1024,
            "decode": [{"batch": int(r["batch"]), "ms_per_step": round(r["ms"], 3),
                        "tokens_per_s": round(r["tokens_per_s"], 1), "TBps": round(r["TBps"], 2)} for r in dec],
            "prefill": {"n_seqs": 4, "prompt_len": 1024, "tokens_per_s": round(pre["tokens_per_s"], 1),
                        "TFLOPs": round(pre["TFLOPs"], 1), "gemm_backend": eng.model.prefill_gemm_backend(4 * 1024)},
            "decode_gemm_backend": "hand-written skinny / mid-batch / K-split tiled kernels up to 256 rows "
                                   "(gate_up, and wo above 208 rows, on hipBLASLt from 129 rows)"}
        # the same decode with W8A16 weight copies (labelled: not the bf16 figure above)
        eng.model.enable_fp8_decode()
        eng._graphs.clear()
        eng.warmup([1, 16, 64])
        d8 = [sb.decode_step_ms(eng, b, 1024, iters=10) for b in (1, 16, 64)]
        out["decode_fp8_weights"] = [
            {"batch": int(r["batch"]), "ms_per_step": round(r["ms"], 3), "tokens_per_s": round(r["tokens_per_s"], 1),
             "TBps": round(r["TBps"], 2)} for r in d8]
        del eng
        gc.collect()
        torch.cuda.empty_cache()
        # and with an e4m3 KV cache on top (large batches, where attention bytes dominate)
        eng = Engine(EngineConfig(model="llama3-8b", device=str(dev), max_num_seqs=256, max_model_len=4096,
                                  kv_blocks=20000, kv_cache_dtype="fp8"))
        eng.warmup([64, 256])
        dk = [sb.decode_step_ms(eng, b, 1024, iters=10) for b in (64, 256)]
        out["decode_fp8_kv_cache"] = [
            {"batch": int(r["batch"]), "ms_per_step": round(r["ms"], 3), "tokens_per_s": round(r["tokens_per_s"], 1),
             "TBps": round(r["TBps"], 2)} for r in dk]
        del eng, dec, pre, d8, dk
        gc.collect()
        torch.cuda.empty_cache()
        return out
    except Exception as e:
        gc.collect()
        torch.cuda.empty_cache()
        return {"error": f"{type(e).__name__}: {e}"}


# ---- the output contract: ONE compact JSON line on stdout, everything else in a file --------------------------------
#: keys of the full result that go into the compact line as they are
_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "flop_per_step", "gemm_variant", "value_dp_kernel", "gemm_variant_dp_kernel",
              "warmup_steps_run", "warmup_seconds", "allreduce_bytes_per_step", "allreduce_busbw_GBps_overlapped",
              "distributed_backend", "rccl_ranks_seen", "launcher", "time_budget_s", "shed", "validation_error")
#: BASELINE.json's metric ("telemetry error vs amdsmi/rocprof (%), sampling overhead, alert latency"): flat scalars of the summary
_SUMMARY_KEYS = ("hbm_error_pct", "gfx_activity_error_points", "power_error_pct", "sampling_overhead_pct", "sampling_overhead_ci95_pct",
                 "sampling_overhead_high_pct", "alert_latency_max_s", "alert_latency_crossings", "sse_latency_max_s", "poll_latency_max_s",
                 "power_alert", "throttle_alert", "vllm_tokens_per_s_error_pct", "vllm_kv_cache_error_points",
                 "xgmi_unit_verdict_tx", "xgmi_tx_error_pct")
LINE_LIMIT = 4096


def detail_path(a=None) -> str:
    """Where the full result goes: --detail-out, else $KLM_BENCH_DETAIL, else bench_detail.json next to bench.py (the
    repository root; the state directory when that is not writable)."""
    import os
    p = getattr(a, "detail_out", "") or os.environ.get("KLM_BENCH_DETAIL", "")
    if p:
        return p
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if os.access(root, os.W_OK):
        return os.path.join(root, "bench_detail.json")
    from ..server.http import validation_report_path
    return os.path.join(os.path.dirname(validation_report_path()), "bench_detail.json")


def _round_floats(x, nd: int = 4):
    if isinstance(x, float):
        return round(x, nd)
    if isinstance(x, dict):
        return {k: _round_floats(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round_floats(v, nd) for v in x]
    return x


def compact(result: Dict[str, Any], detail: str = "") -> Dict[str, Any]:
    """The line bench.py prints: the contract's keys, BASELINE.json's metric as flat scalars, which kernel ran, how many
    ranks RCCL saw and the peak bus bandwidth per n -- nothing that grows with the run (the driver keeps 8 000 characters
    of stdout; round 2's 20.8 KB line was recorded as `parsed: null`)."""
    line = {k: result[k] for k in _LINE_KEYS if k in result}
    summ = result.get("validation") or {}
    for k in _SUMMARY_KEYS:
        if summ.get(k) is not None:
            line[k] = summ[k]
    if summ.get("pass") is not None:
        line["validation_pass"] = bool(summ["pass"]) and all(summ["pass"].values())
        line["validation_failed"] = sorted(k for k, ok in summ["pass"].items() if not ok)
        line["validation_checks"] = len(summ["pass"])
    for name in ("rccl_curves", "rccl_curves_native"):
        cs = result.get(name)
        if isinstance(cs, dict) and "error" not in cs:
            # n = 1 carries the local copy rate (no link traffic); an unverified point is null, never a number
            line["peak_busbw_GBps" + ("_native" if name.endswith("native") else "")] = {
                n: ((c.get("peak_busbw_GBps") if int(n) > 1 else c.get("peak_algbw_GBps")) if c.get("verified") else None)
                for n, c in sorted(cs.items(), key=lambda kv: int(kv[0]))}
        elif isinstance(cs, dict):
            line[name + "_error"] = str(cs["error"])[:200]
    vc = result.get("vendor_comparison") or {}
    for k in ("bf16_klm_over_hipblaslt", "fp8_klm_over_scaled_mm"):
        if k in vc:
            line[k] = vc[k]
    se = vc.get("step_energy") or {}
    if "joule_per_step" in (se.get("klm") or {}):
        line["joule_per_step"] = se["klm"]["joule_per_step"]
        if "joule_per_step" in (se.get("hipblaslt") or {}):
            line["joule_per_step_hipblaslt"] = se["hipblaslt"]["joule_per_step"]
    if isinstance(vc.get("training_step"), dict) and "klm_over_hipblaslt" in vc["training_step"]:
        line["step_klm_over_hipblaslt"] = vc["training_step"]["klm_over_hipblaslt"]
    if "wall_seconds" in result:
        line["wall_total_s"] = result["wall_seconds"].get("total")
    if detail:
        line["detail"] = detail
    line = _round_floats(line)
    # the line must stay a line the driver can keep whole: drop the optional parts, longest first, if it ever grows past the limit
    import json
    optional = [k for k in line if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                             "scaling", "vs_baseline", "dtype", "data", "config", "detail")]
    while len(json.dumps(line)) >= LINE_LIMIT and optional:
        worst = max(optional, key=lambda k: len(json.dumps(line[k])))
        optional.remove(worst)
        line.pop(worst)
        line["truncated"] = True
    return line


def write_detail(result: Dict[str, Any], path: str) -> str:
    """The full result (curves, per-phase rows, llm_serving, vendor_comparison) as a JSON file, and the validation report
    where /api/validation reads it."""
    import json
    import os
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "w", encoding="utf-8") as f:
        json.dump(result, f, indent=1)
    os.replace(tmp, path)
    if result.get("validation") and not (result["validation"] or {}).get("rehearsal"):
        try:
            from ..server.http import validation_report_path
            rp = validation_report_path()
            os.makedirs(os.path.dirname(rp) or ".", exist_ok=True)
            doc = dict(result.get("validation_detail") or {}, summary=result["validation"], timestamp=time.time(), source="bench.py")
            with open(rp + ".tmp", "w", encoding="utf-8") as f:
                json.dump(doc, f, indent=1)
            os.replace(rp + ".tmp", rp)
        except OSError:
            pass
    return path
