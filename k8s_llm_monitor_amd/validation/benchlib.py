"""The timed loop and the validation phases behind ``bench.py``.

No reference counterpart (the reference has no benchmark); the
quantities are the self-defined targets of BASELINE.md: telemetry error, sampling
overhead, alert latency and the RCCL bus bandwidth at N GPUs.
"""
from __future__ import annotations

import time
from typing import Any, Dict


def _sync_all(torch, dist, distributed):
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    if gpu:
        torch.cuda.synchronize()


from ..models.linear_step import LinearLayerStep as _Load   # noqa: E402  (kept under its old name for callers)


def timed_steps(torch, dist, load: _Load, world: int, steps: int, warmup: int, distributed=None) -> float:
    """Seconds for exactly `steps` steps: barrier + synchronize on both sides, MAX over ranks."""
    dd = (world > 1) if distributed is None else bool(distributed)   # a one-rank group still takes the collective path
    for _ in range(warmup):
        load.step()
    load.drain()
    _sync_all(torch, dist, dd)
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


def run(a, rank: int, world: int, dev, ops) -> Dict[str, Any]:
    import torch
    import torch.distributed as dist

    t_run0 = time.perf_counter()
    wall: Dict[str, float] = {}
    size = a.size
    gemm = None
    if dev.type != "cuda":                      # --cpu-rehearsal
        gemm = lambda x, w, out: out.copy_(x.float() @ w.float().t())
    dd = world > 1 or bool(getattr(a, "force_dist", False))       # a process group exists (one rank with --force-dist)
    load = _Load(torch, dist, ops, dev, world, size, gemm=gemm, distributed=dd)
    extras: Dict[str, Any] = {}

    sampler = None
    period = a.sample_period if a.sample_period > 0 else 5.0
    if not a.no_sampler:
        sampler = _start_sampler(period, dev.index or 0)
    extras["sampler"] = (sampler.describe() if sampler is not None else "off")

    extras["flop_per_step"] = load.flops
    wall["setup"] = round(time.perf_counter() - t_run0, 2)
    t_sec = time.perf_counter()
    dt = timed_steps(torch, dist, load, world, a.steps, a.warmup, distributed=dd)
    wall["warmup_and_timed_steps"] = round(time.perf_counter() - t_sec, 3)
    wall["timed_steps_only"] = round(dt, 4)
    ms = dt / a.steps * 1e3
    tflops = load.flops * world / (dt / a.steps) / 1e12
    if dd:
        algbw = load.allreduce_bytes / (dt / a.steps) / 1e9
        extras["allreduce_bytes_per_step"] = load.allreduce_bytes
        extras["allreduce_algbw_GBps_overlapped"] = round(algbw, 2)
        extras["allreduce_busbw_GBps_overlapped"] = round(algbw * 2 * (world - 1) / world, 2)
        extras["distributed_backend"] = dist.get_backend()

    if sampler is not None:
        extras["sampler_stats"] = sampler.stats()

    t_sec = time.perf_counter()
    if not getattr(a, "no_probe", False):
        # the RCCL all-reduce curve at EVERY n in {1, 2, 4, 8} <= world from this one launch (outside the
        # timed region): the first n ranks form their own communicator and sweep it in place
        try:
            from ..parallel import dist_probe, probe
            sizes = probe.sweep_sizes(1024, getattr(a, "probe_max_bytes", 1 << 28), 4)
            if dd:
                curves = dist_probe.sweep_subgroups(dev, (1, 2, 4, 8), sizes, "bf16", warmup=3, iters=10,
                                                    ctrl=getattr(a, "ctrl_group", None))
                extras["rccl_ranks_seen"] = {"process_group_world_size": dist.get_world_size(),
                                             "communicators": [c.get("ranks_seen") for c in curves]}
            elif dev.type == "cuda":
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

    which = getattr(a, "validation_phases", "all") or "all"
    if dd and not a.no_validation and (which == "all" or "xgmi" in which.split(",")):
        # xGMI counter fidelity: only measurable with peers (every rank takes part in the collectives)
        try:
            from . import phases as _ph
            xf = _ph.xgmi_fidelity(torch, dist, dev, rank, world)
            if rank == 0:
                extras["xgmi_fidelity"] = xf
        except Exception as e:
            if rank == 0:
                extras["xgmi_fidelity"] = {"error": f"{type(e).__name__}: {e}"}

    wall["rccl_probe_and_xgmi_check"] = round(time.perf_counter() - t_sec, 2)
    t_sec = time.perf_counter()
    if not a.no_validation and rank == 0 and dev.type == "cuda" and which == "all":
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
        # the serving load generator (Llama-3-8B architecture, random weights): decode and prefill rates
        try:
            from ..serving import Engine, EngineConfig
            from ..serving import bench as sb
            eng = Engine(EngineConfig(model="llama3-8b", device=str(dev), max_num_seqs=256, max_model_len=4096,
                                      kv_blocks=20000))
            eng.warmup([1, 16, 64, 192, 256])
            dec = [sb.decode_step_ms(eng, b, 1024, iters=10) for b in (1, 16, 64, 192, 256)]   # skinny / skinny / mid / tiled GEMM paths
            pre = sb.prefill_ms(eng, 4, 1024, iters=2)
            extras["llm_serving"] = {
                "model": "llama3-8b (random weights, bf16)", "ctx": 1024,
                "decode": [{"batch": int(r["batch"]), "ms_per_step": round(r["ms"], 3),
                            "tokens_per_s": round(r["tokens_per_s"], 1), "TBps": round(r["TBps"], 2)} for r in dec],
                "prefill": {"n_seqs": 4, "prompt_len": 1024, "tokens_per_s": round(pre["tokens_per_s"], 1),
                            "TFLOPs": round(pre["TFLOPs"], 1),
                            "gemm_backend": "hipBLASLt (torch.matmul): NOT a hand-written-kernel figure; only attention, "
                                            "norms, rope and the KV write are this repo's kernels in prefill"},
                "decode_gemm_backend": "hand-written skinny / mid-batch / K-split tiled kernels up to 256 rows "
                                       "(gate_up, and wo above 208 rows, on hipBLASLt from 129 rows)"}
            # the same decode with W8A16 weight copies (labelled: not the bf16 figure above)
            eng.model.enable_fp8_decode()
            eng._graphs.clear()
            eng.warmup([1, 16, 64])
            d8 = [sb.decode_step_ms(eng, b, 1024, iters=10) for b in (1, 16, 64)]
            extras["llm_serving"]["decode_fp8_weights"] = [
                {"batch": int(r["batch"]), "ms_per_step": round(r["ms"], 3), "tokens_per_s": round(r["tokens_per_s"], 1),
                 "TBps": round(r["TBps"], 2)} for r in d8]
            del eng
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            # and with an e4m3 KV cache on top (large batches, where attention bytes dominate)
            eng = Engine(EngineConfig(model="llama3-8b", device=str(dev), max_num_seqs=256, max_model_len=4096,
                                      kv_blocks=20000, kv_cache_dtype="fp8"))
            eng.warmup([64, 256])
            dk = [sb.decode_step_ms(eng, b, 1024, iters=10) for b in (64, 256)]
            extras["llm_serving"]["decode_fp8_kv_cache"] = [
                {"batch": int(r["batch"]), "ms_per_step": round(r["ms"], 3), "tokens_per_s": round(r["tokens_per_s"], 1),
                 "TBps": round(r["TBps"], 2)} for r in dk]
            del eng, dec, pre, d8, dk
            gc.collect()
            torch.cuda.empty_cache()
        except Exception as e:
            extras["llm_serving"] = {"error": f"{type(e).__name__}: {e}"}

    wall["other_kernels_and_llm_serving"] = round(time.perf_counter() - t_sec, 2)
    t_sec = time.perf_counter()
    local_phases = ",".join(p for p in which.split(",") if p != "xgmi")
    if not a.no_validation and local_phases:
        try:
            from . import phases
            t0 = time.time()
            v = phases.run_all(torch, dist, ops, dev, rank, world, load, sampler, a, quick=True, which=local_phases)
            if v:
                extras["validation"] = v.get("summary")
                extras["validation_detail"] = {k: x for k, x in v.items() if k != "summary"}
                extras["validation_seconds"] = round(time.time() - t0, 1)
        except Exception as e:  # the validation phases never invalidate the timed number
            extras["validation_error"] = f"{type(e).__name__}: {e}"

    if sampler is not None:
        sampler.stop()
    wall["validation_phases"] = round(time.perf_counter() - t_sec, 2)
    wall["total"] = round(time.perf_counter() - t_run0, 2)
    # only `timed_steps_only` enters `value`; everything else is reported, untimed work around it
    extras["wall_seconds"] = wall

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
        **extras,
    }
