"""The timed loop and the validation phases behind ``bench.py``.

No reference counterpart (the reference has no benchmark); the
quantities are the self-defined targets of BASELINE.md: telemetry error, sampling
overhead, alert latency and the RCCL bus bandwidth at N GPUs.
"""
from __future__ import annotations

import time
from typing import Any, Dict


def _sync_all(torch, dist, world):
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if gpu:
        torch.cuda.synchronize()


class _Load:
    """One rank's buffers and the step function.

    A step is the GEMM work of one linear layer's training step on the burner kernel, with
    the data-parallel gradient all-reduce overlapped the way DP training overlaps it:

        fwd    Y  = X  @ W^T          (compute stream)
        dgrad  dX = dY @ (W^T)^T      (compute stream)
        wgrad  dW = dY^T @ (X^T)^T    (compute stream)  -> all-reduce(dW) over RCCL on the
                                                           communication stream, overlapping
                                                           the NEXT step's fwd and dgrad
    All operands are square `size` x `size` bf16, stored K-major as the kernel wants them
    (the transposed copies are made once, outside the timed region). dW is double-buffered:
    the wgrad of step i+2 waits for the all-reduce of step i. 6*size^3 FLOP and one
    2*size^2-byte all-reduce per step.
    """

    def __init__(self, torch, dist, ops, dev, world: int, size: int, gemm=None):
        self.torch, self.dist, self.ops, self.world = torch, dist, ops, world
        # `gemm(a, b, out)` computes out = a @ b.T; the default is the hand-written MFMA kernel.
        # CPU tests of the stream/event/all-reduce logic pass a torch.matmul stand-in.
        self.gemm = gemm or (lambda a, b, out: ops.gemm_bf16(a, b, out=out, variant=1))
        self.gpu = dev.type == "cuda"
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + (dist.get_rank() if world > 1 else 0))

        def rnd():
            # uniform [-1, 1) operands: zero-filled operands read ~15-20 % high (DVFS)
            return (torch.rand(size, size, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)

        self.x, self.w, self.dy = rnd(), rnd(), rnd()
        self.wt, self.dyt, self.xt = self.w.t().contiguous(), self.dy.t().contiguous(), self.x.t().contiguous()
        self.a, self.b = self.dyt, self.xt                     # the operands of the all-reduced GEMM (tests)
        self.y = torch.empty(size, size, device=dev, dtype=torch.bfloat16)
        self.dx = torch.empty(size, size, device=dev, dtype=torch.bfloat16)
        self.c = [torch.empty(size, size, device=dev, dtype=torch.bfloat16) for _ in range(2)]    # dW, double-buffered
        self.size = size
        self.flops = 3 * 2.0 * size * size * size
        self.comm = torch.cuda.Stream(device=dev) if (world > 1 and self.gpu) else None
        self.ev_gemm = [torch.cuda.Event() for _ in range(2)] if self.gpu else []
        self.ev_comm = [torch.cuda.Event() for _ in range(2)] if self.gpu else []
        self.i = 0
        self.allreduce_bytes = size * size * 2 if world > 1 else 0

    def step(self):
        torch = self.torch
        k = self.i & 1
        if not self.gpu:                       # CPU rehearsal: same data flow, no streams
            self.gemm(self.x, self.w, self.y)
            self.gemm(self.dy, self.wt, self.dx)
            self.gemm(self.dyt, self.xt, self.c[k])
            if self.world > 1:
                self.dist.all_reduce(self.c[k])
            self.i += 1
            return
        main = torch.cuda.current_stream()
        self.gemm(self.x, self.w, self.y)
        self.gemm(self.dy, self.wt, self.dx)
        if self.comm is not None:
            # dW buffer k was all-reduced two steps ago: wgrad may only overwrite it after that
            main.wait_event(self.ev_comm[k])
        self.gemm(self.dyt, self.xt, self.c[k])
        if self.comm is not None:
            self.ev_gemm[k].record(main)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.ev_gemm[k])
                self.dist.all_reduce(self.c[k])
                self.ev_comm[k].record(self.comm)
        self.i += 1

    def drain(self):
        if self.comm is not None:
            self.torch.cuda.current_stream().wait_stream(self.comm)


def timed_steps(torch, dist, load: _Load, world: int, steps: int, warmup: int) -> float:
    """Seconds for exactly `steps` steps: barrier + synchronize on both sides, MAX over ranks."""
    for _ in range(warmup):
        load.step()
    load.drain()
    _sync_all(torch, dist, world)
    t0 = time.perf_counter()
    for _ in range(steps):
        load.step()
    load.drain()
    _sync_all(torch, dist, world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=load.a.device, dtype=torch.float64)
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


def run(a, rank: int, world: int, dev, ops) -> Dict[str, Any]:
    import torch
    import torch.distributed as dist

    size = a.size
    load = _Load(torch, dist, ops, dev, world, size)
    extras: Dict[str, Any] = {}

    sampler = None
    period = a.sample_period if a.sample_period > 0 else 5.0
    if not a.no_sampler:
        sampler = _start_sampler(period, dev.index or 0)
    extras["sampler"] = (sampler.describe() if sampler is not None else "off")

    extras["flop_per_step"] = load.flops
    dt = timed_steps(torch, dist, load, world, a.steps, a.warmup)
    ms = dt / a.steps * 1e3
    tflops = load.flops * world / (dt / a.steps) / 1e12
    if world > 1:
        algbw = load.allreduce_bytes / (dt / a.steps) / 1e9
        extras["allreduce_bytes_per_step"] = load.allreduce_bytes
        extras["allreduce_algbw_GBps_overlapped"] = round(algbw, 2)
        extras["allreduce_busbw_GBps_overlapped"] = round(algbw * 2 * (world - 1) / world, 2)

    if sampler is not None:
        extras["sampler_stats"] = sampler.stats()

    if world > 1 and not getattr(a, "no_probe", False):
        # the RCCL all-reduce curve at this GPU count (outside the timed region)
        try:
            from ..parallel import dist_probe, probe
            c = dist_probe.sweep(dev, probe.sweep_sizes(1024, getattr(a, "probe_max_bytes", 1 << 28), 4), "bf16",
                                 warmup=3, iters=10)
            if c is not None:
                extras["rccl_allreduce"] = {"peak_busbw_GBps": round(c["peak_busbw_GBps"], 2), "peak_bytes": c["peak_bytes"],
                                            "ceiling_mesh_GBps": c["ceiling_mesh_GBps"], "ceiling_ring_GBps": c["ceiling_ring_GBps"],
                                            "verified": c["verified"],
                                            "curve": [[p["bytes"], round(p["busbw_GBps"], 2)] for p in c["points"]]}
                try:
                    probe.save_results(probe.merge_results(probe.load_results(), probe.results_doc([c], note="bench.py")))
                except OSError:
                    pass
        except Exception as e:
            extras["rccl_allreduce"] = {"error": f"{type(e).__name__}: {e}"}

    if not a.no_validation and rank == 0:
        # the other validation-load kernels, 1 s each (outside the timed region)
        try:
            from ..loadgen.burner import Burner
            k = {}
            for kind in ("gemm_fp8", "hbm_read", "hbm_fill", "hbm_copy"):
                b = Burner(kind, size=size, duty=100.0, device=dev.index or 0)
                o = b.run(1.0)
                k[kind] = round(o.get("tflops_busy") or o.get("TBps_busy") or 0.0, 2)
                del b
            torch.cuda.empty_cache()
            extras["other_kernels"] = dict(k, units={"gemm_fp8": "TFLOP/s", "hbm_*": "TB/s"})
        except Exception as e:
            extras["other_kernels"] = {"error": f"{type(e).__name__}: {e}"}

    if not a.no_validation:
        try:
            from . import phases
            t0 = time.time()
            v = phases.run_all(torch, dist, ops, dev, rank, world, load, sampler, a, quick=True)
            if v:
                extras["validation"] = v.get("summary")
                extras["validation_detail"] = {k: x for k, x in v.items() if k != "summary"}
                extras["validation_seconds"] = round(time.time() - t0, 1)
        except Exception as e:  # the validation phases never invalidate the timed number
            extras["validation_error"] = f"{type(e).__name__}: {e}"

    if sampler is not None:
        sampler.stop()

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
        "data": "synthetic (uniform [-1,1) operands, random init)",
        "config": {
            "model": "klm bf16 GEMM burner, one %dx%d linear layer: fwd+dgrad+wgrad (hand-written gfx950 MFMA kernel) + RCCL all-reduce of dW" % (size, size),
            "global_batch": world,
            "seq_len": size,
            "parallelism": f"dp{world}" if world > 1 else "single",
        },
        **extras,
    }
