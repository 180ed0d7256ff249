"""RCCL all-reduce probe, one process per GPU (``torch.distributed``, backend "nccl" = RCCL).

Same sweep and point records as the native probe (parallel/probe.py); this is the executor
``bench.py`` uses under ``torch.distributed.run`` and the one the CPU tests drive with the
``gloo`` backend at world_size 2 (the 2/4/8-GPU code path must be
exercised without 8 GPUs). No reference counterpart.

Every rank calls ``sweep``; rank 0 gets the curve. Timing per size: warm-up rounds, a
barrier, ``iters`` back-to-back all-reduces bracketed by device synchronisation, MAX over
ranks. Collectives run on large flat buffers (bf16 by default) sized for 288 GB HBM:
the sweep goes to 1 GiB per rank by default.

Launch by hand:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
                     -m k8s_llm_monitor_amd.parallel.dist_probe --max-bytes 1073741824
"""
from __future__ import annotations

import argparse
import json
import os
import time
from typing import Any, Dict, List, Optional, Sequence

from . import probe

OPS = ("allreduce", "allgather", "reducescatter", "alltoall")


def _dtype(torch, name: str):
    return {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[name]


def _sync(torch, dev) -> None:
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def run_op(dist, op: str, x, y, world: int, group=None):
    if op == "allreduce" and world == 1:
        # one rank: the collective is a no-op on the in-place buffer torch.distributed hands it. The n = 1 point is the
        # device-local copy rate, so move the bytes for real, out of place.
        y.copy_(x)
    elif op == "allreduce":
        dist.all_reduce(x, group=group)
    elif op == "allgather":
        dist.all_gather_into_tensor(y, x[: x.numel() // world], group=group)
    elif op == "reducescatter":
        dist.reduce_scatter_tensor(y[: x.numel() // world], x, group=group)
    elif op == "alltoall":
        dist.all_to_all_single(y, x, group=group)
    else:
        raise ValueError(op)


def sweep(dev, sizes: Optional[Sequence[int]] = None, dtype: str = "bf16", op: str = "allreduce",
          warmup: int = 5, iters: int = 20, check: bool = True, group=None) -> Optional[Dict[str, Any]]:
    """Run the size sweep on `group` (default: the default process group); every member of the group
    calls it. Returns the curve on the group's rank 0, None elsewhere. The buffer is reduced IN PLACE
     and the data check covers every element. A one-rank group has no traffic to time: its
    curve is the out-of-place device-local copy rate, labelled mode "local_copy"; every point is held against
    `probe.bound_GBps` and a curve with an implausible point comes back unverified."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    esz = probe.DTYPE_BYTES[dtype]
    sizes = list(sizes) if sizes is not None else probe.sweep_sizes()
    tdt = _dtype(torch, dtype)
    gran = esz * world
    max_b = max(sizes)
    x = torch.empty(max_b // esz, dtype=tdt, device=dev)
    y = torch.empty(max_b // esz, dtype=tdt, device=dev) if (op != "allreduce" or world == 1) else None
    pts: List[Dict[str, Any]] = []
    for nbytes in sizes:
        nbytes -= nbytes % gran
        if nbytes <= 0:
            continue
        n = nbytes // esz
        xv = x[:n]
        yv = y[:n] if y is not None else None
        verdict = None
        if check and op == "allreduce":
            xv.fill_(float(rank + 1))
            dist.all_reduce(xv, group=group)
            _sync(torch, dev)
            want = world * (world + 1) / 2.0            # exact in bf16 up to 22 ranks
            verdict = "ok" if bool((xv == want).all().item()) else "MISMATCH"
            if world == 1:                              # ... and the copy the timed rounds run lands every element
                yv.zero_()
                run_op(dist, op, xv, yv, world, group)
                _sync(torch, dev)
                verdict = "ok" if verdict == "ok" and bool((yv == want).all().item()) else "MISMATCH"
        xv.fill_(1.0 / world if op == "allreduce" else 1.0)      # keeps repeated sums finite
        for _ in range(warmup):
            run_op(dist, op, xv, yv, world, group)
            if op == "allreduce":
                xv.mul_(1.0 / world)
        _sync(torch, dev)
        dist.barrier(group=group)
        _sync(torch, dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            run_op(dist, op, xv, yv, world, group)
        _sync(torch, dev)
        dt = (time.perf_counter() - t0) / iters
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        pts.append(probe.point(world, nbytes, float(t.item()), dtype, op, verdict))
    if rank != 0:
        return None
    ok = all(p["check"] in (None, "ok") for p in pts)
    c = probe.curve(world, pts, dtype, verified=ok, executor="torch.distributed/" + dist.get_backend(group))
    c["op"] = op
    c["ranks_seen"] = world                              # what the communicator itself reports
    return c


def sweep_subgroups(dev, counts: Sequence[int] = (1, 2, 4, 8), sizes: Optional[Sequence[int]] = None,
                    dtype: str = "bf16", warmup: int = 5, iters: int = 20, ctrl=None) -> List[Dict[str, Any]]:
    """The whole 1/2/4/8 curve from ONE launch: for every n <= world the first n ranks form their own
    communicator (``dist.new_group``; RCCL builds it over just those GPUs' xGMI links) and sweep it while
    the others wait on `ctrl` (a gloo group: waiting there submits nothing to the GPU, whereas an RCCL
    barrier would park a spinning kernel on the idle devices). Every rank of the default group calls
    this; rank 0 gets the list of curves, the others []."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    curves: List[Dict[str, Any]] = []
    for n in sorted({int(c) for c in counts if 1 <= int(c) <= world}):
        g = dist.new_group(ranks=list(range(n))) if n < world else None     # collective over the default group
        if rank < n:
            c = sweep(dev, sizes, dtype, "allreduce", warmup, iters, group=g)
            if c is not None:
                curves.append(c)
        dist.barrier(group=ctrl)
        if g is not None:
            dist.destroy_process_group(g)
    return curves


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--backend", default="", help="nccl (RCCL) on GPUs, gloo on CPU; default by device")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--op", default="allreduce", choices=OPS)
    ap.add_argument("--min-bytes", type=int, default=1024)
    ap.add_argument("--max-bytes", type=int, default=1 << 30)
    ap.add_argument("--factor", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="", help="results JSON merged for the dashboard ('' = default state path)")
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    use_gpu = not a.cpu and torch.cuda.device_count() > local
    dev = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(dev)
    backend = a.backend or ("nccl" if use_gpu else "gloo")
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if use_gpu else {}))
    try:
        c = sweep(dev, probe.sweep_sizes(a.min_bytes, a.max_bytes, a.factor), a.dtype, a.op, a.warmup, a.iters)
        if rank == 0 and c is not None:
            doc = probe.merge_results(probe.load_results(a.out), probe.results_doc([c], note=f"{world} rank(s), {backend}"))
            path = probe.save_results(doc, a.out)
            for p in c["points"]:
                print(json.dumps(p))
            print(json.dumps({k: v for k, v in c.items() if k != "points"}))
            print("saved:", path)
    finally:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
