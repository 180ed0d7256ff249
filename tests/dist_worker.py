"""Worker run under ``torch.distributed.run`` by test_distributed.py (gloo, CPU)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from k8s_llm_monitor_amd.parallel import dist_probe, probe
    from k8s_llm_monitor_amd.validation import benchlib
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    out = {}
    # 1. the probe sweep: data check, timing reduction, curve on rank 0 only
    c = dist_probe.sweep(dev, [1024, 4096, 65536], "fp32", warmup=1, iters=3)
    assert (c is not None) == (rank == 0)
    if rank == 0:
        out["curve"] = {k: c[k] for k in ("n_gpus", "verified", "executor", "peak_busbw_GBps")}
        out["checks"] = [p["check"] for p in c["points"]]
        out["bytes"] = [p["bytes"] for p in c["points"]]
    for op in ("allgather", "reducescatter", "alltoall"):
        c2 = dist_probe.sweep(dev, [4096], "fp32", op=op, warmup=1, iters=2)
        if rank == 0:
            out[op] = c2["points"][0]["busbw_GBps"] > 0
    # 2. the bench's step loop: GEMM + all-reduce of its output, timed with barriers, MAX over ranks
    size = 64
    load = benchlib._Load(torch, dist, None, dev, world, size,
                          gemm=lambda a, b, o: o.copy_(a.float() @ b.float().t()))
    dt = benchlib.timed_steps(torch, dist, load, world, steps=3, warmup=1)
    # every rank holds the same all-reduced output: sum over ranks of a_r @ b_r^T
    mine = (load.a.float() @ load.b.float().t())
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    want = sum(parts).to(torch.bfloat16).float()
    got = load.c[(load.i - 1) & 1].float()
    out_ok = bool(torch.allclose(got, want, rtol=2e-2, atol=2e-1))
    t = torch.tensor([dt], dtype=torch.float64)
    ts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    # 3. the xGMI fidelity phase: collectives on every rank, telemetry (absent here) on rank 0 only
    from k8s_llm_monitor_amd.validation import phases
    xf = phases.xgmi_fidelity(torch, dist, dev, rank, world, nbytes=1 << 16, iters=3, settle_s=0.0)
    if rank == 0:
        out["xgmi"] = {k: xf.get(k) for k in ("world", "expected_bytes_per_gpu", "error")}
    else:
        assert xf == {}
    if rank == 0:
        out["steps_s"] = dt
        out["same_time_on_all_ranks"] = all(float(x) == float(ts[0]) for x in ts)
        out["allreduce_ok"] = out_ok
        out["i"] = load.i
        print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
