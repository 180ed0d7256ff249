"""One rank of the xGMI collective tests (tests/test_gpu_tp.py starts WORLD_SIZE of these, all on cuda:0).
Prints one line ``RESULT {json}``."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_llm_monitor_amd.parallel import xgmi  # noqa: E402


def make(r: int, n: int, dev) -> torch.Tensor:
    return (((torch.arange(n, device=dev) % 251).float() - 125.0) * (0.03125 * (r + 1))).to(torch.bfloat16)


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = xgmi.XgmiComm(rank, world, dev, xgmi.gloo_exchange(), slot_bytes=8 << 20, blocks=32, spin_limit_s=1.5)
    out = {"rank": rank, "describe": comm.describe(), "sizes": {}, "modes": {}}

    def expected(n):
        acc = torch.zeros(n, dtype=torch.float32, device=dev)
        for r in range(world):
            acc += make(r, n, dev).float()
        return acc.to(torch.bfloat16)

    # every schedule: one shot, two shot, pieces (10 M elements = 20 MB > the 4 MiB two-shot capacity)
    for n in (8, 4096, 65536, 1 << 20, 10 * (1 << 20) + 64):
        x = make(rank, n, dev)
        comm.all_reduce(x)
        out["sizes"][n] = bool(torch.equal(x, expected(n)))
        out["modes"][n] = sorted({m for (_, _, m) in comm.plan(n * 2)})

    # slot reuse: many back-to-back calls of alternating sizes, every result checked on the device
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    small, big = make(rank, 16384, dev), make(rank, 1 << 19, dev)
    es, eb = expected(16384), expected(1 << 19)
    for i in range(150):
        for src, want in ((small, es), (big, eb)):
            y = src.clone()
            comm.all_reduce(y)
            bad += (y != want).sum()
    out["repeat_mismatches"] = int(bad.item())

    # all-gather of a non-bf16 tensor
    g = comm.all_gather(torch.full((4, 2), float(rank + 1), dtype=torch.float32, device=dev))
    out["gather_ok"] = bool(all(torch.all(g[r] == r + 1) for r in range(world))) and tuple(g.shape) == (world, 4, 2)

    # inside a hipGraph: the launch arguments never change, the epoch lives in device memory
    src = make(rank, 32768, dev)
    buf = torch.empty_like(src)
    gout = torch.empty((world, 32768), dtype=torch.bfloat16, device=dev)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        buf.copy_(src)
        comm.all_reduce(buf)
        comm.all_gather(src, out=gout)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        buf.copy_(src)
        comm.all_reduce(buf)
        comm.all_gather(src, out=gout)
    ok = True
    want = expected(32768)
    for _ in range(40):
        graph.replay()
        ok = ok and bool(torch.equal(buf, want))
    out["graph_ok"] = ok and bool(all(torch.equal(gout[r], make(r, 32768, dev)) for r in range(world)))

    # latency of the decode-sized message and rate of a large one (all ranks share ONE GPU here: not an xGMI number)
    def timed(n, iters):
        y = make(rank, n, dev)
        for _ in range(5):
            comm.all_reduce(y)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            comm.all_reduce(y)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters
    out["us_32KiB"] = round(timed(16384, 200) * 1e6, 2)
    out["us_2MiB"] = round(timed(1 << 20, 50) * 1e6, 2)
    out["status"] = comm.status()

    # ranks whose schedules differ: rank 0 sums 4096 elements where the others sum 8192. Every rank sees a peer's tag
    # differ from its own (status bit 4) instead of quietly summing stale slot bytes; the next matching call is clean.
    dist.barrier()
    y = make(rank, 4096 if rank == 0 else 8192, dev)
    comm.all_reduce(y)
    out["mismatch_status"] = comm.status()
    try:
        comm.check()
        out["mismatch_error"] = None
    except xgmi.XgmiError as e:
        out["mismatch_error"] = str(e)
    dist.barrier()
    y = make(rank, 4096, dev)
    comm.all_reduce(y)
    out["after_mismatch_ok"] = bool(torch.equal(y, expected(4096))) and comm.status() == 0

    # a peer that never launches: ranks != 0 make three calls rank 0 does not make. The first gives up after the spin
    # limit, the next two after a fraction of it (the status word is still set); resync() then puts every rank on the
    # same epoch again and the next call is exact.
    dist.barrier()
    t0 = time.perf_counter()
    if rank != 0:
        for _ in range(3):
            comm.all_reduce(make(rank, 4096, dev))
        torch.cuda.synchronize()
        out["absent_peer_seconds"] = round(time.perf_counter() - t0, 3)
        out["absent_peer_status"] = comm.status()
        try:
            comm.check()
            out["absent_peer_error"] = None
        except xgmi.XgmiError as e:
            out["absent_peer_error"] = str(e)
    dist.barrier()
    comm.resync()                               # rank 0 is three epochs behind: everyone moves to the largest
    y = make(rank, 65536, dev)
    comm.all_reduce(y)
    out["after_absent_peer_ok"] = bool(torch.equal(y, expected(65536))) and comm.status() == 0
    dist.barrier()
    comm.close()
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
