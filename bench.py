#!/usr/bin/env python3
"""Headline benchmark: the monitored validation load on N MI355X GPUs of one node.

BASELINE.json names no throughput number (the reference publishes none); its metric is
"telemetry error vs amdsmi/rocprof (%), sampling overhead, alert latency; RCCL busBW
1-8 GPU" on the configs "single MI355X: amdsmi telemetry + HIP MFMA bf16 GEMM burner"
and "8xMI355X: RCCL all-reduce probe over xGMI at 1/2/4/8 GPUs". One bench STEP is one
iteration of exactly that load, on every rank, with the telemetry sampler running:

    1. the three GEMMs of one 8192x8192 linear layer's training step (fwd, dgrad, wgrad;
       6 * 8192^3 FLOP) on the hand-written bf16 MFMA burner kernel, on the compute stream;
    2. for N > 1: the data-parallel gradient all-reduce (RCCL, sum, bf16, 128 MiB of dW) on a
       communication stream, overlapped with the next step's fwd and dgrad (dW double-buffered).

``value`` is the whole-job burner rate in TFLOP/s (sum over ranks, weak scaling); the
JSON line also carries the all-reduce bus bandwidth and, from short phases outside the
timed region, the monitor's own figures: sampling overhead (burner rate with the sampler
on vs off), telemetry error (HBM used vs. the allocator's ground truth, GFX activity vs.
the known duty cycle) and alert latency (HBM threshold crossing -> alert visible).

Launch: ``python bench.py`` (1 GPU) or
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=8192, help="GEMM M=N=K")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--no-validation", action="store_true",
                    help="skip the fidelity / overhead / alert-latency phases")
    ap.add_argument("--validation-phases", default="all",
                    help="comma list of hbm,activity,mem,overhead,latency,llm,vllm,xgmi (xgmi needs a process group)")
    ap.add_argument("--no-probe", action="store_true", help="skip the RCCL all-reduce size sweep (N > 1)")
    ap.add_argument("--probe-max-bytes", type=int, default=1 << 28)
    ap.add_argument("--cpu-rehearsal", action="store_true",
                    help="run the same step loop on CPU with gloo and a torch.matmul stand-in (tests the launch / "
                         "reduction / JSON contract without a GPU; the number it prints is not a benchmark)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path (process group, communication stream, probe, xGMI check) even with "
                         "one rank: a one-rank RCCL communicator rehearses the launch on a single-GPU box")
    ap.add_argument("--sample-period", type=float, default=0.0,
                    help="telemetry period in s during the timed region (0 = config default 5 s)")
    return ap.parse_args()


def main():
    a = parse_args()
    import torch
    import torch.distributed as dist
    from k8s_llm_monitor_amd import ops

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"[bench] note: WORLD_SIZE={world} but --gpus {a.gpus}; using WORLD_SIZE", file=sys.stderr)
    if a.cpu_rehearsal:
        dev = torch.device("cpu")
        a.no_sampler = a.no_validation = a.no_probe = True
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    distributed = world > 1 or a.force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if a.cpu_rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from k8s_llm_monitor_amd.validation import benchlib
    result = benchlib.run(a, rank=rank, world=world, dev=dev, ops=ops)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
