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

Launch: ``python bench.py --gpus N`` starts its own N ranks (one process per GPU under
``torch.distributed.run``, as a CHILD of a parent that never touches the GPU), or launch the ranks
yourself with ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``.
Either way one command yields the RCCL all-reduce curve at EVERY n in {1, 2, 4, 8} with n <= N
(``rccl_curves``): the first n ranks form their own communicator and sweep it. A WORLD_SIZE that
disagrees with ``--gpus`` is an error, never a silent one-rank run.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--no-native-probe", action="store_true",
                    help="self-launched runs only: skip the C++ probe (klm_rccl_probe, one process driving n devices) "
                         "that the GPU-free parent runs for every n after the ranks have exited")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="self-launched runs: limit for the ranks, s")
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _result_line(text: str):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def launch_ranks(a, argv) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks as a child and relay their one
    JSON line. This parent never imports torch.cuda and never replaces itself, so the GPUs are touched
    by the ranks only; when they are gone it drives the native probe for n = 1, 2, 4, 8 <= N."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    print(f"[bench] launching {a.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    env = dict(os.environ, KLM_BENCH_LAUNCHER="self")
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, timeout=a.launch_timeout)
    except subprocess.TimeoutExpired:
        print(f"[bench] ranks did not finish within {a.launch_timeout:.0f} s", file=sys.stderr)
        return 124
    res = _result_line(r.stdout or "")
    if r.returncode != 0 or res is None:
        sys.stderr.write((r.stdout or "")[-2000:])
        print(f"[bench] ranks failed: rc={r.returncode}, result line {'missing' if res is None else 'present'}", file=sys.stderr)
        return r.returncode or 1
    res["launcher"] = "bench.py self-launch (child torch.distributed.run)"
    if not (a.cpu_rehearsal or a.no_probe or a.no_native_probe):
        # the same curve from the other executor, with no rank process left on the devices
        try:
            from k8s_llm_monitor_amd.parallel import probe
            counts = [n for n in (1, 2, 4, 8) if n <= a.gpus]
            doc = probe.run_sweep(counts, available=a.gpus, max_bytes=a.probe_max_bytes, warmup=3, iters=10, timeout=240.0)
            res["rccl_curves_native"] = {str(c["n_gpus"]): probe.brief(c) for c in doc["curves"] if c.get("executor") == "native"}
        except Exception as e:
            res["rccl_curves_native"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(res), flush=True)
    return 0


def main():
    a = parse_args()
    # RCCL and tensor sharing between processes need dmabuf IPC on this driver (without it: hipIpcGetMemHandle: invalid
    # argument). The variable is exported on the build and GPU boxes; set it for the ranks wherever it is not.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        raise SystemExit(launch_ranks(a, sys.argv[1:]))
    if env_world is not None and int(env_world) != a.gpus:
        # a flat scaling curve recorded under the wrong rank count is worse than no record
        print(f"[bench] error: --gpus {a.gpus} but the launcher started WORLD_SIZE={env_world} ranks", file=sys.stderr)
        raise SystemExit(2)
    import torch
    import torch.distributed as dist
    from k8s_llm_monitor_amd import ops

    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not a.cpu_rehearsal and torch.cuda.device_count() < world:
        print(f"[bench] error: {world} ranks but {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        raise SystemExit(2)
    if a.cpu_rehearsal:
        dev = torch.device("cpu")
        a.no_sampler = a.no_validation = True
        a.probe_max_bytes = min(a.probe_max_bytes, 1 << 16)      # the sweep logic, not a bandwidth
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    distributed = world > 1 or a.force_dist
    a.ctrl_group = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if a.cpu_rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # host-side control group: ranks that sit out a sub-communicator sweep wait here, off the GPU
            a.ctrl_group = dist.new_group(backend="gloo")

    from k8s_llm_monitor_amd.validation import benchlib
    result = benchlib.run(a, rank=rank, world=world, dev=dev, ops=ops)
    if rank == 0:
        result.setdefault("launcher", "torch.distributed.run" if env_world is not None and
                          os.environ.get("KLM_BENCH_LAUNCHER") != "self" else "single process")
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier(group=a.ctrl_group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
