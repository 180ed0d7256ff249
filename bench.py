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

``value`` is the whole-job burner rate in TFLOP/s (sum over ranks, weak scaling), taken at steady
state: the warm-up is ``--warmup`` steps and at least two seconds of steps, so the figure does not
move with ``--steps``. From short phases outside the timed region come the monitor's own figures,
which are BASELINE.json's metric: telemetry error (HBM used vs. the allocator's ground truth, GFX
activity vs. the known duty cycle, power gauge vs. energy counter), sampling overhead (burner rate
with the whole product on vs. off), alert latency (threshold crossing -> alert visible, per delivery
path) and the power-cap / throttle alerts firing on real hardware.

Output contract: the LAST line of stdout is ONE compact JSON object (< 4 KB): the required keys, the
BASELINE.json metric as flat scalars, ``gemm_variant``, ``rccl_ranks_seen`` and the peak bus bandwidth
per n. Everything else (per-phase rows, the full curves, llm_serving, vendor_comparison) goes to
``bench_detail.json`` (its path is in the line as ``detail``) and, as progress, to stderr.
``--time-budget S`` (default 540) sheds sections and validation phases, longest first, rather than
overrunning; what was shed is listed in the line (``shed``). A section that was started and never returns
(a wedged collective) is ended by a watchdog ``--watchdog-grace`` seconds after the budget: rank 0 prints the
line from the timed number it already holds, with ``watchdog`` naming the section.

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
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
T_PROCESS_START = time.time()


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
    ap.add_argument("--time-budget", type=float, default=540.0,
                    help="wall-clock budget of the whole run in s (0 = unlimited): sections outside the timed region and "
                         "validation phases that do not fit are shed, longest first, instead of overrunning")
    ap.add_argument("--min-warmup-seconds", type=float, default=2.0,
                    help="warm up for at least this long (on top of --warmup steps) so that `value` is a steady-state figure")
    ap.add_argument("--watchdog-grace", type=float, default=40.0,
                    help="with a time budget: this long after the budget has run out every rank ends itself, rank 0 printing the "
                         "result line from the timed number it already holds (a section that never returns must not cost the record)")
    ap.add_argument("--detail-out", default="", help="where the full result goes (default: bench_detail.json next to bench.py)")
    ap.add_argument("--rehearsal-phases", default="",
                    help="--cpu-rehearsal only: stand-in validation phases 'name:priced_s:actual_s,...' (tests of the time budget)")
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
    env = dict(os.environ, KLM_BENCH_LAUNCHER="self", KLM_BENCH_T0=str(time.time()))
    t_start = time.monotonic()
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
    from k8s_llm_monitor_amd.validation import benchlib
    res["launcher"] = "bench.py self-launch (child torch.distributed.run)"
    spent = time.monotonic() - t_start
    left = None if not a.time_budget else a.time_budget - spent
    if not (a.cpu_rehearsal or a.no_probe or a.no_native_probe) and (left is None or left > 60.0):
        # the same curve from the other executor, with no rank process left on the devices
        native = None
        try:
            from k8s_llm_monitor_amd.parallel import probe
            counts = [n for n in (1, 2, 4, 8) if n <= a.gpus]
            per = 240.0 if left is None else max(20.0, (left - 15.0) / len(counts))
            doc = probe.run_sweep(counts, available=a.gpus, max_bytes=a.probe_max_bytes, warmup=3, iters=10, timeout=min(240.0, per))
            native = {str(c["n_gpus"]): probe.brief(c) for c in doc["curves"] if c.get("executor") == "native"}
        except Exception as e:
            native = {"error": f"{type(e).__name__}: {e}"}
        # the full curves join the detail file; the line gets the peak per n
        detail = res.get("detail")
        full = None
        if detail:
            try:
                with open(detail, "r", encoding="utf-8") as f:
                    full = json.load(f)
            except (OSError, ValueError):
                full = None
        if full is not None:
            full["rccl_curves_native"] = native
            full["launcher"] = res["launcher"]
            benchlib.write_detail(full, detail)
        res.update({k: v for k, v in benchlib.compact({"rccl_curves_native": native}).items() if k.startswith(("peak_busbw", "rccl_curves_native"))})
    elif not (a.cpu_rehearsal or a.no_probe or a.no_native_probe):
        res.setdefault("shed", []).append("native_probe")
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

    from k8s_llm_monitor_amd.validation import benchlib, phases
    # the budget's clock starts where the run started: at this process's start, or at the self-launching parent's
    spent = max(0.0, time.time() - float(os.environ.get("KLM_BENCH_T0", T_PROCESS_START)))
    a.budget = phases.Budget(a.time_budget if a.time_budget > 0 else None, t0=time.monotonic() - spent)
    if os.environ.get("KLM_BENCH_LAUNCHER") == "self" and a.budget.seconds is not None and not (a.no_probe or a.no_native_probe or a.cpu_rehearsal):
        a.budget.reserve_s += 25.0 * len([n for n in (1, 2, 4, 8) if n <= a.gpus])      # the parent's native probe comes after the ranks
    a.section, a.provisional, dog = "start", None, None
    if a.budget.seconds is not None and a.watchdog_grace > 0:
        dog = benchlib.Watchdog(a, rank, a.budget.seconds + a.watchdog_grace, a.budget.t0).start()
    result = benchlib.run(a, rank=rank, world=world, dev=dev, ops=ops)
    if dog is not None:
        dog.cancel()
    if rank == 0:
        result.setdefault("launcher", "torch.distributed.run" if env_world is not None and
                          os.environ.get("KLM_BENCH_LAUNCHER") != "self" else "single process")
        path = benchlib.detail_path(a)
        try:
            benchlib.write_detail(result, path)
        except OSError as e:
            print(f"[bench] could not write {path}: {e}", file=sys.stderr)
            path = ""
        # the detail for a reader of the log; the contract's line is the last thing on stdout
        print("[bench] detail: " + json.dumps({k: result[k] for k in ("validation", "wall_seconds", "shed", "rccl_curves") if k in result})[:6000],
              file=sys.stderr, flush=True)
        print(json.dumps(benchlib.compact(result, path)), flush=True)
    if distributed:
        try:
            dist.barrier(group=a.ctrl_group)
            dist.destroy_process_group()
        except RuntimeError as e:
            # a peer that is gone (its own failure is what the launcher reports) must not turn this rank's finished work into one
            print(f"[bench] rank {rank}: closing barrier: {str(e).splitlines()[0][:200]}", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
