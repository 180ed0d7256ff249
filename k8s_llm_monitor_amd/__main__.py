"""Command line: ``python -m k8s_llm_monitor_amd <command>``.

The reference has one entry point, ``node monitor_server.js`` (README.md:99), with no
flags. ``serve`` is that entry point; the other commands drive the
MI355X validation toolkit.

    serve      run the monitor (HTTP API + dashboard + background sampler)
    demo       the same with every source faked (no GPU, no cluster): BASELINE config 1
    sample     print one telemetry sample as JSON and exit
    build      compile the native targets for gfx950
    burner     run the MFMA GEMM / HBM load generator
    probe      run the RCCL all-reduce probe and store its curve for the dashboard
    validate   run the fidelity / overhead / alert-latency phases on this GPU
    serve-llm  run the vLLM-compatible serving load generator (Llama-3 architecture, random weights)
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import sys
import threading
import time
from typing import Optional, Sequence


def _cmd_serve(a, demo: bool = False) -> int:
    from .config import demo_config, load_config
    from .server.http import MonitorServer
    from .server.sampler import Monitor
    over = {}
    if a.port is not None:
        over.setdefault("server", {})["port"] = a.port
    if a.host:
        over.setdefault("server", {})["host"] = a.host
    if getattr(a, "scenario", None):
        over.setdefault("sources", {})["fake_scenario"] = a.scenario
    if demo:
        if a.port is None:
            over.setdefault("server", {})["port"] = 8888
        cfg = demo_config(**over)
    else:
        cfg = load_config(a.config, overrides=over or None)
    mon = Monitor(cfg).start()
    srv = MonitorServer(mon, cfg)
    print(f"🚀 监控面板已启动: {srv.url}  (gpu={mon.gpu.describe()}, kube={mon.kube.describe()}, "
          f"vllm={mon.vllm.describe()})", flush=True)
    stop = threading.Event()
    for sig in (signal.SIGINT, signal.SIGTERM):
        signal.signal(sig, lambda *_: stop.set())
    srv.start()
    try:
        while not stop.wait(0.5):
            if a.duration and time.time() - mon.started > a.duration:
                break
    finally:
        srv.stop()
        mon.stop()
    return 0


def _cmd_sample(a) -> int:
    from .collectors.gpu import open_gpu_source
    from .collectors.host import HostCollector
    src = open_gpu_source(a.source)
    src.sample()
    time.sleep(a.interval)
    out = {"host": HostCollector().collect(), "gpus": src.sample(), "source": src.describe(), "error": src.last_error}
    src.close()
    print(json.dumps(out, indent=1, default=str))
    return 0


def _cmd_build(a) -> int:
    from . import build
    return build.main((["--force"] if a.force else []) + list(a.targets))


def _cmd_burner(a) -> int:
    from .loadgen import burner
    return burner.main_from_args(a)


def _cmd_probe(a) -> int:
    from .parallel import probe
    doc = probe.run_sweep(gpu_counts=[int(x) for x in a.gpus.split(",")], dtypes=a.dtypes.split(","),
                          path=a.out, max_bytes=a.max_bytes, iters=a.iters, warmup=a.warmup)
    for s in doc["scaling"]:
        print(json.dumps(s))
    print("saved:", a.out or probe.default_results_path())
    return 0


def _cmd_validate(a) -> int:
    from .validation import phases
    return phases.main_from_args(a)


def _cmd_serve_llm_replicas(a) -> int:
    """--gpus N: one engine process per GPU (data parallel) behind the router on --port."""
    from .serving.router import ReplicaSet, Router
    child = ["--model", a.model, "--max-num-seqs", str(a.max_num_seqs), "--max-model-len", str(a.max_model_len),
             "--kv-blocks", str(a.kv_blocks), "--kv-fraction", str(a.kv_fraction), "--decode-weights", a.decode_weights,
             "--kv-cache-dtype", a.kv_cache_dtype, "--load-rps", str(a.load_rps / a.gpus), "--prompt-len", str(a.prompt_len),
             "--output-len", str(a.output_len), "--max-num-batched-tokens", str(a.max_num_batched_tokens)] + \
        (["--no-graphs"] if a.no_graphs else []) + (["--no-mixed-batches"] if a.no_mixed_batches else [])
    kind = "cpu" if a.device == "cpu" else "cuda"
    base = a.replica_base_port or a.port + 1
    rs = ReplicaSet(list(range(a.gpus)), base, child, host="127.0.0.1", device_kind=kind).start()
    stop = threading.Event()
    for sig in (signal.SIGINT, signal.SIGTERM):
        signal.signal(sig, lambda *_: stop.set())
    try:
        if not rs.wait_ready(a.ready_timeout):
            print("a replica did not come up", file=sys.stderr, flush=True)
            return 1
        with Router(rs.urls, host=a.host, port=a.port) as router:
            print(f"router on {router.url} -> {a.gpus} replica(s): {rs.urls}", flush=True)
            t0 = time.time()
            while not stop.wait(1.0):
                if a.duration and time.time() - t0 > a.duration:
                    break
            print(json.dumps({k: v for k, v in router.stats().items() if k != "replicas"}), flush=True)
    finally:
        rs.stop()
    return 0


def _cmd_serve_llm_tp_launch(a, argv: Sequence[str]) -> int:
    """--tp N: start N rank processes of this same command (one per GPU) and wait for them."""
    import socket
    import subprocess
    port = a.port
    while port == a.port:                      # the rendezvous port must not be the one rank 0 will serve HTTP on
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(a.tp):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.tp), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KLM_TP_RANK=str(r), KLM_CHILD_OF=str(os.getpid()), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PYTHONUNBUFFERED="1")
        procs.append(subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    stopping = threading.Event()

    def _stop(*_):
        stopping.set()
        procs[0].terminate()                                         # rank 0 stops the engine, the others follow it out

    for sig in (signal.SIGINT, signal.SIGTERM):
        signal.signal(sig, _stop)
    rc = 0
    try:
        # a rank that dies takes the job with it: the others would otherwise wait for it in a collective forever
        while procs[0].poll() is None:
            dead = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() is not None]
            if stopping.is_set():
                # a requested stop: a follower may be out before rank 0 is (its exit status 0 is the stop working, not a rank
                # lost in mid-job); only a rank that FAILS on the way out is reported
                dead = [d for d in dead if d[1] != 0]
            if dead:
                print(f"[tp] rank {dead[0][0]} exited with {dead[0][1]} while the job was running: stopping all ranks",
                      file=sys.stderr, flush=True)
                rc = dead[0][1] or 1
                break
            time.sleep(0.5)
        else:
            rc = procs[0].returncode
            for p in procs[1:]:
                try:
                    p.wait(timeout=60)
                except subprocess.TimeoutExpired:
                    pass
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def _die_with_parent() -> None:
    """A rank / replica process must not outlive its launcher (a killed launcher would otherwise leave ranks waiting in a
    collective for ever): ask the kernel for SIGKILL when the parent dies. Linux only, best effort."""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, signal.SIGKILL, 0, 0, 0)          # PR_SET_PDEATHSIG
        if os.getppid() != int(os.environ["KLM_CHILD_OF"]):                         # the launcher was gone before the call
            os._exit(1)
    except (OSError, AttributeError, KeyError, ValueError):
        pass


def _tp_group(a):
    """This process is rank KLM_TP_RANK of a --tp job: join the process group, pick the device, build the TP group."""
    import torch
    import torch.distributed as dist
    from .parallel.tp import TPGroup
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if a.device == "cpu":
        dev, backend = torch.device("cpu"), "gloo"
    elif a.tp_same_device or torch.cuda.device_count() < world:
        if not a.tp_same_device:
            raise SystemExit(f"--tp {world} needs {world} GPUs ({torch.cuda.device_count()} visible); "
                             f"--tp-same-device rehearses the ranks on one")
        dev, backend = torch.device("cuda", 0), "gloo"        # RCCL refuses two ranks on one device: mesh kernels only
    else:
        dev, backend = torch.device("cuda", rank), "nccl"
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    return TPGroup(dev, data=a.tp_comm if dev.type == "cuda" else "gloo"), dev


def _cmd_serve_llm(a) -> int:
    if a.gpus > 1:
        return _cmd_serve_llm_replicas(a)
    if a.tp > 1 and "KLM_TP_RANK" not in os.environ:
        return _cmd_serve_llm_tp_launch(a, a.argv)
    if "KLM_CHILD_OF" in os.environ:             # a rank of --tp or a replica of --gpus: never outlive the launcher
        _die_with_parent()
    from .serving import Engine, EngineConfig
    from .serving.loadclient import OpenLoopLoad
    from .serving.server import LlmServer
    tp, device = (None, a.device)
    if a.tp > 1:
        tp, dev = _tp_group(a)
        device = str(dev)
    eng = Engine(EngineConfig(model=a.model, device=device, max_num_seqs=a.max_num_seqs, max_model_len=a.max_model_len,
                              kv_blocks=a.kv_blocks, kv_fraction=a.kv_fraction, use_graphs=not a.no_graphs,
                              decode_weights=a.decode_weights, kv_cache_dtype=a.kv_cache_dtype,
                              max_num_batched_tokens=a.max_num_batched_tokens, mixed_batches=not a.no_mixed_batches), tp=tp)
    eng.warmup()
    if tp is not None and tp.rank != 0:
        for sig in (signal.SIGINT, signal.SIGTERM):
            signal.signal(sig, signal.SIG_IGN)          # rank 0 decides when the job ends
        eng.follow()
        tp.close()
        return 0
    eng.start()
    srv = LlmServer(eng, host=a.host, port=a.port).start()
    print(f"llm engine: {json.dumps(eng.describe())}", flush=True)
    print(f"serving {eng.mcfg.name} on {srv.url}  (/metrics, /v1/completions, /stats)", flush=True)
    load = None
    if a.load_rps > 0:
        load = OpenLoopLoad(eng, a.load_rps, a.prompt_len, a.output_len).start(a.duration or 1e9)
    stop = threading.Event()
    for sig in (signal.SIGINT, signal.SIGTERM):
        signal.signal(sig, lambda *_: stop.set())
    t0 = time.time()
    try:
        while not stop.wait(1.0):
            if a.duration and time.time() - t0 > a.duration:
                break
    finally:
        if load is not None:
            load.stop()
        srv.stop()
        eng.stop()
        if tp is not None:
            tp.close()
        print(json.dumps(eng.metrics.truth()), flush=True)
    return 0


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="k8s_llm_monitor_amd", description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("serve", "demo"):
        p = sub.add_parser(name)
        p.add_argument("--config", default=None, help="JSON/YAML config file (or KLM_CONFIG)")
        p.add_argument("--port", type=int, default=None)
        p.add_argument("--host", default="")
        p.add_argument("--scenario", default="", help="fake scenario name (demo / fake sources)")
        p.add_argument("--duration", type=float, default=0.0, help="exit after N seconds (0 = run until signalled)")
    p = sub.add_parser("sample")
    p.add_argument("--source", default="auto")
    p.add_argument("--interval", type=float, default=1.0)
    p = sub.add_parser("build")
    p.add_argument("--force", action="store_true")
    p.add_argument("targets", nargs="*")
    p = sub.add_parser("burner")
    p.add_argument("--kind", default="gemm_bf16", choices=["gemm_bf16", "gemm_fp8", "gemm_fp4", "hbm_copy", "hbm_read", "hbm_fill", "spin"])
    p.add_argument("--duty", type=float, default=100.0, help="busy share in percent")
    p.add_argument("--period-ms", type=float, default=100.0, help="duty-cycle period")
    p.add_argument("--seconds", type=float, default=10.0)
    p.add_argument("--size", type=int, default=8192)
    p.add_argument("--hbm-gib", type=float, default=0.0, help="also hold this much HBM (fill injector)")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--json", action="store_true")
    p.add_argument("--monitored", action="store_true", help="run the whole monitor (sampler, server, SSE + polling clients) beside the load")
    p.add_argument("--monitor-period", type=float, default=1.0)
    p = sub.add_parser("probe")
    p.add_argument("--gpus", default="1,2,4,8")
    p.add_argument("--dtypes", default="bf16")
    p.add_argument("--max-bytes", type=int, default=1 << 30)
    p.add_argument("--iters", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--out", default="")
    p = sub.add_parser("validate")
    p.add_argument("--out", default="")
    p.add_argument("--quick", action="store_true")
    p.add_argument("--phases", default="all", help="all or a comma list of hbm,activity,mem,overhead,latency,llm,vllm")
    p.add_argument("--device", type=int, default=0)
    p = sub.add_parser("serve-llm")
    p.add_argument("--model", default="llama3-8b", help="llama3-8b, llama3-70b, llama3-405b (needs --tp), mistral-7b, wide, small, tiny")
    p.add_argument("--device", default="cuda")
    p.add_argument("--host", default="127.0.0.1", help="bind address (0.0.0.0 to serve other hosts: the API is unauthenticated)")
    p.add_argument("--port", type=int, default=8000)
    p.add_argument("--max-num-seqs", type=int, default=256)
    p.add_argument("--max-model-len", type=int, default=4096)
    p.add_argument("--kv-blocks", type=int, default=0, help="KV blocks of 16 tokens (0: from --kv-fraction)")
    p.add_argument("--kv-fraction", type=float, default=0.5, help="share of the free HBM given to the KV cache")
    p.add_argument("--no-graphs", action="store_true")
    p.add_argument("--max-num-batched-tokens", type=int, default=8192, help="token budget of one prefill launch (longer prompts go in chunks)")
    p.add_argument("--no-mixed-batches", action="store_true",
                   help="alternate prefill and decode launches instead of decoding the running sequences inside prefill launches")
    p.add_argument("--gpus", type=int, default=1, help="N > 1: one engine process per GPU behind a router on --port")
    p.add_argument("--tp", type=int, default=1,
                   help="tensor parallel over N GPUs (2, 4, 8): one process per GPU, all-reduce on the xGMI mesh kernels; "
                        "for models beyond one GPU's 288 GB (llama3-405b)")
    p.add_argument("--tp-comm", default="auto", choices=["auto", "xgmi", "rccl"],
                   help="auto: mesh kernels for small (decode) messages, RCCL for large (prefill)")
    p.add_argument("--tp-same-device", action="store_true", help="rehearsal: every rank on GPU 0 (mesh kernels only)")
    p.add_argument("--replica-base-port", type=int, default=0, help="first replica port (default: --port + 1)")
    p.add_argument("--ready-timeout", type=float, default=300.0)
    p.add_argument("--decode-weights", default="bf16", choices=["bf16", "fp8"],
                   help="fp8: W8A16 weight copies for decode up to batch 128 (half the bytes per token; costs half the model size in HBM)")
    p.add_argument("--kv-cache-dtype", default="bf16", choices=["bf16", "fp8"],
                   help="fp8: e4m3 KV cache, twice the tokens per GB and half the attention bytes")
    p.add_argument("--load-rps", type=float, default=0.0, help="built-in open-loop load (requests per second)")
    p.add_argument("--prompt-len", type=int, default=512)
    p.add_argument("--output-len", type=int, default=128)
    p.add_argument("--duration", type=float, default=0.0)
    a = ap.parse_args(argv)
    a.argv = list(argv) if argv is not None else sys.argv[1:]      # --tp re-runs this command once per rank
    if a.cmd == "serve":
        return _cmd_serve(a)
    if a.cmd == "demo":
        return _cmd_serve(a, demo=True)
    return {"sample": _cmd_sample, "build": _cmd_build, "burner": _cmd_burner, "probe": _cmd_probe,
            "validate": _cmd_validate,
            "serve-llm": _cmd_serve_llm}[a.cmd](a)


if __name__ == "__main__":
    sys.exit(main())
