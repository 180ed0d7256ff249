"""Multi-process CPU rehearsal of the distributed paths (gloo, world_size 2): the RCCL probe's
sweep/timing/reporting logic and the bench's step loop. On MI355X the same code runs with
backend "nccl" (= RCCL); 2/4/8-GPU numbers themselves can only come from an 8-GPU node
("RCCL probe")."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun(nproc, args, timeout=180):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), *args]
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(240)
def test_probe_and_bench_loop_on_two_ranks():
    r = torchrun(2, [os.path.join(ROOT, "tests", "dist_worker.py")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert out["curve"]["n_gpus"] == 2 and out["curve"]["verified"] and out["curve"]["executor"] == "torch.distributed/gloo"
    assert out["checks"] == ["ok", "ok", "ok"] and out["bytes"] == [1024, 4096, 65536]
    assert out["curve"]["peak_busbw_GBps"] > 0 and out["allgather"] and out["reducescatter"] and out["alltoall"]
    assert out["allreduce_ok"] and out["same_time_on_all_ranks"] and out["i"] == 4 and out["steps_s"] > 0


@pytest.mark.timeout(240)
def test_dist_probe_cli_writes_dashboard_results(tmp_path):
    out = tmp_path / "probe.json"
    r = torchrun(2, ["-m", "k8s_llm_monitor_amd.parallel.dist_probe", "--cpu", "--dtype", "fp32", "--max-bytes", "16384",
                     "--iters", "2", "--warmup", "1", "--out", str(out)])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    doc = json.loads(out.read_text())
    assert doc["curves"][0]["n_gpus"] == 2 and doc["scaling"][0]["ceiling_mesh_GBps"] == 153.0
    assert [p["bytes"] for p in doc["curves"][0]["points"]] == [1024, 4096, 16384]
