"""Tensor-parallel pieces on one MI355X: the xGMI mesh collectives (parallel/csrc/xgmi_allreduce.hip) run between
several PROCESSES that share the one GPU of the test box (hipIpc-mapped buffers, flag barriers: the kernels only see
pointers, so the protocol is the one eight GPUs run)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(script, world, args=(), timeout=240):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", script), *args], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=timeout)
            outs.append((p.returncode, so, se))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = []
    for rc, so, se in outs:
        assert rc == 0, so[-1500:] + se[-3000:]
        res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    return res


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_xgmi_collectives_between_processes(world):
    res = run_ranks("xgmi_worker.py", world)
    assert sorted(r["rank"] for r in res) == list(range(world))
    for r in res:
        assert all(r["sizes"].values()), r["sizes"]                  # bit-exact against the fp32 rank-order sum
        assert r["modes"]["4096"] == [0] and r["modes"]["1048576"] == [1], r["modes"]      # one shot / two shot
        assert r["repeat_mismatches"] == 0 and r["gather_ok"] and r["graph_ok"] and r["status"] == 0, r
    print("xgmi collectives:", {k: res[0][k] for k in ("describe", "us_32KiB", "us_2MiB")})
