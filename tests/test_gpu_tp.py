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
        # a call of another size on one rank is seen by every rank, and costs nothing afterwards
        assert r["mismatch_status"] & 4 and "different size or mode" in r["mismatch_error"] and r["after_mismatch_ok"], r
        if r["rank"] != 0:
            # three calls against an absent peer: one spin limit (1.5 s) plus two short ones, not three limits
            assert r["absent_peer_status"] & 1 and "did not reach" in r["absent_peer_error"] and 1.4 < r["absent_peer_seconds"] < 2.6, r
        assert r["after_absent_peer_ok"], r
    print("xgmi collectives:", {k: res[0][k] for k in ("describe", "us_32KiB", "us_2MiB")})


@pytest.mark.timeout(400)
def test_tensor_parallel_model_and_engine_two_ranks_on_one_gpu():
    """Two processes, each holding half of the `small` model, on the one GPU: every projection all-reduce and the
    sharded argmax run on the mesh kernels (also inside the decode hipGraphs), the engine runs in lock-step."""
    res = run_ranks("tp_worker.py", 2, ["--gpu", "--model", "small"], timeout=380)
    for r in res:
        assert r["tp"]["backend"] == "xgmi" and r["tp"]["xgmi"]["world"] == 2 and r["tp"]["self_test"] == "passed"
        assert r["local_shapes"] == [8, 2, 4096, 16000]
        assert r["model_max_abs_err"] < 0.04 * max(1.0, r["model_logit_scale"]), r          # bf16 partial sums
        assert r["sharded_argmax_agrees"] and r["loop_ok"], r
        assert r["engine_counts"] == res[0]["engine_counts"] and r["graphs"] == res[0]["graphs"] and r["graphs"]
    r0 = [r for r in res if r["rank"] == 0][0]
    assert r0["engine_done"] and r0["engine_tokens"] == 24 and r0["graph_equals_eager"], r0
    print("tp2 on one GPU:", {k: r0[k] for k in ("model_max_abs_err", "model_logit_scale", "engine_tokens_match",
                                                 "engine_first_tokens_match", "engine_counts", "graphs")})


@pytest.mark.timeout(300)
def test_a_launch_that_fails_on_one_rank_only_resets_every_rank():
    """Rank 1's prefill raises, rank 0's does not: rank 0's mesh collectives give up after the spin limit, its step fails
    its check, the control exchange resets both ranks, and the next request is served with the right tokens."""
    res = run_ranks("tp_worker.py", 2, ["--gpu", "--model", "small", "--what", "follower", "--spin-limit", "2.0"], timeout=280)
    r0 = [r for r in res if r["rank"] == 0][0]
    r1 = [r for r in res if r["rank"] == 1][0]
    assert r0["follower_failed_request_error"] and r0["follower_recovered"], r0
    assert r0["follower_resets"] >= 1 and r1["follower_resets"] >= 1 and r1["follower_errors"] == 1, (r0, r1)
    assert r0["follower_seconds_to_fail"] < 30.0, r0              # one spin limit, not one per collective of the step
    print("follower-only failure:", {k: r0[k] for k in ("follower_failed_request_error", "follower_seconds_to_fail", "follower_resets")})


@pytest.mark.timeout(400)
def test_serve_llm_tp_cli_two_ranks_share_the_gpu():
    """`serve-llm --tp 2 --tp-same-device`: launcher, two rank processes, HTTP at rank 0, hipGraph decode with the mesh
    all-reduce inside, clean exit of both ranks on SIGTERM."""
    import signal
    import time
    import urllib.request
    port = free_port()
    p = subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", "serve-llm", "--model", "small", "--tp", "2", "--tp-same-device",
                          "--host", "127.0.0.1", "--port", str(port), "--max-num-seqs", "16", "--max-model-len", "1024",
                          "--kv-blocks", "512"], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    try:
        url = f"http://127.0.0.1:{port}"
        t_end = time.time() + 240
        while True:
            try:
                urllib.request.urlopen(url + "/health", timeout=1).read()
                break
            except OSError:
                assert p.poll() is None and time.time() < t_end, "server did not come up"
                time.sleep(0.5)
        outs = []
        for i in range(3):
            req = urllib.request.Request(url + "/v1/completions", headers={"Content-Type": "application/json"},
                                         data=json.dumps({"prompt_len": 200 + 50 * i, "max_tokens": 24, "seed": i}).encode())
            outs.append(json.loads(urllib.request.urlopen(req, timeout=120).read()))
        assert all(len(o["choices"][0]["token_ids"]) == 24 for o in outs)
        st = json.loads(urllib.request.urlopen(url + "/stats", timeout=5).read())["engine"]
        assert st["tensor_parallel"]["backend"] == "xgmi" and st["tensor_parallel"]["world"] == 2 and st["graphs"], st
        assert st["tensor_parallel"]["self_test"] == "passed"
        assert st["errors"] == 0 and st["step_counts"]["decode"] >= 69
        text = urllib.request.urlopen(url + "/metrics", timeout=5).read().decode()
        assert "vllm:generation_tokens_total" in text
        p.send_signal(signal.SIGTERM)
        assert p.wait(timeout=120) == 0
    finally:
        if p.poll() is None:
            p.kill()
