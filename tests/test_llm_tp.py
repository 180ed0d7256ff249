"""Tensor parallelism of the serving model and engine on CPU: two gloo ranks, fp32, against the unsharded model.
The GPU version (xGMI mesh kernels, hipGraph decode) is tests/test_gpu_tp.py."""
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


RENDEZVOUS_ERRORS = ("Address already in use", "errno: 98", "Connection refused", "Connection reset by peer", "Broken pipe")


def start_tp_server(attempts: int = 2):
    """`serve-llm --tp 2` on CPU, up and answering /health. The ports (HTTP here, the ranks' rendezvous inside the launcher) are
    picked free and bound a moment later, so a start that dies on a port somebody else took in between is started once more
    (nothing else is retried: a server that stays alive and never answers fails the test)."""
    import time
    import urllib.request
    err = ""
    for _ in range(attempts):
        port = free_port()
        p = subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", "serve-llm", "--model", "tiny", "--device", "cpu", "--tp", "2",
                              "--host", "127.0.0.1", "--port", str(port), "--max-num-seqs", "4", "--max-model-len", "128", "--kv-blocks", "32"],
                             cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2"), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True)
        url = f"http://127.0.0.1:{port}"
        t_end = time.time() + 120
        while p.poll() is None:
            try:
                urllib.request.urlopen(url + "/health", timeout=1).read()
                return p, url
            except OSError:
                if time.time() > t_end:
                    p.kill()
                    raise AssertionError("server did not come up")
                time.sleep(0.3)
        err = (p.stderr.read() or "")[-1500:]
        if not any(e in err for e in RENDEZVOUS_ERRORS):
            break
    raise AssertionError("server exited during start-up: " + err)


def run_ranks(world, args=(), timeout=240, attempts=2):
    """Runs tests/tp_worker.py as `world` processes. The rendezvous port is found by binding and releasing it, which another
    process can win in between: such a start-up failure (and only that) gets a second attempt on a new port."""
    for attempt in range(attempts):
        port = free_port()
        procs = [subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "tp_worker.py"), *args], cwd=ROOT,
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                           PYTHONPATH=ROOT, OMP_NUM_THREADS="2"),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
        res, errs = [], []
        try:
            for p in procs:
                so, se = p.communicate(timeout=timeout)
                if p.returncode != 0:
                    errs.append(so[-1500:] + se[-3000:])
                    continue
                res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        if not errs:
            return res
        if attempt + 1 < attempts and any(m in e for e in errs for m in RENDEZVOUS_ERRORS) and not any("RESULT " in e for e in errs):
            continue
        raise AssertionError("\n".join(errs))


def test_config_shards():
    from k8s_llm_monitor_amd.models.llama import PRESETS
    big = PRESETS["llama3-405b"]
    assert 400e9 < big.params() < 410e9
    l = big.shard(8)
    assert (l.n_q, l.n_kv, l.ffn, l.vocab, l.hidden) == (16, 1, 6656, 16032, 16384) and l.ffn % 128 == 0 and l.vocab % 16 == 0
    assert big.shard(1) is big
    with pytest.raises(ValueError):
        PRESETS["tiny"].shard(4)                     # 2 kv heads do not split over 4 ranks


@pytest.mark.timeout(300)
def test_two_rank_model_and_engine_match_the_unsharded_model():
    res = run_ranks(2)
    for r in res:
        assert r["tp"]["world"] == 2 and r["tp"]["backend"] == "gloo"
        assert r["local_shapes"] == [2, 1, 256, 256]
        assert r["model_max_abs_err"] < 2e-4 * max(1.0, r["model_logit_scale"]), r          # fp32: only the sum order differs
        assert r["sharded_argmax_agrees"] and r["loop_ok"], r
        assert r["engine_steps"] == res[0]["engine_steps"] and r["engine_counts"] == res[0]["engine_counts"]   # lock-step
    r0 = [r for r in res if r["rank"] == 0][0]
    assert r0["engine_tokens_match"] and r0["engine_done"] and r0["engine_tokens"] == 24, r0
    assert "injected launch failure" in r0["failed_request_error"] and all(r["engine_errors"] == 1 for r in res)
    assert r0["engine_counts"]["prefill"] + r0["engine_counts"]["mixed"] >= 3 and r0["engine_counts"]["mixed"] >= 1   # 242 prompt tokens at 96 per launch: chunked, running sequences ride along


@pytest.mark.timeout(300)
def test_ranks_with_unequal_kv_space_agree_on_the_smallest_and_stay_in_step():
    res = run_ranks(2, args=("--what", "unequal"))
    for r in res:
        assert r["unequal_counts"] == [40, 28] and r["unequal_blocks_total"] == 28, r
        assert r["unequal_steps"] == res[0]["unequal_steps"] and r["unequal_step_counts"] == res[0]["unequal_step_counts"]
        assert r["unequal_resets"] == 0 and r["unequal_preemptions"] == res[0]["unequal_preemptions"]
    r0 = [r for r in res if r["rank"] == 0][0]
    assert r0["unequal_tokens_match"] and r0["unequal_done"], r0


class _ScriptedGroup:
    """Stands in for a TPGroup in `_tp_sync`: this rank's four integers (failed, state hash, stop, new requests) go in, the
    scripted rows of all ranks come out; rank 0's request specs travel in a second message only when it announces any."""

    def __init__(self, rank, script, specs=None):
        self.rank, self.world, self.script, self.sent, self.specs, self.broadcasts = rank, 2, script, [], specs, 0

    def gather_ints(self, values):
        self.sent.append(list(values))
        return [list(r) for r in self.script(list(values))]

    def broadcast_object(self, obj=None):
        self.broadcasts += 1
        return obj if self.rank == 0 else self.specs

    def gather_objects(self, obj):
        raise AssertionError("the per-step message is four integers: nothing is pickled (ADVICE r2)")

    def resync(self):
        self.resyncs = getattr(self, "resyncs", 0) + 1


def _tiny_engine():
    import torch
    from k8s_llm_monitor_amd.serving.engine import Engine, EngineConfig
    return Engine(EngineConfig(model="tiny", device="cpu", max_num_seqs=4, max_model_len=128, kv_blocks=16, use_graphs=False))


def test_tp_sync_resets_every_rank_when_any_rank_failed_or_drifted():
    from k8s_llm_monitor_amd.serving.engine import Request
    # a follower: rank 0 reports two new requests, nobody failed
    e = _tiny_engine()
    e.tp = _ScriptedGroup(1, lambda mine: [[0, mine[1], 0, 2], mine], specs=[(0, [1, 2, 3], 4), (1, [4, 5], 2)])
    assert e._tp_sync() is False and [r.rid for r in e.waiting] == [0, 1] and e.tp_resets == 0 and e.tp.broadcasts == 1
    assert e.tp.sent[0][0] == 0 and e.tp.sent[0][2:] == [0, 0] and 0 <= e.tp.sent[0][1] < 2 ** 63      # followers send no control part
    # a step without new requests is ONE small collective: no second message
    e.tp = _ScriptedGroup(1, lambda mine: [[0, mine[1], 0, 0], mine])
    assert e._tp_sync() is False and e.tp.broadcasts == 0 and len(e.tp.sent) == 1
    # the follower failed its last step: it says so, and drops what it holds like everyone else
    e = _tiny_engine()
    e._tp_reset = True
    e.tp = _ScriptedGroup(1, lambda mine: [[0, mine[1], 0, 1], mine], specs=[(2, [7], 1)])
    e._tp_sync()
    assert e.tp.sent[0][0] == 1 and [r.rid for r in e.waiting] == [2] and e.tp_resets == 1 and e._tp_reset is False
    # rank 0 hears that rank 1 failed: its requests in flight and the just-submitted one all end with an error
    e0 = _tiny_engine()
    e0.tp = _ScriptedGroup(0, lambda mine: [mine, [1, mine[1], 0, 0]])
    held = Request(5, [1, 2, 3], 4, 0.0)
    e0.waiting.append(held)
    fresh = Request(6, [9], 2, 0.0)
    e0._inbox.append(fresh)
    e0._tp_sync()
    assert e0.tp.sent[0][3] == 1 and e0.tp.broadcasts == 1                      # it announced the one new request
    assert held.done.is_set() and "rank(s) [1] failed" in held.error
    # ... while a request that arrived with this very message was no part of the failed step and is kept, on every rank
    assert not fresh.done.is_set() and list(e0.waiting) == [fresh] and not e0._inbox and e0.tp_resets == 1
    # equal failure flags but a different scheduler state on rank 1 (another hash): also a reset, before any launch
    e0 = _tiny_engine()
    e0.tp = _ScriptedGroup(0, lambda mine: [mine, [0, mine[1] ^ 1, 0, 0]])
    held = Request(7, [1, 2, 3], 4, 0.0)
    e0.waiting.append(held)
    e0._tp_sync()
    assert "scheduled differently" in held.error and e0.tp_resets == 1 and e0.tp.resyncs == 1
    # the state hash moves with everything scheduling depends on: queue length, block pool
    e1, e2 = _tiny_engine(), _tiny_engine()
    e1.tp = _ScriptedGroup(0, lambda mine: [mine, mine])
    e2.tp = _ScriptedGroup(0, lambda mine: [mine, mine])
    e2.waiting.append(Request(8, [1], 1, 0.0))
    e1._tp_sync(), e2._tp_sync()
    assert e1.tp.sent[0][1] != e2.tp.sent[0][1]
    # the stop wish travels in rank 0's row
    e = _tiny_engine()
    e.tp = _ScriptedGroup(1, lambda mine: [[0, mine[1], 1, 0], mine])
    assert e._tp_sync() is True


@pytest.mark.timeout(300)
def test_serve_llm_tp_cli_two_cpu_ranks():
    """`serve-llm --tp 2` end to end on CPU: the launcher starts both ranks, rank 0 answers HTTP, a completion runs
    through both shards, SIGTERM at the launcher ends rank 0 and, through the control channel, rank 1."""
    import signal
    import time
    import urllib.request
    p, url = start_tp_server()
    try:
        req = urllib.request.Request(url + "/v1/completions", data=json.dumps({"prompt_len": 30, "max_tokens": 5, "seed": 1}).encode(),
                                     headers={"Content-Type": "application/json"})
        out = json.loads(urllib.request.urlopen(req, timeout=60).read())
        assert len(out["choices"][0]["token_ids"]) == 5 and out["usage"]["prompt_tokens"] == 30
        st = json.loads(urllib.request.urlopen(url + "/stats", timeout=5).read())
        assert st["engine"]["tensor_parallel"] == {"world": 2, "rank": 0, "backend": "gloo", "self_test": "not run (no mesh kernels in this group)"}
        p.send_signal(signal.SIGTERM)
        assert p.wait(timeout=90) == 0
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.timeout(300)
def test_tp_launcher_stops_the_job_when_a_rank_dies():
    import signal
    import time
    import urllib.request
    p, _ = start_tp_server()
    try:
        # find rank 1 among the launcher's children and kill exactly that process
        kids = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
        assert len(kids) == 2, kids
        rank1 = None
        for k in kids:
            env = open(f"/proc/{k}/environ", "rb").read().split(b"\0")
            if b"KLM_TP_RANK=1" in env:
                rank1 = int(k)
        assert rank1 is not None
        os.kill(rank1, signal.SIGKILL)
        assert p.wait(timeout=60) != 0
        assert "rank 1 exited" in p.stderr.read()
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.timeout(300)
def test_ranks_do_not_outlive_a_killed_launcher():
    """SIGKILL at the launcher (it cannot clean up): the kernel delivers the ranks' parent-death signal."""
    import signal
    import time
    import urllib.request
    port = free_port()
    p = subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", "serve-llm", "--model", "tiny", "--device", "cpu", "--tp", "2",
                          "--host", "127.0.0.1", "--port", str(port), "--max-num-seqs", "4", "--max-model-len", "128", "--kv-blocks", "32"],
                         cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2"), stdout=subprocess.DEVNULL,
                         stderr=subprocess.DEVNULL)
    kids = []
    try:
        t_end = time.time() + 120
        while True:
            try:
                urllib.request.urlopen(f"http://127.0.0.1:{port}/health", timeout=1).read()
                break
            except OSError:
                assert p.poll() is None and time.time() < t_end, "server did not come up"
                time.sleep(0.3)
        kids = [int(k) for k in subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()]
        assert len(kids) == 2
        p.kill()
        p.wait(timeout=30)
        t_end = time.time() + 30
        while time.time() < t_end and any(os.path.exists(f"/proc/{k}") and "Z" not in open(f"/proc/{k}/stat").read().split(")")[-1].split()[0]
                                          for k in kids):
            time.sleep(0.2)
        alive = [k for k in kids if os.path.exists(f"/proc/{k}") and open(f"/proc/{k}/stat").read().split(")")[-1].split()[0] != "Z"]
        assert not alive, f"ranks {alive} survived their launcher"
    finally:
        if p.poll() is None:
            p.kill()
        for k in kids:
            try:
                os.kill(k, signal.SIGKILL)
            except OSError:
                pass
