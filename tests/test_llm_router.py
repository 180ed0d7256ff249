"""Data-parallel serving: the router over in-process replicas, and `serve-llm --gpus 2` as real
processes (CPU engines, tiny model)."""
import json
import os
import signal
import socket
import subprocess
import sys
import time
import urllib.error
import urllib.request

import pytest
import torch

from k8s_llm_monitor_amd.collectors.vllm import VllmDirectSource, parse_exposition
from k8s_llm_monitor_amd.models.llama import PRESETS, LlamaModel
from k8s_llm_monitor_amd.serving import Engine, EngineConfig
from k8s_llm_monitor_amd.serving.router import Router, label_exposition
from k8s_llm_monitor_amd.serving.server import LlmServer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _post(url, body, timeout=60):
    req = urllib.request.Request(url + "/v1/completions", data=json.dumps(body).encode(),
                                 headers={"Content-Type": "application/json"})
    return json.load(urllib.request.urlopen(req, timeout=timeout))


def test_label_exposition_adds_the_replica_label():
    text = '# HELP x y\nvllm:num_requests_running{model_name="m"} 3.0\nfoo 1\nbar{le="+Inf",a="b"} 7\n'
    assert label_exposition(text, "2") == ['vllm:num_requests_running{model_name="m",replica="2"} 3.0',
                                            'foo{replica="2"} 1', 'bar{le="+Inf",a="b",replica="2"} 7']


def test_router_spreads_requests_and_merges_metrics():
    model = LlamaModel(PRESETS["tiny"], "cpu", seed=1, dtype=torch.float32)
    engines = [Engine(EngineConfig(model="tiny", device="cpu", max_num_seqs=4, max_model_len=128, kv_blocks=32,
                                   use_graphs=False), model=model).start() for _ in range(2)]
    servers = [LlmServer(e).start() for e in engines]
    try:
        with Router([s.url for s in servers]) as router:
            outs = [_post(router.url, {"prompt": [3, 1, 4, 1, 5], "max_tokens": 4}) for _ in range(6)]
            assert {o["replica"] for o in outs} == {0, 1}                       # both replicas serve
            assert len({tuple(o["choices"][0]["token_ids"]) for o in outs}) == 1   # same weights, same answer
            assert router.sent == [3, 3] and router.outstanding == [0, 0]
            text = urllib.request.urlopen(router.url + "/metrics", timeout=5).read().decode()
            samples = parse_exposition(text)
            gen = [(l.get("replica"), v) for (n, l, v) in samples if n == "vllm:generation_tokens_total"]
            assert sorted(gen) == [("0", 12.0), ("1", 12.0)]
            # the monitor's collector reads the router like one endpoint and sums the replicas
            v = VllmDirectSource([router.url]).sample()
            assert v["available"] and v["running"] == 0 and v["models"] == ["tiny"]
            assert json.load(urllib.request.urlopen(router.url + "/health", timeout=5)) == {"replicas_up": 2, "replicas": 2}
            with pytest.raises(urllib.error.HTTPError) as e:                      # a replica's 400 is passed through
                _post(router.url, {"max_tokens": 2})
            assert e.value.code == 400
            # sizes are judged before anything is built from them: a prompt_len of a billion costs nothing
            import time as _t
            t0 = _t.monotonic()
            for bad in ({"prompt_len": 10 ** 9}, {"prompt_len": 5, "max_tokens": -3}, {"prompt_len": 5, "max_tokens": 10 ** 9},
                        {"prompt": list(range(500)), "max_tokens": 2}, [1, 2, 3]):
                with pytest.raises(urllib.error.HTTPError) as e:
                    _post(servers[1].url, bad)
                assert e.value.code == 400, bad
            assert _t.monotonic() - t0 < 2.0
            # the router bounds the body like the replicas do
            import http.client
            c = http.client.HTTPConnection(router.host, router.port, timeout=5)
            c.putrequest("POST", "/v1/completions")
            c.putheader("Content-Length", str(1 << 30))
            c.endheaders()
            assert c.getresponse().status == 400
            c.close()
            # one replica dies: requests keep being answered by the other
            servers[0].stop()
            engines[0].stop()
            outs = [_post(router.url, {"prompt": [9, 9], "max_tokens": 2}) for _ in range(3)]
            assert {o["replica"] for o in outs} == {1} and sum(router.failed) >= 1
    finally:
        for s in servers[1:]:
            s.stop()
        for e in engines:
            e.stop()


def test_router_counts_a_replica_that_answers_garbage_as_failed():
    import http.server
    import threading

    class Garbage(http.server.BaseHTTPRequestHandler):
        def do_POST(self):
            self.rfile.read(int(self.headers.get("Content-Length") or 0))
            body = b"<html>not json</html>"
            self.send_response(200)
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

        def log_message(self, *a):
            pass

    srv = http.server.ThreadingHTTPServer(("127.0.0.1", 0), Garbage)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    try:
        with Router([f"http://127.0.0.1:{srv.server_address[1]}"]) as router:
            with pytest.raises(urllib.error.HTTPError) as e:
                _post(router.url, {"prompt": [1, 2], "max_tokens": 2})
            assert e.value.code == 503 and router.failed == [1] and router.outstanding == [0]
    finally:
        srv.shutdown()
        srv.server_close()


def _free_port_block(n):
    """Base of n consecutive ports that can all be bound right now."""
    for _ in range(50):
        socks = []
        try:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            socks.append(s)
            base = s.getsockname()[1]
            if base + n > 65535:
                continue
            for k in range(1, n):
                t = socket.socket()
                socks.append(t)
                t.bind(("127.0.0.1", base + k))
            return base
        except OSError:
            continue
        finally:
            for s in socks:
                s.close()
    raise RuntimeError("no block of free ports")


def test_serve_llm_gpus_2_runs_two_engine_processes_behind_the_router():
    base = _free_port_block(3)
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.Popen([sys.executable, "-m", "k8s_llm_monitor_amd", "serve-llm", "--gpus", "2", "--device", "cpu",
                          "--model", "tiny", "--host", "127.0.0.1", "--port", str(base + 2), "--replica-base-port", str(base),
                          "--kv-blocks", "64", "--max-model-len", "128", "--max-num-seqs", "4", "--ready-timeout", "120"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    url = f"http://127.0.0.1:{base + 2}"
    try:
        t_end = time.time() + 150
        up = False
        while time.time() < t_end and p.poll() is None:
            try:
                up = json.load(urllib.request.urlopen(url + "/health", timeout=2))["replicas_up"] == 2
                if up:
                    break
            except (urllib.error.URLError, OSError, ValueError):
                time.sleep(0.5)
        assert up, "router / replicas did not come up: " + (p.stdout.read() if p.poll() is not None else "")
        outs = [_post(url, {"prompt_len": 6, "seed": 1, "max_tokens": 3}) for _ in range(4)]
        assert {o["replica"] for o in outs} == {0, 1}
        assert all(len(o["choices"][0]["token_ids"]) == 3 for o in outs)
    finally:
        p.send_signal(signal.SIGTERM)
        try:
            p.wait(timeout=40)
        except subprocess.TimeoutExpired:
            p.kill()
