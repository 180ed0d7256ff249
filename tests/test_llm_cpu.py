"""The Llama decoder and the serving engine on CPU (tiny model, fp32): the paged, batched,
preempting engine must generate exactly the tokens of a dense full-recompute forward, and its
vLLM-named metrics must be readable by the monitor's scraper (collectors/vllm.py)."""
import json
import math
import urllib.request

import pytest
import torch

from k8s_llm_monitor_amd.collectors.vllm import VllmDirectSource, parse_exposition
from k8s_llm_monitor_amd.models.llama import PRESETS, LlamaConfig, LlamaModel
from k8s_llm_monitor_amd.ops import llm, ref
from k8s_llm_monitor_amd.serving import BlockAllocator, Engine, EngineConfig, exact_quantile
from k8s_llm_monitor_amd.serving.loadclient import OpenLoopLoad
from k8s_llm_monitor_amd.serving.server import LlmServer

TINY = PRESETS["tiny"]


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    return LlamaModel(TINY, "cpu", seed=1, dtype=torch.float32)


def dense_next(model, seq):
    """Full-sequence forward without a cache, written independently of the paged path."""
    cfg = model.cfg
    T = len(seq)
    h = model.embed[torch.tensor(seq)]
    pos = torch.arange(T)
    for L in model.layers:
        x = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + cfg.eps)    # gains are folded into the weights
        qkv = ref.rope(x @ L["wqkv"].t(), model.cos_sin, pos, cfg.n_q, cfg.n_kv).view(T, -1, 128)
        q, k, v = qkv[:, :cfg.n_q], qkv[:, cfg.n_q:cfg.n_q + cfg.n_kv], qkv[:, cfg.n_q + cfg.n_kv:]
        g = cfg.n_q // cfg.n_kv
        k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
        s = torch.einsum("thd,chd->htc", q, k) / math.sqrt(128)
        s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1)[None], float("-inf"))
        a = torch.einsum("htc,chd->thd", torch.softmax(s, -1), v).reshape(T, -1)
        h = h + a @ L["wo"].t()
        x = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + cfg.eps)
        gu = (x @ L["w_gate_up"].t()).view(T, cfg.ffn // 8, 2, 8)       # gate/up interleaved by 8
        h = h + (torch.nn.functional.silu(gu[:, :, 0]) * gu[:, :, 1]).reshape(T, cfg.ffn) @ L["w_down"].t()
    x = h[-1] * torch.rsqrt(h[-1].pow(2).mean(-1, keepdim=True) + cfg.eps)
    return int((x @ model.lm_head.t()).argmax())


def dense_generate(model, prompt, n):
    seq = list(prompt)
    for _ in range(n):
        seq.append(dense_next(model, seq))
    return seq[len(prompt):]


def _engine(model, **kw):
    cfg = EngineConfig(model="tiny", device="cpu", max_num_seqs=kw.pop("max_num_seqs", 8),
                       max_model_len=kw.pop("max_model_len", 256), kv_blocks=kw.pop("kv_blocks", 64),
                       max_num_batched_tokens=kw.pop("max_num_batched_tokens", 512), use_graphs=False)
    return Engine(cfg, model=model)


def _drain(eng, limit=2000):
    n = 0
    while not eng.idle():
        eng.step()
        n += 1
        assert n < limit, "engine did not drain"
    return n


def test_presets_have_the_published_shapes():
    c = PRESETS["llama3-8b"]
    assert (c.hidden, c.layers, c.n_q, c.n_kv, c.ffn, c.vocab) == (4096, 32, 32, 8, 14336, 128256)
    assert abs(c.params() / 1e9 - 8.03) < 0.01                      # Llama-3-8B: 8.03 B parameters
    assert c.kv_bytes_per_token() == 131072                          # 128 KiB of KV per token in bf16
    assert abs(PRESETS["llama3-70b"].params() / 1e9 - 70.55) < 0.05
    assert abs(PRESETS["mistral-7b"].params() / 1e9 - 7.25) < 0.01                 # Mistral-7B-v0.3


def test_block_allocator():
    a = BlockAllocator(4)
    x = a.alloc(3)
    assert len(set(x)) == 3 and a.free == 1 and a.usage == 0.75
    with pytest.raises(MemoryError):
        a.alloc(2)
    a.release(x)
    assert a.free == 4 and a.usage == 0.0


def test_single_request_matches_the_dense_forward(model):
    eng = _engine(model)
    prompt = [5, 17, 300, 2, 9, 44, 101]
    r = eng.submit(prompt, 12)
    _drain(eng)
    assert r.done.is_set() and r.error is None
    assert r.out == dense_generate(model, prompt, 12)
    assert eng.alloc.free == eng.alloc.total                         # every block came back


def test_batched_requests_of_mixed_length_match_the_dense_forward(model):
    eng = _engine(model)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, TINY.vocab, (n,), generator=g).tolist() for n in (1, 15, 16, 17, 40, 3)]
    reqs = [eng.submit(p, m) for p, m in zip(prompts, (5, 9, 1, 20, 7, 33))]
    _drain(eng)
    for p, r in zip(prompts, reqs):
        assert r.out == dense_generate(model, p, r.max_tokens), f"request {r.rid}"
    t = eng.metrics.truth()
    assert t["totals"]["generation_tokens"] == sum(len(r.out) for r in reqs)
    assert t["totals"]["prompt_tokens"] == sum(len(p) for p in prompts)
    assert t["totals"]["finished"] == len(reqs)


def test_late_arrivals_join_the_running_batch(model):
    eng = _engine(model)
    a = eng.submit([1, 2, 3, 4, 5], 10)
    for _ in range(4):
        eng.step()
    b = eng.submit([9, 8, 7], 6)                                      # arrives while `a` decodes
    _drain(eng)
    assert a.out == dense_generate(model, [1, 2, 3, 4, 5], 10)
    assert b.out == dense_generate(model, [9, 8, 7], 6)


@pytest.mark.parametrize("mixed", [False, True])
def test_long_prompts_are_prefilled_in_chunks_next_to_decode_steps(model, mixed):
    """A 16-token budget turns 45- and 70-token prompts into 3 and 5 launches whose later chunks attend to
    the earlier ones through the paged cache. A request already decoding gets a token between chunks
    (mixed_batches off) or inside every chunk launch (on: it never waits for a prefill)."""
    eng = _engine(model, max_num_batched_tokens=16, max_model_len=128)
    eng.cfg.mixed_batches = mixed
    g = torch.Generator().manual_seed(9)
    short = [4, 8, 15, 16, 23, 42]
    a = eng.submit(short, 12)
    eng.step()                                                        # `a` is decoding from here on
    longs = [torch.randint(0, TINY.vocab, (n,), generator=g).tolist() for n in (45, 70)]
    rs = [eng.submit(p, 5) for p in longs]
    kinds, gained = [], []
    while not eng.idle():
        before, had, was_running = dict(eng.step_counts), len(a.out), not a.done.is_set()
        eng.step()
        kinds.append("P" if eng.step_counts["prefill"] > before["prefill"] else
                     "M" if eng.step_counts["mixed"] > before["mixed"] else "D")
        if was_running:
            gained.append(len(a.out) - had)
    k = "".join(kinds)
    if mixed:
        assert eng.step_counts["mixed"] >= 3 + 5 - 1 and k.startswith("MMMMMMM")
        assert all(n == 1 for n in gained)                             # a token in EVERY launch, prefill or not
        assert eng.step_counts["mixed"] + eng.step_counts["prefill"] + eng.step_counts["decode"] == len(kinds) + 1
    else:
        assert eng.step_counts["mixed"] == 0
        assert eng.step_counts["prefill"] >= 1 + 3 + 5 - 1             # chunks of different requests share launches
        assert "PP" not in k[:12]                                      # while `a` runs, every chunk is followed by a decode
    assert a.out == dense_generate(model, short, 12)
    for p, r in zip(longs, rs):
        assert r.out == dense_generate(model, p, 5)
    assert eng.alloc.free == eng.alloc.total


def test_preemption_recomputes_and_still_matches(model):
    # 6 blocks = 96 token slots for three requests that grow to 3 x 40 tokens: someone must be preempted
    eng = _engine(model, kv_blocks=6, max_model_len=64)
    prompts = [[3] * 10, [4] * 10, [5] * 10]
    reqs = [eng.submit(p, 30) for p in prompts]
    _drain(eng)
    assert eng.metrics.preemptions >= 1 and sum(r.preempted for r in reqs) >= 1
    for p, r in zip(prompts, reqs):
        assert r.out == dense_generate(model, p, 30)
    assert eng.alloc.free == eng.alloc.total


def test_oversized_and_empty_prompts_are_rejected(model):
    eng = _engine(model, max_model_len=32)
    assert eng.submit([], 4).error
    assert eng.submit([1] * 32, 4).error
    ok = eng.submit([1] * 31, 100)
    assert ok.error is None and ok.max_tokens == 1                    # clipped to the model length
    _drain(eng)
    assert len(ok.out) == 1 and eng.metrics.rejected == 2


def test_metrics_text_is_what_the_monitor_scrapes(model):
    t = [100.0]
    eng = Engine(EngineConfig(model="tiny", device="cpu", max_num_seqs=4, max_model_len=128, kv_blocks=32,
                              use_graphs=False), model=model, clock=lambda: t[0])
    src = VllmDirectSource(["fake://engine/metrics"], fetch=lambda _u: eng.metrics.render(), clock=lambda: t[0])
    s0 = src.sample()
    assert s0["available"] and s0["running"] == 0 and s0["models"] == ["tiny"]
    reqs = [eng.submit([7] * 20, 40) for _ in range(3)]
    for _ in range(11):                                                # 1 prefill + 10 decode steps, 0.1 s each
        t[0] += 0.1
        eng.step()
    s1 = src.sample()
    assert s1["running"] == 3 and s1["waiting"] == 0
    assert s1["kvCachePercent"] == pytest.approx(eng.alloc.usage * 100)
    # 3 first tokens + 30 decoded tokens in 1.1 s
    assert s1["tokensPerSec"] == pytest.approx(33 / 1.1, rel=1e-6)
    assert s1["promptTokensPerSec"] == pytest.approx(60 / 1.1, rel=1e-6)
    assert 0.08 <= s1["ttftP50"] <= 0.1                                # exact 0.1, bucket (0.08, 0.1]
    truth = eng.metrics.truth()
    assert truth["ttft_p50"] == pytest.approx(0.1) and truth["itl_p50"] == pytest.approx(0.1)
    names = {n for (n, _, _) in parse_exposition(eng.metrics.render())}
    for want in ("vllm:num_requests_running", "vllm:num_requests_waiting", "vllm:kv_cache_usage_perc",
                 "vllm:gpu_cache_usage_perc", "vllm:generation_tokens_total", "vllm:prompt_tokens_total",
                 "vllm:time_to_first_token_seconds_bucket", "vllm:e2e_request_latency_seconds_bucket",
                 "vllm:inter_token_latency_seconds_bucket", "vllm:num_preemptions_total"):
        assert want in names, want
    del reqs


def test_http_server_and_open_loop_load(model):
    eng = _engine(model).start()
    try:
        with LlmServer(eng) as srv:
            body = json.dumps({"prompt": [1, 2, 3], "max_tokens": 5}).encode()
            req = urllib.request.Request(srv.url + "/v1/completions", data=body, headers={"Content-Type": "application/json"})
            out = json.load(urllib.request.urlopen(req, timeout=30))
            assert out["choices"][0]["token_ids"] == dense_generate(model, [1, 2, 3], 5)
            assert out["usage"] == {"prompt_tokens": 3, "completion_tokens": 5, "total_tokens": 8}
            req = urllib.request.Request(srv.url + "/v1/completions", data=b'{"max_tokens": 2}')
            with pytest.raises(urllib.error.HTTPError) as e:
                urllib.request.urlopen(req, timeout=30)
            assert e.value.code == 400
            load = OpenLoopLoad(eng, rate_rps=50, prompt_len=12, output_len=6, seed=1).start(0.4)
            load.join(5)
            assert load.wait_all(60) and len(load.requests) >= 3
            text = urllib.request.urlopen(srv.url + "/metrics", timeout=5).read().decode()
            assert 'vllm:generation_tokens_total{model_name="tiny"}' in text
            st = json.load(urllib.request.urlopen(srv.url + "/stats?window=60", timeout=5))
            assert st["truth"]["totals"]["finished"] == len(load.requests) + 1
            assert st["engine"]["model"] == "tiny"
            assert json.load(urllib.request.urlopen(srv.url + "/v1/models", timeout=5))["data"][0]["id"] == "tiny"
            tr = json.load(urllib.request.urlopen(srv.url + "/trace", timeout=5))
            ev = [e for e in tr["traceEvents"] if e["ph"] == "X"]
            assert {"prefill", "decode"} <= {e["cat"] for e in ev} <= {"prefill", "mixed", "decode"} and all(e["dur"] > 0 for e in ev)
            assert sum(e["args"]["tokens"] for e in ev if e["cat"] == "decode") + sum(1 for e in ev if e["cat"] == "prefill") >= 5
    finally:
        eng.stop()


def test_stop_fails_the_requests_it_leaves_behind(model):
    eng = _engine(model)
    a = eng.submit([1, 2, 3], 50)
    eng.step()                                    # prefill: `a` is running
    b = eng.submit([4, 5], 5)                     # still waiting
    eng.stop()
    assert a.done.is_set() and b.done.is_set()
    assert a.error == "engine stopped" and b.error == "engine stopped"
    assert eng.idle() and eng.alloc.free == eng.alloc.total


def test_fp8_decode_weights_generate_what_the_dequantised_model_generates():
    m = LlamaModel(TINY, "cpu", seed=5, dtype=torch.float32).enable_fp8_decode()
    assert m.fp8_decode and m.layers[0]["wqkv8"].dtype == torch.float8_e4m3fn
    w8 = llm.untile_weights(m.layers[0]["wqkv8"])
    assert torch.equal(m.layers[0]["wqkv"], w8.float() * m.layers[0]["wqkv_s"][:, None])   # bf16 path = dequantised fp8
    assert m.decode_stream_bytes() * 2 == m.weight_bytes() - TINY.vocab * TINY.hidden * 2
    eng = _engine(m)
    prompts = [[3, 1, 4, 1, 5, 9, 2, 6], [2, 7, 1, 8]]
    reqs = [eng.submit(p, 10) for p in prompts]
    _drain(eng)
    for p, r in zip(prompts, reqs):
        assert r.out == dense_generate(m, p, 10)      # prefill on the dequantised weights, decode on fp8 + scales
    assert "fp8" in eng.describe()["decode_weights"]


def test_a_failing_model_launch_fails_the_requests_instead_of_hanging_them(model):
    eng = _engine(model)
    boom = {"n": 0}
    real = eng._prefill

    def broken(batch):
        boom["n"] += 1
        if boom["n"] == 1:
            raise RuntimeError("HIP out of memory (simulated)")
        return real(batch)

    eng._prefill = broken
    eng.start()
    try:
        a = eng.submit([1, 2, 3], 4)
        assert a.done.wait(10) and "out of memory" in a.error
        b = eng.submit([1, 2, 3], 4)                     # the engine keeps serving afterwards
        assert b.done.wait(30) and b.error is None and len(b.out) == 4
    finally:
        eng.stop()
    d = eng.describe()
    assert d["errors"] == 1 and "simulated" in d["last_error"] and eng.alloc.free == eng.alloc.total


def test_fp8_kv_cache_serves_and_halves_the_cache_bytes(model):
    a = Engine(EngineConfig(model="tiny", device="cpu", max_num_seqs=4, max_model_len=128, kv_blocks=32, use_graphs=False),
               model=model)
    b = Engine(EngineConfig(model="tiny", device="cpu", max_num_seqs=4, max_model_len=128, kv_blocks=32, use_graphs=False,
                            kv_cache_dtype="fp8"), model=model)
    assert b.cache.k.dtype == torch.float8_e4m3fn and b.cache.nbytes * (4 if model.dtype == torch.float32 else 2) == a.cache.nbytes
    prompts = [[5, 6, 7, 8, 9], [11] * 20]
    ra = [a.submit(p, 8) for p in prompts]
    rb = [b.submit(p, 8) for p in prompts]
    _drain(a)
    _drain(b)
    assert all(len(r.out) == 8 and r.error is None for r in rb) and b.alloc.free == b.alloc.total
    # the first token comes from the prefill of the same prompt: identical up to the quantised context
    same = sum(x.out[0] == y.out[0] for x, y in zip(ra, rb))
    assert same >= 1
    with pytest.raises(ValueError):
        Engine(EngineConfig(model="tiny", device="cpu", kv_cache_dtype="int4", kv_blocks=8), model=model)


def test_exact_quantile():
    assert exact_quantile([], 0.5) is None
    assert exact_quantile([1.0, 2.0, 3.0, 4.0], 0.5) == 2.5
    assert exact_quantile([5.0], 0.99) == 5.0


def test_cpu_dispatch_of_the_ops_matches_the_reference():
    torch.manual_seed(0)
    x, w, res = torch.randn(5, 256), torch.randn(256), torch.randn(5, 256)
    r0 = res.clone()
    o = llm.rmsnorm(x, w, 1e-5, residual=res)
    want, nr = ref.rmsnorm(x, w, 1e-5, residual=r0)
    assert torch.equal(o, want) and torch.equal(res, nr)
    assert llm.prefill_tiles([5, 40], 4, 1).tolist() == [[1, 32], [1, 16], [1, 0], [0, 0]]     # longest context first
    assert llm.prefill_tiles([5, 40], 4, 2).tolist() == [[1, 32], [1, 0], [0, 0]]
    assert llm.prefill_tiles([8, 8], 4, 1, ctx_lens=[8, 300]).tolist() == [[1, 0], [0, 0]]
    assert llm.pick_splits(1, 8, 8192) == 32 and llm.pick_splits(64, 8, 8192) == 1 and llm.pick_splits(1, 8, 256) == 2
    assert llm.pick_q_tiles([2048], 4, 8) == 1 and llm.pick_q_tiles([4096], 4, 8) == 2


@pytest.mark.parametrize("mixed,blocks", [(False, 10), (True, 10), (True, 6)])
def test_random_traffic_with_a_tight_cache_keeps_every_invariant(model, mixed, blocks):
    """Random arrivals and lengths against 10 blocks: whatever the interleaving of prefill, decode
    and preemption, no block is held twice or leaked, every request finishes with exactly the
    tokens of the dense forward, and the counters add up."""
    import random
    rng = random.Random(7)
    eng = _engine(model, kv_blocks=blocks, max_model_len=96, max_num_seqs=5, max_num_batched_tokens=24)   # prompts get chunked
    eng.cfg.mixed_batches = mixed
    reqs, steps = [], 0
    while len(reqs) < 14 or not eng.idle():
        if len(reqs) < 14 and rng.random() < 0.4:
            p = [rng.randrange(TINY.vocab) for _ in range(rng.randint(1, 40))]
            reqs.append((p, eng.submit(p, rng.randint(1, 30))))
        eng.step()
        steps += 1
        assert steps < 5000
        held = [b for r in eng.running + eng.prefilling for b in r.blocks]
        assert len(held) == len(set(held)), "a block is owned twice"
        assert len(held) + eng.alloc.free == eng.alloc.total, "blocks leaked"
        assert all(not r.blocks and r.prefilled == 0 for r in eng.waiting)
        assert len(eng.running) + len(eng.prefilling) <= 5
    if (mixed, blocks) != (True, 10):          # mixed batches drain this traffic before 10 blocks run out
        assert eng.metrics.preemptions > 0, "the scenario is meant to preempt"
    for p, r in reqs:
        assert r.done.is_set() and r.error is None
        assert r.out == dense_generate(model, p, r.max_tokens), f"request {r.rid} (preempted {r.preempted}x)"
    t = eng.metrics.truth()["totals"]
    assert t["finished"] == 14 and t["generation_tokens"] >= sum(len(r.out) for _, r in reqs)
    assert eng.alloc.free == eng.alloc.total


def test_mid_batch_stack_computes_what_the_library_stack_computes():
    """models/llama.py `_stack_mid` (fused residual + norm / SwiGLU epilogues) against `_stack`, on the CPU fallbacks."""
    from k8s_llm_monitor_amd.models.llama import PRESETS, LlamaModel
    m = LlamaModel(PRESETS["tiny"], "cpu", seed=2, dtype=torch.float32)
    T = 24
    toks = torch.randint(0, m.cfg.vocab, (T,))
    pos = torch.arange(T, dtype=torch.int32)
    slots = torch.arange(T, dtype=torch.int32)

    def attend(qkv, kc, vc):                           # any function of qkv will do: the stacks must agree around it
        return qkv[:, :m.cfg.q_dim].contiguous()
    a = m._stack(toks, pos, slots, m.new_cache(4), attend)
    b = m._stack_mid(toks, pos, slots, m.new_cache(4), attend)
    assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)


def test_decode_buckets_are_powers_of_two_their_midpoints_and_quarter_points_from_128():
    from k8s_llm_monitor_amd.serving.engine import _buckets
    assert _buckets(256) == [1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256]
    assert _buckets(200) == [1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 200] and _buckets(512)[-5:] == [256, 320, 384, 448, 512]
    assert _buckets(8) == [1, 2, 4, 8] and _buckets(1) == [1] and _buckets(40) == [1, 2, 4, 8, 16, 24, 32, 40]
    for lim in (1, 3, 16, 17, 100, 256):
        b = _buckets(lim)
        assert b == sorted(set(b)) and b[-1] == lim                       # every batch up to the limit has a bucket


def test_kernel_path_policy_per_model_class(model):
    """mid_spans / lib_projections are tables of measurements (docs/KERNELS.md): pin them, by model width."""
    from dataclasses import replace
    m = LlamaModel(TINY, "cpu", seed=1, dtype=torch.float32)
    assert m.mid_spans() == [(1, 128), (129, 256)]                              # 8B class
    assert m.lib_projections(192) == {"w_gate_up"} and m.lib_projections(230) == {"w_gate_up", "wo"}
    m.cfg = replace(m.cfg, hidden=8192)                                         # 70B class
    assert m.mid_spans() == [(33, 128), (129, 256)] and m.lib_projections(192) == {"w_gate_up"} and m.lib_projections(256) == frozenset()
    m.cfg = replace(m.cfg, hidden=16384)                                        # a 405B / tp8 shard
    assert m.mid_spans() == [(97, 128), (129, 256)] and m.lib_projections(256) == {"wo"}
    m.fp8_decode = True
    assert m.mid_spans() == [(1, 128), (129, 256)]                              # e4m3 copies up to 128 rows, bf16 tiles above
    m.wide = False
    assert m.mid_spans() == [(1, 128)]
    m.mid_limit = 40
    assert m.mid_spans() == [(1, 40)]
    assert not m.mid(20)                                                        # and never on the CPU
    eng = _engine(model)
    d = eng.describe()["kernel_paths"]
    assert d["skinny_rows"] == 16 and d["mid_spans"] == [[1, 128], [129, 256]] and d["fused_rope_attention"] is False
