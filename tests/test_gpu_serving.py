"""The Llama decoder and the serving engine on a real MI355X: the whole GPU stack (HIP kernels +
hipBLASLt GEMMs + hipGraph replay) against the fp32 CPU path of the same model, and the engine's
vLLM metrics scraped over HTTP by the monitor's own collector."""
import json
import time
import urllib.request

import pytest
import torch

pytestmark = pytest.mark.gpu

from k8s_llm_monitor_amd.collectors.vllm import VllmDirectSource  # noqa: E402
from k8s_llm_monitor_amd.models.llama import PRESETS, DecodeMeta, LlamaModel, PrefillMeta  # noqa: E402
from k8s_llm_monitor_amd.ops import llm  # noqa: E402
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402
from k8s_llm_monitor_amd.serving.loadclient import OpenLoopLoad  # noqa: E402
from k8s_llm_monitor_amd.serving.server import LlmServer  # noqa: E402

SMALL = PRESETS["small"]


@pytest.fixture(scope="module")
def gpu_model():
    return LlamaModel(SMALL, "cuda", seed=3)


@pytest.fixture(scope="module")
def cpu_model(gpu_model):
    return LlamaModel(SMALL, "cpu", seed=0, dtype=torch.float32).load_from(gpu_model)


def _prefill_inputs(prompts, dev, max_blocks, first_block=0):
    toks, pos, slots, info, last = [], [], [], [], []
    bt = torch.zeros((len(prompts), max_blocks), dtype=torch.int32)
    blk = first_block
    for i, p in enumerate(prompts):
        n = len(p)
        nb = -(-(n + 1) // 16)
        ids = list(range(blk, blk + nb))
        blk += nb
        bt[i, :nb] = torch.tensor(ids, dtype=torch.int32)
        info.append((len(toks), n, n))
        toks += p
        pos += list(range(n))
        slots += [ids[j // 16] * 16 + j % 16 for j in range(n)]
        last.append(len(toks) - 1)
    meta = PrefillMeta.build(SMALL, [len(p) for p in prompts], bt.to(dev),
                             torch.tensor(info, dtype=torch.int32, device=dev),
                             torch.tensor(last, dtype=torch.int64, device=dev), dev)
    return (torch.tensor(toks, dtype=torch.int64, device=dev), torch.tensor(pos, dtype=torch.int32, device=dev),
            torch.tensor(slots, dtype=torch.int32, device=dev), meta)


def _rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("n_seqs", [3, 20, 40, 150, 230])  # decode on the skinny kernel (3), the mid-batch kernel (20, 40), the K-split GEMM (150, 230: wo on the library)
def test_gpu_logits_match_the_fp32_cpu_model(gpu_model, cpu_model, n_seqs):
    assert gpu_model.skinny(3) and gpu_model.mid(20) and gpu_model.mid(40) and gpu_model.mid(129) and gpu_model.mid(256) and not gpu_model.mid(257)
    _compare_logits(gpu_model, cpu_model, n_seqs)


def test_gpu_logits_with_the_separate_rope_kernel(gpu_model, cpu_model):
    gpu_model.fused_rope = False                         # skinny path: rope_cache launch + attention instead of the fused kernel
    try:
        _compare_logits(gpu_model, cpu_model, 3)
    finally:
        gpu_model.fused_rope = True


def test_gpu_logits_on_the_library_gemm_path(gpu_model, cpu_model):
    gpu_model.mid_limit = 0                              # batches above 16 rows on hipBLASLt + norm / SwiGLU kernels
    try:
        assert not gpu_model.mid(20)
        _compare_logits(gpu_model, cpu_model, 20)
    finally:
        gpu_model.mid_limit = None


def test_fp8_decode_weights_match_the_fp32_cpu_model_with_the_dequantised_weights():
    g = LlamaModel(SMALL, "cuda", seed=4).enable_fp8_decode()
    c = LlamaModel(SMALL, "cpu", seed=0, dtype=torch.float32).load_from(g)     # the dequantised weights, no fp8
    assert g.layers[0]["w_down8"].dtype == torch.float8_e4m3fn and not c.fp8_decode
    _compare_logits(g, c, 5)
    # past 128 rows the same model decodes on the tiled GEMM over its bf16 weights (which hold the dequantised values)
    assert g.mid(150) and g.decode_stream_bytes(150) == 2 * g.decode_stream_bytes(5)
    _compare_logits(g, c, 150)


def test_fp8_kv_cache_stays_close_to_the_bf16_cache_model(gpu_model, cpu_model):
    _compare_logits(gpu_model, cpu_model, 4, kv_dtype=torch.float8_e4m3fn, tol=0.08)
    _compare_logits(gpu_model, cpu_model, 20, kv_dtype=torch.float8_e4m3fn, tol=0.08)


def _compare_logits(gpu_model, cpu_model, n_seqs, kv_dtype=None, tol=0.03):
    g = torch.Generator().manual_seed(n_seqs)
    lens = [int(x) for x in torch.randint(5, 90, (n_seqs,), generator=g)]
    prompts = [torch.randint(0, SMALL.vocab, (n,), generator=g).tolist() for n in lens]
    mb, nblocks = 8, max(300, 6 * n_seqs + 8)
    out = {}
    for name, model in (("cpu", cpu_model), ("gpu", gpu_model)):
        dev = model.device
        cache = model.new_cache(nblocks, kv_dtype if dev.type == "cuda" else None)
        t, p, s, meta = _prefill_inputs(prompts, dev, mb)
        pre = model.prefill(t, p, s, cache, meta, logits=True)
        # one decode step on top of the prefilled cache, fed with the same (CPU-chosen) tokens
        nxt = out["cpu_next"] if name == "gpu" else pre.argmax(-1)
        out.setdefault("cpu_next", nxt.cpu())
        ctx = torch.tensor([n + 1 for n in lens], dtype=torch.int32, device=dev)
        posd = torch.tensor(lens, dtype=torch.int32, device=dev)
        slotd = torch.stack([meta.block_tables[i, n // 16] * 16 + n % 16 for i, n in enumerate(lens)]).to(torch.int32)
        dmeta = DecodeMeta(meta.block_tables, ctx, num_splits=2 if dev.type == "cuda" else 1)
        dec = model.decode(out["cpu_next"].to(dev), posd, slotd, cache, dmeta, logits=True)
        out[name] = (pre, dec)
    assert _rel_err(out["gpu"][0], out["cpu"][0]) < tol, "prefill logits"
    assert _rel_err(out["gpu"][1], out["cpu"][1]) < tol, "decode logits"


@pytest.mark.parametrize("n_ride,new_lens", [(5, [40, 70]), (30, [33]), (60, [100, 90, 17])])   # 115 and 63 rows: mid-batch kernels; 267: library
def test_mixed_batch_logits_match_the_fp32_cpu_model(gpu_model, cpu_model, n_ride, new_lens):
    """New prompts and one decode token of each running sequence in ONE launch: chunk rows through the prefill attention
    kernel, the riders' rows through the decode kernel, the projections over all rows together."""
    g = torch.Generator().manual_seed(n_ride)
    lens = [int(x) for x in torch.randint(5, 90, (n_ride,), generator=g)]
    running = [torch.randint(0, SMALL.vocab, (n,), generator=g).tolist() for n in lens]
    fresh = [torch.randint(0, SMALL.vocab, (n,), generator=g).tolist() for n in new_lens]
    mb, nblocks, out = 8, 600, {}
    for name, model in (("cpu", cpu_model), ("gpu", gpu_model)):
        dev = model.device
        cache = model.new_cache(nblocks)
        t, p, s, meta = _prefill_inputs(running, dev, mb)
        pre = model.prefill(t, p, s, cache, meta, logits=True)
        out.setdefault("next", pre.argmax(-1).cpu())                  # the CPU model's tokens feed both
        used = sum(-(-(n + 1) // 16) for n in lens)
        tb, pb, sb_, mB = _prefill_inputs(fresh, dev, mb, first_block=used)
        slotd = torch.stack([meta.block_tables[i, n // 16] * 16 + n % 16 for i, n in enumerate(lens)]).to(torch.int32)
        rows = torch.cat([mB.last_rows, tb.shape[0] + torch.arange(n_ride, device=dev)])
        mixed = PrefillMeta(mB.block_tables, mB.seq_info, mB.tiles, rows, mB.q_tiles)
        riders = DecodeMeta(meta.block_tables, torch.tensor([n + 1 for n in lens], dtype=torch.int32, device=dev),
                            num_splits=2 if dev.type == "cuda" else 1)
        out[name] = model.prefill(torch.cat([tb, out["next"].to(dev)]), torch.cat([pb, torch.tensor(lens, dtype=torch.int32, device=dev)]),
                                  torch.cat([sb_, slotd]), cache, mixed, logits=True, riders=riders)
    assert tuple(out["gpu"].shape) == (len(new_lens) + n_ride, SMALL.vocab)
    k = len(new_lens)
    assert _rel_err(out["gpu"][:k], out["cpu"][:k]) < 0.03, "chunk rows"
    assert _rel_err(out["gpu"][k:], out["cpu"][k:]) < 0.03, "rider rows"


def test_engine_with_mixed_batches_keeps_running_sequences_moving(gpu_model):
    """Arrivals while others decode: with mixed batches no launch leaves the running sequences out, and both modes produce
    (near-ties aside) the same text."""
    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, SMALL.vocab, (n,), generator=g).tolist() for n in (30, 200, 120, 64, 90, 150)]
    outs, counts = [], []
    for mixed in (True, False):
        eng = Engine(EngineConfig(model="small", max_num_seqs=16, max_model_len=512, kv_blocks=512, max_num_batched_tokens=128,
                                  mixed_batches=mixed), model=gpu_model)
        reqs = []
        for i, p in enumerate(prompts):                 # one arrival every third step
            reqs.append(eng.submit(p, 12))
            for _ in range(3):
                eng.step()
        n = 0
        while not eng.idle():
            eng.step()
            n += 1
            assert n < 2000
        outs.append([r.out for r in reqs])
        counts.append(dict(eng.step_counts))
        assert eng.alloc.free == eng.alloc.total and all(len(r.out) == 12 for r in reqs)
    assert counts[0]["mixed"] >= 5 and counts[1]["mixed"] == 0
    assert sum(counts[0].values()) < sum(counts[1].values())          # fewer launches for the same work
    same = sum(a[:4] == b[:4] for a, b in zip(*outs))
    assert same >= len(prompts) - 2, outs


def _run(engine, prompts, max_tokens):
    reqs = [engine.submit(p, max_tokens) for p in prompts]
    n = 0
    while not engine.idle():
        engine.step()
        n += 1
        assert n < 5000
    return [r.out for r in reqs]


def test_graph_replay_and_eager_decode_generate_the_same_tokens(gpu_model):
    g = torch.Generator().manual_seed(11)
    prompts = [torch.randint(0, SMALL.vocab, (n,), generator=g).tolist() for n in (7, 33, 64, 100, 16)]
    outs = []
    for graphs in (True, False):
        eng = Engine(EngineConfig(model="small", max_num_seqs=32, max_model_len=512, kv_blocks=256, use_graphs=graphs),
                     model=gpu_model)
        outs.append(_run(eng, prompts, 24))
        assert eng.alloc.free == eng.alloc.total
        if graphs:
            assert eng._graphs, "decode did not run from a hipGraph"
    assert outs[0] == outs[1]
    assert all(len(o) == 24 for o in outs[0])


def test_engine_under_load_is_scraped_by_the_monitor_collector(gpu_model):
    eng = Engine(EngineConfig(model="small", max_num_seqs=64, max_model_len=1024, kv_blocks=2048), model=gpu_model)
    eng.warmup([1, 2, 4, 8, 16, 32, 64])
    eng.start()
    try:
        with LlmServer(eng) as srv:
            src = VllmDirectSource([srv.url])
            assert src.sample()["available"]
            load = OpenLoopLoad(eng, rate_rps=200, prompt_len=128, output_len=48, seed=2).start(3.0)
            time.sleep(1.5)
            mid = src.sample()
            load.join(10)
            assert load.wait_all(60)
            end = src.sample()
            truth = json.load(urllib.request.urlopen(srv.url + "/stats", timeout=5))["truth"]
    finally:
        eng.stop()
    assert mid["running"] > 0 and mid["tokensPerSec"] > 0 and mid["kvCachePercent"] > 0
    assert end["running"] == 0 and end["waiting"] == 0 and end["kvCachePercent"] == 0
    assert truth["totals"]["finished"] == len(load.requests) and truth["totals"]["generation_tokens"] > 0
    assert 0 < truth["ttft_p50"] < 1.0
    # the scraped median sits in the same histogram bucket as the exact one
    from k8s_llm_monitor_amd.serving.metrics import TTFT_BUCKETS
    import bisect
    b = bisect.bisect_left(TTFT_BUCKETS, truth["ttft_p50"])
    lo, hi = (TTFT_BUCKETS[b - 1] if b else 0.0), TTFT_BUCKETS[min(b, len(TTFT_BUCKETS) - 1)]
    assert end["ttftP50"] is None or lo * 0.5 <= end["ttftP50"] <= hi * 2


def test_decode_step_timing_helper_reports_the_roofline_terms(gpu_model):
    eng = Engine(EngineConfig(model="small", max_num_seqs=16, max_model_len=512, kv_blocks=512), model=gpu_model)
    r = sb.decode_step_ms(eng, 4, 256, iters=5)
    assert r["ms"] > 0 and r["tokens_per_s"] > 0 and r["bytes"] > eng.model.weight_bytes() // 2


def test_chunked_prefill_on_the_gpu_gives_the_first_token_of_the_unchunked_run(gpu_model):
    """The same prompts through one big prefill launch and through 64-token chunks: the chunks attend to
    their predecessors through the paged cache (prefill kernel with q_len < ctx_len). bf16 GEMMs of
    different shapes may flip a near-tie, so most (not all) first tokens must agree."""
    g = torch.Generator().manual_seed(21)
    prompts = [torch.randint(0, SMALL.vocab, (n,), generator=g).tolist() for n in (150, 200, 97, 310, 64, 180)]
    firsts = []
    for budget in (4096, 64):
        eng = Engine(EngineConfig(model="small", max_num_seqs=16, max_model_len=512, kv_blocks=512,
                                  max_num_batched_tokens=budget), model=gpu_model)
        firsts.append([o[0] for o in _run(eng, prompts, 2)])
        if budget == 64:
            assert eng.step_counts["prefill"] + eng.step_counts["mixed"] >= sum(-(-len(p) // 64) for p in prompts) // 2
        assert eng.alloc.free == eng.alloc.total
    agree = sum(a == b for a, b in zip(*firsts))
    assert agree >= len(prompts) - 1, firsts


def test_two_replica_processes_on_the_gpu_behind_the_router():
    """The per-GPU replica launcher with real GPU children. One MI355X is all a test box has, so both
    children are pinned to device 0 (on a node each gets its own); what is checked is the launch path:
    HIP_VISIBLE_DEVICES set before the child touches HIP, readiness, routing, merged metrics."""
    import socket
    from k8s_llm_monitor_amd.collectors.vllm import parse_exposition
    from k8s_llm_monitor_amd.serving.router import ReplicaSet, Router
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    base = s.getsockname()[1]
    s.close()
    rs = ReplicaSet([0, 0], base, ["--model", "small", "--kv-blocks", "256", "--max-model-len", "512",
                                   "--max-num-seqs", "8"]).start()
    try:
        assert rs.wait_ready(240), "replicas did not come up"
        with Router(rs.urls) as router:
            outs = []
            for i in range(4):
                req = urllib.request.Request(router.url + "/v1/completions",
                                             data=json.dumps({"prompt_len": 40, "seed": i, "max_tokens": 6}).encode(),
                                             headers={"Content-Type": "application/json"})
                outs.append(json.load(urllib.request.urlopen(req, timeout=120)))
            assert {o["replica"] for o in outs} == {0, 1} and all(len(o["choices"][0]["token_ids"]) == 6 for o in outs)
            text = urllib.request.urlopen(router.url + "/metrics", timeout=10).read().decode()
            gen = {l.get("replica"): v for (n, l, v) in parse_exposition(text) if n == "vllm:generation_tokens_total"}
            assert gen == {"0": 12.0, "1": 12.0}
    finally:
        rs.stop()
