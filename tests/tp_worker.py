"""One rank of the tensor-parallel model / engine tests. CPU + gloo under torch.distributed.run (test_llm_tp.py), or
GPU ranks sharing cuda:0 with the xGMI mesh kernels (test_gpu_tp.py, --gpu). Prints one line ``RESULT {json}``."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_llm_monitor_amd.models.llama import BLOCK, PRESETS, DecodeMeta, LlamaModel, PrefillMeta  # noqa: E402
from k8s_llm_monitor_amd.parallel.tp import TPGroup  # noqa: E402
from k8s_llm_monitor_amd.serving.engine import Engine, EngineConfig  # noqa: E402


def model_check(cfg, dev, dtype, tp, out):
    """The sharded model against the whole one: prefill logits of two prompts, then three decode steps."""
    full = LlamaModel(cfg, dev, seed=3, dtype=dtype)
    part = LlamaModel(cfg, dev, seed=0, dtype=dtype, tp=tp).load_shard_from(full)
    lens = [37, 21]
    gen = torch.Generator().manual_seed(11)
    prompts = [torch.randint(0, cfg.vocab, (n,), generator=gen).tolist() for n in lens]
    nb = 16
    worst, agree = 0.0, True
    caches = {m: m.new_cache(nb) for m in (full, part)}
    bt = torch.tensor([[0, 1, 2, 3], [4, 5, 6, 7]], dtype=torch.int32, device=dev)
    toks = torch.tensor(sum(prompts, []), dtype=torch.int64, device=dev)
    pos = torch.tensor(sum([list(range(n)) for n in lens], []), dtype=torch.int32, device=dev)
    slots = torch.tensor(sum([[int(bt[i, p // BLOCK]) * BLOCK + p % BLOCK for p in range(n)] for i, n in enumerate(lens)], []),
                         dtype=torch.int32, device=dev)
    info = torch.tensor([[0, lens[0], lens[0]], [lens[0], lens[1], lens[1]]], dtype=torch.int32, device=dev)
    last = torch.tensor([lens[0] - 1, lens[0] + lens[1] - 1], dtype=torch.int64, device=dev)
    res = {}
    for m in (full, part):
        meta = PrefillMeta.build(m.lcfg, lens, bt, info, last, dev)
        res[m] = m.prefill(toks, pos, slots, caches[m], meta, logits=True).float()
    worst = max(worst, float((res[full] - res[part]).abs().max()))
    scale = float(res[full].abs().max())
    cur = res[full].argmax(-1)
    agree = agree and bool(torch.equal(cur, res[part].argmax(-1)))
    n = list(lens)
    for _ in range(3):
        p = torch.tensor(n, dtype=torch.int32, device=dev)
        sl = torch.tensor([int(bt[i, n[i] // BLOCK]) * BLOCK + n[i] % BLOCK for i in range(2)], dtype=torch.int32, device=dev)
        ctx = torch.tensor([x + 1 for x in n], dtype=torch.int32, device=dev)
        lg = {m: m.decode(cur, p, sl, caches[m], DecodeMeta(bt, ctx), logits=True).float() for m in (full, part)}
        ids = part.decode(cur, p, sl, caches[part], DecodeMeta(bt, ctx))          # the sharded argmax
        worst = max(worst, float((lg[full] - lg[part]).abs().max()))
        agree = agree and bool(torch.equal(ids, lg[part].argmax(-1)))
        cur = lg[full].argmax(-1)
        n = [x + 1 for x in n]
    out.update(model_max_abs_err=worst, model_logit_scale=scale, sharded_argmax_agrees=agree,
               local_shapes=[part.lcfg.n_q, part.lcfg.n_kv, part.lcfg.ffn, part.lcfg.vocab])


def engine_check(cfg, dev, dtype, tp, out, use_graphs):
    """The lock-step engine: rank 0 submits, every rank steps; the tokens must be those of a one-rank engine that
    holds the whole model."""
    full = LlamaModel(cfg, dev, seed=5, dtype=dtype)
    part = LlamaModel(cfg, dev, seed=0, dtype=dtype, tp=tp).load_shard_from(full)
    ecfg = EngineConfig(model=cfg.name, device=str(dev), max_num_seqs=8, max_model_len=256, max_num_batched_tokens=96,
                        kv_blocks=64, use_graphs=use_graphs)
    gen = torch.Generator().manual_seed(23)
    prompts = [torch.randint(0, cfg.vocab, (n,), generator=gen).tolist() for n in (40, 130, 7, 65)]
    want = None
    if tp.rank == 0:                        # the reference run needs no collective: only rank 0 pays for it
        ref = Engine(ecfg, model=full)
        rr = [ref.submit(p, 6) for p in prompts]
        while not ref.idle():
            ref.step()
        want = [r.out for r in rr]
    def run(e):
        reqs = [e.submit(p, 6) for p in prompts] if tp.rank == 0 else []
        e.step_all()
        steps = 1
        while not e.idle():
            e.step_all()
            steps += 1
        tp.check()
        return reqs, steps

    eng = Engine(ecfg, model=part)
    reqs, steps = run(eng)
    out.update(engine_steps=steps, engine_counts=dict(eng.step_counts), graphs=sorted(eng._graphs))
    if tp.rank == 0:
        got = [r.out for r in reqs]
        out.update(engine_tokens_match=got == want, engine_done=all(r.done.is_set() and not r.error for r in reqs),
                   engine_tokens=sum(len(o) for o in got),
                   engine_first_tokens_match=[g[0] for g in got] == [w[0] for w in want])
    if use_graphs:                          # the same sharded arithmetic without hipGraphs must give the same tokens
        from dataclasses import replace
        eager = Engine(replace(ecfg, use_graphs=False), model=part)
        reqs2, steps2 = run(eager)
        if tp.rank == 0:
            out.update(graph_equals_eager=[r.out for r in reqs2] == got and steps2 == steps)
    # the background loop: rank 0 serves from a thread, the others follow until it stops
    # ... and a launch that fails (on every rank: a bad shape would) costs the requests in flight, not the lock-step
    real_prefill, state = eng.model.prefill, {"failed": False}

    def failing_once(*a, **kw):
        if not state["failed"]:
            state["failed"] = True
            raise RuntimeError("injected launch failure")
        return real_prefill(*a, **kw)
    eng.model.prefill = failing_once
    if tp.rank == 0:
        eng.start()
        bad = eng.submit(prompts[2], 3)
        bad.done.wait(120)
        r = eng.submit(prompts[0], 4)
        ok = r.done.wait(120)
        eng.stop()
        out.update(loop_ok=bool(ok) and r.out == want[0][:4], failed_request_error=bad.error, engine_errors=eng.errors)
    else:
        eng.follow()
        out.update(loop_ok=True, engine_errors=eng.errors)
    eng.model.prefill = real_prefill


def unequal_check(cfg, dev, dtype, tp, out):
    """Ranks that would size their KV cache differently (their own free memory, their own config) agree on the
    smallest count at construction and then schedule identically."""
    full = LlamaModel(cfg, dev, seed=5, dtype=dtype)
    part = LlamaModel(cfg, dev, seed=0, dtype=dtype, tp=tp).load_shard_from(full)
    ecfg = EngineConfig(model=cfg.name, device=str(dev), max_num_seqs=8, max_model_len=256, max_num_batched_tokens=96,
                        kv_blocks=40 - 12 * tp.rank, use_graphs=False)          # rank 0: 40 blocks, rank 1: 28
    eng = Engine(ecfg, model=part)
    gen = torch.Generator().manual_seed(29)
    # 4 x (100 + 24) tokens = 32 blocks: more than 28, so the engine with the agreed count must preempt or queue --
    # identically on both ranks -- where a 40-block rank alone would not
    prompts = [torch.randint(0, cfg.vocab, (100,), generator=gen).tolist() for _ in range(4)]
    reqs = [eng.submit(p, 24) for p in prompts] if tp.rank == 0 else []
    eng.step_all()
    steps = 1
    while not eng.idle() and steps < 2000:
        eng.step_all()
        steps += 1
    out.update(unequal_blocks_total=eng.alloc.total, unequal_counts=list(eng.kv_blocks_per_rank), unequal_steps=steps,
               unequal_step_counts=dict(eng.step_counts), unequal_resets=eng.tp_resets,
               unequal_preemptions=eng.metrics.truth()["totals"]["preemptions"])
    if tp.rank == 0:
        ref = Engine(EngineConfig(model=cfg.name, device=str(dev), max_num_seqs=8, max_model_len=256, max_num_batched_tokens=96,
                                  kv_blocks=28, use_graphs=False), model=full)
        rr = [ref.submit(p, 24) for p in prompts]
        while not ref.idle():
            ref.step()
        out.update(unequal_tokens_match=[r.out for r in reqs] == [r.out for r in rr],
                   unequal_done=all(r.done.is_set() and not r.error for r in reqs))


def follower_failure_check(cfg, dev, dtype, tp, out):
    """GPU + mesh kernels only: a launch fails on rank 1 ALONE. Rank 0's collectives give up after the spin limit (once:
    later barriers of the step wait for a fraction of it), its tp.check() fails the step, and the next control exchange
    resets every rank; the request after that is served correctly."""
    import time
    full = LlamaModel(cfg, dev, seed=5, dtype=dtype)
    part = LlamaModel(cfg, dev, seed=0, dtype=dtype, tp=tp).load_shard_from(full)
    ecfg = EngineConfig(model=cfg.name, device=str(dev), max_num_seqs=8, max_model_len=256, max_num_batched_tokens=96,
                        kv_blocks=64, use_graphs=False)
    eng = Engine(ecfg, model=part)
    gen = torch.Generator().manual_seed(31)
    prompts = [torch.randint(0, cfg.vocab, (n,), generator=gen).tolist() for n in (33, 50)]
    real_prefill, state = eng.model.prefill, {"failed": False}

    def failing_once(*a, **kw):
        if tp.rank == 1 and not state["failed"]:
            state["failed"] = True
            raise RuntimeError("injected failure on rank 1 only")
        return real_prefill(*a, **kw)
    eng.model.prefill = failing_once
    if tp.rank == 0:
        ref = Engine(ecfg, model=full)
        rr = ref.submit(prompts[1], 5)
        while not ref.idle():
            ref.step()
        t0 = time.monotonic()
        eng.start()
        bad = eng.submit(prompts[0], 3)
        bad.done.wait(120)
        t_fail = time.monotonic() - t0
        good = eng.submit(prompts[1], 5)
        ok = good.done.wait(120)
        eng.stop()
        out.update(follower_failed_request_error=bad.error, follower_recovered=bool(ok) and good.out == rr.out and not good.error,
                   follower_seconds_to_fail=t_fail, follower_resets=eng.tp_resets, follower_errors=eng.errors)
    else:
        eng.follow()
        out.update(follower_resets=eng.tp_resets, follower_errors=eng.errors)
    eng.model.prefill = real_prefill


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--what", default="model,engine")
    ap.add_argument("--spin-limit", type=float, default=20.0)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if a.gpu:
        dev, dtype = torch.device("cuda", 0), torch.bfloat16
        torch.cuda.set_device(dev)
    else:
        dev, dtype = torch.device("cpu"), torch.float32
    tp = TPGroup(dev, data="xgmi" if a.gpu else "gloo", slot_bytes=8 << 20, spin_limit_s=a.spin_limit)
    out = {"rank": rank, "tp": tp.describe()}
    cfg = PRESETS[a.model]
    with torch.no_grad():
        if "model" in a.what:
            model_check(cfg, dev, dtype, tp, out)
        if "engine" in a.what:
            engine_check(cfg, dev, dtype, tp, out, use_graphs=a.gpu)
        if "unequal" in a.what:
            unequal_check(cfg, dev, dtype, tp, out)
        if "follower" in a.what:
            follower_failure_check(cfg, dev, dtype, tp, out)
    tp.barrier()
    tp.close()
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
