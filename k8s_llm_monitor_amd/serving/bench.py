"""Direct timing of the engine's two model launches (no scheduler, no HTTP): decode step time per
batch bucket and context length, prefill time per prompt length, with the memory-roofline each
is bounded by. Used by scripts/gpu_llm_bench.py, bench.py extras and the GPU tests."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from ..models.llama import BLOCK, PrefillMeta
from .engine import Engine


def _synthetic_tables(eng: Engine, batch: int, ctx: int) -> None:
    """Point `batch` rows of the static decode inputs at distinct blocks holding `ctx` tokens."""
    nblk = -(-ctx // BLOCK)
    if batch * nblk > eng.alloc.total:
        raise ValueError(f"batch {batch} x ctx {ctx} needs {batch * nblk} blocks, cache has {eng.alloc.total}")
    bt = np.zeros((eng.cfg.max_num_seqs, eng.max_blocks), dtype=np.int32)
    for i in range(batch):
        bt[i, :nblk] = np.arange(i * nblk, (i + 1) * nblk)
    eng._d_bt.copy_(torch.from_numpy(bt))
    i32 = np.zeros((3, eng.cfg.max_num_seqs), dtype=np.int32)
    i32[1] = -1
    i32[0, :batch] = ctx - 1
    i32[1, :batch] = bt[:batch, nblk - 1] * BLOCK + (ctx - 1) % BLOCK
    i32[2, :batch] = ctx
    eng._d_i32.copy_(torch.from_numpy(i32))
    eng._d_tokens.random_(0, eng.mcfg.vocab)


def decode_step_ms(eng: Engine, batch: int, ctx: int, iters: int = 20, warmup: int = 3) -> Dict[str, float]:
    """Time of one decode step (graph replay when graphs are on) for `batch` sequences of `ctx` tokens."""
    assert eng.idle(), "benchmark the engine while it serves nothing"
    b = next(x for x in eng.buckets if x >= batch)
    _synthetic_tables(eng, batch, ctx)
    for _ in range(warmup):
        eng._run_decode(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng._run_decode(b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    c, lc, world = eng.mcfg, eng.lcfg, eng.tp.world        # figures are per rank: this GPU's bytes and FLOPs
    weight_bytes = eng.model.decode_stream_bytes(b)
    kv_bytes = batch * ctx * lc.kv_bytes_per_token() // (2 if eng.cfg.kv_cache_dtype == "fp8" else 1)
    return {"batch": batch, "bucket": b, "ctx": ctx, "ms": ms, "tokens_per_s": batch / ms * 1e3,
            "bytes": weight_bytes + kv_bytes, "TBps": (weight_bytes + kv_bytes) / ms / 1e9,
            "TFLOPs": batch * c.flops_per_token(ctx) / world / ms / 1e9}


def prefill_ms(eng: Engine, n_seqs: int, prompt_len: int, iters: int = 3, warmup: int = 1) -> Dict[str, float]:
    assert eng.idle()
    c = eng.mcfg
    nblk = -(-prompt_len // BLOCK)
    if n_seqs * nblk > eng.alloc.total:
        raise ValueError("cache too small for this prefill")
    dev = eng.device
    T = n_seqs * prompt_len
    bt = np.zeros((n_seqs, eng.max_blocks), dtype=np.int32)
    slots = np.zeros(T, dtype=np.int32)
    for i in range(n_seqs):
        blocks = np.arange(i * nblk, (i + 1) * nblk)
        bt[i, :nblk] = blocks
        idx = np.arange(prompt_len)
        slots[i * prompt_len:(i + 1) * prompt_len] = blocks[idx // BLOCK] * BLOCK + idx % BLOCK
    info = [(i * prompt_len, prompt_len, prompt_len) for i in range(n_seqs)]
    meta = PrefillMeta.build(eng.lcfg, [prompt_len] * n_seqs, torch.from_numpy(bt).to(dev),
                             torch.tensor(info, dtype=torch.int32, device=dev),
                             torch.tensor([(i + 1) * prompt_len - 1 for i in range(n_seqs)], dtype=torch.int64, device=dev),
                             dev)
    toks = torch.randint(0, c.vocab, (T,), device=dev)
    pos = torch.arange(prompt_len, dtype=torch.int32, device=dev).repeat(n_seqs)
    sl = torch.from_numpy(slots).to(dev)
    for _ in range(warmup):
        eng.model.prefill(toks, pos, sl, eng.cache, meta)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng.model.prefill(toks, pos, sl, eng.cache, meta)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = sum(c.flops_per_token(p) for p in range(prompt_len)) * n_seqs / eng.tp.world
    eng.cache.k.zero_()
    eng.cache.v.zero_()
    return {"n_seqs": n_seqs, "prompt_len": prompt_len, "ms": ms, "tokens_per_s": T / ms * 1e3,
            "TFLOPs": flops / ms / 1e9}


def sweep(eng: Engine, batches: List[int], ctx: int, iters: int = 20) -> List[Dict[str, float]]:
    return [decode_step_ms(eng, b, ctx, iters=iters) for b in batches]
