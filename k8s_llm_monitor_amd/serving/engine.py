"""Continuous-batching LLM engine on one MI355X: paged KV cache, iteration-level scheduling,
prefill-first admission, recompute preemption, hipGraph decode.

This is the "vLLM-ROCm pod" of BASELINE.json config 4 as a self-contained load generator: it
serves a Llama-3 architecture model with random weights (models/llama.py), exposes vLLM's
metric names (serving/metrics.py) and knows its own ground truth, so the monitor's vLLM panel
can be validated on hardware without vLLM, a checkpoint or a network. No reference counterpart
(the reference only mentions vLLM, README.md:73).

Scheduling (one `step()` = one model launch):
  1. prefill: the requests already in prefill continue, then waiting requests are admitted while
     blocks and sequence slots allow, up to `max_num_batched_tokens` tokens per launch. A prompt
     longer than what is left of the budget is CHUNKED: its next chunk attends to the chunks already
     in the paged cache (the prefill kernel takes q_len < ctx_len). A request gets its first token
     (TTFT) from the launch that finishes its prompt;
  2. decode one token for every running sequence. After a prefill launch the running sequences get
     a decode step before the next prefill chunk, so a stream of long prompts delays them by one
     chunk at a time, not by whole prompts. A sequence that needs a new block when none is free
     preempts the most recently admitted one (prefilling before running): its blocks are freed, it
     re-enters the queue at the front and is recomputed from prompt + generated tokens.
Decode runs from a hipGraph captured per batch-size bucket (launch-bound otherwise: ~10 kernel
launches per layer); the graph's inputs are static device buffers that each step refills.
"""
from __future__ import annotations

import threading
import time
from collections import deque
from dataclasses import dataclass, field
from typing import Callable, Deque, Dict, List, Optional, Sequence

import numpy as np
import torch

from ..models.llama import BLOCK, PRESETS, DecodeMeta, KVCache, LlamaConfig, LlamaModel, PrefillMeta
from ..ops import llm
from .metrics import EngineMetrics


@dataclass
class EngineConfig:
    model: str = "llama3-8b"
    device: str = "cuda"
    max_num_seqs: int = 256
    max_model_len: int = 4096
    max_num_batched_tokens: int = 8192
    kv_blocks: int = 0                 # 0: size the cache from kv_fraction of the free HBM
    kv_fraction: float = 0.5
    use_graphs: bool = True
    kv_cache_dtype: str = "bf16"       # "fp8": e4m3 KV cache (half the bytes per token; prefill then uses the register kernel)
    decode_weights: str = "bf16"       # "fp8": W8A16 copies for decode up to batch 128 (LlamaModel.enable_fp8_decode)
    mixed_batches: bool = True         # running sequences decode inside prefill launches instead of waiting for them
    seed: int = 0


class BlockAllocator:
    def __init__(self, num_blocks: int):
        self.total = int(num_blocks)
        self._free: List[int] = list(range(num_blocks - 1, -1, -1))

    @property
    def free(self) -> int:
        return len(self._free)

    @property
    def usage(self) -> float:
        return 1.0 - len(self._free) / self.total if self.total else 0.0

    def alloc(self, n: int) -> List[int]:
        if n > len(self._free):
            raise MemoryError(f"{n} blocks wanted, {len(self._free)} free")
        out = self._free[-n:] if n else []
        del self._free[len(self._free) - n:]
        return out

    def release(self, blocks: Sequence[int]) -> None:
        self._free.extend(blocks)


@dataclass
class Request:
    rid: int
    prompt: List[int]
    max_tokens: int
    arrival: float
    out: List[int] = field(default_factory=list)
    blocks: List[int] = field(default_factory=list)
    t_first: Optional[float] = None
    t_last: Optional[float] = None
    t_done: Optional[float] = None
    preempted: int = 0
    prefilled: int = 0                # tokens of prompt + out whose K/V are in the cache (chunked prefill)
    error: Optional[str] = None
    done: threading.Event = field(default_factory=threading.Event)

    @property
    def n_tokens(self) -> int:
        return len(self.prompt) + len(self.out)

    @property
    def ttft(self) -> Optional[float]:
        return None if self.t_first is None else self.t_first - self.arrival


def _buckets(limit: int) -> List[int]:
    """Decode batch sizes that get a hipGraph: powers of two, from 16 on the midpoints (24, 48, 96, ...) and from 128 on the
    quarter points (160, 224, ...): a step costs what its bucket costs (Llama-3-8B: 5.6 ms at 48, 6.2 ms at 64; 10.5 ms at
    160, 11.4 ms at 192), so 40 running sequences should not pay for 64, nor 150 for 192."""
    out, b = [], 1
    while b < limit:
        out.append(b)
        if b >= 128:
            out.extend(x for x in (b + b // 4, b + b // 2, b + 3 * b // 4) if x < limit)
        elif b >= 16 and b + b // 2 < limit:
            out.append(b + b // 2)
        b *= 2
    out.append(limit)
    return out


class Engine:
    def __init__(self, cfg: Optional[EngineConfig] = None, model: Optional[LlamaModel] = None,
                 clock: Callable[[], float] = time.monotonic, tp=None):
        """`tp`: a parallel.tp.TPGroup. Every rank then runs this same engine in lock-step (see `_tp_sync`):
        requests enter at rank 0, every rank schedules them identically and holds its shard of model and cache."""
        self.cfg = cfg or EngineConfig()
        self.clock = clock
        mcfg: LlamaConfig = model.cfg if model is not None else PRESETS[self.cfg.model]
        self.device = torch.device(self.cfg.device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.on_gpu = self.device.type == "cuda"
        self.max_len = min(self.cfg.max_model_len, mcfg.max_pos)
        self.model = model or LlamaModel(mcfg, self.device, seed=self.cfg.seed, tp=tp)
        self.tp = self.model.tp
        self.lcfg: LlamaConfig = self.model.lcfg          # this rank's shapes (== mcfg without tensor parallelism)
        self._inbox: List[Request] = []                   # tensor parallel: submitted at rank 0, not yet seen by the others
        self._tp_reset = False                            # this rank failed a step: the next control exchange makes every rank start clean
        self.tp_resets = 0
        if self.cfg.decode_weights == "fp8" and not self.model.fp8_decode:
            self.model.enable_fp8_decode()
        elif self.cfg.decode_weights not in ("bf16", "fp8"):
            raise ValueError(f"decode_weights must be bf16 or fp8, not {self.cfg.decode_weights!r}")
        self.mcfg = mcfg
        self.max_blocks = -(-self.max_len // BLOCK)
        if self.cfg.kv_cache_dtype not in ("bf16", "fp8"):
            raise ValueError(f"kv_cache_dtype must be bf16 or fp8, not {self.cfg.kv_cache_dtype!r}")
        self.kv_dtype = torch.float8_e4m3fn if self.cfg.kv_cache_dtype == "fp8" else None
        nblocks = self.cfg.kv_blocks or self._auto_blocks()
        if self.tp.world > 1:
            # lock-step scheduling needs the SAME block count everywhere: free HBM differs per rank (ranks sharing a GPU see
            # what earlier ranks left; a config file may differ), and one rank admitting or preempting alone desynchronises
            # the collectives. Everyone takes the smallest.
            counts = self.tp.gather_objects(int(nblocks))
            self.kv_blocks_per_rank = counts
            nblocks = min(counts)
        self.cache: KVCache = self.model.new_cache(nblocks, self.kv_dtype)
        self.alloc = BlockAllocator(nblocks)
        self.metrics = EngineMetrics(mcfg.name, clock=clock)
        self.waiting: Deque[Request] = deque()
        self.running: List[Request] = []
        self.prefilling: List[Request] = []        # admitted (blocks held), prompt not fully in the cache yet
        self._last_was_prefill = False
        self._lock = threading.Lock()
        self._next_id = 0
        self._wake = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()
        self.errors = 0
        self.last_error: Optional[str] = None
        # the last model launches, for /trace: (start, seconds, kind, sequences, tokens)
        self.step_log: Deque[tuple] = deque(maxlen=4096)
        self.step_seconds = {"prefill": 0.0, "mixed": 0.0, "decode": 0.0}
        self.step_counts = {"prefill": 0, "mixed": 0, "decode": 0}
        # decode: static device inputs (graph capture needs stable pointers) + pinned host mirrors
        B = self.cfg.max_num_seqs
        self.buckets = _buckets(B)
        dev = self.device
        self._d_tokens = torch.zeros(B, dtype=torch.int64, device=dev)
        self._d_i32 = torch.zeros((3, B), dtype=torch.int32, device=dev)      # positions, slots, ctx lens
        self._d_i32[1].fill_(-1)
        self._d_pos, self._d_slots, self._d_ctx = self._d_i32[0], self._d_i32[1], self._d_i32[2]
        self._d_bt = torch.zeros((B, self.max_blocks), dtype=torch.int32, device=dev)
        self._d_next = torch.zeros(B, dtype=torch.int64, device=dev)

        def host(*shape, dtype):
            t = torch.zeros(shape, dtype=dtype)
            return t.pin_memory() if self.on_gpu else t

        self._h_tokens_t = host(B, dtype=torch.int64)
        self._h_i32_t = host(3, B, dtype=torch.int32)
        self._h_bt_t = host(B, self.max_blocks, dtype=torch.int32)
        self._h_tokens, self._h_i32, self._h_bt = (self._h_tokens_t.numpy(), self._h_i32_t.numpy(),
                                                   self._h_bt_t.numpy())
        self._graphs: Dict[int, object] = {}
        self._workspace = llm.DecodeWorkspace(B, self.lcfg.n_q, 64, dev) if self.on_gpu else None

    # ------------------------------------------------------------------ sizing
    def _auto_blocks(self) -> int:
        per_block = KVCache.bytes_per_block(self.lcfg, self.kv_dtype or torch.bfloat16)
        if self.on_gpu:
            free, _total = torch.cuda.mem_get_info(self.device)
            n = int(free * self.cfg.kv_fraction) // per_block
        else:
            n = 256
        return max(n, self.max_blocks + 1)

    @property
    def kv_capacity_tokens(self) -> int:
        return self.alloc.total * BLOCK

    # ------------------------------------------------------------------ submission
    def submit(self, prompt: Sequence[int], max_tokens: int = 16) -> Request:
        now = self.clock()
        with self._lock:
            r = Request(self._next_id, [int(t) % self.mcfg.vocab for t in prompt], int(max_tokens), now)
            self._next_id += 1
            if not r.prompt or r.max_tokens < 1 or len(r.prompt) + 1 > self.max_len or \
                    -(-(len(r.prompt) + 1) // BLOCK) > self.alloc.total:
                r.error = "prompt empty or longer than max_model_len"
                r.t_done = now
                r.done.set()
                self.metrics.on_reject()
                return r
            r.max_tokens = min(r.max_tokens, self.max_len - len(r.prompt),
                               self.alloc.total * BLOCK - len(r.prompt))
            if self.tp.world > 1:
                if self.tp.rank != 0:
                    raise RuntimeError("tensor parallel: requests are submitted at rank 0")
                self._inbox.append(r)              # enters `waiting` on every rank at the next _tp_sync
            else:
                self.waiting.append(r)
            self.metrics.set_queue(len(self.running), len(self.waiting) + len(self._inbox), self.alloc.usage)
        self._wake.set()
        return r

    def idle(self) -> bool:
        with self._lock:
            return not self.waiting and not self.running and not self.prefilling

    # ------------------------------------------------------------------ tensor parallel lock-step
    def _tp_sync(self, stop: bool = False) -> bool:
        """Collective over the control group, once before every step: rank 0's new requests (and its wish to
        stop) reach every rank. Scheduling is a deterministic function of the queue contents and the sampled
        tokens are the same bits everywhere, so the ranks stay in step without exchanging scheduler output.
        Returns True when the engine is to stop."""
        if self.tp.world == 1:
            return stop
        # One small all-gather per step, nothing pickled: from EVERY rank whether its last step failed and a 63-bit hash of
        # its scheduler state, from rank 0 also its wish to stop and how many new requests it holds. Scheduling is a
        # deterministic function of that state, so equal hashes mean equal launches; a rank that failed (out of memory, a
        # bad launch, its own tp.check()) or drifted is seen by all ranks here, BEFORE the next collective is launched, and
        # all of them start clean together. Only a step that brings new requests pays for a second collective (their
        # prompts, as objects from rank 0). Before: an all_gather_object of the whole digest -- a pickle of one tuple per
        # running request and two gloo collectives -- on the critical path of every decode step.
        with self._lock:
            digest = hash((len(self.waiting), tuple(r.rid for r in self.prefilling), tuple((r.rid, len(r.out)) for r in self.running),
                           self.alloc.free, self.alloc.total)) & 0x7FFFFFFFFFFFFFFF
        failed, self._tp_reset = self._tp_reset, False
        new: List[Request] = []
        if self.tp.rank == 0:
            with self._lock:
                new, self._inbox = self._inbox, []
        msgs = self.tp.gather_ints((int(bool(failed)), digest, int(bool(stop)), len(new)))
        stop_all, n_new = bool(msgs[0][2]), int(msgs[0][3])
        new_specs = []
        if n_new:
            new_specs = self.tp.broadcast_object([(r.rid, r.prompt, r.max_tokens) for r in new] if self.tp.rank == 0 else None)
        bad = [i for i, m in enumerate(msgs) if m[0]]
        drift = [i for i, m in enumerate(msgs) if m[1] != msgs[0][1]]
        if bad or drift:
            why = (f"rank(s) {bad} failed a step" if bad else f"rank(s) {drift} scheduled differently from rank 0")
            self.tp_resets += 1
            self.last_error = self.last_error or why
            self._fail_all("tensor-parallel reset: " + why)
            if self.on_gpu:
                torch.cuda.synchronize(self.device)
            self.tp.resync()                    # the ranks launched different numbers of collectives in the failed step
        now = self.clock()
        with self._lock:
            # requests that arrived with this message were not part of the failed step: every rank takes them, reset or not
            if self.tp.rank == 0:
                self.waiting.extend(new)
            else:
                for rid, prompt, max_tokens in new_specs:
                    self.waiting.append(Request(rid, list(prompt), int(max_tokens), now))
        return bool(stop_all)

    def step_all(self) -> int:
        """`step()` for callers that drive a tensor-parallel engine by hand: every rank calls it the same
        number of times."""
        self._tp_sync()
        return self.step()

    # ------------------------------------------------------------------ one iteration
    def step(self) -> int:
        """Run one model launch; returns the number of tokens it produced (0 = nothing to do)."""
        chunks: List[tuple] = []                    # (request, tokens of it in this launch)
        riders: List[Request] = []                  # running sequences that decode inside a prefill launch
        mixed = self.cfg.mixed_batches
        with self._lock:
            decode_turn = not mixed and self._last_was_prefill and bool(self.running)
            if not decode_turn and (self.prefilling or self.waiting):
                budget = self.cfg.max_num_batched_tokens
                if mixed and self.running:
                    self._make_room()
                    riders = list(self.running)
                    budget = max(budget - len(riders), BLOCK)
                for r in self.prefilling:
                    take = min(r.n_tokens - r.prefilled, budget)
                    if take <= 0:
                        break
                    chunks.append((r, take))
                    budget -= take
                while budget > 0 and self.waiting and \
                        len(self.running) + len(self.prefilling) < self.cfg.max_num_seqs:
                    r = self.waiting[0]
                    n = r.n_tokens
                    need = -(-(n + 1) // BLOCK)
                    if need > self.alloc.free:
                        break
                    r.blocks = self.alloc.alloc(need)
                    r.prefilled = 0
                    self.waiting.popleft()
                    self.prefilling.append(r)
                    take = min(n, budget)
                    chunks.append((r, take))
                    budget -= take
        if chunks:
            self._last_was_prefill = True
            return self._prefill(chunks, riders) if riders else self._prefill(chunks)
        self._last_was_prefill = False
        if self.running:
            return self._decode()
        return 0

    def _finish(self, r: Request, now: float) -> None:
        r.t_done = now
        self.alloc.release(r.blocks)
        r.blocks = []
        self.metrics.on_finish(now - r.arrival)
        r.done.set()

    def _prefill(self, chunks: List[tuple], riders: Sequence[Request] = ()) -> int:
        """One launch over prompt chunks and, behind them, one decode token of each of `riders`: the projections see
        one batch of rows (the weights are streamed once for both), attention runs the prefill kernel on the chunk rows
        and the decode kernel on the riders' rows."""
        t0 = self.clock()
        toks: List[int] = []
        pos: List[int] = []
        slots: List[int] = []
        info, last_rows = [], []
        bt = np.zeros((len(chunks), self.max_blocks), dtype=np.int32)
        for i, (r, take) in enumerate(chunks):
            seq = r.prompt + r.out
            lo, hi = r.prefilled, r.prefilled + take
            info.append((len(toks), take, hi))                    # the chunk's queries see the hi tokens cached so far
            toks.extend(seq[lo:hi])
            pos.extend(range(lo, hi))
            blk = np.asarray(r.blocks, dtype=np.int64)
            idx = np.arange(lo, hi)
            slots.extend((blk[idx // BLOCK] * BLOCK + idx % BLOCK).tolist())
            bt[i, :len(r.blocks)] = r.blocks
            last_rows.append(len(toks) - 1)
        dev = self.device
        g = self.lcfg.n_q // self.lcfg.n_kv
        q_lens = [n for (_, n, _) in info]
        qt = llm.pick_q_tiles(q_lens, g, self.lcfg.n_kv)
        n_chunk_rows = len(toks)
        dmeta = None
        if riders:
            bt_d = np.zeros((len(riders), self.max_blocks), dtype=np.int32)
            ctx_d = []
            for j, r in enumerate(riders):
                nt = r.n_tokens
                toks.append(r.out[-1])
                pos.append(nt - 1)
                slots.append(r.blocks[(nt - 1) // BLOCK] * BLOCK + (nt - 1) % BLOCK)
                bt_d[j, :len(r.blocks)] = r.blocks
                ctx_d.append(nt)
                last_rows.append(n_chunk_rows + j)
            dmeta = DecodeMeta(torch.from_numpy(bt_d).to(dev), torch.tensor(ctx_d, dtype=torch.int32, device=dev),
                               num_splits=llm.pick_splits(len(riders), self.lcfg.n_kv, self.max_len) if self.on_gpu else 1,
                               workspace=self._workspace)
        meta = PrefillMeta(torch.from_numpy(bt).to(dev), torch.tensor(info, dtype=torch.int32, device=dev),
                           llm.prefill_tiles(q_lens, g, qt, ctx_lens=[c for (_, _, c) in info]).to(dev),
                           torch.tensor(last_rows, dtype=torch.int64, device=dev), qt)
        nxt = self.model.prefill(torch.tensor(toks, dtype=torch.int64, device=dev),
                                 torch.tensor(pos, dtype=torch.int32, device=dev),
                                 torch.tensor(slots, dtype=torch.int32, device=dev), self.cache, meta, riders=dmeta)
        nxt = nxt.tolist()                      # synchronises
        now = self.clock()
        new_prompt = produced = 0
        with self._lock:
            for (r, take), t in zip(chunks, nxt):
                if r.t_first is None:
                    new_prompt += take
                r.prefilled += take
                if r.prefilled < r.n_tokens:
                    continue                    # more chunks to go: the sampled token of a partial prompt is dropped
                self.prefilling.remove(r)
                produced += 1
                if r.t_first is None:
                    r.t_first = now
                    self.metrics.on_first_token(now - r.arrival)
                else:
                    self.metrics.on_token_gap(now - (r.t_last or now))
                r.out.append(int(t))
                r.t_last = now
                if len(r.out) >= r.max_tokens:
                    self._finish(r, now)
                else:
                    self.running.append(r)
            for r, t in zip(riders, nxt[len(chunks):]):
                self.metrics.on_token_gap(now - (r.t_last or now))
                r.out.append(int(t))
                r.t_last = now
                produced += 1
                if len(r.out) >= r.max_tokens or r.n_tokens >= self.max_len:
                    self.running.remove(r)
                    self._finish(r, now)
            self.metrics.on_step(produced, new_prompt, len(self.running) + len(self.prefilling), len(self.waiting),
                                 self.alloc.usage)
        kind = "mixed" if riders else "prefill"
        self.step_seconds[kind] = self.step_seconds.get(kind, 0.0) + now - t0
        self.step_counts[kind] = self.step_counts.get(kind, 0) + 1
        self.step_log.append((t0, now - t0, kind, len(chunks) + len(riders), len(toks)))
        return max(produced, 1)

    def _make_room(self) -> None:
        """Give every running sequence the block its next token needs, preempting from the back."""
        i = 0
        while i < len(self.running):
            r = self.running[i]
            need = -(-r.n_tokens // BLOCK)
            if len(r.blocks) >= need:
                i += 1
                continue
            if self.alloc.free == 0:
                # most recently admitted first: a request still in prefill loses the least work
                victim = self.prefilling.pop() if self.prefilling else self.running.pop()
                self.alloc.release(victim.blocks)
                victim.blocks = []
                victim.prefilled = 0
                victim.preempted += 1
                self.waiting.appendleft(victim)
                self.metrics.on_preempt()
                if victim is r:
                    break
                continue
            r.blocks.extend(self.alloc.alloc(1))
            i += 1

    def _decode(self) -> int:
        t0 = self.clock()
        with self._lock:
            self._make_room()
            seqs = list(self.running)
        n = len(seqs)
        if n == 0:
            return 0
        b = next(x for x in self.buckets if x >= n)
        ht, hv, hb = self._h_tokens, self._h_i32, self._h_bt
        ht[:b] = 0
        hv[:, :b] = 0
        hv[1, :b] = -1                                    # padded rows write no cache slot
        for i, r in enumerate(seqs):
            nt = r.n_tokens
            ht[i] = r.out[-1]
            hv[0, i] = nt - 1
            hv[1, i] = r.blocks[(nt - 1) // BLOCK] * BLOCK + (nt - 1) % BLOCK
            hv[2, i] = nt
            hb[i, :len(r.blocks)] = r.blocks
        # the pinned host buffers are not touched again before the .tolist() below has synchronised
        self._d_tokens[:b].copy_(self._h_tokens_t[:b], non_blocking=True)
        self._d_i32.copy_(self._h_i32_t, non_blocking=True)
        self._d_bt[:b].copy_(self._h_bt_t[:b], non_blocking=True)
        self._run_decode(b)
        nxt = self._d_next[:n].tolist()         # synchronises
        now = self.clock()
        with self._lock:
            still: List[Request] = []
            for r, t in zip(seqs, nxt):
                self.metrics.on_token_gap(now - (r.t_last or now))
                r.out.append(int(t))
                r.t_last = now
                if len(r.out) >= r.max_tokens or r.n_tokens >= self.max_len:
                    self._finish(r, now)
                else:
                    still.append(r)
            self.running = still
            self.metrics.on_step(n, 0, len(self.running) + len(self.prefilling), len(self.waiting), self.alloc.usage)
        self.step_seconds["decode"] += now - t0
        self.step_counts["decode"] += 1
        self.step_log.append((t0, now - t0, "decode", n, n))
        return n

    def _decode_call(self, b: int) -> None:
        meta = DecodeMeta(self._d_bt[:b], self._d_ctx[:b],
                          num_splits=llm.pick_splits(b, self.lcfg.n_kv, self.max_len) if self.on_gpu else 1,
                          workspace=self._workspace)
        out = self.model.decode(self._d_tokens[:b], self._d_pos[:b], self._d_slots[:b], self.cache, meta)
        self._d_next[:b].copy_(out)

    def _run_decode(self, b: int) -> None:
        if not (self.on_gpu and self.cfg.use_graphs):
            self._decode_call(b)
            return
        g = self._graphs.get(b)
        if g is None:
            # warm up on a side stream (library handles, allocator pools), then capture
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                self._decode_call(b)
            torch.cuda.current_stream(self.device).wait_stream(s)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_call(b)
            self._graphs[b] = g
        g.replay()

    def warmup(self, buckets: Optional[Sequence[int]] = None) -> None:
        """Capture the decode graphs ahead of traffic (padded rows only: no cache slot is written)."""
        if not (self.on_gpu and self.cfg.use_graphs):
            return
        self._d_i32.zero_()
        self._d_i32[1].fill_(-1)
        for b in (buckets or self.buckets):
            self._run_decode(b)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ background loop
    def start(self) -> "Engine":
        if self._thread is None:
            self._stop.clear()
            self._thread = threading.Thread(target=self._loop, name="klm-llm-engine", daemon=True)
            self._thread.start()
        return self

    def _loop(self) -> None:
        if self.on_gpu:
            torch.cuda.set_device(self.device)
        while True:
            if self.tp.world > 1:
                if self.tp.rank == 0 and not self._stop.is_set() and self.idle() and not self._inbox:
                    self._wake.wait(0.05)          # nothing to do: the other ranks wait in the control broadcast
                    self._wake.clear()
                    continue
                if self._tp_sync(self._stop.is_set()):
                    break
            elif self._stop.is_set():
                break
            try:
                n = self.step()
                if self.tp.world > 1 and n:
                    self.tp.check()     # a peer that missed a collective left this step's results invalid
            except Exception as e:      # a failed launch (out of memory, bad shape) must not strand the callers
                self.last_error = f"{type(e).__name__}: {e}"
                self.errors += 1
                self._fail_all(self.last_error)
                self._tp_reset = True   # tensor parallel: the next control exchange tells every rank (from ANY rank) to start clean
                n = 0
            if n == 0 and self.tp.world == 1:
                self._wake.wait(0.05)
                self._wake.clear()

    def follow(self) -> None:
        """Ranks other than 0 of a tensor-parallel engine: run the loop in this thread until rank 0 stops."""
        self._loop()

    def _fail_all(self, why: str) -> None:
        now = self.clock()
        with self._lock:
            for r in list(self._inbox) + list(self.waiting) + list(self.prefilling) + list(self.running):
                if r.blocks:
                    self.alloc.release(r.blocks)
                r.blocks = []
                r.error = why
                r.t_done = now
                r.done.set()
            self.waiting.clear()
            self._inbox = []
            self.running = []
            self.prefilling = []
            self.metrics.set_queue(0, 0, self.alloc.usage)

    def stop(self, abort_pending: bool = True) -> None:
        """Stop the background loop; requests still queued or running are failed (their `done` event
        fires with `error` set) so that no caller waits for a step that will never run."""
        self._stop.set()
        self._wake.set()
        if self._thread is not None:
            self._thread.join(timeout=30)
            self._thread = None
        if abort_pending:
            self._fail_all("engine stopped")

    def trace(self) -> Dict[str, object]:
        """The recent model launches as a Chrome trace (chrome://tracing, Perfetto): one track per kind,
        complete events with the batch shape as arguments."""
        ev = []
        for (t0, dt, kind, seqs, toks) in list(self.step_log):
            ev.append({"name": f"{kind} {seqs}x" if kind == "decode" else f"{kind} {toks} tok", "cat": kind, "ph": "X",
                       "ts": t0 * 1e6, "dur": dt * 1e6, "pid": 1, "tid": 2 if kind == "decode" else 1,
                       "args": {"sequences": seqs, "tokens": toks}})
        meta = [{"name": "thread_name", "ph": "M", "pid": 1, "tid": 1, "args": {"name": "prefill launches"}},
                {"name": "thread_name", "ph": "M", "pid": 1, "tid": 2, "args": {"name": "decode launches"}},
                {"name": "process_name", "ph": "M", "pid": 1, "args": {"name": f"klm engine {self.mcfg.name}"}}]
        return {"traceEvents": meta + ev, "displayTimeUnit": "ms"}

    def describe(self) -> Dict[str, object]:
        return {"model": self.mcfg.name, "params": self.mcfg.params(), "weight_bytes": self.model.weight_bytes(),
                "kv_blocks": self.alloc.total, "kv_capacity_tokens": self.kv_capacity_tokens,
                "kv_bytes": self.cache.nbytes, "max_num_seqs": self.cfg.max_num_seqs, "max_model_len": self.max_len,
                "graphs": sorted(self._graphs), "device": str(self.device),
                "decode_weights": "fp8 (W8A16, batch <= 128)" if self.model.fp8_decode else "bf16",
                "kv_cache_dtype": self.cfg.kv_cache_dtype, "mixed_batches": self.cfg.mixed_batches,
                # which kernels a launch of n rows runs on (rows outside these ranges: hipBLASLt + separate norm / rope / SwiGLU)
                "kernel_paths": {"skinny_rows": self.model.skinny_rows(), "mid_spans": [list(x) for x in self.model.mid_spans()],
                                 "fused_rope_attention": bool(self.model.fused_rope and self.on_gpu and self.kv_dtype is None)},
                "max_num_batched_tokens": self.cfg.max_num_batched_tokens,
                "step_seconds": dict(self.step_seconds), "step_counts": dict(self.step_counts),
                "tensor_parallel": self.tp.describe(), "errors": self.errors, "last_error": self.last_error}
