"""Llama-3 family decoder on the hand-written gfx950 kernels (random-initialised weights).

BASELINE.json config 4 names "vLLM-ROCm Llama-3-8B pod"; there is no vLLM, no checkpoint and no
network in the build environment, so the serving workload whose metrics the monitor scrapes is
this model of the same architecture with random weights of the same shapes: the GPU work per
token (weight bytes streamed, KV bytes read, FLOPs) is that of the real model, which is what a
telemetry check needs. No reference counterpart (vLLM is a README bullet, reference README.md:73).

MI355X-first choices:
  * 288 GB of HBM3E: 8B (16 GB) and 70B (141 GB) both fit ONE GPU in bf16, so replicas scale as
    one process per GPU (data parallel serving) and no tensor-parallel all-reduce sits on the
    token path; the KV cache gets the rest of the HBM. A model beyond that (llama3-405b) is sharded
    over a tensor-parallel group (parallel/tp.py): the same code with local shapes and two all-reduces
    per layer on the xGMI mesh kernels.
  * attention, RMSNorm(+residual), RoPE+cache write and SwiGLU are the kernels of ops/csrc/llm_*.hip.
    Decode is a weight-streaming problem up to ~128 rows, and four paths cover it (`skinny`, `mid_spans`,
    `_trunk` pick by row count, model width and weight format, all from measurements):
    the skinny MFMA kernel (batch <= 16: norm, residual add and SwiGLU fused in, RoPE + cache write inside the
    attention kernel: 4 weight passes + attention per layer), the mid-batch kernel (17 - 128 rows: x staged through
    LDS, split-K, epilogues that fuse the residual add + RMSNorm, SwiGLU and RoPE + cache write), the K-split
    256 x 256 MFMA GEMM behind the same epilogues (129 - 256 rows), and beyond that, as for long prefill, plain library
    GEMMs (hipBLASLt through torch.matmul), fused where the operands allow it (one qkv GEMM, one gate_up GEMM).
  * weight layout chosen for those kernels: RMSNorm gains are folded into the columns of the
    projection that consumes the norm, and gate/up rows are interleaved by 8 (ops.llm.interleave_gate_up).
  * decode is launch-bound at small batch (about 10 launches x layers), so the engine captures it
    in a hipGraph per batch bucket (serving/engine.py); everything here is capture-safe: no
    host synchronisation and no shape-dependent Python branching inside ``decode``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace
from typing import Dict, List, Optional

import torch

from ..ops import llm
from ..parallel.tp import SingleGroup

HEAD_DIM = llm.HEAD_DIM
BLOCK = llm.BLOCK


@dataclass(frozen=True)
class LlamaConfig:
    name: str
    vocab: int
    hidden: int
    layers: int
    n_q: int
    n_kv: int
    ffn: int
    rope_theta: float = 500000.0
    eps: float = 1e-5
    max_pos: int = 8192

    @property
    def q_dim(self) -> int:
        return self.n_q * HEAD_DIM

    @property
    def qkv_dim(self) -> int:
        return (self.n_q + 2 * self.n_kv) * HEAD_DIM

    def shard(self, world: int) -> "LlamaConfig":
        """The shapes one of `world` tensor-parallel ranks holds: heads, FFN columns and vocabulary rows are
        split, the hidden size (the residual stream, replicated) is not."""
        if world == 1:
            return self
        if self.n_q % world or self.n_kv % world or self.ffn % (8 * world) or self.vocab % world:
            raise ValueError(f"{self.name}: {self.n_q} q heads / {self.n_kv} kv heads / ffn {self.ffn} / vocab {self.vocab} "
                             f"do not split over {world} ranks")
        return replace(self, n_q=self.n_q // world, n_kv=self.n_kv // world, ffn=self.ffn // world, vocab=self.vocab // world)

    def params(self) -> int:
        per_layer = self.hidden * self.qkv_dim + self.q_dim * self.hidden + 3 * self.hidden * self.ffn + 2 * self.hidden
        return self.layers * per_layer + 2 * self.vocab * self.hidden + self.hidden

    def kv_bytes_per_token(self) -> int:
        return self.layers * 2 * self.n_kv * HEAD_DIM * 2

    def flops_per_token(self, ctx: int = 0) -> float:
        """Forward FLOPs of one token (GEMMs + attention over `ctx` keys), embedding lookup excluded."""
        gemm = 2.0 * (self.params() - self.vocab * self.hidden)
        attn = 4.0 * self.layers * self.n_q * HEAD_DIM * ctx
        return gemm + attn


PRESETS: Dict[str, LlamaConfig] = {
    "llama3-8b": LlamaConfig("llama3-8b", vocab=128256, hidden=4096, layers=32, n_q=32, n_kv=8, ffn=14336),
    "llama3-70b": LlamaConfig("llama3-70b", vocab=128256, hidden=8192, layers=80, n_q=64, n_kv=8, ffn=28672),
    # the same decoder block with other published shapes (GQA, head_dim 128, SwiGLU, no biases)
    # 810 GB in bf16: the one preset that does not fit a GPU and needs tensor parallelism (serve-llm --tp 4 / 8)
    "llama3-405b": LlamaConfig("llama3-405b", vocab=128256, hidden=16384, layers=126, n_q=128, n_kv=8, ffn=53248, max_pos=8192),
    "mistral-7b": LlamaConfig("mistral-7b", vocab=32768, hidden=4096, layers=32, n_q=32, n_kv=8, ffn=14336,
                              rope_theta=1000000.0, max_pos=32768),
    # same block structure, sized for CPU tests and smoke runs
    "tiny": LlamaConfig("tiny", vocab=512, hidden=256, layers=2, n_q=4, n_kv=2, ffn=512, max_pos=1024),
    "small": LlamaConfig("small", vocab=32000, hidden=2048, layers=4, n_q=16, n_kv=4, ffn=8192, max_pos=8192),
    # the 405B block at 4 layers: every tensor-parallel shape of the big model (16 q heads per kv head, hidden 16384)
    "wide": LlamaConfig("wide", vocab=32000, hidden=16384, layers=4, n_q=128, n_kv=8, ffn=53248, max_pos=8192),
}


class KVCache:
    """Paged cache of every layer: K [L, blocks, n_kv, 16, 128], V [L, blocks, n_kv, 128, 16]; zero
    initialised (the attention kernels read stale slots and multiply them by zero). dtype bf16 (or the
    model's dtype on CPU), or ``torch.float8_e4m3fn``: e4m3 bytes with unit scale, converted back to bf16
    fragments inside the attention kernels; half the bytes per token, twice the tokens per GB."""

    def __init__(self, cfg: LlamaConfig, num_blocks: int, device, dtype=torch.bfloat16):
        self.num_blocks = int(num_blocks)
        self.dtype = dtype
        self.k = torch.zeros((cfg.layers, num_blocks, cfg.n_kv, BLOCK, HEAD_DIM), dtype=dtype, device=device)
        self.v = torch.zeros((cfg.layers, num_blocks, cfg.n_kv, HEAD_DIM, BLOCK), dtype=dtype, device=device)

    @staticmethod
    def bytes_per_block(cfg: LlamaConfig, dtype=torch.bfloat16) -> int:
        return cfg.kv_bytes_per_token() * BLOCK * (1 if dtype == torch.float8_e4m3fn else 2) // 2

    @property
    def nbytes(self) -> int:
        return self.k.numel() * self.k.element_size() * 2


@dataclass
class DecodeMeta:
    block_tables: torch.Tensor          # int32 [B, max_blocks]
    ctx_lens: torch.Tensor              # int32 [B]  (context length INCLUDING the token being decoded)
    num_splits: int = 1
    workspace: Optional[llm.DecodeWorkspace] = None


@dataclass
class PrefillMeta:
    block_tables: torch.Tensor          # int32 [S, max_blocks]
    seq_info: torch.Tensor              # int32 [S, 3] = (q_start, q_len, ctx_len)
    tiles: torch.Tensor                 # int32 [n_tiles, 2], built for `q_tiles`
    last_rows: torch.Tensor             # int64 [S]: row of each sequence's last token
    q_tiles: int = llm.PREFILL_Q_TILES

    @staticmethod
    def build(cfg: "LlamaConfig", q_lens, block_tables, seq_info, last_rows, device) -> "PrefillMeta":
        """Chooses the attention tiling for these prompt lengths and builds the tile table."""
        g = cfg.n_q // cfg.n_kv
        qt = llm.pick_q_tiles(q_lens, g, cfg.n_kv)
        return PrefillMeta(block_tables, seq_info, llm.prefill_tiles(q_lens, g, qt).to(device), last_rows, qt)


class LlamaModel:
    WIDE_ROWS = 256      # largest row count an 8B-class model sends to the K-split tiled GEMM (see lib_projections)
    # rows from which `_stack`'s projections run on ops.gemm_bf16 (token rows padded to 256). 256: whole prefills of 256 .. 2048
    # tokens are 1.00 - 1.08x the hipBLASLt path since the 128 x 128 ring kernel took the few-tile shapes
    # (profiles/r3_llm/prefill_small_rows.json; 2048 before)
    OWN_GEMM_MIN_ROWS = 256

    """`tp`: a parallel.tp group. With more than one rank this instance holds ONE shard (Megatron layout: q/k/v
    heads, gate/up rows and vocabulary rows split by rank; wo and w_down split along their input, their partial
    products summed by an all-reduce of the residual-stream update) and `lcfg` are its local shapes."""

    def __init__(self, cfg: LlamaConfig, device="cuda", seed: int = 0, dtype=torch.bfloat16, tp=None):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.tp = tp if tp is not None else SingleGroup()
        self.lcfg = cfg.shard(self.tp.world)
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)

        def w(*shape, std=0.02):
            return torch.empty(shape, dtype=dtype, device=self.device).normal_(0.0, std, generator=g)

        self.embed = w(cfg.vocab, cfg.hidden)                      # replicated: the same on every rank
        if self.tp.world > 1:
            g.manual_seed(seed * 7919 + 1 + self.tp.rank)          # every rank draws its own shard
        cfg = self.lcfg
        self.lm_head = w(cfg.vocab, cfg.hidden)                    # final norm gain folded in
        # the norm gains live inside wqkv / w_gate_up / lm_head; the stand-alone norm kernel gets ones
        self.unit_gain = torch.ones(cfg.hidden, dtype=dtype, device=self.device)
        out_std = 0.02 / math.sqrt(2 * cfg.layers)
        self.layers: List[Dict[str, torch.Tensor]] = []
        for _ in range(cfg.layers):
            self.layers.append({
                "wqkv": w(cfg.qkv_dim, cfg.hidden),
                "wo": w(cfg.hidden, cfg.q_dim, std=out_std),
                "w_gate_up": w(2 * cfg.ffn, cfg.hidden),            # gate/up rows interleaved by 8
                "w_down": w(cfg.hidden, cfg.ffn, std=out_std),
            })
        self.cos_sin = llm.rope_table(cfg.max_pos, cfg.rope_theta, self.device)
        self.scale = 1.0 / math.sqrt(HEAD_DIM)
        self.fp8_decode = False
        self.skinny_limit: Optional[int] = None                    # overrides skinny_rows() (benchmarks)
        self.mid_limit: Optional[int] = None                       # overrides mid_spans(): rows 1 .. mid_limit
        self.wide_from: Optional[int] = None                       # rows from which the tiled GEMM also replaces the mid-batch kernel
        self.lib_proj: Optional[frozenset] = None                 # overrides lib_projections() (benchmarks)
        self.wide = True                                           # 129 - 256 rows on the K-split tiled GEMM (else hipBLASLt)
        # the library path's projections (`_stack`: prefill, batches beyond 256 rows) on this repo's persistent bf16 stream
        # from OWN_GEMM_MIN_ROWS rows on
        self.own_gemm = True
        self._pad: Dict[tuple, torch.Tensor] = {}                  # token-dimension padding scratch, per (rows, K)
        self.fused_rope = True                                     # skinny decode path: RoPE + cache write inside the attention kernel
        self.lm_head8: Optional[torch.Tensor] = None
        self.lm_head_s: Optional[torch.Tensor] = None

    _PROJ = ("wqkv", "wo", "w_gate_up", "w_down")

    def enable_fp8_decode(self) -> "LlamaModel":
        """W8A16 for the weight-streaming decode paths (skinny kernel up to 16 rows, mid-batch kernel up to 128): every projection gets an e4m3 copy
        with one scale per output row, stored in the skinny kernel's streaming order (ops.llm.tile_weights);
        activations stay bf16. The bf16 weights are replaced by the dequantised values, so the library-GEMM
        path (prefill, larger batches) and the fp8 path compute with the same weights. Costs half the
        model size in extra HBM (8 GB for 8B) and halves the bytes a decode step streams."""
        def q(w):
            w8, s = llm.quantize_fp8_rows(w)
            w.copy_(llm.dequantize_fp8_rows(w8, s, w.dtype))
            return llm.tile_weights(w8), s

        for L in self.layers:
            for name in self._PROJ:
                L[name + "8"], L[name + "_s"] = q(L[name])
        self.lm_head8, self.lm_head_s = q(self.lm_head)
        self.fp8_decode = True
        return self

    def prefill_gemm_backend(self, rows: int = 4096) -> str:
        """Which GEMMs carry the library path's projections (`_stack`: prefill, decode beyond 256 rows) at `rows` rows, for
        bench labels."""
        c = self.lcfg
        own = [n for n, (N, K) in (("wqkv", (c.qkv_dim, c.hidden)), ("wo", (c.hidden, c.q_dim)), ("w_gate_up", (2 * c.ffn, c.hidden)),
                                   ("w_down", (c.hidden, c.ffn))) if self._own(rows, N, K)]
        if len(own) == 4:
            return ("klm persistent bf16 MFMA stream (ops.gemm_bf16, token rows padded to 256) for wqkv / wo / w_gate_up / w_down; "
                    "the lm_head on the prompts' last rows on the skinny weight-streaming kernel (up to 16 rows; hipBLASLt beyond)")
        if own:
            return f"klm persistent bf16 MFMA stream for {' / '.join(own)}; hipBLASLt (torch.matmul) for the rest"
        return ("hipBLASLt (torch.matmul): NOT a hand-written-kernel figure; only attention, norms, rope and the KV write are "
                "this repo's kernels in prefill")

    def _own(self, rows: int, N: int, K: int) -> bool:
        return bool(self.own_gemm and self.device.type == "cuda" and self.dtype == torch.bfloat16 and rows >= self.OWN_GEMM_MIN_ROWS
                    and N % 256 == 0 and K % 128 == 0 and K >= 256)

    def _proj(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """``x[T, K] @ w[N, K]^T`` for the library path. From OWN_GEMM_MIN_ROWS rows on, and when N and K fit its tiles (every
        Llama-3 projection does: 6144 / 4096 / 28672 / 14336), on this repo's persistent MFMA stream: the token dimension is
        padded to whole 256-row tiles in a persistent scratch buffer (the extra rows hold stale values; their outputs are cut
        off), a last partial round of tiles is cut into K slices (ops.gemm_stream_plan). Else hipBLASLt through torch.matmul,
        and `prefill_gemm_backend` says so."""
        T, K = x.shape
        N = w.shape[0]
        if not self._own(T, N, K):
            return torch.matmul(x, w.t())
        from .. import ops
        Mp = (T + 255) // 256 * 256
        if Mp != T or not x.is_contiguous():
            buf = self._pad.get((Mp, K))
            if buf is None:
                buf = self._pad[(Mp, K)] = torch.zeros((Mp, K), dtype=x.dtype, device=x.device)
            buf[:T].copy_(x)
            x = buf
        out = torch.empty((Mp, N), dtype=x.dtype, device=x.device)
        ops.gemm_bf16(x, w, out=out)
        return out[:T]

    def weight_bytes(self) -> int:
        """bf16 bytes of the model (what a decode step streams without fp8 decode weights)."""
        n = self.embed.numel() + self.lm_head.numel()
        for L in self.layers:
            n += sum(L[k].numel() for k in self._PROJ)
        return n * 2

    def decode_stream_bytes(self, rows: int = 1) -> int:
        """Weight bytes one decode step of `rows` sequences reads (embedding rows excluded): fp8 copies where the
        path for that batch uses them (the lm_head is a library GEMM on the mid-batch path: bf16)."""
        layers = sum(L[k].numel() for L in self.layers for k in self._PROJ)
        head = self.lm_head.numel()
        if self.fp8_decode and self.skinny(rows):
            return layers + head
        if self.fp8_decode and self.mid(rows) and rows <= llm.MID_MAX_ROWS:
            return layers + 2 * head
        return 2 * (layers + head)

    def new_cache(self, num_blocks: int, dtype=None) -> KVCache:
        return KVCache(self.lcfg, num_blocks, self.device, dtype or self.dtype)

    # ------------------------------------------------------------------ the decoder stack
    def _stack(self, tokens, positions, slots, cache: KVCache, attend) -> torch.Tensor:
        """Library-GEMM path (prefill, decode at batch > 16) -> the final residual stream's norm."""
        cfg, g, tp = self.lcfg, self.unit_gain, self.tp
        h = self.embed[tokens]                                   # [T, hidden]
        residual = None
        for li, L in enumerate(self.layers):
            if residual is None:
                residual = h
                x = llm.rmsnorm(h, g, cfg.eps)
            else:
                x = llm.rmsnorm(h, g, cfg.eps, residual=residual)
            qkv = self._proj(x, L["wqkv"])
            llm.rope_cache(qkv, self.cos_sin, positions, slots, cache.k[li], cache.v[li], cfg.n_q, cfg.n_kv)
            a = attend(qkv, cache.k[li], cache.v[li])
            h = tp.all_reduce(self._proj(a, L["wo"]))
            x = llm.rmsnorm(h, g, cfg.eps, residual=residual)
            act = llm.silu_mul(self._proj(x, L["w_gate_up"]))
            h = tp.all_reduce(self._proj(act, L["w_down"]))
        return llm.rmsnorm(h, g, cfg.eps, residual=residual)

    def _stack_skinny(self, tokens, positions, slots, cache: KVCache, attend, rope: bool = True) -> torch.Tensor:
        """Batch <= 16: every projection is one pass of the skinny kernel over its weights, with the
        norm / residual add / SwiGLU fused in; `h` is the residual stream, updated in place.
        `rope=False`: `attend` takes the un-rotated qkv and does RoPE + the cache write itself."""
        cfg, tp = self.lcfg, self.tp
        f8 = self.fp8_decode
        # the residual add rides on rank 0's partial product; the all-reduce then yields the new residual stream
        res = (lambda h: {"epilogue": "residual", "residual": h}) if tp.rank == 0 else (lambda h: {})

        def lin(x, L, name, **kw):
            if f8:
                return llm.skinny_linear(x, L[name + "8"], weight_scale=L[name + "_s"], tiled=True, eps=cfg.eps, **kw)
            return llm.skinny_linear(x, L[name], eps=cfg.eps, **kw)

        h = self.embed[tokens]
        for li, L in enumerate(self.layers):
            qkv = lin(h, L, "wqkv", rms=True)
            if rope:
                llm.rope_cache(qkv, self.cos_sin, positions, slots, cache.k[li], cache.v[li], cfg.n_q, cfg.n_kv)
            a = attend(qkv, cache.k[li], cache.v[li])
            tp.all_reduce(lin(a, L, "wo", out=h, **res(h)))
            act = lin(h, L, "w_gate_up", rms=True, epilogue="swiglu")
            tp.all_reduce(lin(act, L, "w_down", out=h, **res(h)))
        return h

    def _stack_mid(self, tokens, positions, slots, cache: KVCache, attend) -> torch.Tensor:
        """Batches (or prompt chunks) of up to 128 rows: every projection on the mid-batch weight-streaming kernel
        (ops/csrc/llm_mid.hip), whose epilogue does what `_stack` needs a kernel of its own for: the residual add +
        RMSNorm after wo and w_down, SwiGLU after gate_up. Returns the final norm of the residual stream."""
        cfg, tp = self.lcfg, self.tp
        f8 = self.fp8_decode and tokens.shape[0] <= llm.MID_MAX_ROWS      # the e4m3 copies serve the weight-streaming kernel only
        residual = self.embed[tokens]
        x = llm.rmsnorm(residual, self.unit_gain, cfg.eps)

        tiled = self.wide_from is not None and tokens.shape[0] >= self.wide_from and not f8   # below 129 rows: by policy

        def lin(y, L, name, **kw):
            if f8:
                return llm.mid_linear(y, L[name + "8"], weight_scale=L[name + "_s"], eps=cfg.eps, **kw)
            return llm.mid_linear(y, L[name], eps=cfg.eps, wide=tiled, **kw)

        lib = self.lib_projections(tokens.shape[0]) if tiled or tokens.shape[0] > llm.MID_MAX_ROWS else frozenset()

        def close(y, L, name):
            """residual += y @ W^T (summed over the tensor-parallel ranks); -> its norm"""
            if name in lib:
                h = tp.all_reduce(torch.matmul(y, L[name].t()))
                return llm.rmsnorm(h, self.unit_gain, cfg.eps, residual=residual)
            if tp.world == 1:
                return lin(y, L, name, kind="res_rms", residual=residual)
            h = tp.all_reduce(lin(y, L, name))
            return llm.rmsnorm(h, self.unit_gain, cfg.eps, residual=residual)

        # above 64 rows the widest bf16 projection is the one shape where the library GEMM still wins (54 vs 63 us at
        # 128 rows of the 8B gate_up): x staging per 224-row block then rivals the weight stream
        # (from 129 rows on the K-split tiled GEMM takes every projection)
        wide_lib = 64 < tokens.shape[0] <= llm.MID_MAX_ROWS and not f8 and not tiled
        for li, L in enumerate(self.layers):
            # the kernel that sums the qkv projection's K splits also rotates q / k and fills the cache slots
            qkv = llm.mid_qkv_rope(x, L["wqkv8"] if f8 else L["wqkv"], self.cos_sin, positions, slots, cache.k[li], cache.v[li],
                                   cfg.n_q, cfg.n_kv, weight_scale=L["wqkv_s"] if f8 else None, wide=tiled)
            x = close(attend(qkv, cache.k[li], cache.v[li]), L, "wo")
            if wide_lib or "w_gate_up" in lib:
                act = llm.silu_mul(torch.matmul(x, L["w_gate_up"].t()))
            else:
                act = lin(x, L, "w_gate_up", kind="swiglu")
            x = close(act, L, "w_down")
        return x

    def mid_spans(self):
        """Row intervals [lo, hi] (beyond skinny_rows()) whose projections run on the mid-batch path: the weight-streaming
        kernel up to 128 rows, the K-split tiled GEMM above, both with fused epilogues; everything else is hipBLASLt +
        separate norm / rope / SwiGLU kernels. Measured per model class on one MI355X, whole decode step at ctx 1024:

        hidden <= 4096 (Llama-3-8B): ahead at every row count (15 - 23 % per layer at 24 - 128 rows,
            scripts/gpu_mid_bench.py; 9.9 vs 10.2 ms at 136 rows, 11.4 vs 13.4 at 192, 13.6 vs 13.9 at 256 with the
            projections of lib_projections() left on the library; scripts/exp/ab_wide.py, ab_wide_mix.py);
        hidden <= 8192 (Llama-3-70B): level up to 32 rows (31.8 vs 30.7 ms), ahead beyond: 31.0 vs 35.2 ms at 40 rows,
            36.8 vs 45.0 at 80, 43.5 vs 49.1 at 128, 45.1 vs 58.1 at 136, 58.2 vs 74.7 at 256 (scripts/exp/ab_wide_70b.py);
        wider (a llama3-405b / tp8 shard: narrow, deep projections): the library wins up to 96 rows (34.7 vs 36.7 ms at 64);
            ahead at 128 (45.9 vs 47.8), by 1 % at 136 - 192 and by 16 % at 256 (57.2 vs 66.3 ms) with wo left on the
            library (scripts/exp/ab_wide_405b_shard.py);
        fp8 decode weights: the weight-streaming kernel reads the e4m3 copies, up to its 128 rows."""
        if self.mid_limit is not None:
            return [(1, self.mid_limit)]
        h = self.cfg.hidden
        # past 128 rows the tiled GEMM reads the bf16 weights, whatever the decode weight format
        above = [] if not self.wide else [(129, self.WIDE_ROWS)] if h <= 4096 else [(129, llm.WIDE_MAX_ROWS)] if h <= 8192 \
            else [(129, llm.WIDE_MAX_ROWS)]
        if self.fp8_decode:
            return [(1, llm.MID_MAX_ROWS)] + above
        if h <= 4096:
            return [(1, llm.MID_MAX_ROWS)] + above
        if h <= 8192:
            return [(33, llm.MID_MAX_ROWS)] + above
        return [(97, llm.MID_MAX_ROWS)] + above

    def lib_projections(self, rows: int) -> frozenset:
        """Projections that stay on hipBLASLt inside the tiled-GEMM range (129 - 256 rows). Llama-3-8B, whole decode step
        (scripts/exp/ab_wide_mix.py): with gate_up (28672 x 4096: 112 tiles, two K splits, 59 MB of partial sums) on the
        library 12.06 -> 11.40 ms at 192 rows; with wo there as well 13.01 -> 12.50 ms at 224 and 14.25 -> 13.63 at 256
        (all library: 13.37 / 12.76 / 13.94)."""
        if self.lib_proj is not None:
            return self.lib_proj
        if self.cfg.hidden <= 4096:
            return frozenset(("w_gate_up", "wo")) if rows > 208 else frozenset(("w_gate_up",))
        if self.cfg.hidden <= 8192:
            # Llama-3-70B: gate_up on the library is worth 1 - 3 % up to 224 rows (49.7 -> 48.3 ms at 192); at 256 the tiled
            # GEMM with SwiGLU in its tile epilogue (224 tiles, no K split) is ahead (57.1 vs 58.2 ms). w_down is where the
            # library loses: 72.9 ms at 256 rows with it there too
            return frozenset(("w_gate_up",)) if rows <= 224 else frozenset()
        # a llama3-405b / tp8 shard (scripts/exp/ab_wide_405b_shard.py mix): wo (16384 x 2048) is the library's, gate_up is
        # not (68.6 ms at 256 rows with it there): 48.5 / 53.2 / 57.2 ms at 136 / 192 / 256 rows, all library 49.2 / 53.6 / 66.3
        return frozenset(("wo",))

    def mid(self, rows: int) -> bool:
        c = self.lcfg
        if rows > llm.MID_MAX_ROWS and (c.hidden % 256 or c.qkv_dim % 256 or (2 * c.ffn) % 256):
            return False                     # the tiled GEMM writes 256 output columns per block
        return (not self.skinny(rows) and any(lo <= rows <= hi for lo, hi in self.mid_spans()) and self.device.type == "cuda"
                and c.hidden % 512 == 0 and c.ffn % 512 == 0 and c.qkv_dim % 64 == 0 and c.q_dim % 512 == 0
                and (2 * c.ffn) % 64 == 0)

    def _trunk(self, tokens, positions, slots, cache: KVCache, attend) -> torch.Tensor:
        """The decoder stack on the path that fits the number of rows -> the final norm [T, hidden]."""
        if self.mid(tokens.shape[0]):
            return self._stack_mid(tokens, positions, slots, cache, attend)
        return self._stack(tokens, positions, slots, cache, attend)

    def skinny_rows(self) -> int:
        """Largest batch that decodes on the skinny kernel. Measured per model on MI355X (scripts/gpu_tp_shard_bench.py,
        profiles/r1_tp/): with 4096-wide rows it wins up to its 16 rows. With rows of 8192 and more (70B, 405B shards)
        its x fragments (M x K x 2 bytes per 16-row block, from L2) grow with K: on bf16 weights the library GEMMs win
        from 8 rows on (12 % at 8, 25 % at 16); on fp8 weights the mid-batch kernel takes over, from 5 rows for the
        70B shapes (16.9 vs 20.2 ms per step at 16) and from 9 for a 405B / tp8 shard (17.5 vs 19.5 ms at 8 in favour
        of this kernel, 24.7 vs 23.0 at 16 against it)."""
        if self.skinny_limit is not None:
            return self.skinny_limit
        if self.cfg.hidden <= 4096:
            return llm.SKINNY_MAX_ROWS
        return 8 if (self.fp8_decode and self.cfg.hidden > 8192) else 4

    def skinny(self, batch: int) -> bool:
        c = self.lcfg
        ok = batch <= self.skinny_rows() and c.hidden % 128 == 0 and c.ffn % 128 == 0 and c.vocab % 16 == 0
        return ok and (not self.fp8_decode or (c.hidden % 64 == 0 and c.ffn % 64 == 0))

    def decode(self, tokens, positions, slots, cache: KVCache, meta: DecodeMeta, logits: bool = False) -> torch.Tensor:
        """One token per sequence -> greedy next token ids [B] (int64), or the logits [B, vocab].
        `positions` are `meta.ctx_lens - 1` (the token being decoded is the last of its context). Capture-safe."""
        cfg = self.lcfg

        def attend(qkv, kc, vc):
            return llm.paged_attn_decode(qkv, kc, vc, meta.block_tables, meta.ctx_lens, self.scale, cfg.n_q,
                                         num_splits=meta.num_splits, workspace=meta.workspace)

        def attend_rope(qkv, kc, vc):
            # launch-bound regime: rotation and cache write ride inside the attention kernel (one launch less per layer)
            return llm.paged_attn_decode_rope(qkv, kc, vc, meta.block_tables, meta.ctx_lens, slots, self.cos_sin, self.scale,
                                              cfg.n_q, num_splits=meta.num_splits, workspace=meta.workspace)

        if self.skinny(tokens.shape[0]):
            fused = self.fused_rope and self.device.type == "cuda" and cache.k.dtype == torch.bfloat16
            h = self._stack_skinny(tokens, positions, slots, cache, attend_rope if fused else attend, rope=not fused)
            if self.fp8_decode:
                out = llm.skinny_linear(h, self.lm_head8, rms=True, eps=cfg.eps, weight_scale=self.lm_head_s, tiled=True)
            else:
                out = llm.skinny_linear(h, self.lm_head, rms=True, eps=cfg.eps)
        else:
            out = torch.matmul(self._trunk(tokens, positions, slots, cache, attend), self.lm_head.t())
        return self._sample(out, logits)

    def _sample(self, out: torch.Tensor, logits: bool) -> torch.Tensor:
        """Local logits [B, vocab / world] -> greedy token ids [B] (or the full logits [B, vocab])."""
        tp = self.tp
        if tp.world == 1:
            return out if logits else out.argmax(dim=-1)
        B, vl = out.shape
        if logits:
            return tp.all_gather(out.contiguous()).permute(1, 0, 2).reshape(B, tp.world * vl)
        # every rank offers its best (value, global id); ids fit fp32 exactly (< 2^24); rows padded to 16 bytes
        vals, idx = out.float().max(dim=-1)
        cand = torch.zeros((B + (B & 1), 2), dtype=torch.float32, device=out.device)
        cand[:B, 0] = vals
        cand[:B, 1] = (idx + tp.rank * vl).float()
        allc = tp.all_gather(cand)                                   # [world, B', 2]: the same bits on every rank
        best = allc[:, :B, 0].argmax(dim=0)
        return allc[:, :B, 1].gather(0, best[None])[0].long()

    def prefill(self, tokens, positions, slots, cache: KVCache, meta: PrefillMeta, logits: bool = False,
                riders: Optional[DecodeMeta] = None) -> torch.Tensor:
        """Variable-length batch of prompts (or prompt chunks) -> next token id of each row in `meta.last_rows`
        (or the logits of those rows). `riders`: the last ``riders.block_tables.shape[0]`` rows are one decode token each
        of running sequences (a mixed batch): they share the projections with the chunks and attend through the
        decode kernel."""
        cfg = self.lcfg
        n_ride = 0 if riders is None else riders.block_tables.shape[0]
        n_chunk = tokens.shape[0] - n_ride

        def attend(qkv, kc, vc):
            if not n_ride:
                return llm.paged_attn_prefill(qkv, kc, vc, meta.block_tables, meta.seq_info, meta.tiles, self.scale,
                                              cfg.n_q, q_tiles=meta.q_tiles)
            o = torch.empty((qkv.shape[0], cfg.q_dim), dtype=qkv.dtype, device=qkv.device)
            llm.paged_attn_prefill(qkv[:n_chunk], kc, vc, meta.block_tables, meta.seq_info, meta.tiles, self.scale,
                                   cfg.n_q, out=o[:n_chunk], q_tiles=meta.q_tiles)
            llm.paged_attn_decode(qkv[n_chunk:], kc, vc, riders.block_tables, riders.ctx_lens, self.scale, cfg.n_q,
                                  out=o[n_chunk:], num_splits=riders.num_splits, workspace=riders.workspace)
            return o

        x = self._trunk(tokens, positions, slots, cache, attend)
        return self._sample(self._head(x[meta.last_rows]), logits)

    def _head(self, x: torch.Tensor) -> torch.Tensor:
        """The lm_head on already-normed rows (prefill: the prompts' last rows). Up to 16 rows: one pass of the skinny
        weight-streaming kernel over the bf16 head (the decode path's kernel, without its fused norm); more rows: `_proj`."""
        c = self.lcfg
        if (self.own_gemm and x.is_cuda and x.dtype == torch.bfloat16 and 0 < x.shape[0] <= llm.SKINNY_MAX_ROWS
                and c.hidden % 128 == 0 and c.vocab % 16 == 0):
            return llm.skinny_linear(x.contiguous(), self.lm_head)
        return self._proj(x, self.lm_head)

    def load_shard_from(self, full: "LlamaModel") -> "LlamaModel":
        """Cut this rank's shard out of an unsharded instance (tests: the sharded model must compute what the
        whole one computes)."""
        c, l, r = full.cfg, self.lcfg, self.tp.rank
        if full.tp.world != 1 or c != self.cfg:
            raise ValueError("load_shard_from wants the unsharded model of the same config")
        D = HEAD_DIM
        q0, k0, v0 = 0, c.n_q * D, (c.n_q + c.n_kv) * D
        rows = torch.cat([torch.arange(q0 + r * l.n_q * D, q0 + (r + 1) * l.n_q * D),
                          torch.arange(k0 + r * l.n_kv * D, k0 + (r + 1) * l.n_kv * D),
                          torch.arange(v0 + r * l.n_kv * D, v0 + (r + 1) * l.n_kv * D)]).to(full.device)
        self.embed.copy_(full.embed)
        self.lm_head.copy_(full.lm_head[r * l.vocab:(r + 1) * l.vocab])
        for mine, theirs in zip(self.layers, full.layers):
            mine["wqkv"].copy_(theirs["wqkv"][rows])
            mine["wo"].copy_(theirs["wo"][:, r * l.q_dim:(r + 1) * l.q_dim])
            mine["w_gate_up"].copy_(theirs["w_gate_up"][2 * r * l.ffn:2 * (r + 1) * l.ffn])      # interleaved by 8: contiguous
            mine["w_down"].copy_(theirs["w_down"][:, r * l.ffn:(r + 1) * l.ffn])
        return self

    def load_from(self, other: "LlamaModel") -> "LlamaModel":
        """Copy another instance's weights (any device / dtype) into this one."""
        self.embed.copy_(other.embed)
        self.lm_head.copy_(other.lm_head)
        for mine, theirs in zip(self.layers, other.layers):
            for k in mine:
                mine[k].copy_(theirs[k])
        return self
