"""Plain PyTorch fp32 references of the LLM kernels in ``csrc/llm_*.hip``.

Two users: the numerics tests (``tests/test_gpu_llm.py`` compares every HIP kernel against
these on the same inputs) and the CPU path of ``ops.llm`` (a tiny model steps through the
serving engine in the CPU test suite). On a GPU the HIP kernels run, never these.

Cache layouts (see llm_attention.hip): K ``[blocks, n_kv, 16, 128]``, V ``[blocks, n_kv, 128, 16]``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

HEAD_DIM = 128
BLOCK = 16


def rope_table(max_pos: int, theta: float = 500000.0, device="cpu") -> torch.Tensor:
    """fp32 [max_pos, 128]: 64 cosines then 64 sines of pos * theta^(-2i/128)."""
    inv = 1.0 / (theta ** (torch.arange(0, HEAD_DIM, 2, dtype=torch.float64) / HEAD_DIM))
    ang = torch.arange(max_pos, dtype=torch.float64)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1).to(torch.float32).to(device)


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float,
            residual: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """-> (out, new_residual); both in x.dtype, math in fp32."""
    if residual is not None:
        h = (x.float() + residual.float()).to(x.dtype)
        new_res = h
    else:
        h, new_res = x, None
    hf = h.float()
    out = hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()
    return out.to(x.dtype), new_res


def silu_mul(gate_up: torch.Tensor) -> torch.Tensor:
    """gate/up columns interleaved by 8: [.., 16j+c] gate, [.., 16j+8+c] up -> [.., 8j+c]."""
    f = gate_up.shape[-1] // 2
    v = gate_up.float().reshape(*gate_up.shape[:-1], f // 8, 2, 8)
    return (torch.nn.functional.silu(v[..., 0, :]) * v[..., 1, :]).reshape(*gate_up.shape[:-1], f).to(gate_up.dtype)


def rope(qkv: torch.Tensor, cos_sin: torch.Tensor, positions: torch.Tensor, n_q: int, n_kv: int) -> torch.Tensor:
    """Rotate-half RoPE on the q and k heads of qkv [T, (n_q + 2 n_kv) * 128]; v untouched."""
    T = qkv.shape[0]
    x = qkv.float().view(T, n_q + 2 * n_kv, HEAD_DIM)
    cs = cos_sin[positions.long().clamp(0, cos_sin.shape[0] - 1)]
    c, s = cs[:, None, :64], cs[:, None, 64:]
    rot = x[:, : n_q + n_kv]
    x1, x2 = rot[..., :64], rot[..., 64:]
    out = x.clone()
    out[:, : n_q + n_kv, :64] = x1 * c - x2 * s
    out[:, : n_q + n_kv, 64:] = x2 * c + x1 * s
    return out.view(T, -1).to(qkv.dtype)


def cache_write(qkv_rot: torch.Tensor, slots: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                n_q: int, n_kv: int) -> None:
    T = qkv_rot.shape[0]
    x = qkv_rot.view(T, n_q + 2 * n_kv, HEAD_DIM)
    ok = (slots >= 0) & (slots < k_cache.shape[0] * BLOCK)
    idx = torch.nonzero(ok).flatten()
    if idx.numel() == 0:
        return
    sl = slots[idx].long()
    blk, off = sl // BLOCK, sl % BLOCK
    k_cache[blk, :, off, :] = x[idx, n_q:n_q + n_kv].to(k_cache.dtype)
    v_cache[blk, :, :, off] = x[idx, n_q + n_kv:].to(v_cache.dtype)


def gather_kv(k_cache: torch.Tensor, v_cache: torch.Tensor, table: torch.Tensor, ctx: int):
    """-> K, V as fp32 [n_kv, ctx, 128] for one sequence."""
    nb = (ctx + BLOCK - 1) // BLOCK
    ids = table[:nb].long()
    k = k_cache[ids].permute(1, 0, 2, 3).reshape(k_cache.shape[1], nb * BLOCK, HEAD_DIM)[:, :ctx]
    v = v_cache[ids].permute(1, 0, 3, 2).reshape(v_cache.shape[1], nb * BLOCK, HEAD_DIM)[:, :ctx]
    return k.float(), v.float()


def attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_pos: torch.Tensor, scale: float) -> torch.Tensor:
    """q [Tq, n_q, 128] (fp32), k/v [n_kv, ctx, 128], q_pos [Tq]: causal attention with GQA."""
    n_q, n_kv = q.shape[1], k.shape[0]
    g = n_q // n_kv
    kk = k.repeat_interleave(g, dim=0)
    vv = v.repeat_interleave(g, dim=0)
    s = torch.einsum("thd,hcd->htc", q, kk) * scale
    key_pos = torch.arange(k.shape[1], device=q.device)
    mask = key_pos[None, None, :] > q_pos.long()[None, :, None]
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("htc,hcd->thd", p, vv)


def paged_attn_decode(q, k_cache, v_cache, block_tables, ctx_lens, scale: float) -> torch.Tensor:
    """q [B, n_q*128 (+ more, strided view ok)] -> [B, n_q*128]."""
    B = block_tables.shape[0]
    n_kv = k_cache.shape[1]
    n_q = q.shape[-1] // HEAD_DIM
    out = torch.zeros((B, n_q * HEAD_DIM), dtype=q.dtype, device=q.device)
    for b in range(B):
        ctx = int(ctx_lens[b])
        if ctx <= 0:
            continue
        k, v = gather_kv(k_cache, v_cache, block_tables[b], ctx)
        qq = q[b].float().view(1, n_q, HEAD_DIM)
        o = attend(qq, k, v, torch.tensor([ctx - 1], device=q.device), scale)
        out[b] = o.reshape(-1).to(q.dtype)
    del n_kv
    return out


def paged_attn_prefill(q, k_cache, v_cache, block_tables, seq_info, scale: float) -> torch.Tensor:
    """q [T, n_q*128]; seq_info [S, 3] = (q_start, q_len, ctx_len) -> [T, n_q*128]."""
    n_q = q.shape[-1] // HEAD_DIM
    out = torch.zeros((q.shape[0], n_q * HEAD_DIM), dtype=q.dtype, device=q.device)
    for s in range(seq_info.shape[0]):
        q0, ql, ctx = (int(x) for x in seq_info[s])
        if ql <= 0 or ctx < ql:
            continue
        k, v = gather_kv(k_cache, v_cache, block_tables[s], ctx)
        qq = q[q0:q0 + ql].float().view(ql, n_q, HEAD_DIM)
        pos = torch.arange(ctx - ql, ctx, device=q.device)
        out[q0:q0 + ql] = attend(qq, k, v, pos, scale).reshape(ql, -1).to(q.dtype)
    return out
