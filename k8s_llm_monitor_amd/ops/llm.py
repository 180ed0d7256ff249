"""LLM serving ops: RMSNorm, SwiGLU, RoPE + paged-cache write, paged attention (decode, prefill).

On a GPU every call runs the hand-written gfx950 kernel of ``csrc/llm_*.hip`` (ctypes, current
torch stream, no allocation inside the launcher, so the calls can be captured into a hipGraph);
a missing library is an error there. On CPU tensors the fp32 reference of ``ops.ref`` runs, which
is what lets the serving engine be tested without a GPU.

No reference counterpart (vLLM is a README bullet in the reference, README.md:73).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import KlmOpsError, _check, load
from . import _stream_of as _stream
from . import ref

HEAD_DIM = ref.HEAD_DIM
BLOCK = ref.BLOCK
rope_table = ref.rope_table

_bound = False


def _lib():
    global _bound
    lib = load()
    if not _bound:
        vp, i, f, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
        if not hasattr(lib, "klm_paged_attn_decode"):
            raise KlmOpsError("libklm_ops.so was built without the llm_*.hip kernels: run `python -m k8s_llm_monitor_amd.build --force`")
        lib.klm_rmsnorm.argtypes = [vp, vp, vp, vp, i, i, f, vp]
        lib.klm_silu_mul.argtypes = [vp, vp, i, i, vp]
        lib.klm_rope_cache.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i64, i, vp]
        lib.klm_paged_attn_decode.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, f, i, vp]
        lib.klm_paged_attn_decode_rope.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, f, i, vp]
        lib.klm_paged_attn_decode_rope.restype = i
        lib.klm_paged_attn_prefill.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, f, i, i, i, vp]
        lib.klm_skinny_linear.argtypes = [vp, i, vp, vp, vp, i, i, i, i, i, f, i, i, i, i, vp]
        lib.klm_skinny_linear_w8.argtypes = [vp, i, vp, vp, vp, vp, i, i, i, i, i, f, i, i, i, i, vp]
        lib.klm_skinny_linear_w8.restype = i
        ip = ctypes.POINTER(i)
        lib.klm_mid_linear_plan.argtypes = [i, i, i, ip, ip, ip]
        lib.klm_mid_linear.argtypes = [vp, i, vp, i, vp, i, i, i, i, i, i, vp]
        lib.klm_mid_linear_w8.argtypes = [vp, i, vp, vp, vp, i, i, i, i, i, i, vp]
        lib.klm_mid_epilogue.argtypes = [vp, i, i, i, i, vp, vp, f, vp]
        lib.klm_rope_cache_from_partials.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, i, i, i, i, i64, i, vp]
        lib.klm_mid_linear_swiglu.argtypes = [vp, i, vp, i, vp, vp, i, i, i, i, i, vp]
        lib.klm_gemm_bf16_splitk_plan.argtypes = [i, i, i, ip]
        lib.klm_gemm_bf16_splitk.argtypes = [vp, i, vp, i, vp, i, i, i, i, vp]
        lib.klm_gemm_bf16_splitk_plan.restype = lib.klm_gemm_bf16_splitk.restype = i
        lib.klm_gemm_bf16_swiglu.argtypes = [vp, i, vp, i, vp, i, i, i, vp]
        lib.klm_gemm_bf16_swiglu.restype = i
        for n in ("klm_mid_linear_plan", "klm_mid_linear", "klm_mid_linear_w8", "klm_mid_epilogue", "klm_rope_cache_from_partials",
                  "klm_mid_linear_swiglu"):
            getattr(lib, n).restype = i
        for n in ("klm_rmsnorm", "klm_silu_mul", "klm_rope_cache", "klm_paged_attn_decode", "klm_paged_attn_prefill",
                  "klm_skinny_linear"):
            getattr(lib, n).restype = i
        _bound = True
    return lib


def _bf16(*ts) -> None:
    for t in ts:
        if t.dtype != torch.bfloat16:
            raise KlmOpsError("llm ops need bfloat16 tensors")


def _i32(*ts) -> None:
    for t in ts:
        if t.dtype != torch.int32 or not t.is_contiguous() or not t.is_cuda:
            raise KlmOpsError("index tensors must be contiguous int32 on the GPU")


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-5,
            residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out = rmsnorm(x) * weight``; with `residual`: ``residual += x`` in place first and the
    norm is taken of the sum (the fused add of every transformer sub-layer)."""
    if not x.is_cuda:
        o, r = ref.rmsnorm(x, weight, eps, residual)
        if residual is not None:
            residual.copy_(r)
        if out is not None:
            out.copy_(o)
            return out
        return o
    _bf16(x, weight)
    if x.dim() != 2 or not x.is_contiguous() or weight.shape != (x.shape[1],):
        raise KlmOpsError(f"rmsnorm: x {tuple(x.shape)} / weight {tuple(weight.shape)}")
    if residual is not None and (residual.shape != x.shape or not residual.is_contiguous() or residual.dtype != x.dtype):
        raise KlmOpsError("rmsnorm: residual must match x")
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or not out.is_contiguous() or out.dtype != x.dtype:
        raise KlmOpsError("rmsnorm: bad out")
    T, H = x.shape
    rc = _lib().klm_rmsnorm(x.data_ptr(), residual.data_ptr() if residual is not None else None,
                            weight.data_ptr(), out.data_ptr(), T, H, float(eps), _stream(x))
    _check(rc, f"klm_rmsnorm(T={T},H={H})")
    return out


def silu_mul(gate_up: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SwiGLU on the fused gate_up activation (gate/up columns interleaved by 8, see
    interleave_gate_up): ``out[:, 8j+c] = silu(gate_up[:, 16j+c]) * gate_up[:, 16j+8+c]``."""
    if not gate_up.is_cuda:
        return ref.silu_mul(gate_up)
    _bf16(gate_up)
    if gate_up.dim() != 2 or not gate_up.is_contiguous() or gate_up.shape[1] % 16:
        raise KlmOpsError(f"silu_mul: gate_up {tuple(gate_up.shape)}")
    T, F2 = gate_up.shape
    if out is None:
        out = torch.empty((T, F2 // 2), dtype=gate_up.dtype, device=gate_up.device)
    _check(_lib().klm_silu_mul(gate_up.data_ptr(), out.data_ptr(), T, F2 // 2, _stream(gate_up)), "klm_silu_mul")
    return out


SKINNY_MAX_ROWS = 16
_EPILOGUES = {"none": 0, "residual": 1, "swiglu": 2}


def skinny_ok(M: int, N: int, K: int) -> bool:
    """Shapes the skinny kernel takes (decode at small batch)."""
    return 0 < M <= SKINNY_MAX_ROWS and N % 16 == 0 and K % 128 == 0


def interleave_gate_up(w_gate: torch.Tensor, w_up: torch.Tensor) -> torch.Tensor:
    """[F,K] gate and up weights -> the fused [2F,K] layout of this package: rows 16j..16j+7 are gate
    rows 8j..8j+7, rows 16j+8..16j+15 the matching up rows (see silu_mul / skinny_linear)."""
    F, K = w_gate.shape
    return torch.stack((w_gate.view(F // 8, 8, K), w_up.view(F // 8, 8, K)), dim=1).reshape(2 * F, K).contiguous()


FP8_MAX = 448.0      # largest finite OCP e4m3 value


def quantize_fp8_rows(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[N,K] weights -> (e4m3 [N,K], fp32 scale [N]) with ``w ~= w8.float() * scale[:, None]``
    (one scale per output row, chosen so that the row's largest magnitude maps to 448)."""
    wf = w.float()
    scale = (wf.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX)
    w8 = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    return w8.contiguous(), scale.to(torch.float32).contiguous()


def tile_weights(w: torch.Tensor) -> torch.Tensor:
    """[N,K] row-major (bf16 or e4m3) -> the skinny kernel's streaming order, same shape and dtype:
    tiles [N/16][K/kt] of 64 lanes x 16 bytes, lane l holding row 16 nt + l%16, k = kt ks + (l/16)*e .. +e
    (kt, e = 32, 8 for bf16; 64, 16 for fp8). Pass ``tiled=True`` to skinny_linear with the result."""
    N, K = w.shape
    e = 16 if w.dtype == torch.float8_e4m3fn else 8
    kt = 4 * e
    if N % 16 or K % kt:
        raise KlmOpsError(f"tile_weights: {tuple(w.shape)} is not a multiple of 16 x {kt}")
    v = w.view(torch.uint8 if e == 16 else w.dtype).view(N // 16, 16, K // kt, 4, e)     # [nt, fr, ks, fq, e]
    t = v.permute(0, 2, 3, 1, 4).contiguous().view(N, K)                                  # [nt, ks, fq, fr, e]
    return t.view(w.dtype)


def untile_weights(t: torch.Tensor) -> torch.Tensor:
    N, K = t.shape
    e = 16 if t.dtype == torch.float8_e4m3fn else 8
    kt = 4 * e
    v = t.view(torch.uint8 if e == 16 else t.dtype).view(N // 16, K // kt, 4, 16, e)
    return v.permute(0, 3, 1, 2, 4).contiguous().view(N, K).view(t.dtype)


def dequantize_fp8_rows(w8: torch.Tensor, scale: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    return (w8.float() * scale[:, None].float()).to(dtype)


def skinny_linear(x: torch.Tensor, weight: torch.Tensor, rms: bool = False, epilogue: str = "none",
                  residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  eps: float = 1e-5, waves: int = 0, row_tiles: int = 0,
                  weight_scale: Optional[torch.Tensor] = None, tiled: bool = False) -> torch.Tensor:
    """``out = x[M,K] @ weight[N,K].T`` for M <= 16, the weight streamed from HBM once.

    `weight` is bf16, or e4m3 with `weight_scale` [N] (see quantize_fp8_rows): half the bytes, converted
    to bf16 in registers (W8A16).
    rms: rows scaled by ``rsqrt(mean(x^2) + eps)`` (an RMSNorm whose gain is folded into `weight`);
    epilogue "residual": ``out = residual + product`` (`out` may be `residual`); "swiglu": `weight`
    rows are gate/up interleaved by 8 and ``out[M, N/2] = silu(gate) * up``."""
    epi = _EPILOGUES[epilogue]
    N, K = weight.shape
    M = x.shape[0]
    if (epi == 1) != (residual is not None):
        raise KlmOpsError("skinny_linear: `residual` goes with epilogue='residual'")
    w8 = weight.dtype == torch.float8_e4m3fn
    if w8 != (weight_scale is not None):
        raise KlmOpsError("skinny_linear: `weight_scale` goes with float8_e4m3fn weights")
    if not x.is_cuda:
        if tiled:
            weight = untile_weights(weight)
        xf = x[:, :K].float()
        if rms:
            xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
        o = xf @ weight.float().t()
        if w8:
            o = o * weight_scale.float()[None, :]
        if epi == 1:
            o = o + residual.float()
        elif epi == 2:
            o = ref.silu_mul(o)
        o = o.to(x.dtype)
        if out is not None:
            out.copy_(o)
            return out
        return o
    _bf16(x) if w8 else _bf16(x, weight)
    if w8 and (tuple(weight_scale.shape) != (N,) or weight_scale.dtype != torch.float32 or not weight_scale.is_contiguous()
               or not weight_scale.is_cuda):
        raise KlmOpsError("skinny_linear: weight_scale must be contiguous fp32 [N] on the GPU")
    # row-major weights may be a column slice of a wider allocation (rows further apart than K)
    w_ok = weight.is_contiguous() or (not tiled and weight.stride(1) == 1 and weight.stride(0) >= K)
    if x.dim() != 2 or x.stride(1) != 1 or x.shape[1] < K or not w_ok or not skinny_ok(M, N, K):
        raise KlmOpsError(f"skinny_linear: x {tuple(x.shape)} weight {tuple(weight.shape)}")
    n_out = N // 2 if epi == 2 else N
    if residual is not None and (tuple(residual.shape) != (M, N) or not residual.is_contiguous() or residual.dtype != x.dtype):
        raise KlmOpsError("skinny_linear: residual must be contiguous [M,N]")
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (M, n_out) or not out.is_contiguous() or out.dtype != x.dtype:
        raise KlmOpsError("skinny_linear: bad out")
    rp = residual.data_ptr() if residual is not None else None
    if w8:
        rc = _lib().klm_skinny_linear_w8(x.data_ptr(), x.stride(0), weight.data_ptr(), weight_scale.data_ptr(),
                                         out.data_ptr(), rp, M, N, K, int(bool(rms)), epi, float(eps), int(waves),
                                         int(row_tiles), int(bool(tiled)), int(weight.stride(0)), _stream(x))
    else:
        rc = _lib().klm_skinny_linear(x.data_ptr(), x.stride(0), weight.data_ptr(), out.data_ptr(), rp, M, N, K,
                                      int(bool(rms)), epi, float(eps), int(waves), int(row_tiles), int(bool(tiled)),
                                      int(weight.stride(0)), _stream(x))
    _check(rc, f"klm_skinny_linear(M={M},N={N},K={K},rms={rms},{epilogue})")
    return out


MID_MAX_ROWS = 128
WIDE_MAX_ROWS = 256      # 129 - 256 rows: the K-split 256 x 256 MFMA GEMM (gemm_bf16.hip) writes the same fp32 partials
_MID_KINDS = {"none": 0, "swiglu": 1, "res_rms": 2}


def mid_ok(M: int, N: int, K: int, fp8: bool = False) -> bool:
    """Shapes the mid-batch path takes: up to 128 rows on the weight-streaming kernel (decode at batch 17 - 128, short
    prefill chunks), 129 - 256 rows of bf16 weights on the K-split 256 x 256 GEMM."""
    if 0 < M <= MID_MAX_ROWS:
        return N % 64 == 0 and K % 512 == 0
    return MID_MAX_ROWS < M <= WIDE_MAX_ROWS and not fp8 and N % 256 == 0 and K % 512 == 0


def wide_plan(M: int, N: int, K: int, ksplit: int = 0) -> int:
    """K split of the 256 x 256 GEMM for these shapes (0: the launcher's choice)."""
    ks = ctypes.c_int(ksplit)
    _check(_lib().klm_gemm_bf16_splitk_plan(M, N, K, ctypes.byref(ks)), f"klm_gemm_bf16_splitk_plan({M},{N},{K})")
    return ks.value


def mid_plan(M: int, N: int, K: int, ksplit: int = 0, waves: int = 0, row_tiles: int = 0):
    """(ksplit, waves, row_tiles) the launcher uses; zeros are filled in by its heuristic."""
    ks, nw, rt = ctypes.c_int(ksplit), ctypes.c_int(waves), ctypes.c_int(row_tiles)
    _check(_lib().klm_mid_linear_plan(M, N, K, ctypes.byref(ks), ctypes.byref(nw), ctypes.byref(rt)), f"klm_mid_linear_plan({M},{N},{K})")
    return ks.value, nw.value, rt.value


def _wide(M: int, N: int, K: int, fp8: bool, wide: Optional[bool]) -> bool:
    """Does the K-split tiled GEMM produce the partial sums? Always past 128 rows; below only when asked and possible."""
    return M > MID_MAX_ROWS or bool(wide and not fp8 and N % 256 == 0 and K % 512 == 0)


def _mid_partials(x, weight, weight_scale, ksplit: int = 0, waves: int = 0, row_tiles: int = 0, wide: Optional[bool] = None):
    """The main kernel of the mid-batch path: fp32 partial sums [ksplit][M][N] (operands already checked)."""
    N, K = weight.shape
    M = x.shape[0]
    st = _stream(x)
    if _wide(M, N, K, weight_scale is not None, wide):   # compute-heavy side of the range: the tiled GEMM, same partial layout
        ks = wide_plan(M, N, K, ksplit)
        part = torch.empty((ks, M, N), dtype=torch.float32, device=x.device)
        _check(_lib().klm_gemm_bf16_splitk(x.data_ptr(), x.stride(0), weight.data_ptr(), weight.stride(0), part.data_ptr(),
                                           M, N, K, ks, st), f"klm_gemm_bf16_splitk(M={M},N={N},K={K},ks={ks})")
        return part, ks
    ks, nw, rt = mid_plan(M, N, K, ksplit, waves, row_tiles)
    part = torch.empty((ks, M, N), dtype=torch.float32, device=x.device)
    if weight_scale is not None:
        rc = _lib().klm_mid_linear_w8(x.data_ptr(), x.stride(0), weight.data_ptr(), weight_scale.data_ptr(), part.data_ptr(),
                                      M, N, K, ks, nw, rt, st)
    else:
        rc = _lib().klm_mid_linear(x.data_ptr(), x.stride(0), weight.data_ptr(), weight.stride(0), part.data_ptr(), M, N, K,
                                   ks, nw, rt, st)
    _check(rc, f"klm_mid_linear(M={M},N={N},K={K},ks={ks},nw={nw},rt={rt},fp8={weight_scale is not None})")
    return part, ks


def _mid_check(x, weight, weight_scale, what: str) -> None:
    N, K = weight.shape
    w8 = weight.dtype == torch.float8_e4m3fn
    if w8 != (weight_scale is not None):
        raise KlmOpsError(f"{what}: `weight_scale` goes with float8_e4m3fn weights (tiled)")
    _bf16(x) if w8 else _bf16(x, weight)
    if w8 and (not weight.is_contiguous() or tuple(weight_scale.shape) != (N,) or weight_scale.dtype != torch.float32
               or not weight_scale.is_contiguous() or not weight_scale.is_cuda):
        raise KlmOpsError(f"{what}: fp8 weights must be contiguous (tiled) with a contiguous fp32 scale [N] on the GPU")
    if x.dim() != 2 or x.stride(1) != 1 or x.shape[1] < K or weight.stride(1) != 1 or weight.stride(0) < K or \
            not mid_ok(x.shape[0], N, K, w8):
        raise KlmOpsError(f"{what}: x {tuple(x.shape)} weight {tuple(weight.shape)}")


def mid_qkv_rope(x: torch.Tensor, wqkv: torch.Tensor, cos_sin: torch.Tensor, positions: torch.Tensor, slots: torch.Tensor,
                 k_cache: torch.Tensor, v_cache: torch.Tensor, n_q: int, n_kv: int,
                 weight_scale: Optional[torch.Tensor] = None, wide: Optional[bool] = None) -> torch.Tensor:
    """``rope_cache(mid_linear(x, wqkv))`` in two launches instead of three: the kernel that sums the projection's K
    splits also rotates q and k and writes k / v into their cache slots. Bit-identical to the separate kernels."""
    T = x.shape[0]
    N = wqkv.shape[0]
    if not x.is_cuda:
        return rope_cache(mid_linear(x, wqkv, weight_scale=weight_scale), cos_sin, positions, slots, k_cache, v_cache, n_q, n_kv)
    nb = _check_caches(k_cache, v_cache, n_kv)
    if N != (n_q + 2 * n_kv) * HEAD_DIM or positions.shape != (T,) or slots.shape != (T,):
        raise KlmOpsError(f"mid_qkv_rope: wqkv {tuple(wqkv.shape)} for {n_q} + 2 x {n_kv} heads, T = {T}")
    if cos_sin.dim() != 2 or cos_sin.shape[1] != HEAD_DIM or cos_sin.dtype != torch.float32 or not cos_sin.is_contiguous():
        raise KlmOpsError("mid_qkv_rope: cos_sin must be fp32 [max_pos, 128]")
    _mid_check(x, wqkv, weight_scale, "mid_qkv_rope")
    _i32(positions, slots)
    part, ks = _mid_partials(x, wqkv, weight_scale, wide=wide)
    qkv = torch.empty((T, N), dtype=x.dtype, device=x.device)
    rc = _lib().klm_rope_cache_from_partials(part.data_ptr(), ks, qkv.data_ptr(), cos_sin.data_ptr(), positions.data_ptr(),
                                             slots.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), T, n_q, n_kv,
                                             cos_sin.shape[0], nb * BLOCK, _kv8(k_cache), _stream(x))
    _check(rc, f"klm_rope_cache_from_partials(T={T},ks={ks})")
    return qkv


def mid_linear(x: torch.Tensor, weight: torch.Tensor, kind: str = "none", residual: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None, eps: float = 1e-5, ksplit: int = 0, waves: int = 0,
               row_tiles: int = 0, weight_scale: Optional[torch.Tensor] = None, wide: Optional[bool] = None) -> torch.Tensor:
    """``x[M,K] @ weight[N,K].T`` for M <= 128 with the weight streamed from HBM once (llm_mid.hip), or for M <= 256 (and on
    request, `wide=True`, below 129 rows too) on the K-split 256 x 256 GEMM (gemm_bf16.hip, bf16 weights), finished by
    `kind`: "none" -> [M,N]; "swiglu" (gate/up rows interleaved by 8) -> [M,N/2]; "res_rms": `residual` [M,N] is
    updated in place with the product and its RMSNorm (unit gain) is returned.
    `weight` is bf16 row-major (rows may be further apart than K), or e4m3 in the tiled order of `tile_weights` with
    `weight_scale` [N] (W8A16: half the bytes)."""
    k = _MID_KINDS[kind]
    N, K = weight.shape
    M = x.shape[0]
    if (k == 2) != (residual is not None):
        raise KlmOpsError("mid_linear: `residual` goes with kind='res_rms'")
    w8 = weight.dtype == torch.float8_e4m3fn
    if w8 != (weight_scale is not None):
        raise KlmOpsError("mid_linear: `weight_scale` goes with float8_e4m3fn weights (tiled)")
    n_out = N // 2 if k == 1 else N
    if not x.is_cuda:
        wf = untile_weights(weight).float() * weight_scale.float()[:, None] if w8 else weight.float()
        o = x[:, :K].float() @ wf.t()
        if k == 1:
            o = ref.silu_mul(o)
        elif k == 2:
            o, nr = ref.rmsnorm(o.to(x.dtype), torch.ones(N, dtype=x.dtype), eps, residual=residual)
            residual.copy_(nr)
        o = o.to(x.dtype)
        if out is not None:
            out.copy_(o)
            return out
        return o
    _mid_check(x, weight, weight_scale, "mid_linear")
    if residual is not None and (tuple(residual.shape) != (M, N) or not residual.is_contiguous() or residual.dtype != x.dtype):
        raise KlmOpsError("mid_linear: residual must be contiguous [M,N]")
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (M, n_out) or not out.is_contiguous() or out.dtype != x.dtype:
        raise KlmOpsError("mid_linear: bad out")
    st = _stream(x)
    if k == 1 and _wide(M, N, K, w8, wide) and ksplit in (0, 1) and wide_plan(M, N, K, ksplit) == 1:
        # enough tiles without a K split (70B gate_up): SwiGLU in the tile epilogue, no partial sums
        _check(_lib().klm_gemm_bf16_swiglu(x.data_ptr(), x.stride(0), weight.data_ptr(), weight.stride(0), out.data_ptr(), M, N, K, st),
               f"klm_gemm_bf16_swiglu(M={M},N={N},K={K})")
        return out
    if k == 1 and not _wide(M, N, K, w8, wide):
        ks, nw, rt = mid_plan(M, N, K, ksplit, waves, row_tiles)
        if ks == 1:                       # no K split: the activation is applied inside the main kernel, no epilogue launch
            _check(_lib().klm_mid_linear_swiglu(x.data_ptr(), x.stride(0), weight.data_ptr(), weight.stride(0),
                                                weight_scale.data_ptr() if w8 else None, out.data_ptr(), M, N, K, nw, rt, st),
                   f"klm_mid_linear_swiglu(M={M},N={N},K={K},nw={nw},rt={rt},fp8={w8})")
            return out
    part, ks = _mid_partials(x, weight, weight_scale, ksplit, waves, row_tiles, wide=wide)
    _check(_lib().klm_mid_epilogue(part.data_ptr(), ks, M, N, k, out.data_ptr(),
                                   residual.data_ptr() if residual is not None else None, float(eps), st),
           f"klm_mid_epilogue(M={M},N={N},{kind})")
    return out


def _check_caches(k_cache: torch.Tensor, v_cache: torch.Tensor, n_kv: int) -> int:
    nb = k_cache.shape[0]
    if tuple(k_cache.shape) != (nb, n_kv, BLOCK, HEAD_DIM) or tuple(v_cache.shape) != (nb, n_kv, HEAD_DIM, BLOCK):
        raise KlmOpsError(f"cache shapes: K {tuple(k_cache.shape)} V {tuple(v_cache.shape)} for n_kv={n_kv}")
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()):
        raise KlmOpsError("caches must be contiguous")
    if k_cache.dtype != v_cache.dtype:
        raise KlmOpsError("K and V caches must have the same dtype")
    return nb


def _kv8(k_cache: torch.Tensor) -> int:
    """1 when the cache holds e4m3 bytes (fp8 KV cache, unit scale), 0 for bf16."""
    if k_cache.dtype == torch.float8_e4m3fn:
        return 1
    if k_cache.dtype != torch.bfloat16:
        raise KlmOpsError("the KV cache is bfloat16 or float8_e4m3fn")
    return 0


def rope_cache(qkv: torch.Tensor, cos_sin: torch.Tensor, positions: torch.Tensor, slots: torch.Tensor,
               k_cache: torch.Tensor, v_cache: torch.Tensor, n_q: int, n_kv: int) -> torch.Tensor:
    """RoPE on q and k of the fused qkv activation (in place) and k/v -> cache slot ``slots[t]``."""
    T = qkv.shape[0]
    nb = _check_caches(k_cache, v_cache, n_kv)
    if qkv.dim() != 2 or qkv.shape[1] != (n_q + 2 * n_kv) * HEAD_DIM or not qkv.is_contiguous():
        raise KlmOpsError(f"rope_cache: qkv {tuple(qkv.shape)}")
    if positions.shape != (T,) or slots.shape != (T,):
        raise KlmOpsError("rope_cache: positions / slots must be [T]")
    if cos_sin.dim() != 2 or cos_sin.shape[1] != HEAD_DIM or cos_sin.dtype != torch.float32 or not cos_sin.is_contiguous():
        raise KlmOpsError("rope_cache: cos_sin must be fp32 [max_pos, 128]")
    if not qkv.is_cuda:
        r = ref.rope(qkv, cos_sin, positions, n_q, n_kv)
        qkv.copy_(r)
        ref.cache_write(qkv, slots, k_cache, v_cache, n_q, n_kv)
        return qkv
    _bf16(qkv)
    _i32(positions, slots)
    rc = _lib().klm_rope_cache(qkv.data_ptr(), cos_sin.data_ptr(), positions.data_ptr(), slots.data_ptr(),
                               k_cache.data_ptr(), v_cache.data_ptr(), T, n_q, n_kv, cos_sin.shape[0],
                               nb * BLOCK, _kv8(k_cache), _stream(qkv))
    _check(rc, f"klm_rope_cache(T={T})")
    return qkv


def pick_splits(batch: int, n_kv: int, max_ctx: int, target_blocks: int = 256) -> int:
    """Flash-decoding split count: one block per CU is enough (each holds 4 waves x 16 KiB in flight) and
    every further split costs a block start and a partial result; at least 4 steps of 32 tokens each.
    Measured (scripts/gpu_attn_bench.py, ctx 1024): batch 1 -> 16-32 splits, 4 -> 8, 16 -> 2, 64+ -> 1."""
    by_fill = -(-target_blocks // max(1, batch * n_kv))
    by_len = max(1, max_ctx // 128)
    return int(max(1, min(by_fill, by_len, 64)))


class DecodeWorkspace:
    """Scratch of the split reduction, allocated once (hipGraph capture needs stable pointers)."""

    def __init__(self, max_batch: int, n_q: int, max_splits: int, device):
        self.max_batch, self.n_q, self.max_splits = max_batch, n_q, max_splits
        self.tmp_o = torch.empty((max_batch, n_q, max_splits, HEAD_DIM), dtype=torch.float32, device=device)
        self.tmp_ml = torch.empty((max_batch, n_q, max_splits, 2), dtype=torch.float32, device=device)


def paged_attn_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor,
                      ctx_lens: torch.Tensor, scale: float, n_q: int, out: Optional[torch.Tensor] = None,
                      num_splits: int = 1, workspace: Optional[DecodeWorkspace] = None) -> torch.Tensor:
    """One query token per sequence. `q` is [B, >= n_q*128] with unit inner stride (a view of the
    fused qkv activation is fine); returns [B, n_q*128]."""
    B = block_tables.shape[0]
    n_kv = k_cache.shape[1]
    nb = _check_caches(k_cache, v_cache, n_kv)
    if q.dim() != 2 or q.shape[0] != B or q.shape[1] < n_q * HEAD_DIM or q.stride(1) != 1:
        raise KlmOpsError(f"paged_attn_decode: q {tuple(q.shape)} strides {q.stride()}")
    if ctx_lens.shape != (B,):
        raise KlmOpsError("paged_attn_decode: ctx_lens must be [B]")
    if not q.is_cuda:
        o = ref.paged_attn_decode(q[:, : n_q * HEAD_DIM], k_cache, v_cache, block_tables, ctx_lens, scale)
        if out is not None:
            out.copy_(o)
            return out
        return o
    _bf16(q)
    _i32(block_tables, ctx_lens)
    if out is None:
        out = torch.empty((B, n_q * HEAD_DIM), dtype=q.dtype, device=q.device)
    elif tuple(out.shape) != (B, n_q * HEAD_DIM) or not out.is_contiguous():
        raise KlmOpsError("paged_attn_decode: bad out")
    tmp_o = tmp_ml = None
    if num_splits > 1:
        if workspace is None:
            workspace = DecodeWorkspace(B, n_q, num_splits, q.device)
        if workspace.max_batch < B or workspace.n_q != n_q or workspace.max_splits < num_splits:
            raise KlmOpsError("paged_attn_decode: workspace too small")
        tmp_o, tmp_ml = workspace.tmp_o.data_ptr(), workspace.tmp_ml.data_ptr()
    rc = _lib().klm_paged_attn_decode(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(),
                                      ctx_lens.data_ptr(), out.data_ptr(), tmp_o, tmp_ml, B, q.stride(0), n_q, n_kv,
                                      block_tables.shape[1], nb, int(num_splits), float(scale), _kv8(k_cache),
                                      _stream(q))
    _check(rc, f"klm_paged_attn_decode(B={B},splits={num_splits})")
    return out


def paged_attn_decode_rope(qkv: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor,
                           ctx_lens: torch.Tensor, slots: torch.Tensor, cos_sin: torch.Tensor, scale: float, n_q: int,
                           out: Optional[torch.Tensor] = None, num_splits: int = 1,
                           workspace: Optional[DecodeWorkspace] = None) -> torch.Tensor:
    """`rope_cache` + `paged_attn_decode` in one launch (bf16 cache, GPU): `qkv` [B, (n_q + 2 n_kv) * 128] is the UN-rotated
    projection; sequence b's token sits at position ctx_lens[b] - 1 and goes to cache slot slots[b] (< 0: padded row).
    q and the newest key are rotated in registers, key / value are written by the one wave that attends to them."""
    B = block_tables.shape[0]
    n_kv = k_cache.shape[1]
    nb = _check_caches(k_cache, v_cache, n_kv)
    if not qkv.is_cuda or _kv8(k_cache):
        raise KlmOpsError("paged_attn_decode_rope: GPU and a bf16 cache only (use rope_cache + paged_attn_decode)")
    if qkv.dim() != 2 or qkv.shape[0] != B or qkv.shape[1] != (n_q + 2 * n_kv) * HEAD_DIM or qkv.stride(1) != 1:
        raise KlmOpsError(f"paged_attn_decode_rope: qkv {tuple(qkv.shape)} strides {qkv.stride()}")
    if ctx_lens.shape != (B,) or slots.shape != (B,) or cos_sin.dim() != 2 or cos_sin.shape[1] != HEAD_DIM or \
            cos_sin.dtype != torch.float32 or not cos_sin.is_contiguous():
        raise KlmOpsError("paged_attn_decode_rope: ctx_lens [B], slots [B], cos_sin fp32 [max_pos, 128]")
    _bf16(qkv)
    _i32(block_tables, ctx_lens, slots)
    if out is None:
        out = torch.empty((B, n_q * HEAD_DIM), dtype=qkv.dtype, device=qkv.device)
    elif tuple(out.shape) != (B, n_q * HEAD_DIM) or not out.is_contiguous():
        raise KlmOpsError("paged_attn_decode_rope: bad out")
    tmp_o = tmp_ml = None
    if num_splits > 1:
        if workspace is None:
            workspace = DecodeWorkspace(B, n_q, num_splits, qkv.device)
        if workspace.max_batch < B or workspace.n_q != n_q or workspace.max_splits < num_splits:
            raise KlmOpsError("paged_attn_decode_rope: workspace too small")
        tmp_o, tmp_ml = workspace.tmp_o.data_ptr(), workspace.tmp_ml.data_ptr()
    rc = _lib().klm_paged_attn_decode_rope(qkv.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(),
                                           ctx_lens.data_ptr(), slots.data_ptr(), cos_sin.data_ptr(), out.data_ptr(), tmp_o, tmp_ml,
                                           B, qkv.stride(0), n_q, n_kv, block_tables.shape[1], nb, int(num_splits), float(scale),
                                           cos_sin.shape[0], _stream(qkv))
    _check(rc, f"klm_paged_attn_decode_rope(B={B},splits={num_splits})")
    return out


PREFILL_Q_TILES = 2      # 16-row query tiles per wave of the prefill kernel (see pick_q_tiles)
PREFILL_VARIANT = 1      # 1: K/V staged once per block through the LDS ring; 0: per-wave global loads


def pick_q_tiles(q_lens: Sequence[int], group: int, n_kv: int) -> int:
    """2 query tiles per wave (half the LDS reads per MFMA) once that still leaves every CU four
    blocks; 1 for shorter prompts, where the grid would not fill the chip (scripts/gpu_attn_bench.py,
    one sequence: 2048 tokens 80 us vs 93 us, 4096 tokens 257 us vs 211 us, 8192 tokens 927 us vs 743 us)."""
    per = 2 * 64 // group
    blocks = sum(-(-int(n) // per) for n in q_lens) * n_kv
    return 2 if blocks >= 1024 else 1


def prefill_tiles(q_lens: Sequence[int], group: int, q_tiles: int = PREFILL_Q_TILES,
                  ctx_lens: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Host-side tile table of the prefill kernel: int32 [n_tiles, 2] = (sequence, first query token);
    one tile = q_tiles * 64 / group query tokens. Tiles are ordered by the number of keys they attend
    to, longest first: causal attention makes the last tile of a 4096-token prompt 128x the work of
    the first, and in launch order the chip ran about 40 % empty (rocprofv3 SQ_WAVE_CYCLES) while the
    late tiles of the last head finished alone. `ctx_lens` (default: q_lens) gives the context
    length of each sequence when the queries are a chunk of a longer context."""
    per = q_tiles * 64 // group
    ctx = list(q_lens) if ctx_lens is None else list(ctx_lens)
    rows: List[Tuple[int, int, int]] = []
    for s, n in enumerate(q_lens):
        n = int(n)
        for t in range(0, n, per):
            rows.append((int(ctx[s]) - n + min(t + per, n), s, t))       # keys seen by the tile's last query
    rows.sort(key=lambda r: -r[0])
    return torch.tensor([(s, t) for (_, s, t) in rows], dtype=torch.int32).reshape(-1, 2)


def paged_attn_prefill(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor,
                       seq_info: torch.Tensor, tiles: torch.Tensor, scale: float, n_q: int,
                       out: Optional[torch.Tensor] = None, q_tiles: int = PREFILL_Q_TILES,
                       variant: int = PREFILL_VARIANT) -> torch.Tensor:
    """Causal attention of the last ``q_len`` tokens of each sequence over its paged context.
    q [T, >= n_q*128]; seq_info int32 [S, 3] = (q_start, q_len, ctx_len); tiles from prefill_tiles() built with the same `q_tiles`."""
    n_kv = k_cache.shape[1]
    nb = _check_caches(k_cache, v_cache, n_kv)
    S = seq_info.shape[0]
    if q.dim() != 2 or q.shape[1] < n_q * HEAD_DIM or q.stride(1) != 1:
        raise KlmOpsError(f"paged_attn_prefill: q {tuple(q.shape)}")
    if seq_info.shape != (S, 3) or block_tables.shape[0] != S or tiles.dim() != 2 or tiles.shape[1] != 2:
        raise KlmOpsError("paged_attn_prefill: seq_info [S,3], block_tables [S,*], tiles [n,2]")
    if not q.is_cuda:
        o = ref.paged_attn_prefill(q[:, : n_q * HEAD_DIM], k_cache, v_cache, block_tables, seq_info, scale)
        if out is not None:
            out.copy_(o)
            return out
        return o
    _bf16(q)
    _i32(block_tables, seq_info, tiles)
    if out is None:
        out = torch.zeros((q.shape[0], n_q * HEAD_DIM), dtype=q.dtype, device=q.device)
    rc = _lib().klm_paged_attn_prefill(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(),
                                       seq_info.data_ptr(), tiles.data_ptr(), out.data_ptr(), tiles.shape[0], S,
                                       q.stride(0), n_q, n_kv, block_tables.shape[1], nb, float(scale), int(q_tiles),
                                       int(variant), _kv8(k_cache), _stream(q))
    _check(rc, f"klm_paged_attn_prefill(tiles={tiles.shape[0]})")
    return out
