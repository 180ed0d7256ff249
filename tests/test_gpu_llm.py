"""Numerics of the LLM serving kernels (ops/csrc/llm_*.hip) against the fp32 PyTorch references of
ops/ref.py on the same inputs. GPU only; the CPU dispatch of the same functions is covered by
tests/test_llm_cpu.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from k8s_llm_monitor_amd.ops import llm, ref  # noqa: E402

DEV = "cuda"


def _rand(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(torch.bfloat16)


def _close(a, b, atol, rtol=2e-2):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    assert bool((err <= bound).all()), f"max err {err.max().item():.4g} (at |ref| {b.abs().flatten()[err.argmax()].item():.4g})"


@pytest.mark.parametrize("T,H", [(37, 4096), (3, 256), (130, 8192), (9, 16384), (5, 12288)])
def test_rmsnorm_matches_reference(T, H):
    torch.manual_seed(1)
    x, w = _rand(T, H), _rand(H)
    o = llm.rmsnorm(x, w, 1e-5)
    r, _ = ref.rmsnorm(x, w, 1e-5)
    _close(o, r, 2e-2)
    res = _rand(T, H)
    res0 = res.clone()
    o2 = llm.rmsnorm(x, w, 1e-5, residual=res)
    r2, nr = ref.rmsnorm(x, w, 1e-5, residual=res0)
    assert torch.equal(res, nr)                       # the residual update is exact (one bf16 rounding)
    _close(o2, r2, 2e-2)


def test_silu_mul_matches_reference():
    torch.manual_seed(2)
    gu = _rand(33, 2 * 14336, scale=2.0)
    _close(llm.silu_mul(gu), ref.silu_mul(gu), 1e-2)


def _caches(nb, n_kv, fill=True):
    if fill:   # random everywhere: stale slots hold finite garbage, which the kernels must mask
        return _rand(nb, n_kv, 16, 128), _rand(nb, n_kv, 128, 16)
    z = torch.zeros
    return z(nb, n_kv, 16, 128, device=DEV, dtype=torch.bfloat16), z(nb, n_kv, 128, 16, device=DEV, dtype=torch.bfloat16)


def test_rope_and_cache_write_match_reference():
    torch.manual_seed(3)
    T, n_q, n_kv, nb = 50, 32, 8, 40
    qkv = _rand(T, (n_q + 2 * n_kv) * 128)
    cs = llm.rope_table(4096, 500000.0, DEV)
    pos = torch.randint(0, 4096, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(nb * 16, device=DEV)[:T].to(torch.int32)
    slots[7] = -1                                       # padding row: no cache write
    kc, vc = _caches(nb, n_kv, fill=False)
    kr, vr = kc.clone(), vc.clone()
    want = ref.rope(qkv, cs, pos, n_q, n_kv)
    ref.cache_write(want, slots, kr, vr, n_q, n_kv)
    got = qkv.clone()
    llm.rope_cache(got, cs, pos, slots, kc, vc, n_q, n_kv)
    _close(got, want, 1e-2, 1e-2)
    _close(kc, kr, 1e-2, 1e-2)
    assert torch.equal(vc, vr)                          # v is copied, bit for bit
    assert float(kc.float().abs().sum()) > 0


def _tables(ctx_lens, nb, max_blocks):
    perm = torch.randperm(nb)
    bt = torch.full((len(ctx_lens), max_blocks), -7, dtype=torch.int32)   # unused entries hold junk ids
    at = 0
    for i, c in enumerate(ctx_lens):
        n = (c + 15) // 16
        bt[i, :n] = perm[at:at + n].to(torch.int32)
        at += n
    return bt.to(DEV)


@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (4, 4), (16, 2), (16, 1)])
@pytest.mark.parametrize("splits", [1, 4, 0])
def test_paged_decode_attention_matches_reference(n_q, n_kv, splits):
    torch.manual_seed(4)
    ctx = [1, 15, 16, 17, 33, 700, 129, 0]
    nb, mb = 80, 48
    kc, vc = _caches(nb, n_kv)
    bt = _tables(ctx, nb, mb)
    cl = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    B = len(ctx)
    qkv = _rand(B, (n_q + 2 * n_kv) * 128)
    q = qkv[:, : n_q * 128]                              # strided view, as the model passes it
    scale = 1.0 / math.sqrt(128)
    ns = splits or llm.pick_splits(B, n_kv, mb * 16)
    got = llm.paged_attn_decode(q, kc, vc, bt, cl, scale, n_q, num_splits=ns)
    want = ref.paged_attn_decode(q, kc, vc, bt, cl, scale)
    _close(got, want, 2e-2)
    assert float(got[-1].float().abs().sum()) == 0.0     # ctx 0 -> zeros


@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (4, 4), (16, 1)])
@pytest.mark.parametrize("splits", [1, 4, 16, 0])
def test_decode_attention_with_rope_inside_equals_the_two_kernel_sequence(n_q, n_kv, splits):
    """rope_cache + paged_attn_decode against the one launch that rotates q and the newest key in registers and has the
    wave that attends to the newest token write its key / value: same output, the same bytes in the caches."""
    torch.manual_seed(40)
    ctx = [1, 15, 16, 17, 32, 33, 64, 700, 129, 0, 48, 113]          # newest token on every block / step boundary
    nb, mb = 120, 48
    kc, vc = _caches(nb, n_kv)
    bt = _tables(ctx, nb, mb)
    cl = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    B = len(ctx)
    slots = torch.tensor([bt[i, (c - 1) // 16].item() * 16 + (c - 1) % 16 if c else -1 for i, c in enumerate(ctx)],
                         dtype=torch.int32, device=DEV)
    pos = (cl - 1).clamp(min=0)
    qkv = _rand(B, (n_q + 2 * n_kv) * 128)
    cs = llm.rope_table(4096, 500000.0, DEV)
    scale = 1.0 / math.sqrt(128)
    ns = splits or llm.pick_splits(B, n_kv, mb * 16)
    k2, v2, two = kc.clone(), vc.clone(), qkv.clone()
    llm.rope_cache(two, cs, pos, slots, k2, v2, n_q, n_kv)
    want = llm.paged_attn_decode(two, k2, v2, bt, cl, scale, n_q, num_splits=ns)
    raw = qkv.clone()
    got = llm.paged_attn_decode_rope(raw, kc, vc, bt, cl, slots, cs, scale, n_q, num_splits=ns)
    assert torch.equal(raw, qkv)                                       # the activation is left as it came
    assert torch.equal(kc, k2) and torch.equal(vc, v2)
    _close(got, want, 1e-3, 1e-3)
    assert float((got.float() - want.float()).abs().max()) == 0.0      # in fact the same bits
    _close(got, ref.paged_attn_decode(two[:, : n_q * 128], k2, v2, bt, cl, scale), 2e-2)
    assert float(got[9].float().abs().sum()) == 0.0                    # ctx 0 -> zeros, nothing written
    with pytest.raises(llm.KlmOpsError):
        llm.paged_attn_decode_rope(raw, kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn), bt, cl, slots, cs, scale, n_q)


@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (2, 2), (8, 1)])
@pytest.mark.parametrize("q_tiles", [1, 2, 4])
@pytest.mark.parametrize("variant", [0, 1])
def test_paged_prefill_attention_matches_reference(n_q, n_kv, q_tiles, variant):
    torch.manual_seed(5)
    # (q_len, ctx_len): fresh prompts, and chunks whose earlier context is already cached
    seqs = [(5, 5), (64, 64), (100, 100), (17, 300), (1, 40), (33, 64)]
    nb, mb = 60, 24
    kc, vc = _caches(nb, n_kv)
    bt = _tables([c for _, c in seqs], nb, mb)
    T = sum(q for q, _ in seqs)
    info, at = [], 0
    for ql, c in seqs:
        info.append((at, ql, c))
        at += ql
    seq_info = torch.tensor(info, dtype=torch.int32, device=DEV)
    tiles = llm.prefill_tiles([q for q, _ in seqs], n_q // n_kv, q_tiles).to(DEV)
    qkv = _rand(T, (n_q + 2 * n_kv) * 128)
    q = qkv[:, : n_q * 128]
    scale = 1.0 / math.sqrt(128)
    got = llm.paged_attn_prefill(q, kc, vc, bt, seq_info, tiles, scale, n_q, q_tiles=q_tiles, variant=variant)
    want = ref.paged_attn_prefill(q, kc, vc, bt, seq_info, scale)
    _close(got, want, 2e-2)


def test_decode_attention_can_be_captured_in_a_hip_graph():
    torch.manual_seed(6)
    n_q, n_kv, nb, mb, B = 32, 8, 64, 16, 4
    kc, vc = _caches(nb, n_kv)
    bt = _tables([200, 50, 256, 3], nb, mb)
    cl = torch.tensor([200, 50, 256, 3], dtype=torch.int32, device=DEV)
    q = _rand(B, n_q * 128)
    out = torch.empty_like(q)
    ws = llm.DecodeWorkspace(B, n_q, 8, DEV)
    scale = 1.0 / math.sqrt(128)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        llm.paged_attn_decode(q, kc, vc, bt, cl, scale, n_q, out=out, num_splits=8, workspace=ws)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        llm.paged_attn_decode(q, kc, vc, bt, cl, scale, n_q, out=out, num_splits=8, workspace=ws)
    q.copy_(_rand(B, n_q * 128))
    cl.copy_(torch.tensor([100, 50, 256, 1], dtype=torch.int32, device=DEV))
    g.replay()
    torch.cuda.synchronize()
    _close(out, ref.paged_attn_decode(q, kc, vc, bt, cl, scale), 2e-2)


def _lin_ref(x, w, rms, epilogue, residual, eps=1e-5):
    xf = x.float()
    if rms:
        xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    o = xf @ w.float().t()
    if epilogue == "residual":
        o = o + residual.float()
    elif epilogue == "swiglu":
        o = ref.silu_mul(o)
    return o


@pytest.mark.parametrize("M", [1, 5, 16])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (256, 128), (128256, 4096)])
@pytest.mark.parametrize("rms,epilogue", [(False, "none"), (False, "residual"), (True, "none"), (True, "swiglu"),
                                          (True, "residual")])
@pytest.mark.parametrize("waves,row_tiles", [(0, 0), (16, 1), (4, 4)])
def test_skinny_linear_matches_reference(M, N, K, rms, epilogue, waves, row_tiles):
    if N == 128256 and (M != 5 or epilogue != "none" or not rms):
        pytest.skip("the lm_head shape is checked once")
    torch.manual_seed(7)
    w = _rand(N, K, scale=0.05)
    x = _rand(M, K)
    residual = _rand(M, N) if epilogue == "residual" else None
    want = _lin_ref(x, w, rms, epilogue, residual)
    got = llm.skinny_linear(x, w, rms=rms, epilogue=epilogue, residual=residual, waves=waves, row_tiles=row_tiles)
    _close(got, want, 5e-3 * float(want.abs().max()) + 1e-3, 2e-2)
    if residual is not None:                            # in place on the residual stream
        r2 = residual.clone()
        llm.skinny_linear(x, w, rms=rms, epilogue=epilogue, residual=r2, out=r2, waves=waves, row_tiles=row_tiles)
        assert torch.equal(r2, got)


def test_skinny_linear_reads_a_strided_input():
    torch.manual_seed(8)
    big = _rand(4, 3 * 4096)
    x = big[:, 4096:8192]                                # row stride 12288
    w = _rand(4096, 4096, scale=0.05)
    _close(llm.skinny_linear(x, w), _lin_ref(x, w, False, "none", None), 5e-2, 2e-2)


@pytest.mark.parametrize("M,rt", [(1, 1), (9, 4)])
def test_skinny_linear_reads_weights_with_a_row_stride(M, rt):
    """The weight may be a column slice of a wider allocation (w_stride > K): the columns beyond K are never read."""
    torch.manual_seed(18)
    N, K, PAD = 1024, 16384, 64
    big = torch.full((N, K + PAD), float("nan"), dtype=torch.bfloat16, device=DEV)
    w = big[:, :K]
    w.copy_(_rand(N, K, scale=0.03))
    x, res = _rand(M, K), _rand(M, N)
    want = _lin_ref(x, w.contiguous(), True, "residual", res)
    _close(llm.skinny_linear(x, w, rms=True, epilogue="residual", residual=res, row_tiles=rt), want, 5e-2, 2e-2)


def test_interleaved_gate_up_is_the_plain_swiglu():
    torch.manual_seed(9)
    F, K, T = 512, 256, 7
    wg, wu, x = _rand(F, K, scale=0.1), _rand(F, K, scale=0.1), _rand(T, K)
    w = llm.interleave_gate_up(wg, wu)
    want = torch.nn.functional.silu(x.float() @ wg.float().t()) * (x.float() @ wu.float().t())
    got = llm.silu_mul((x.float() @ w.float().t()).to(torch.bfloat16))
    _close(got, want, 3e-2, 2e-2)
    _close(llm.skinny_linear(x, w, epilogue="swiglu"), want, 3e-2, 2e-2)


@pytest.mark.parametrize("M", [1, 7, 16])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (256, 128), (28672, 4096)])
@pytest.mark.parametrize("rms,epilogue", [(False, "none"), (True, "swiglu"), (False, "residual")])
def test_skinny_linear_with_fp8_weights_matches_the_dequantised_reference(M, N, K, rms, epilogue):
    torch.manual_seed(10)
    w = _rand(N, K, scale=0.05)
    w8, ws = llm.quantize_fp8_rows(w)
    wd = w8.float() * ws[:, None]                       # what the kernel multiplies by, exactly
    assert float((wd - w.float()).abs().max()) < 0.07 * float(w.float().abs().max())   # e4m3: 3 mantissa bits
    x = _rand(M, K)
    residual = _rand(M, N) if epilogue == "residual" else None
    want = _lin_ref(x, wd, rms, epilogue, residual)
    got = llm.skinny_linear(x, w8, rms=rms, epilogue=epilogue, residual=residual, weight_scale=ws)
    _close(got, want, 5e-3 * float(want.abs().max()) + 1e-3, 2e-2)


def test_fp8_fragment_conversion_keeps_the_k_order():
    """One hot weight per row at a different k: any mix-up of the byte order inside the converted
    fragments would move the product to another column of x."""
    N, K = 16, 128
    w = torch.zeros(N, K, device=DEV)
    for n in range(N):
        w[n, (n * 37 + 5) % K] = 1.0
    w8, ws = llm.quantize_fp8_rows(w)
    x = torch.arange(K, device=DEV, dtype=torch.float32).remainder(97).to(torch.bfloat16)[None, :].contiguous()
    got = llm.skinny_linear(x, w8, weight_scale=ws).float()[0]
    want = torch.stack([x[0, (n * 37 + 5) % K].float() for n in range(N)])
    assert torch.allclose(got, want, rtol=1e-2, atol=1e-2), (got, want)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M,N,K,rt", [(1, 6144, 4096, 0), (9, 28672, 4096, 4), (16, 4096, 14336, 1), (3, 256, 128, 0)])
def test_skinny_linear_with_tiled_weights_equals_the_row_major_run(fp8, M, N, K, rt):
    torch.manual_seed(11)
    w = _rand(N, K, scale=0.05)
    x = _rand(M, K)
    ws = None
    if fp8:
        w, ws = llm.quantize_fp8_rows(w)
    t = llm.tile_weights(w)
    assert torch.equal(llm.untile_weights(t).view(torch.uint8), w.view(torch.uint8))
    a = llm.skinny_linear(x, w, rms=True, epilogue="swiglu", weight_scale=ws, row_tiles=rt)
    b = llm.skinny_linear(x, t, rms=True, epilogue="swiglu", weight_scale=ws, row_tiles=rt, tiled=True)
    assert torch.equal(a, b)                      # same products in the same order: bit-identical


# ---------------------------------------------------------------- fp8 (e4m3) KV cache
FP8 = torch.float8_e4m3fn


def test_rope_cache_writes_an_fp8_cache_like_torch_converts_it():
    torch.manual_seed(12)
    T, n_q, n_kv, nb = 40, 32, 8, 12
    qkv = _rand(T, (n_q + 2 * n_kv) * 128, scale=3.0)
    cs = llm.rope_table(2048, 500000.0, DEV)
    pos = torch.randint(0, 2048, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(nb * 16, device=DEV)[:T].to(torch.int32)
    kr, vr = _caches(nb, n_kv, fill=False)
    want = ref.rope(qkv, cs, pos, n_q, n_kv)
    ref.cache_write(want, slots, kr, vr, n_q, n_kv)
    k8 = torch.zeros(nb, n_kv, 16, 128, device=DEV, dtype=FP8)
    v8 = torch.zeros(nb, n_kv, 128, 16, device=DEV, dtype=FP8)
    got = qkv.clone()
    llm.rope_cache(got, cs, pos, slots, k8, v8, n_q, n_kv)
    _close(got, want, 1e-2, 1e-2)                                   # q and the in-place k stay bf16
    assert torch.equal(v8.view(torch.uint8), vr.to(FP8).view(torch.uint8))
    # k goes through the kernel's own fp32 rotation before the two roundings: allow the last fp8 bit
    dk = (k8.float() - kr.to(FP8).float()).abs()
    assert float((dk > 0).float().mean()) < 0.02 and bool((dk <= 0.13 * kr.float().abs() + 1e-3).all())


@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (4, 4), (16, 1)])
@pytest.mark.parametrize("splits", [1, 4])
def test_decode_attention_over_an_fp8_cache_matches_the_reference_on_the_same_values(n_q, n_kv, splits):
    torch.manual_seed(13)
    ctx = [1, 16, 17, 33, 700, 129, 0]
    nb, mb = 80, 48
    kc, vc = _caches(nb, n_kv)
    k8, v8 = kc.to(FP8), vc.to(FP8)
    bt = _tables(ctx, nb, mb)
    cl = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    qkv = _rand(len(ctx), (n_q + 2 * n_kv) * 128)
    q = qkv[:, : n_q * 128]
    scale = 1.0 / math.sqrt(128)
    got = llm.paged_attn_decode(q, k8, v8, bt, cl, scale, n_q, num_splits=splits)
    want = ref.paged_attn_decode(q, k8.to(torch.bfloat16), v8.to(torch.bfloat16), bt, cl, scale)
    _close(got, want, 2e-2)


@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (2, 2), (8, 1)])
@pytest.mark.parametrize("q_tiles", [1, 2])
def test_prefill_attention_over_an_fp8_cache_matches_the_reference_on_the_same_values(n_q, n_kv, q_tiles):
    torch.manual_seed(14)
    seqs = [(5, 5), (64, 64), (100, 100), (17, 300), (33, 64)]
    nb, mb = 60, 24
    kc, vc = _caches(nb, n_kv)
    k8, v8 = kc.to(FP8), vc.to(FP8)
    bt = _tables([c for _, c in seqs], nb, mb)
    info, at = [], 0
    for ql, c in seqs:
        info.append((at, ql, c))
        at += ql
    seq_info = torch.tensor(info, dtype=torch.int32, device=DEV)
    tiles = llm.prefill_tiles([q for q, _ in seqs], n_q // n_kv, q_tiles, ctx_lens=[c for _, c in seqs]).to(DEV)
    qkv = _rand(at, (n_q + 2 * n_kv) * 128)
    q = qkv[:, : n_q * 128]
    scale = 1.0 / math.sqrt(128)
    got = llm.paged_attn_prefill(q, k8, v8, bt, seq_info, tiles, scale, n_q, q_tiles=q_tiles)   # variant 0 is forced
    want = ref.paged_attn_prefill(q, k8.to(torch.bfloat16), v8.to(torch.bfloat16), bt, seq_info, scale)
    _close(got, want, 2e-2)


# ------------------------------------------------------------------ mid-batch linear (llm_mid.hip)
@pytest.mark.parametrize("M", [17, 48, 64, 100, 128])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (256, 512), (28672, 4096)])
def test_mid_linear_matches_reference(M, N, K):
    torch.manual_seed(31)
    x, w = _rand(M, K), _rand(N, K, scale=0.03)
    want = x.float() @ w.float().t()
    _close(llm.mid_linear(x, w), want, 5e-2, 2e-2)
    _close(llm.mid_linear(x, w, kind="swiglu"), ref.silu_mul(want), 5e-2, 2e-2)
    if N > 16384:                                        # res_rms keeps a row in one block's registers
        with pytest.raises(llm.KlmOpsError):
            llm.mid_linear(x, w, kind="res_rms", residual=_rand(M, N))
        return
    res = _rand(M, N)
    res0 = res.clone()
    got = llm.mid_linear(x, w, kind="res_rms", residual=res)
    new_res = (want + res0.float()).to(torch.bfloat16)
    _close(res, new_res, 1e-2, 2 ** -7)                  # one bf16 rounding of (fp32 product + residual); sum order differs
    hf = res.float()
    _close(got, hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5), 3e-2, 2e-2)


def test_tiled_gemm_with_swiglu_in_the_tile_epilogue():
    """Shapes with enough tiles for the chip without a K split (Llama-3-70B gate_up): no partial sums, the activation leaves
    the GEMM in bf16."""
    torch.manual_seed(35)
    for M, N, K in ((256, 57344, 1024), (150, 65536, 512), (129, 57344, 512)):
        assert llm.wide_plan(M, N, K) == 1
        x, w = _rand(M, K), _rand(N, K, scale=0.05)
        want = ref.silu_mul(x.float() @ w.float().t())
        guard = torch.full((M + 1, N // 2), 7.0, device=DEV, dtype=torch.bfloat16)
        got = llm.mid_linear(x, w, kind="swiglu", out=guard[:M])
        _close(got, want, 5e-2, 2e-2)
        assert bool((guard[M] == 7.0).all())
        _close(got, llm.mid_linear(x, w, kind="swiglu", ksplit=2), 2e-2, 2e-2)     # the partial-sum route


@pytest.mark.parametrize("M", [129, 200, 256])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (28672, 4096), (256, 512)])
def test_mid_linear_from_129_rows_runs_the_k_split_tiled_gemm(M, N, K):
    """129 - 256 rows: the 256 x 256 MFMA GEMM with a K split writes the fp32 partials, the mid-batch epilogues finish
    them (rows past M of the last tile are neither read out of bounds nor stored)."""
    torch.manual_seed(33)
    x, w = _rand(M, K), _rand(N, K, scale=0.03)
    assert llm.mid_ok(M, N, K) and not llm.mid_ok(M, N, K, fp8=True) and not llm.mid_ok(257, N, K)
    ks = llm.wide_plan(M, N, K)
    assert K % (ks * 64) == 0 and (N // 256) * ks <= 256
    want = x.float() @ w.float().t()
    guard = torch.full((M + 1, N), 7.0, device=DEV, dtype=torch.bfloat16)          # a row behind `out` must stay untouched
    got = llm.mid_linear(x, w, out=guard[:M])
    _close(got, want, 5e-2, 2e-2)
    assert bool((guard[M] == 7.0).all())
    _close(llm.mid_linear(x, w, kind="swiglu"), ref.silu_mul(want), 5e-2, 2e-2)
    for forced in (1, 2, 8):
        if K % (forced * 64) == 0 and K // forced >= 128:
            _close(llm.mid_linear(x, w, ksplit=forced), want, 5e-2, 2e-2)
    if N <= 16384:
        res = _rand(M, N)
        new_res = (res.float() + want).to(torch.bfloat16)
        r2 = res.clone()
        y = llm.mid_linear(x, w, kind="res_rms", residual=r2)
        _close(r2, new_res, 5e-2, 2e-2)
        _close(y, ref.rmsnorm(new_res, torch.ones(N, device=DEV, dtype=torch.bfloat16), 1e-5)[0], 5e-2, 2e-2)
    xs = _rand(M, K + 64)[:, :K]                                                # strided activation rows
    ws = _rand(N, K + 128, scale=0.03)[:, :K]
    _close(llm.mid_linear(xs, ws), xs.float() @ ws.float().t(), 5e-2, 2e-2)


@pytest.mark.parametrize("ksplit,waves,rt", [(1, 4, 1), (4, 8, 1), (8, 4, 2), (2, 8, 2), (8, 4, 1), (16, 8, 2)])
def test_mid_linear_every_configuration_gives_the_same_product(ksplit, waves, rt):
    torch.manual_seed(32)
    M, N, K = 72, 1024, 4096
    x, w = _rand(M, K), _rand(N, K, scale=0.03)
    assert llm.mid_plan(M, N, K, ksplit, waves, rt) == (ksplit, waves, rt)
    want = x.float() @ w.float().t()
    _close(llm.mid_linear(x, w, ksplit=ksplit, waves=waves, row_tiles=rt), want, 5e-2, 2e-2)
    # SwiGLU: inside the main kernel when there is no K split, in the epilogue kernel otherwise
    _close(llm.mid_linear(x, w, kind="swiglu", ksplit=ksplit, waves=waves, row_tiles=rt), ref.silu_mul(want), 5e-2, 2e-2)


@pytest.mark.parametrize("M", [20, 48, 64])
@pytest.mark.parametrize("fp8", [False, True])
def test_mid_linear_fused_swiglu_on_the_gate_up_shape(M, fp8):
    """N = 28672: 7-wave blocks, no K split up to 64 rows -> the activation never leaves the main kernel."""
    torch.manual_seed(34)
    N, K = 28672, 4096
    assert llm.mid_plan(M, N, K)[0] == 1 and llm.mid_plan(M, N, K)[1] == 7
    x, w = _rand(M, K), _rand(N, K, scale=0.03)
    if fp8:
        w8, sc = llm.quantize_fp8_rows(w)
        want = ref.silu_mul(x.float() @ (w8.float() * sc[:, None]).t())
        got = llm.mid_linear(x, llm.tile_weights(w8), weight_scale=sc, kind="swiglu")
    else:
        want = ref.silu_mul(x.float() @ w.float().t())
        got = llm.mid_linear(x, w, kind="swiglu")
    assert tuple(got.shape) == (M, N // 2)
    _close(got, want, 5e-2, 2e-2)


def test_mid_linear_reads_strided_operands_and_rejects_bad_shapes():
    torch.manual_seed(33)
    big = _rand(40, 3 * 1024)
    x = big[:, 1024:2048]
    wbig = _rand(512, 1024 + 64, scale=0.05)
    w = wbig[:, :1024]
    _close(llm.mid_linear(x, w), x.float() @ w.float().t(), 5e-2, 2e-2)
    with pytest.raises(llm.KlmOpsError):
        llm.mid_linear(_rand(257, 1024), w)                # beyond the tiled GEMM's range
    with pytest.raises(llm.KlmOpsError):
        llm.mid_linear(_rand(129, 1024), _rand(320, 1024)) # 129 rows and more need N % 256 == 0
    with pytest.raises(llm.KlmOpsError):
        llm.mid_linear(_rand(8, 1000), _rand(512, 1000))


@pytest.mark.parametrize("M", [20, 64, 128])
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 14336), (28672, 4096)])
def test_mid_linear_with_fp8_tiled_weights_matches_the_dequantised_reference(M, N, K):
    torch.manual_seed(41)
    x, w = _rand(M, K), _rand(N, K, scale=0.03)
    w8, sc = llm.quantize_fp8_rows(w)
    wd = llm.dequantize_fp8_rows(w8, sc, torch.bfloat16)
    want = x.float() @ (w8.float() * sc[:, None]).t()       # the kernel multiplies the e4m3 value by the fp32 scale: no bf16 rounding
    w8t = llm.tile_weights(w8)
    _close(llm.mid_linear(x, w8t, weight_scale=sc), want, 5e-2, 2e-2)
    _close(llm.mid_linear(x, w8t, weight_scale=sc, kind="swiglu"), ref.silu_mul(want), 5e-2, 2e-2)
    # the bf16 kernel on the dequantised (bf16-rounded) weights: the same sums up to that rounding
    _close(llm.mid_linear(x, w8t, weight_scale=sc), llm.mid_linear(x, wd), 5e-2, 1e-2)
    with pytest.raises(llm.KlmOpsError):
        llm.mid_linear(x, w8t)                          # fp8 without scales


@pytest.mark.parametrize("kv8", [False, True])
@pytest.mark.parametrize("T", [17, 64, 100, 200])
def test_mid_qkv_rope_equals_the_three_kernel_sequence(T, kv8):
    """mid_qkv_rope = mid_linear + rope_cache with the split-sum folded into the rope kernel: same bits everywhere."""
    torch.manual_seed(51)
    n_q, n_kv, K, nb = 8, 2, 1024, max(12, T // 16 + 2)
    N = (n_q + 2 * n_kv) * 128
    x, w = _rand(T, K), _rand(N, K, scale=0.05)
    cos_sin = llm.rope_table(512, 500000.0, DEV)
    pos = torch.randint(0, 512, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(nb * 16, device=DEV)[:T].to(torch.int32)
    slots[3] = -1                                       # a padded row writes no cache slot
    dt = torch.float8_e4m3fn if kv8 else torch.bfloat16
    caches = [(torch.zeros((nb, n_kv, 16, 128), dtype=dt, device=DEV), torch.zeros((nb, n_kv, 128, 16), dtype=dt, device=DEV))
              for _ in range(2)]
    a = llm.rope_cache(llm.mid_linear(x, w), cos_sin, pos, slots, *caches[0], n_q, n_kv)
    b = llm.mid_qkv_rope(x, w, cos_sin, pos, slots, *caches[1], n_q, n_kv)
    assert torch.equal(a, b)
    for ca, cb in zip(caches[0], caches[1]):
        assert torch.equal(ca.view(torch.uint8), cb.view(torch.uint8))
