"""Numerics of the hand-written gfx950 kernels against plain PyTorch fp32 references.
Runs on a real MI355X only (``-m gpu``). The library is loaded with build_if_missing=False:
a missing ``libklm_ops.so`` is a failure here, never a silent fallback."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from k8s_llm_monitor_amd import ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ops.load(build_if_missing=False)
    return torch, ops, torch.device("cuda:0")


@pytest.mark.parametrize("variant,shape", [(0, (128, 128, 64)), (0, (256, 384, 192)), (0, (1024, 512, 2048)),
                                           (1, (256, 256, 128)), (1, (512, 768, 320)), (1, (2048, 1024, 4096)),
                                           (2, (256, 256, 256)), (2, (512, 768, 384)), (2, (2048, 1024, 4096)), (2, (256, 512, 4224)),
                                           (3, (256, 256, 256)), (3, (512, 768, 384)), (3, (1024, 512, 4096)),
                                           (4, (256, 256, 256)), (4, (512, 768, 384)), (4, (2048, 1024, 4096)), (4, (256, 512, 4352)),
                                           (10, (512, 768, 384)), (11, (512, 768, 1024)), (12, (512, 768, 1024)), (13, (256, 256, 4352)), (11, (1280, 1536, 2048)),
                                           (8, (256, 256, 256)), (8, (512, 768, 384)), (8, (2048, 1024, 4096)), (8, (256, 512, 4352)),
                                           (9, (512, 768, 384)), (9, (2048, 1024, 4096)),
                                           (14, (128, 128, 64)), (14, (256, 384, 128)), (14, (384, 640, 192)), (14, (1024, 512, 2048)), (14, (128, 256, 4160)),
                                           (16, (256, 128, 64)), (16, (512, 384, 192)), (16, (1024, 640, 2048)), (17, (128, 256, 128)), (17, (384, 512, 320)), (17, (640, 1024, 2112)),
                                           (None, (2560, 2560, 512)),
                                           (None, (1024, 1024, 1024)), (None, (2048, 2048, 2048)), (None, (384, 128, 64)), (None, (256, 256, 192))])
def test_gemm_bf16_matches_fp32_reference(env, variant, shape):
    torch, ops, dev = env
    M, N, K = shape
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    c = ops.gemm_bf16(a, b, variant=variant)
    ref = a.float() @ b.float().t()
    # bf16 output rounding: relative 2^-8 of the value, plus accumulation-order noise
    torch.testing.assert_close(c.float(), ref, rtol=1.6e-2, atol=1e-2 * K ** 0.5)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13, 14, 16, 17])
def test_gemm_bf16_exact_on_small_integers_with_asymmetric_operands(env, variant):
    torch, ops, dev = env
    M, N, K = 512, 256, (192 if variant < 2 else 384)
    g = torch.Generator(device=dev).manual_seed(7)
    a = torch.randint(-3, 4, (M, K), device=dev, generator=g).to(torch.bfloat16)
    b = torch.randint(-3, 4, (N, K), device=dev, generator=g).to(torch.bfloat16)
    b[:, 0] = torch.arange(N, device=dev).to(torch.bfloat16) % 5          # row-dependent: catches a transposed C
    c = ops.gemm_bf16(a, b, variant=variant)
    ref = (a.float() @ b.float().t()).to(torch.bfloat16)
    assert torch.equal(c, ref)


def test_gemm_bf16_identity_picks_rows_not_columns(env):
    torch, ops, dev = env
    n = 256
    eye = torch.eye(n, device=dev, dtype=torch.bfloat16)
    b = (torch.arange(n * n, device=dev).reshape(n, n) % 251).to(torch.bfloat16)      # asymmetric
    for v in (0, 1, 2, 3, 4, 8, 10, 13, 14, 16, 17):
        assert torch.equal(ops.gemm_bf16(eye, b, variant=v), b.t().contiguous())      # C = I @ B^T


def test_gemm_bf16_is_deterministic_run_to_run(env):
    torch, ops, dev = env
    a = torch.randn(1024, 2048, device=dev, dtype=torch.bfloat16)
    b = torch.randn(1024, 2048, device=dev, dtype=torch.bfloat16)
    c0 = ops.gemm_bf16(a, b, variant=1).clone()
    for v in (1, 2, 3, 4, 8, 9, 13):                                  # every pipeline gives the same bits, every run
        for _ in range(20):
            assert torch.equal(ops.gemm_bf16(a, b, variant=v), c0)   # a ring-buffer race would show as a rare diff
    # the eight-wave 128 x 128 ring (its own K order: compared with its own first run), on K tile counts 1..5 and a long one
    for K in (64, 128, 192, 256, 320, 2048):
        x, y = a[:, :K].contiguous(), b[:, :K].contiguous()
        c14 = ops.gemm_bf16(x, y, variant=14).clone()
        torch.testing.assert_close(c14.float(), x.float() @ y.float().t(), rtol=1.6e-2, atol=1e-2 * K ** 0.5)
        for _ in range(20):
            assert torch.equal(ops.gemm_bf16(x, y, variant=14), c14), K
        # its 256 x 128 / 128 x 256 forms (three-stage ring) sum in the same order: the same bits
        for _ in range(10):
            assert torch.equal(ops.gemm_bf16(x, y, variant=16), c14) and torch.equal(ops.gemm_bf16(x, y, variant=17), c14), K


@pytest.mark.parametrize("shape", [(256, 256, 256), (512, 768, 384), (1024, 2048, 4096), (256, 512, 512), (768, 256, 4608),
                                   (128, 128, 128), (384, 640, 640), (128, 256, 4224), (2048, 2048, 2048)])
def test_gemm_fp8_matches_fp32_reference(env, shape):
    torch, ops, dev = env
    M, N, K = shape
    torch.manual_seed(K)
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.float8_e4m3fn)
    b = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.float8_e4m3fn)
    c = ops.gemm_fp8(a, b)                               # the hand-scheduled stream where K allows, else the ring kernel
    ref = a.float() @ b.float().t()                      # the fp8 VALUES are exact in fp32
    torch.testing.assert_close(c.float(), ref, rtol=1.6e-2, atol=2e-3 * K ** 0.5)
    # every pipeline sums an element's products in the same order: same bits (14: the eight-wave 128 x 128 ring kernel, the
    # default for shapes of few tiles; K tile counts 1, 3, 5, 33 and even ones)
    assert torch.equal(c, ops.gemm_fp8(a, b, variant=14))
    if N % 256 == 0:
        assert torch.equal(c, ops.gemm_fp8(a, b, variant=17))              # ... and its 128 x 256 / 256 x 128 forms
    if M % 256 == 0:
        assert torch.equal(c, ops.gemm_fp8(a, b, variant=16))
    if M % 256 or N % 256:
        return
    assert torch.equal(c, ops.gemm_fp8(a, b, variant=0)) and torch.equal(c, ops.gemm_fp8(a, b, variant=1))
    if K % 256 == 0 and K >= 512:
        assert torch.equal(c, ops.gemm_fp8(a, b, variant=2)) and torch.equal(c, ops.gemm_fp8(a, b, variant=7))
        assert torch.equal(c, ops.gemm_fp8(a, b, variant=4)) and torch.equal(c, ops.gemm_fp8(a, b, variant=5))
        if ops.gemm_has_diag():                                    # 3 / 6: A/B-record streams, in diagnostic builds only (6 and the ring kernels use the block-scaled MFMA)
            assert torch.equal(c, ops.gemm_fp8(a, b, variant=3)) and torch.equal(c, ops.gemm_fp8(a, b, variant=6))


def test_gemm_fp8_exact_on_small_integers(env):
    torch, ops, dev = env
    M, N, K = 256, 512, 768
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randint(-2, 3, (M, K), device=dev, generator=g).float()
    b = torch.randint(-2, 3, (N, K), device=dev, generator=g).float()
    b[:, 5] = (torch.arange(N, device=dev) % 3).float()
    a[:, 200] = (torch.arange(M, device=dev) % 4).float() - 1        # position-dependent along K too
    c = ops.gemm_fp8(a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
    assert torch.equal(c, (a @ b.t()).to(torch.bfloat16))


@pytest.mark.parametrize("shape", [(256, 256, 512), (512, 768, 1024), (1024, 512, 8192), (256, 512, 1536), (512, 256, 768),
                                   (128, 128, 256), (384, 640, 1280), (128, 256, 8448), (2048, 2048, 2048)])
def test_gemm_fp4_is_exact_against_the_decoded_reference(env, shape):
    """e2m1 values are multiples of 0.5 up to 6: products and fp32 sums of a few thousand of them are
    exact, so the only rounding is the bf16 store."""
    torch, ops, dev = env
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(K)
    a = torch.randint(0, 256, (M, K // 2), device=dev, generator=g, dtype=torch.uint8)
    b = torch.randint(0, 256, (N, K // 2), device=dev, generator=g, dtype=torch.uint8)
    a[:, 3] = (torch.arange(M, device=dev) % 251).to(torch.uint8)            # position- and row-dependent bytes
    b[:, K // 2 - 5] = (torch.arange(N, device=dev) % 239).to(torch.uint8)
    c = ops.gemm_fp4(a, b)                                                    # the hand-scheduled stream / the 128 x 128 ring kernel, by shape
    ref = ops.fp4_decode(a).double() @ ops.fp4_decode(b).double().t()
    assert torch.equal(c, ref.float().to(torch.bfloat16))
    assert torch.equal(c, ops.gemm_fp4(a, b, variant=14))                     # the eight-wave 128 x 128 ring kernel: K tile counts 1, 5, 33, even
    if N % 256 == 0:
        assert torch.equal(c, ops.gemm_fp4(a, b, variant=17))
    if M % 256 == 0:
        assert torch.equal(c, ops.gemm_fp4(a, b, variant=16))
    if M % 256 or N % 256:
        return
    assert torch.equal(c, ops.gemm_fp4(a, b, variant=0)) and torch.equal(c, ops.gemm_fp4(a, b, variant=1))
    if (K // 2) % 256 == 0 and K // 2 >= 512:
        assert torch.equal(c, ops.gemm_fp4(a, b, variant=2)) and torch.equal(c, ops.gemm_fp4(a, b, variant=3))
        assert torch.equal(c, ops.gemm_fp4(a, b, variant=4)) and torch.equal(c, ops.gemm_fp4(a, b, variant=6))
        if ops.gemm_has_diag():
            assert torch.equal(c, ops.gemm_fp4(a, b, variant=5))
    if hasattr(torch, "float4_e2m1fn_x2"):
        assert torch.equal(ops.gemm_fp4(a.view(torch.float4_e2m1fn_x2), b.view(torch.float4_e2m1fn_x2)), c)


@pytest.mark.parametrize("grid", [1, 2, 3, 5, 0])
def test_persistent_gemms_walk_several_tiles_per_workgroup(env, grid):
    """The persistent kernels take their output tiles from a per-workgroup table and fetch the next tile's operands during
    the current tile's last pass: with 1 / 2 / 3 / 5 workgroups for 12 tiles every workgroup walks several (unevenly
    many), with the default (one per CU) one each. Same bits as the one-workgroup-per-tile streams."""
    torch, ops, dev = env
    M, N, K = 768, 1024, 1024
    g = torch.Generator(device=dev).manual_seed(grid)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    a4 = torch.randint(0, 256, (M, K), device=dev, generator=g, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (N, K), device=dev, generator=g, dtype=torch.uint8)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    want = ops.gemm_bf16(a, b, variant=4), ops.gemm_fp8(a8, b8, variant=2), ops.gemm_fp4(a4, b4, variant=2)
    before = ops.gemm_persistent_grid(grid)
    try:
        for _ in range(3):
            got = ops.gemm_bf16(a, b, variant=8), ops.gemm_fp8(a8, b8, variant=4), ops.gemm_fp4(a4, b4, variant=3)
            assert all(torch.equal(x, y) for x, y in zip(got, want))
    finally:
        ops.gemm_persistent_grid(before)
    torch.testing.assert_close(want[0].float(), a.float() @ b.float().t(), rtol=1.6e-2, atol=0.4)


@pytest.mark.parametrize("pad_bytes", [256, 1024])
def test_gemm_streams_take_row_pitched_operands(env, pad_bytes):
    """Operands in a padded allocation (the burner's answer to row pitches the memory channels camp on): same bits as dense."""
    torch, ops, dev = env
    M, N, K = 512, 768, 1536
    g = torch.Generator(device=dev).manual_seed(pad_bytes)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)

    def padded(t):
        p = torch.full((t.shape[0], t.shape[1] + pad_bytes // t.element_size()), 3, dtype=torch.uint8 if t.dtype == torch.uint8 else torch.float32,
                       device=dev).to(t.dtype)[:, :t.shape[1]]          # the padding holds values that would show in a sum
        p.copy_(t)
        assert not p.is_contiguous()
        return p

    want = ops.gemm_bf16(a, b, variant=4)
    for v in (4, 8, 9, 13):
        assert torch.equal(ops.gemm_bf16(padded(a), padded(b), variant=v), want)
    # the default may cut these six tiles into K slices (another summation order): same bits as on dense operands, close to fp32
    assert torch.equal(ops.gemm_bf16(padded(a), padded(b)), ops.gemm_bf16(a, b, variant=10))
    assert torch.equal(ops.gemm_bf16(padded(a), padded(b), variant=11), ops.gemm_bf16(a, b, variant=11))
    torch.testing.assert_close(ops.gemm_bf16(padded(a), padded(b)).float(), a.float() @ b.float().t(), rtol=1.6e-2, atol=0.3)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    want8 = ops.gemm_fp8(a8, b8, variant=2)
    for v in (2, 4):
        assert torch.equal(ops.gemm_fp8(padded(a8), padded(b8), variant=v), want8)
    assert torch.equal(ops.gemm_fp8(padded(a8), padded(b8)), ops.gemm_fp8(a8, b8, variant=7))
    a4 = torch.randint(0, 256, (M, K), device=dev, generator=g, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (N, K), device=dev, generator=g, dtype=torch.uint8)
    want4 = ops.gemm_fp4(a4, b4, variant=2)
    for v in (2, 3, None):
        assert torch.equal(ops.gemm_fp4(padded(a4), padded(b4), variant=v), want4)              # fp4 sums are exact: any order gives the same bits
    # what cannot take a pitch says so instead of reading the wrong rows
    with pytest.raises(ops.KlmOpsError, match="pitched"):
        ops.gemm_bf16(padded(a), padded(b), variant=2)
    with pytest.raises(ops.KlmOpsError, match="one row pitch"):
        ops.gemm_bf16(padded(a), b, variant=8)
    with pytest.raises(ops.KlmOpsError, match="contiguous"):
        ops.gemm_bf16(a.t(), b.t())
    # padded_operand pads exactly the camping pitches
    assert ops.padded_operand(256, 12288, torch.bfloat16, dev).stride(0) == 12288 + 128
    assert ops.padded_operand(256, 8192, torch.bfloat16, dev).is_contiguous() and ops.padded_operand(256, 12032, torch.bfloat16, dev).is_contiguous()


def test_default_dispatch_rates_guard_the_kernel_choice(env):
    """Loose floors (a third below what the kernels measure) that a wrong default trips: 2048^3 on the 256 x 256 stream runs at
    ~460 TFLOP/s where the 128 x 128 ring kernel gives ~800; 6144^3 without its split round loses 10 %; 8192^3 on a ring kernel
    would read ~1000."""
    torch, ops, dev = env
    g = torch.Generator(device=dev).manual_seed(99)

    def rate(fn, flop, iters):
        fn()
        best = 0.0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); e1.synchronize()
            best = max(best, flop * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        return best

    for (M, N, K), floor_bf16, floor_fp8 in (((2048, 2048, 2048), 560.0, 950.0), ((1024, 4096, 4096), 620.0, 1100.0),
                                             ((6144, 6144, 6144), 1150.0, 2100.0), ((8192, 8192, 8192), 1250.0, 2500.0)):
        a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        flop, iters = 2.0 * M * N * K, max(3, int(2e12 / (2.0 * M * N * K)))
        r = rate(lambda: ops.gemm_bf16(a, b, out=c), flop, iters)
        assert r > floor_bf16, f"bf16 {M}x{N}x{K}: {r:.0f} TFLOP/s on variant {ops.pick_bf16_variant(M, N, K)}"
        a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
        r8 = rate(lambda: ops.gemm_fp8(a8, b8, out=c), flop, iters)
        assert r8 > floor_fp8, f"fp8 {M}x{N}x{K}: {r8:.0f} TFLOP/s on variant {ops.pick_mx_variant(M, N, K, 0, 7)}"
        del a, b, c, a8, b8


def test_superblock_tile_order_gives_the_same_product(env):
    """The persistent streams' second tile order (KlmQuadPlan.map = 1: one 16 x 16 super-block of tiles per round of the chip,
    an 8 x 4 patch per XCD; picked by itself for operands well past the Infinity Cache): every tile exactly once, the same bits
    as the chunked order."""
    torch, ops, dev = env
    if torch.cuda.get_device_properties(dev).multi_processor_count != 256:
        pytest.skip("the super-block order is defined for 256 workgroups")
    g = torch.Generator(device=dev).manual_seed(15)
    for M, N, K in ((4096, 8192, 512), (8192, 4096, 768)):
        assert ops.gemm_stream_plan(M, N, 2 * K, mode=5)["map"] == 1
        a = torch.randint(-3, 4, (M, K), device=dev, generator=g).float()
        b = torch.randint(-3, 4, (N, K), device=dev, generator=g).float()
        b[:, 0] = (torch.arange(N, device=dev) % 5).float()
        a[:, 1] = (torch.arange(M, device=dev) % 7).float() - 3
        want = (a @ b.t()).to(torch.bfloat16)
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        assert torch.equal(ops.gemm_bf16(a.to(torch.bfloat16), b.to(torch.bfloat16), out=c, variant=15), want)
        c.fill_(float("nan"))
        assert torch.equal(ops.gemm_fp8(a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn), out=c, variant=9), want)
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        y = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        assert torch.equal(ops.gemm_bf16(x, y, variant=15), ops.gemm_bf16(x, y, variant=13))


def test_default_dispatch_over_random_shapes(env):
    """`variant=None` over random tile-aligned shapes (every kernel family gets picked: 128 x 128, ring, persistent stream,
    padded operands): bf16 against the fp32 product, fp8 / fp4 against their ring kernels bit for bit."""
    import random
    torch, ops, dev = env
    rnd = random.Random(1234)
    g = torch.Generator(device=dev).manual_seed(1234)
    for _ in range(24):
        M, N = 128 * rnd.randint(1, 20), 128 * rnd.randint(1, 20)
        K = 64 * rnd.randint(2, 40)
        a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        c = ops.gemm_bf16(a, b)
        torch.testing.assert_close(c.float(), a.float() @ b.float().t(), rtol=1.6e-2, atol=6e-3 * K ** 0.5, msg=lambda m: f"{M}x{N}x{K}: {m}")
        if K % 128 == 0 and (M % 256 or N % 256):
            # only the 128 x 128 ring kernel takes these: fp8 against the fp32 product, fp4 (rows of K bytes) against its decode
            a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
            torch.testing.assert_close(ops.gemm_fp8(a8, b8).float(), a8.float() @ b8.float().t(), rtol=1.6e-2, atol=2e-3 * K ** 0.5)
            a4 = torch.randint(0, 256, (M, K), device=dev, generator=g, dtype=torch.uint8)
            b4 = torch.randint(0, 256, (N, K), device=dev, generator=g, dtype=torch.uint8)
            assert torch.equal(ops.gemm_fp4(a4, b4), (ops.fp4_decode(a4).double() @ ops.fp4_decode(b4).double().t()).float().to(torch.bfloat16)), (M, N, K)
        if M % 256 == 0 and N % 256 == 0 and K % 128 == 0 and K >= 256:
            a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
            assert torch.equal(ops.gemm_fp8(a8, b8), ops.gemm_fp8(a8, b8, variant=1)), (M, N, K)
            a4 = torch.randint(0, 256, (M, K), device=dev, generator=g, dtype=torch.uint8)
            b4 = torch.randint(0, 256, (N, K), device=dev, generator=g, dtype=torch.uint8)
            assert torch.equal(ops.gemm_fp4(a4, b4), ops.gemm_fp4(a4, b4, variant=1)), (M, N, K)
            # the same operands in padded allocations
            pa, pb = ops.padded_operand(M, K, torch.bfloat16, dev, pad_bytes=256), ops.padded_operand(N, K, torch.bfloat16, dev, pad_bytes=256)
            pa.copy_(a); pb.copy_(b)
            if not pa.is_contiguous():
                plan = ops.gemm_stream_plan(M, N, 2 * K, mode=1)
                if plan["split"] <= 1:             # whole tiles only: the same bits as the one-workgroup-per-tile stream
                    assert torch.equal(ops.gemm_bf16(pa, pb), ops.gemm_bf16(a, b, variant=4)), (M, N, K)
                else:                              # K slices: another summation order
                    torch.testing.assert_close(ops.gemm_bf16(pa, pb).float(), a.float() @ b.float().t(), rtol=1.6e-2, atol=6e-3 * K ** 0.5)


@pytest.mark.parametrize("grid,shape", [(4, (768, 768, 1024)), (4, (512, 768, 512)), (16, (1280, 1024, 2048)), (6, (1024, 1280, 3072)), (0, (1536, 1536, 4096))])
def test_partial_round_split_into_k_slices_and_staggered_k_loops(env, grid, shape):
    """The unit-based persistent stream (gemm_quad_asm.h): with `grid` workgroups the tile count is no multiple of the grid, so
    the last round is cut into K slices whose fp32 partial sums a second kernel adds up (variants 11 / fp8 8 / fp4 7 force it);
    variant 12 starts the K loop of XCD x's workgroups x * P / 8 passes in. Small-integer operands: every form must give the
    exact product, whatever its summation order; on random operands they stay within bf16 rounding of fp32."""
    torch, ops, dev = env
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(M + K)
    a = torch.randint(-3, 4, (M, K), device=dev, generator=g).to(torch.bfloat16)
    b = torch.randint(-3, 4, (N, K), device=dev, generator=g).to(torch.bfloat16)
    want = (a.float() @ b.float().t()).to(torch.bfloat16)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    a4 = torch.randint(0, 256, (M, K), device=dev, generator=g, dtype=torch.uint8)
    b4 = torch.randint(0, 256, (N, K), device=dev, generator=g, dtype=torch.uint8)
    want4 = (ops.fp4_decode(a4).double() @ ops.fp4_decode(b4).double().t()).float().to(torch.bfloat16)
    x = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    y = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    before = ops.gemm_persistent_grid(grid)
    try:
        plan = ops.gemm_stream_plan(M, N, 2 * K, mode=2)
        assert plan["rest"] > 0 and plan["split"] >= 2, plan          # the shapes are chosen so that there IS a partial round to split
        for _ in range(2):
            for v in (10, 11, 12, 13):
                c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
                assert torch.equal(ops.gemm_bf16(a, b, out=c, variant=v), want), v
            assert torch.equal(ops.gemm_fp8(a8, b8, variant=8), want) and torch.equal(ops.gemm_fp8(a8, b8, variant=7), want)
            assert torch.equal(ops.gemm_fp4(a4, b4, variant=7), want4) and torch.equal(ops.gemm_fp4(a4, b4, variant=6), want4)
        ref = x.float() @ y.float().t()
        for v in (11, 12):
            torch.testing.assert_close(ops.gemm_bf16(x, y, variant=v).float(), ref, rtol=1.6e-2, atol=6e-3 * K ** 0.5)
        assert torch.equal(ops.gemm_bf16(x, y, variant=13), ops.gemm_bf16(x, y, variant=4))      # no slices, no stagger: the plain stream's bits
        assert torch.equal(ops.gemm_bf16(x, y, variant=11), ops.gemm_bf16(x, y, variant=11))     # and the split is deterministic
    finally:
        ops.gemm_persistent_grid(before)


def test_dense_operands_on_a_bad_pitch_are_repitched_with_the_same_bits(env):
    """ops.repitched copies rows into padded scratch (klm_hbm_repitch); `gemm_bf16(variant=None)` does that by itself for dense
    operands on a 24 KiB pitch once the problem is large enough, and the result has the bits of the dense run."""
    torch, ops, dev = env
    g = torch.Generator(device=dev).manual_seed(3)
    t = torch.randint(0, 255, (777, 4096), device=dev, generator=g, dtype=torch.uint8)
    p = ops.repitched(t, "t", pad_bytes=256)
    assert p.shape == t.shape and p.stride(0) == 4096 + 256 and torch.equal(p, t)
    M = N = 6144
    K = 12288
    assert ops.repack_worthwhile(M, N, 2 * K)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    c = ops.gemm_bf16(a, b)                                  # re-pitched, and its partial round split
    pa, pb = ops.repitched(a, "x"), ops.repitched(b, "y")
    assert not pa.is_contiguous() and torch.equal(c, ops.gemm_bf16(pa, pb, variant=10))
    torch.testing.assert_close(c[:512, :512].float(), a[:512].float() @ b[:512].float().t(), rtol=1.6e-2, atol=1.0)


def test_ops_follow_the_tensors_device_not_the_current_one(env):
    """A process whose current device is 0 launches on device 1 (per-device kernel attributes, the persistent kernels' CU count,
    the launch itself all act on the CURRENT device: the wrappers make that the tensor's). Needs two GPUs."""
    torch, ops, dev = env
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    d1 = torch.device("cuda:1")
    assert torch.cuda.current_device() == 0
    a = torch.randn(512, 1024, device=d1, dtype=torch.bfloat16)
    b = torch.randn(768, 1024, device=d1, dtype=torch.bfloat16)
    ref = a.float() @ b.float().t()
    for v in (None, 8, 4, 2, 0):
        torch.testing.assert_close(ops.gemm_bf16(a, b, variant=v).float(), ref, rtol=1.6e-2, atol=0.4)
    a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
    assert torch.equal(ops.gemm_fp8(a8, b8), ops.gemm_fp8(a8, b8, variant=1))
    x = torch.empty(1 << 26, dtype=torch.uint8, device=d1)
    y = torch.empty_like(x)
    ops.hbm_fill(x, 0x01020304)
    ops.hbm_copy(y, x)
    torch.cuda.synchronize(d1)
    assert torch.equal(x, y) and int(y.view(torch.int32)[12345]) == 0x01020304
    assert torch.cuda.current_device() == 0
    from k8s_llm_monitor_amd.loadgen.burner import Burner
    o = Burner("gemm_bf16", size=2048, duty=100.0, device=1).run(0.5)
    assert o["kernels"] > 0 and o["tflops_busy"] > 50


def test_fp4_decode_table(env):
    torch, ops, dev = env
    codes = torch.arange(256, device=dev, dtype=torch.uint8).reshape(1, 256)
    v = ops.fp4_decode(codes)[0]
    assert v.shape == (512,) and v[2 * 0x21].item() == 0.5 and v[2 * 0x21 + 1].item() == 1.0     # low nibble first
    assert sorted(set(v.abs().tolist())) == [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]


def test_shape_and_dtype_errors_are_raised_not_launched(env):
    torch, ops, dev = env
    a = torch.zeros(100, 64, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(128, 64, device=dev, dtype=torch.bfloat16)
    with pytest.raises(ops.KlmOpsError, match="multiple of the tile"):
        ops.gemm_bf16(a, b, variant=0)
    with pytest.raises(ops.KlmOpsError, match="bfloat16"):
        ops.gemm_bf16(a.float(), b.float())
    with pytest.raises(ops.KlmOpsError, match="do not match"):
        ops.gemm_bf16(torch.zeros(128, 64, device=dev, dtype=torch.bfloat16), torch.zeros(128, 128, device=dev, dtype=torch.bfloat16))
    with pytest.raises(ops.KlmOpsError, match="GPU only"):
        ops.gemm_bf16(a.cpu(), b.cpu())
    with pytest.raises(ops.KlmOpsError):
        ops.hbm_fill(torch.zeros(17, dtype=torch.uint8, device=dev))          # not a multiple of 16 bytes


def test_hbm_kernels(env):
    torch, ops, dev = env
    n = 64 << 20
    x = torch.zeros(n, dtype=torch.uint8, device=dev)
    ops.hbm_fill(x, 0x01020304)
    w = x.view(torch.int32)
    assert int(w[0]) == 0x01020304 and int(w[-1]) == 0x01020304 and bool((w == 0x01020304).all())
    assert ops.hbm_checksum(x) == 0x01020304 * (n // 4)
    y = torch.zeros_like(x)
    src = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
    ops.hbm_copy(y, src)
    assert torch.equal(y, src)
    assert ops.hbm_checksum(src) == int(src.view(torch.int32).to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    ops.hbm_read(src, sink)
    torch.cuda.synchronize()
    odd = torch.zeros(16 * 3, dtype=torch.uint8, device=dev)                   # smaller than one grid pass
    ops.hbm_fill(odd, 0xFFFFFFFF)
    assert bool((odd == 255).all())


def test_hbm_kernels_past_one_launch_of_the_grid_limit(env):
    """64 GiB and more cannot be one launch (HIP bounds gridDim.x * blockDim.x by 2^32 - 1): the default kernels walk such a
    buffer in sub-range launches (klm_hbm_launch_plan). 66 GiB here: filled, summed, copied, spot-checked on both sides of
    every launch boundary and at the very end."""
    torch, ops, dev = env
    GIB = 1 << 30
    n = 66 * GIB
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 2 * n + 8 * GIB:
        pytest.skip(f"needs {2 * n >> 30} GiB of free HBM, {free >> 30} GiB are")
    launches, first, last = ops.hbm_launch_plan(n)
    assert launches == 2 and first * 256 <= 0xFFFFFFFF and (launches - 1) * first + last == n // 4096
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    try:
        ops.hbm_fill(x, 0x0A0B0C0D)
        w = x.view(torch.int32)
        edge = first * 4096 // 4                                   # first int32 of the second launch
        for i in (0, 1, edge - 1, edge, edge + 1, n // 4 - 1):
            assert int(w[i]) == 0x0A0B0C0D, i
        assert ops.hbm_checksum(x) == 0x0A0B0C0D * (n // 4)
        # a copy across the boundary: mark positions on both sides, copy, read them back
        marks = {0: 1, edge - 1: 2, edge: 3, n // 4 - 1: 4, (n // 4) // 3: 5}
        for i, v in marks.items():
            w[i] = v
        y = torch.empty(n, dtype=torch.uint8, device=dev)
        ops.hbm_copy(y, x)
        wy = y.view(torch.int32)
        for i, v in marks.items():
            assert int(wy[i]) == v, i
        assert int(wy[edge + 1]) == 0x0A0B0C0D
        assert ops.hbm_checksum(y) == ops.hbm_checksum(x)
        sink = torch.zeros(4, dtype=torch.int32, device=dev)
        ops.hbm_read(y, sink)
        torch.cuda.synchronize()
    finally:
        del x
        y = w = wy = None
        torch.cuda.empty_cache()


def test_kernels_run_on_the_current_stream_and_capture_into_a_graph(env):
    torch, ops, dev = env
    a = torch.randn(512, 256, device=dev, dtype=torch.bfloat16)
    b = torch.randn(512, 256, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(512, 512, device=dev, dtype=torch.bfloat16)
    ref = ops.gemm_bf16(a, b).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.gemm_bf16(a, b, out=out)          # warm (sets the LDS attribute outside capture)
        s.synchronize()
        out.zero_()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                ops.gemm_bf16(a, b, out=out)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
