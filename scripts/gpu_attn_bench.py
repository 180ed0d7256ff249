"""Paged attention kernels alone (Llama-3-8B head shapes): prefill q_tiles sweep, decode sweep."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_llm_monitor_amd.ops import llm  # noqa: E402

DEV, n_q, n_kv = "cuda", 32, 8
scale = 1 / math.sqrt(128)


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


nb = 4096
kc = torch.randn(nb, n_kv, 16, 128, device=DEV).to(torch.bfloat16)
vc = torch.randn(nb, n_kv, 128, 16, device=DEV).to(torch.bfloat16)
for n_seqs, L in ((1, 512), (1, 2048), (1, 4096), (1, 8192), (8, 1024)):
    nblk = -(-L // 16)
    bt = torch.arange(n_seqs * nblk, dtype=torch.int32, device=DEV).view(n_seqs, nblk).contiguous()
    info = torch.tensor([(i * L, L, L) for i in range(n_seqs)], dtype=torch.int32, device=DEV)
    q = torch.randn(n_seqs * L, (n_q + 2 * n_kv) * 128, device=DEV).to(torch.bfloat16)
    flops = 4.0 * n_q * 128 * n_seqs * L * (L + 1) / 2
    line = f"prefill {n_seqs}x{L}:"
    for var, qt in ((0, 1), (0, 2), (1, 1), (1, 2)):
        tiles = llm.prefill_tiles([L] * n_seqs, n_q // n_kv, qt).to(DEV)
        us = t(lambda: llm.paged_attn_prefill(q, kc, vc, bt, info, tiles, scale, n_q, q_tiles=qt, variant=var))
        line += f"  v{var}qt{qt} {us:7.1f}us {flops / us / 1e6:5.0f}TF"
    print(line, flush=True)
# decode: a cache larger than the 256 MB last-level cache, launches timed from a hipGraph
del kc, vc
nb = 32768
kc = torch.randn(nb, n_kv, 16, 128, device=DEV).to(torch.bfloat16)
vc = torch.randn(nb, n_kv, 128, 16, device=DEV).to(torch.bfloat16)
for B, ctx in ((1, 1024), (1, 4096), (4, 1024), (16, 1024), (64, 1024), (256, 1024), (64, 4000)):
    nblk = -(-ctx // 16)
    if B * nblk > nb:
        continue
    bt = (torch.randperm(nb)[:B * nblk]).to(torch.int32).to(DEV).view(B, nblk).contiguous()
    cl = torch.full((B,), ctx, dtype=torch.int32, device=DEV)
    q = torch.randn(B, (n_q + 2 * n_kv) * 128, device=DEV).to(torch.bfloat16)
    ws = llm.DecodeWorkspace(B, n_q, 64, DEV)
    out = torch.empty(B, n_q * 128, device=DEV, dtype=torch.bfloat16)
    line = f"decode B={B} ctx={ctx}:"
    for ns in sorted({1, 2, 4, 8, 16, 32, 64}):
        def call():
            llm.paged_attn_decode(q, kc, vc, bt, cl, scale, n_q, out=out, num_splits=ns, workspace=ws)
        call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                call()
        us = t(g.replay, iters=10) / 10
        line += f"  s{ns} {us:6.1f}us {B * ctx * 4096 / us / 1e6:4.2f}"
    print(line + f"  (pick {llm.pick_splits(B, n_kv, 4096)})", flush=True)
