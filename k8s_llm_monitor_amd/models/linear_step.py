"""The flagship validation "model": one linear layer's training step on the burner kernels.

The reference has no model of any kind (no forward pass, no training entry
point). The validation toolkit needs a load that looks like what the monitored cluster runs, so
that the dashboard's GFX / HBM / xGMI readings can be checked against it: a transformer-style
linear layer is three GEMMs per step plus, under data parallelism, an all-reduce of the weight
gradient. ``bench.py`` times exactly this step; ``__graft_entry__.smoke`` runs a tiny one.
"""
from __future__ import annotations


class LinearLayerStep:
    """One rank's buffers and the step function.

    A step is the GEMM work of one linear layer's training step on the burner kernel, with
    the data-parallel gradient all-reduce overlapped the way DP training overlaps it:

        fwd    Y  = X  @ W^T          (compute stream)
        dgrad  dX = dY @ (W^T)^T      (compute stream)
        wgrad  dW = dY^T @ (X^T)^T    (compute stream)  -> all-reduce(dW) over RCCL on the
                                                           communication stream, overlapping
                                                           the NEXT step's fwd and dgrad
    All operands are square `size` x `size` bf16, stored K-major as the kernel wants them
    (the transposed copies are made once, outside the timed region). dW is double-buffered:
    the wgrad of step i+2 waits for the all-reduce of step i. 6*size^3 FLOP and one
    2*size^2-byte all-reduce per step.
    """

    def __init__(self, torch, dist, ops, dev, world: int, size: int, gemm=None, distributed=None, dp_kernel=None):
        self.torch, self.dist, self.ops, self.world = torch, dist, ops, world
        dd = self.distributed = (world > 1) if distributed is None else bool(distributed)
        # `gemm(a, b, out)` computes out = a @ b.T; the default is the hand-written MFMA kernel.
        # CPU tests of the stream/event/all-reduce logic pass a torch.matmul stand-in.
        # Under data parallelism the stream with ONE WORKGROUP PER TILE (variant 4), not the persistent default: a persistent
        # workgroup owns its CU for the whole GEMM and fills every SIMD's register file, so the all-reduce kernel of the
        # communication stream could only start between GEMMs and would then keep the CUs it took away from the next
        # GEMM's statically assigned workgroups (their tiles would run after everyone else's). With one workgroup per tile
        # the collective gets CUs as tiles retire and the remaining tiles flow over the CUs that are left.
        # `dp_kernel=True` forces that stream without a process group: the one-GPU point of a scaling curve measured on the
        # kernel the N > 1 points run (bench.py's `value_dp_kernel`), so that scaling efficiency compares like with like.
        quad = size % 256 == 0 and size >= 256
        use_dp = dd if dp_kernel is None else bool(dp_kernel)
        variant = 4 if (use_dp and quad) else None
        #: the bf16 kernel variant every GEMM of the step runs (None: a caller-supplied `gemm`)
        self.gemm_variant = None if gemm is not None else (variant if variant is not None else ops.pick_bf16_variant(size, size, size))
        self.gemm = gemm or (lambda a, b, out: ops.gemm_bf16(a, b, out=out, variant=variant))
        self.gpu = dev.type == "cuda"
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + (dist.get_rank() if dd else 0))

        def rnd():
            # uniform [-1, 1) operands: zero-filled operands read ~15-20 % high (DVFS)
            return (torch.rand(size, size, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)

        self.x, self.w, self.dy = rnd(), rnd(), rnd()
        self.wt, self.dyt, self.xt = self.w.t().contiguous(), self.dy.t().contiguous(), self.x.t().contiguous()
        self.a, self.b = self.dyt, self.xt                     # the operands of the all-reduced GEMM (tests)
        self.y = torch.empty(size, size, device=dev, dtype=torch.bfloat16)
        self.dx = torch.empty(size, size, device=dev, dtype=torch.bfloat16)
        self.c = [torch.empty(size, size, device=dev, dtype=torch.bfloat16) for _ in range(2)]    # dW, double-buffered
        self.size = size
        self.flops = 3 * 2.0 * size * size * size
        self.comm = torch.cuda.Stream(device=dev) if (dd and self.gpu) else None
        self.ev_gemm = [torch.cuda.Event() for _ in range(2)] if self.gpu else []
        self.ev_comm = [torch.cuda.Event() for _ in range(2)] if self.gpu else []
        self.i = 0
        self.allreduce_bytes = size * size * 2 if dd else 0

    def step(self):
        torch = self.torch
        k = self.i & 1
        if not self.gpu:                       # CPU rehearsal: same data flow, no streams
            self.gemm(self.x, self.w, self.y)
            self.gemm(self.dy, self.wt, self.dx)
            self.gemm(self.dyt, self.xt, self.c[k])
            if self.distributed:
                self.dist.all_reduce(self.c[k])
            self.i += 1
            return
        main = torch.cuda.current_stream()
        self.gemm(self.x, self.w, self.y)
        self.gemm(self.dy, self.wt, self.dx)
        if self.comm is not None:
            # dW buffer k was all-reduced two steps ago: wgrad may only overwrite it after that
            main.wait_event(self.ev_comm[k])
        self.gemm(self.dyt, self.xt, self.c[k])
        if self.comm is not None:
            self.ev_gemm[k].record(main)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.ev_gemm[k])
                self.dist.all_reduce(self.c[k])
                self.ev_comm[k].record(self.comm)
        self.i += 1

    def drain(self):
        if self.comm is not None:
            self.torch.cuda.current_stream().wait_stream(self.comm)
