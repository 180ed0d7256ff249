"""Tensor-parallel process group for the serving engine: one process per GPU.

Three channels, chosen for what each carries:

  data, small (decode: [batch, hidden] bf16 twice per layer, candidates of the sharded argmax)
      the xGMI mesh kernels (parallel/xgmi.py): one launch, hipGraph-safe, latency-bound design;
  data, large (prefill: thousands of tokens x hidden)
      RCCL through ``torch.distributed`` when every rank owns a GPU (bandwidth-bound: the library's
      pipelined schedule is fine there), else the two-shot mesh kernel in pieces;
  control (which requests arrived, stop)
      a gloo CPU group: a few bytes per engine step, never on a GPU stream.

On CPU (tests, rehearsal) everything runs over gloo. ``SingleGroup`` is the world-size-1 stand-in,
so model and engine code has one path.

No reference counterpart.
"""
from __future__ import annotations

from typing import Any, List, Optional


class SingleGroup:
    rank, world = 0, 1
    backend = "none"

    def all_reduce(self, x):
        return x

    def all_gather(self, x):
        return x.unsqueeze(0)

    def broadcast_object(self, obj: Any = None) -> Any:
        return obj

    def gather_objects(self, obj: Any) -> List[Any]:
        return [obj]

    def gather_ints(self, values) -> List[List[int]]:
        return [[int(v) for v in values]]

    def barrier(self) -> None:
        pass

    def check(self) -> None:
        pass

    def resync(self) -> None:
        pass

    def close(self) -> None:
        pass

    def describe(self) -> dict:
        return {"world": 1, "backend": "none"}


class TPGroup:
    """Collective construction: every rank of the default ``torch.distributed`` group calls it.

    ``data``: "xgmi" (mesh kernels for everything), "rccl" (torch.distributed for everything: the
    default group must then be nccl), "auto" (mesh kernels up to ``xgmi_max_bytes``, RCCL above when
    the default group is nccl), "gloo" (CPU)."""

    def __init__(self, device, data: str = "auto", xgmi_max_bytes: int = 4 << 20, slot_bytes: int = 16 << 20,
                 blocks: int = 32, ctrl_timeout_s: float = 7 * 24 * 3600.0, spin_limit_s: float = 20.0):
        import datetime

        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device)
        default_backend = dist.get_backend()
        # followers sit in a control broadcast while the engine is idle: that must not time out
        self.ctrl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=ctrl_timeout_s))
        on_gpu = self.device.type == "cuda"
        if data == "auto":
            data = "xgmi+rccl" if (on_gpu and default_backend == "nccl") else ("xgmi" if on_gpu else "gloo")
        if data in ("rccl", "xgmi+rccl") and default_backend != "nccl":
            raise ValueError(f"data={data!r} needs the default process group on nccl (it is {default_backend})")
        if data == "gloo" and on_gpu:
            raise ValueError("data='gloo' is the CPU path")
        self.backend = data
        self.xgmi_max_bytes = int(xgmi_max_bytes)
        self.comm = None
        self.self_test_error: Optional[str] = None
        if data.startswith("xgmi"):
            from . import xgmi
            try:
                self.comm = xgmi.XgmiComm(self.rank, self.world, self.device, xgmi.gloo_exchange(self.ctrl),
                                          slot_bytes=slot_bytes, blocks=blocks, spin_limit_s=spin_limit_s)
                mine = None
            except xgmi.XgmiError as e:          # e.g. cached memory across different GPUs
                self.comm, mine = None, f"rank {self.rank}: {e}"
            # construction can fail on some ranks only (an allocation, an IPC open): agree before anyone launches
            made: List[Optional[str]] = [None] * self.world
            dist.all_gather_object(made, mine, group=self.ctrl)
            if any(made):
                self.self_test_error = "; ".join(m for m in made if m)
                if self.comm is not None:
                    self.comm.close()
                    self.comm = None
            else:
                self.self_test_error = self._self_test()
            if self.self_test_error:
                # the same verdict on every rank (it is gathered), so every rank takes the same branch
                if self.comm is not None:
                    self.comm.close()
                self.comm = None
                if default_backend != "nccl":
                    raise RuntimeError(f"xGMI mesh collectives failed their self-test and there is no RCCL group to fall "
                                       f"back to: {self.self_test_error}")
                print(f"[tp] xGMI mesh collectives failed their self-test ({self.self_test_error}); using RCCL", flush=True)
                self.backend = "rccl"

    def _self_test(self) -> Optional[str]:
        """One all-reduce per schedule and one all-gather with known operands, before any model traffic. Returns
        None, or what went wrong on ANY rank."""
        torch = self.torch
        mine: Optional[str] = None
        try:
            for n in (4096, 1 << 20):                       # one-shot, two-shot
                x = torch.full((n,), float(self.rank + 1), dtype=torch.bfloat16, device=self.device)
                self.comm.all_reduce(x)
                want = self.world * (self.world + 1) / 2.0
                if not bool((x.float() == want).all().item()):
                    mine = f"rank {self.rank}: all-reduce of {n} elements is wrong"
            g = self.comm.all_gather(torch.full((8,), float(self.rank), dtype=torch.float32, device=self.device))
            if not bool((g[:, 0] == torch.arange(self.world, device=self.device)).all().item()):
                mine = f"rank {self.rank}: all-gather is wrong"
            if self.comm.status():
                mine = f"rank {self.rank}: a peer did not arrive (status {self.comm.status():#x})"
        except Exception as e:
            mine = f"rank {self.rank}: {type(e).__name__}: {e}"
        verdicts: List[Optional[str]] = [None] * self.world
        self.dist.all_gather_object(verdicts, mine, group=self.ctrl)
        bad = [v for v in verdicts if v]
        return "; ".join(bad) if bad else None

    # ------------------------------------------------------------------ data
    def all_reduce(self, x):
        """In-place sum (bf16 on GPUs)."""
        if self.comm is not None and (self.backend == "xgmi" or x.numel() * x.element_size() <= self.xgmi_max_bytes):
            return self.comm.all_reduce(x)
        self.dist.all_reduce(x)
        return x

    def all_gather(self, x):
        """-> [world, *x.shape]"""
        if self.comm is not None and x.numel() * x.element_size() <= self.comm.slot_bytes:
            return self.comm.all_gather(x)
        out = self.torch.empty((self.world, *x.shape), dtype=x.dtype, device=x.device)
        if self.backend == "xgmi":                          # no library under us: move the rows in slot-sized pieces
            flat, of = x.reshape(-1), out.view(self.world, -1)
            step = self.comm.slot_bytes // x.element_size()
            for o in range(0, flat.numel(), step):
                piece = self.comm.all_gather(flat[o:o + step].contiguous())
                of[:, o:o + step] = piece
            return out
        if self.backend == "gloo":
            self.dist.all_gather(list(out.unbind(0)), x.contiguous())
        else:
            self.dist.all_gather_into_tensor(out.view(-1), x.reshape(-1))
        return out

    # ------------------------------------------------------------------ control
    def broadcast_object(self, obj: Any = None) -> Any:
        """Rank 0's object on every rank (CPU, gloo)."""
        box: List[Optional[Any]] = [obj if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0, group=self.ctrl)
        return box[0]

    def gather_objects(self, obj: Any) -> List[Any]:
        """Every rank's object on every rank, in rank order (CPU, gloo)."""
        out: List[Any] = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.ctrl)
        return out

    def gather_ints(self, values) -> List[List[int]]:
        """Every rank's small vector of integers on every rank, in rank order: ONE gloo all-gather of a fixed-size int64
        tensor, nothing pickled (the per-step control message of the serving engine; `gather_objects` costs a pickle and two
        collectives per call)."""
        import torch
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.ctrl)
        return [o.tolist() for o in out]

    def barrier(self) -> None:
        self.dist.barrier(group=self.ctrl)

    def check(self) -> None:
        if self.comm is not None:
            self.comm.check()

    def resync(self) -> None:
        """Collective: bring the mesh collectives' epochs back in step after a step that not every rank completed."""
        if self.comm is not None:
            self.comm.resync()
        else:
            self.barrier()

    def close(self) -> None:
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def describe(self) -> dict:
        d = {"world": self.world, "rank": self.rank, "backend": self.backend,
             # the start-up self-test of the mesh kernels (every schedule + an all-gather with known operands), as `/stats` shows it
             "self_test": ("passed" if self.comm is not None else
                           f"failed, fell back to RCCL: {self.self_test_error}" if self.self_test_error else "not run (no mesh kernels in this group)")}
        if self.comm is not None:
            d["xgmi"] = self.comm.describe()
        return d
