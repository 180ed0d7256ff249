"""Python side of the xGMI mesh collectives (csrc/xgmi_allreduce.hip): buffer set-up over hipIpc,
the launch wrappers and the choice between the one-shot and the two-shot schedule.

One ``XgmiComm`` per rank (= per process = per GPU). Creating it is collective: every rank
allocates its buffer, the 64-byte IPC handles are exchanged through ``exchange`` (any all-gather of
bytes: ``gloo_exchange`` uses a ``torch.distributed`` CPU group) and every rank maps its peers'
buffers. After that a collective is ONE kernel launch per rank with constant arguments: capturable
in a hipGraph, no host synchronisation, no library call on the token path.

The ranks' processes may also share one GPU (``tests/test_gpu_tp.py`` does that on the one-GPU box):
the kernels only see pointers.

No reference counterpart.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Callable, List, Optional, Sequence

LIB_NAME = "libklm_xgmi.so"
ONE_SHOT, TWO_SHOT, GATHER = 0, 1, 2
_lib = None


def lib_path() -> Path:
    return Path(__file__).resolve().parent.parent / "_native" / "lib" / LIB_NAME


def load_lib():
    """The shared library, or a RuntimeError that says how to build it (never a silent fallback)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not p.exists():
            raise RuntimeError(f"{p} is missing: run `python -m k8s_llm_monitor_amd build`")
        L = ctypes.CDLL(str(p))
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.klm_xgmi_alloc.argtypes = [sz, ctypes.POINTER(vp), ctypes.c_char_p, ctypes.POINTER(ci)]
        L.klm_xgmi_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
        L.klm_xgmi_close.argtypes = [vp]
        L.klm_xgmi_free.argtypes = [vp]
        L.klm_xgmi_collective.argtypes = [ctypes.POINTER(vp), vp, vp, ci, ci, vp, vp, sz, ci, ci, vp, vp, sz, ctypes.c_double, vp]
        L.klm_xgmi_limits.argtypes = [ctypes.POINTER(ci)] * 4
        for f in (L.klm_xgmi_alloc, L.klm_xgmi_open, L.klm_xgmi_close, L.klm_xgmi_free, L.klm_xgmi_collective, L.klm_xgmi_limits):
            f.restype = ci
        _lib = L
    return _lib


def limits() -> dict:
    v = [ctypes.c_int() for _ in range(4)]
    load_lib().klm_xgmi_limits(*[ctypes.byref(x) for x in v])
    return {"max_ranks": v[0].value, "max_blocks": v[1].value, "sig_bytes": v[2].value, "counter_bytes": v[3].value}


def gloo_exchange(group=None) -> Callable[[bytes], List[bytes]]:
    """All-gather of one bytes object per rank over a torch.distributed group (use a gloo group)."""
    import torch.distributed as dist

    def exchange(mine: bytes) -> List[bytes]:
        out: List[Optional[bytes]] = [None] * dist.get_world_size(group)
        dist.all_gather_object(out, mine, group=group)
        return [bytes(b) for b in out]               # type: ignore[arg-type]
    return exchange


class XgmiError(RuntimeError):
    pass


class XgmiComm:
    """``slot_bytes``: capacity of one of the two data slots (one-shot and gather messages up to this size,
    two-shot up to half of it; larger tensors go through in pieces). ``blocks``: workgroups per launch; it is
    part of the protocol (block b pairs with block b of the peers), so it is fixed for the communicator's life
    and must be the same on every rank."""

    SIG_PAD = 1 << 16

    def __init__(self, rank: int, world: int, device, exchange: Callable[[bytes], List[bytes]],
                 slot_bytes: int = 16 << 20, blocks: int = 32, one_shot_max: int = 256 << 10, spin_limit_s: float = 20.0):
        import torch
        lim = limits()
        if world not in (1, 2, 4, 8) or world > lim["max_ranks"] or not 0 <= rank < world:
            raise ValueError(f"world {world} / rank {rank}: 1, 2, 4 or 8 ranks")
        if not 1 <= blocks <= lim["max_blocks"] or slot_bytes % (32 * world) or lim["sig_bytes"] > self.SIG_PAD:
            raise ValueError("blocks or slot_bytes out of range")
        self.torch, self.L = torch, load_lib()
        self._exchange = exchange
        self.spin_limit_s = float(spin_limit_s)        # how long a kernel waits for a peer before it gives up (status != 0)
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self.slot_bytes, self.blocks, self.one_shot_max = int(slot_bytes), int(blocks), int(one_shot_max)
        torch.cuda.set_device(self.device)
        torch.cuda.current_stream(self.device).synchronize()
        # one allocation per rank: [flags, padded to 64 KiB][slot 0][slot 1]
        self.nbytes = self.SIG_PAD + 2 * self.slot_bytes
        base, fine = ctypes.c_void_p(), ctypes.c_int()
        handle = ctypes.create_string_buffer(64)
        rc = self.L.klm_xgmi_alloc(self.nbytes, ctypes.byref(base), handle, ctypes.byref(fine))
        # From here on every rank makes the SAME exchanges whatever happens to it locally: a rank that raised before an
        # exchange would leave its peers waiting in it. Local trouble travels as `err` and everyone raises at the end.
        err: Optional[str] = None
        if rc != 0:
            err = f"klm_xgmi_alloc({self.nbytes}) failed ({rc}): is HSA_ENABLE_IPC_MODE_LEGACY=0 set?"
        self.base, self.fine_grained = base.value, bool(fine.value)
        self._closed = False
        self.peer_base: List[int] = []
        self._opened: List[int] = []
        # everyone publishes (handle, pid, geometry, device, memory kind); a rank whose geometry differs would deadlock the protocol
        mine = handle.raw + f"|{os.getpid()}|{self.slot_bytes}|{self.blocks}|{self.one_shot_max}|{self._device_id()}|{int(self.fine_grained)}|{int(err is None)}".encode()
        infos = exchange(mine)
        if len(infos) != world:
            err = err or f"exchange returned {len(infos)} entries for {world} ranks"
            infos = []
        metas = [info[65:].decode().split("|") for info in infos]
        failed = [r for r, m in enumerate(metas) if m[-1] != "1"]
        if failed and err is None:
            err = f"rank(s) {failed} could not allocate their buffer"
        # cached memory is only coherent between processes that share ONE GPU (the single-box tests); across devices
        # the flag protocol needs the uncached allocation
        self.distinct_devices = len({m[4] for m in metas}) > 1
        coarse = [r for r, m in enumerate(metas) if m[5] == "0"]
        if err is None and self.distinct_devices and coarse:
            err = (f"rank(s) {coarse} got ordinary (cached) device memory instead of an uncached allocation and the ranks are "
                   f"on different GPUs: the flag protocol is not coherent across devices on such memory")
        if err is None:
            for r, (info, meta) in enumerate(zip(infos, metas)):
                if meta[1:4] != [str(self.slot_bytes), str(self.blocks), str(self.one_shot_max)]:
                    err = f"rank {r} was created with a different geometry: {meta[1:4]}"
                    break
                if r == rank:
                    self.peer_base.append(self.base)
                    continue
                p = ctypes.c_void_p()
                rc = self.L.klm_xgmi_open(info[:64], ctypes.byref(p))
                if rc != 0:
                    err = f"hipIpcOpenMemHandle of rank {r}'s buffer failed ({rc})"
                    break
                self.peer_base.append(p.value)
                self._opened.append(p.value)
        if err is None:
            self._data = (ctypes.c_void_p * world)(*[b + self.SIG_PAD for b in self.peer_base])
            self._sig_table = torch.tensor(self.peer_base, dtype=torch.int64, device=self.device)
            self._counters = torch.zeros(lim["max_blocks"], dtype=torch.int32, device=self.device)
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
            torch.cuda.synchronize(self.device)
        # nobody launches before every buffer is mapped and zeroed -- and nobody at all if any rank failed
        verdicts = exchange(b"ready" if err is None else b"failed: " + err.encode())
        bad = [f"rank {r}: {v.decode()[8:]}" for r, v in enumerate(verdicts) if v != b"ready"]
        if bad:
            if self.base:
                self.close()
            self._closed = True
            raise XgmiError("; ".join(bad))

    def _device_id(self) -> str:
        """Something that names the physical GPU across processes (each process may call it cuda:0)."""
        try:
            pr = self.torch.cuda.get_device_properties(self.device)
            return str(getattr(pr, "uuid", None) or f"{pr.pci_domain_id}:{pr.pci_bus_id}:{pr.pci_device_id}")
        except Exception:
            return f"unknown-{self.device}"

    # ------------------------------------------------------------------ launches
    def _launch(self, x_ptr: int, out_ptr: int, nbytes: int, mode: int) -> None:
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        rc = self.L.klm_xgmi_collective(self._data, self._sig_table.data_ptr(), self.base, self.rank, self.world, x_ptr, out_ptr,
                                        nbytes, mode, self.blocks, self._counters.data_ptr(), self._status.data_ptr(),
                                        self.slot_bytes, self.spin_limit_s, s)
        if rc != 0:
            raise XgmiError(f"klm_xgmi_collective rejected the call ({rc}): {nbytes} bytes, mode {mode}")

    def _check(self, x, what: str) -> int:
        if self._closed:
            raise XgmiError("communicator is closed")
        if x.device != self.device or not x.is_contiguous():
            raise ValueError(f"{what}: a contiguous tensor on {self.device} is needed")
        n = x.numel() * x.element_size()
        if n == 0 or n % 16 or x.data_ptr() % 16:
            raise ValueError(f"{what}: {n} bytes at {x.data_ptr():#x}: size and address must be multiples of 16")
        return n

    def plan(self, nbytes: int) -> List[tuple]:
        """(offset, bytes, mode) pieces of one all-reduce; the same on every rank by construction."""
        if nbytes <= self.one_shot_max or nbytes % (16 * self.world):
            mode, cap = ONE_SHOT, self.slot_bytes
        else:
            mode, cap = TWO_SHOT, self.slot_bytes // 2
        return [(o, min(cap, nbytes - o), mode) for o in range(0, nbytes, cap)]

    def all_reduce(self, x):
        """In-place sum over the ranks of a bf16 tensor (fp32 accumulation in rank order: every rank gets the same bits)."""
        if x.dtype != self.torch.bfloat16:
            raise ValueError("all_reduce sums bf16")
        n = self._check(x, "all_reduce")
        p = x.data_ptr()
        for off, nb, mode in self.plan(n):
            self._launch(p + off, p + off, nb, mode)
        return x

    def all_gather(self, x, out=None):
        """-> [world, *x.shape]; any dtype (bytes are moved)."""
        n = self._check(x, "all_gather")
        if n > self.slot_bytes:
            raise ValueError(f"all_gather of {n} bytes exceeds the slot ({self.slot_bytes})")
        if out is None:
            out = self.torch.empty((self.world, *x.shape), dtype=x.dtype, device=self.device)
        elif out.numel() * out.element_size() != n * self.world or not out.is_contiguous():
            raise ValueError("all_gather: out has the wrong size")
        self._launch(x.data_ptr(), out.data_ptr(), n, GATHER)
        return out

    def status(self) -> int:
        """0, or bits: 1, 2 = a barrier stage gave up waiting for a peer; 4 = a peer ran a call of another size or
        mode at the same step (synchronises the stream)."""
        return int(self._status.item())

    def check(self) -> None:
        st = self.status()
        if st:
            self._status.zero_()           # reported once: the protocol itself is in step again when the peer arrives
            what = []
            if st & 3:
                what.append(f"a peer did not reach barrier stage(s) {st & 3:#x} within the spin limit")
            if st & 4:
                what.append("a peer launched a collective of a different size or mode (the ranks' schedules differ)")
            raise XgmiError("; ".join(what) + ": results since then are invalid")

    def resync(self) -> None:
        """Collective (every rank, through the exchange the communicator was built with): after a step in which some rank
        launched collectives its peers never launched, the ranks' epoch counters differ for good -- the rank that is behind
        would sail through every barrier on stale flags while the one ahead waits out the limit each time. All ranks drain
        their streams, take the largest epoch any of them reached, and start from there with a clear status word."""
        if self._closed:
            raise XgmiError("communicator is closed")
        torch = self.torch
        torch.cuda.synchronize(self.device)
        mine = int(self._counters.max().item())
        every = [int(v.decode()) for v in self._exchange(str(mine).encode())]
        self._counters.fill_(max(every))
        self._status.zero_()
        torch.cuda.synchronize(self.device)
        self._exchange(b"resynced")              # nobody launches before every rank's counters are set

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        self.torch.cuda.synchronize(self.device)
        for p in self._opened:
            self.L.klm_xgmi_close(p)
        self.L.klm_xgmi_free(self.base)

    def describe(self) -> dict:
        return {"rank": self.rank, "world": self.world, "slot_bytes": self.slot_bytes, "blocks": self.blocks,
                "one_shot_max": self.one_shot_max, "fine_grained": self.fine_grained,
                "distinct_devices": getattr(self, "distinct_devices", None)}
