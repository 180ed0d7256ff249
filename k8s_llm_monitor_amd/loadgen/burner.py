"""Validation load generator: a known, steady, controllable load on one MI355X.

No reference counterpart (BASELINE.json north_star item 4). Kinds:

  gemm_bf16 / gemm_fp8 / gemm_fp4   the hand-written MFMA GEMM kernels (ops.gemm_bf16 / gemm_fp8 / gemm_fp4)
  hbm_copy / hbm_read / hbm_fill   HBM bandwidth burners (known MEM/UMC activity)
  spin                   busy waves with no memory traffic and no matrix work

Duty-cycle control: time is cut into periods of ``period_ms``; in each period kernels are
launched back to back for ``duty`` percent of it (a whole number of kernels, from a
calibrated per-kernel time) and the stream then idles to the period's end. The achieved
duty is measured with HIP events around every busy window, not assumed, and is what the
fidelity check compares amdsmi's GFX activity against.

``HbmHold`` is the HBM exhaustion injector: it allocates and touches N GiB (or fills the
GPU to a target percentage of its 288 GB) so the 70 / 85 / 95 % alert tiers can be
crossed on real hardware, and frees it reliably.
"""
from __future__ import annotations

import json
import threading
import time
from typing import Any, Callable, Dict, List, Optional

KINDS = ("gemm_bf16", "gemm_fp8", "gemm_fp4", "hbm_copy", "hbm_read", "hbm_fill", "spin")
GIB = 1 << 30


def plan_window(duty: float, period_s: float, kernel_s: float) -> int:
    """Kernels per busy window: the whole number whose run time is closest to duty*period,
    at least 1 when duty > 0 (so the achievable duty is quantised by the kernel time)."""
    if duty <= 0:
        return 0
    if kernel_s <= 0:
        raise ValueError("kernel time must be positive")
    busy = min(1.0, duty / 100.0) * period_s
    return max(1, int(round(busy / kernel_s)))


class Burner:
    GRAPH_CHUNK = 32          # kernels per captured hipGraph
    GRAPH_BELOW_S = 100e-6    # "auto": capture when one kernel is shorter than this (launch-bound)

    def __init__(self, kind: str = "gemm_bf16", size: int = 8192, duty: float = 100.0, period_ms: float = 100.0,
                 device: int = 0, hbm_bytes: int = 2 * GIB, graph: Any = "auto"):
        if kind not in KINDS:
            raise ValueError(f"unknown burner kind {kind!r}; have {KINDS}")
        if not (0.0 <= duty <= 100.0):
            raise ValueError("duty must be within 0..100")
        import torch
        from .. import ops
        self.torch, self.ops = torch, ops
        self.kind, self.size, self.duty, self.period_s = kind, size, float(duty), period_ms / 1e3
        self.dev = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=self.dev)
        g = torch.Generator(device=self.dev)
        g.manual_seed(4321)
        self.flops = 0.0          # per kernel
        self.bytes = 0.0          # HBM bytes moved per kernel
        if kind == "gemm_fp4":
            # every byte is two valid e2m1 codes: random bytes are random fp4 operands
            self.a = self._operand(torch.randint(0, 256, (size, size // 2), device=self.dev, generator=g, dtype=torch.uint8))
            self.b = self._operand(torch.randint(0, 256, (size, size // 2), device=self.dev, generator=g, dtype=torch.uint8))
            self.c = torch.empty(size, size, device=self.dev, dtype=torch.bfloat16)
            self.flops = 2.0 * size ** 3
        elif kind in ("gemm_bf16", "gemm_fp8"):
            a = torch.rand(size, size, device=self.dev, generator=g) * 2 - 1
            b = torch.rand(size, size, device=self.dev, generator=g) * 2 - 1
            dt = torch.bfloat16 if kind == "gemm_bf16" else torch.float8_e4m3fn
            self.a, self.b = self._operand(a.to(dt)), self._operand(b.to(dt))
            del a, b
            self.c = torch.empty(size, size, device=self.dev, dtype=torch.bfloat16)
            self.flops = 2.0 * size ** 3
        elif kind.startswith("hbm"):
            n = hbm_bytes - hbm_bytes % 16
            self.x = torch.empty(n, dtype=torch.uint8, device=self.dev)
            ops.hbm_fill(self.x, 0x5A5A5A5A)
            if kind == "hbm_copy":
                self.y = torch.empty(n, dtype=torch.uint8, device=self.dev)
                self.bytes = 2.0 * n
            else:
                self.bytes = float(n)
            self.sink = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.kernel_s = self.calibrate()
        # Short kernels are launch-bound from Python (a ctypes call + a HIP launch per kernel is
        # ~10 us): capture GRAPH_CHUNK launches into one hipGraph and replay that instead.
        self.graph = None
        if graph is True or (graph == "auto" and self.kernel_s < self.GRAPH_BELOW_S):
            self._capture()
            self.kernel_s = self.calibrate()
        self._stop = threading.Event()
        self._th: Optional[threading.Thread] = None
        self._lock = threading.Lock()
        self._stats = self._zero()

    def _operand(self, t):
        """The burner owns its operands, so it may lay them out: when the hand-scheduled stream takes the shape and the dense
        row pitch is one the memory channels camp on (ops.camping_pitch: 12288 bf16 columns = 24 KiB rows run at 1380
        TFLOP/s, with 256 bytes of padding per row at 1617), the operand lives in a padded allocation."""
        row = t.shape[1] * t.element_size()
        if t.shape[0] % 256 or row % 256 or row < 512 or not self.ops.camping_pitch(row):
            return t
        p = self.ops.padded_operand(t.shape[0], t.shape[1], t.dtype, t.device)
        p.copy_(t)
        return p

    # ---- one kernel ---------------------------------------------------------------------
    def launch(self) -> None:
        k, ops = self.kind, self.ops
        if k == "gemm_bf16":
            ops.gemm_bf16(self.a, self.b, out=self.c)
        elif k == "gemm_fp8":
            ops.gemm_fp8(self.a, self.b, out=self.c)
        elif k == "gemm_fp4":
            ops.gemm_fp4(self.a, self.b, out=self.c)
        elif k == "hbm_copy":
            ops.hbm_copy(self.y, self.x)
        elif k == "hbm_read":
            ops.hbm_read(self.x, self.sink)
        elif k == "hbm_fill":
            ops.hbm_fill(self.x, 0x5A5A5A5A)
        else:
            ops.spin(2_400_000, 1024, self.dev)      # ~1 ms (s_memtime ticks ~2.4 GHz here); calibrated anyway

    def _capture(self) -> None:
        torch = self.torch
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.stream):
            self.launch()                       # first launch outside capture (one-time function attributes)
            self.stream.synchronize()
            with torch.cuda.graph(g, stream=self.stream):
                for _ in range(self.GRAPH_CHUNK):
                    self.launch()
        self.graph = g

    def launch_many(self, n: int) -> None:
        """`n` kernels on the burner's stream: whole graph chunks, then single launches."""
        if self.graph is not None:
            while n >= self.GRAPH_CHUNK:
                self.graph.replay()
                n -= self.GRAPH_CHUNK
        for _ in range(n):
            self.launch()

    def calibrate(self, iters: int = 10) -> float:
        torch = self.torch
        with torch.cuda.stream(self.stream):
            for _ in range(3):
                self.launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if getattr(self, "graph", None) is not None:
                iters = max(iters, 4 * self.GRAPH_CHUNK)
            e0.record(self.stream)
            self.launch_many(iters) if hasattr(self, "graph") else [self.launch() for _ in range(iters)]
            e1.record(self.stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / 1e3 / iters

    @staticmethod
    def _zero() -> Dict[str, float]:
        return {"kernels": 0, "busy_s": 0.0, "wall_s": 0.0, "periods": 0}

    # ---- run ----------------------------------------------------------------------------
    def run(self, seconds: float, on_period: Optional[Callable[[Dict[str, Any]], None]] = None) -> Dict[str, Any]:
        """Run the duty-cycled load for `seconds` on this thread and return its stats."""
        torch = self.torch
        full = self.duty >= 100.0
        t_start = time.perf_counter()
        t_end = t_start + seconds
        st = self._zero()
        pending: List[Any] = []
        # closed loop: the per-kernel time moves with clocks and temperature (a cold calibration
        # reads ~15 % slow), so every retired window updates the estimate the next window is sized by
        est = {"k": self.kernel_s}

        def retire(block_on_oldest: bool) -> None:
            if block_on_oldest and pending:
                pending[0][1].synchronize()
            while pending and pending[0][1].query():
                e0, e1, n = pending.pop(0)
                busy = e0.elapsed_time(e1) / 1e3
                st["busy_s"] += busy
                st["kernels"] += n
                if n:
                    est["k"] = 0.7 * est["k"] + 0.3 * busy / n

        with torch.cuda.stream(self.stream):
            k = 0
            while not self._stop.is_set() and time.perf_counter() < t_end:
                if not full:
                    # partial duty: windows start on the period grid
                    p0 = t_start + k * self.period_s
                    now = time.perf_counter()
                    if p0 > now:
                        time.sleep(min(p0 - now, max(0.0, t_end - now)))
                        if time.perf_counter() >= t_end:
                            break
                n_per = plan_window(self.duty, self.period_s, est["k"])
                if n_per:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                    self.launch_many(n_per)
                    e1.record(self.stream)
                    pending.append((e0, e1, n_per))
                    if full:
                        retire(block_on_oldest=len(pending) >= 4)    # keep the queue fed, never drained
                    else:
                        e1.synchronize()                              # the idle part starts after the window
                        retire(False)
                else:
                    time.sleep(min(self.period_s, max(0.0, t_end - time.perf_counter())))
                st["periods"] += 1
                k += 1
                if on_period is not None:
                    on_period(st)
            while pending:
                retire(block_on_oldest=True)
        self.kernel_s = est["k"]
        st["wall_s"] = time.perf_counter() - t_start
        with self._lock:
            self._stats = st
        return self.summary(st)

    def summary(self, st: Optional[Dict[str, float]] = None) -> Dict[str, Any]:
        st = st or self._stats
        wall, busy = max(st["wall_s"], 1e-9), max(st["busy_s"], 1e-12)
        out = {"kind": self.kind, "size": self.size, "duty_target": self.duty, "period_ms": self.period_s * 1e3,
               "hip_graph": self.graph is not None,
               "kernel_ms": self.kernel_s * 1e3, "kernels": st["kernels"], "wall_s": wall,
               "duty_achieved": min(100.0, st["busy_s"] / wall * 100.0)}
        if self.flops:
            out["tflops_busy"] = self.flops * st["kernels"] / busy / 1e12
            out["tflops_avg"] = self.flops * st["kernels"] / wall / 1e12
        if self.bytes:
            out["TBps_busy"] = self.bytes * st["kernels"] / busy / 1e12
            out["TBps_avg"] = self.bytes * st["kernels"] / wall / 1e12
        return out

    def start(self, seconds: float = 3600.0) -> "Burner":
        self._stop.clear()
        self._th = threading.Thread(target=self.run, args=(seconds,), name="klm-burner", daemon=True)
        self._th.start()
        return self

    def stop(self) -> Dict[str, Any]:
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=30)
            self._th = None
        self.torch.cuda.synchronize(self.dev)
        return self.summary()


class HbmHold:
    """Allocate and touch HBM in chunks; ``release()`` (or the context manager) frees it."""

    def __init__(self, device: int = 0, chunk_gib: float = 4.0):
        import torch
        from .. import ops
        self.torch, self.ops = torch, ops
        self.dev = torch.device("cuda", device)
        self.chunk = int(chunk_gib * GIB) & ~0xF
        self.blocks: List[Any] = []

    def held_bytes(self) -> int:
        return sum(b.numel() for b in self.blocks)

    def add(self, nbytes: int) -> int:
        """Allocate and touch `nbytes` more. Returns the bytes actually added (stops at OOM)."""
        torch = self.torch
        added = 0
        while added < nbytes:
            n = min(self.chunk, nbytes - added) & ~0xF
            if n <= 0:
                break
            try:
                b = torch.empty(n, dtype=torch.uint8, device=self.dev)
            except torch.OutOfMemoryError:
                break
            self.ops.hbm_fill(b, 0xA5A5A5A5)      # touch every page: an untouched allocation may not be backed
            self.blocks.append(b)
            added += n
        torch.cuda.synchronize(self.dev)
        return added

    def drop(self, nbytes: int) -> int:
        """Free at least `nbytes` again (whole blocks, newest first) and hand them back to the driver. Returns the bytes freed."""
        freed = 0
        while self.blocks and freed < nbytes:
            freed += self.blocks.pop().numel()
        self.torch.cuda.synchronize(self.dev)
        self.torch.cuda.empty_cache()
        return freed

    def fill_to_percent(self, percent: float, total_bytes: Optional[int] = None, used_bytes: Optional[int] = None) -> int:
        """Grow the hold until device-wide HBM use reaches `percent` of the total (as the driver
        reports it through ``mem_get_info``, which is what amdsmi's VRAM figure tracks)."""
        torch = self.torch
        free, total = torch.cuda.mem_get_info(self.dev)
        total = total_bytes or total
        used = used_bytes if used_bytes is not None else total - free
        want = int(total * percent / 100.0) - used
        return self.add(want) if want > 0 else 0

    def verify(self) -> bool:
        """The held region still holds its pattern (nothing else scribbled on it)."""
        for b in self.blocks[:2]:
            if self.ops.hbm_checksum(b) != (0xA5A5A5A5 * (b.numel() // 4)) & 0xFFFFFFFFFFFFFFFF:
                return False
        return True

    def release(self) -> None:
        self.blocks.clear()
        self.torch.cuda.synchronize(self.dev)
        self.torch.cuda.empty_cache()

    def __enter__(self) -> "HbmHold":
        return self

    def __exit__(self, *exc) -> None:
        self.release()


def main_from_args(a) -> int:
    b = Burner(a.kind, size=a.size, duty=a.duty, period_ms=a.period_ms, device=a.device)
    hold = None
    if a.hbm_gib > 0:
        hold = HbmHold(a.device)
        got = hold.add(int(a.hbm_gib * GIB))
        print(json.dumps({"hbm_hold_gib": got / GIB}), flush=True)
    last = [time.time()]

    def tick(st):
        if a.json and time.time() - last[0] >= 1.0:
            last[0] = time.time()
            print(json.dumps({"t": last[0], "kernels": st["kernels"], "busy_s": round(st["busy_s"], 4)}), flush=True)

    product = None
    if getattr(a, "monitored", False):
        # the whole monitor beside the load (validation/phases.py::Product): what rocprofv3 compares kernel times under
        from ..validation.phases import Product
        product = Product(a.monitor_period, pollers=4)
    try:
        out = b.run(a.seconds, on_period=tick)
    finally:
        if hold is not None:
            hold.release()
        if product is not None:
            out_m = product.stop()
            print(json.dumps({"monitor": out_m}), flush=True)
    print(json.dumps(out), flush=True)
    return 0
