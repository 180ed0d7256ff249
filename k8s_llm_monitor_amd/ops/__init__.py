"""Python bindings of the hand-written gfx950 kernels (``libklm_ops.so``, C ABI, ctypes).

The kernels take raw device pointers and a HIP stream, so they run on PyTorch-ROCm
tensors and on the current torch stream without a torch C++ extension. There is no
PyTorch fallback: on a machine with a GPU a missing library is an error, never a silent
eager path (the library can be built in-tree with ``python -m k8s_llm_monitor_amd.build``).

No reference counterpart: the reference has no native code; these are the
load generators of SURVEY.md 2.4 (N2 GEMM burner, N3 HBM fill, N4 bandwidth burner).
"""
from __future__ import annotations

import ctypes
import functools
import threading
from pathlib import Path
from typing import Optional

_LIB_PATH = Path(__file__).resolve().parent.parent / "_native" / "lib" / "libklm_ops.so"
_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()

STATUS = {0: "ok", -1: "shape not a multiple of the tile", -2: "pointer not 16-byte aligned",
          -3: "unknown variant", -4: "HIP launch error", -5: "bad argument"}


class KlmOpsError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def available() -> bool:
    """True when the compiled library exists (it may still need a GPU to run)."""
    return _LIB_PATH.exists()


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load ``libklm_ops.so``; build it with hipcc when it is missing. Raises if neither works."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not _LIB_PATH.exists() and build_if_missing:
            from .. import build as _build
            _build.build_all(only=["klm_ops"], verbose=False)
        if not _LIB_PATH.exists():
            raise KlmOpsError(f"{_LIB_PATH} is missing: run `python -m k8s_llm_monitor_amd.build`")
        lib = ctypes.CDLL(str(_LIB_PATH))
        vp, i, sz, u32, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64
        lib.klm_gemm_bf16.argtypes = [vp, vp, vp, i, i, i, i, vp]
        lib.klm_gemm_bf16.restype = i
        lib.klm_gemm_persistent_grid.argtypes = [i]
        lib.klm_gemm_persistent_grid.restype = i
        lib.klm_gemm_bf16_pitched.argtypes = [vp, vp, vp, i, i, i, i, i, vp]
        lib.klm_gemm_bf16_pitched.restype = i
        lib.klm_gemm_mx_pitched.argtypes = [i, vp, vp, vp, i, i, i, i, i, vp]
        lib.klm_gemm_mx_pitched.restype = i
        if hasattr(lib, "klm_gemm_stream"):
            lib.klm_gemm_stream.argtypes = [i, vp, vp, vp, i, i, i, i, i, i, i, vp, sz, vp]
            lib.klm_gemm_stream.restype = i
            lib.klm_gemm_stream_plan.argtypes = [i, i, i, i, i, i, sz, ctypes.POINTER(ctypes.c_int)]
            if hasattr(lib, "klm_gemm_pick"):
                lib.klm_gemm_pick.argtypes = [i, i, i, i]
                lib.klm_gemm_pick.restype = i
            lib.klm_gemm_stream_plan.restype = i
            lib.klm_gemm_has_diag.restype = i
        if hasattr(lib, "klm_gemm_fp8"):
            lib.klm_gemm_fp8.argtypes = [vp, vp, vp, i, i, i, i, vp]
            lib.klm_gemm_fp8.restype = i
        if hasattr(lib, "klm_gemm_fp4"):
            lib.klm_gemm_fp4.argtypes = [vp, vp, vp, i, i, i, i, vp]
            lib.klm_gemm_fp4.restype = i
        lib.klm_hbm_fill.argtypes = [vp, sz, u32, vp]
        lib.klm_hbm_fill.restype = i
        lib.klm_hbm_copy.argtypes = [vp, vp, sz, vp]
        lib.klm_hbm_copy.restype = i
        if hasattr(lib, "klm_hbm_copy_ex"):
            lib.klm_hbm_copy_ex.argtypes = [vp, vp, sz, i, i, vp]
            lib.klm_hbm_copy_ex.restype = i
            lib.klm_hbm_fill_ex.argtypes = [vp, sz, u32, i, i, vp]
            lib.klm_hbm_fill_ex.restype = i
        if hasattr(lib, "klm_hbm_read_ex"):
            lib.klm_hbm_read_ex.argtypes = [vp, sz, vp, i, i, vp]
            lib.klm_hbm_read_ex.restype = i
        lib.klm_hbm_read.argtypes = [vp, sz, vp, vp]
        lib.klm_hbm_read.restype = i
        if hasattr(lib, "klm_hbm_repitch"):
            lib.klm_hbm_repitch.argtypes = [vp, sz, vp, sz, i, sz, vp]
            lib.klm_hbm_repitch.restype = i
        lib.klm_hbm_checksum.argtypes = [vp, sz, vp, vp]
        lib.klm_hbm_checksum.restype = i
        lib.klm_spin.argtypes = [u64, i, vp]
        lib.klm_spin.restype = i
        lib.klm_version.restype = ctypes.c_char_p
        if hasattr(lib, "klm_src_hash"):
            lib.klm_src_hash.restype = ctypes.c_char_p
        if hasattr(lib, "klm_hbm_launch_plan"):
            lib.klm_hbm_launch_plan.argtypes = [sz, i, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]
            lib.klm_hbm_launch_plan.restype = i
        _lib = lib
        return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise KlmOpsError(f"{what}: {STATUS.get(rc, rc)} (status {rc})")


def _stream_of(t) -> int:
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def _on_device(t):
    """The launchers act on the CURRENT HIP device (per-device kernel attributes, the launch itself): make that the
    tensor's device for the duration of the call, so `--device 1` in a process whose current device is 0 works."""
    import torch
    return torch.cuda.device(t.device)


def _require_cuda(*ts) -> None:
    for t in ts:
        if not t.is_cuda:
            raise KlmOpsError("klm ops run on the GPU only (tensor is on %s)" % t.device)
        if not t.is_contiguous():
            raise KlmOpsError("klm ops need contiguous tensors")


def _require_rows(*ts) -> None:
    """GEMM operands: on the GPU, rows contiguous (stride 1 along K); the rows themselves may be pitched."""
    for t in ts:
        if not t.is_cuda:
            raise KlmOpsError("klm ops run on the GPU only (tensor is on %s)" % t.device)
        if t.dim() == 2 and not t.is_contiguous() and not (t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
            raise KlmOpsError("klm ops need contiguous tensors (GEMM operands: contiguous rows, optionally pitched)")


def _pitch_bytes(a, b) -> int:
    """0 for dense operands; the common row pitch in bytes when a / b are row-pitched views (`padded_operand`)."""
    if a.dim() != 2 or b.dim() != 2 or (a.is_contiguous() and b.is_contiguous()):
        return 0
    pa, pb = a.stride(0) * a.element_size(), b.stride(0) * b.element_size()
    if pa != pb or pa % 16:
        raise KlmOpsError(f"pitched GEMM operands need one row pitch, a multiple of 16 bytes (got {pa} and {pb})")
    return pa


#: row pitches (bytes) the memory channels camp on: all rows of a 256-row panel start in the same few channels, and the
#: GEMMs' 128-byte-wide column reads queue there (8192 x 8192 x 12288 bf16, pitch 24 KiB: 1400 TFLOP/s; 12032: 1591).
#: Measured bad: multiples of 2 - 4 KiB that are not a power of two; 256 bytes of padding per row is enough
#: (scripts/exp/gemm_stride_sweep.py, gemm_pitch_check.py). Power-of-two pitches are fine (the channel hash handles them).
def camping_pitch(nbytes: int) -> bool:
    return nbytes >= 4096 and nbytes % 2048 == 0 and (nbytes & (nbytes - 1)) != 0


def padded_operand(rows: int, cols: int, dtype, device, pad_bytes: int = 256):
    """An uninitialised [rows, cols] GEMM operand; when its dense row pitch is one the memory channels camp on
    (`camping_pitch`) it is a view into a [rows, cols + pad] allocation, which the hand-scheduled GEMM streams take as is."""
    import torch
    es = torch.empty((), dtype=dtype).element_size()
    if not camping_pitch(cols * es):
        return torch.empty((rows, cols), dtype=dtype, device=device)
    return torch.empty((rows, cols + pad_bytes // es), dtype=dtype, device=device)[:, :cols]


def version() -> str:
    return load().klm_version().decode()


def src_hash() -> str:
    """The hash of the sources the LOADED library was built from ("unhashed": built by other means than build.py)."""
    lib = load()
    return lib.klm_src_hash().decode() if hasattr(lib, "klm_src_hash") else "unhashed"


def tree_hash() -> str:
    """The hash of the sources in the tree, as build.py computes it for libklm_ops.so."""
    from .. import build as _build
    return next(t for t in _build.targets() if t.name == "klm_ops").source_hash()


def assert_current() -> None:
    """Raise unless the loaded library was built from the sources in this tree: a stale `.so` (a snapshot copy or a checkout
    rewrote the mtimes, the build was skipped) would otherwise pass tests of code that is not the code in the tree."""
    have, want = src_hash(), tree_hash()
    if have != want:
        raise KlmOpsError(f"{_LIB_PATH} was built from other sources (library src:{have}, tree src:{want}): "
                          "run `python -m k8s_llm_monitor_amd.build`")


def hbm_launch_plan(nbytes: int, accesses_per_thread: int = 1):
    """(launches, first grid, last grid) the default HBM kernels use for a buffer of `nbytes` (HIP's grid limit is 2^32 - 1
    threads per launch: 64 GiB at one 16-byte access per thread; larger buffers are walked in several launches)."""
    first, last = ctypes.c_uint(0), ctypes.c_uint(0)
    n = load().klm_hbm_launch_plan(int(nbytes), int(accesses_per_thread), ctypes.byref(first), ctypes.byref(last))
    _check(min(n, 0), "klm_hbm_launch_plan")
    return n, first.value, last.value


#: the cost model behind the default kernel choice (microseconds; fitted to profiles/r3_prof/gemm_small.json and
#: gemm_units_first.json). A workgroup's time per BYTE of a row's K extent on a quarter-full chip and on a full one (fewer busy
#: CUs: higher clocks, more L2 bandwidth each); per byte, because both kernels stage 128-byte K tiles whatever the format:
_STREAM_US_PER_KB = (0.0073, 0.0106)    # 256 x 256 tile, persistent stream (bf16 10, fp8 7, fp4 6)
_P128_US_PER_KB = (0.0020, 0.0039)      # 128 x 128 tile, eight-wave ring kernel (14)
_P256_US_PER_KB = (0.0056, 0.0070)      # 256 x 128 / 128 x 256 tile, the same kernel (16 / 17): 7 % less per FLOP than 14 on a full chip
_P128_SECOND_ROUND_US_PER_KB = 0.0032   # measured: 2304^3, 2560^3, 1024 x 6144 x 4096 (profiles/r3_prof/gemm_mid.json)
_STREAM_FIXED_US, _P128_FIXED_US = 8.0, 6.5
_SPLIT_PASS_US, _SPLIT_MB_US, _SPLIT_LAUNCH_US = 2.2, 0.30, 8.0      # klm_quad_plan's decision rule (gemm_quad_asm.h)
_SPLIT_TIME_MB_US = 0.13               # what the partial sums were measured to add (64 MB: 8 us on top of the reduce launch)


def _per_kb(pair, occupancy: float, power: int = 1) -> float:
    lo, hi = pair
    return lo + (hi - lo) * min(1.0, max(0.0, (occupancy - 0.25) / 0.75)) ** power


def stream_plan_py(M: int, N: int, Kbytes: int, cus: int = 256, mode: int = 1):
    """klm_quad_plan's arithmetic (rounds, rest, split) without the library: what the default kernel choice prices."""
    ntiles, P = (M // 256) * (N // 256), Kbytes // 256
    rounds, rest = divmod(ntiles, cus)
    S = 0
    if mode and rest:
        s = min(cus // rest, 8, P // 2)
        if s > 1:
            saved = _SPLIT_PASS_US * (P - -(-P // s))
            cost = _SPLIT_LAUNCH_US + _SPLIT_MB_US * rest * s * 0.25
            if mode == 2 or saved > cost:
                S = s
    return rounds, rest, S


def stream_time_us(M: int, N: int, Kbytes: int, cus: int = 256) -> float:
    """Modelled time of the persistent 256 x 256 stream on a problem whose rows are `Kbytes` long."""
    rounds, rest, S = stream_plan_py(M, N, Kbytes, cus)
    t = _STREAM_FIXED_US + rounds * Kbytes * _STREAM_US_PER_KB[1]
    if rest:
        if S > 1:
            t += Kbytes * _per_kb(_STREAM_US_PER_KB, rest * S / cus, 2) / S + _SPLIT_LAUNCH_US + _SPLIT_TIME_MB_US * rest * S * 0.25
        else:
            t += Kbytes * _per_kb(_STREAM_US_PER_KB, rest / cus, 2)      # measured convex in the occupancy (3072^3, 3584^3)
    return t


def p128_time_us(M: int, N: int, Kbytes: int, cus: int = 256) -> float:
    """Modelled time of the eight-wave 128 x 128 ring kernel (one workgroup per CU at a time: 128 KiB of LDS each)."""
    rounds, rest = divmod((M // 128) * (N // 128), cus)
    per = _per_kb(_P128_US_PER_KB, rest / cus)
    if rounds:
        per = max(per, _P128_SECOND_ROUND_US_PER_KB)      # a partial round BEHIND a full one never runs at the empty chip's rate
    return _P128_FIXED_US + rounds * Kbytes * _P128_US_PER_KB[1] + (Kbytes * per if rest else 0.0)


def p256_time_us(M: int, N: int, Kbytes: int, cus: int = 256) -> float:
    """Modelled time of the ring kernel on 256 x 128 (or 128 x 256) tiles: three 48 KiB stages, one workgroup per CU."""
    rounds, rest = divmod((M * N) // (256 * 128), cus)
    return _P128_FIXED_US + rounds * Kbytes * _P256_US_PER_KB[1] + (Kbytes * _per_kb(_P256_US_PER_KB, rest / cus) if rest else 0.0)


def p256_shape_ok(M: int, N: int, Kbytes: int) -> bool:
    return p128_shape_ok(M, N, Kbytes) and (M % 256 == 0 or N % 256 == 0)


def stream_shape_ok(M: int, N: int, Kbytes: int) -> bool:
    return M > 0 and N > 0 and M % 256 == 0 and N % 256 == 0 and Kbytes % 256 == 0 and Kbytes >= 512


def p128_shape_ok(M: int, N: int, Kbytes: int) -> bool:
    return M > 0 and N > 0 and M % 128 == 0 and N % 128 == 0 and Kbytes % 128 == 0 and Kbytes >= 128


@functools.lru_cache(maxsize=4096)
def pick_tile_kernel(M: int, N: int, Kbytes: int, cus: int = 256) -> Optional[str]:
    """"stream" (256 x 256 persistent), "p128" / "p256" (the eight-wave ring kernel on 128 x 128 / 256 x 128 tiles) or None (none
    takes the shape): whichever the cost model prices lowest. Pure: no device, no library."""
    cands = []
    if stream_shape_ok(M, N, Kbytes):
        cands.append((stream_time_us(M, N, Kbytes, cus), 0, "stream"))
    if p256_shape_ok(M, N, Kbytes):
        cands.append((p256_time_us(M, N, Kbytes, cus), 1, "p256"))
    if p128_shape_ok(M, N, Kbytes):
        cands.append((p128_time_us(M, N, Kbytes, cus), 2, "p128"))
    return min(cands)[2] if cands else None


def _ring_fallback(M: int, N: int, Kbytes: int, cus: int) -> bool:
    """Whole 256 x 256 tiles with an odd number of K tiles: the 256 x 256 ring kernel's schedule 0 takes them, and wins once
    its tiles fill half the chip."""
    return (not stream_shape_ok(M, N, Kbytes) and M % 256 == 0 and N % 256 == 0 and Kbytes >= 256 and (M // 256) * (N // 256) >= cus // 2)


def native_pick(M: int, N: int, Kbytes: int, cus: int = 256) -> Optional[str]:
    """The library's own copy of `pick_tile_kernel` (klm_gemm_pick: what klm_gemm_auto and the C++ burner dispatch by)."""
    return {0: None, 1: "stream", 2: "p128", 3: "p256"}[int(load().klm_gemm_pick(M, N, Kbytes, cus))]


def pick_bf16_variant(M: int, N: int, K: int, pitch: int = 0, cus: int = 256) -> int:
    """The kernel `gemm_bf16(variant=None)` runs for a shape (pure: no device, no library): the persistent 256 x 256 stream
    (10) or, where too few 256 x 256 tiles leave most of the chip idle, the eight-wave ring kernel on 128 x 128 tiles (14: 2048^3
    807 TFLOP/s against 465) or on 256 x 128 / 128 x 256 tiles (16 / 17: 2560^3 43.4 -> 38.8 us), whichever the cost model prices
    lowest; the 256 x 256 ring / old 128 x 128 kernels for shapes none of them takes."""
    if pitch:
        return 10              # row-pitched operands: the streams are the kernels that take a pitch
    kind = pick_tile_kernel(M, N, 2 * K, cus)
    if kind == "stream":
        return 10
    if kind in ("p128", "p256"):
        if _ring_fallback(M, N, 2 * K, cus):
            return 1
        return 14 if kind == "p128" else (16 if M % 256 == 0 else 17)
    return 0


def pick_mx_variant(M: int, N: int, Kbytes: int, pitch: int, stream_variant: int, cus: int = 256) -> int:
    """fp8 / fp4 (`Kbytes`: a row's bytes): the persistent stream with its plan, the ring kernel on 128 x 128 (14) or 128 x 256 /
    256 x 128 tiles (17 / 16) where the model prices it lower, else the 256 x 256 ring kernel (1)."""
    if pitch:
        return stream_variant
    kind = pick_tile_kernel(M, N, Kbytes, cus)
    if kind in ("p128", "p256") and _ring_fallback(M, N, Kbytes, cus):
        return 1
    if kind == "p256":
        return 17 if N % 256 == 0 else 16
    return {"stream": stream_variant, "p128": 14, None: 1}[kind]


#: fp32 partial sums of one K slice of one 256 x 256 tile
SLOT_BYTES = 256 * 256 * 4
_ws_cache: dict = {}


def stream_workspace(device, nbytes: int = 0):
    """The scratch the persistent streams cut their last partial round into K slices with: one buffer per (device, stream),
    sized for the worst case (one slot per CU), allocated on first use through torch's caching allocator and kept."""
    import torch
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is not None and ws.numel() >= nbytes:
        return ws
    need = max(int(nbytes), torch.cuda.get_device_properties(dev).multi_processor_count * SLOT_BYTES)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    return ws


_repack_cache: dict = {}

#: Dense bf16 operands whose row pitch is a non-power-of-two multiple of 8 KiB (24 KiB = K 12288, 40 KiB = K 20480) cost the
#: GEMM 17 - 21 % (8192 x 8192 x 12288: 1381 TFLOP/s dense, 1617 on rows 256 bytes further apart; the milder camping pitches --
#: 20 KiB, 28 KiB: 7 % -- are not worth a copy: profiles/r2_prof/gemm_pitch_check.json, profiles/r3_prof/gemm_repack.json).
#: From this many rows on, `gemm_bf16(variant=None)` copies both operands into padded scratch first (6.4 TB/s, one launch each).
REPACK_MIN_ROWS = 6144


def repack_worthwhile(M: int, N: int, Kbytes: int, dtype_bytes: int = 2) -> bool:
    """True when `gemm_bf16(variant=None)` re-pitches dense operands before the GEMM (pure)."""
    bad_pitch = Kbytes >= 24576 and Kbytes % 8192 == 0 and (Kbytes & (Kbytes - 1)) != 0
    return dtype_bytes == 2 and bad_pitch and min(M, N) >= REPACK_MIN_ROWS and M % 256 == 0 and N % 256 == 0


def repitched(t, role: str = "a", pad_bytes: int = 256):
    """A row-padded copy of the 2-D tensor `t` in a scratch buffer kept per (device, stream, role): the persistent streams
    take it as a pitched operand. One 16-byte load + store per thread (klm_hbm_repitch)."""
    import torch
    _require_cuda(t)
    rows, cols = t.shape
    es = t.element_size()
    key = (t.device.index, torch.cuda.current_stream(t.device).cuda_stream, role)
    pitch_elems = cols + pad_bytes // es
    buf = _repack_cache.get(key)
    if buf is None or buf.dtype != t.dtype or buf.numel() < rows * pitch_elems:
        buf = torch.empty(rows * pitch_elems, dtype=t.dtype, device=t.device)
        _repack_cache[key] = buf
    view = buf[: rows * pitch_elems].view(rows, pitch_elems)[:, :cols]
    with _on_device(t):
        _check(load().klm_hbm_repitch(view.data_ptr(), pitch_elems * es, t.data_ptr(), cols * es, rows, cols * es, _stream_of(t)), "klm_hbm_repitch")
    return view


def gemm_has_diag() -> bool:
    """True when the library carries the A/B-record and diagnostic streams (a -DKLM_GEMM_DIAG build, never the shipped one)."""
    lib = load()
    return bool(hasattr(lib, "klm_gemm_has_diag") and lib.klm_gemm_has_diag())


def gemm_stream_plan(M: int, N: int, Kbytes: int, cus: int = 0, mode: int = 1, rot: int = 0, ws_bytes: int = 1 << 40) -> dict:
    """How the persistent stream runs an M x N problem whose rows hold `Kbytes` bytes of K on `cus` workgroups (0: this
    device's CUs): `rounds` full rounds of whole tiles on `grid` workgroups, then `rest` tiles, each cut into `split` K slices
    (0: run whole); `rot`: the K-loop stagger in passes; `map`: 1 = the super-block tile order (mode + 4 asks for it)."""
    out = (ctypes.c_int * 6)()
    _check(load().klm_gemm_stream_plan(M, N, Kbytes, cus, mode, rot, ws_bytes, out), f"klm_gemm_stream_plan({M},{N},{Kbytes})")
    return dict(zip(("grid", "rounds", "rest", "split", "rot", "map"), (int(x) for x in out)))


#: K-loop stagger for dense operands whose row pitch the memory channels camp on: the workgroups of XCD x start their K loop
#: x * CAMPING_ROT passes (of two K tiles, 256 bytes of a row) in; -1 = an eighth of the K extent per XCD
CAMPING_ROT = -1
#: what `variant=None` does on a camping pitch: 0 = nothing (set from the measurement in scripts/exp/gemm_units.py)
AUTO_ROT = 0
#: the default streams' plan mode: 1 = split the partial round where it pays; + 4 = the super-block tile order where the shape allows
STREAM_MODE = 1


def _stream(dtype_code: int, a, b, out, M, N, Kbytes, pitch, store=0, mode=1, rot=None, what="gemm"):
    lib = load()
    ld = pitch or Kbytes
    if rot is None:
        rot = AUTO_ROT if (not pitch and camping_pitch(Kbytes)) else 0
    if rot < 0:
        rot = max(1, (Kbytes // 256) // 8)
    ws = stream_workspace(a.device) if mode else None
    with _on_device(a):
        rc = lib.klm_gemm_stream(dtype_code, a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, Kbytes, ld if pitch else 0, store, mode, rot,
                                 ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream_of(a))
    _check(rc, f"klm_gemm_stream({what},M={M},N={N},Kbytes={Kbytes},pitch={pitch},mode={mode},rot={rot})")
    return out


def gemm_bf16(a, b, out=None, variant: Optional[int] = None):
    """``out[M,N] = a[M,K] @ b[N,K].T`` in bf16 with fp32 accumulation on the MFMA kernels.

    variant None picks the fastest kernel the shape allows (`pick_bf16_variant`, a cost model fitted to measurements):
      10  the hand-scheduled 4-wave 256 x 256 stream in its persistent form with its plan (`gemm_stream_plan`): one workgroup
          per CU walks units (K ranges of output tiles), the next unit's operands fetched under the epilogue; a last, partial
          round of tiles is cut into K slices (fp32 partial sums + a reduce kernel) where that pays; a one-round problem stores
          C past the caches; operands well past the Infinity Cache are walked in 16 x 16 super-blocks of tiles; dense
          operands on a badly camping row pitch are copied into padded scratch first (`repack_worthwhile`)
      14 / 16 / 17  the eight-wave LDS-DMA ring kernel on 128 x 128 / 256 x 128 / 128 x 256 tiles, for shapes of too few
          256 x 256 tiles to fill the chip
      1   the 8-wave 256 x 256 ring kernel (whole 256 x 256 tiles with an odd number of K tiles); 0: the round-1 128 x 128 kernel
    Explicit variants: 8 the persistent stream without a plan, 13 the same through the plan machinery, 11 the split forced
    wherever the shape allows, 12 the K loops staggered per XCD, 15 the super-block order forced, 4 the stream with one workgroup
    per tile (what `LinearLayerStep` runs under data parallelism), 9: 8 with C stored past the caches (`sc1 nt`), 2 the ring
    kernel's schedule 1, 3 the compiler-scheduled 4-wave kernel (gemm_quad128.h). Diagnostic builds (-DKLM_GEMM_DIAG) add 5 .. 7
    (earlier placements of the stream), 48 / 49 (8 without its stores / epilogue) and 40 .. 43, whose RESULT IS MEANINGLESS
    (scripts/exp/quad_asm_ablate.py).
    """
    import torch
    _require_rows(a, b)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise KlmOpsError("gemm_bf16 needs bfloat16 inputs")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise KlmOpsError(f"gemm_bf16 shapes {tuple(a.shape)} x {tuple(b.shape)}^T do not match")
    M, K = a.shape
    N = b.shape[0]
    pitch = _pitch_bytes(a, b)
    if variant is None:
        variant = pick_bf16_variant(M, N, K, pitch)
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    else:
        _require_cuda(out)
        if out.shape != (M, N) or out.dtype != torch.bfloat16:
            raise KlmOpsError("gemm_bf16: bad `out`")
    if variant == 10 and not pitch and repack_worthwhile(M, N, 2 * K):
        # dense operands on a row pitch the memory channels camp on, and a problem large enough to pay for a copy: both
        # operands go through padded scratch (the same bits as on the dense operands: the pitch does not enter the sum)
        a, b = repitched(a, "a"), repitched(b, "b")
        pitch = _pitch_bytes(a, b)
    if variant in (10, 11, 12, 13, 15):
        # the persistent stream with its plan: 10 the default (the partial round split into K slices where that pays, the K loops
        # staggered on a camping pitch); 11: split wherever the shape allows, 12: stagger always, 13: neither (= 8);
        # 15: 10 with the super-block tile order where the shape allows it
        return _stream(0, a, b, out, M, N, 2 * K, pitch, mode={10: STREAM_MODE, 11: 2, 12: 0, 13: 0, 15: 5}[variant],
                       rot={10: None, 11: 0, 12: CAMPING_ROT, 13: 0, 15: None}[variant], what="bf16")
    with _on_device(a):
        if pitch:
            # row-pitched operands (padded_operand): the hand-scheduled streams only
            if variant not in (4, 8, 9):
                raise KlmOpsError(f"gemm_bf16: pitched operands need the hand-scheduled stream (variant 4 / 8 / 9 / 10), shape {M}x{N}x{K} gives {variant}")
            rc = load().klm_gemm_bf16_pitched(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, pitch, {4: 0, 8: 1, 9: 2}[variant], _stream_of(a))
        else:
            rc = load().klm_gemm_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, variant, _stream_of(a))
    _check(rc, f"klm_gemm_bf16(M={M},N={N},K={K},variant={variant},pitch={pitch})")
    return out


def gemm_persistent_grid(grid: int = -1) -> int:
    """Workgroup count of the persistent GEMM kernels (bf16 variant 8, fp8 4, fp4 3): 0 = one per CU (the default), a
    positive value for tests that want several output tiles per workgroup on a small problem; < 0 only reads.
    Returns the previous setting."""
    return int(load().klm_gemm_persistent_grid(int(grid)))


def gemm_fp8(a, b, out=None, variant: Optional[int] = None):
    """``out[M,N] (bf16) = a[M,K] @ b[N,K].T`` with OCP e4m3 inputs on the f8f6f4 MFMA (16x16x128).
    variant None: the hand-scheduled 4-wave stream in its persistent form with its plan (7; 4 without the plan) when K is a
    multiple of 256 and >= 512, or the eight-wave 128 x 128 ring kernel (14) where fewer 256 x 256 tiles than CUs make it
    the faster one (`pick_mx_variant`), else the ring kernel (1: schedule 1 when the K tile count allows, else schedule 0; 0: schedule 0). 2: the stream with one
    workgroup per tile; 3: another placement of stream 2; 5: 4 with C stored past the caches (see gemm_bf16, variant 9);
    6: 4 on the block-scaled MFMA with unit scales (the streams use the plain v_mfma_f32_16x16x128_f8f6f4: same bits, one
    instruction word instead of two, +1.6 %; the ring kernels keep the block-scaled form)."""
    import torch
    _require_rows(a, b)
    if a.dtype != torch.float8_e4m3fn or b.dtype != torch.float8_e4m3fn:
        raise KlmOpsError("gemm_fp8 needs float8_e4m3fn (OCP) inputs")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise KlmOpsError("gemm_fp8 shapes do not match")
    M, K = a.shape
    N = b.shape[0]
    if variant is None:
        variant = pick_mx_variant(M, N, K, _pitch_bytes(a, b), 7)
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    lib = load()
    if not hasattr(lib, "klm_gemm_fp8"):
        raise KlmOpsError("libklm_ops.so was built without the fp8 kernel")
    pitch = _pitch_bytes(a, b)
    if variant in (7, 8, 9):
        # 7: the persistent stream with its plan (as bf16 variant 10); 8: the partial round split wherever the shape allows;
        # 9: 7 with the super-block tile order where the shape allows it
        return _stream(1, a, b, out, M, N, K, pitch, mode={7: STREAM_MODE, 8: 2, 9: 5}[variant], rot=None if variant != 8 else 0, what="fp8")
    with _on_device(a):
        if pitch:
            if variant not in (2, 4, 5):
                raise KlmOpsError(f"gemm_fp8: pitched operands need the hand-scheduled stream (variant 2 / 4 / 5), K={K} gives {variant}")
            rc = lib.klm_gemm_mx_pitched(1, a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, pitch, {2: 0, 4: 1, 5: 2}[variant], _stream_of(a))
        else:
            rc = lib.klm_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, variant, _stream_of(a))
    _check(rc, f"klm_gemm_fp8(M={M},N={N},K={K},variant={variant},pitch={pitch})")
    return out


#: the 16 e2m1 code points (sign, 2-bit exponent, 1-bit mantissa), index = the 4-bit code
FP4_VALUES = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0)


def fp4_decode(packed):
    """uint8 [.., K/2] (two e2m1 codes per byte, low nibble first) -> float32 [.., K]."""
    import torch
    lut = torch.tensor(FP4_VALUES, dtype=torch.float32, device=packed.device)
    p = packed.view(torch.uint8).to(torch.int64)
    lo, hi = lut[p & 15], lut[p >> 4]
    return torch.stack((lo, hi), dim=-1).reshape(*packed.shape[:-1], packed.shape[-1] * 2)


def gemm_fp4(a, b, out=None, variant: Optional[int] = None):
    """``out[M,N] (bf16) = A[M,K] @ B[N,K].T`` with e2m1 (fp4) inputs packed two per byte:
    `a` is uint8 (or ``float4_e2m1fn_x2``) of shape [M, K/2], `b` of shape [N, K/2].
    variant None: the hand-scheduled 4-wave stream in its persistent form with its plan (6; 3 without) when the packed row
    is a multiple of 256 bytes and >= 512, or the eight-wave 128 x 128 ring kernel (14) where `pick_mx_variant` prices it
    lower, else the ring kernel (1: schedule 1 where the tile count allows; 0: schedule 0). 2: the stream with
    one workgroup per tile; 4: 3 with C stored past the caches; 5: 3 on the block-scaled MFMA with unit scales (same bits,
    -3.3 %)."""
    import torch
    _require_rows(a, b)
    ok = (torch.uint8, getattr(torch, "float4_e2m1fn_x2", torch.uint8))
    if a.dtype not in ok or b.dtype not in ok:
        raise KlmOpsError("gemm_fp4 needs packed uint8 / float4_e2m1fn_x2 inputs")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise KlmOpsError("gemm_fp4 shapes do not match")
    M, Kb = a.shape
    N = b.shape[0]
    if variant is None:
        variant = pick_mx_variant(M, N, Kb, _pitch_bytes(a, b), 6)
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    lib = load()
    if not hasattr(lib, "klm_gemm_fp4"):
        raise KlmOpsError("libklm_ops.so was built without the fp4 kernel")
    pitch = _pitch_bytes(a, b)
    if variant in (6, 7):
        # 6: the persistent stream with its plan (as bf16 variant 10); 7: the partial round split wherever the shape allows
        return _stream(2, a, b, out, M, N, Kb, pitch, mode=1 if variant == 6 else 2, rot=None if variant == 6 else 0, what="fp4")
    with _on_device(a):
        if pitch:
            if variant not in (2, 3, 4):
                raise KlmOpsError(f"gemm_fp4: pitched operands need the hand-scheduled stream (variant 2 / 3 / 4), K={Kb * 2} gives {variant}")
            rc = lib.klm_gemm_mx_pitched(2, a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, Kb, pitch, {2: 0, 3: 1, 4: 2}[variant], _stream_of(a))
        else:
            rc = lib.klm_gemm_fp4(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, Kb * 2, variant, _stream_of(a))
    _check(rc, f"klm_gemm_fp4(M={M},N={N},K={Kb * 2},variant={variant},pitch={pitch})")
    return out


def _nbytes(t) -> int:
    return t.numel() * t.element_size()


def hbm_fill(t, pattern: int = 0xA5A5A5A5):
    """Write `pattern` (32-bit) over the whole tensor with streaming 16-byte stores."""
    _require_cuda(t)
    with _on_device(t):
        _check(load().klm_hbm_fill(t.data_ptr(), _nbytes(t), pattern & 0xFFFFFFFF, _stream_of(t)), "klm_hbm_fill")
    return t


def hbm_copy(dst, src):
    _require_cuda(dst, src)
    if _nbytes(dst) != _nbytes(src):
        raise KlmOpsError("hbm_copy: size mismatch")
    with _on_device(src):
        _check(load().klm_hbm_copy(dst.data_ptr(), src.data_ptr(), _nbytes(src), _stream_of(src)), "klm_hbm_copy")
    return dst


def hbm_copy_ex(dst, src, flags: int = 3, grid: int = 0):
    """Tuning variant of hbm_copy (see klm_ops.h)."""
    _require_cuda(dst, src)
    if _nbytes(dst) != _nbytes(src):
        raise KlmOpsError("hbm_copy_ex: size mismatch")
    with _on_device(src):
        _check(load().klm_hbm_copy_ex(dst.data_ptr(), src.data_ptr(), _nbytes(src), flags, grid, _stream_of(src)), "klm_hbm_copy_ex")
    return dst


def hbm_fill_ex(t, pattern: int = 0xA5A5A5A5, flags: int = 2, grid: int = 0):
    _require_cuda(t)
    with _on_device(t):
        _check(load().klm_hbm_fill_ex(t.data_ptr(), _nbytes(t), pattern & 0xFFFFFFFF, flags, grid, _stream_of(t)), "klm_hbm_fill_ex")
    return t


def hbm_read_ex(src, sink, flags: int = 1, grid: int = 0):
    """Tuning variant of hbm_read (see klm_ops.h)."""
    _require_cuda(src, sink)
    with _on_device(src):
        _check(load().klm_hbm_read_ex(src.data_ptr(), _nbytes(src), sink.data_ptr(), flags, grid, _stream_of(src)), "klm_hbm_read_ex")
    return sink


def hbm_read(src, sink):
    _require_cuda(src, sink)
    with _on_device(src):
        _check(load().klm_hbm_read(src.data_ptr(), _nbytes(src), sink.data_ptr(), _stream_of(src)), "klm_hbm_read")


def hbm_checksum(src) -> int:
    """Sum of the tensor's 32-bit words (mod 2^64); synchronises to return the value."""
    import torch
    _require_cuda(src)
    acc = torch.zeros(1, dtype=torch.int64, device=src.device)
    with _on_device(src):
        _check(load().klm_hbm_checksum(src.data_ptr(), _nbytes(src), acc.data_ptr(), _stream_of(src)),
               "klm_hbm_checksum")
    return int(acc.item()) & 0xFFFFFFFFFFFFFFFF


def spin(cycles: int, blocks: int = 256, device=None):
    import torch
    s = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):            # None keeps the current device
        _check(load().klm_spin(int(cycles), int(blocks), s), "klm_spin")
