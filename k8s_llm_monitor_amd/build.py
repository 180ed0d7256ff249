"""In-tree native build for gfx950 (MI355X).

Compiles every HIP/C++ source of the package with explicit ``hipcc`` / ``g++`` command
lines (no JIT cache, no pip install): the built ``.so`` and executables stay inside the
package directory so that they travel with a ``gpurun`` snapshot and show up as in-tree
native code in the loader's ``/proc/self/maps``.

The reference has no build step at all (it is two interpreted files);
this module exists because the MI355X validation toolkit is native.

Usage::

    python -m k8s_llm_monitor_amd.build            # build what is stale
    python -m k8s_llm_monitor_amd.build --force    # rebuild everything
    python -m k8s_llm_monitor_amd.build --list     # show targets
"""
from __future__ import annotations

import argparse
import hashlib
import os
import shutil
import subprocess
import sys
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Sequence

PKG = Path(__file__).resolve().parent
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
ARCH = "gfx950"
BIN_DIR = PKG / "_native" / "bin"
LIB_DIR = PKG / "_native" / "lib"


@dataclass
class Target:
    name: str
    out: Path
    sources: List[Path]
    kind: str  # "hip-shared" | "hip-exe" | "cxx-exe"
    extra: List[str] = field(default_factory=list)
    deps: List[Path] = field(default_factory=list)  # headers etc. for staleness

    def source_hash(self) -> str:
        """sha256 over what the artefact is made from: kind, extra flags, and the CONTENT of every source and header.
        Passed to the compiler as KLM_SRC_HASH and embedded in the binary (marker ``KLM_SRC_HASH=<hash>``)."""
        h = hashlib.sha256()
        h.update(("%s|%s|%s" % (self.kind, ARCH, " ".join(self.extra))).encode())
        for p in sorted([*self.sources, *self.deps], key=lambda q: q.name):
            h.update(b"\0" + p.name.encode() + b"\0")
            if p.exists():
                h.update(p.read_bytes())
        return h.hexdigest()[:16]

    def stale(self) -> bool:
        """Stale = the binary does not carry the hash of the tree's sources. Content, not mtimes: a snapshot copy or a
        `git checkout` rewrites every mtime, and a `.so` older or newer than its sources proves nothing either way."""
        return not self.out.exists() or embedded_hash(self.out) != self.source_hash()


HASH_MARKER = b"KLM_SRC_HASH="


def embedded_hash(path: Path) -> Optional[str]:
    """The hash a built artefact carries (None: built by other means, or before hashes existed)."""
    try:
        data = path.read_bytes()
    except OSError:
        return None
    i = data.find(HASH_MARKER)
    while i >= 0:
        tail = data[i + len(HASH_MARKER): i + len(HASH_MARKER) + 16]
        if len(tail) == 16 and all(c in b"0123456789abcdef" for c in tail):
            return tail.decode()
        i = data.find(HASH_MARKER, i + 1)
    return None


def hipcc() -> Optional[str]:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), str(ROCM / "bin" / "hipcc")):
        if c and Path(c).exists():
            return c
    return None


def targets() -> List[Target]:
    ops = PKG / "ops" / "csrc"
    hdr = ops / "klm_ops.h"
    ops_src = [ops / "gemm_bf16.hip", ops / "gemm_mx.hip", ops / "hbm.hip", ops / "llm_elementwise.hip", ops / "llm_attention.hip",
               ops / "llm_skinny.hip", ops / "llm_mid.hip"]
    ops_src = [s for s in ops_src if s.exists()]
    out: List[Target] = [
        Target("klm_ops", LIB_DIR / "libklm_ops.so", ops_src, "hip-shared", deps=[hdr, ops / "llm_common.h", ops / "gemm_ring256.h", ops / "gemm_p128.h", ops / "gemm_quad128.h", ops / "klm_launch.h", ops / "gemm_quad_asm.inc", ops / "gemm_quad_asm.h"]),
    ]
    burner = PKG / "loadgen" / "native" / "klm_burner.cpp"
    if burner.exists():
        # the standalone burner carries the load-generator kernels only (GEMMs, HBM, spin), not the serving engine's
        out.append(Target("klm_burner", BIN_DIR / "klm_burner", [burner, *[x for x in ops_src if not x.name.startswith("llm_")]], "hip-exe",
                          extra=[f"-I{ops}"], deps=[hdr, ops / "llm_common.h", ops / "gemm_ring256.h", ops / "gemm_p128.h", ops / "gemm_quad128.h", ops / "klm_launch.h", ops / "gemm_quad_asm.inc", ops / "gemm_quad_asm.h"]))
    probe = PKG / "parallel" / "csrc" / "rccl_probe.cpp"
    if probe.exists():
        out.append(Target("klm_rccl_probe", BIN_DIR / "klm_rccl_probe", [probe], "hip-exe",
                          extra=[f"-I{ROCM}/include/rccl", f"-L{ROCM}/lib", "-lrccl",
                                 f"-Wl,-rpath,{ROCM}/lib", "-lpthread"]))
    xgmi = PKG / "parallel" / "csrc" / "xgmi_allreduce.hip"
    if xgmi.exists():
        out.append(Target("klm_xgmi", LIB_DIR / "libklm_xgmi.so", [xgmi], "hip-shared"))
    smi = PKG / "collectors" / "native" / "amdsmi_sampler.cpp"
    if smi.exists():
        out.append(Target("klm_amdsmi_sampler", BIN_DIR / "klm_amdsmi_sampler", [smi], "cxx-exe",
                          extra=[f"-I{ROCM}/include", f"-L{ROCM}/lib", "-lamd_smi",
                                 f"-Wl,-rpath,{ROCM}/lib", "-lpthread"]))
    return out


def _cmd(t: Target) -> List[str]:
    srcs = [str(s) for s in t.sources]
    if t.kind in ("hip-shared", "hip-exe"):
        cc = hipcc()
        if cc is None:
            raise RuntimeError("hipcc not found (set HIPCC or ROCM_PATH)")
        base = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-Wall", "-Wno-unused-function",
                f'-DKLM_SRC_HASH="{t.source_hash()}"', "-x", "hip"]
        if t.kind == "hip-shared":
            base += ["-shared", "-fPIC"]
        return [*base, *srcs, "-o", str(t.out), *t.extra]
    if t.kind == "cxx-exe":
        cxx = shutil.which("g++") or shutil.which("c++") or "g++"
        return [cxx, "-O2", "-std=c++17", "-Wall", f'-DKLM_SRC_HASH="{t.source_hash()}"', *srcs, "-o", str(t.out), *t.extra]
    raise ValueError(t.kind)


SANITIZE_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def sanitized(t: Target, out_dir: Optional[Path] = None) -> Target:
    """The debug preset of SURVEY.md section 5 ("build native code with -fsanitize=address,
    undefined"): host code only. GPU sanitizers are not available on the MI355X pool, so HIP
    targets get the flags on the host side alone (``-Xarch_host``)."""
    d = out_dir or (PKG / "_native" / "bin-asan")
    extra = list(t.extra)
    if t.kind == "cxx-exe":
        extra = SANITIZE_FLAGS + extra
    else:
        for f in SANITIZE_FLAGS[3:]:
            extra = ["-Xarch_host", f] + extra
        extra = ["-g"] + extra
    return Target(t.name + "-asan", d / t.out.name, t.sources, t.kind, extra=extra, deps=t.deps)


def build_target(t: Target, force: bool = False, verbose: bool = True) -> bool:
    """Build one target. Returns True when a compile ran."""
    if not force and not t.stale():
        return False
    t.out.parent.mkdir(parents=True, exist_ok=True)
    cmd = _cmd(t)
    if verbose:
        print("[build]", t.name, "->", t.out.relative_to(PKG.parent), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"build of {t.name} failed:\n{' '.join(cmd)}\n{r.stdout}")
    if verbose and r.stdout.strip():
        print(r.stdout)
    return True


def build_all(force: bool = False, only: Optional[Sequence[str]] = None, verbose: bool = True) -> List[str]:
    built = []
    for t in targets():
        if only and t.name not in only:
            continue
        if build_target(t, force=force, verbose=verbose):
            built.append(t.name)
    return built


def artifact(name: str) -> Path:
    for t in targets():
        if t.name == name:
            return t.out
    raise KeyError(name)


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--sanitize", action="store_true", help="also build the host-sanitizer preset (bin-asan/) of the C++ and HIP executables")
    ap.add_argument("targets", nargs="*")
    a = ap.parse_args(argv)
    if a.list:
        for t in targets():
            print(f"{t.name:22s} {t.kind:10s} {'stale' if t.stale() else 'ok':5s} src:{t.source_hash()} built:{embedded_hash(t.out)} {t.out}")
        return 0
    built = build_all(force=a.force, only=a.targets or None)
    if a.sanitize:
        for t in targets():
            # every executable: plain C++ ones whole, HIP ones on the host side (argument parsing, JSON printing, the launch loop)
            if t.kind in ("cxx-exe", "hip-exe") and (not a.targets or t.name in a.targets):
                if build_target(sanitized(t), force=a.force):
                    built.append(t.name + "-asan")
    print("built:", ", ".join(built) if built else "(nothing stale)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
