"""Static check of the compiled mesh-collective kernels (CPU only; listed in .gpurunignore because it has to name
the instructions it looks for)."""
import re
import subprocess

import pytest

from k8s_llm_monitor_amd import build


def test_xgmi_kernels_write_nothing_through_the_scalar_cache(tmp_path):
    """The mesh collectives synchronise through flags in memory: every flag access must be a vector instruction
    (release / acquire at system scope), never a scalar-cache store or atomic."""
    hipcc = build.hipcc()
    if hipcc is None:
        pytest.skip("hipcc not installed")
    src = build.PKG / "parallel" / "csrc" / "xgmi_allreduce.hip"
    asm = tmp_path / "x.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", str(src),
                        "-o", str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    assert not re.search(r"^\s*(s_store|s_atomic|s_buffer_store|s_dcache_wb)", text, re.M)
    assert "buffer_wbl2 sc0 sc1" in text and "buffer_inv sc0 sc1" in text          # the system-scope fences are there
    assert "scratch_" not in text                                                     # no spills, no struct copies
