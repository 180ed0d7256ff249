"""Host-side checks of the native code on a machine without a GPU: every executable is built
for gfx950, fails with a JSON error (not a crash) when there is no device, and the sampler's
JSON writer runs clean under AddressSanitizer + UBSan and feeds the Python reader."""
import json
import os
import shutil
import subprocess

import pytest

from k8s_llm_monitor_amd import build
from k8s_llm_monitor_amd.collectors.gpu import GpuSource, NativeSamplerReader

HAVE_HIPCC = build.hipcc() is not None


def run(cmd, **kw):
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=60, env=env, **kw)


@pytest.fixture(scope="module")
def built():
    if not HAVE_HIPCC:
        pytest.skip("hipcc not installed")
    build.build_all(verbose=False)
    return {t.name: t.out for t in build.targets()}


def test_all_targets_exist_and_carry_gfx950_code(built):
    assert set(built) == {"klm_ops", "klm_burner", "klm_rccl_probe", "klm_xgmi", "klm_amdsmi_sampler"}
    for name, path in built.items():
        assert path.exists(), name
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if os.path.exists(objdump):
        import tempfile
        import shutil
        with tempfile.TemporaryDirectory() as td:      # --offloading writes the code objects next to its input: work on a copy
            for lib in ("klm_ops", "klm_xgmi"):
                copy = shutil.copy(built[lib], td)
                r = subprocess.run([objdump, "--offloading", copy], capture_output=True, text=True, cwd=td)
                assert "gfx950" in r.stdout + r.stderr, lib
        assert not [f for f in os.listdir(built["klm_ops"].parent) if "hipv4" in f or ".host-" in f]


def test_executables_fail_with_json_when_no_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for name, args in (("klm_amdsmi_sampler", ["--once"]), ("klm_rccl_probe", ["--gpus", "1", "--max", "4096"]),
                       ("klm_burner", ["--seconds", "1"])):
        r = run([str(built[name]), *args])
        assert r.returncode not in (0, -11, -6, 139, 134), (name, r.returncode, r.stderr[-300:])
        msg = json.loads(r.stdout.strip().splitlines()[-1])
        assert "error" in msg, (name, msg)


def test_bad_arguments_are_rejected(built):
    assert run([str(built["klm_rccl_probe"]), "--bogus"]).returncode == 1
    assert run([str(built["klm_burner"]), "--size", "1000"]).returncode == 1
    assert run([str(built["klm_amdsmi_sampler"]), "--period", "0"]).returncode == 1


def test_sampler_json_writer_under_asan_ubsan_feeds_the_python_reader(built, tmp_path):
    if not shutil.which("g++"):
        pytest.skip("g++ not installed")
    t = next(t for t in build.targets() if t.name == "klm_amdsmi_sampler")
    san = build.sanitized(t, tmp_path)
    build.build_target(san, force=True, verbose=False)
    r = run([str(san.out), "--selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(line["gpus"]) == 2 and line["poll_us_mean"] == 300
    g1 = line["gpus"][1]
    assert g1["name"] == 'AMD Instinct "MI355X"'                   # escaping
    assert g1["temp_edge"] is None                                   # unsupported sentinel -> null
    assert g1["xgmi_link_status"][1] is None and g1["xgmi_link_status"][0] == 1
    assert g1["peak"]["power_w"] == 900 and g1["power_w"] == 600     # mean and peak of the period
    # the same line through the production reader and normaliser
    src = GpuSource(NativeSamplerReader(argv=[str(san.out), "--selftest"]))
    try:
        s = src.sample()
        assert [x["index"] for x in s] == [0, 1] and s[0]["memoryTotal"] == 294896 and s[1]["memoryUsed"] == 1284
        assert s[0]["temperature"] == 46 and s[0]["utilization"] == 20 and s[0]["powerCapW"] == 1400
        assert len(s[0]["xgmi"]["links"]) == 7 and s[0]["peak"]["gfx_activity"] == 30
    finally:
        src.close()

