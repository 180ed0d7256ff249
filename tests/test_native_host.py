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
    # PCIe health, per-XCD accumulators and the process table: what the Python reader emits
    assert g1["pcie_replay_acc"] == 8 and g1["pcie_l0_recovery_acc"] == 2 and g1["pcie_nak_sent_acc"] == 0 and g1["pcie_nak_rcvd_acc"] is None
    assert g1["pcie_link_speed"] == 320 and g1["pcie_link_width"] == 16 and g1["pcie_bandwidth_acc"] == 123456789 and g1["pcie_replay_rollover_acc"] == 0
    assert g1["xcd_busy_acc"] == [1000000 * (x + 1) + 1 for x in range(8)] and g1["xcd_busy_inst"] == [10 * x for x in range(8)]
    assert g1["processes"] == 2 and g1["process_list"][0] == {"pid": 4242, "name": 'python3 "vllm"', "vramBytes": 17179869184}
    assert "processes" not in line["gpus"][0] and g1["peak"]["temp_edge"] is None
    from k8s_llm_monitor_amd.collectors.gpu import metrics_to_raw
    py_keys = set(metrics_to_raw({k: 1 for k in (
        "average_gfx_activity", "average_umc_activity", "gfx_activity_acc", "mem_activity_acc", "accumulation_counter", "firmware_timestamp",
        "current_socket_power", "energy_accumulator", "temperature_hotspot", "temperature_mem", "temperature_edge", "current_gfxclk", "current_uclk",
        "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc",
        "xgmi_read_data_acc", "xgmi_write_data_acc", "xgmi_link_status", "xgmi_link_width", "xgmi_link_speed", "pcie_bandwidth_inst",
        "pcie_bandwidth_acc", "pcie_link_speed", "pcie_link_width", "pcie_replay_count_acc", "pcie_replay_rover_count_acc",
        "pcie_l0_to_recov_count_acc", "pcie_nak_sent_count_acc", "pcie_nak_rcvd_count_acc")} | {"xcp_stats.gfx_busy_acc": [[1] * 8], "xcp_stats.gfx_busy_inst": [[1] * 8]}))
    assert py_keys <= set(g1), sorted(py_keys - set(g1))             # every metrics key of the Python reader is in the sampler's line
    # the same line through the production reader and normaliser
    src = GpuSource(NativeSamplerReader(argv=[str(san.out), "--selftest"]))
    try:
        s = src.sample()
        assert [x["index"] for x in s] == [0, 1] and s[0]["memoryTotal"] == 294896 and s[1]["memoryUsed"] == 1284
        assert s[0]["temperature"] == 46 and s[0]["utilization"] == 20 and s[0]["powerCapW"] == 1400
        assert len(s[0]["xgmi"]["links"]) == 7 and s[0]["peak"]["gfx_activity"] == 30
        assert s[1]["pcie"]["errors"]["replay"]["count"] == 8 and s[1]["pcie"]["linkSpeedGTs"] == 32.0 and s[1]["xcdActivity"] == [10 * x for x in range(8)]
        assert s[1]["processes"] == 2 and s[1]["processList"][1]["name"] == "klm_burner"
    finally:
        src.close()



@pytest.fixture(scope="module")
def asan_bins(built):
    """Host-side AddressSanitizer + UBSan builds of the two HIP executables (device code unsanitized: the GPU
    sanitizers are not available on the MI355X pool). Cached under _native/bin-asan, rebuilt when a source changed."""
    out = {}
    for t in build.targets():
        if t.kind == "hip-exe":
            san = build.sanitized(t)
            build.build_target(san, verbose=False)
            out[t.name] = san.out
    return out


def _clean(r):
    return all(s not in r.stderr for s in ("ERROR: AddressSanitizer", "runtime error", "LeakSanitizer"))


def test_probe_and_burner_host_code_under_asan_ubsan(asan_bins):
    probe, burner = str(asan_bins["klm_rccl_probe"]), str(asan_bins["klm_burner"])
    for cmd, rc in (([probe, "--help"], 0), ([probe, "--bogus"], 1), ([probe, "--gpus"], 1), ([probe, "--gpus", "0"], 1),
                    ([probe, "--dtype", 'x"y'], 1), ([probe, "--min", "1", "--dtype", "fp32"], 1), ([probe, "--factor", "1"], 1),
                    ([probe, "--gpus", "99999999999999999999"], None),
                    ([burner, "--help"], 0), ([burner, "--bogus"], 1), ([burner, "--kind"], 1), ([burner, "--kind", 'a"b\\'], 1),
                    ([burner, "--size", "1000"], 1), ([burner, "--duty", "101"], 1), ([burner, "--hbm-gib", "-1"], 1),
                    ([burner, "--device", "-3"], 1)):
        r = run(cmd)
        assert _clean(r), (cmd, r.stderr[-1500:])
        if rc is not None:
            assert r.returncode == rc, (cmd, r.returncode, r.stdout, r.stderr[-300:])
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                json.loads(line)                                  # errors stay valid JSON whatever the argument held
    # without a GPU both stop at the first HIP call with a JSON error, cleanly
    import torch
    if not torch.cuda.is_available():
        for cmd in ([probe, "--gpus", "1"], [burner, "--seconds", "1"]):
            r = run(cmd)
            assert r.returncode == 2 and _clean(r) and "error" in json.loads(r.stdout.strip().splitlines()[-1])


def test_probe_selftest_output_feeds_the_curve_builder(asan_bins):
    from k8s_llm_monitor_amd.parallel import probe
    r = run([str(asan_bins["klm_rccl_probe"]), "--selftest", "--max", str(1 << 28), "--factor", "4"])
    assert r.returncode == 0 and _clean(r), r.stderr[-1500:]
    parsed = probe.parse_probe_output(r.stdout)
    h = parsed["header"]
    assert h["in_place"] is True and h["n_gpus"] == 8 and h["device"] == 'AMD "Instinct" MI355X\\' and "max over ranks" in h["timing"]
    sizes = [p["bytes"] for p in parsed["points"]]
    assert sizes == probe.sweep_sizes(1024, 1 << 28, 4) and not parsed["errors"]
    for p in parsed["points"]:
        assert p["time_us"] == p["time_us_max_rank"] >= p["time_us_rank0"] > 0
        assert p["elements_checked"] == p["bytes"] // 2 * 8 and p["elements_mismatched"] == 0 and p["check"] == "ok"
        assert p["busbw_GBps"] == pytest.approx(p["algbw_GBps"] * 2 * 7 / 8, rel=1e-3)
    c = probe.curve(8, parsed["points"], "bf16", verified=True, executor="native")
    assert c["peak_bytes"] == 1 << 28 and 300 < c["peak_busbw_GBps"] < 400
    assert h["mode"] == "allreduce" and h["bound_GBps"] == pytest.approx(7 * 153.0 * 1.1)
    # the physical bound: a 0.436 us "copy" of 256 MiB and a point above the mesh ceiling are implausible
    r = run([str(asan_bins["klm_rccl_probe"]), "--selftest-bounds", "--max", "4096"])
    assert r.returncode == 0 and _clean(r), r.stderr[-1500:]
    extra = probe.parse_probe_output(r.stdout)["points"][-4:]
    assert [(p["n_gpus"], p["check"]) for p in extra] == [(1, "implausible"), (1, "ok"), (8, "implausible"), (8, "ok")]
    assert extra[0]["algbw_GBps"] > 600000 and [probe.judged(dict(p, check="ok"))["check"] for p in extra] == [p["check"] for p in extra]
    c = probe.curve(1, extra[:2], "bf16", verified=True, executor="native")
    assert c["verified"] is False and c["mode"] == "local_copy" and c["peak_algbw_GBps"] == pytest.approx(7900.0, rel=1e-3)
    # an odd max and a factor whose product would overflow size_t end the list at max without looping
    r = run([str(asan_bins["klm_rccl_probe"]), "--selftest", "--min", "3", "--max", str((1 << 63) + 5), "--factor", "1000000"])
    assert r.returncode == 0 and _clean(r), r.stderr[-1500:]
    pts = probe.parse_probe_output(r.stdout)["points"]
    assert [p["bytes"] for p in pts] == [2, 3000000 - 3000000 % 2, 3000000000000, 3000000000000000000, (1 << 63) + 4]


def test_burner_selftest_planner_matches_the_python_planner(asan_bins):
    from k8s_llm_monitor_amd.loadgen.burner import plan_window
    r = run([str(asan_bins["klm_burner"]), "--selftest"])
    assert r.returncode == 0 and _clean(r), r.stderr[-1500:]
    lines = [json.loads(x) for x in r.stdout.splitlines()]
    rows = lines[0]["rows"]
    assert len(rows) == 28
    for duty, kernel_s, n in rows:
        assert n == plan_window(duty, lines[0]["period_s"], kernel_s), (duty, kernel_s, n)
    # which dense row pitches get padded: the same rule in both burners
    from k8s_llm_monitor_amd.ops import camping_pitch, pick_tile_kernel
    # which kernel a square problem of each size runs on: the library's dispatch (klm_gemm_pick) = the Python cost model
    assert lines[1]["selftest"] == "gemm_pick" and lines[1]["cus"] == 256 and len(lines[1]["rows"]) == 9
    code = {None: 0, "stream": 1, "p128": 2, "p256": 3}
    for n, bf16, fp8, fp4 in lines[1]["rows"]:
        assert (bf16, fp8, fp4) == tuple(code[pick_tile_kernel(n, n, kb)] for kb in (2 * n, n, n // 2)), n
    lines = [lines[0]] + lines[2:]
    assert lines[1]["selftest"] == "camping_pitch" and len(lines[1]["rows"]) == 12
    for nbytes, bad in lines[1]["rows"]:
        assert bool(bad) == camping_pitch(nbytes), nbytes
    assert camping_pitch(24576) and camping_pitch(20480) and not camping_pitch(16384) and not camping_pitch(24064)
    assert lines[2]["tflops"] == pytest.approx(2 * 8192 ** 3 * 2700 / 1.97 / 1e12, rel=1e-4) and lines[3]["TBps"] == pytest.approx(3.006, abs=1e-3)
    assert lines[4]["kind"] == 'gemm"bf16\\\n' and lines[4]["hbm_hold_gib"] == 3.0 and lines[4]["duty_achieved"] == 98.5
