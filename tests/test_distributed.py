"""Multi-process CPU rehearsal of the distributed paths (gloo, world_size 2): the RCCL probe's
sweep/timing/reporting logic and the bench's step loop. On MI355X the same code runs with
backend "nccl" (= RCCL); 2/4/8-GPU numbers themselves can only come from an 8-GPU node
("RCCL probe")."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun(nproc, args, timeout=180):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), *args]
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(240)
def test_probe_and_bench_loop_on_two_ranks():
    r = torchrun(2, [os.path.join(ROOT, "tests", "dist_worker.py")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert out["curve"]["n_gpus"] == 2 and out["curve"]["verified"] and out["curve"]["executor"] == "torch.distributed/gloo"
    assert out["checks"] == ["ok", "ok", "ok"] and out["bytes"] == [1024, 4096, 65536]
    assert out["curve"]["peak_busbw_GBps"] > 0 and out["allgather"] and out["reducescatter"] and out["alltoall"]
    assert out["xgmi"]["world"] == 2 and out["xgmi"]["expected_bytes_per_gpu"] == 3 * (1 << 16) and out["xgmi"]["error"]   # no amdsmi here
    assert out["allreduce_ok"] and out["same_time_on_all_ranks"] and out["i"] == 4 and out["steps_s"] > 0


@pytest.mark.timeout(240)
def test_dist_probe_cli_writes_dashboard_results(tmp_path):
    out = tmp_path / "probe.json"
    r = torchrun(2, ["-m", "k8s_llm_monitor_amd.parallel.dist_probe", "--cpu", "--dtype", "fp32", "--max-bytes", "16384",
                     "--iters", "2", "--warmup", "1", "--out", str(out)])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    doc = json.loads(out.read_text())
    assert doc["curves"][0]["n_gpus"] == 2 and doc["scaling"][0]["ceiling_mesh_GBps"] == 153.0
    assert [p["bytes"] for p in doc["curves"][0]["points"]] == [1024, 4096, 16384]


REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


def check_bench_line(stdout: str, n: int, steps: int, warmup: int):
    # exactly ONE JSON line, from rank 0 (gloo's C++ layer prints its own "[Gloo] Rank ..." chatter on CPU), and it is the
    # LAST line of stdout and short enough for the driver to keep whole (it keeps 8 000 characters)
    lines = [l[l.index('{"metric"'):] for l in stdout.splitlines() if '{"metric"' in l]
    assert len(lines) == 1, stdout
    assert '{"metric"' in stdout.strip().splitlines()[-1]
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert isinstance(d[k], t), (k, d[k])
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md has no number to divide by
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "bf16" and "synthetic" in d["data"]
    assert set(d["config"]) >= {"model", "global_batch", "seq_len", "parallelism"} and d["config"]["parallelism"] == (f"dp{n}" if n > 1 else "single")
    # value is the whole-job aggregate: n ranks x FLOP per step / step time
    assert d["value"] == pytest.approx(n * d["flop_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12, rel=2e-2, abs=0.006)   # printed with 2 decimals
    assert "validation_detail" not in d and "rccl_curves" not in d and "llm_serving" not in d      # those live in the detail file
    with open(d["detail"], "r", encoding="utf-8") as f:
        full = json.load(f)
    assert full["value"] == d["value"] and full["metric"] == d["metric"] and "wall_seconds" in full
    return d, full


@pytest.mark.timeout(240)
def test_bench_contract_single_process_rehearsal(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-rehearsal", "--size", "256", "--steps", "3", "--warmup", "1",
                        "--detail-out", str(tmp_path / "d.json")], cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    d, _ = check_bench_line(r.stdout, 1, 3, 1)
    assert d["time_budget_s"] == 540.0 and d["shed"] == [] and d["warmup_steps_run"] == 1     # no wall-time warm-up on the CPU stand-in


@pytest.mark.timeout(240)
def test_bench_contract_two_ranks_rehearsal(tmp_path):
    r = torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "256", "--steps", "4", "--warmup", "2",
                     "--detail-out", str(tmp_path / "d.json")])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d, _ = check_bench_line(r.stdout, 2, 4, 2)
    assert d["allreduce_bytes_per_step"] == 256 * 256 * 2 and d["allreduce_busbw_GBps_overlapped"] > 0


@pytest.mark.timeout(240)
def test_bench_launches_its_own_ranks(tmp_path):
    """Plain ``python bench.py --gpus 2`` (no torchrun): the parent starts the ranks as a child, relays ONE compact JSON line,
    and the line proves two ranks ran and carries the peak at n = 1 AND n = 2; the curves
    themselves are in the detail file."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "256",
                        "--steps", "3", "--warmup", "1", "--detail-out", str(tmp_path / "d.json")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=220)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d, full = check_bench_line(r.stdout, 2, 3, 1)
    assert d["launcher"].startswith("bench.py self-launch")
    assert d["rccl_ranks_seen"] == {"process_group_world_size": 2, "communicators": [1, 2]}
    assert sorted(d["peak_busbw_GBps"]) == ["1", "2"] and d["peak_busbw_GBps"]["1"] > 0 and d["peak_busbw_GBps"]["2"] > 0
    assert sorted(full["rccl_curves"]) == ["1", "2"]
    c1, c2 = full["rccl_curves"]["1"], full["rccl_curves"]["2"]
    assert c1["verified"] and c2["verified"] and c1["ranks_seen"] == 1 and c2["ranks_seen"] == 2
    assert c1["curve_key"] == "algbw_GBps" and c1["mode"] == "local_copy" and c2["curve_key"] == "busbw_GBps" and c2["ceiling_mesh_GBps"] == 153.0
    assert [b for b, _ in c2["curve"]] == [1024, 4096, 16384, 65536] and c2["curve"][-1][1] > 0
    assert "peak_busbw_GBps_native" not in d and "rccl_curves_native" not in full       # the C++ probe needs GPUs


@pytest.mark.timeout(240)
def test_bench_four_ranks_give_the_1_2_4_curve(tmp_path):
    r = torchrun(4, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--cpu-rehearsal", "--size", "128", "--steps", "2", "--warmup", "1",
                     "--probe-max-bytes", "4096", "--detail-out", str(tmp_path / "d.json")])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d, full = check_bench_line(r.stdout, 4, 2, 1)
    assert d["launcher"] == "torch.distributed.run"
    assert sorted(d["peak_busbw_GBps"]) == ["1", "2", "4"] and d["rccl_ranks_seen"]["communicators"] == [1, 2, 4]
    assert full["rccl_curves"]["4"]["ceiling_mesh_GBps"] == 3 * 153.0 and full["rccl_allreduce"]["n_gpus"] == 4


@pytest.mark.timeout(240)
def test_four_ranks_wait_for_rank0_validation_and_the_budget_sheds_phases(tmp_path):
    """rank 0 runs (stand-in, slow) validation phases while ranks 1-3 wait on the gloo control
    group and then finish cleanly; phases that do not fit --time-budget are shed, longest first, and listed in the line."""
    import time
    t0 = time.time()
    r = torchrun(4, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--cpu-rehearsal", "--size", "128", "--steps", "2", "--warmup", "1",
                     "--no-probe", "--time-budget", "60", "--rehearsal-phases", "quick:5:1.0,long:400:0.1,mid:20:1.5,huge:900:0.1",
                     "--detail-out", str(tmp_path / "d.json")])
    took = time.time() - t0
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]           # every rank exited 0: nobody was left at a barrier
    d, full = check_bench_line(r.stdout, 4, 2, 1)
    assert d["shed"] == ["huge", "long"] and d["time_budget_s"] == 60.0
    assert full["validation_detail"]["ran"] == ["quick", "mid"] and d["validation_pass"] is True and d["validation_checks"] == 2
    assert full["wall_seconds"]["validation_phases"] >= 2.4 and took < 120
    # a budget already spent sheds everything that is optional and still prints the headline
    r = torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "128", "--steps", "2", "--warmup", "1",
                     "--time-budget", "1", "--rehearsal-phases", "quick:5:1.0", "--detail-out", str(tmp_path / "e.json")])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d, _ = check_bench_line(r.stdout, 2, 2, 1)
    assert "probe" in d["shed"] and "quick" in d["shed"] and "peak_busbw_GBps" not in d


@pytest.mark.timeout(240)
def test_a_section_that_never_returns_still_leaves_the_result_line(tmp_path):
    """The budget sheds sections before they start; a section that was started and hangs (on hardware: a collective one rank
    never joined) is ended by the watchdog. Rank 0 prints the compact line from the timed number it already holds, every
    rank exits 0, and the line says which section was cut."""
    import time
    t0 = time.time()
    r = torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "128", "--steps", "2", "--warmup", "1",
                     "--no-probe", "--time-budget", "16", "--watchdog-grace", "1", "--rehearsal-phases", "hang:0.5:600",
                     "--detail-out", str(tmp_path / "d.json")])
    took = time.time() - t0
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d, full = check_bench_line(r.stdout, 2, 2, 1)
    assert d["watchdog"]["section"] == "validation_phases" and 16.5 < d["watchdog"]["fired_after_s"] < 30 and took < 60
    assert full["watchdog"] == d["watchdog"] and "did not return" in r.stderr


def test_watchdog_unit():
    import io
    import types
    from k8s_llm_monitor_amd.validation import benchlib
    import time
    codes = []
    # no timed result yet: nothing to print, exit status 3
    a = types.SimpleNamespace(provisional=None, section="start", detail_out="")
    out = io.StringIO()
    benchlib.Watchdog(a, 0, 0.05, time.monotonic(), out=out, exit_fn=codes.append).start()._th.join(5)
    assert codes == [3] and out.getvalue() == ""
    # cancelled in time: nothing happens
    w = benchlib.Watchdog(a, 0, 0.2, time.monotonic(), out=out, exit_fn=codes.append).start()
    w.cancel()
    w._th.join(5)
    assert codes == [3]
    # a rank other than 0 prints nothing
    a.provisional = ({"metric": "m", "value": 1.0}, {"shed": []})
    benchlib.Watchdog(a, 1, 0.0, time.monotonic() - 5, out=out, exit_fn=codes.append).start()._th.join(5)
    assert codes == [3, 0] and out.getvalue() == ""


def test_compact_line_stays_under_the_limit_whatever_the_run_produced():
    """The line is built from a fixed set of scalars; if a future field ever bloats it, optional parts are dropped, never the
    contract's keys (and the 8-rank shape of the result, with four curves from both executors, fits as it is)."""
    from k8s_llm_monitor_amd.parallel import probe
    from k8s_llm_monitor_amd.validation import benchlib
    curves = {str(n): probe.brief(probe.curve(n, [probe.point(n, b, b / 100e9 + 2e-5) for b in probe.sweep_sizes(1024, 1 << 28)])) for n in (1, 2, 4, 8)}
    summ = {k: 0.123456789 for k in benchlib._SUMMARY_KEYS}
    summ.update(power_alert=True, throttle_alert=True, xgmi_unit_verdict_tx="KiB (x1024)", **{"pass": {f"check{i}": True for i in range(14)}})
    res = {"metric": "m" * 150, "value": 12345.67, "unit": "TFLOP/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 2.1234, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "config": {"model": "x" * 160, "global_batch": 8, "seq_len": 8192, "parallelism": "dp8"},
           "flop_per_step": 3.3e12, "gemm_variant": 4, "rccl_curves": curves, "rccl_curves_native": curves, "validation": summ,
           "validation_detail": {"rows": list(range(5000))}, "llm_serving": {"x": "y" * 9000}, "wall_seconds": {"total": 431.2},
           "rccl_ranks_seen": {"process_group_world_size": 8, "communicators": [1, 2, 4, 8]}, "shed": [], "launcher": "torch.distributed.run"}
    line = benchlib.compact(res, "/root/repo/bench_detail.json")
    assert len(json.dumps(line)) < 4096 and "truncated" not in line and line["validation_pass"] is True
    assert line["peak_busbw_GBps"]["8"] == pytest.approx(curves["8"]["peak_busbw_GBps"]) and line["peak_busbw_GBps_native"]["1"] == pytest.approx(curves["1"]["peak_algbw_GBps"])
    res["shed"] = ["phase-%d" % i for i in range(600)]
    line = benchlib.compact(res, "/root/repo/bench_detail.json")
    assert len(json.dumps(line)) < 4096 and line["truncated"] is True and "shed" not in line and line["value"] == 12345.67 and line["config"]["parallelism"] == "dp8"


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """WORLD_SIZE=1 with --gpus 2 used to run one rank and report n_gpus 1 with rc 0."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "64", "--steps", "1",
                        "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and '{"metric"' not in r.stdout


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_one_rank_rccl_takes_the_multi_gpu_path(tmp_path):
    """The N > 1 launch on real RCCL, as far as one GPU allows: torchrun, backend "nccl", the communication stream with
    the double-buffered dW all-reduce, the size sweep and the xGMI counter check, all on a one-rank communicator."""
    r = torchrun(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "6", "--warmup", "2",
                     "--probe-max-bytes", str(1 << 26), "--validation-phases", "xgmi", "--detail-out", str(tmp_path / "d.json")], timeout=280)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d, full = check_bench_line(r.stdout, 1, 6, 2)
    assert d["distributed_backend"] == "nccl" and d["allreduce_bytes_per_step"] == 8192 * 8192 * 2 and d["gemm_variant"] == 4
    assert d["value"] > 500, d["value"]                                  # the GEMMs ran at speed next to the collective
    assert d["warmup_steps_run"] > 100 and d["warmup_seconds"] >= 2.0     # warmed up by wall time, not by 2 steps
    assert "error" not in full["rccl_allreduce"] and full["rccl_allreduce"]["verified"], full["rccl_allreduce"]
    c1 = full["rccl_curves"]["1"]
    assert c1["executor"] == "torch.distributed/nccl" and c1["mode"] == "local_copy" and d["rccl_ranks_seen"]["communicators"] == [1]
    assert 50.0 < d["peak_busbw_GBps"]["1"] < 8000.0                     # a real local copy, bounded (64 MiB: cache-resident or not)
    assert full["xgmi_fidelity"]["world"] == 1 and full["xgmi_fidelity"]["expected_bytes_per_gpu"] == 0, full["xgmi_fidelity"]
    # the xGMI check's verdict is part of the validation report (and so of /api/validation)
    assert full["validation_detail"]["xgmi_fidelity"]["world"] == 1 and full["validation"]["xgmi_world"] == 1


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_bench_value_does_not_move_with_steps(tmp_path):
    """--steps 20 --warmup 5 (the driver's run) within 1.5 % of --steps 100 --warmup 10."""
    vals = {}
    for steps, warm in ((20, 5), (100, 10)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--no-validation", "--no-probe",
                            "--detail-out", str(tmp_path / f"d{steps}.json")], cwd=ROOT, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr[-2000:]
        d, _ = check_bench_line(r.stdout, 1, steps, warm)
        assert d["gemm_variant"] == 10 and d["gemm_variant_dp_kernel"] == 4 and d["value_dp_kernel"] > 0.9 * d["value"]
        vals[steps] = d["value"]
    assert abs(vals[20] - vals[100]) / vals[100] < 0.015, vals
