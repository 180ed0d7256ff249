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
    # exactly ONE JSON line, from rank 0 (gloo's C++ layer prints its own "[Gloo] Rank ..." chatter on CPU)
    lines = [l[l.index('{"metric"'):] for l in stdout.splitlines() if '{"metric"' in l]
    assert len(lines) == 1, stdout
    d = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert isinstance(d[k], t), (k, d[k])
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md has no number to divide by
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "bf16" and "synthetic" in d["data"]
    assert set(d["config"]) >= {"model", "global_batch", "seq_len", "parallelism"} and d["config"]["parallelism"] == (f"dp{n}" if n > 1 else "single")
    # value is the whole-job aggregate: n ranks x FLOP per step / step time
    assert d["value"] == pytest.approx(n * d["flop_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12, rel=2e-2, abs=0.006)   # printed with 2 decimals
    return d


@pytest.mark.timeout(240)
def test_bench_contract_single_process_rehearsal():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-rehearsal", "--size", "256", "--steps", "3", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    check_bench_line(r.stdout, 1, 3, 1)


@pytest.mark.timeout(240)
def test_bench_contract_two_ranks_rehearsal():
    r = torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "256", "--steps", "4", "--warmup", "2"])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = check_bench_line(r.stdout, 2, 4, 2)
    assert d["allreduce_bytes_per_step"] == 256 * 256 * 2 and d["allreduce_busbw_GBps_overlapped"] > 0


@pytest.mark.timeout(240)
def test_bench_launches_its_own_ranks():
    """Plain ``python bench.py --gpus 2`` (no torchrun): the parent starts the ranks as a child, relays ONE JSON line,
    and the line proves two ranks ran and carries the curve at n = 1 AND n = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "256",
                        "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=220)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = check_bench_line(r.stdout, 2, 3, 1)
    assert d["launcher"].startswith("bench.py self-launch")
    assert d["rccl_ranks_seen"] == {"process_group_world_size": 2, "communicators": [1, 2]}
    assert sorted(d["rccl_curves"]) == ["1", "2"]
    c1, c2 = d["rccl_curves"]["1"], d["rccl_curves"]["2"]
    assert c1["verified"] and c2["verified"] and c1["ranks_seen"] == 1 and c2["ranks_seen"] == 2
    assert c1["curve_key"] == "algbw_GBps" and c2["curve_key"] == "busbw_GBps" and c2["ceiling_mesh_GBps"] == 153.0
    assert [b for b, _ in c2["curve"]] == [1024, 4096, 16384, 65536] and c2["curve"][-1][1] > 0
    assert "rccl_curves_native" not in d                                 # the C++ probe needs GPUs


@pytest.mark.timeout(240)
def test_bench_four_ranks_give_the_1_2_4_curve():
    r = torchrun(4, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--cpu-rehearsal", "--size", "128", "--steps", "2", "--warmup", "1",
                     "--probe-max-bytes", "4096"])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = check_bench_line(r.stdout, 4, 2, 1)
    assert d["launcher"] == "torch.distributed.run"
    assert sorted(d["rccl_curves"]) == ["1", "2", "4"] and d["rccl_ranks_seen"]["communicators"] == [1, 2, 4]
    assert d["rccl_curves"]["4"]["ceiling_mesh_GBps"] == 3 * 153.0 and d["rccl_allreduce"]["n_gpus"] == 4


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """WORLD_SIZE=1 with --gpus 2 used to run one rank and report n_gpus 1 with rc 0."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-rehearsal", "--size", "64", "--steps", "1",
                        "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and '{"metric"' not in r.stdout


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_one_rank_rccl_takes_the_multi_gpu_path():
    """The N > 1 launch on real RCCL, as far as one GPU allows: torchrun, backend "nccl", the communication stream with
    the double-buffered dW all-reduce, the size sweep and the xGMI counter check, all on a one-rank communicator."""
    r = torchrun(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "6", "--warmup", "2",
                     "--probe-max-bytes", str(1 << 24), "--validation-phases", "xgmi"], timeout=280)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [l[l.index('{"metric"'):] for l in r.stdout.splitlines() if '{"metric"' in l]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["distributed_backend"] == "nccl" and d["allreduce_bytes_per_step"] == 8192 * 8192 * 2
    assert d["value"] > 500, d["value"]                                  # the GEMMs ran at speed next to the collective
    assert "error" not in d["rccl_allreduce"] and d["rccl_allreduce"]["verified"], d["rccl_allreduce"]
    assert d["rccl_curves"]["1"]["executor"] == "torch.distributed/nccl" and d["rccl_ranks_seen"]["communicators"] == [1]
    assert d["xgmi_fidelity"]["world"] == 1 and d["xgmi_fidelity"]["expected_bytes_per_gpu"] == 0, d["xgmi_fidelity"]
