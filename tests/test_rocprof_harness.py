"""rocprofv3 harness logic on synthetic CSVs (the GPU runs themselves live in scripts/gpu_prof.sh)."""
import pytest

from k8s_llm_monitor_amd.validation import rocprof


def test_pass_slot_limits():
    for name, counters in rocprof.PASSES.items():
        rocprof.check_pass(counters)                           # every shipped pass fits
    with pytest.raises(ValueError, match="SQ_"):
        rocprof.check_pass([f"SQ_X{i}" for i in range(9)])
    with pytest.raises(ValueError, match="TCC_"):
        rocprof.check_pass(["FETCH_SIZE", "WRITE_SIZE"])      # 3 + 2 > 4 slots
    with pytest.raises(ValueError, match="GRBM_"):
        rocprof.check_pass(["GRBM_A", "GRBM_B", "GRBM_C"])
    rocprof.check_pass(["TCC_HIT_sum", "TCC_HIT_avr", "TCC_MISS_sum"])    # _sum/_avr of one counter count once


def test_busy_fraction_union_of_intervals():
    rows = [{"name": "gemm", "start": 0, "end": 10}, {"name": "gemm", "start": 5, "end": 20}, {"name": "gemm", "start": 50, "end": 60},
            {"name": "other", "start": 20, "end": 50}]
    b = rocprof.busy_fraction(rows, "gemm")
    assert b["kernels"] == 3 and b["busy_ns"] == 30 and b["span_ns"] == 60 and b["busy_fraction"] == 0.5
    assert rocprof.busy_fraction(rows, "gemm", t0=0, t1=100)["busy_fraction"] == 0.3
    assert rocprof.busy_fraction([], "x")["busy_fraction"] == 0.0


def test_read_csvs_and_derive(tmp_path):
    kt = tmp_path / "1_kernel_trace.csv"
    kt.write_text('"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"\n"KERNEL_DISPATCH","gemm_bf16_256(a, b)",100,300\n"KERNEL_DISPATCH","bad",x,y\n')
    assert rocprof.read_kernel_trace(str(kt)) == [{"name": "gemm_bf16_256(a, b)", "start": 100, "end": 300}]
    cc = tmp_path / "1_counter_collection.csv"
    lines = ['"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"']
    # the numbers of one real 8192^3 bf16 dispatch on MI355X (profiles/r1_prof/pmc_summary.json), two dispatches
    for d in (1, 2):
        for k, v in (("SQ_INSTS_VALU_MFMA_MOPS_BF16", 2147483648), ("SQ_VALU_MFMA_BUSY_CYCLES", 1073741824),
                     ("GRBM_GUI_ACTIVE", 11647072), ("GRBM_COUNT", 11647072), ("SQ_WAVE_CYCLES", 708680380), ("SQ_WAIT_ANY", 229680745),
                     ("TCC_EA0_RDREQ_sum", 16777528), ("TCC_EA0_RDREQ_32B_sum", 0), ("TCC_EA0_WRREQ_sum", 33557412), ("TCC_EA0_WRREQ_64B_sum", 33557412)):
            lines.append(f'{d},"gemm_bf16_256(x)","{k}",{v}')
    cc.write_text("\n".join(lines) + "\n")
    per = rocprof.read_counters(str(cc))
    c = per["gemm_bf16_256(x)"]
    assert c["dispatches"] == 2 and c["GRBM_COUNT"] == 2 * 11647072
    d = rocprof.derive(c)
    assert d["mfma_flop"] == 2.0 * 8192 ** 3                       # MOPS x 512 is exact
    assert d["mfma_busy_share"] == pytest.approx(0.72, abs=0.01) and d["gpu_busy_share"] == 1.0
    assert d["wait_share"] == pytest.approx(0.324, abs=0.001)
    assert d["hbm_read_bytes"] == pytest.approx(2 * 2 ** 30, rel=1e-4) and d["hbm_write_bytes"] == pytest.approx(2 * 2 ** 30, rel=1e-4)


def test_burner_kernels_are_recognised_under_every_name_they_had():
    names = ["void gemm_bf16_256_t<false, false, 1>(...)", "gemm_bf16_128(...)", "void gemm_quad_asm<0>(void const*, ...)",
             "void gemm_quad_asm_p<0>(void const*, void const*, unsigned short*, int, int, int, int)"]
    assert all(rocprof.is_burner_gemm(n) for n in names)
    assert not rocprof.is_burner_gemm("void hbm_fill_tile_kernel<false, 1>(...)") and not rocprof.is_burner_gemm("Cijk_Alik_Bljk_BBS")
    rows = [{"name": names[3], "start": 1000 * i, "end": 1000 * i + 700} for i in range(40)] + [{"name": "other", "start": 5, "end": 9}]
    d = rocprof.kernel_durations(rows, rocprof.is_burner_gemm, skip=16)
    assert len(d) == 24 and d[0] == 0.7
    assert rocprof.kernel_durations(rows, "other", skip=0) == [0.004]


def test_burner_cmd_has_no_shell_or_env_hop():
    cmd = rocprof.burner_cmd(duty=50)
    assert cmd[0].endswith("python") or "python" in cmd[0]
    assert cmd[1:4] == ["-m", "k8s_llm_monitor_amd", "burner"] and "env" not in cmd and "bash" not in cmd
