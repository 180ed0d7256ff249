#!/bin/bash
# Profiling session: HBM variant sweep, kernel-trace stats of the bench, raw PMC passes of the
# hot kernels (one pass per run, never mixed with trace domains), duty cross-check.
set -o pipefail
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
echo "=== hbm sweep"; timeout -k 10 300 python -u scripts/hbm_sweep.py > $O/hbm_sweep.log 2>&1 || { tail -5 $O/hbm_sweep.log; exit 1; }
tail -3 $O/hbm_sweep.log
echo "=== kernel trace of bench"; timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --no-validation --steps 50 --warmup 5 > $O/bench_trace.log 2>&1 || { tail -5 $O/bench_trace.log; exit 1; }
tail -1 $O/bench_trace.log
for pass in mfma stall hbm; do
  C=$(python3 -c "from k8s_llm_monitor_amd.validation.rocprof import PASSES; print(' '.join(PASSES['$pass']))")
  echo "=== pmc $pass: $C"
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$pass -- python3 scripts/prof_gemm.py > $O/pmc_$pass.log 2>&1 || { tail -5 $O/pmc_$pass.log; exit 1; }
done
echo "=== duty cross-check"; timeout -k 10 300 python3 -u scripts/rocprof_crosscheck.py $O/duty > $O/duty.log 2>&1 || { tail -8 $O/duty.log; exit 1; }
cat $O/duty.log | tail -4
# keep only the small summaries
find $O -name "*.csv" -size +3M -delete
echo "=== done"
