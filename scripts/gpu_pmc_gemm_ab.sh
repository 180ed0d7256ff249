#!/bin/bash
# rocprofv3 PMC passes (one pass per run, never mixed with trace domains) over a workload script ($2, default scripts/prof_gemm_ab.py),
# then a kernel trace for the durations. Output: gpurun_out/$1/.
set -o pipefail
O=gpurun_out/${1:-pmc_ab}
W=${2:-scripts/prof_gemm_ab.py}      # the workload: any script that launches the kernels back to back
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pass in mfma stall hbm; do
  C=$(python3 -c "from k8s_llm_monitor_amd.validation.rocprof import PASSES; print(' '.join(PASSES['$pass']))")
  echo "=== pmc $pass: $C"
  timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$pass -- python3 $W > $O/pmc_$pass.log 2>&1 || { tail -5 $O/pmc_$pass.log; exit 1; }
done
echo "=== kernel trace"
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $W > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
echo "=== lds pass (optional)"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lds -- python3 $W > $O/pmc_lds.log 2>&1 || { tail -5 $O/pmc_lds.log; rm -rf $O/pmc_lds; }
python3 scripts/summarize_pmc.py $O "gemm_bf16,gemm_mx,gemm_quad,Cijk,gemm" > $O/pmc_summary.json 2> $O/summarize.err || { tail -5 $O/summarize.err; }
find $O -name "*.csv" -size +3M -delete
ls $O
echo "=== done"
