#!/bin/bash
# Round 3's default GEMM kernels under rocprofv3: per size (8192, 6144, 2048) one run per PMC pass (never mixed with trace
# domains) and one kernel trace, over scripts/prof_gemm_r3.py. Output: gpurun_out/$1/n<size>/pmc_summary.json.
set -o pipefail
R=gpurun_out/${1:-pmc_r3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in ${KLM_PMC_SIZES:-8192 6144 2048}; do
  O=$R/n$n
  mkdir -p $O
  export KLM_PROF_N=$n
  for pass in mfma stall hbm; do
    C=$(python3 -c "from k8s_llm_monitor_amd.validation.rocprof import PASSES; print(' '.join(PASSES['$pass']))")
    echo "=== n=$n pmc $pass"
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$pass -- python3 scripts/prof_gemm_r3.py > $O/pmc_$pass.log 2>&1 || { tail -5 $O/pmc_$pass.log; exit 1; }
  done
  echo "=== n=$n lds pass"
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lds -- python3 scripts/prof_gemm_r3.py > $O/pmc_lds.log 2>&1 || { tail -5 $O/pmc_lds.log; exit 1; }
  echo "=== n=$n kernel trace"
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scripts/prof_gemm_r3.py > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
  python3 scripts/summarize_pmc.py $O "gemm_p128,gemm_quad,quad_partial,Cijk,gemm" > $O/pmc_summary.json 2> $O/summarize.err || { tail -5 $O/summarize.err; }
  find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
  find $O -name "*.csv" -size +1M -delete
done
ls -R $R | head -60
echo "=== done"
