#!/bin/bash
# fp4 MFMA-busy share against K (8192 x 8192 x K): one rocprofv3 --pmc run (the mfma pass) per K. Output: gpurun_out/$1/k<K>/.
set -o pipefail
R=gpurun_out/${1:-pmc_fp4_k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=$(python3 -c "from k8s_llm_monitor_amd.validation.rocprof import PASSES; print(' '.join(PASSES['mfma']))")
for k in 4096 8192 16384 32768; do
  O=$R/k$k
  mkdir -p $O
  export KLM_PROF_K=$k
  echo "=== K=$k"
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pmc_mfma -- python3 scripts/prof_gemm_fp4_k.py > $O/pmc_mfma.log 2>&1 || { tail -5 $O/pmc_mfma.log; exit 1; }
  python3 scripts/summarize_pmc.py $O "gemm_quad" > $O/pmc_summary.json 2> $O/summarize.err || { tail -5 $O/summarize.err; }
  find $O -name "*.csv" -size +1M -delete
done
echo "=== done"
