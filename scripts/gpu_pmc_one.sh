#!/bin/bash
# mfma + stall PMC passes over scripts/prof_gemm_one.py (arms from KLM_ARMS); output gpurun_out/$1/pmc_summary.json
set -o pipefail
O=gpurun_out/${1:-pmc_one}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pass in mfma stall; do
  C=$(python3 -c "from k8s_llm_monitor_amd.validation.rocprof import PASSES; print(' '.join(PASSES['$pass']))")
  timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$pass -- python3 ${KLM_PROF_SCRIPT:-scripts/prof_gemm_one.py} > $O/pmc_$pass.log 2>&1 || { tail -5 $O/pmc_$pass.log; exit 1; }
done
python3 scripts/summarize_pmc.py $O "gemm_bf16,gemm_mx,Cijk,gemm" > $O/pmc_summary.json 2> $O/summarize.err
python3 - <<PY
import json
d=json.load(open("$O/pmc_summary.json"))
for k,v in d.items():
    dd={}
    for p,e in v.items(): dd.update(e["derived"])
    print(k[:60], {a:round(b,3) for a,b in dd.items() if a in ("mfma_busy_share","wait_share","issue_stall_share","active_share","gfx_cycles_per_dispatch")})
PY
find $O -name "*.csv" -size +3M -delete
