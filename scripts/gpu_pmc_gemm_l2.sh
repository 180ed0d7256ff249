#!/bin/bash
# rocprofv3 PMC passes over scripts/prof_gemm_one.py at a chosen size (KLM_PROF_N, KLM_ARMS): MFMA, fabric requests, L2 hits / misses,
# vector-L1 -> L2 requests. One pass per run. Output: gpurun_out/$1/pmc_summary.json
set -o pipefail
O=gpurun_out/${1:-pmc_gemm_l2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { timeout -s KILL 150 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_$1 -- python3 scripts/prof_gemm_one.py > $O/pmc_$1.log 2>&1 || { tail -5 $O/pmc_$1.log; exit 1; }; }
run mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT"
run hbm "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE GRBM_COUNT"
run l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE GRBM_COUNT"
run stallmem "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE GRBM_COUNT"
python3 scripts/summarize_pmc.py $O "gemm_bf16,gemm_mx,Cijk,gemm" > $O/pmc_summary.json 2> $O/summarize.err
python3 - <<PY
import json
d=json.load(open("$O/pmc_summary.json"))
for k,v in d.items():
    print(k[:70])
    for p,e in v.items():
        print("   ",p,{a:round(b,1) for a,b in e["per_dispatch"].items() if not a.startswith("GRBM_COUNT")})
PY
find $O -name "*.csv" -size +3M -delete
