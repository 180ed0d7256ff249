#!/bin/bash
# rocprofv3 PMC passes over the HBM kernels (scripts/prof_hbm.py): request bytes at the L2's fabric port, and the port's
# stall counters (why a copy runs below a pure read). One pass per run. Output: gpurun_out/$1/pmc_summary.json
set -o pipefail
O=gpurun_out/${1:-pmc_hbm}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_$1 -- python3 scripts/prof_hbm.py > $O/pmc_$1.log 2>&1 || { tail -5 $O/pmc_$1.log; exit 1; }; }
run hbm "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE GRBM_COUNT"
run stallmem "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT"
run level "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_CYCLE_sum TCC_BUSY_sum GRBM_GUI_ACTIVE GRBM_COUNT"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scripts/prof_hbm.py > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 scripts/summarize_pmc.py $O "hbm_,elementwise,copy" > $O/pmc_summary.json 2> $O/summarize.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
find $O -name "*.csv" -size +2M -delete
head -12 $O/kernel_stats.csv
