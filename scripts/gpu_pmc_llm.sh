#!/bin/bash
# rocprofv3 counter passes (one pass per run, never mixed with trace domains) over the LLM kernels.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_llm
mkdir -p $O
for cfg in "decode 1 1024 4" "decode 64 1024 4" "prefill 1 4096 2"; do
  tag=$(echo $cfg | tr ' ' '_')
  for pass in mfma stall hbm; do
    C=$(python3 -c "from k8s_llm_monitor_amd.validation.rocprof import PASSES; print(' '.join(PASSES['$pass']))")
    ( cd /tmp && timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/$tag/pmc_$pass -- python3 $R/scripts/prof_llm.py $cfg ) > $O/${tag}_$pass.log 2>&1 || { tail -5 $O/${tag}_$pass.log; exit 1; }
  done
  python3 scripts/summarize_pmc.py $O/$tag skinny_linear,paged_attn,rmsnorm,rope_cache,silu_mul > $O/${tag}_summary.json || exit 1
  echo "== $cfg"; python3 - $O/${tag}_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    m, h, st = v.get("pmc_mfma", {}).get("derived", {}), v.get("pmc_hbm", {}).get("derived", {}), v.get("pmc_stall", {}).get("derived", {})
    print("%-60s n=%5d cyc=%9.0f mfma_busy=%5.3f rdMB=%8.2f wrMB=%7.2f lds_conf=%s" % (
        k[-60:], v.get("pmc_hbm", {}).get("dispatches", 0), h.get("gfx_cycles_per_dispatch") or 0, m.get("mfma_busy_share") or 0,
        (h.get("hbm_read_bytes") or 0) / 1e6, (h.get("hbm_write_bytes") or 0) / 1e6, st.get("lds_conflict_per_active")))
PY
done
find $O -name "*.csv" -size +2M -delete
