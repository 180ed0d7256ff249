#!/bin/bash
# rocprofv3 over the mid-batch decode path (batch 48 and 128, Llama-3-8B): kernel-time breakdown, then one counter
# pass per run (never mixed with trace domains) for MFMA-busy, stalls / LDS conflicts and HBM bytes.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_mid
mkdir -p $O
for cfg in "decode 48 1024 10" "decode 128 1024 10"; do
  tag=$(echo $cfg | tr ' ' '_')
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/scripts/prof_llm.py $cfg ) > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; exit 1; }
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/${tag}_kernel_stats.csv
  echo "== $cfg"; tail -1 $O/$tag.log | cut -c1-300
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:10]:
    print("%6.2f%% %8d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total kernel ms", tot / 1e6)
PY
done
for cfg in "decode 48 1024 3" "decode 128 1024 3"; do
  tag=$(echo $cfg | tr ' ' '_')
  for pass in mfma stall hbm; do
    C=$(python3 -c "from k8s_llm_monitor_amd.validation.rocprof import PASSES; print(' '.join(PASSES['$pass']))")
    ( cd /tmp && timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/$tag/pmc_$pass -- python3 $R/scripts/prof_llm.py $cfg ) > $O/${tag}_$pass.log 2>&1 || { tail -5 $O/${tag}_$pass.log; exit 1; }
  done
  python3 scripts/summarize_pmc.py $O/$tag mid_linear,mid_epilogue,mid_res_rms,paged_attn,rope_cache > $O/${tag}_summary.json || exit 1
  echo "== $cfg"; python3 - $O/${tag}_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    m, h, st = v.get("pmc_mfma", {}).get("derived", {}), v.get("pmc_hbm", {}).get("derived", {}), v.get("pmc_stall", {}).get("derived", {})
    print("%-70s n=%5d cyc=%9.0f mfma_busy=%5.3f rdMB=%8.2f wrMB=%7.2f lds_conf=%s" % (
        k[-70:], v.get("pmc_hbm", {}).get("dispatches", 0), h.get("gfx_cycles_per_dispatch") or 0, m.get("mfma_busy_share") or 0,
        (h.get("hbm_read_bytes") or 0) / 1e6, (h.get("hbm_write_bytes") or 0) / 1e6, st.get("lds_conflict_per_active")))
PY
done
find $O -name "*.csv" -size +2M -delete
