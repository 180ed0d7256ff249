#!/bin/bash
# kernel-time breakdown of one decode configuration and one prefill
set -o pipefail
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof_llm
for cfg in "decode 1 1024 20" "decode 64 1024 20" "prefill 1 2048 3"; do
  tag=$(echo $cfg | tr ' ' '_')
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/scripts/prof_llm.py $cfg ) > $R/gpurun_out/prof_llm/$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_llm/$tag.log; exit 1; }
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/prof_llm/${tag}_kernel_stats.csv
  echo "== $cfg"; tail -2 $R/gpurun_out/prof_llm/$tag.log | cut -c1-300
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%6.2f%% %8d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print("total kernel ms", tot / 1e6)
PY
done
