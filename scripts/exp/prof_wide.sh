set -o pipefail
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof_wide
for cfg in "decode 256 1024 5" "decode 192 1024 5"; do
  tag=$(echo $cfg | tr ' ' '_')
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/scripts/prof_llm.py $cfg ) > $R/gpurun_out/prof_wide/$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_wide/$tag.log; exit 1; }
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/prof_wide/${tag}_kernel_stats.csv
  echo "== $cfg"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%6.2f%% %8d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
done
