#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-fill_trace}
mkdir -p $O
timeout -k 10 120 python3 -u scripts/exp/fill_vs_torch.py $O/rates.json || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 scripts/exp/fill_vs_torch.py $O/rates_traced.json > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:90], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(k, []).append(d)
for k, v in agg.items():
    v.sort()
    print(k, "n", len(v), "median_us", round(v[len(v)//2], 1))
PY
find $O -name "*.csv" -size +2M -delete
