#!/bin/bash
# Round-end record: smoke, the whole bench line, every GPU test, a rocprofv3 kernel trace of the bench's timed loop.
set -o pipefail
O=gpurun_out/${1:-final}
mkdir -p $O
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 500 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -20 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --no-validation --no-probe --steps 50 --warmup 5 > $O/bench_trace.log 2>&1 || { tail -5 $O/bench_trace.log; exit 1; }
cp $(find $O/bench_trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
head -6 $O/bench_kernel_stats.csv
find $O -name "*.csv" -size +2M -delete
