#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1
timeout -k 10 500 python -u bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; rc=$?
tail -c 3000 gpurun_out/bench_full.json; echo; tail -5 gpurun_out/bench_full.err | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
