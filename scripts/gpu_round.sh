#!/bin/bash
# One GPU session: GPU tests, the ops check, the validation phases, the bench and a rocprof
# kernel trace. Every step has its own time limit; the chain stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { echo "=== $1 ($(date +%T))"; }
if [ -z "$SKIP_TESTS" ]; then
step tests && timeout -k 10 420 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -40 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc"; exit $rc; }
fi
step check_ops && timeout -k 10 180 python -u scripts/gpu_check_ops.py > gpurun_out/check_ops.log 2>&1; rc=$?
tail -25 gpurun_out/check_ops.log
[ $rc -ne 0 ] && { echo "check_ops rc=$rc"; exit $rc; }
step validate && KLM_STATE_DIR=gpurun_out timeout -k 10 400 python -u -m k8s_llm_monitor_amd validate --quick --out gpurun_out/validation.json > gpurun_out/validate.log 2>&1; rc=$?
tail -15 gpurun_out/validate.log
[ $rc -ne 0 ] && { echo "validate rc=$rc"; exit $rc; }
step bench && timeout -k 10 300 python -u bench.py --no-validation > gpurun_out/bench.log 2>&1; rc=$?
tail -3 gpurun_out/bench.log
[ $rc -ne 0 ] && { echo "bench rc=$rc"; exit $rc; }
echo "=== done"
