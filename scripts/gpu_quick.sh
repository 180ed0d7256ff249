#!/bin/bash
# Quick GPU pass: ops check, HBM sweep, ops tests.
set -o pipefail
mkdir -p gpurun_out/prof
timeout -k 10 200 python -u scripts/gpu_check_ops.py > gpurun_out/check_ops.log 2>&1 || { tail -20 gpurun_out/check_ops.log; exit 1; }
tail -24 gpurun_out/check_ops.log | head -18
timeout -k 10 200 python -u scripts/hbm_sweep.py > gpurun_out/prof/hbm_sweep2.log 2>&1 || { tail -20 gpurun_out/prof/hbm_sweep2.log; exit 1; }
tail -12 gpurun_out/prof/hbm_sweep2.log | cut -c1-200
timeout -k 10 200 python -u -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
