#!/bin/bash
# Timing-only builds of the bf16 ring kernel with some LDS reads removed (outputs are wrong):
# how much does LDS read traffic cost under the power cap?
set -e
S=k8s_llm_monitor_amd/ops/csrc
O=k8s_llm_monitor_amd/_native/lib
for v in "base:" "skipb1:-DKLM_EXP_SKIP_B1" "skipa1:-DKLM_EXP_SKIP_A1" "skipboth:-DKLM_EXP_SKIP_B1 -DKLM_EXP_SKIP_A1" "noglobal:-DKLM_EXP_NO_GLOBAL" "mfmaonly:-DKLM_EXP_NO_GLOBAL -DKLM_EXP_SKIP_B1 -DKLM_EXP_SKIP_A1"; do
  n=${v%%:*}; f=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -shared -fPIC $f $S/gemm_bf16.hip -I$S -o $O/libklm_exp_$n.so
done
ls -la $O
