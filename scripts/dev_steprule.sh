#!/bin/bash
# GPU scratch driver: accuracy (re-synchronised, identical QPs) and time of the step rule / lower start variants
set -o pipefail
mkdir -p gpurun_out/r3
AB_STEP=1e-5 python scripts/dev_trace.py 11 8 > gpurun_out/r3/trace_11.log 2>&1 || exit 1
AB_STEP=1e-5 AB_MU=100 python scripts/dev_trace.py 11 8 > gpurun_out/r3/trace_11_low.log 2>&1 || exit 1
AB_STEP=1e-5 python scripts/dev_trace.py 77 8 > gpurun_out/r3/trace_77.log 2>&1 || exit 1
python scripts/dev_accuracy.py 256 6 "1e-15:0:0,1e-15:1e-5:0,1e-15:3e-5:0,1e-15:1e-5:100,1e-15:3e-5:100" > gpurun_out/r3/acc_step.log 2>&1 || exit 1
LIB=bilevel-gait-gen_amd/libsrbm_rti.so
rm -f gpurun_out/r3/ab_step.log
for v in "0 0" "1e-5 0" "3e-5 0" "1e-5 1000" "1e-5 100" "3e-5 100"; do
  set -- $v
  AB_WINDOWS=1 AB_STEP=$1 AB_MU=$2 python scripts/dev_ab.py $LIB >> gpurun_out/r3/ab_step.log 2>&1 || exit 1
done
