#!/bin/bash
# one gpurun call: A/B of library builds under build_ab/ on Config B and D in the bench's solver mode (usage: gpurun -- bash scripts/gpu_ab.sh libA.so libB.so ...)
mkdir -p gpurun_out/ab
for wl in B D; do
  AB_WORKLOAD=$wl AB_STEP=0 AB_MU=0.1 AB_WINDOWS=1 timeout -k 10 300 python scripts/dev_ab.py "$@" < /dev/null 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/ab_$wl.log
done
