#!/bin/bash
# GPU scratch driver: the experimental co-resident build (256 threads, M in L2, 2 workgroups per CU) against the standard one on Config D / B
mkdir -p gpurun_out/r3
rm -f gpurun_out/r3/ab_co.log
L=bilevel-gait-gen_amd
for wl in D B; do
  for lib in $L/libsrbm_rti.so $L/ab/lib_co.so; do
    AB_WORKLOAD=$wl AB_WINDOWS=1 python scripts/dev_ab.py $lib >> gpurun_out/r3/ab_co.log 2>&1
  done
done
