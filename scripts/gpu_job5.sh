#!/bin/bash
D=gpurun_out/r4
mkdir -p $D
L=bilevel-gait-gen_amd
for wl in D B; do
  AB_WORKLOAD=$wl AB_STEP=1e-5 AB_MU=0.1 AB_WINDOWS=1 timeout -k 10 400 python scripts/dev_ab.py $L/ab/libr03.so $L/libsrbm_rti.so $L/ab/libr03.so $L/libsrbm_rti.so 2>&1 | grep windows | sed "s/^/$wl: /" | cut -c1-200
done
