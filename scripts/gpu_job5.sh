#!/bin/bash
D=gpurun_out/r4
mkdir -p $D
L=bilevel-gait-gen_amd
AB_WORKLOAD=B AB_STEP=0 AB_MU=0 AB_WINDOWS=1 timeout -k 10 400 python scripts/dev_ab.py $L/libsrbm_rti.so > $D/ab_ref.log 2>&1
grep "windows\|ms/step" $D/ab_ref.log | cut -c1-330
AB_WORKLOAD=B AB_STEP=0 AB_MU=0.1 AB_WINDOWS=1 timeout -k 10 400 python scripts/dev_ab.py $L/libsrbm_rti.so > $D/ab_ref_low.log 2>&1
grep "windows\|ms/step" $D/ab_ref_low.log | cut -c1-330
AB_WORKLOAD=B AB_STEP=0 AB_MU=1.0 AB_WINDOWS=1 timeout -k 10 400 python scripts/dev_ab.py $L/libsrbm_rti.so > $D/ab_ref_low1.log 2>&1
grep "windows\|ms/step" $D/ab_ref_low1.log | cut -c1-330
AB_WORKLOAD=B AB_STEP=1e-5 AB_MU=0.1 AB_WINDOWS=1 timeout -k 10 400 python scripts/dev_ab.py $L/libsrbm_rti.so > $D/ab_fast.log 2>&1
grep "windows\|ms/step" $D/ab_fast.log | cut -c1-330
