#!/bin/bash
# A/B of workgroups-per-CU variants on Config D (512 instances, N = 50): standard set, 256-thread co set, 512-thread M-in-L2 set at 2 and 4 waves/SIMD
set -e
mkdir -p gpurun_out/r5
export AB_WORKLOAD=D AB_STEP=1e-5 AB_MU=0.1 AB_WINDOWS=1
{
AB_SET=0 python scripts/dev_ab.py build_ab/libbase.so
AB_SET=1 python scripts/dev_ab.py build_ab/libbase.so
AB_SET=1 python scripts/dev_ab.py build_ab/lib512x2.so
AB_SET=1 python scripts/dev_ab.py build_ab/lib512x4.so
} > gpurun_out/r5/occ_D.log 2>&1
tail -20 gpurun_out/r5/occ_D.log
