#!/bin/bash
D=gpurun_out/r4
mkdir -p $D
timeout -k 10 300 python scripts/dev_ownpath.py > $D/ownpath.log 2>&1
timeout -k 10 300 python scripts/dev_chol_ab.py bilevel-gait-gen_amd/ab/libchol2b.so bilevel-gait-gen_amd/libsrbm_rti.so > $D/chol_ab.log 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -x -q > $D/dense.log 2>&1
tail -3 $D/dense.log
cat $D/chol_ab.log | tail -40
