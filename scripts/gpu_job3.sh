#!/bin/bash
D=gpurun_out/r4
mkdir -p $D
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -x -q > $D/dense.log 2>&1
tail -3 $D/dense.log
timeout -k 10 300 python scripts/dev_chol_ab.py bilevel-gait-gen_amd/ab/libchol2b.so bilevel-gait-gen_amd/libsrbm_rti.so > $D/chol_ab.log 2>&1
grep "ticks\|DIFF" $D/chol_ab.log
SRBM_LIB=ab/libstamps.so NS=32,108 timeout -k 10 120 python scripts/dev_chol.py > $D/stamps_chol.log 2>&1
grep stamps $D/stamps_chol.log
timeout -k 10 600 python -m pytest tests/test_gpu_ownpath.py -x -q -s > $D/ownpath_test.log 2>&1
tail -3 $D/ownpath_test.log
grep "own path\|instances beyond" $D/ownpath_test.log | cut -c1-900
