#!/bin/bash
D=gpurun_out/r4
mkdir -p $D
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -x -q > $D/dense.log 2>&1
tail -5 $D/dense.log
timeout -k 10 120 python scripts/dev_trtri.py > $D/trtri.log 2>&1
tail -8 $D/trtri.log
SRBM_LIB=ab/libstamps.so timeout -k 10 120 python scripts/dev_trtri.py > $D/stamps_fi.log 2>&1
grep stamps $D/stamps_fi.log
