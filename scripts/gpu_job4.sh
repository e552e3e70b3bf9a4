#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 300 python scripts/dev_repeat_check.py 2>&1 | tee gpurun_out/r4/repeat_check.log | tail -20
