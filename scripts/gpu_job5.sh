#!/bin/bash
D=gpurun_out/r4
mkdir -p $D
timeout -k 10 600 python -m pytest tests/test_gpu_gait.py tests/test_cpp_facade.py tests/test_gpu_bench.py -m gpu -x -q > $D/gait_tests.log 2>&1
tail -4 $D/gait_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --closed-loop-steps 0 --wbc-ticks 0 > $D/bench_gait.json 2> $D/bench_gait.err
python - <<PY
import json
d=json.loads([l for l in open('$D/bench_gait.json') if l.startswith('{')][0])
print('value', d['value'], 'gait', d['gait'])
PY
