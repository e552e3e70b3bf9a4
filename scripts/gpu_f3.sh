#!/bin/bash
# one gpurun call: the control-tick kernels (row f3) -- parity tests of the whole-body QP and the IK, then the bench's fourth segment under
# rocprofv3 --kernel-trace --stats (usage: gpurun -- bash scripts/gpu_f3.sh gpurun_out/r5b)
set -o pipefail
D=${1:-gpurun_out/f3}
mkdir -p $D
export TMPDIR=/tmp
timeout -k 10 240 python -m pytest tests/test_gpu_wbc.py tests/test_gpu_ik.py -q -s -x > $D/f3_tests.log 2>&1 < /dev/null
echo "tests rc=$?"; tail -3 $D/f3_tests.log; grep -a "IPM iterations" $D/f3_tests.log | head -1
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $D/trace_f3 -o trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --gait-steps 0 --closed-loop-steps 0 --steady-from 0 > $D/bench_f3.json 2> $D/bench_f3.err < /dev/null
echo "bench rc=$?"
python3 - $D <<'PY'
import json, sys, glob, csv
D = sys.argv[1]
try:
    d = json.loads([l for l in open(D + '/bench_f3.json') if l.startswith('{')][0])
    w = d['wbc']; print({k: w[k] for k in w if k != 'workload'})
except Exception as e:
    print('no bench line:', e)
for f in glob.glob(D + '/trace_f3/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r['Name'] for s in ('qp_control', 'targets_from', 'eval_traj')):
            print(r['Name'][:40], 'calls', r['Calls'], 'avg us', float(r['AverageNs']) / 1e3, 'min', float(r['MinNs']) / 1e3, 'max', float(r['MaxNs']) / 1e3)
PY
