#!/bin/bash
# one gpurun call of the round's routine: GPU test suite, then the bench line (logs under gpurun_out/<dir>).  The bench runs even when a test fails.
set -o pipefail
D=${1:-gpurun_out/r5}
mkdir -p $D
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $D/gpu_tests.log 2>&1
rc=$?
tail -8 $D/gpu_tests.log
# a crashed or killed test run (GPU fault, timeout): no further GPU step in this call
if [ $rc -ge 124 ] || grep -q "Memory access fault" $D/gpu_tests.log; then echo "test run crashed (rc $rc): bench not started"; exit 1; fi
timeout -k 10 400 python bench.py > $D/bench.json 2> $D/bench.err || { tail -20 $D/bench.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open('$D/bench.json') if l.startswith('{')][0])
print('value', d['value'], d['solver_mode'], 'ms/step', d['ms_per_step'], 'regions', d['region_ms'], 'median-based', d['median_region']['value'], 'steady', d.get('steady_state',{}).get('value'))
for k in ('reference_criterion','step_rule_mode','one_region','config_d','config_e'):
    if k in d: print(k, d[k]['value'], d[k]['value_from_median_region'], d[k]['ms_per_step'], d[k]['mean_ipm_iterations'], d[k]['all_solved'])
if 'config_d' in d and 'step_rule_mode' in d['config_d']: print('config_d step rule', d['config_d']['step_rule_mode']['value'], d['config_d']['step_rule_mode']['value_from_median_region'])
print('gait', d.get('gait',{}).get('ms_per_step'), 'gait step-rule', d.get('gait',{}).get('step_rule_mode',{}).get('ms_per_step'), 'cl', d.get('closed_loop',{}).get('rti_iterations_per_s'), 'wbc', d.get('wbc',{}).get('device_resident',{}).get('ms_per_tick_of_the_batch'))
print('roofline', {k:d['roofline'][k] for k in ('frac','executed_mfma_frac_of_peak','avg_launch_ms','ipm_iterations_per_solve')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('all_cores',{}).get('value'), d.get('cpu_baseline',{}).get('like_for_like'))
PY
exit $rc
