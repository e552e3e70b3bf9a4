#!/bin/bash
# GPU scratch driver: iteration budget of the lower-start attempt (K3_LOW_CAP): Config B and the closed-loop segment through bench.py
mkdir -p gpurun_out/r3
rm -f gpurun_out/r3/cap_ab.log
for lib in bilevel-gait-gen_amd/ab/lib_cap14.so bilevel-gait-gen_amd/ab/lib_cap17.so bilevel-gait-gen_amd/ab/lib_cap20.so bilevel-gait-gen_amd/libsrbm_rti.so; do
  SRBM_RTI_LIB=$PWD/$lib python bench.py --no-cpu-baseline --gait-steps 0 --wbc-ticks 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'value %.0f  regions %s  mean its %.2f  solver %s  closed loop %.0f it/s not solved %d' % (d['value'], ' '.join('%.1f' % v for v in d['region_ms']), d['config']['mean_ipm_iterations'], {k: d['config']['solver'][k] for k in ('lower_start_attempts','attempts_repeated_from_standard_start')}, d['closed_loop']['rti_iterations_per_s'], d['closed_loop']['not_solved_all_steps']))" >> gpurun_out/r3/cap_ab.log
done
cat gpurun_out/r3/cap_ab.log
