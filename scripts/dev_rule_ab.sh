#!/bin/bash
# GPU scratch driver: variants of the step rule (dual criterion, refinement of the last step): time on Config B and the strict parity tests
mkdir -p gpurun_out/r3
L=bilevel-gait-gen_amd
rm -f gpurun_out/r3/rule_ab.log
for lib in $L/ab/lib_d0_r0.so $L/ab/lib_d0_r1.so $L/ab/lib_d1_r0.so $L/libsrbm_rti.so; do
  AB_WINDOWS=1 python scripts/dev_ab.py $lib 2>&1 | grep windows >> gpurun_out/r3/rule_ab.log
  SRBM_RTI_LIB=$PWD/$lib python -m pytest -q -m gpu tests/test_gpu_resync.py::test_config_b_all_256_instances_entrywise_over_20_steps tests/test_gpu_resync.py::test_config_c_values_at_n20_entrywise tests/test_gpu_parity.py::test_full_batch_minimisers_on_identical_qps 2>&1 | grep -E "^E   *Assertion|^E   *assert np|passed|failed" | cut -c1-200 >> gpurun_out/r3/rule_ab.log
done
cat gpurun_out/r3/rule_ab.log
