#!/bin/bash
# GPU scratch driver: library variants of the IPM (build flags) -- time on Config B (the bench's windows) and the strict parity tests
mkdir -p gpurun_out/r3
L=bilevel-gait-gen_amd
rm -f gpurun_out/r3/rule_ab.log
for lib in "$@"; do
  AB_WINDOWS=1 python scripts/dev_ab.py $lib 2>&1 | grep -E "windows|counters" | cut -c1-260 >> gpurun_out/r3/rule_ab.log
  SRBM_RTI_LIB=$PWD/$lib python -m pytest -q -m gpu tests/test_gpu_resync.py::test_config_b_all_256_instances_entrywise_over_20_steps tests/test_gpu_parity.py::test_full_batch_minimisers_on_identical_qps 2>&1 | grep -E "^E   *Assertion|^E   *assert np|passed|failed" | cut -c1-200 >> gpurun_out/r3/rule_ab.log
done
cat gpurun_out/r3/rule_ab.log
