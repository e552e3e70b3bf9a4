#!/bin/bash
# Round profiles of bench.py on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh
# One unprofiled run, then rocprofv3 runs of the same command: kernel trace + stats, and the PMC passes each ON ITS OWN (the
# guide's recipe: FETCH_SIZE and WRITE_SIZE cannot share a pass; no tracing beside --pmc; the program itself directly after `--`):
#   pmc_fetch  FETCH_SIZE                      pmc_write  WRITE_SIZE
#   pmc_sq1    SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
#   pmc_sq2    SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_WAIT_INST_LDS
# Raw output goes to gpurun_out/prof/, the summaries judged are written by scripts/summarize_pmc.py into profiles/<round>/.
set -o pipefail
OUT=gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --gait-steps 0 --closed-loop-steps 0 --wbc-ticks 0"
python3 $CMD > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err || exit 1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
pass() {   # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- python3 $CMD > $OUT/bench_$name.json 2> $OUT/$name.err
    echo "$name rc=$?" >> $OUT/passes.log
}
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass pmc_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU
# the control-tick kernels of row f3 (targets from the trajectory, whole-body QP): kernel trace of the bench WITH its fourth segment
rocprofv3 --kernel-trace --stats -d $OUT/trace_f3 -o trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --gait-steps 0 --closed-loop-steps 0 > $OUT/bench_f3_under_rocprof.json 2> $OUT/trace_f3.err
echo "trace_f3 rc=$?" >> $OUT/passes.log
# kernel stats of the other workloads and segments, occupancy / wait split of Config D, PMC of the control-tick kernels
rocprofv3 --kernel-trace --stats -d $OUT/trace_D -o trace --output-format csv -- python3 bench.py --workload D --no-cpu-baseline --closed-loop-steps 0 > $OUT/bench_D_under_rocprof.json 2> $OUT/trace_D.err
echo "trace_D rc=$?" >> $OUT/passes.log
rocprofv3 --kernel-trace --stats -d $OUT/trace_E -o trace --output-format csv -- python3 bench.py --workload E --no-cpu-baseline --closed-loop-steps 0 > $OUT/bench_E_under_rocprof.json 2> $OUT/trace_E.err
echo "trace_E rc=$?" >> $OUT/passes.log
rocprofv3 --kernel-trace --stats -d $OUT/trace_gait -o trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --closed-loop-steps 0 --wbc-ticks 0 > $OUT/bench_gait_under_rocprof.json 2> $OUT/trace_gait.err
echo "trace_gait rc=$?" >> $OUT/passes.log
CMD_D="bench.py --workload D --no-cpu-baseline --closed-loop-steps 0"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_D_sq1 -o pmc --output-format csv -- python3 $CMD_D > $OUT/bench_pmc_D_sq1.json 2> $OUT/pmc_D_sq1.err
echo "pmc_D_sq1 rc=$?" >> $OUT/passes.log
CMD_F3="bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --gait-steps 0 --closed-loop-steps 0"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_f3_sq1 -o pmc --output-format csv -- python3 $CMD_F3 > $OUT/bench_pmc_f3_sq1.json 2> $OUT/pmc_f3_sq1.err
echo "pmc_f3_sq1 rc=$?" >> $OUT/passes.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_f3_sq2 -o pmc --output-format csv -- python3 $CMD_F3 > $OUT/bench_pmc_f3_sq2.json 2> $OUT/pmc_f3_sq2.err
echo "pmc_f3_sq2 rc=$?" >> $OUT/passes.log
python3 scripts/dev_prof.py > $OUT/phase_shares.txt 2> $OUT/phase_shares.err
PROF_MODE=ref python3 scripts/dev_prof.py > $OUT/phase_shares_reference_criterion.txt 2>> $OUT/phase_shares.err
echo "dev_prof rc=$?" >> $OUT/passes.log
SRBM_RTI_LIB=bilevel-gait-gen_amd/libsrbm_rti_prof.so python3 scripts/dev_prof_wbc.py > $OUT/wbc_phase_shares.txt 2>> $OUT/phase_shares.err
echo "dev_prof_wbc rc=$?" >> $OUT/passes.log
cat $OUT/passes.log
find $OUT -name "*.csv" | sort
