#!/bin/bash
# Round profiles of bench.py on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r01
# Three rocprofv3 runs of the same command -- kernel trace + stats, then the two PMC passes on their own (the guide's
# HBM recipe: FETCH_SIZE and WRITE_SIZE in separate passes, no other tracing) -- plus one unprofiled run.  Raw output goes
# to gpurun_out/prof/, the summaries judged are written by scripts/summarize_pmc.py into profiles/<round>/ afterwards.
set -e -o pipefail
ROUND=${1:-r01}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --no-cpu-baseline --gait-steps 0 --closed-loop-steps 0"
python3 $CMD > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- python3 $CMD > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- python3 $CMD > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
