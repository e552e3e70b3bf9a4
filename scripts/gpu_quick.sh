#!/bin/bash
# quick GPU check of a change in the dense phase: unit tests of the dense blocks, stamps of the triangular inverse (if the A/B library exists),
# Config-B timing of the bench protocol (headline segment only).  usage (through gpurun): bash scripts/gpu_quick.sh [extra pytest files]
set -o pipefail
OUT=gpurun_out/r4
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py "$@" -q -m gpu -x 2>&1 | tail -5 > $OUT/quick_tests.txt
rc=$?
cat $OUT/quick_tests.txt
[ $rc -ne 0 ] && exit $rc
if [ -f bilevel-gait-gen_amd/ab/libtrtri_stamps.so ]; then
    SRBM_LIB=ab/libtrtri_stamps.so timeout -k 10 200 python scripts/dev_trtri.py > $OUT/quick_stamps.txt 2>&1
    grep "n 108 \|^108\|^120" $OUT/quick_stamps.txt | sort
fi
if [ -f bilevel-gait-gen_amd/ab/libchol_stamps.so ]; then
    NS=108,120 SRBM_LIB=ab/libchol_stamps.so timeout -k 10 200 python scripts/dev_chol.py > $OUT/quick_chol_stamps.txt 2>&1
    cat $OUT/quick_chol_stamps.txt | tail -4
fi
for v in $QUICK_VARIANTS; do
    SRBM_LIB=ab/$v timeout -k 10 200 python scripts/dev_trtri.py > $OUT/quick_$v.txt 2>&1
    echo "variant $v"; grep "n 108 \|^108\|^120" $OUT/quick_$v.txt | sort
done
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-reference-criterion --extra-workloads 0 --gait-steps 0 --closed-loop-steps 0 --wbc-ticks 0 > $OUT/quick_bench.json 2> $OUT/quick_bench.err || exit 1
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/quick_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'regions', [round(x, 2) for x in d['region_ms']], 'its', d['config']['solver'].get('mean_ipm_iterations') if isinstance(d['config'].get('solver'), dict) else None)
PY
done
if [ -n "$QUICK_PROF" ]; then
    timeout -k 10 300 python scripts/dev_prof.py > $OUT/quick_prof.txt 2>&1
    grep -A8 "M dense + factor" $OUT/quick_prof.txt; head -16 $OUT/quick_prof.txt
fi
