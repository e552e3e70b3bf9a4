#!/bin/bash
D=gpurun_out/r4
mkdir -p $D
python3 scripts/dev_prof.py > $D/phase_shares_fast.txt 2>&1
PROF_MODE=ref python3 scripts/dev_prof.py > $D/phase_shares_ref.txt 2>&1
(./scripts/ubench/lat; ./scripts/ubench/lat2) > $D/ubench.txt 2>&1
cat $D/ubench.txt; head -18 $D/phase_shares_fast.txt
