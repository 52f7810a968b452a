#!/bin/bash
# Round 6, call O: the TrajNet suite with the opt-in clip-resident step's tests (reference goldens through it, every clip -> XCD
# split against the launch-per-layer loop, the missing-partner fallback), and the A/B timing record.
TAG=${1:-r6_o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 1200 python -m pytest tests/test_gpu_trajnet.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) 2>&1 | tee $OUT/pytest_trajnet.txt
( timeout 600 python scripts/resident_ab.py 1 8 32 64 2>&1 | tail -90 ) 2>&1 | tee $OUT/resident_ab.txt
