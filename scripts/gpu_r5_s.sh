#!/bin/bash
# Round 5, call S: the whole GPU suite + smoke on the HEAD tree (after the LBS change).
TAG=${1:-r5_s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( time timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -18 ) 2>&1 | tee $OUT/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
