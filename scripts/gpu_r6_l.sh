#!/bin/bash
# Round 6, call L: the new timeline test and the extended closing-phase test on hardware.
TAG=${1:-r6_l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -s -p no:cacheprovider -k "timeline or one_launch or whole_rounds" 2>&1 | grep -E "B=|max\||passed|failed|Error|error|assert" | tail -30 ) 2>&1 | tee $OUT/pytest_chain_new.txt
