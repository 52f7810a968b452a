#!/bin/bash
# Round 6, call N: first hardware run of the clip-resident TrajNet step (csrc/trajnet_resident.hip): A/B against the launch-per-layer loop
# (difference of the samples + wall time), then the TrajNet parity suite (reference goldens) with the resident step as the default.
TAG=${1:-r6_n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 600 python scripts/resident_ab.py ${AB_BATCHES:-1 2 8 9 32 64} 2>&1 | tail -80 ) 2>&1 | tee $OUT/resident_ab.txt
( timeout 900 python -m pytest tests/test_gpu_trajnet.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) 2>&1 | tee $OUT/pytest_trajnet.txt
