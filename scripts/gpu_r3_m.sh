#!/bin/bash
# Round 3: kernel-level tests of the LayerNorm fold (tests/test_gpu_planes.py) + the fold A/B of gpu_r3_l.sh
TAG=${1:-r3_m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_planes.py -x -q -p no:cacheprovider -k "ln_fold or shape_errors" 2>&1 | tail -25 | tee $OUT/pytest_planes_fold.txt
bash scripts/gpu_r3_l.sh $TAG
