#!/bin/bash
# Round 6, call R: the clip-resident TrajNet step as the DEFAULT for TrajNet: the whole GPU suite (default environment), then the suites that
# run TrajNet / TrajControl inside the inference scheme once more with ROHM_TRAJ_RESIDENT=1 (TrajControl resident too), smoke.
TAG=${1:-r6_r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 2700 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 ) 2>&1 | tee $OUT/pytest_gpu.txt
( time ROHM_TRAJ_RESIDENT=1 timeout 1500 python -m pytest tests/test_gpu_scheme.py tests/test_gpu_config_batches.py tests/test_gpu_eval_losses.py tests/test_dropin.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_scheme_all_resident.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
