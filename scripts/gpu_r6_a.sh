#!/bin/bash
# Round 6, call A: the new config-batch parity tests (stack kernels under reference fixtures at B = 32 / 64) + a headline sanity leg.
TAG=${1:-r6_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_config_batches.py -m gpu -q -s -p no:cacheprovider --durations=12 2>&1 | tail -60 ) 2>&1 | tee $OUT/pytest_config_batches.txt
( time timeout 600 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_headline.json 2> $OUT/bench_headline.err ) 2>&1 | tail -3
tail -c 1500 $OUT/bench_headline.json
