#!/bin/bash
# Round 6, call M: the recorded one-launch steps (graph replay) on hardware; and the whole PoseNet / chain / exchange suites with the closing phase switched OFF
# (the three-launch step must stay green: it is the path of every batch size the stack does not serve in the loops' fallback).
TAG=${1:-r6_m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_chain.py -m gpu -q -s -p no:cacheprovider -k "recorded_into_a_graph_draw" 2>&1 | tail -15 ) 2>&1 | tee $OUT/pytest_graph.txt
( time ROHM_POSENET_STACK_TAIL=0 timeout 2400 python -m pytest tests/test_gpu_posenet.py tests/test_gpu_exchange.py tests/test_gpu_config_batches.py tests/test_gpu_guidance.py tests/test_gpu_scheme.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_tail_off.txt
